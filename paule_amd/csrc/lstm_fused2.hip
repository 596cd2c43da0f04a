// The forward launch of lstm_fused.hip at TWO workgroups per CU (round 4; `fused_fwd2_kernel`).
//
// Why.  A chain-step of the forward launch keeps its workgroup busy for 3.2 - 3.7 us of which 1.0 us is matrix time (the rest: the h tile
// on its way into LDS, the poll's answer, the memory pipe taking the next tile's requests, cell arithmetic, stores, drain), and with one
// wave per SIMD those phases run one after the other: nothing fills the matrix core while a wave does cell arithmetic, and nothing issues
// loads while it multiplies.  lstm_fused.hip answers with CHAINS (a second group whose hand-off flies meanwhile); cutting the work of one
// workgroup into wave stages that overlap was tried twice and lost to the lock-step of its barriers.  This file lets the
// HARDWARE do the overlapping: the same roles, written to fit 256 registers and 80 KB of LDS, so that TWO workgroups share a CU -- two
// waves per SIMD that belong to different workgroups (different groups, often different roles), each with its own barriers, and the CU's
// schedulers interleave one's MFMA chain with the other's memory and cell phases.
//
// What had to go to get there (the slice of W_hh alone takes 184 of the 256 registers):
//  * the prefetched h tile (48 registers) -- the tile goes global -> LDS by LDS-DMA (`global_load_lds_dwordx4`), no registers, no
//    ds_write phase.  The image is laid out for that: [64-byte column block kb][row][4 x 16 B], one wave instruction fills 16 rows x 64 B
//    = 1 KB of contiguous LDS, and the 16-byte chunk c of row r sits in slot c ^ s(r) so that the MFMA operand reads (ds_read_b128 in the
//    lane groups of gfx950) stay conflict-free -- the permutation is applied on the GLOBAL side (which chunk of its 64-byte piece a lane
//    fetches), so coalescing is untouched;
//  * operands of the NEXT chain-step in flight under this one (one image, one set of rows): the latency this exposes is what the other
//    workgroup on the CU covers.  One chain per workgroup is the normal shape now; two are supported (cell state in LDS);
//  * the bias in 16 registers -> LDS; B fragments three k-steps ahead instead of six.
// Arithmetic, MFMA shapes and k order are those of lstm_fused.hip / the per-layer sweeps: bit-identical results
// (tests/test_hip_parity.py::test_fused_forward_two_per_cu_is_bit_identical).  Hand-off protocol, flags, bounded waits, census: unchanged
// (fused_common.h); the role table is the same, planned for 2 x n_cu workgroup slots (planner.hip: plan_fused).
#include "fused_common.h"

#ifndef FUSED2_STREAMED_W
#define FUSED2_STREAMED_W 4   // W_hh fragments (k-steps) of a recurrence role that are fetched per chain-step instead of staying in registers
#endif
#ifndef FUSED2_PF
#define FUSED2_PF 3           // B fragments read ahead of the MFMA that consumes them (recurrence roles)
#endif
#ifndef FUSED2_AHEAD
#define FUSED2_AHEAD 1        // 1: a two-chain recurrence role requests the OTHER chain's h tile at the end of a chain-step (behind its hand-off store, in front of
                              // the drain and the flag) when that chain's flags were already seen up: the tile lands under the flag, the stash stores and the loop top
#endif
#ifndef FUSED2_X_AT_END
#define FUSED2_X_AT_END 1     // with a tile requested ahead: the write-through input rows are waited for behind the whole recurrent chain instead of half way through it
#endif
#ifndef FUSED2_XCD
#define FUSED2_XCD 0          // 1: the recurrence roles' own exchange through the XCD's L2 is compiled in (PAULE_HIP_FUSED2_XCD=1 then switches it on)
#endif
#ifndef FUSED2_X_LATE_MIN_KS
#define FUSED2_X_LATE_MIN_KS 16   // narrower recurrences stage their input rows in front of the chain: their tile is six pieces, the half-way stop only costs
#endif
#ifndef FUSED2_SPLIT_TILE
#define FUSED2_SPLIT_TILE 1   // 1: the h tile's landing is waited for in two halves around the MFMA chain (measured: DESIGN.md 4.0d)
#endif

namespace pl {
namespace {

constexpr int kFused2MaxChains = 2;

typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2_t o;
    o[0] = (bf16_t)a; o[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, o);
}
__device__ __forceinline__ float bf16_lo_hi(unsigned u, int k) { return __uint_as_float(k ? (u & 0xffff0000u) : (u << 16)); }   // element k of a bf16 pair, widened (exact)

// slot of the 16-byte chunk c of image row r: LSTM roles read B fragments as (row lane & 31, chunk 2 (ks & 1) + (lane >> 5)), product roles
// read A fragments as (row lane & 15 (+ 16), chunk lane >> 4) -- each needs its own permutation to keep the 16 lanes of a ds_read_b128
// group on 16 different bank quads (worked through in DESIGN.md 4.0d)
template <bool GEMM>
__device__ __forceinline__ int img_swz(int row) {
    const int q = (row >> 2) & 3;
    return GEMM ? ((0x78 >> (2 * q)) & 3) : q;
}

// rows 32 g .. 32 g + 31 of a row-major [Bp][Hp] bf16 slab -> image, by LDS-DMA (write-through reads).  Wave w fills the half tiles
// (kb = 2 j + (w >> 1), rows 16 (w & 1) ...): P tiles x 2 halves, 12 or 11 pieces a wave at Hp = 736.  Rows beyond the batch read row
// Bp - 1 (their results are never stored: every store's offset goes through the range check).
// N pieces of one wave in ONE asm statement: piece j reads 128 bytes further along the rows (the instruction's immediate offset, which moves
// the global AND the LDS address) and lands 4096 bytes further in the image (M0 moves by 4096 - 128 between pieces).  Piece by piece through
// glds16_sc1, each piece paid a 64-bit address (two v_readlane of the spilled slab pointer, s_add_u32, s_addc_u32), its LDS address and a
// save / restore of M0: ten scalar instructions beside a DMA whose own issue costs about as much (round 5).
#define PL_GLDS_FIRST "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\t"
#define PL_GLDS_NEXT(OFF) "s_add_u32 m0, m0, 3968\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:" #OFF " sc1\n\t"
#define PL_GLDS_LAST "s_mov_b32 m0, %0"
#define PL_GLDS_ARGS : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory", "scc"
template <int N>
__device__ __forceinline__ void dma_pieces_sc1(const unsigned char* src, unsigned voff, unsigned dst) {
    unsigned keep;
    static_assert(N == 1 || N == 2 || N == 3 || N == 11 || N == 12, "piece counts of P = 3, 6, 23");
    if constexpr (N == 1) asm volatile(PL_GLDS_FIRST PL_GLDS_LAST PL_GLDS_ARGS);
    if constexpr (N == 2) asm volatile(PL_GLDS_FIRST PL_GLDS_NEXT(128) PL_GLDS_LAST PL_GLDS_ARGS);
    if constexpr (N == 3) asm volatile(PL_GLDS_FIRST PL_GLDS_NEXT(128) PL_GLDS_NEXT(256) PL_GLDS_LAST PL_GLDS_ARGS);
    if constexpr (N == 11)
        asm volatile(PL_GLDS_FIRST PL_GLDS_NEXT(128) PL_GLDS_NEXT(256) PL_GLDS_NEXT(384) PL_GLDS_NEXT(512) PL_GLDS_NEXT(640) PL_GLDS_NEXT(768) PL_GLDS_NEXT(896)
                     PL_GLDS_NEXT(1024) PL_GLDS_NEXT(1152) PL_GLDS_NEXT(1280) PL_GLDS_LAST PL_GLDS_ARGS);
    if constexpr (N == 12)
        asm volatile(PL_GLDS_FIRST PL_GLDS_NEXT(128) PL_GLDS_NEXT(256) PL_GLDS_NEXT(384) PL_GLDS_NEXT(512) PL_GLDS_NEXT(640) PL_GLDS_NEXT(768) PL_GLDS_NEXT(896)
                     PL_GLDS_NEXT(1024) PL_GLDS_NEXT(1152) PL_GLDS_NEXT(1280) PL_GLDS_NEXT(1408) PL_GLDS_LAST PL_GLDS_ARGS);
}

template <int P, int ROWB, bool GEMM>
__device__ __forceinline__ void dma_image(const void* slab, int g, int Bp, unsigned img_lds, int wave, int lane) {
    const int half = wave & 1, hi = wave >> 1;
    const int row = 16 * half + (lane >> 2);
    int rb = 32 * g + row;
    rb = rb < Bp ? rb : Bp - 1;
    const unsigned voff = (unsigned)(rb * ROWB + (((lane & 3) ^ img_swz<GEMM>(row)) * 16));
    const unsigned char* src = uni(static_cast<const unsigned char*>(slab) + hi * 64);
    const unsigned dst = (unsigned)uni((int)(img_lds + (unsigned)(hi * 2048 + half * 1024)));
    if (hi == 0) dma_pieces_sc1<(P + 1) / 2>(src, voff, dst);   // tiles kb = 2 j + hi < P
    else dma_pieces_sc1<P / 2>(src, voff, dst);
}

// The same image from the role's PRIVATE copy of the hand-off (round 5): tile-major [slice kb][32 rows][64 B] -- a slice is 2 KB contiguous,
// 16 whole lines stored plainly by the one workgroup that owns it -- fetched through the XCD's L2 (nt).  One wave instruction = one
// contiguous 1-KB piece; the chunk permutation of the image sits on the global side as before.
template <int P, bool GEMM>
__device__ __forceinline__ void dma_image_tiles(const unsigned char* tiles, unsigned img_lds, int wave, int lane) {
    const int half = wave & 1, hi = wave >> 1;
    const int row = 16 * half + (lane >> 2);
    const unsigned voff = (unsigned)(row * 64 + (((lane & 3) ^ img_swz<GEMM>(row)) * 16));
    const unsigned char* src = tiles + hi * 2048;
    const unsigned dst = img_lds + (unsigned)(hi * 2048 + half * 1024);
#pragma unroll
    for (int j = 0; j < (P + 1) / 2; ++j)
        if (2 * j + hi < P) glds16_nt(uni(src + j * 4096), voff, (unsigned)uni((int)(dst + (unsigned)(j * 4096))));
}

// ---------------------------------------------------------------------------------------------------------------------
// forward recurrence of one layer (arithmetic of lstm_fwd_sweep_kernel / fused_lstm_fwd)
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int KSX>
struct LstmFwd2Lds {
    static constexpr int Hp = 16 * KS, P = Hp / 32;
    static constexpr int HRS = 64 + 16;           // outgoing tiles [32 rows][32 units] bf16
    static constexpr int XRS = KSX * 32 + 16;
    static constexpr int GRS = 4 * 64 + 16;       // KSX = 0: the workgroup's projection rows [32 rows][4 gates x 32 units] bf16
    static constexpr int O_HIMG = 0;
    static constexpr int O_HST = O_HIMG + P * 2048;
    static constexpr int O_XIMG = O_HST + 6 * 32 * HRS;
    static constexpr int O_CST = O_XIMG + (KSX ? 32 * XRS : 32 * GRS);
    static constexpr int O_BIAS = O_CST + kFused2MaxChains * 256 * 16;
    static constexpr int O_FLAG = O_BIAS + 512;
    static constexpr int BYTES = O_FLAG + 64;
};

// SC1: the role's input rows (x / G) are written by a role of this launch: write-through loads, issued behind the flag wait.  A compile-time
// fact of the instantiation, because the waits differ: rows that nobody in the launch writes are fetched in front of the flag wait and have
// landed before the tile's pieces go out, and the compiler's s_waitcnt for a load still in flight on ONE path of a run-time choice would sit
// on both (SC1 = true is correct for any role, only slower for rows that could have come early).
// AHEAD (with FUSED2_AHEAD): the other chain's h tile is requested at the end of a chain-step (see there).  On in the role-fused launch
// (cfg3 4.19 -> 4.14 ms, cfg3_setB 4.14 -> 4.05), off in the per-layer sweep kernel, where it measured 0.4 % slower (2048 rows).
template <int KS, int KSX, bool SC1, bool AHEAD>
__device__ __forceinline__ void fused_lstm_fwd2(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmFwd2Lds<KS, KSX>;
    static_assert(KS % 2 == 0, "whole 32-unit tiles");
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, P = Hp / 32;
    constexpr int ROWB = Hp * 2, HRS = L::HRS, XRS = L::XRS;
    constexpr int PF = FUSED2_PF;                     // B-fragment read-ahead
    constexpr int NP1 = FUSED2_SPLIT_TILE ? (P + 3) / 4 : (P + 1) / 2;   // pieces per wave in the first half of the tile (tiles 0 .. 2 NP1 - 1)
    constexpr int KH = FUSED2_SPLIT_TILE ? 4 * NP1 : KS;                 // ... = k-steps 0 .. KH - 1
    constexpr int PK = KS / 2;                        // k-step of the look at the next chain-step's flags (two chains)
    constexpr int INP = KSX ? 16 * KSX : 16, XC = INP / 8;
    unsigned char* himg = lds + L::O_HIMG;
    unsigned char* hst = lds + L::O_HST;
    unsigned char* ximg = lds + L::O_XIMG;
    float4* cst = reinterpret_cast<float4*>(lds + L::O_CST);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);
    const unsigned himg_lds = (unsigned)(uintptr_t)(lds_ptr_t)himg;

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(R.W);

    // weights -> registers: A-operand row (lane & 31) = gate (row >> 3), unit 32p + 8 wave + (row & 7).  All but the first NWT k-steps'
    // fragments stay resident; those NWT are fetched again every chain-step (from L2, in front of the flag wait, into registers the cell
    // update uses later): resident, the compiler spilled exactly these to scratch and reloaded each in front of its MFMA behind an
    // s_waitcnt vmcnt(0) -- which also waited for the second half of the tile and for the last chain-step's stash stores
    constexpr int NWT = KS >= 16 ? FUSED2_STREAMED_W : 0;
    uint4 wreg[KS];
    const unsigned woff = (unsigned)((((lane & 31) >> 3) * Hp + 32 * p + 8 * wave + (lane & 7)) * Hp + 8 * (lane >> 5)) * 2u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W, (unsigned)((size_t)G4 * Hp * 2));
    {
        const int ar = lane & 31;
        const bf16_t* wrow = W + (size_t)((ar >> 3) * Hp + 32 * p + 8 * wave + (ar & 7)) * Hp + 8 * (lane >> 5);
#pragma unroll
        for (int ks = NWT; ks < KS; ++ks) wreg[ks] = gld<uint4>(wrow + 16 * ks);
#pragma unroll
        for (int ks = NWT; ks < KS; ++ks) pin(wreg[ks]);   // pinned after ALL loads are out: load + pin in one loop waited for every load by itself
    }
    // the input projection's weight fragments are fetched per chain-step as well, half way through the MFMA chain (they multiply last):
    // they take the registers the streamed W_hh fragments have left by then
    uint4 wx[KSX ? KSX : 1];
    const unsigned wxoff = (unsigned)((((lane & 31) >> 3) * Hp + 32 * p + 8 * wave + (lane & 7)) * INP + 8 * (lane >> 5)) * 2u;
    const __amdgpu_buffer_rsrc_t rwx = make_rsrc(KSX ? R.Wih : R.W, (unsigned)((size_t)G4 * INP * 2));
    auto fetch_wx = [&]() {
        if constexpr (KSX > 0) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rwx, wxoff + 32u * ks, 0, 0);
                wx[ks] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    if constexpr (KSX > 0) {
        // bias of accumulator element r of (wave w, lane half h): [w][h][r] in LDS, read back as four float4 per chain-step
        if (wave < 2) {
            const int w2 = tid >> 5, h2 = (tid >> 4) & 1, r = tid & 15;
            reinterpret_cast<float*>(lds + L::O_BIAS)[tid] = gld<float>(R.bias + (r >> 2) * Hp + 32 * p + 8 * w2 + 4 * h2 + (r & 3));
        }
    }
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G);
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(R.h);
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(R.c);
    constexpr bool src_sc1 = SC1;   // x / G rows come from a role of this launch: write-through loads
    const bf16_t* const x_in = static_cast<const bf16_t*>(R.x_in);
    // The role's OWN exchange through the XCD's L2 (round 5; tools/microbench/allgather_step.hip: 5.9 -> 4.5 us a step at two workgroups per CU):
    // written through, the h slices drop out of the L2 and all 23 workgroups of a set fetch the same 47 KB from the memory side, 64-byte half
    // lines at a time.  Every workgroup ALSO stores its slice plainly into a private tile-major copy (R.hx: [2 slots][groups][P][32 rows][64 B],
    // whole lines) and raises a second, plain flag (R.fast_flags); it publishes its XCD with its first hand-off, the members of a set compare at
    // chain-step (c = 0, t = 1) -- behind their first completed wait on each other -- and a set that is whole on one XCD reads the private copy
    // (nt LDS-DMA), polls the plain flags (nt), and moves the write-through h store and flag, which only OTHER roles read, out of its own way:
    // behind the plain flag, the write-through flag one chain-step late (a wave's stores complete in order: the next drain covers them).
    // Placement is never assumed, the bits do not change.
    unsigned char* const hx = FUSED2_XCD ? static_cast<unsigned char*>(R.hx) : nullptr;   // (compiled out by default: see FUSED2_XCD)
    int* const xtab = (hx && R.xtab) ? R.xtab + set * 64 : nullptr;
    const long fdelta = (xtab && R.fast_flags) ? (long)(R.fast_flags - R.flags) : 0;
    const size_t hx_grp = (size_t)P * 2048, hx_slot = (size_t)a.n_groups * hx_grp;
    bool fast = false;
    int wt_g = -1, wt_t = -1;   // fast: the write-through flag of the chain-step before is still to be raised
    // B fragment of k-step ks: tile ks >> 1, row bl, chunk 2 (ks & 1) + hh (addresses: per chain-step, from the opaque lane index -- held across
    // the loop they were spilled, and the reload in front of the first MFMA carried an s_waitcnt vmcnt(0) that waited for the WHOLE tile)

    // the five stash arrays of a chain-step (gates i f g o, c; nobody inside the launch waits for them) leave LDS BEHIND the flag: issued
    // in front of it they sat on the group's critical path (0.9 us a chain-step; behind the next chain-step's tile requests they delayed
    // the tile's landing by as much)
    auto stash_stores = [&](const int g, const int t, const int tid) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + 256 * i;   // piece: array e / 128, row (e % 128) / 4, quarter e % 4
        if (i < 2 || wave < 2) {   // 640 pieces: all threads twice, waves 0 and 1 a third time
            const int arr = e >> 7, row = (e & 127) >> 2, qt = e & 3, rb = 32 * g + row;
            const uint4 sv = *reinterpret_cast<const uint4*>(hst + (arr + 1) * 32 * HRS + row * HRS + qt * 16);
            u32x4 d;
            d[0] = sv.x; d[1] = sv.y; d[2] = sv.z; d[3] = sv.w;
            if (i < 2) {   // pieces 0 .. 511: the four gate arrays (128 pieces = 2 waves each); 512 .. 639: c
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
                __builtin_amdgcn_raw_buffer_store_b128(d, rg, rb < Bp ? (unsigned)(((size_t)rb * G4 + arr * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cs + (size_t)t * slabH, (unsigned)(slabH * 2));
                __builtin_amdgcn_raw_buffer_store_b128(d, rc, rb < Bp ? (unsigned)(((size_t)rb * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
            }
        }
    }
    };
    PL_ST_DECL
    int c = 0, t = 0;
    bool ready = false;   // the flags of the coming chain-step were seen up during the last one
    bool ahead = false;   // ... and its h tile was requested at the end of the last one (FUSED2_AHEAD): the pieces are in flight, older than everything below
    for (;;) {
        // per-lane_q indices of this chain-step, opaque to the optimizer: derived from plain `tid` every address of the x loads, the staging
        // image, the stores ... is loop-invariant (one chain), gets hoisted in front of the loop and held in registers the kernel does not
        // have -- the weight fragments paid for that in scratch.  Recomputing them per chain-step is a few dozen VALU operations.
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane_q = tq & 63, bl_q = lane_q & 31, hh_q = lane_q >> 5;
        const unsigned char* const bsrc0 = himg + bl_q * 64 + ((hh_q ^ img_swz<false>(bl_q)) * 16);
        const unsigned char* const bsrc1 = himg + bl_q * 64 + (((2 + hh_q) ^ img_swz<false>(bl_q)) * 16);
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t + 1; }
        const bool has_next = tn < T;
        const int gn = set * RC + cn;

        // the streamed weight fragments of this chain-step (they do not depend on anybody's flag)
#pragma unroll
        for (int ks = 0; ks < NWT; ++ks) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, woff + 32u * ks, 0, 0);
            wreg[ks] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        // Input rows that no role of this launch writes (the predictor's CP frames) do not depend on anybody's flag either: fetched here, they
        // have landed by the time the flags are up
        uint4 xv = make_uint4(0, 0, 0, 0);
        constexpr bool kXEarly = KSX > 0 && !SC1;
        if constexpr (KSX > 0) {
            if (!src_sc1 && wave < XC / 2) {   // 32 x XC threads = XC / 2 whole waves: a scalar branch
                int rb = 32 * g + tq / XC;
                rb = rb < Bp ? rb : Bp - 1;
                xv = gld<uint4>(x_in + ((size_t)t * Bp + rb) * INP + (tq % XC) * 8);
            }
        }
        // 0. what this chain-step waits for (bounded; the barrier inside also closes the last chain-step's LDS reads)
        if (!ready) {
            FlagPoll s0 = step_flags(a, WT_, g, t, p);
            if (fast && s0.na > 0) { s0.fa += fdelta; s0.fa_nt = 1; }   // the set's own flags of step t - 1: the plain set
            if (!poll_empty(s0) && !flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        }
        PL_ST(0);   // wait for the chain-step's flags

        // A. operands: the small x / projection rows into registers, the h tile straight into LDS
        uint2 gxn[4] = {};
        uint4 gv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if constexpr (KSX > 0) {
            if (src_sc1 && wave < XC / 2) {
                int rb = 32 * g + tq / XC;
                rb = rb < Bp ? rb : Bp - 1;
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(x_in + (size_t)t * Bp * INP, (unsigned)((size_t)Bp * INP * 2));
                xv = ld16_sc1(rx, (unsigned)((rb * INP + (tq % XC) * 8) * 2));
            }
        } else {
            if (src_sc1) {
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = tq + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                    const int rb = 32 * g + row;
                    gv[q] = ld16_sc1(rg, rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2) : kOob);
                }
            } else {
                int b2 = 32 * g + bl_q;
                b2 = b2 < Bp ? b2 : Bp - 1;
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)b2 * G4 + (32 * p + 8 * wave + 4 * hh_q);
#pragma unroll
                for (int q = 0; q < 4; ++q) gxn[q] = gld<uint2>(g_row + q * Hp);
            }
        }
        // the input rows go to their LDS image where waiting for them costs nothing: rows fetched in front of the flag wait at once, rows a role
        // of this launch wrote (issued just now, in front of the tile's pieces) behind the SECOND half's landing wait -- the x-projection
        // multiplies last, and the cell update reads the projection rows after the chain
        auto stage_x = [&]() {
            if constexpr (KSX > 0) {
                if (wave < XC / 2) *reinterpret_cast<uint4*>(ximg + (tq / XC) * XRS + (tq % XC) * 16) = xv;
            } else {
                if (src_sc1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = tq + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                        *reinterpret_cast<uint4*>(ximg + row * L::GRS + gate * 64 + q4 * 16) = gv[q];
                    }
                }
            }
        };
        auto read_gx = [&]() {
            if constexpr (KSX == 0) {
                if (src_sc1) {
                    const unsigned char* gsrc = ximg + bl_q * L::GRS + (8 * wave + 4 * hh_q) * 2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) gxn[q] = *reinterpret_cast<const uint2*>(gsrc + q * 64);
                }
            }
        };
        const bool x_late = t > 0 && src_sc1 && FUSED2_SPLIT_TILE && KS >= FUSED2_X_LATE_MIN_KS;   // staged at the half-way point of the MFMA chain
        // Everything the compiler itself has in flight is waited for HERE, in front of the tile's pieces (on every path: a wait the compiler
        // still owes on ONE path comes out as s_waitcnt vmcnt(0) behind the merge): it does not count the pieces (inline asm), so its wait for
        // any load it issued earlier came out as s_waitcnt vmcnt(0) somewhere between the first-half wait and the first MFMA -- which waited
        // for the WHOLE tile, and the split landing never overlapped anything (round 4).  The streamed weight fragments and the
        // early input rows were fetched in front of the flag wait: they have landed.
#pragma unroll
        for (int ks = 0; ks < NWT; ++ks) asm volatile("" ::"v"(wreg[ks].x), "v"(wreg[ks].y), "v"(wreg[ks].z), "v"(wreg[ks].w));
        if constexpr (kXEarly) asm volatile("" ::"v"(xv.x), "v"(xv.y), "v"(xv.z), "v"(xv.w));
        // B. with a second chain: a look at ITS flags from inside the MFMA chain (with one chain the next chain-step waits for the
        // flag this one raises at its end)
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr, 0};
        const bool look = has_next && Ca > 1;
        if (look) pn = step_flags(a, WT_, gn, tn, p);
        if (look && fast && pn.na > 0) { pn.fa += fdelta; pn.fa_nt = 1; }
        int pv = 1;
        const bool poll_here = wave == 0 && look;
        // C. gates = W_hh h_{t-1} (+ W_ih x_t + b)
        f32x16 acc;
        auto acc_init = [&]() {
            if constexpr (KSX > 0) {
                const float4* bs = reinterpret_cast<const float4*>(lds + L::O_BIAS) + (wave * 2 + hh_q) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 bv = bs[i];
                    acc[4 * i] = bv.x; acc[4 * i + 1] = bv.y; acc[4 * i + 2] = bv.z; acc[4 * i + 3] = bv.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
        };
        // (ONE branch on t > 0 for operands and chain together: with two, the loads of the t = 0 path -- x-projection fragments, the look's
        // poll -- were still owed at the merge in front of the chain, and the compiler put its s_waitcnt vmcnt(0) in front of the first MFMA)
        if (t > 0) {
            if (ahead) {
                // requested a chain-step ago: the pieces are older than this chain-step's streamed fragments and write-through input rows (NWT
                // + NSC of them at most, on every wave) -- all but those have completed means the whole tile has landed
                constexpr int NSC = (KSX == 0 && SC1) ? 2 : 0;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWT + NSC) : "memory");
            } else {
            if (fast) dma_image_tiles<P, false>(hx + (size_t)((t - 1) & 1) * hx_slot + (size_t)g * hx_grp, himg_lds, wave, lane_q);
            else dma_image<P, ROWB, false>(Hs + (size_t)(t - 1) * slabH, g, Bp, himg_lds, wave, lane_q);
            // the tile in two halves: the first KH k-steps' pieces (a wave's first NP1, and everything older) have landed when all but
            // its youngest pieces have; the second half lands under the first half's MFMAs
            if (!FUSED2_SPLIT_TILE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P + 1) / 2 - NP1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P / 2 - NP1 > 0 ? P / 2 - NP1 : 0) : "memory");
            }
            if (!x_late) stage_x();
            __syncthreads();
            if (!x_late) read_gx();
            PL_ST(1);   // operands: issue -> landed -> barrier
            acc_init();
            uint4 bq[PF];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int K0[2] = {0, KH}, K1[2] = {KH, KS};
                if (half == 1 && K0[1] < K1[1] && !(FUSED2_X_AT_END && ahead)) {   // the second half of the tile (and, with it, input rows that a role of this launch wrote)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (x_late) stage_x();
                    __syncthreads();
                    if (x_late) read_gx();
                }
                if (half == 1) fetch_wx();
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (K0[half] + i < K1[half]) bq[i] = *reinterpret_cast<const uint4*>((((K0[half] + i) & 1) ? bsrc1 : bsrc0) + ((K0[half] + i) >> 1) * 2048);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = K0[half]; ks < K1[half]; ++ks) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks]), __builtin_bit_cast(bf16x8, bq[(ks - K0[half]) % PF]), acc, 0, 0, 0);
                    if (ks + PF < K1[half])
                        bq[(ks - K0[half]) % PF] = *reinterpret_cast<const uint4*>((((ks + PF) & 1) ? bsrc1 : bsrc0) + ((ks + PF) >> 1) * 2048);
                    if (ks == PK && poll_here) pv = poll_load(pn, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (FUSED2_X_AT_END && ahead && x_late) {   // the tile was there from the start (requested a chain-step ago): no half-way stop; the write-through
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // input rows -- issued at this chain-step's top -- are staged behind the whole recurrent chain
                stage_x();
                __syncthreads();
                read_gx();
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage_x();
            __syncthreads();
            read_gx();
            PL_ST(1);
            acc_init();
            fetch_wx();
            if (poll_here) pv = poll_load(pn, lane);
        }
        if constexpr (KSX > 0) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                const uint4 xb = *reinterpret_cast<const uint4*>(ximg + bl_q * XRS + ks * 32 + hh_q * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wx[ks]), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
            }
        }
        PL_ST(2);   // MFMA chain
        if (wave == 0) {   // the look's answer, handed to everybody behind the barrier of the output staging
            const bool rdy = look && __all(pv != 0);
            if (lane_q == 0) lflag[0] = rdy ? 1 : 0;
        }

        // F. cell update: acc[4 * gate + unit], two units at a time (the six outputs of a pair leave for the staging image before the next
        // pair is touched: all four at once kept 24 outputs + 16 widened inputs alive on top of the accumulators, and the weight
        // fragments paid for it in scratch)
        // G. the h tile (hand-off) and the five stash arrays leave through LDS as whole 64-byte row pieces
        {
            float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t > 0) cs = cst[c * 256 + tq];
            float c_state[4] = {cs.x, cs.y, cs.z, cs.w};
            unsigned char* o = hst + bl_q * HRS + (8 * wave + 4 * hh_q) * 2;
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const unsigned xi = u2 ? gxn[0].y : gxn[0].x, xf = u2 ? gxn[1].y : gxn[1].x, xg = u2 ? gxn[2].y : gxn[2].x, xo = u2 ? gxn[3].y : gxn[3].x;
                float vi[2], vf[2], vg[2], vo[2], vh[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int u = 2 * u2 + k;
                    vi[k] = sigmoid_fast(acc[u] + bf16_lo_hi(xi, k));
                    vf[k] = sigmoid_fast(acc[4 + u] + bf16_lo_hi(xf, k));
                    vg[k] = tanh_fast(acc[8 + u] + bf16_lo_hi(xg, k));
                    vo[k] = sigmoid_fast(acc[12 + u] + bf16_lo_hi(xo, k));
                    c_state[u] = cell_c(vf[k], c_state[u], vi[k], vg[k]);
                    vh[k] = vo[k] * tanh_fast(c_state[u]);
                }
                *reinterpret_cast<unsigned*>(o + 4 * u2) = pack_bf16x2(vh[0], vh[1]);
                *reinterpret_cast<unsigned*>(o + 4 * u2 + 32 * HRS) = pack_bf16x2(vi[0], vi[1]);
                *reinterpret_cast<unsigned*>(o + 4 * u2 + 2 * 32 * HRS) = pack_bf16x2(vf[0], vf[1]);
                *reinterpret_cast<unsigned*>(o + 4 * u2 + 3 * 32 * HRS) = pack_bf16x2(vg[0], vg[1]);
                *reinterpret_cast<unsigned*>(o + 4 * u2 + 4 * 32 * HRS) = pack_bf16x2(vo[0], vo[1]);
                *reinterpret_cast<unsigned*>(o + 4 * u2 + 5 * 32 * HRS) = pack_bf16x2(c_state[2 * u2], c_state[2 * u2 + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            cst[c * 256 + tq] = make_float4(c_state[0], c_state[1], c_state[2], c_state[3]);
        }
        if (xtab && c == 0 && t == 1 && wave == 1) {   // every member's XCD id is in place (stored and drained in front of its first flag)
            const int mine = xcc_id_plus1();
            int v = mine;
            if (lane_q < P) v = flag_load(xtab + lane_q);
            const bool same = __all(v == mine);
            if (lane_q == 0) lflag[2] = same ? 1 : 0;
        }
        __syncthreads();
        PL_ST(3);   // cell update, staging, barrier
        ready = lflag[0] != 0;
        if (xtab && c == 0 && t == 1) fast = uni(lflag[2]) != 0;
        int* const wflag = rflags + ((size_t)g * T + t) * a.flag_stride + p;
        if (wave < 2) {
            const int row = tq >> 2, qt = tq & 3;
            const int rb = 32 * g + row;
            const uint4 hvv = *reinterpret_cast<const uint4*>(hst + row * HRS + qt * 16);
            if (hx) {   // the private tile-major copy, plain: what the set itself reads once it has found itself on one XCD
                u32x4 d;
                d[0] = hvv.x; d[1] = hvv.y; d[2] = hvv.z; d[3] = hvv.w;
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(hx + (size_t)(t & 1) * hx_slot + (size_t)g * hx_grp + (size_t)p * 2048, 2048u);
                __builtin_amdgcn_raw_buffer_store_b128(d, rx, (unsigned)(row * 64 + qt * 16), 0, 0);
            }
            if (!fast) {
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
                st16_sc1(ro, rb < Bp ? (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2) : kOob, hvv);
            }
        }
        if (xtab && c == 0 && t == 0 && tq == 0) flag_store(xtab + p, xcc_id_plus1());   // with the first hand-off: drained before its flag
        PL_ST(4);   // hand-off store issue
        // FUSED2_AHEAD: the other chain's flags were seen up during this chain-step's MFMAs (ready), and behind the barrier above nobody reads the
        // h image any more: its next tile goes out NOW, behind this chain-step's hand-off store -- the drain in front of the flag waits for
        // everything but the pieces (a wave's memory operations complete in order), and the tile lands under the flag, the stash stores and
        // the next chain-step's top instead of in front of its MFMAs
        const bool ahead_next = FUSED2_AHEAD && AHEAD && ready && has_next && tn > 0 && !fast;
        if (ahead_next) dma_image<P, ROWB, false>(Hs + (size_t)(tn - 1) * slabH, gn, Bp, himg_lds, wave, lane_q);
        ahead = ahead_next;
        if (!fast) {
            if (ahead_next) raise_flag<(P + 1) / 2>(wflag);   // waves 0 and 1 (the storing ones) requested (P + 1) / 2 pieces each; waves 2, 3 one fewer and stored nothing
            else raise_flag<0>(wflag);   // only the hand-off is in flight: the stash stores follow the flag
            if (fdelta != 0 && wave == 0 && lane_q == 0) flag_store_plain(wflag + fdelta, 1);   // the plain set holds every step: the set may switch to it
        } else {
            // drained here: this step's private slice and everything older -- the write-through h and stash stores of the chain-step before
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 0 && lane_q == 0) {
                flag_store_plain(wflag + fdelta, 1);
                if (wt_g >= 0) flag_store(rflags + ((size_t)wt_g * T + wt_t) * a.flag_stride + p, 1);
            }
            wt_g = g; wt_t = t;
            if (wave < 2) {   // the write-through copy for the roles that read this layer's h (mel head, projection of the layer above)
                const int row = tq >> 2, qt = tq & 3;
                const int rb = 32 * g + row;
                const uint4 hvv = *reinterpret_cast<const uint4*>(hst + row * HRS + qt * 16);
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
                st16_sc1(ro, rb < Bp ? (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2) : kOob, hvv);
            }
        }
        PL_ST(5);   // drain + barrier + flag
        stash_stores(g, t, tq);
        PL_ST(6);   // stash store issue (behind the flag: they drain under the next wait)
        if (!has_next) break;
        c = cn;
        t = tn;
    }
    if (wt_g >= 0) raise_flag<0>(rflags + ((size_t)wt_g * T + wt_t) * a.flag_stride + p);   // the last chain-step's write-through flag
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// row-tile products on a producing layer's h (arithmetic of gemm_nt_kernel / fused_gemm_fwd): the input projection of the
// layer above (G_t = h_t Wih^T + b, bf16) and the mel head with its pooling
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct GemmFwd2Lds {
    static constexpr int Hp = 16 * KS, P = Hp / 32;
    static constexpr int ORS = 128 * 2 + 16;
    static constexpr int O_IMG = 0;
    static constexpr int O_OST = O_IMG + P * 2048;
    static constexpr int O_YB = O_OST + 32 * ORS;
    static constexpr int O_FLAG = O_YB + kFused2MaxChains * 256 * 32;
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, bool HEAD>
__device__ __forceinline__ void fused_gemm_fwd2(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = GemmFwd2Lds<KS>;
    static_assert(KS % 2 == 0, "whole 32-unit tiles");
    constexpr int Hp = 16 * KS, KB = KS / 2, P = Hp / 32, ORS = L::ORS, ROWB = Hp * 2;
    constexpr int NJ = HEAD ? 1 : 2;
    constexpr int PF = 2;
    unsigned char* img = lds + L::O_IMG;
    unsigned char* ost = lds + L::O_OST;
    float4* yb = reinterpret_cast<float4*>(lds + L::O_YB);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);
    const unsigned img_lds = (unsigned)(uintptr_t)(lds_ptr_t)img;

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const int G4 = 4 * Hp;   // PROJ: gate columns of the consuming layer (same hidden size)
    const bf16_t* __restrict__ Wg = static_cast<const bf16_t*>(R.Wg);
    uint4 wreg[NJ][KB];
    float bias_v[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int col = HEAD ? 16 * wave + lr : wave * Hp + 32 * p + 16 * jj + lr;
        const bf16_t* wrow = Wg + (size_t)col * Hp + 8 * kq;
#pragma unroll
        for (int n = 0; n < KB; ++n) wreg[jj][n] = gld<uint4>(wrow + 32 * n);
        bias_v[jj] = R.bias ? gld<float>(R.bias + col) : 0.f;
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int n = 0; n < KB; ++n) pin(wreg[jj][n]);
    const size_t slabH = (size_t)Bp * Hp;
    const bf16_t* __restrict__ Hsrc = static_cast<const bf16_t*>(R.src_h);
    const int out_dim = R.out_dim;
    float* const out_bm = R.out_bm;
    void* const out_ptr = R.out;
    // A fragment of k-step n (32 columns = tile n): rows lr and 16 + lr (same permutation: (row >> 2) & 3 agrees), chunk kq
    const unsigned char* const a0 = img + lr * 64 + ((kq ^ img_swz<true>(lr)) * 16);
    const unsigned char* const a1 = a0 + 16 * 64;

    int c = 0, t = 0;
    bool ready = false;
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t + 1; }
        const bool has_next = tn < T;
        const int gn = set * RC + cn;

        if (!ready) {
            const FlagPoll s0 = step_flags(a, WT_, g, t, p);
            if (!flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        }
        dma_image<P, ROWB, true>(Hsrc + (size_t)t * slabH, g, Bp, img_lds, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr, 0};
        if (has_next) pn = step_flags(a, WT_, gn, tn, p);
        int pv = 1;
        const bool poll_here = wave == 0 && has_next;
        __builtin_amdgcn_sched_barrier(0);

        f32x4 acc[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            uint4 f0[PF], f1[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < KB) {
                    f0[i] = *reinterpret_cast<const uint4*>(a0 + i * 2048);
                    f1[i] = *reinterpret_cast<const uint4*>(a1 + i * 2048);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < KB; ++n) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[n % PF]), __builtin_bit_cast(bf16x8, wreg[jj][n]), acc[0][jj], 0, 0, 0);
                    acc[1][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[n % PF]), __builtin_bit_cast(bf16x8, wreg[jj][n]), acc[1][jj], 0, 0, 0);
                }
                if (n + PF < KB) {
                    f0[n % PF] = *reinterpret_cast<const uint4*>(a0 + (n + PF) * 2048);
                    f1[n % PF] = *reinterpret_cast<const uint4*>(a1 + (n + PF) * 2048);
                }
                if (n == KB / 2 && poll_here) pv = poll_load(pn, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wave == 0) {   // the answer is read behind the barrier of the output staging
            const bool rdy = has_next && __all(pv != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }

        // epilogue: D[row 16 i + 4 kq + r][column 16 jj + lr (of this wave's columns)]
        if constexpr (!HEAD) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<bf16_t*>(ost + (16 * i + 4 * kq + r) * ORS + (32 * wave + 16 * jj + lr) * 2) = (bf16_t)(acc[i][jj][r] + bias_v[jj]);
            __syncthreads();
            ready = lflag[0] != 0;
            bf16_t* Gout = static_cast<bf16_t*>(out_ptr);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(Gout + (size_t)t * Bp * G4, (unsigned)((size_t)Bp * G4 * 2));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                const int rb = 32 * g + row;
                const uint4 v = *reinterpret_cast<const uint4*>(ost + row * ORS + (32 * gate + 8 * q4) * 2);
                st16_sc1(ro, rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2) : kOob, v);
            }
            raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);
        } else {
            float y[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[4 * i + r] = acc[i][0][r] + bias_v[0];
            if ((t & 1) == 0) {   // even frame: kept for its partner
                yb[(c * 256 + tid) * 2] = make_float4(y[0], y[1], y[2], y[3]);
                yb[(c * 256 + tid) * 2 + 1] = make_float4(y[4], y[5], y[6], y[7]);
                __syncthreads();   // the image is rewritten at the top of the next chain-step
                ready = lflag[0] != 0;
            } else {
                const float4 e0 = yb[(c * 256 + tid) * 2], e1 = yb[(c * 256 + tid) * 2 + 1];
                const float ye[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                const int tp = t >> 1, Tp = T >> 1, col = 16 * wave + lr;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * i + 4 * kq + r, bb = 32 * g + row;
                        const bool live = bb < a.B && col < out_dim;
                        const float v = live ? 0.5f * (ye[4 * i + r] + y[4 * i + r]) : 0.f;
                        const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(out_bm, (unsigned)((size_t)a.B * Tp * out_dim * 4));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rb_, live ? (unsigned)((((size_t)bb * Tp + tp) * out_dim + col) * 4) : kOob, 0, 0);
                        *reinterpret_cast<bf16_t*>(ost + row * ORS + col * 2) = (bf16_t)v;
                    }
                __syncthreads();
                ready = lflag[0] != 0;
                {   // pooled frame, time-major activation [tp][Bp][64]: the input of the embedder's first layer (hand-off)
                    const int row = tid >> 3, q8 = tid & 7, rb = 32 * g + row;
                    bf16_t* Mout = static_cast<bf16_t*>(out_ptr);
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Mout + (size_t)tp * Bp * 64, (unsigned)((size_t)Bp * 64 * 2));
                    const uint4 v = *reinterpret_cast<const uint4*>(ost + row * ORS + q8 * 16);
                    st16_sc1(ro, rb < Bp ? (unsigned)((rb * 64 + 8 * q8) * 2) : kOob, v);
                }
                raise_flag<0>(rflags + ((size_t)g * Tp + tp) * a.flag_stride);
            }
        }
        if (!has_next) break;
        c = cn;
        t = tn;
    }
}

template <int KS>
constexpr int fused_fwd2_lds_bytes() {
    int m = LstmFwd2Lds<KS, 0>::BYTES;
    m = m > LstmFwd2Lds<KS, 2>::BYTES ? m : LstmFwd2Lds<KS, 2>::BYTES;
    m = m > LstmFwd2Lds<KS, 4>::BYTES ? m : LstmFwd2Lds<KS, 4>::BYTES;
    m = m > GemmFwd2Lds<KS>::BYTES ? m : GemmFwd2Lds<KS>::BYTES;
    return (m + 15) / 16 * 16;
}

template <int KS>
__device__ __forceinline__ void fused_fwd2_role(const FusedArgs& a, const FusedRole& R, int set, int p, unsigned char* lds) {
    switch (R.type) {
        case FR_LSTM_FWD:   // (the role tables give the CP-fed layer rows nobody in the launch writes, every other layer rows of a role)
            if (R.ksx == 2 && !R.src_sc1) fused_lstm_fwd2<KS, 2, false, true>(a, R, set, p, lds);
            else if (R.ksx == 2) fused_lstm_fwd2<KS, 2, true, true>(a, R, set, p, lds);
            else if (R.ksx == 4) fused_lstm_fwd2<KS, 4, true, true>(a, R, set, p, lds);
            else fused_lstm_fwd2<KS, 0, true, true>(a, R, set, p, lds);
            break;
        case FR_PROJ_FWD: fused_gemm_fwd2<KS, false>(a, R, set, p, lds); break;
        case FR_HEAD_FWD: fused_gemm_fwd2<KS, true>(a, R, set, p, lds); break;
        default: break;
    }
}

template <int KSP, int KSE>
__global__ __launch_bounds__(256, 2) void fused_fwd2_kernel(FusedArgs a) {
    constexpr int kLds = fused_fwd2_lds_bytes<KSP>() > fused_fwd2_lds_bytes<KSE>() ? fused_fwd2_lds_bytes<KSP>() : fused_fwd2_lds_bytes<KSE>();
    static_assert(kLds <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    if (a.prio > 0 && R.C <= a.prio) __builtin_amdgcn_s_setprio(3);
    const int set_step = a.gpp > 0 ? uni(a.gpp / R.C) : 0;   // passes: as in fused_fwd_kernel
    for (int s2 = set;; s2 += set_step) {
        if constexpr (KSP == KSE) {
            fused_fwd2_role<KSE>(a, R, s2, p, lds);
        } else {
            if (R.wide) fused_fwd2_role<KSE>(a, R, s2, p, lds);
            else fused_fwd2_role<KSP>(a, R, s2, p, lds);
        }
        if (set_step <= 0 || (s2 + set_step) * R.C >= a.n_groups) break;
        __syncthreads();
        if (uni(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
    }
#ifdef PL_STAMPS
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)(role + 1) | ((unsigned long long)set << 8) | ((unsigned long long)p << 16);   // who this block was (tools/fused2_stamps.py)
#endif
}

// ---- the recurrence role alone as a per-layer forward sweep (batches of more groups than one pass of lstm_fwd_sweep_kernel holds) ------
// lstm_persist.hip's forward sweep runs one workgroup per CU: 11 groups of 23 workgroups a pass at H = 720, 64 groups (cfg4's 2048 rows on
// one GPU) in 6 passes of 300 x 4.2 us.  The two-per-CU role holds 22 groups a pass at 5.9 us a step: 3 passes.  Workgroup b serves slice
// b % P of the groups b / P, b / P + sets, ... one after the other; groups are independent, every group's own flags [T][flag_stride] are
// the sweep's `counters`, and the role function is the fused launch's (one chain, no waits on other roles): same bits in G, h, c.
template <int KS>
__global__ __launch_bounds__(256, 2) void lstm_fwd2_sweep_kernel(LstmSweepArgs s, int sets) {
    constexpr int P = 16 * KS / 32;
    constexpr int kLds = fused_fwd2_lds_bytes<KS>();
    static_assert(kLds <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    const int set = uni((int)blockIdx.x / P), p = uni((int)blockIdx.x % P);
    if (set >= sets) return;
    FusedArgs a{};
    a.Bp = s.Bp; a.B = s.Bp; a.n_groups = (s.Bp + 31) / 32; a.flag_stride = s.flag_stride;
    a.status = s.status; a.spin_ticks = s.spin_ticks; a.poll_mask = s.poll_mask; a.stamps = s.stamps;
    FusedRole R{};
    R.type = FR_LSTM_FWD; R.C = 1; R.T = s.T; R.flags = s.counters;
    R.wait[0] = FusedWait{s.counters, s.T, P, 0, 0, -1};
    R.G = s.G; R.W = s.W; R.h = s.h; R.c = s.c;
    R.ksx = s.x_in ? s.in_p / 16 : 0; R.x_in = s.x_in; R.Wih = s.Wih; R.bias = s.bias;
    for (int g = set; g < a.n_groups; g += sets) {
        if (R.ksx == 2) fused_lstm_fwd2<KS, 2, false, false>(a, R, g, p, lds);
        else if (R.ksx == 4) fused_lstm_fwd2<KS, 4, false, false>(a, R, g, p, lds);
        else fused_lstm_fwd2<KS, 0, false, false>(a, R, g, p, lds);
        __syncthreads();   // nobody starts the next group's LDS images while a wave still reads this one's
        if (uni(__hip_atomic_load(s.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;   // a wait gave up: everybody leaves
    }
}

}  // namespace

#define PL_FUSED_FWD2_KS(X) X(6) X(12) X(46)
bool lstm_fwd2_sweep_supported(int Hp) {
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_FUSED_FWD2_KS(PL_CASE)
#undef PL_CASE
    return false;
}
// sets: groups served at once (<= 2 n_cu / P); the grid is sets x P workgroups
void launch_lstm_fwd2_sweep(hipStream_t stream, int Hp, int sets, const LstmSweepArgs& s) {
#define PL_CASE(K)                                                                                              \
    if (Hp == 16 * K) {                                                                                         \
        hipLaunchKernelGGL(lstm_fwd2_sweep_kernel<K>, dim3(sets * (16 * K / 32)), dim3(256), 0, stream, s, sets); \
        return;                                                                                                 \
    }
    PL_FUSED_FWD2_KS(PL_CASE)
#undef PL_CASE
}

#define PL_FUSED_FWD2_PAIRS(X) X(6, 6) X(46, 46) X(12, 46)

bool fused_fwd2_xcd_compiled() { return FUSED2_XCD != 0; }
bool fused_fwd2_supported(int Hp_pred, int Hp_emb) {
#define PL_CASE(KP, KE) if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) return true;
    PL_FUSED_FWD2_PAIRS(PL_CASE)
#undef PL_CASE
    return false;
}

void launch_fused_fwd2(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a) {
#define PL_CASE(KP, KE)                                                                               \
    if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) {                                                    \
        hipLaunchKernelGGL((fused_fwd2_kernel<KP, KE>), dim3(a.grid), dim3(256), 0, stream, a);       \
        return;                                                                                       \
    }
    PL_FUSED_FWD2_PAIRS(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
