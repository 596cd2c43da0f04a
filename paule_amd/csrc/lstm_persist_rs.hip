// Backward LSTM sweep, "reduce-scatter" form (bf16).
//
// The all-gather form (lstm_persist.hip) makes every workgroup of a batch group read the group's whole
// dA_{t+1} (32 rows x 4*Hp bf16 = 188 KB per workgroup per step at H = 720): 3.3 of its 5.4 us per step.
// Here each workgroup multiplies only the dA it produced itself (its 32 hidden units x 4 gates = 128 gate
// rows, straight from LDS) with the matching 128 rows of W_hh, i.e. its PARTIAL contribution to dh_t of ALL
// hidden units, and hands the partials over; a workgroup then sums the P partial 32 x 32 tiles that belong to
// its own hidden units.  Per workgroup and step: 47 KB written + 47 KB read instead of 188 KB read.
//
//   partial_p[b][n] = sum_{k in gate rows of slice p} dA_t[b][k] * W_hh[k][n]        (MFMA, K = 128, N = Hp)
//   dh_{t-1}[b][j]  = dh_ext_{t-1}[b][j] + sum_p partial_p[b][j]                      (j in own slice)
//
// Price: the partials cross the exchange rounded to bf16 (each is a 128-term f32 sum; 23 of them are then added
// in f32) -- the same order of rounding as dA itself being bf16; covered by the bf16 parity tests.
// W_hh^T rows stay in registers (wave w owns N tiles w, w+4, ...: 6 x 8 k-steps x 4 VGPRs = 192), the running dc
// never leaves registers.  Exchange protocol, bounded waits, zeroing: as in lstm_persist.hip.
#include "sweep_common.h"

namespace pl {

// NW = waves per workgroup.  4: one wave per SIMD, a wave owns N tiles w, w + 4, ...  8 (round 3, the default): waves 0 .. 3 do
// what they did (poll, ingest, cell update: same threads, same summation order), and the tile products are dealt over EIGHT
// waves, two per SIMD -- a tile's epilogue (accumulators -> bf16 -> LDS -> read-back by rows -> hand-off stores) is a dependent
// chain that leaves the matrix pipe idle, and the SIMD's other wave now multiplies meanwhile.  A tile is still produced by one
// wave with the same 8 MFMAs, so the results are bit-identical to NW = 4.
template <int KS, int NW>   // KS = Hp / 16
__global__ __launch_bounds__(64 * NW, 1) void lstm_bwd_rs_sweep_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;                   // slices = N tiles of 32 hidden units
    constexpr int NT = (P + NW - 1) / NW;        // N tiles per wave (wave w: tiles w, w + NW, ...)
    constexpr int DRS = 128 * 2 + 16;            // dA^T image: [32 batch rows][128 local gate rows] bf16, odd chunk stride
    constexpr int ORS = Hp * 2 + 16;             // partial image: [32 batch rows][Hp] bf16
    constexpr int NTH = 64 * NW;
    constexpr int NST = (P * 128 + NTH - 1) / NTH;   // hand-off stores (16 B) per thread per step
    __shared__ __attribute__((aligned(16))) unsigned char da_img[32 * DRS];
    __shared__ __attribute__((aligned(16))) unsigned char out_img[32 * ORS];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cellw = wave < 4;                 // the waves that own cells (a scalar condition: whole waves)
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);   // Whh^T packed [Hp][4*Hp]

    // weights -> registers: tile nt = wave + 4 i covers hidden columns n = 32 nt + (lane & 31); local k = 16 ks + 8 (lane >> 5) + jj
    // maps to gate row (ks / 2) * Hp + 32 p + 16 (ks % 2) + 8 (lane >> 5) + jj of this slice
    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + NW * i;
        const int n = 32 * (nt < P ? nt : 0) + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wreg[i][ks] = *reinterpret_cast<const uint4*>(WT + (size_t)n * G4 + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
    }

    // cell ownership (threads of waves 0 .. 3): thread -> batch row (tid >> 3), hidden units 32p + 4 (tid & 7) .. +3
    const int erow = (tid & 255) >> 3, jq = tid & 7;
    const int j = 32 * p + 4 * jq;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);
    // exchange layout [2 slots][groups][P destinations][P sources][32 rows][32 columns]: a destination reads ONE contiguous
    // 2 KB x P block (a wave instruction = 512 contiguous bytes), a source writes P contiguous 2-KB tiles
    bf16_t* __restrict__ X = static_cast<bf16_t*>(a.xchg);
    constexpr size_t TILE = 32 * 32;                                  // elements of one (destination, source) tile
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + erow;
        const bool ok = erow < gs && b < Bp;   // (waves 0 .. 3)
        const int bc = ok ? b : Bp - 1;
        float dc_next[4] = {0.f, 0.f, 0.f, 0.f};
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;   // flags [group][step][flag_stride]

        for (int t = T - 1; t >= 0; --t) {
            // stash operands of this step (written by the forward launch): plain loads, issued before the wait
            uint2 sg[4] = {}, sc = make_uint2(0u, 0u), scp = make_uint2(0u, 0u), sdh = make_uint2(0u, 0u);
            if (cellw) {
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) sg[q] = *reinterpret_cast<const uint2*>(g_row + q * Hp);
                sc = *reinterpret_cast<const uint2*>(Cs + (size_t)t * slabH + (size_t)bc * Hp + j);
                if (t > 0) scp = *reinterpret_cast<const uint2*>(Cs + (size_t)(t - 1) * slabH + (size_t)bc * Hp + j);
                if (dhe) sdh = *reinterpret_cast<const uint2*>(dhe + (size_t)t * slabH + (size_t)bc * Hp + j);
                else if (dhl && t == T - 1) sdh = *reinterpret_cast<const uint2*>(dhl + (size_t)bc * Hp + j);
            }

            float dh[4];
            unpack_bf16x4(sdh, dh);
            PL_ST(0);
            if (t + 1 < T) {
                if (!wait_arrivals(cnt + (size_t)(t + 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == T - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                if (cellw) {
                    // sum the P partial tiles of step t+1 that belong to this thread's cells (sc1 loads: handed-off bytes)
                    const bf16_t* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 2));
                    const unsigned o0 = (unsigned)((erow * 32 + 4 * jq) * 2);
                    u32x2 pv[P];
#pragma unroll
                    for (int s = 0; s < P; ++s)
                        pv[s] = plain_handoff ? __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxNt)
                                              : __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxSc1);
#pragma unroll
                    for (int s = 0; s < P; ++s) {
                        float f[4];
                        unpack_bf16x4(make_uint2(pv[s][0], pv[s][1]), f);
                        dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
                    }
                }
            }
            PL_ST(2);   // partial ingest

            if (cellw) {
                float gi[4], gf[4], gg[4], go[4], c[4], cp[4];
                unpack_bf16x4(sg[0], gi);
                unpack_bf16x4(sg[1], gf);
                unpack_bf16x4(sg[2], gg);
                unpack_bf16x4(sg[3], go);
                unpack_bf16x4(sc, c);
                unpack_bf16x4(scp, cp);
                float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], c[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
                const uint2 pi = pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), pf = pack_bf16x4(daf[0], daf[1], daf[2], daf[3]);
                const uint2 pg = pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), po = pack_bf16x4(dao[0], dao[1], dao[2], dao[3]);
                if (ok) {   // dA_t overwrites the gate stash in place (read later by the dX / dH GEMM launches)
                    bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                    *reinterpret_cast<uint2*>(go_) = pi;
                    *reinterpret_cast<uint2*>(go_ + Hp) = pf;
                    *reinterpret_cast<uint2*>(go_ + 2 * Hp) = pg;
                    *reinterpret_cast<uint2*>(go_ + 3 * Hp) = po;
                }
                if (t > 0) {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 32 + unit]
                    unsigned char* drow = da_img + erow * DRS + jq * 8;
                    *reinterpret_cast<uint2*>(drow) = pi;
                    *reinterpret_cast<uint2*>(drow + 64) = pf;
                    *reinterpret_cast<uint2*>(drow + 128) = pg;
                    *reinterpret_cast<uint2*>(drow + 192) = po;
                }
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == T - 1 && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            PL_ST(3);   // cell + stash stores + dA image
            uint4 bfr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                bfr[ks] = *reinterpret_cast<const uint4*>(da_img + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            const bool own_store = (a.stash_via_lds & 2) != 0;   // A/B: every wave hands its own tiles over right behind their MFMAs
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int nt = wave + NW * i;
                if (NW * i + NW - 1 >= P && nt >= P) break;   // a compile-time fact for all but a wave's last tile
                // (the break sits IN FRONT of a tile's MFMAs, behind the previous tile's epilogue: no MFMA result is read across it --
                // tools/isa_mfma_hazard_scan.py, profiles/r04_isa_stale_accumulator.txt)
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                // acc[r] = partial[n = 32 nt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][batch lane & 31] -> bf16 image [batch][n]
                unsigned char* orow = out_img + (lane & 31) * ORS + (32 * nt + 4 * (lane >> 5)) * 2;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<uint2*>(orow + rg * 16) = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
                if (own_store) {
                    // tile nt is this wave's alone: read it back by rows (LDS operations of a wave are ordered) and hand it over now,
                    // under the MFMAs of the wave's next tile; 128 16-byte chunks = 2 per lane, the tile is 2 KB contiguous
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
                        const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * nt + 8 * c4) * 2);
                        u32x4 d;
                        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                        const unsigned off = (unsigned)(((size_t)nt * P * TILE + cidx * 8) * 2);
                        if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                    }
                }
            }
            PL_ST(4);   // MFMA + partial image
            if (!own_store) {   // hand-off after a barrier: this workgroup's partial tile rows, whole 16-byte chunks, write-through
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    const int e = tid + NTH * i;      // 16-byte chunk: destination e / 128, row (e % 128) / 4, quarter e % 4
                    if (e < P * 128) {
                        const int dst = e >> 7, r = (e & 127) >> 2, c4 = e & 3;
                        const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * dst + 8 * c4) * 2);
                        u32x4 d;
                        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                        const unsigned off = (unsigned)(((size_t)dst * P * TILE + (e & 127) * 8) * 2);
                        if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                    }
                }
            }
            PL_ST(5);   // hand-off store issue
            publish<0>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);   // drain + barrier + arrival add
        }
    }
    PL_ST_DUMP(a.stamps);
}

// LDS-DMA: 64 lanes x 16 bytes from base + lane_off land at lds_dst + 16 * lane.  Inline asm: the compiler neither sees the LDS write
// nor counts the operation -- the issuing wave waits with s_waitcnt vmcnt(0) and a barrier follows before anybody reads the bytes.
__device__ __forceinline__ void glds16(const void* gsrc_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst_uniform)
        : "memory");
}
typedef __attribute__((address_space(3))) unsigned char* rs_lds_ptr_t;

// ---------------------------------------------------------------------------------------------------------------------
// Streamed form of the reduce-scatter (round 3, the default).  In the kernel above a step is a chain of whole-workgroup phases:
// all tiles -> drain -> barrier -> ONE flag -> (consumers) poll -> 23 tile loads -> cell -> tiles ...; the last tile's way to its
// consumer starts only when the slowest wave of the producer has drained, and the consumer's 47 KB of tile loads start only when
// the last producer has signalled (profiles/r03_sweep_phase_stamps.txt: 1.49 us wait + 1.13 us ingest of a 4.55-us step).  Here
// every 32 x 32 tile travels on its own:
//   * a tile is produced by ONE wave, so that wave alone drains its two stores (counted vmcnt, one tile behind: the wait sits under
//     the next tile's MFMAs) and raises the tile's OWN flag -- tflags[slot][group][destination][source] = step + 1;
//   * source p produces its tiles in the rotated order destination p + 1, p + 2, ... (mod P), eight waves at a time, so every
//     destination receives about a third of its tiles per round instead of all of them in one round (destination 20 would get all 23
//     in the last round and hold the whole group up);
//   * the four cell-owning waves of a destination poll the 23 flags of their destination with one wave instruction, load whatever
//     has newly arrived, and poll again -- two thirds of the ingest is under way by the time the last round's flags come up.
// Same tiles, same 8 MFMAs per tile, same fixed-order sum over the sources: bit-identical to the kernel above.  One workgroup
// barrier per step (the dA image is double-buffered); hand-off rules as before (write-through sc1 both sides, or plain / nt through
// the shared L2 for a group that verified it sits on one XCD); every spin bounded.
// DMA = 1 (round 4): the stash rows of a step (gates, c_t, c_{t-1}, dL/dh from above: 14 KB per workgroup) are fetched ONE STEP AHEAD by
// LDS-DMA, issued by waves 4 .. 7 (they own no cells and idle through the ingest); the cell waves find them in LDS.  Before, the cell
// waves loaded them at the step's top -- ~1 us from HBM -- and since a wave's vector-memory operations retire in order, the first flag
// poll's answer and every tile load queued behind them.  dA_t leaves for the gate stash from the LDS image as 16-byte pieces, stored by
// waves 4 .. 7 behind the barrier (2 store instructions per lane instead of 4 in the cell waves' critical phase).  Same values, same
// order of every sum: bit-identical to DMA = 0 (tests/test_hip_parity.py::test_backward_sweep_forms_are_bit_identical).
// XT = 1 (round 4): the partial input gradient of the step rides along -- the last wave's free tile slot (P tiles on 8 x NT slots)
// holds this workgroup's 128 rows of W_ih^T, its product with the dA image (the B operand every tile of the step uses) is one more
// tile of 8 MFMAs, produced behind the wave's exchange tiles and stored as f32 straight from the accumulators (no flag, nobody waits
// for it in the launch).  launch_dx_reduce sums the P partials afterwards; the batched product, which re-read the whole dA stash,
// is gone (cfg3: 107 us -> ~45 us).  The recurrence itself is untouched: every dA and every exchange tile keep their bits.
// Prefetcher workgroups of the streamed sweeps (LstmSweepArgs::n_pf; sweep_common.h: stash_prefetch_walk), paced on the per-tile flags:
// the row of destination 0 holds step tokens u + 1 of the steps with u's parity, and later steps overwrite them with smaller ones
template <int KS>
__device__ __forceinline__ void rs_prefetch_role(const LstmSweepArgs& a, const int q, const int n_res) {
    constexpr int Hp = 16 * KS, P = Hp / 32;
    const int gs = a.group_rows, n_groups = (a.Bp + gs - 1) / gs;
    const int lane = threadIdx.x & 63;
    stash_prefetch_walk(a, Hp, 2, q, n_res, a.T - 1, 0, [&](int g, int u) {
        if (u < 1 || !a.tflags) return;
        const int* frow = a.tflags + ((size_t)(u & 1) * n_groups + g) * P * 32;
        const __amdgpu_buffer_rsrc_t rf = make_rsrc(frow, (unsigned)(P * 4));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            int v = 1;
            if (lane < P) v = (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
            if (__all(v != 0 && v <= u + 1)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000ull) break;   // 20 us without news: go on unpaced
            __builtin_amdgcn_s_sleep(16);
        }
    });
}

#ifndef RS_DA_LATE
#define RS_DA_LATE 0
#endif
template <int KS, int DMAV, int XT = 0>   // DMAV bit 0: stash rows by LDS-DMA a step ahead; bit 1: dA_t stored by waves 4 .. 7 from the image
__global__ __launch_bounds__(512, 1) void lstm_bwd_rs_stream_kernel(LstmSweepArgs a) {
    constexpr bool DMA = (DMAV & 1) != 0, DAW = (DMAV & 2) != 0;
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;
    constexpr int NW = 8;
    constexpr int NT = (P + NW - 1) / NW;
    constexpr int DRS = 128 * 2 + 16;
    constexpr int ORS = Hp * 2 + 16;
    static_assert(P <= 32, "one flag word per source in a 32-int row");
    static_assert(!XT || NW * NT > P, "the ride-along tile needs a free tile slot in the last wave");
    __shared__ __attribute__((aligned(16))) unsigned char da_img[2][32 * DRS];
    __shared__ __attribute__((aligned(16))) unsigned char out_img[32 * ORS];
    __shared__ __attribute__((aligned(16))) unsigned char st_img[DMA ? 2 : 1][DMA ? 7 : 1][DMA ? 2048 : 16];   // stash rows: [i, f, g, o, c_t, c_{t-1}, dh][32 rows][64 B]
    __shared__ int lds_flag, lds_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cellw = wave < 4;
    const int n_main = (int)gridDim.x - a.n_pf;
    const int n_res = n_main / P;
    if ((int)blockIdx.x >= n_main) {   // prefetcher workgroups (speed only)
        rs_prefetch_role<KS>(a, (int)blockIdx.x - n_main, n_res);
        return;
    }
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);   // Whh^T packed [Hp][4*Hp]

    // this wave's tiles: positions k = wave + 8 i of the rotated order, destination (p + 1 + k) mod P
    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int k = wave + NW * i;
        const int nt = k < P ? (p + 1 + k) % P : 0;
        const int n = 32 * nt + (lane & 31);
        // the last wave's last slot is free (k >= P): with XT it holds W_ih^T -- row n = input column lane & 31, the same 128 gate rows
        const bf16_t* wsrc = (XT && i == NT - 1 && wave == NW - 1) ? static_cast<const bf16_t*>(a.WihT) + (size_t)(lane & 31) * G4 : WT + (size_t)n * G4;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wreg[i][ks] = *reinterpret_cast<const uint4*>(wsrc + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
    }

    const int erow = (tid & 255) >> 3, jq = tid & 7;
    const int j = 32 * p + 4 * jq;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);
    bf16_t* __restrict__ X = static_cast<bf16_t*>(a.xchg);
    constexpr size_t TILE = 32 * 32;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    const unsigned st_lds = (unsigned)(uintptr_t)(rs_lds_ptr_t)&st_img[0][0][0];
    if (tid == 0) lds_abort = 0;
    __syncthreads();

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + erow;
        const bool ok = erow < gs && b < Bp;
        const int bc = ok ? b : Bp - 1;
        float dc_next[4] = {0.f, 0.f, 0.f, 0.f};
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;
        // tile flags [2 slots][groups][P destinations][32]: the row of destination d holds one word per source
        int* const tf = a.tflags;
        // DMA: stash rows of step ts -> LDS image ts & 1, by waves 4 .. 7: 14 pieces of 1 KB (an array's rows 0 .. 15 / 16 .. 31, 64 B a row);
        // wave 4 + w takes pieces w, w + 4, w + 8, w + 12: a lane's byte offset inside a time slab is the same for every step
        unsigned pf_off[4] = {0u, 0u, 0u, 0u};
        if (DMA) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = (wave & 3) + 4 * qi, arr = q >> 1, row = 16 * (q & 1) + (lane >> 2);
                const int br = gs * g + row;
                const int bcl = (row < gs && br < Bp) ? br : Bp - 1;
                pf_off[qi] = arr < 4 ? (unsigned)(((size_t)bcl * G4 + (size_t)arr * Hp + 32 * p + 8 * (lane & 3)) * 2)
                                     : (unsigned)(((size_t)bcl * Hp + 32 * p + 8 * (lane & 3)) * 2);
            }
        }
        auto prefetch = [&](int ts) {
            const bf16_t* const gb = G + (size_t)ts * slabG;
            const bf16_t* const cb = Cs + (size_t)ts * slabH;
            const bf16_t* const db = dhe ? dhe + (size_t)ts * slabH : ((dhl && ts == T - 1) ? dhl : nullptr);
            const unsigned dst0 = st_lds + (unsigned)((ts & 1) * 7 * 2048);
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = (wave & 3) + 4 * qi, arr = q >> 1;   // wave-uniform
                if (q < 14) {
                    const bf16_t* base = arr < 4 ? gb : (arr == 4 ? cb : (arr == 5 ? (ts > 0 ? cb - slabH : nullptr) : db));
                    if (base) glds16(base, pf_off[qi], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + (unsigned)(q * 1024))));
                }
            }
        };
        if (DMA) {
            if (!cellw) {
                prefetch(T - 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }

        uint2 c_carry = make_uint2(0u, 0u);   // c_{t-1} of the step before = c_t of this one: loaded once (one stash row of seven less in front of the polls)
        for (int t = T - 1; t >= 0; --t) {
            uint2 sg[4] = {}, sc = make_uint2(0u, 0u), scp = make_uint2(0u, 0u), sdh = make_uint2(0u, 0u);
            if (DMA && !cellw && t > 0) prefetch(t - 1);   // next step's stash rows, under this step's ingest and cell update
            if (!DMA && cellw) {
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) sg[q] = *reinterpret_cast<const uint2*>(g_row + q * Hp);
                if (t == T - 1) sc = *reinterpret_cast<const uint2*>(Cs + (size_t)t * slabH + (size_t)bc * Hp + j);
                else sc = c_carry;
                if (t > 0) scp = *reinterpret_cast<const uint2*>(Cs + (size_t)(t - 1) * slabH + (size_t)bc * Hp + j);
                c_carry = scp;
                if (dhe) sdh = *reinterpret_cast<const uint2*>(dhe + (size_t)t * slabH + (size_t)bc * Hp + j);
                else if (dhl && t == T - 1) sdh = *reinterpret_cast<const uint2*>(dhl + (size_t)bc * Hp + j);
            }
            if (DMA && cellw) {   // only dL/dh from above is needed before the partial sums are added (the order of that f32 sum is part of the result)
                if (dhe || (dhl && t == T - 1)) sdh = *reinterpret_cast<const uint2*>(&st_img[t & 1][6][0] + erow * 64 + jq * 8);
            }
            float dh[4];
            unpack_bf16x4(sdh, dh);
            PL_ST(0);
            if (t + 1 < T) {
                if (cellw) {
                    // The P tiles of step t + 1 for this workgroup's units, loaded as their flags come up: each cell wave polls the row
                    // of its destination with one wave instruction and issues the loads of whatever is new.  (vmcnt retires in order, so
                    // a poll's answer is seen after the tile loads issued before it have landed -- loads that are needed anyway.
                    // Measured and dropped: a fifth wave that only polls and hands the arrival mask round through LDS -- its polls
                    // queue behind the cell waves' tile loads in the CU's memory pipe, 5.54 vs 5.22 ms per iteration.)
                    const int token = t + 2;
                    const int* frow = tf + ((size_t)((t + 1) & 1) * n_groups + g) * P * 32 + (size_t)p * 32;
                    const __amdgpu_buffer_rsrc_t rf = make_rsrc(frow, (unsigned)(P * 4));
                    const bf16_t* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 2));
                    const unsigned o0 = (unsigned)((erow * 32 + 4 * jq) * 2);
                    constexpr unsigned full = P == 32 ? 0xffffffffu : ((1u << P) - 1u);
                    unsigned issued = 0;
                    u32x2 pv[P];
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 1;; ++spin) {
                        int v = 0;
                        if (lane < P)
                            v = plain_handoff ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxNt)
                                              : (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
                        const unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(v == token) & full;
                        const unsigned newly = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask & ~issued));
#pragma unroll
                        for (int s = 0; s < P; ++s)
                            if ((newly >> s) & 1u)
                                pv[s] = plain_handoff ? __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxNt)
                                                      : __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxSc1);
                        issued |= newly;
                        if (issued == full) break;
                        if ((spin & a.poll_mask) == 0 &&
                            (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 ||
                             __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks)) {
                            if (lane == 0) {
                                __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                lds_abort = 1;
                            }
#pragma unroll
                            for (int s = 0; s < P; ++s)
                                if (!((issued >> s) & 1u)) pv[s] = u32x2{0u, 0u};
                            break;
                        }
                    }
                    PL_ST(1);   // polls + tile loads issued
#pragma unroll
                    for (int s = 0; s < P; ++s) {
                        float f[4];
                        unpack_bf16x4(make_uint2(pv[s][0], pv[s][1]), f);
                        dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
                    }
                }
                if (t == T - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
            }
            PL_ST(2);   // tiles landed + sums

            unsigned char* const dimg = da_img[t & 1];
            if (cellw) {
                if (DMA) {
                    const unsigned char* sb = &st_img[t & 1][0][0] + erow * 64 + jq * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sg[q] = *reinterpret_cast<const uint2*>(sb + q * 2048);
                    sc = *reinterpret_cast<const uint2*>(sb + 4 * 2048);
                    if (t > 0) scp = *reinterpret_cast<const uint2*>(sb + 5 * 2048);
                }
                float gi[4], gf[4], gg[4], go[4], c[4], cp[4];
                unpack_bf16x4(sg[0], gi);
                unpack_bf16x4(sg[1], gf);
                unpack_bf16x4(sg[2], gg);
                unpack_bf16x4(sg[3], go);
                unpack_bf16x4(sc, c);
                unpack_bf16x4(scp, cp);
                float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], c[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
                const uint2 pi = pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), pf = pack_bf16x4(daf[0], daf[1], daf[2], daf[3]);
                const uint2 pg = pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), po = pack_bf16x4(dao[0], dao[1], dao[2], dao[3]);
                if (ok && (!DAW || t == 0) && !(XT && a.skip_dA) && (!RS_DA_LATE || t == 0)) {   // dA_t overwrites the gate stash in place (read later by the dX / dH GEMM launches)
                    bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                    *reinterpret_cast<uint2*>(go_) = pi;
                    *reinterpret_cast<uint2*>(go_ + Hp) = pf;
                    *reinterpret_cast<uint2*>(go_ + 2 * Hp) = pg;
                    *reinterpret_cast<uint2*>(go_ + 3 * Hp) = po;
                }
                if (t > 0 || XT) {   // (XT: step 0 has an input gradient too, though nobody consumes its recurrent partials)
                    unsigned char* drow = dimg + erow * DRS + jq * 8;
                    *reinterpret_cast<uint2*>(drow) = pi;
                    *reinterpret_cast<uint2*>(drow + 64) = pf;
                    *reinterpret_cast<uint2*>(drow + 128) = pg;
                    *reinterpret_cast<uint2*>(drow + 192) = po;
                }
            }
            if (t == 0 && !XT) break;   // nobody consumes the partials of step 0
            if (t == T - 1 && tid == 0) {   // this workgroup's XCD, in place before ANY of its flags (they are raised behind the barrier below)
                __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (DMA && !cellw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stash rows of step t - 1 have landed in LDS
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(lds_abort) != 0) return;
            PL_ST(3);   // cell + stash stores + dA image + barrier
            uint4 bfr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                bfr[ks] = *reinterpret_cast<const uint4*>(dimg + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            int* const fcol = tf + ((size_t)(t & 1) * n_groups + g) * P * 32 + p;   // + 32 * destination
            const __amdgpu_buffer_rsrc_t rfl = make_rsrc(fcol, (unsigned)(((P - 1) * 32 + 1) * 4));
            auto raise = [&](int nt) {   // the wave's own stores of that tile are acknowledged: its flag
                if (lane == 0) {
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 32 * 4), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 32 * 4), 0, kAuxSc1);
                }
            };
            int nt_prev = -1;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int k = wave + NW * i;
                if (XT && t == 0) break;                      // step 0: nobody consumes recurrent partials (only the input gradient below)
                if (NW * i + NW - 1 >= P && k >= P) break;   // a compile-time fact for all but a wave's last tile
                // (the break sits IN FRONT of a tile's MFMAs, behind the previous tile's epilogue: no MFMA result is read across it --
                // tools/isa_mfma_hazard_scan.py, profiles/r04_isa_stale_accumulator.txt)
                const int nt = (p + 1 + k) % P;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                unsigned char* orow = out_img + (lane & 31) * ORS + (32 * nt + 4 * (lane >> 5)) * 2;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<uint2*>(orow + rg * 16) = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
                    const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * nt + 8 * c4) * 2);
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    const unsigned off = (unsigned)(((size_t)nt * P * TILE + cidx * 8) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
                if (nt_prev >= 0) {   // the tile before: all but this tile's two stores have been acknowledged
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    raise(nt_prev);
                }
                nt_prev = nt;
            }
            if (nt_prev >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise(nt_prev);
            }
            if (XT && wave == NW - 1) {
                // the ride-along tile, BEHIND the wave's last flag (nobody in the launch waits for its stores): partial dX[n = input column][batch] over this workgroup's 128 gate rows, f32 from the accumulators;
                // acc[4 rg + e] = partial[n = 8 rg + 4 (lane >> 5) + e][batch lane & 31] -> 16 bytes at rg * 1 KB + lane * 16
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[NT - 1][ks]),
                                                                  __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                float* xp = a.xpart + (((size_t)t * n_groups + g) * P + p) * 1024;
                const __amdgpu_buffer_rsrc_t rp = make_rsrc(xp, 4096u);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    // (through scalars: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 -- hipcc stored acc[0] sixteen times)
                    const float f0 = acc[4 * rg], f1 = acc[4 * rg + 1], f2 = acc[4 * rg + 2], f3 = acc[4 * rg + 3];
                    u32x4 d;
                    d[0] = __float_as_uint(f0); d[1] = __float_as_uint(f1); d[2] = __float_as_uint(f2); d[3] = __float_as_uint(f3);
                    __builtin_amdgcn_raw_buffer_store_b128(d, rp, (unsigned)(rg * 1024 + lane * 16), 0, 0);
                }
            }
            if (XT && t == 0) break;
            if (RS_DA_LATE && !DAW && cellw && ok && !(XT && a.skip_dA)) {   // experiment: the cell waves' dA stores BEHIND their tiles and flags, read back from the image
                const unsigned char* drow = dimg + erow * DRS + jq * 8;
                bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                *reinterpret_cast<uint2*>(go_) = *reinterpret_cast<const uint2*>(drow);
                *reinterpret_cast<uint2*>(go_ + Hp) = *reinterpret_cast<const uint2*>(drow + 64);
                *reinterpret_cast<uint2*>(go_ + 2 * Hp) = *reinterpret_cast<const uint2*>(drow + 128);
                *reinterpret_cast<uint2*>(go_ + 3 * Hp) = *reinterpret_cast<const uint2*>(drow + 192);
            }
            if (DAW && !cellw) {   // dA_t -> the gate stash, from the image (double-buffered: intact until step t - 2), as 16-byte pieces (gate ch >> 2,
                // units 8 (ch & 3) .. + 7) -- BEHIND the tiles: in front of them the acknowledge of these stores (HBM, ~1 us) sat in the way of every
                // tile flag's counted wait (tile phase 1.23 -> 2.20 us, profiles/r04_ab_bwd_dma.txt)
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = (tid & 255) + 256 * q, row = e >> 4, ch = e & 15;
                    const uint4 v = *reinterpret_cast<const uint4*>(dimg + row * DRS + ch * 16);
                    const int br = gs * g + row;
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    __builtin_amdgcn_raw_buffer_store_b128(d, rg, (row < gs && br < Bp) ? (unsigned)(((size_t)br * G4 + (size_t)(ch >> 2) * Hp + 32 * p + 8 * (ch & 3)) * 2) : 0x80000000u, 0, 0);
                }
            }
            PL_ST(4);   // tiles + flags
        }
        __syncthreads();   // a workgroup that sweeps several groups in turn: nobody writes the next group's dA image while a wave still reads this one's
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chained form of the streamed reduce-scatter (round 5): a workgroup serves C batch groups ("chains") with ONE copy of its W_hh^T slice
// and takes a step of each in turn -- t, chain 0; t, chain 1; ...; t - 1, chain 0; ...  A group's step is a latency chain (tiles out,
// acknowledge, flag, poll, tiles in: ~2.5 of the 4.05 us of lstm_bwd_rs_stream_kernel's step, during which the CU idles); with the other
// chains' steps in between, a chain finds its tiles in place when its turn comes again, and a sweep needs groups / C x P workgroups
// instead of groups x P -- what lets several layers' recurrences run side by side on the chip.  Per (group, step) everything is the
// streamed kernel's: the same tiles from the same 8 MFMAs in the same rotated order, the same per-tile flags and exchange buffer
// (indexed by the group), the same fixed-order sum, the same-XCD form per group -- results are bit-identical
// (tests/test_hip_parity.py::test_backward_sweep_forms_are_bit_identical).  The running dL/dc of the chains lives in LDS.
// Grid: n_slots x P with set = blockIdx % n_slots (8 slots: a set's workgroups share blockIdx % 8, one XCD under the observed dealing --
// speed only, verified at run time like every same-XCD choice); sets beyond ceil(groups / C) leave at once.
constexpr int kRsMaxChains = 4;
template <int KS, int XT>
__global__ __launch_bounds__(512, 1) void lstm_bwd_rs_chain_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;
    constexpr int NW = 8;
    constexpr int NT = (P + NW - 1) / NW;
    constexpr int DRS = 128 * 2 + 16;
    constexpr int ORS = Hp * 2 + 16;
    static_assert(P <= 32, "one flag word per source in a 32-int row");
    static_assert(!XT || NW * NT > P, "the ride-along tile needs a free tile slot in the last wave");
    __shared__ __attribute__((aligned(16))) unsigned char da_img[2][32 * DRS];
    __shared__ __attribute__((aligned(16))) unsigned char out_img[32 * ORS];
    __shared__ float4 dcs[kRsMaxChains][256];
    __shared__ int lds_flag, lds_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cellw = wave < 4;
    const int n_slots = gridDim.x / P;
    const int set = blockIdx.x % n_slots, p = blockIdx.x / n_slots;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const int C = a.chains < 1 ? 1 : (a.chains > kRsMaxChains ? kRsMaxChains : a.chains);
    const int g0 = set * C;
    if (g0 >= n_groups) return;
    const int Ca = n_groups - g0 < C ? n_groups - g0 : C;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);   // Whh^T packed [Hp][4*Hp]

    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int k = wave + NW * i;
        const int nt = k < P ? (p + 1 + k) % P : 0;
        const int n = 32 * nt + (lane & 31);
        const bf16_t* wsrc = (XT && i == NT - 1 && wave == NW - 1) ? static_cast<const bf16_t*>(a.WihT) + (size_t)(lane & 31) * G4 : WT + (size_t)n * G4;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wreg[i][ks] = *reinterpret_cast<const uint4*>(wsrc + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
    }

    const int erow = (tid & 255) >> 3, jq = tid & 7;
    const int j = 32 * p + 4 * jq;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);
    bf16_t* __restrict__ X = static_cast<bf16_t*>(a.xchg);
    constexpr size_t TILE = 32 * 32;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    int* const tf = a.tflags;
    if (tid == 0) lds_abort = 0;
    __syncthreads();

    unsigned plain_mask = 0;   // bit c: chain c's group verified that it sits on one XCD (a scalar)
    int kpar = 0;              // parity of the chain-step: which dA image it writes
    for (int t = T - 1; t >= 0; --t) {
        for (int c = 0; c < Ca; ++c) {
            const int g = g0 + c;
            const int b = gs * g + erow;
            const bool ok = erow < gs && b < Bp;
            const int bc = ok ? b : Bp - 1;
            int* xtab = a.xcc_tab + (size_t)g * 64;
            const bool plain_handoff = ((plain_mask >> c) & 1u) != 0;
            uint2 sg[4] = {}, sc = make_uint2(0u, 0u), scp = make_uint2(0u, 0u), sdh = make_uint2(0u, 0u);
            if (cellw) {
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) sg[q] = *reinterpret_cast<const uint2*>(g_row + q * Hp);
                sc = *reinterpret_cast<const uint2*>(Cs + (size_t)t * slabH + (size_t)bc * Hp + j);
                if (t > 0) scp = *reinterpret_cast<const uint2*>(Cs + (size_t)(t - 1) * slabH + (size_t)bc * Hp + j);
                if (dhe) sdh = *reinterpret_cast<const uint2*>(dhe + (size_t)t * slabH + (size_t)bc * Hp + j);
                else if (dhl && t == T - 1) sdh = *reinterpret_cast<const uint2*>(dhl + (size_t)bc * Hp + j);
            }
            float dh[4];
            unpack_bf16x4(sdh, dh);
            PL_ST(0);
            if (t + 1 < T) {
                if (cellw) {
                    const int token = t + 2;
                    const int* frow = tf + ((size_t)((t + 1) & 1) * n_groups + g) * P * 32 + (size_t)p * 32;
                    const __amdgpu_buffer_rsrc_t rf = make_rsrc(frow, (unsigned)(P * 4));
                    const bf16_t* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 2));
                    const unsigned o0 = (unsigned)((erow * 32 + 4 * jq) * 2);
                    constexpr unsigned full = P == 32 ? 0xffffffffu : ((1u << P) - 1u);
                    unsigned issued = 0;
                    u32x2 pv[P];
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 1;; ++spin) {
                        int v = 0;
                        if (lane < P)
                            v = plain_handoff ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxNt)
                                              : (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
                        const unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(v == token) & full;
                        const unsigned newly = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask & ~issued));
#pragma unroll
                        for (int s = 0; s < P; ++s)
                            if ((newly >> s) & 1u)
                                pv[s] = plain_handoff ? __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxNt)
                                                      : __builtin_amdgcn_raw_buffer_load_b64(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxSc1);
                        issued |= newly;
                        if (issued == full) break;
                        if ((spin & a.poll_mask) == 0 &&
                            (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 ||
                             __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks)) {
                            if (lane == 0) {
                                __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                lds_abort = 1;
                            }
#pragma unroll
                            for (int s = 0; s < P; ++s)
                                if (!((issued >> s) & 1u)) pv[s] = u32x2{0u, 0u};
                            break;
                        }
                    }
                    PL_ST(1);   // polls + tile loads issued
#pragma unroll
                    for (int s = 0; s < P; ++s) {
                        float f[4];
                        unpack_bf16x4(make_uint2(pv[s][0], pv[s][1]), f);
                        dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
                    }
                }
                if (t == T - 2 && a.xcd_fast) {
                    if (group_on_one_xcd(xtab, P, &lds_flag)) plain_mask |= 1u << c;
                }
            }
            PL_ST(2);   // tiles landed + sums

            unsigned char* const dimg = da_img[kpar];
            if (cellw) {
                float gi[4], gf[4], gg[4], go[4], cv[4], cp[4];
                unpack_bf16x4(sg[0], gi);
                unpack_bf16x4(sg[1], gf);
                unpack_bf16x4(sg[2], gg);
                unpack_bf16x4(sg[3], go);
                unpack_bf16x4(sc, cv);
                unpack_bf16x4(scp, cp);
                float4 dcv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t + 1 < T) dcv = dcs[c][tid];
                float dc_next[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
                float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], cv[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
                dcs[c][tid] = make_float4(dc_next[0], dc_next[1], dc_next[2], dc_next[3]);
                const uint2 pi = pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), pf = pack_bf16x4(daf[0], daf[1], daf[2], daf[3]);
                const uint2 pg = pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), po = pack_bf16x4(dao[0], dao[1], dao[2], dao[3]);
                if (ok && !(XT && a.skip_dA)) {
                    bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                    *reinterpret_cast<uint2*>(go_) = pi;
                    *reinterpret_cast<uint2*>(go_ + Hp) = pf;
                    *reinterpret_cast<uint2*>(go_ + 2 * Hp) = pg;
                    *reinterpret_cast<uint2*>(go_ + 3 * Hp) = po;
                }
                if (t > 0 || XT) {
                    unsigned char* drow = dimg + erow * DRS + jq * 8;
                    *reinterpret_cast<uint2*>(drow) = pi;
                    *reinterpret_cast<uint2*>(drow + 64) = pf;
                    *reinterpret_cast<uint2*>(drow + 128) = pg;
                    *reinterpret_cast<uint2*>(drow + 192) = po;
                }
            }
            if (t == 0 && !XT) continue;   // nobody consumes the partials of step 0 (uniform: no barrier is skipped by part of the workgroup)
            if (t == T - 1 && tid == 0) {
                __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(lds_abort) != 0) return;
            kpar ^= 1;
            PL_ST(3);   // cell + stash stores + dA image + barrier
            uint4 bfr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                bfr[ks] = *reinterpret_cast<const uint4*>(dimg + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            int* const fcol = tf + ((size_t)(t & 1) * n_groups + g) * P * 32 + p;   // + 32 * destination
            const __amdgpu_buffer_rsrc_t rfl = make_rsrc(fcol, (unsigned)(((P - 1) * 32 + 1) * 4));
            auto raise = [&](int nt) {
                if (lane == 0) {
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 32 * 4), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 32 * 4), 0, kAuxSc1);
                }
            };
            int nt_prev = -1;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int k = wave + NW * i;
                if (XT && t == 0) break;
                if (NW * i + NW - 1 >= P && k >= P) break;   // in front of a tile's MFMAs, behind the previous tile's epilogue (tools/isa_mfma_hazard_scan.py)
                const int nt = (p + 1 + k) % P;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                unsigned char* orow = out_img + (lane & 31) * ORS + (32 * nt + 4 * (lane >> 5)) * 2;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<uint2*>(orow + rg * 16) = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
                    const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * nt + 8 * c4) * 2);
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    const unsigned off = (unsigned)(((size_t)nt * P * TILE + cidx * 8) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
                if (nt_prev >= 0) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    raise(nt_prev);
                }
                nt_prev = nt;
            }
            if (nt_prev >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise(nt_prev);
            }
            if (XT && wave == NW - 1) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[NT - 1][ks]),
                                                                  __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                float* xp = a.xpart + (((size_t)t * n_groups + g) * P + p) * 1024;
                const __amdgpu_buffer_rsrc_t rp = make_rsrc(xp, 4096u);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float f0 = acc[4 * rg], f1 = acc[4 * rg + 1], f2 = acc[4 * rg + 2], f3 = acc[4 * rg + 3];
                    u32x4 d;
                    d[0] = __float_as_uint(f0); d[1] = __float_as_uint(f1); d[2] = __float_as_uint(f2); d[3] = __float_as_uint(f3);
                    __builtin_amdgcn_raw_buffer_store_b128(d, rp, (unsigned)(rg * 1024 + lane * 16), 0, 0);
                }
            }
            PL_ST(4);   // tiles + flags
        }
    }
    PL_ST_DUMP(a.stamps);
}

#ifdef PL_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------
// Token form of the reduce-scatter (round 4; EXPERIMENT, compiled with -DPL_EXPERIMENTS only: measured slower than the flag form,
// profiles/r04_token_handoff.txt): the tiles validate THEMSELVES, there is no flag and no drain.
// In the streamed kernel above a tile's way to its consumer is four dependent memory-system round trips -- the producing wave waits
// for the acknowledge of its stores, stores the tile's flag, the consumer's poll finds it, the consumer loads the tile
// (profiles/r03_sweep_phase_stamps_stream.txt: 2.1 of a 4.04-us step).  Here every 8-byte granule of a tile (four bf16 partial sums:
// what ONE consumer lane loads) carries a 2-bit step token in the lowest mantissa bit of its first two values, and the consumer
// simply loads the granules, looks at the tokens and re-loads the tiles that still show the older step: two of the four round
// trips are gone (store -> visible -> load).  What this rests on:
//   * an aligned 8-byte granule written by ONE store instruction (here: half of a lane's 16-byte store) is observed whole -- the
//     MI355X guide's R2 granule; tools/microbench/granule_tear.hip checks exactly this access shape (16-byte stores, 8-byte
//     loads, slot rewritten every round; same XCD with plain stores / nt loads and across XCDs with sc1 both sides, idle and
//     under load): profiles/r04_granule_tear.txt;
//   * tokens: tiles of step t carry pattern A or B by the parity of t >> 1.  A slot (t & 1) is rewritten every second step, its
//     granules go ... -> step t + 2 -> step t, and a consumer reads step t's only after it has seen ALL of step t + 2's (it
//     produced its own step-(t + 1) tiles behind that), so a granule it looks at holds step t + 2's pattern or step t's: one bit
//     would do; the second makes a THIRD state, 0 = "retired": the destination zeroes its incoming tiles after the last two ingests
//     of a launch (steps 1 and 0; 2 x 47 KB per workgroup), so the next launch on this buffer -- whatever its T -- never finds
//     a stale tile that looks current.  The buffer is the token kernel's alone (planner.hip: sweep_xchg_tok, zeroed at
//     allocation and again after a launch that was abandoned on a timeout).
//   * price: the first two of a granule's four values lose their last mantissa bit to the token (the consumer leaves the bit in
//     place: +-1 ulp of bf16 on half of the partial sums, zero mean -- both patterns set one of the two bits); oracle/bf16_emul.py
//     models it bit for bit (token_bits=True).  Not bit-identical to the flag forms above (which stay selectable:
//     PAULE_HIP_BWD_STREAM=1 / 0); gradient-level bars as for every bf16 kernel.
// Tile production (rotated order, eight waves, own-tile epilogue) as in the streamed kernel, minus the counted waits and flags.
__device__ __forceinline__ unsigned rs_token(int t) { return ((t >> 1) & 1) ? 0x00010000u : 0x00000001u; }
constexpr unsigned kRsTokMask = 0x00010001u;

__device__ __forceinline__ unsigned or3(unsigned a, unsigned b, unsigned c) { return a | b | c; }

// EARLY = 1: the cell waves load the tiles that other workgroups produced in their first / second round of the CURRENT step while they
// produce their own second / third tile (the loads fly under the MFMAs and epilogues); at the next step's top only the last round's
// tiles remain to be loaded.  The stash rows of a step (gates, c_t, c_{t-1}, dL/dh from above: 14 KB per workgroup) are fetched one
// step ahead by LDS-DMA, issued by waves 4 .. 7 (which own no cells and idle through the ingest): the cell waves find them in LDS.
template <int KS, int EARLY>
__global__ __launch_bounds__(512, 1) void lstm_bwd_rs_token_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;
    constexpr int NW = 8;
    constexpr int NT = (P + NW - 1) / NW;
    constexpr int DRS = 128 * 2 + 16;
    constexpr int ORS = Hp * 2 + 16;
    constexpr int KE = EARLY ? (8 * (NT - 1) < P ? 8 * (NT - 1) : P) : 0;   // rotated positions 0 .. KE-1 are loaded during the tile phase
    static_assert(P <= 32, "one bit per source in the pending mask");
    __shared__ __attribute__((aligned(16))) unsigned char da_img[2][32 * DRS];
    __shared__ __attribute__((aligned(16))) unsigned char out_img[32 * ORS];
    __shared__ __attribute__((aligned(16))) unsigned char st_img[2][7][2048];   // stash rows of a step: [gate i, f, g, o, c_t, c_{t-1}, dh][32 rows][64 B]
    __shared__ int lds_flag, lds_abort;
    __shared__ int tile_rdy[4][4];   // [cell wave][tile index]: step whose tile image is complete in out_img (waves 4 .. 7 store it)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cellw = wave < 4;
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);   // Whh^T packed [Hp][4*Hp]

    // this wave's tiles: positions k = wave + 8 i of the rotated order, destination (p + 1 + k) mod P
    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int k = wave + NW * i;
        const int nt = k < P ? (p + 1 + k) % P : 0;
        const int n = 32 * nt + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wreg[i][ks] = *reinterpret_cast<const uint4*>(WT + (size_t)n * G4 + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
    }

    const int erow = (tid & 255) >> 3, jq = tid & 7;
    const int j = 32 * p + 4 * jq;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);
    bf16_t* __restrict__ X = static_cast<bf16_t*>(a.xchg);
    constexpr size_t TILE = 32 * 32;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    const unsigned st_lds = (unsigned)(uintptr_t)(rs_lds_ptr_t)&st_img[0][0][0];
    if (wave == 0 && lane == 0) lds_abort = 0;
    if (tid < 16) (&tile_rdy[0][0])[tid] = -1;
    __syncthreads();

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + erow;
        const bool ok = erow < gs && b < Bp;
        float dc_next[4] = {0.f, 0.f, 0.f, 0.f};
        int* xtab = a.xcc_tab + (size_t)g * 64;
        int seq = 0;       // sequence number of the tile phase (the LDS hand-over of the cell waves' tile images to the storing waves)
        int plain_i = 0;   // 1 once the group has verified that it sits on one XCD; read through readfirstlane: a SCALAR wherever it is used
        // this destination's incoming tiles of a slot, all P sources: one contiguous block of P x 2 KB
        const bf16_t* const xin = X + (size_t)g * grp_stride + (size_t)p * P * TILE;
        auto retire = [&](int slot) {
            const __amdgpu_buffer_rsrc_t rz = make_rsrc(xin + (size_t)slot * slot_stride, (unsigned)(P * TILE * 2));
            const u32x4 z = {0u, 0u, 0u, 0u};
            const bool plain_handoff = __builtin_amdgcn_readfirstlane(plain_i) != 0;
            for (int e0 = 0; e0 < P * 128; e0 += 512) {   // uniform trip count; chunks beyond the block: dropped by the range check
                const unsigned zo = e0 + tid < P * 128 ? (unsigned)((e0 + tid) * 16) : 0xfffffff0u;
                if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(z, rz, zo, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(z, rz, zo, 0, kAuxSc1);
            }
        };
        // stash rows of step ts -> LDS image ts & 1, by waves 4 .. 7: 14 pieces of 1 KB (an array's rows 0 .. 15 / 16 .. 31, 64 B a row);
        // wave 4 + w takes pieces w, w + 4, w + 8, w + 12: a lane's byte offset inside a time slab is the same for every step
        unsigned pf_off[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int q = (wave & 3) + 4 * qi, arr = q >> 1, row = 16 * (q & 1) + (lane >> 2);
            const int br = gs * g + row;
            const int bcl = (row < gs && br < Bp) ? br : Bp - 1;
            pf_off[qi] = arr < 4 ? (unsigned)(((size_t)bcl * G4 + (size_t)arr * Hp + 32 * p + 8 * (lane & 3)) * 2)
                                 : (unsigned)(((size_t)bcl * Hp + 32 * p + 8 * (lane & 3)) * 2);
        }
        auto prefetch = [&](int ts) {
            const bf16_t* const gb = G + (size_t)ts * slabG;
            const bf16_t* const cb = Cs + (size_t)ts * slabH;
            const bf16_t* const db = dhe ? dhe + (size_t)ts * slabH : ((dhl && ts == T - 1) ? dhl : nullptr);
            const unsigned dst0 = st_lds + (unsigned)((ts & 1) * 7 * 2048);
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = (wave & 3) + 4 * qi, arr = q >> 1;   // wave-uniform
                if (q < 14) {
                    const bf16_t* base = arr < 4 ? gb : (arr == 4 ? cb : (arr == 5 ? (ts > 0 ? cb - slabH : nullptr) : db));
                    if (base) glds16(base, pf_off[qi], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + (unsigned)(q * 1024))));
                }
            }
        };
        if (!cellw) {
            prefetch(T - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rx0 = make_rsrc(xin, (unsigned)(P * TILE * 2));
        const __amdgpu_buffer_rsrc_t rx1 = make_rsrc(xin + slot_stride, (unsigned)(P * TILE * 2));
        const unsigned o0 = (unsigned)((erow * 32 + 4 * jq) * 2);
        u32x2 pv[P];   // by rotated position k: the tile of source (p - 1 - k) mod P -- what that source produces in its round k / 8
#pragma unroll
        for (int k = 0; k < P; ++k) pv[k] = u32x2{0u, 0u};
        // (p is laundered through an empty asm so that the 23 offsets are recomputed -- three scalar instructions -- instead of being kept
        // in scalar registers across the step loop, where they spill)
        auto src_off = [&](int k) { int pp = p; asm volatile("" : "+s"(pp)); int sidx = pp - 1 - k; sidx = sidx < 0 ? sidx + P : sidx; return (unsigned)(sidx * (int)(TILE * 2)); };
        // tiles at rotated positions k0 .. k1-1 of a slot -> pv (one scalar branch for the cache policy, not one per load)
        auto tile_loads = [&](int slot, auto k0c, auto k1c) {
            constexpr int k0 = decltype(k0c)::value, k1 = decltype(k1c)::value;
            const __amdgpu_buffer_rsrc_t rx = slot ? rx1 : rx0;
            if (__builtin_amdgcn_readfirstlane(plain_i) != 0) {
#pragma unroll
                for (int k = k0; k < k1; ++k) pv[k] = __builtin_amdgcn_raw_buffer_load_b64(rx, o0, src_off(k), kAuxNt);
            } else {
#pragma unroll
                for (int k = k0; k < k1; ++k) pv[k] = __builtin_amdgcn_raw_buffer_load_b64(rx, o0, src_off(k), kAuxSc1);
            }
        };
        auto tile_reloads = [&](int slot, unsigned pending) {
            const __amdgpu_buffer_rsrc_t rx = slot ? rx1 : rx0;
            if (__builtin_amdgcn_readfirstlane(plain_i) != 0) {
#pragma unroll
                for (int k = 0; k < P; ++k)
                    if ((pending >> k) & 1u) pv[k] = __builtin_amdgcn_raw_buffer_load_b64(rx, o0, src_off(k), kAuxNt);
            } else {
#pragma unroll
                for (int k = 0; k < P; ++k)
                    if ((pending >> k) & 1u) pv[k] = __builtin_amdgcn_raw_buffer_load_b64(rx, o0, src_off(k), kAuxSc1);
            }
        };

        for (int t = T - 1; t >= 0; --t) {
            if (!cellw && t > 0) prefetch(t - 1);   // next step's stash rows, under this step's ingest and cell update
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
            PL_ST(0);
            if (t + 1 < T) {
                if (cellw) {
                    // The P tiles of step t + 1 for this workgroup's units.  A lane's granule of a tile = the four partial sums of its
                    // units; a tile is taken when all 64 granules of the wave's piece show step t + 1's token, otherwise loaded again.
                    const int slot = (t + 1) & 1;
                    const unsigned E = rs_token(t + 1);
                    tile_loads(slot, std::integral_constant<int, KE>{}, std::integral_constant<int, P>{});
                    // fast check: the slot held step t + 3's tiles before (the OTHER pattern), so "no granule shows the other pattern's
                    // bit" is "all arrived" -- except in a launch's first two ingests, when the slot was retired (no bit set at all)
                    bool all_in = false;
                    if (t + 3 < T) {
                        unsigned acc_or = 0u;
#pragma unroll
                        for (int k = 0; k + 1 < P; k += 2) acc_or = or3(acc_or, pv[k][0], pv[k + 1][0]);
                        if (P & 1) acc_or |= pv[P - 1][0];
                        all_in = __builtin_amdgcn_ballot_w64((acc_or & (kRsTokMask ^ E)) != 0u) == 0ull;
                    }
                    if (!all_in) {
                        unsigned pending = 0u;
#pragma unroll
                        for (int k = 0; k < P; ++k)
                            if (__builtin_amdgcn_ballot_w64((pv[k][0] & kRsTokMask) != E) != 0ull) pending |= 1u << k;
#if defined(PL_STAMPS)
                        if (pending) st_acc[5] += 100;
#endif
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        for (unsigned spin = 1; pending != 0u; ++spin) {
                            asm volatile("" ::: "memory");   // every sweep re-issues its loads
#if defined(PL_STAMPS)
                            st_acc[6] += 100 * __builtin_popcount(pending);
#endif
                            tile_reloads(slot, pending);
#pragma unroll
                            for (int k = 0; k < P; ++k)
                                if ((pending >> k) & 1u)
                                    if (__builtin_amdgcn_ballot_w64((pv[k][0] & kRsTokMask) != E) == 0ull) pending &= ~(1u << k);
                            if (pending == 0u) break;
                            if ((spin & a.poll_mask) == 0 &&
                                (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 ||
                                 __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks)) {
                                if (lane == 0) {
                                    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    lds_abort = 1;
                                }
#pragma unroll
                                for (int k = 0; k < P; ++k)
                                    if ((pending >> k) & 1u) pv[k] = u32x2{0u, 0u};
                                break;
                            }
                        }
                    }
                    PL_ST(1);   // tile loads + token checks
#pragma unroll
                    for (int k = 0; k < P; ++k) {
                        float f[4];
                        unpack_bf16x4(make_uint2(pv[k][0], pv[k][1]), f);
                        dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
                    }
                }
                if (t == T - 2 && a.xcd_fast) plain_i = __builtin_amdgcn_readfirstlane((int)group_on_one_xcd(xtab, P, &lds_flag));
            }
            PL_ST(2);   // sums

            unsigned char* const dimg = da_img[t & 1];
            if (cellw) {
                const unsigned char* sb = &st_img[t & 1][0][0] + erow * 64 + jq * 8;
                float gi[4], gf[4], gg[4], go[4], c[4], cp[4], dx[4];
                unpack_bf16x4(*reinterpret_cast<const uint2*>(sb), gi);
                unpack_bf16x4(*reinterpret_cast<const uint2*>(sb + 2048), gf);
                unpack_bf16x4(*reinterpret_cast<const uint2*>(sb + 2 * 2048), gg);
                unpack_bf16x4(*reinterpret_cast<const uint2*>(sb + 3 * 2048), go);
                unpack_bf16x4(*reinterpret_cast<const uint2*>(sb + 4 * 2048), c);
                unpack_bf16x4(t > 0 ? *reinterpret_cast<const uint2*>(sb + 5 * 2048) : make_uint2(0u, 0u), cp);
                unpack_bf16x4((dhe || (dhl && t == T - 1)) ? *reinterpret_cast<const uint2*>(sb + 6 * 2048) : make_uint2(0u, 0u), dx);
                // dL/dh from above first, then the partial sums in the order they were added above (an f32 sum: the order is part of the result)
#pragma unroll
                for (int u = 0; u < 4; ++u) dh[u] = dx[u] + dh[u];
                float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], c[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
                const uint2 pi = pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), pf = pack_bf16x4(daf[0], daf[1], daf[2], daf[3]);
                const uint2 pg = pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), po = pack_bf16x4(dao[0], dao[1], dao[2], dao[3]);
                if (t == 0) {   // dA_0 (no image, no tile phase): the cell waves store it themselves; rows that do not exist: dropped by the range check
                    const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
                    const unsigned go_ = ok ? (unsigned)(((size_t)b * G4 + j) * 2) : 0x80000000u;
                    st8_handoff(rg, go_, pi, true);
                    st8_handoff(rg, go_ + (unsigned)(Hp * 2), pf, true);
                    st8_handoff(rg, go_ + (unsigned)(2 * Hp * 2), pg, true);
                    st8_handoff(rg, go_ + (unsigned)(3 * Hp * 2), po, true);
                }
                if (t > 0) {
                    unsigned char* drow = dimg + erow * DRS + jq * 8;
                    *reinterpret_cast<uint2*>(drow) = pi;
                    *reinterpret_cast<uint2*>(drow + 64) = pf;
                    *reinterpret_cast<uint2*>(drow + 128) = pg;
                    *reinterpret_cast<uint2*>(drow + 192) = po;
                }
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == T - 1 && wave == 0) {   // this workgroup's XCD, in place before ANY of its tiles (they are stored behind the barrier below)
                if (lane == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!cellw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stash rows of step t - 1 have landed in LDS
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(lds_abort) != 0) return;
            PL_ST(3);   // cell + stash stores + dA image + barrier
            if (t == 1 && T > 2) retire(0);   // step 2's tiles (slot 0) were this launch's last use of that slot: every wave has summed them
            if (!cellw) {
                // dA_t -> the gate stash, in place, from the image: the cell waves issue NO global store in a step (vmcnt retires in order:
                // their tile loads would wait for the acknowledge of every older store of theirs, ~2 us -- profiles/r04_token_stamps_v2.txt)
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = (tid & 255) + 256 * q, row = e >> 4, ch = e & 15;   // 16-byte chunk: gate ch >> 2, units 8 (ch & 3) .. + 7
                    const uint4 v = *reinterpret_cast<const uint4*>(dimg + row * DRS + ch * 16);
                    const int br = gs * g + row;
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    __builtin_amdgcn_raw_buffer_store_b128(d, rg, (row < gs && br < Bp) ? (unsigned)(((size_t)br * G4 + (size_t)(ch >> 2) * Hp + 32 * p + 8 * (ch & 3)) * 2) : 0x80000000u, 0, 0);
                }
            }
            uint4 bfr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                bfr[ks] = *reinterpret_cast<const uint4*>(dimg + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            const unsigned Et = rs_token(t);
            const bool plain_handoff = __builtin_amdgcn_readfirstlane(plain_i) != 0;
            ++seq;
            auto store_tile = [&](int ntile) {   // a tile's image, read back by rows: 2 KB contiguous, two 16-byte stores per lane
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
                    const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * ntile + 8 * c4) * 2);
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    const unsigned off = (unsigned)(((size_t)ntile * P * TILE + cidx * 8) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
            };
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (EARLY && i >= 1 && cellw) {
                    // what the others produced in their round i - 1 of THIS step is on its way or there: loaded now, checked at the next step's top
                    if (i == 1) tile_loads(t & 1, std::integral_constant<int, 0>{}, std::integral_constant<int, (KE < 8 ? KE : 8)>{});
                    if (i == 2) tile_loads(t & 1, std::integral_constant<int, (KE < 8 ? KE : 8)>{}, std::integral_constant<int, (KE < 16 ? KE : 16)>{});
                    if (i == 3) tile_loads(t & 1, std::integral_constant<int, (KE < 16 ? KE : 16)>{}, std::integral_constant<int, (KE < 24 ? KE : 24)>{});
                }
                const int k = wave + NW * i;
                const bool have = !(NW * i + NW - 1 >= P && k >= P);   // a compile-time fact for all but a wave's last tile
                if (have) {
                    const int nt = (p + 1 + k) % P;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[i][ks]),
                                                                      __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                    // a lane's four consecutive values of one batch row = one consumer granule: its first two values carry the step token
                    unsigned char* orow = out_img + (lane & 31) * ORS + (32 * nt + 4 * (lane >> 5)) * 2;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        uint2 pk = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
                        pk.x = (pk.x & ~kRsTokMask) | Et;
                        *reinterpret_cast<uint2*>(orow + rg * 16) = pk;
                    }
                    if (cellw) {   // the image is complete (release: the flag follows the image's LDS writes): wave + 4 stores it
                        if (lane == 0) __hip_atomic_store(&tile_rdy[wave][i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        store_tile(nt);
                    }
                }
                if (!cellw) {
                    // the SIMD's other wave's tile of this round (it owns cells and issues no global store), as soon as its image is complete
                    const int kp = (wave - 4) + NW * i;
                    if (!(NW * i + 3 >= P && kp >= P)) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        bool got = true;
                        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&tile_rdy[wave - 4][i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq) {
                            __builtin_amdgcn_s_sleep(1);
                            if (__builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks) {   // never in a healthy launch: give up loudly
                                if (lane == 0) {
                                    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    lds_abort = 1;
                                }
                                got = false;
                                break;
                            }
                        }
                        if (got) store_tile((p + 1 + kp) % P);
                    }
                }
            }
            PL_ST(4);   // tiles
        }
        __syncthreads();   // every wave has summed step 1's tiles; and nobody writes the next group's dA image / stash image while a wave still reads this one's
        if (T > 1 && __builtin_amdgcn_readfirstlane(lds_abort) == 0) retire(1);
    }
    PL_ST_DUMP(a.stamps);
}

#endif  // PL_EXPERIMENTS

#define PL_SWEEP_KS_LIST(X) X(2) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(46) X(48)

size_t lstm_rs_exchange_bytes(int Hp, int Bp) {
    const size_t groups = (Bp + 7) / 8, P = Hp / 32;   // upper bound on the group count (groups hold >= 8 rows)
    return 2 * groups * P * P * 32 * 32 * 2;
}

// fixed-order sum of the P partial input-gradient tiles of every (step, group): dX f32 [T][Bp][32]
__global__ __launch_bounds__(256) void dx_reduce_kernel(const float* __restrict__ xpart, int P, int n_groups, int Bp, float* __restrict__ dX) {
    const int tg = blockIdx.x, t = tg / n_groups, g = tg % n_groups;
    const int rg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4* src = reinterpret_cast<const float4*>(xpart + (size_t)tg * P * 1024) + rg * 64 + lane;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P == 23) {   // the ride-along's shape: all 23 loads in flight, summed in the same order
        float4 v[23];
#pragma unroll
        for (int p = 0; p < 23; ++p) v[p] = src[(size_t)p * 256];
#pragma unroll
        for (int p = 0; p < 23; ++p) { s.x += v[p].x; s.y += v[p].y; s.z += v[p].z; s.w += v[p].w; }
    } else {
        for (int p = 0; p < P; ++p) {
            const float4 v = src[(size_t)p * 256];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    const int b = 32 * g + (lane & 31), n = 8 * rg + 4 * (lane >> 5);
    if (b < Bp) *reinterpret_cast<float4*>(dX + ((size_t)t * Bp + b) * 32 + n) = s;
}
bool lstm_rs_ride_along_supported(int Hp, int in_p) { return in_p == 32 && Hp == 16 * 46; }   // P = 23: one free slot on 8 waves x 3 tiles
size_t lstm_rs_xpart_bytes(int Hp, int Bp, int T) { return (size_t)T * ((Bp + 31) / 32) * (Hp / 32) * 4096; }
void launch_dx_reduce(hipStream_t stream, const float* xpart, int Hp, int Bp, int T, float* dX) {
    const int ng = (Bp + 31) / 32;
    hipLaunchKernelGGL(dx_reduce_kernel, dim3(T * ng), dim3(256), 0, stream, xpart, Hp / 32, ng, Bp, dX);
}

#ifdef PL_EXPERIMENTS   // the token form and the LDS-DMA variants of the streamed form (profiles/r04_token_handoff.txt)
#define PL_CASE_EXPERIMENTS(K)                                                                        \
        if (a.token_handoff && K <= 64) {                                                                 \
            if (a.token_handoff == 2) hipLaunchKernelGGL((lstm_bwd_rs_token_kernel<(K <= 64 ? K : 2), 0>), dim3(grid), dim3(512), 0, stream, a); \
            else hipLaunchKernelGGL((lstm_bwd_rs_token_kernel<(K <= 64 ? K : 2), 1>), dim3(grid), dim3(512), 0, stream, a);               \
            return;                                                                                       \
        }                                                                                                 \
        if (a.tflags && a.bwd_waves != 4 && K <= 64 && ((a.stash_via_lds >> 3) & 3)) {                    \
            const int dv = (a.stash_via_lds >> 3) & 3;                                                    \
            if (dv == 3) hipLaunchKernelGGL((lstm_bwd_rs_stream_kernel<(K <= 64 ? K : 2), 3>), dim3(grid), dim3(512), 0, stream, a); \
            else if (dv == 2) hipLaunchKernelGGL((lstm_bwd_rs_stream_kernel<(K <= 64 ? K : 2), 2>), dim3(grid), dim3(512), 0, stream, a); \
            else hipLaunchKernelGGL((lstm_bwd_rs_stream_kernel<(K <= 64 ? K : 2), 1>), dim3(grid), dim3(512), 0, stream, a); \
            return;                                                                                       \
        }
#else
#define PL_CASE_EXPERIMENTS(K)
#endif

void launch_lstm_bwd_rs_sweep(hipStream_t stream, int Hp, int grid, const LstmSweepArgs& a) {
#define PL_CASE(K)                                                                                        \
    if (Hp == 16 * K) {                                                                                   \
        PL_CASE_EXPERIMENTS(K)                                                                            \
        if (a.tflags && a.bwd_waves != 4 && K == 46 && a.chains > 0 && a.xpart)                           \
            hipLaunchKernelGGL((lstm_bwd_rs_chain_kernel<46, 1>), dim3(grid), dim3(512), 0, stream, a);   \
        else if (a.tflags && a.bwd_waves != 4 && K == 46 && a.chains > 0)                                 \
            hipLaunchKernelGGL((lstm_bwd_rs_chain_kernel<46, 0>), dim3(grid), dim3(512), 0, stream, a);   \
        else if (a.tflags && a.bwd_waves != 4 && K == 46 && a.xpart)                                      \
            hipLaunchKernelGGL((lstm_bwd_rs_stream_kernel<46, 0, 1>), dim3(grid + a.n_pf), dim3(512), 0, stream, a);   \
        else if (a.tflags && a.bwd_waves != 4 && K <= 64)                                                 \
            hipLaunchKernelGGL((lstm_bwd_rs_stream_kernel<(K <= 64 ? K : 2), 0>), dim3(grid + a.n_pf), dim3(512), 0, stream, a); \
        else if (a.bwd_waves == 4)                                                                        \
            hipLaunchKernelGGL((lstm_bwd_rs_sweep_kernel<K, 4>), dim3(grid), dim3(256), 0, stream, a);    \
        else                                                                                              \
            hipLaunchKernelGGL((lstm_bwd_rs_sweep_kernel<K, 8>), dim3(grid), dim3(512), 0, stream, a);    \
        return;                                                                                           \
    }
    PL_SWEEP_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
