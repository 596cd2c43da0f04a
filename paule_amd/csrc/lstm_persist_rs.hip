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
template <int KS>
__global__ __launch_bounds__(512, 1) void lstm_bwd_rs_stream_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;
    constexpr int NW = 8;
    constexpr int NT = (P + NW - 1) / NW;
    constexpr int DRS = 128 * 2 + 16;
    constexpr int ORS = Hp * 2 + 16;
    static_assert(P <= 32, "one flag word per source in a 32-int row");
    __shared__ __attribute__((aligned(16))) unsigned char da_img[2][32 * DRS];
    __shared__ __attribute__((aligned(16))) unsigned char out_img[32 * ORS];
    __shared__ int lds_flag, lds_abort;

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

        for (int t = T - 1; t >= 0; --t) {
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
                if (t > 0) {
                    unsigned char* drow = dimg + erow * DRS + jq * 8;
                    *reinterpret_cast<uint2*>(drow) = pi;
                    *reinterpret_cast<uint2*>(drow + 64) = pf;
                    *reinterpret_cast<uint2*>(drow + 128) = pg;
                    *reinterpret_cast<uint2*>(drow + 192) = po;
                }
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == T - 1 && tid == 0) {   // this workgroup's XCD, in place before ANY of its flags (they are raised behind the barrier below)
                __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
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
                if (NW * i + NW - 1 >= P && k >= P) break;   // a compile-time fact for all but a wave's last tile
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
            PL_ST(4);   // tiles + flags
        }
        __syncthreads();   // a workgroup that sweeps several groups in turn: nobody writes the next group's dA image while a wave still reads this one's
    }
    PL_ST_DUMP(a.stamps);
}

#define PL_SWEEP_KS_LIST(X) X(2) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(46) X(48)

size_t lstm_rs_exchange_bytes(int Hp, int Bp) {
    const size_t groups = (Bp + 7) / 8, P = Hp / 32;   // upper bound on the group count (groups hold >= 8 rows)
    return 2 * groups * P * P * 32 * 32 * 2;
}

void launch_lstm_bwd_rs_sweep(hipStream_t stream, int Hp, int grid, const LstmSweepArgs& a) {
#define PL_CASE(K)                                                                                        \
    if (Hp == 16 * K) {                                                                                   \
        if (a.tflags && a.bwd_waves != 4 && K <= 64)                                                      \
            hipLaunchKernelGGL(lstm_bwd_rs_stream_kernel<(K <= 64 ? K : 2)>, dim3(grid), dim3(512), 0, stream, a); \
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
