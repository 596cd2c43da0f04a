// Persistent LSTM layer sweeps (bf16): ONE launch runs all T time steps of one layer.
//
// Why: a launch per time step has to pull its W_hh slice (94-188 KB per CU) out of L2 again every step,
// at the ~70 GB/s a single CU ingests, plus a kernel boundary per step (measured 12.4 us per step).  Here
// every workgroup keeps its W_hh slice in REGISTERS for the whole sweep (184 VGPRs per lane at H = 720) and
// only the recurrent operand moves.
//
// Decomposition.  Utterances are independent, so the batch is cut into groups of 32 rows; a group is served
// by P = Hp / 32 workgroups (one per CU), workgroup p owning hidden units [32p, 32p + 32) for all four
// gates.  grid = P x (resident groups); further groups are processed in sequence by the same workgroups.
//   forward : wave w owns 8 hidden units x 4 gates = 32 gate rows (MFMA 32x32x16 A operand, rows ordered
//             [i(8) f(8) g(8) o(8)] so that a lane ends up with all four gates of 4 units of one batch
//             row); B operand = h_{t-1} of the group's 32 batch rows, staged once per step in LDS.
//             The running cell state c never leaves registers.
//   backward: contraction over all 4*Hp gate rows; wave w takes gate block w (K split over the waves),
//             A operand = W_hh^T rows of the 32 hidden units; the four partial 32 x 32 tiles are reduced
//             through LDS, then each thread runs the cell backward for (1 batch row x 4 hidden units).
//             The running dc never leaves registers.
//
// In-launch exchange (all P workgroups of a group need the whole h_t / dA_t of the group; sweep_common.h): the producers
// store their slice WRITE-THROUGH (sc1), every storing wave drains its hand-off stores with a counted s_waitcnt (the
// stash stores issued behind them stay in flight), the workgroup barriers, ONE lane raises the workgroup's arrival
// flag for the step.  Consumers: one wave polls the P flags of the step with one sc1 load per poll (lane i reads flag
// i), the workgroup barriers, then EVERY load of the handed-off bytes is an sc1 buffer load (L1 bypass) -- the form of
// the MI355X guide's hand-off table, row 1.  Flags are zeroed before every launch by a kernel of ours (sc1 stores;
// hipMemset nodes are not ordered under back-to-back graph replay).  Every spin is bounded (s_memrealtime); on a
// timeout a status word is set and every workgroup leaves.  Results do not depend on placement or dispatch order;
// a group whose workgroups find themselves on one XCD may hand off through the shared L2 (verified at run time).
#include "sweep_common.h"

namespace pl {

// ---------------------------------------------------------------------------------------------------
// forward sweep
// ---------------------------------------------------------------------------------------------------
// KSX > 0: the input projection W_ih x_t + b is fused (layers with a narrow input: CP 32, mel 64 columns): the group's
// x_t tile rides along as KSX extra k-steps, the bias initialises the accumulator, and the batched projection GEMM
// with its [T][Bp][4Hp] write + read-back disappears.  KSX = 0: G holds the precomputed projection.
template <int KS, int KSX>   // KS = Hp / 16, KSX = in_p / 16 MFMA k-steps
__global__ __launch_bounds__(256, 1) void lstm_fwd_sweep_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int XRS = KSX * 32 + 16;           // row stride of the x_t image [32 rows][in_p] bf16
    constexpr int ROWB = Hp * 2;                 // bytes of one h row
    constexpr int RS = ROWB + 16;                // LDS row stride: odd number of 16-byte chunks -> conflict-free b128 reads
    constexpr int KSH = KS / 2;                  // k-steps per K half
    constexpr int CH = Hp / 16;                  // 16-byte chunks per row half
    constexpr int NLH = (32 * CH + 255) / 256;   // loads per thread per half
    constexpr int PF = 6;                        // B-fragment read-ahead
    static_assert(KS % 2 == 0, "Hp is a multiple of 32");
    constexpr int HRS = 64 + 16;                 // row stride of the outgoing h tile [32 rows][32 units] bf16
    __shared__ __attribute__((aligned(16))) unsigned char himg[32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char hst[6 * 32 * HRS];   // [h, gates i f g o, c][32 rows][32 units]
    __shared__ __attribute__((aligned(16))) unsigned char ximg[KSX ? 32 * XRS : 16];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = Hp / 32;
    const int n_res = gridDim.x / P;             // resident groups
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;                 // batch rows per group (<= 32; the MFMA tile stays 32 wide)
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(a.W);

    // weights -> registers: A-operand row (lane & 31) = gate (row >> 3), unit 32p + 8 wave + (row & 7)
    uint4 wreg[KS];
    {
        const int ar = lane & 31;
        const bf16_t* wrow = W + (size_t)((ar >> 3) * Hp + 32 * p + 8 * wave + (ar & 7)) * Hp + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wreg[ks] = *reinterpret_cast<const uint4*>(wrow + 16 * ks);
    }
    const int bl = lane & 31, hh = lane >> 5;
    uint4 wx[KSX ? KSX : 1];
    float bias_r[16];
    if constexpr (KSX > 0) {
        constexpr int INP = 16 * KSX;
        const int ar = lane & 31;
        const bf16_t* xrow = static_cast<const bf16_t*>(a.Wih) + (size_t)((ar >> 3) * Hp + 32 * p + 8 * wave + (ar & 7)) * INP + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) wx[ks] = *reinterpret_cast<const uint4*>(xrow + 16 * ks);
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = a.bias[(r >> 2) * Hp + 32 * p + 8 * wave + 4 * hh + (r & 3)];
    }

    const int j = 32 * p + 8 * wave + 4 * hh;     // this lane's 4 hidden units
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(a.h);
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(a.c);

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + bl;
        const bool ok = bl < gs && b < Bp;
        const int bc = ok ? b : Bp - 1;
        float c_state[4] = {0.f, 0.f, 0.f, 0.f};
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;   // flags [group][step][flag_stride]
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        for (int t = 0; t < T; ++t) {
            // input projection of this step: either precomputed (G, written by the preceding GEMM launch) or fused: both are
            // plain loads of data from earlier launches, issued before the wait
            uint2 gx[4] = {};
            f32x16 acc;
            if constexpr (KSX > 0) {
                constexpr int INP = 16 * KSX, XC = INP / 8;     // 16-byte chunks per x row
                if (tid < 32 * XC) {
                    const int row = tid / XC, c = tid % XC;
                    int rb = gs * g + row;
                    rb = rb < Bp ? rb : Bp - 1;
                    const uint4 xv = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.x_in) +
                                                                     ((size_t)t * Bp + rb) * INP + c * 8);
                    *reinterpret_cast<uint4*>(ximg + row * XRS + c * 16) = xv;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bias_r[r];
                if (t == 0) __syncthreads();   // later steps: the barrier of the arrival wait publishes the x image
            } else {
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) gx[q] = *reinterpret_cast<const uint2*>(g_row + q * Hp);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            PL_ST(0);   // top of step (prefetch issue)
            if (t > 0) {
                if (!wait_arrivals(cnt + (size_t)(t - 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == 1 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);   // waiting for the group's arrivals
                // h_{t-1} of the group's 32 batch rows -> LDS (sc1 loads: handed-off bytes), in two K halves: all
                // loads are issued at once, the second half is still in flight while the first half multiplies
                const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t - 1) * slabH, (unsigned)(slabH * 2));
                uint4 v[2][NLH];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int i = 0; i < NLH; ++i) {
                        const int q = tid + 256 * i;
                        const int row = q / CH, c = q % CH + hf * CH;
                        int rb = gs * g + row;
                        rb = rb < Bp ? rb : Bp - 1;
                        // rows beyond the group are never used (their MFMA columns are discarded): skip their traffic
                        v[hf][i] = (q < 32 * CH && row < gs) ? ld16_handoff(rh, (unsigned)(rb * ROWB + c * 16), plain_handoff) : make_uint4(0, 0, 0, 0);
                    }
                const unsigned char* bsrc = himg + bl * RS + hh * 16;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int i = 0; i < NLH; ++i) {
                        const int q = tid + 256 * i;
                        if (q < 32 * CH) *reinterpret_cast<uint4*>(himg + (q / CH) * RS + (q % CH + hf * CH) * 16) = v[hf][i];
                    }
                    __syncthreads();
                    if (hf == 0) PL_ST(2);   // first half of the h tile landed
                    // B fragments are read PF k-steps ahead of the MFMA that consumes them; sched_barrier pins that order
                    // (left alone, the scheduler serialises ds_read -> wait -> MFMA on one register)
                    uint4 bq[PF];
#pragma unroll
                    for (int i = 0; i < PF; ++i)
                        if (i < KSH) bq[i] = *reinterpret_cast<const uint4*>(bsrc + (hf * KSH + i) * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[hf * KSH + ks]),
                                                                      __builtin_bit_cast(bf16x8, bq[ks % PF]), acc, 0, 0, 0);
                        if (ks + PF < KSH) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (hf * KSH + ks + PF) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }

            if constexpr (KSX > 0) {
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) {
                    const uint4 xb = *reinterpret_cast<const uint4*>(ximg + bl * XRS + ks * 32 + hh * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wx[ks]),
                                                                  __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
                }
            }
            PL_ST(3);   // MFMA chain
            // cell update: acc[4 * gate + unit]
            float gxi[4], gxf[4], gxg[4], gxo[4];
            unpack_bf16x4(gx[0], gxi);
            unpack_bf16x4(gx[1], gxf);
            unpack_bf16x4(gx[2], gxg);
            unpack_bf16x4(gx[3], gxo);
            float vi[4], vf[4], vg[4], vo[4], vc[4], vh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vi[u] = sigmoid_fast(acc[u] + gxi[u]);
                vf[u] = sigmoid_fast(acc[4 + u] + gxf[u]);
                vg[u] = tanh_fast(acc[8 + u] + gxg[u]);
                vo[u] = sigmoid_fast(acc[12 + u] + gxo[u]);
                c_state[u] = cell_c(vf[u], c_state[u], vi[u], vg[u]);
                vc[u] = c_state[u];
                vh[u] = vo[u] * tanh_fast(vc[u]);
            }
            if (t == 0 && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // hand-off first: the workgroup's h tile (32 rows x 32 units) goes through LDS so that each row's 64 bytes
            // leave as ONE instruction's contiguous piece (4 lanes x 16 B): whole sectors, which the consumers' reads
            // of the shared L2 need (16-byte pieces from four different waves read back at half the rate)
            *reinterpret_cast<uint2*>(hst + bl * HRS + (8 * wave + 4 * hh) * 2) = pack_bf16x4(vh[0], vh[1], vh[2], vh[3]);
            const bool stash_lds = (a.stash_via_lds & 1) != 0;
            if (stash_lds) {   // the five stash arrays leave through LDS too: whole 64-byte row pieces instead of 8-byte scatters
                unsigned char* o = hst + 32 * HRS + bl * HRS + (8 * wave + 4 * hh) * 2;
                *reinterpret_cast<uint2*>(o) = pack_bf16x4(vi[0], vi[1], vi[2], vi[3]);
                *reinterpret_cast<uint2*>(o + 32 * HRS) = pack_bf16x4(vf[0], vf[1], vf[2], vf[3]);
                *reinterpret_cast<uint2*>(o + 2 * 32 * HRS) = pack_bf16x4(vg[0], vg[1], vg[2], vg[3]);
                *reinterpret_cast<uint2*>(o + 3 * 32 * HRS) = pack_bf16x4(vo[0], vo[1], vo[2], vo[3]);
                *reinterpret_cast<uint2*>(o + 4 * 32 * HRS) = pack_bf16x4(vc[0], vc[1], vc[2], vc[3]);
            }
            __syncthreads();
            if (tid < 128) {
                const int row = tid >> 2, qt = tid & 3;
                const int rb = gs * g + row;
                if (row < gs && rb < Bp) {
                    const uint4 hv = *reinterpret_cast<const uint4*>(hst + row * HRS + qt * 16);
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
                    u32x4 d;
                    d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
                    const unsigned off = (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
            }
            asm volatile("" ::: "memory");   // keep the stash stores behind it
            if (stash_lds) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int e = tid + 256 * i;   // piece: array e / 128, row (e % 128) / 4, quarter e % 4
                    if (e < 640) {
                        const int arr = e >> 7, row = (e & 127) >> 2, qt = e & 3, rb = gs * g + row;
                        if (row < gs && rb < Bp) {
                            const uint4 sv = *reinterpret_cast<const uint4*>(hst + (arr + 1) * 32 * HRS + row * HRS + qt * 16);
                            bf16_t* dst = arr < 4 ? G + (size_t)t * slabG + (size_t)rb * G4 + arr * Hp + 32 * p + 8 * qt
                                                  : Cs + (size_t)t * slabH + (size_t)rb * Hp + 32 * p + 8 * qt;
                            *reinterpret_cast<uint4*>(dst) = sv;
                        }
                    }
                }
                PL_ST(4);
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // the hand-off store is older than the (at most 3) stash stores
                PL_ST(5);
                publish<3>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
                PL_ST(6);
                continue;
            }
            if (ok) {
                bf16_t* go = G + (size_t)t * slabG + (size_t)b * G4 + j;
                *reinterpret_cast<uint2*>(go) = pack_bf16x4(vi[0], vi[1], vi[2], vi[3]);
                *reinterpret_cast<uint2*>(go + Hp) = pack_bf16x4(vf[0], vf[1], vf[2], vf[3]);
                *reinterpret_cast<uint2*>(go + 2 * Hp) = pack_bf16x4(vg[0], vg[1], vg[2], vg[3]);
                *reinterpret_cast<uint2*>(go + 3 * Hp) = pack_bf16x4(vo[0], vo[1], vo[2], vo[3]);
                *reinterpret_cast<uint2*>(Cs + (size_t)t * slabH + (size_t)b * Hp + j) = pack_bf16x4(vc[0], vc[1], vc[2], vc[3]);
            }
            PL_ST(4);   // cell update + store issue
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            PL_ST(5);   // hand-off store drain (the five stash stores stay in flight)
            publish<5>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);   // also orders this step's LDS reads before the next step's LDS writes
            PL_ST(6);   // barrier + arrival add
        }
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
// backward sweep (backward-DATA only)
// ---------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_sweep_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int NPART = KS >= 8 ? 4 : 2;       // K parts of a gate block (k-step ranges KS*i/NPART)
    constexpr int KPM = (KS + NPART - 1) / NPART;   // k-steps of the largest part = sc1 loads per thread per part
    constexpr int RS = KPM * 32 + 16;            // LDS row stride (odd chunk count -> conflict-free b128 reads)
    constexpr int BLK = 32 * RS;                 // one gate block (= one wave's operand) of a part
    constexpr int IMG = 4 * BLK;                 // one image buffer
    constexpr int PF = 4;                        // B-fragment read-ahead
    constexpr int LDR = 33;
    __shared__ __attribute__((aligned(16))) unsigned char img[2 * IMG];
    __shared__ float red[4 * 32 * LDR];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = Hp / 32;
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);

    // weights -> registers: wave = gate block; A-operand row = hidden unit 32p + (lane & 31); k inside the gate block
    uint4 wreg[KS];
    {
        const bf16_t* wrow = WT + (size_t)(32 * p + (lane & 31)) * G4 + wave * Hp + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wreg[ks] = *reinterpret_cast<const uint4*>(wrow + 16 * ks);
    }

    // epilogue ownership: thread -> batch row (tid >> 3), hidden units 32p + 4 (tid & 7) .. +3
    const int erow = tid >> 3, jq = tid & 7;
    PL_ST_DECL
    const int j = 32 * p + 4 * jq;
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + erow;
        const bool ok = erow < gs && b < Bp;
        const int bc = ok ? b : Bp - 1;
        float dc_next[4] = {0.f, 0.f, 0.f, 0.f};
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;   // flags [group][step][flag_stride]
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        for (int t = T - 1; t >= 0; --t) {
            // stash operands of this step (written by the forward launch): plain loads before the wait
            const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
            uint2 sg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[q] = *reinterpret_cast<const uint2*>(g_row + q * Hp);
            const uint2 sc = *reinterpret_cast<const uint2*>(Cs + (size_t)t * slabH + (size_t)bc * Hp + j);
            uint2 scp = make_uint2(0u, 0u);
            if (t > 0) scp = *reinterpret_cast<const uint2*>(Cs + (size_t)(t - 1) * slabH + (size_t)bc * Hp + j);
            uint2 sdh = make_uint2(0u, 0u);
            if (dhe) sdh = *reinterpret_cast<const uint2*>(dhe + (size_t)t * slabH + (size_t)bc * Hp + j);
            else if (dhl && t == T - 1) sdh = *reinterpret_cast<const uint2*>(dhl + (size_t)bc * Hp + j);

            float dh[4];
            unpack_bf16x4(sdh, dh);
            PL_ST(0);
            if (t + 1 < T) {
                if (!wait_arrivals(cnt + (size_t)(t + 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == T - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)(t + 1) * slabG, (unsigned)(slabG * 2));
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                // dA_{t+1} (32 batch rows x 4 gate blocks) streams through a double-buffered LDS image in NPART K parts:
                // the sc1 loads of part i+2 are issued before the MFMAs of part i, so two parts are always in flight.
                uint4 v[2][KPM];
                auto issue_part = [&](auto pi_c) {
                    constexpr int pi = decltype(pi_c)::value;
                    constexpr int k0 = KS * pi / NPART, kc = KS * (pi + 1) / NPART - k0;
#pragma unroll
                    for (int i = 0; i < kc; ++i) {
                        const int q = tid + 256 * i;
                        const int c = q % (2 * kc), row = (q / (2 * kc)) % 32, blk = q / (64 * kc);
                        int rb = gs * g + row;
                        rb = rb < Bp ? rb : Bp - 1;
                        v[pi % 2][i] = row < gs ? ld16_handoff(rg, (unsigned)((rb * G4 + blk * Hp + 16 * k0) * 2 + c * 16), plain_handoff)
                                                : make_uint4(0, 0, 0, 0);
                    }
                };
                auto run_part = [&](auto pi_c) {
                    constexpr int pi = decltype(pi_c)::value;
                    constexpr int k0 = KS * pi / NPART, kc = KS * (pi + 1) / NPART - k0;
                    unsigned char* buf = img + (pi % 2) * IMG;
#pragma unroll
                    for (int i = 0; i < kc; ++i) {
                        const int q = tid + 256 * i;
                        const int c = q % (2 * kc), row = (q / (2 * kc)) % 32, blk = q / (64 * kc);
                        *reinterpret_cast<uint4*>(buf + blk * BLK + row * RS + c * 16) = v[pi % 2][i];
                    }
                    __syncthreads();
                    if constexpr (pi + 2 < NPART) issue_part(std::integral_constant<int, pi + 2>{});
                    const unsigned char* bsrc = buf + wave * BLK + (lane & 31) * RS + (lane >> 5) * 16;
                    uint4 bq[PF];
#pragma unroll
                    for (int i = 0; i < PF; ++i)
                        if (i < kc) bq[i] = *reinterpret_cast<const uint4*>(bsrc + i * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < kc; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[k0 + ks]),
                                                                      __builtin_bit_cast(bf16x8, bq[ks % PF]), acc, 0, 0, 0);
                        if (ks + PF < kc) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                issue_part(std::integral_constant<int, 0>{});
                issue_part(std::integral_constant<int, 1>{});
                run_part(std::integral_constant<int, 0>{});
                run_part(std::integral_constant<int, 1>{});
                if constexpr (NPART == 4) {
                    run_part(std::integral_constant<int, 2>{});
                    run_part(std::integral_constant<int, 3>{});
                }
                PL_ST(2);   // both halves: sc1 loads + LDS image + MFMA
                // reduce the four gate-block partials: acc[r] = out[row (r&3) + 8 (r>>2) + 4 (lane>>5)][col lane & 31]
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDR + (lane & 31)] = acc[r];
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int w = 0; w < 4; ++w) dh[u] += red[(w * 32 + 4 * jq + u) * LDR + erow];
            }

            PL_ST(3);   // partial reduction
            float gi[4], gf[4], gg[4], go[4], c[4], cp[4];
            unpack_bf16x4(sg[0], gi);
            unpack_bf16x4(sg[1], gf);
            unpack_bf16x4(sg[2], gg);
            unpack_bf16x4(sg[3], go);
            unpack_bf16x4(sc, c);
            unpack_bf16x4(scp, cp);
            float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float tc = tanh_fast(c[u]);
                const float dc = dc_next[u] + dh[u] * go[u] * (1.f - tc * tc);
                dai[u] = dc * gg[u] * gi[u] * (1.f - gi[u]);
                daf[u] = dc * cp[u] * gf[u] * (1.f - gf[u]);
                dag[u] = dc * gi[u] * (1.f - gg[u] * gg[u]);
                dao[u] = dh[u] * tc * go[u] * (1.f - go[u]);
                dc_next[u] = dc * gf[u];
            }
            if (t == T - 1 && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok) {   // dA_t overwrites the gate stash in place: the hand-off of the next (earlier) step
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
                const unsigned o = (unsigned)((b * G4 + j) * 2);
                st8_handoff(ro, o, pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), plain_handoff);
                st8_handoff(ro, o + Hp * 2, pack_bf16x4(daf[0], daf[1], daf[2], daf[3]), plain_handoff);
                st8_handoff(ro, o + 2 * Hp * 2, pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), plain_handoff);
                st8_handoff(ro, o + 3 * Hp * 2, pack_bf16x4(dao[0], dao[1], dao[2], dao[3]), plain_handoff);
            }
            PL_ST(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PL_ST(5);
            publish<0>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);
        }
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
#define PL_SWEEP_KS_LIST(X) X(2) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(46) X(48)

__global__ void zero_counters_kernel(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(p + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_zero_counters(hipStream_t stream, int* p, int n) {
    hipLaunchKernelGGL(zero_counters_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n);
}

bool lstm_sweep_supported(int dt, int Hp) {
    if (dt != BF16) return false;
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_SWEEP_KS_LIST(PL_CASE)
#undef PL_CASE
    return false;
}

// Rows per batch group: 32 (the MFMA tile width).  Smaller groups on more CUs were measured (24 rows x 11 groups = 253
// workgroups at B = 256): the backward ingest got SLOWER (3.3 -> 3.8 us): what bounds it is the chip-wide rate of the
// write-through hand-off traffic (P x B x 4Hp x 2 bytes per step whatever the grouping), not the per-CU rate.
int lstm_sweep_group_rows(int Hp, int Bp, int n_cu) {
    if (n_cu < Hp / 32) return 0;
    return Bp < 32 ? (Bp + 7) / 8 * 8 : 32;
}

int lstm_sweep_grid(int Hp, int Bp, int n_cu, bool spread_small) {
    const int P = Hp / 32, gs = lstm_sweep_group_rows(Hp, Bp, n_cu);
    if (gs == 0) return 0;
    const int groups = (Bp + gs - 1) / gs;
    int res = n_cu / P;
    // fewer groups than XCDs: launch 8 slots anyway (blocks of the empty slots leave at once), so that a group's
    // workgroups (blockIdx % 8 == group) still land on one XCD and can use the verified same-XCD hand-off
    if (spread_small && groups < 8 && res >= 8) return 8 * P;
    if (res > groups) res = groups;
    // Resident groups: as few as sweep all groups in the fewest passes (a workgroup takes its groups in turn), and a multiple of the
    // XCD count where that costs no pass -- it keeps a group's workgroups on one XCD (speed only: the verified same-XCD hand-off).
    // Round 2 rounded DOWN to a multiple of 8 unconditionally: 12 groups of the 4 x 180 predictor then took two passes on 8 slots
    // where 12 fit at once (set B at B = 192: 14.2 ms per iteration against 6.7 ms at B = 256), 64 groups of cfg4_1gpu 8 passes
    // instead of 6.
    if (res >= 1) {
        const int cap = res, passes = (groups + cap - 1) / cap;
        res = (groups + passes - 1) / passes;
        const int r8 = (res + 7) / 8 * 8;
        if (r8 <= cap) res = r8;
    }   // a multiple of the XCD count keeps a group's workgroups on one XCD (speed only)
    return res < 1 ? 0 : res * P;
}

void launch_lstm_sweep(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a) {
    const int ksx = (!backward && a.x_in) ? a.in_p / 16 : 0;   // fused input projection (in_p = 32 or 64)
#define PL_CASE(K)                                                                                               \
    if (Hp == 16 * K) {                                                                                          \
        if (backward) hipLaunchKernelGGL(lstm_bwd_sweep_kernel<K>, dim3(grid), dim3(256), 0, stream, a);         \
        else if (ksx == 2) hipLaunchKernelGGL((lstm_fwd_sweep_kernel<K, 2>), dim3(grid), dim3(256), 0, stream, a); \
        else if (ksx == 4) hipLaunchKernelGGL((lstm_fwd_sweep_kernel<K, 4>), dim3(grid), dim3(256), 0, stream, a); \
        else hipLaunchKernelGGL((lstm_fwd_sweep_kernel<K, 0>), dim3(grid), dim3(256), 0, stream, a);             \
        return;                                                                                                  \
    }
    PL_SWEEP_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
