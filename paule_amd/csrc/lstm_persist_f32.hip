// Persistent LSTM layer sweeps, f32 (exact `v_mfma_f32_16x16x4_f32`): ONE launch runs all T time steps of one layer.
//
// Same idea as the bf16 sweeps (lstm_persist.hip / lstm_persist_rs.hip): W_hh never leaves registers, only the
// recurrent operand moves, in-launch exchange with arrival flags (sweep_common.h).  f32 MFMA is 16x slower than
// bf16, so the work is cut finer to spread it over the chip: groups of 16 batch rows, one workgroup per 16 hidden
// units (P = Hp / 16 workgroups per group: 46 at H = 720; B = 64 fills 184 CUs).
//   forward : wave w owns 4 hidden units x 4 gates = 16 gate rows, ordered [unit slot q (4)][gate (4)] so that the
//             C layout (row = 4 (lane >> 4) + reg, col = lane & 15) leaves a lane with the four gates of ONE unit of
//             one batch row.  A / B fragments are 16-byte (4 k) pieces feeding 4 consecutive MFMAs with the same
//             permuted k order on both operands.  184 weight VGPRs per lane at H = 720.
//   backward: reduce-scatter form: a workgroup multiplies the dA it produced itself (64 gate rows, from LDS) with its
//             64 rows of W_hh -> its f32 partial of dh for all Hp units, 46 tiles of 16 x 16, stored straight from
//             the accumulators (a lane's 4 registers = 16 contiguous bytes of a tile row); a workgroup then sums the
//             46 partial tiles of its own units.  Partials stay f32: the only rounding difference to the
//             launch-per-step kernels is the summation order.
#include "sweep_common.h"

namespace pl {

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ uint4 ld16_ho(__amdgpu_buffer_rsrc_t r, unsigned off, bool same_xcd) { return ld16_handoff(r, off, same_xcd); }

__device__ __forceinline__ void st16_handoff(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v, bool plain) {
    u32x4 d;
    d[0] = __builtin_bit_cast(unsigned, v.x);
    d[1] = __builtin_bit_cast(unsigned, v.y);
    d[2] = __builtin_bit_cast(unsigned, v.z);
    d[3] = __builtin_bit_cast(unsigned, v.w);
    if (plain) __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, kAuxSc1);
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__device__ __forceinline__ void st4_handoff(__amdgpu_buffer_rsrc_t r, unsigned off, float v, bool plain) {
    if (plain) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, kAuxSc1);
}

// Few-row forward product (1 <= NB <= 4 batch rows in use): lane (r = lane & 15, kq = lane >> 4) holds gate row r of its wave at
// k = 16 s + 4 kq .. + 3 (the MFMA A fragment), so the recurrent product of batch row b is a per-lane FMA chain over s and a
// sum over the four kq lanes.  vimg [wave][b][r] hands the sums over to the epilogue layout (lane bl = batch row, kq = unit
// slot, the four gates of that unit): returns them, zero for lanes whose batch row is not in use.
template <int NB, int KS, int KSX>
__device__ __forceinline__ f32x4 fwd_rows_valu(const float4 (&wreg)[KS], const float4* wx, const unsigned char* himg, int RS, bool have_h,
                                               const unsigned char* ximg, int XRS, float* vimg, int lane, int wave) {
    const int kq = lane >> 4, r = lane & 15;
    float pv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pv[b] = 0.f;
    if (have_h) {
        float part[NB][4];   // four independent chains per row (fixed order: deterministic)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[b][j] = 0.f;
        // the h fragments come from LDS through a ring of RA reads in flight: one read per FMA group would serialise the
        // ~100-cycle LDS latency 46 times (the wave is alone on its SIMD)
        constexpr int RA = 8;
        float4 ring[NB][RA];
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (i < KS) ring[b][i] = *reinterpret_cast<const float4*>(himg + b * RS + i * 64 + kq * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                part[b][s & 3] += dot4(wreg[s], ring[b][s % RA]);
                if (s + RA < KS) ring[b][s % RA] = *reinterpret_cast<const float4*>(himg + b * RS + (s + RA) * 64 + kq * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) pv[b] = (part[b][0] + part[b][1]) + (part[b][2] + part[b][3]);
    }
    if constexpr (KSX > 0) {
#pragma unroll
        for (int s = 0; s < KSX; ++s) {
#pragma unroll
            for (int b = 0; b < NB; ++b) pv[b] += dot4(wx[s], *reinterpret_cast<const float4*>(ximg + b * XRS + s * 64 + kq * 16));
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) pv[b] += __shfl_xor(pv[b], 16, 64);
#pragma unroll
    for (int b = 0; b < NB; ++b) pv[b] += __shfl_xor(pv[b], 32, 64);
    // every lane of gate row r now holds the row's sums; the epilogue lane (batch row r', unit slot kq) wants gate rows 4 kq .. + 3
    // of batch row r': four lane reads per row in use (no LDS image, no barriers)
    (void)vimg; (void)wave;
    f32x4 out = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float g4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = __shfl(pv[b], 4 * kq + i, 64);
        if (r == b) out = f32x4{g4[0], g4[1], g4[2], g4[3]};   // r doubles as the batch row of the epilogue layout
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// forward sweep (all-gather of h); KSX > 0: input projection fused (in_p = 16 * KSX = 32 / 64)
// ---------------------------------------------------------------------------------------------------
template <int KS, int KSX, int NV>   // KS = Hp / 16; NV > 0: that many batch rows in use, products as FMA chains (one group only)
__global__ __launch_bounds__(256, 1) void lstm_fwd_sweep_f32_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int ROWB = Hp * 4;                 // bytes of one h row
    constexpr int RS = ROWB + 16;                // LDS row stride: Hp / 4 + 1 chunks (odd): conflict-free b128 reads
    constexpr int CPR = Hp / 4;                  // 16-byte chunks per row
    constexpr int NLD = (16 * CPR + 255) / 256;  // loads per thread per step
    constexpr int PF = 4;                        // B-fragment read-ahead (16-k chunks)
    constexpr int HRS = 64 + 16;                 // outgoing h tile [16 rows][16 units] f32
    constexpr int XRS = KSX * 64 + 16;           // x_t image [16 rows][in_p] f32
    __shared__ __attribute__((aligned(16))) unsigned char himg[16 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char hst[6 * 16 * HRS];   // [h, gates i f g o, c][16 rows][16 units] f32
    __shared__ __attribute__((aligned(16))) unsigned char ximg[KSX ? 16 * XRS : 16];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int P = KS;                        // workgroups per group
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int t_begin = a.t0, t_end = a.t1 > 0 ? a.t1 : T;   // time chunk of this launch
    const int n_groups = (Bp + 15) / 16;         // Bp is a multiple of 16: groups are whole
    const float* __restrict__ W = static_cast<const float*>(a.W);
    const int kq = lane >> 4;
    constexpr int nv = NV;

    // weights -> registers: A row r (= lane & 15): unit slot r >> 2, gate r & 3
    float4 wreg[KS];
    {
        const int r = lane & 15;
        const float* wrow = W + (size_t)((r & 3) * Hp + 16 * p + 4 * wave + (r >> 2)) * Hp + 4 * kq;
#pragma unroll
        for (int s = 0; s < KS; ++s) wreg[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
    }
    // epilogue ownership (C layout): batch column lane & 15, unit 16p + 4 wave + (lane >> 4), gate = accumulator register
    const int bl = lane & 15;
    const int j = 16 * p + 4 * wave + kq;
    float4 wx[KSX ? KSX : 1];
    f32x4 bias_r = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KSX > 0) {
        constexpr int INP = 16 * KSX;
        const int r = lane & 15;
        const float* xrow = static_cast<const float*>(a.Wih) + (size_t)((r & 3) * Hp + 16 * p + 4 * wave + (r >> 2)) * INP + 4 * kq;
#pragma unroll
        for (int s = 0; s < KSX; ++s) wx[s] = *reinterpret_cast<const float4*>(xrow + 16 * s);
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) bias_r[gate] = a.bias[gate * Hp + j];
    }
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    float* __restrict__ G = static_cast<float*>(a.G);
    float* __restrict__ Hs = static_cast<float*>(a.h);
    float* __restrict__ Cs = static_cast<float*>(a.c);

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = 16 * g + bl;
        float c_state = t_begin > 0 ? a.carry[(size_t)b * Hp + j] : 0.f;
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        for (int t = t_begin; t < t_end; ++t) {
            float gx[4] = {0.f, 0.f, 0.f, 0.f};
            f32x4 acc;
            if constexpr (KSX > 0) {
                constexpr int INP = 16 * KSX, XC = INP / 4;
                if (tid < 16 * XC) {
                    const int row = tid / XC, c = tid % XC;
                    const uint4 xv = *reinterpret_cast<const uint4*>(static_cast<const float*>(a.x_in) +
                                                                     ((size_t)t * Bp + 16 * g + row) * INP + c * 4);
                    *reinterpret_cast<uint4*>(ximg + row * XRS + c * 16) = xv;
                }
                acc = bias_r;
                if (t == t_begin) __syncthreads();
            } else {
                const float* g_row = G + (size_t)t * slabG + (size_t)b * G4 + j;
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) gx[gate] = g_row[gate * Hp];
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            PL_ST(0);
            if (t > 0) {
                if (!wait_arrivals(cnt + (size_t)(t - 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == t_begin + 1 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t - 1) * slabH, (unsigned)(slabH * 4));
                const int n_chunks = (nv ? nv : 16) * CPR;   // the few-row path reads the rows in use only
                uint4 v[NLD];
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + 256 * i;
                    v[i] = (e < n_chunks) ? ld16_ho(rh, (unsigned)((16 * g + e / CPR) * ROWB + (e % CPR) * 16), plain_handoff)
                                          : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + 256 * i;
                    if (e < n_chunks) *reinterpret_cast<uint4*>(himg + (e / CPR) * RS + (e % CPR) * 16) = v[i];
                }
                __syncthreads();
                PL_ST(2);
                if constexpr (NV == 0) {
                const unsigned char* bsrc = himg + bl * RS + kq * 16;
                float4 bq[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < KS) bq[i] = *reinterpret_cast<const float4*>(bsrc + i * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc = mfma4(wreg[s], bq[s % PF], acc);
                    if (s + PF < KS) bq[s % PF] = *reinterpret_cast<const float4*>(bsrc + (s + PF) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
            if constexpr (NV > 0) {   // few rows in use: FMA chains instead of the MFMAs (recurrent and, if fused, input product)
                float* vimg = reinterpret_cast<float*>(hst);   // the staging image is free until the cell update below
                const f32x4 add = fwd_rows_valu<NV, KS, KSX>(wreg, wx, himg, RS, t > 0, ximg, XRS, vimg, lane, wave);
                acc[0] += add[0]; acc[1] += add[1]; acc[2] += add[2]; acc[3] += add[3];
            } else if constexpr (KSX > 0) {
#pragma unroll
                for (int s = 0; s < KSX; ++s)
                    acc = mfma4(wx[s], *reinterpret_cast<const float4*>(ximg + bl * XRS + s * 64 + kq * 16), acc);
            }
            PL_ST(3);

            // cell (libm forms: the f32 path carries the 1e-5 parity bar)
            const float vi = sigmoid_f(acc[0] + gx[0]), vf = sigmoid_f(acc[1] + gx[1]);
            const float vg = tanhf(acc[2] + gx[2]), vo = sigmoid_f(acc[3] + gx[3]);
            c_state = vf * c_state + vi * vg;
            const float vh = vo * tanhf(c_state);
            if (t == t_begin && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // hand-off first: the workgroup's h tile (16 rows x 16 units) leaves as whole 64-byte row pieces
            {   // all six outputs leave through LDS as whole 64-byte row pieces (a lane's own values are single floats)
                unsigned char* o = hst + bl * HRS + (4 * wave + kq) * 4;
                *reinterpret_cast<float*>(o) = vh;
                *reinterpret_cast<float*>(o + 1 * 16 * HRS) = vi;
                *reinterpret_cast<float*>(o + 2 * 16 * HRS) = vf;
                *reinterpret_cast<float*>(o + 3 * 16 * HRS) = vg;
                *reinterpret_cast<float*>(o + 4 * 16 * HRS) = vo;
                *reinterpret_cast<float*>(o + 5 * 16 * HRS) = c_state;
            }
            __syncthreads();
            if (tid < 64) {
                const int row = tid >> 2, qt = tid & 3;
                const float4 hv = *reinterpret_cast<const float4*>(hst + row * HRS + qt * 16);
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 4));
                st16_handoff(ro, (unsigned)(((16 * g + row) * Hp + 16 * p + 4 * qt) * 4), hv, plain_handoff);
            }
            asm volatile("" ::: "memory");   // keep the stash stores behind it
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i;   // piece: array e / 64 (gates i f g o, c), row (e % 64) / 4, quarter e % 4
                if (e < 320) {
                    const int arr = e >> 6, row = (e & 63) >> 2, qt = e & 3, rb = 16 * g + row;
                    const float4 sv = *reinterpret_cast<const float4*>(hst + (arr + 1) * 16 * HRS + row * HRS + qt * 16);
                    float* dst = arr < 4 ? G + (size_t)t * slabG + (size_t)rb * G4 + arr * Hp + 16 * p + 4 * qt
                                         : Cs + (size_t)t * slabH + (size_t)rb * Hp + 16 * p + 4 * qt;
                    *reinterpret_cast<float4*>(dst) = sv;
                }
            }
            PL_ST(4);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the hand-off store is older than the (at most 2) stash stores
            PL_ST(5);
            publish<2>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);
        }
        if (t_end < T) a.carry[(size_t)b * Hp + j] = c_state;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
// backward sweep, reduce-scatter of f32 partial dh tiles (backward-DATA only)
// ---------------------------------------------------------------------------------------------------
template <int KS, int NV>
__global__ __launch_bounds__(256, 1) void lstm_bwd_sweep_f32_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = KS;                        // workgroups per group = N tiles of 16 hidden units
    constexpr int NT = (P + 3) / 4;              // N tiles per wave (wave w: tiles w, w + 4, ...)
    constexpr int DRS = 64 * 4 + 16;             // dA^T image [16 batch rows][64 local gate rows] f32
    __shared__ __attribute__((aligned(16))) unsigned char da_img[16 * DRS];
    __shared__ __attribute__((aligned(16))) float red[4][16][16];   // per-wave sums of the partial tiles (wide ingest)
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int t_begin = a.t0, t_end = a.t1 > 0 ? a.t1 : T;   // time chunk of this launch
    const int n_groups = (Bp + 15) / 16;
    const float* __restrict__ WT = static_cast<const float*>(a.W);   // Whh^T packed [Hp][4*Hp]
    const int kq = lane >> 4;
    constexpr int nv = NV;

    // weights -> registers: tile nt = wave + 4 i: A row = hidden column n = 16 nt + (lane & 15); chunk c = gate c:
    // k = 4 kq + e  <->  gate row c * Hp + 16 p + 4 kq + e
    float4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 16 * (nt < P ? nt : 0) + (lane & 15);
#pragma unroll
        for (int c = 0; c < 4; ++c) wreg[i][c] = *reinterpret_cast<const float4*>(WT + (size_t)n * G4 + c * Hp + 16 * p + 4 * kq);
    }

    // cell ownership: thread -> batch row tid >> 4, unit 16p + (tid & 15)
    const int erow = tid >> 4, eu = tid & 15;
    const int j = 16 * p + eu;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    float* __restrict__ G = static_cast<float*>(a.G);
    const float* __restrict__ Cs = static_cast<const float*>(a.c);
    const float* __restrict__ dhe = static_cast<const float*>(a.dh_ext);
    const float* __restrict__ dhl = static_cast<const float*>(a.dh_last);
    // exchange [2 slots][groups][P destinations][P sources][16 rows][16 columns] f32 (1-KB tiles)
    float* __restrict__ X = static_cast<float*>(a.xchg);
    constexpr size_t TILE = 16 * 16;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = 16 * g + erow;
        float dc_next = t_end < T ? a.carry[(size_t)b * Hp + j] : 0.f;
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        float c_carry = 0.f;   // c_{t-1} of the step before = c_t of this one: loaded once
        for (int t = t_end - 1; t >= t_begin; --t) {
            const float* g_row = G + (size_t)t * slabG + (size_t)b * G4 + j;
            const float gi = g_row[0], gf = g_row[Hp], gg = g_row[2 * Hp], go = g_row[3 * Hp];
            const float c = t == t_end - 1 ? Cs[(size_t)t * slabH + (size_t)b * Hp + j] : c_carry;
            const float cp = t > 0 ? Cs[(size_t)(t - 1) * slabH + (size_t)b * Hp + j] : 0.f;
            c_carry = cp;
            float dh = 0.f;
            if (dhe) dh = dhe[(size_t)t * slabH + (size_t)b * Hp + j];
            else if (dhl && t == T - 1) dh = dhl[(size_t)b * Hp + j];
            PL_ST(0);
            if (t + 1 < T) {
                if (!wait_arrivals(cnt + (size_t)(t + 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == t_end - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                const float* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 4));
                if (a.stash_via_lds & 2) {
                    // wide ingest: wave w sums the tiles of sources w * TPG .. with 16-byte loads (a wave instruction = one whole 1-KB
                    // tile instead of a quarter of it: 46 instead of 184 load instructions per workgroup and step), the four waves'
                    // sums meet in LDS.  Fixed order: deterministic.
                    constexpr int TPG = (P + 3) / 4;
                    const int r16 = lane >> 2, quad = lane & 3;
                    uint4 pw[TPG];
#pragma unroll
                    for (int i = 0; i < TPG; ++i) {
                        const int src = wave * TPG + i;
                        pw[i] = src < P ? ld16_ho(rx, (unsigned)(src * TILE * 4 + (r16 * 16 + quad * 4) * 4), plain_handoff) : make_uint4(0, 0, 0, 0);
                    }
                    float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < TPG; ++i) {
                        part.x += __builtin_bit_cast(float, pw[i].x);
                        part.y += __builtin_bit_cast(float, pw[i].y);
                        part.z += __builtin_bit_cast(float, pw[i].z);
                        part.w += __builtin_bit_cast(float, pw[i].w);
                    }
                    *reinterpret_cast<float4*>(&red[wave][r16][quad * 4]) = part;
                    __syncthreads();
                    dh += (red[0][erow][eu] + red[1][erow][eu]) + (red[2][erow][eu] + red[3][erow][eu]);
                } else {
                const unsigned o0 = (unsigned)((erow * 16 + eu) * 4);
                unsigned pv[P];
#pragma unroll
                for (int s = 0; s < P; ++s)
                    pv[s] = plain_handoff ? __builtin_amdgcn_raw_buffer_load_b32(rx, o0 + (unsigned)(s * TILE * 4), 0, kAuxNt)
                                          : __builtin_amdgcn_raw_buffer_load_b32(rx, o0 + (unsigned)(s * TILE * 4), 0, kAuxSc1);
#pragma unroll
                for (int s = 0; s < P; ++s) dh += __builtin_bit_cast(float, pv[s]);
                }
            }
            PL_ST(2);

            const float tc = tanhf(c);
            const float dc = dc_next + dh * go * (1.f - tc * tc);
            const float dai = dc * gg * gi * (1.f - gi);
            const float daf = dc * cp * gf * (1.f - gf);
            const float dag = dc * gi * (1.f - gg * gg);
            const float dao = dh * tc * go * (1.f - go);
            dc_next = dc * gf;
            {   // dA_t overwrites the gate stash in place (read later by the dX / dH GEMM launches)
                float* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                go_[0] = dai;
                go_[Hp] = daf;
                go_[2 * Hp] = dag;
                go_[3 * Hp] = dao;
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == t_end - 1 && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 16 + unit]
                float* drow = reinterpret_cast<float*>(da_img + erow * DRS) + eu;
                drow[0] = dai;
                drow[16] = daf;
                drow[32] = dag;
                drow[48] = dao;
            }
            __syncthreads();
            PL_ST(3);
            float* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 4));
            if constexpr (NV > 0) {
                // few rows in use: lane (n = lane & 15, kq) holds column 16 nt + n of W^T at k = 16 c + 4 kq .. + 3 (the MFMA A
                // fragment): partial[n][b] is a per-lane FMA chain over the four gates and a sum over the four kq lanes; the
                // rows not in use keep the zeros the exchange was allocated with
#pragma unroll
                for (int b = 0; b < nv; ++b) {
                    float4 d4[4];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) d4[c4] = *reinterpret_cast<const float4*>(da_img + b * DRS + c4 * 64 + kq * 16);
                    float q[NT];   // all tiles first, then the two cross-lane steps back to back (their latencies overlap)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        q[i] = (dot4(wreg[i][0], d4[0]) + dot4(wreg[i][1], d4[1])) + (dot4(wreg[i][2], d4[2]) + dot4(wreg[i][3], d4[3]));
#pragma unroll
                    for (int i = 0; i < NT; ++i) q[i] += __shfl_xor(q[i], 16, 64);
#pragma unroll
                    for (int i = 0; i < NT; ++i) q[i] += __shfl_xor(q[i], 32, 64);
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int nt = wave + 4 * i;
                        if (nt < P && kq == 0)
                            st4_handoff(ro, (unsigned)(((size_t)nt * P * TILE + b * 16 + (lane & 15)) * 4), q[i], plain_handoff);
                    }
                }
            } else {
            float4 bfr[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) bfr[c4] = *reinterpret_cast<const float4*>(da_img + (lane & 15) * DRS + c4 * 64 + kq * 16);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int nt = wave + 4 * i;
                if (nt >= P) break;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) acc = mfma4(wreg[i][c4], bfr[c4], acc);
                // acc[r] = partial[n = 16 nt + 4 kq + r][batch lane & 15]: a lane's 4 registers are 16 contiguous bytes of tile row b
                st16_handoff(ro, (unsigned)(((size_t)nt * P * TILE + (lane & 15) * 16 + 4 * kq) * 4),
                             make_float4(acc[0], acc[1], acc[2], acc[3]), plain_handoff);
            }
            }
            PL_ST(4);
            publish<0>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);
        }
        if (t_begin > 0) a.carry[(size_t)b * Hp + j] = dc_next;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
// backward sweep, STREAMED form of the reduce-scatter (round 5): the per-tile hand-off of lstm_persist_rs.hip on the f32 tiles.
// In the kernel above a step is a chain of whole-workgroup phases -- 46 tiles, drain (vmcnt(0)), barrier, ONE flag, (consumers) poll all
// 46 flags, barrier, 46 tile loads, sums -- and the last tile's way to its consumer starts only when the producer's slowest wave has
// drained.  Here every 16 x 16 tile travels on its own:
//   * a tile is produced by ONE wave (16 MFMAs, one 16-byte store per lane), so that wave alone waits for its store (counted vmcnt, one
//     tile behind: the wait sits under the next tile's MFMAs) and raises the tile's OWN flag,
//     tflags[slot][group][destination][source] = step + 1 (a 64-int row per destination, zeroed with the other flags);
//   * source p produces its tiles in the rotated order destination p + 1, p + 2, ... (mod P), four waves at a time, so that every
//     destination receives a share of its tiles per round instead of all of them in one;
//   * wave w of a destination sums the tiles of sources w * TPG ..: it polls THEIR flags with one wave instruction, loads (16 bytes per
//     lane: one instruction = one 1-KB tile) whatever has newly arrived, and polls again.
// Same tiles, same 16 MFMAs per tile, same fixed order of every sum (a wave's tiles in source order, then the four waves' sums):
// bit-identical to the kernel above (tests/test_hip_parity.py::test_f32_streamed_backward_is_bit_identical).  Two workgroup barriers per
// step instead of four; hand-off rules as before (write-through sc1 both sides, or plain / nt through the shared L2 for a group that
// verified it sits on one XCD); every spin bounded.  Whole sequences on the MFMA path only (time chunks and the few-row kernels keep
// the form above).
__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n) for a value the unrolled caller knows at compile time
    switch (n) {
#define PL_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        PL_VM(0) PL_VM(1) PL_VM(2) PL_VM(3) PL_VM(4) PL_VM(5) PL_VM(6) PL_VM(7) PL_VM(8) PL_VM(9) PL_VM(10) PL_VM(11) PL_VM(12) PL_VM(13) PL_VM(14) PL_VM(15) PL_VM(16)
#undef PL_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#ifndef PL_F32_STREAM_D
#define PL_F32_STREAM_D 6
#endif
template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_stream_f32_kernel(LstmSweepArgs a) {
    constexpr int D = PL_F32_STREAM_D < (KS + 3) / 4 ? PL_F32_STREAM_D : (KS + 3) / 4;   // tiles in flight behind a flag (<= 8: the exact counts fit wait_vm)
    constexpr int Hp = 16 * KS;
    constexpr int P = KS;                        // workgroups per group = N tiles of 16 hidden units
    constexpr int NT = (P + 3) / 4;              // tiles per wave: positions k = wave + 4 i of the rotated order
    constexpr int TPG = (P + 3) / 4;             // sources per wave of the ingest: w * TPG ..
    constexpr int DRS = 64 * 4 + 16;             // dA^T image [16 batch rows][64 local gate rows] f32
    static_assert(P <= 64, "one flag word per source in a 64-int row");
    __shared__ __attribute__((aligned(16))) unsigned char da_img[16 * DRS];
    __shared__ __attribute__((aligned(16))) float red[4][16][16];   // per-wave sums of the partial tiles
    __shared__ int lds_flag, lds_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int n_groups = (Bp + 15) / 16;
    const float* __restrict__ WT = static_cast<const float*>(a.W);   // Whh^T packed [Hp][4*Hp]
    const int kq = lane >> 4;

    // weights -> registers: the wave's tile at position k = wave + 4 i goes to destination nt = (p + 1 + k) mod P: A row = hidden column
    // n = 16 nt + (lane & 15); chunk c = gate c: k = 4 kq + e  <->  gate row c * Hp + 16 p + 4 kq + e
    float4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int k = wave + 4 * i;
        const int nt = k < P ? (p + 1 + k) % P : 0;
        const int n = 16 * nt + (lane & 15);
#pragma unroll
        for (int c = 0; c < 4; ++c) wreg[i][c] = *reinterpret_cast<const float4*>(WT + (size_t)n * G4 + c * Hp + 16 * p + 4 * kq);
    }

    // cell ownership: thread -> batch row tid >> 4, unit 16p + (tid & 15)
    const int erow = tid >> 4, eu = tid & 15;
    const int j = 16 * p + eu;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    float* __restrict__ G = static_cast<float*>(a.G);
    const float* __restrict__ Cs = static_cast<const float*>(a.c);
    const float* __restrict__ dhe = static_cast<const float*>(a.dh_ext);
    const float* __restrict__ dhl = static_cast<const float*>(a.dh_last);
    // exchange [2 slots][groups][P destinations][P sources][16 rows][16 columns] f32 (1-KB tiles)
    float* __restrict__ X = static_cast<float*>(a.xchg);
    constexpr size_t TILE = 16 * 16;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    int* const tf = a.tflags;   // [2 slots][groups][P destinations][64]: the row of destination d holds one word per source
    // the sources this wave sums: w * TPG .. (the last wave's share may be short)
    const int n_src = P - wave * TPG < TPG ? P - wave * TPG : TPG;
    const unsigned full = n_src >= 32 ? 0xffffffffu : ((1u << n_src) - 1u);
    if (tid == 0) lds_abort = 0;
    __syncthreads();

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = 16 * g + erow;
        float dc_next = 0.f;
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        float c_carry = 0.f;   // c_{t-1} of the step before = c_t of this one: loaded once
        for (int t = T - 1; t >= 0; --t) {
            const float* g_row = G + (size_t)t * slabG + (size_t)b * G4 + j;
            const float gi = g_row[0], gf = g_row[Hp], gg = g_row[2 * Hp], go = g_row[3 * Hp];
            const float c = t == T - 1 ? Cs[(size_t)t * slabH + (size_t)b * Hp + j] : c_carry;
            const float cp = t > 0 ? Cs[(size_t)(t - 1) * slabH + (size_t)b * Hp + j] : 0.f;
            c_carry = cp;
            float dh = 0.f;
            if (dhe) dh = dhe[(size_t)t * slabH + (size_t)b * Hp + j];
            else if (dhl && t == T - 1) dh = dhl[(size_t)b * Hp + j];
            PL_ST(0);
            if (t + 1 < T) {
                // the tiles of step t + 1 for this workgroup's units, loaded as their flags come up (vmcnt retires in order, so a poll's answer
                // is seen after the tile loads issued before it have landed -- loads the step needs anyway)
                const int token = t + 2;
                const int* frow = tf + ((size_t)((t + 1) & 1) * n_groups + g) * P * 64 + (size_t)p * 64 + wave * TPG;
                const __amdgpu_buffer_rsrc_t rf = make_rsrc(frow, (unsigned)(TPG * 4));
                const float* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 4));
                const int r16 = lane >> 2, quad = lane & 3;
                const unsigned o0 = (unsigned)((wave * TPG * TILE + r16 * 16 + quad * 4) * 4);
                uint4 pw[TPG];
#pragma unroll
                for (int i = 0; i < TPG; ++i) pw[i] = make_uint4(0u, 0u, 0u, 0u);
                unsigned issued = 0;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (unsigned spin = 1;; ++spin) {
                    int v = 0;
                    if (lane < TPG)
                        v = plain_handoff ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxNt)
                                          : (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
                    const unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(v == token) & full;
                    const unsigned newly = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask & ~issued));
#pragma unroll
                    for (int i = 0; i < TPG; ++i)
                        if ((newly >> i) & 1u) pw[i] = ld16_ho(rx, o0 + (unsigned)(i * TILE * 4), plain_handoff);
                    issued |= newly;
                    if (issued == full) break;
                    if ((spin & a.poll_mask) == 0 &&
                        (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0 ||
                         __builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks)) {
                        if (lane == 0) {
                            __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            lds_abort = 1;
                        }
                        break;   // (what was not issued stays zero; the workgroup leaves behind the barrier below)
                    }
                }
                PL_ST(1);   // polls + tile loads issued
                float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < TPG; ++i) {
                    part.x += __builtin_bit_cast(float, pw[i].x);
                    part.y += __builtin_bit_cast(float, pw[i].y);
                    part.z += __builtin_bit_cast(float, pw[i].z);
                    part.w += __builtin_bit_cast(float, pw[i].w);
                }
                *reinterpret_cast<float4*>(&red[wave][r16][quad * 4]) = part;
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(lds_abort) != 0) return;
                dh += (red[0][erow][eu] + red[1][erow][eu]) + (red[2][erow][eu] + red[3][erow][eu]);
                if (t == T - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
            }
            PL_ST(2);   // tiles landed + sums

            const float tc = tanhf(c);
            const float dc = dc_next + dh * go * (1.f - tc * tc);
            const float dai = dc * gg * gi * (1.f - gi);
            const float daf = dc * cp * gf * (1.f - gf);
            const float dag = dc * gi * (1.f - gg * gg);
            const float dao = dh * tc * go * (1.f - go);
            dc_next = dc * gf;
            {   // dA_t overwrites the gate stash in place (read later by the dX / dH GEMM launches)
                float* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                go_[0] = dai;
                go_[Hp] = daf;
                go_[2 * Hp] = dag;
                go_[3 * Hp] = dao;
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == T - 1 && tid == 0) {   // this workgroup's XCD, in place before ANY of its flags (they are raised behind the barrier below)
                __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 16 + unit]
                float* drow = reinterpret_cast<float*>(da_img + erow * DRS) + eu;
                drow[0] = dai;
                drow[16] = daf;
                drow[32] = dag;
                drow[48] = dao;
            }
            __syncthreads();
            PL_ST(3);   // cell + stash stores + dA image + barrier
            float* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 4));
            int* const fcol = tf + ((size_t)(t & 1) * n_groups + g) * P * 64 + p;   // + 64 * destination
            const __amdgpu_buffer_rsrc_t rfl = make_rsrc(fcol, (unsigned)(((P - 1) * 64 + 1) * 4));
            auto raise = [&](int nt) {   // the wave's own store of that tile is acknowledged: its flag
                if (lane == 0) {
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 64 * 4), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rfl, (unsigned)(nt * 64 * 4), 0, kAuxSc1);
                }
            };
            float4 bfr[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) bfr[c4] = *reinterpret_cast<const float4*>(da_img + (lane & 15) * DRS + c4 * 64 + kq * 16);
            int n_done = 0;   // tiles this wave has stored in this step
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int k = wave + 4 * i;
                // (the test is a compile-time fact for all but a wave's last tile and sits IN FRONT of the tile's MFMAs: no MFMA result is read
                // across it.  An `if`, not a `break`: with a break hipcc gave up unrolling at KS = 46 and moved the weights to scratch)
                if (4 * i + 3 < P || k < P) {
                    const int nt = (p + 1 + k) % P;
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) acc = mfma4(wreg[i][c4], bfr[c4], acc);
                    // acc[r] = partial[n = 16 nt + 4 kq + r][batch lane & 15]: a lane's 4 registers are 16 contiguous bytes of tile row b
                    st16_handoff(ro, (unsigned)(((size_t)nt * P * TILE + (lane & 15) * 16 + 4 * kq) * 4),
                                 make_float4(acc[0], acc[1], acc[2], acc[3]), plain_handoff);
                    // The flag of the tile D positions back: its store is acknowledged once at most N younger operations are outstanding -- the D tiles
                    // behind it and the flags raised meanwhile, min(D, i - D) of them (exact: a looser count would let the flag overtake its tile).
                    // D tiles in flight because a write-through acknowledge takes ~1.5 us and a tile's MFMAs 0.2: waiting one tile back (D = 1,
                    // the bf16 kernel's distance, whose stores are acknowledged by the L2) stalls the wave on every tile: cfg2 3.79 -> 4.16 ms.
                    if (i >= D) {
                        wait_vm(D + (i - D < D ? i - D : D));
                        raise((p + 1 + wave + 4 * (i - D)) % P);
                    }
                    n_done = i + 1;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last D tiles: flags behind the wave's final drain (the last tile's flag cannot be earlier)
#pragma unroll
            for (int r = D; r >= 1; --r)
                if (n_done - r >= 0) raise((p + 1 + wave + 4 * (n_done - r)) % P);
            PL_ST(4);   // tiles + flags
        }
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
#ifndef PL_F32_FEW_ROWS
#define PL_F32_FEW_ROWS 2
#endif
#define PL_SWEEP_F32_KS_LIST(X) X(2) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(46) X(48)

bool lstm_sweep_f32_supported(int Hp) {
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_SWEEP_F32_KS_LIST(PL_CASE)
#undef PL_CASE
    return false;
}

// Workgroups to launch, 0 = use the launch-per-step kernels.  A workgroup sweeps its groups one after the other
// (~7 us per step each), the launch-per-step kernels take ~24-28 us per step whatever the batch (they stream the f32
// weights every step): measured break-even is 4 rounds (B = 256 at H = 720: 36.6 vs 34.4 ms per iteration).
int lstm_sweep_f32_grid(int Hp, int Bp, int n_cu) {
    const int P = Hp / 16, groups = (Bp + 15) / 16;
    int res = n_cu / P;
    if (res < 1) return 0;
    if (res > groups) res = groups;
    if (res >= 8) res = res / 8 * 8;
    if ((groups + res - 1) / res > 3) return 0;
    return res * P;
}

size_t lstm_f32_exchange_bytes(int Hp, int Bp) {
    const size_t groups = (Bp + 15) / 16, P = Hp / 16;
    return 2 * groups * P * P * 16 * 16 * 4;
}

// rows in use for the few-row FMA kernels (0 = MFMA kernels): one group with one or two rows in use (B = 1, the reference's
// operating point: 6.85 -> 4.78 ms per iteration; B = 2: 6.83 -> 6.00; a third row would cost more than the MFMAs)
static int few_rows(const LstmSweepArgs& a) { return (a.Bp == 16 && a.n_valid >= 1 && a.n_valid <= PL_F32_FEW_ROWS) ? a.n_valid : 0; }

template <int K, int NV>
static void launch_f32_nv(hipStream_t stream, bool backward, int ksx, int grid, const LstmSweepArgs& a) {
    if (backward && NV == 0 && a.tflags && a.t0 == 0 && (a.t1 == 0 || a.t1 == a.T) && (a.stash_via_lds & 2))   // whole sequence on the MFMAs: streamed hand-off
        hipLaunchKernelGGL((lstm_bwd_stream_f32_kernel<K>), dim3(grid), dim3(256), 0, stream, a);
    else if (backward) hipLaunchKernelGGL((lstm_bwd_sweep_f32_kernel<K, NV>), dim3(grid), dim3(256), 0, stream, a);
    else if (ksx == 2) hipLaunchKernelGGL((lstm_fwd_sweep_f32_kernel<K, 2, NV>), dim3(grid), dim3(256), 0, stream, a);
    else if (ksx == 4) hipLaunchKernelGGL((lstm_fwd_sweep_f32_kernel<K, 4, NV>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((lstm_fwd_sweep_f32_kernel<K, 0, NV>), dim3(grid), dim3(256), 0, stream, a);
}

void launch_lstm_sweep_f32(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a) {
    const int ksx = (!backward && a.x_in) ? a.in_p / 16 : 0;
    const int nv = few_rows(a);
#define PL_CASE(K)                                                                   \
    if (Hp == 16 * K) {                                                              \
        if (nv == 1) launch_f32_nv<K, 1>(stream, backward, ksx, grid, a);            \
        else if (nv == 2) launch_f32_nv<K, 2>(stream, backward, ksx, grid, a);       \
        else launch_f32_nv<K, 0>(stream, backward, ksx, grid, a);                    \
        return;                                                                      \
    }
    PL_SWEEP_F32_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
