// Persistent LSTM sweeps for SMALL batches (bf16): batch groups of 16 rows on `v_mfma_f32_16x16x32_bf16`.
//
// The 32-row kernels (lstm_persist.hip, lstm_persist_rs.hip) spend the same time per step on one utterance as on
// 32: the 32 x 32 x 16 MFMA tile, the cell arithmetic of 4 cells per lane and the 47-KB operand tile are all sized
// for 32 batch rows.  The reference plans ONE utterance (paule/paule.py:585-588), continued learning uses mini-batches
// of 8 (:404), cfg5 has 16 utterances per GPU: for batches of up to 128 rows (8 groups x 23 workgroups = 184 CUs)
// the groups are cut to 16 rows instead, which halves the MFMA passes, the cells per lane (2) and the bytes of every
// hand-off.  Same decomposition otherwise: P = Hp / 32 workgroups per group, workgroup p owns hidden units
// [32p, 32p + 32) for all four gates, W_hh lives in registers (184-192 VGPRs per lane), in-launch exchange with
// arrival flags (sweep_common.h), forward = all-gather of h, backward = reduce-scatter of bf16 partial dh tiles.
//
//   forward : wave w owns units 8w .. 8w + 7 = two A tiles of 16 gate rows ordered [unit (4)][gate (4)]; the C layout
//             (row = 4 (lane >> 4) + reg, col = lane & 15) leaves lane (b, u) with the four gates of unit 8w + 4j + u
//             of batch row b in the four registers of tile j.  One B fragment (16 B of h per lane) feeds both tiles.
//   backward: the workgroup's 128 gate rows (K) x 46 N tiles of 16 hidden units, 4 MFMAs each; wave w takes tiles
//             w, w + 4, ...; exchange tiles are [16 rows][32 columns] bf16 (1 KB).
// All six outputs of a forward step (h hand-off + 4 gates + c stash) leave through LDS as whole 64-byte row pieces.
#include "sweep_common.h"

#ifndef PL16_OCC
#define PL16_OCC 1   // workgroups per CU the kernels are compiled for (2: <= 256 registers per lane)
#endif

namespace pl {

namespace {

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (bf16_t)v); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

}  // namespace

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <int KS, int KSX>   // KS = Hp / 16; KSX = in_p / 32 (0: G holds the precomputed input projection)
__global__ __launch_bounds__(256, PL16_OCC) void lstm_fwd16_sweep_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int KS32 = KS / 2;                 // MFMA k-steps of 32
    constexpr int ROWB = Hp * 2;
    constexpr int RS = ROWB + 16;                // h image row stride: odd number of 16-byte chunks
    constexpr int CPR = Hp / 8;                  // 16-byte chunks per h row
    constexpr int NLD = (16 * CPR + 255) / 256;
    constexpr int PF = 4;                        // B-fragment read-ahead (k-steps of 32)
    constexpr int XRS = KSX * 64 + 16;
    constexpr int ORS = 64 + 16;                 // staged outputs: [6 arrays][16 rows][32 units] bf16
    static_assert(KS % 2 == 0, "Hp is a multiple of 32");
    __shared__ __attribute__((aligned(16))) unsigned char himg[16 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char ost[6 * 16 * ORS];   // 0: h, 1..4: gates i f g o, 5: c
    __shared__ __attribute__((aligned(16))) unsigned char ximg[KSX ? 16 * XRS : 16];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = Hp / 32;
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int t_begin = a.t0, t_end = a.t1 > 0 ? a.t1 : T;   // time chunk of this launch
    const int gs = a.group_rows;                 // <= 16
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(a.W);
    const int lr = lane & 15, kq = lane >> 4;

    // weights -> registers: A row lr of tile j = unit 8w + 4j + (lr >> 2), gate lr & 3; k = 32 ks + 8 kq .. +7
    uint4 wreg[2][KS32];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const bf16_t* wrow = W + (size_t)((lr & 3) * Hp + 32 * p + 8 * wave + 4 * jt + (lr >> 2)) * Hp + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) wreg[jt][ks] = *reinterpret_cast<const uint4*>(wrow + 32 * ks);
    }
    uint4 wx[2][KSX ? KSX : 1];
    float bias_r[2][4];
    if constexpr (KSX > 0) {
        constexpr int INP = 32 * KSX;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const bf16_t* xrow = static_cast<const bf16_t*>(a.Wih) + (size_t)((lr & 3) * Hp + 32 * p + 8 * wave + 4 * jt + (lr >> 2)) * INP + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) wx[jt][ks] = *reinterpret_cast<const uint4*>(xrow + 32 * ks);
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) bias_r[jt][gate] = a.bias[gate * Hp + 32 * p + 8 * wave + 4 * jt + kq];
        }
    }
    // cell ownership (C layout): batch row lr, units 8w + kq (tile 0) and 8w + 4 + kq (tile 1); accumulator register = gate
    const int ul[2] = {8 * wave + kq, 8 * wave + 4 + kq};   // unit index inside the workgroup's slice
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(a.h);
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(a.c);

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + lr;
        const bool ok = lr < gs && b < Bp;
        const int bc = ok ? b : Bp - 1;
        float c_state[2] = {0.f, 0.f};
        if (t_begin > 0) {
            c_state[0] = a.carry[(size_t)bc * Hp + 32 * p + ul[0]];
            c_state[1] = a.carry[(size_t)bc * Hp + 32 * p + ul[1]];
        }
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;

        for (int t = t_begin; t < t_end; ++t) {
            float gx[2][4];
            f32x4 acc[2];
            if constexpr (KSX > 0) {
                constexpr int INP = 32 * KSX, XC = INP / 8;
                if (tid < 16 * XC) {
                    const int row = tid / XC, c = tid % XC;
                    int rb = gs * g + row;
                    rb = rb < Bp ? rb : Bp - 1;
                    const uint4 xv = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.x_in) + ((size_t)t * Bp + rb) * INP + c * 8);
                    *reinterpret_cast<uint4*>(ximg + row * XRS + c * 16) = xv;
                }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    acc[jt] = f32x4{bias_r[jt][0], bias_r[jt][1], bias_r[jt][2], bias_r[jt][3]};
#pragma unroll
                    for (int gate = 0; gate < 4; ++gate) gx[jt][gate] = 0.f;
                }
                if (t == t_begin) __syncthreads();
            } else {
                const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + 32 * p;
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
                    for (int gate = 0; gate < 4; ++gate) gx[jt][gate] = (float)g_row[gate * Hp + ul[jt]];
                    acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            PL_ST(0);
            if (t > 0) {
                if (!wait_arrivals(cnt + (size_t)(t - 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == t_begin + 1 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t - 1) * slabH, (unsigned)(slabH * 2));
                uint4 v[NLD];
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int q = tid + 256 * i, row = q / CPR, c = q % CPR;
                    int rb = gs * g + row;
                    rb = rb < Bp ? rb : Bp - 1;
                    v[i] = (q < 16 * CPR && row < gs) ? ld16_handoff(rh, (unsigned)(rb * ROWB + c * 16), plain_handoff) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int q = tid + 256 * i;
                    if (q < 16 * CPR) *reinterpret_cast<uint4*>(himg + (q / CPR) * RS + (q % CPR) * 16) = v[i];
                }
                __syncthreads();
                PL_ST(2);
                const unsigned char* bsrc = himg + lr * RS + kq * 16;
                uint4 bq[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < KS32) bq[i] = *reinterpret_cast<const uint4*>(bsrc + i * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[ks % PF]);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[0][ks]), bf, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[1][ks]), bf, acc[1], 0, 0, 0);
                    if (ks + PF < KS32) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (KSX > 0) {
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) {
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ximg + lr * XRS + ks * 64 + kq * 16));
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[0][ks]), xb, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[1][ks]), xb, acc[1], 0, 0, 0);
                }
            }
            PL_ST(3);
            // cell update (2 cells per lane) -> all six outputs into the staging image [array][row][unit]
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const float vi = sigmoid_fast(acc[jt][0] + gx[jt][0]), vf = sigmoid_fast(acc[jt][1] + gx[jt][1]);
                const float vg = tanh_fast(acc[jt][2] + gx[jt][2]), vo = sigmoid_fast(acc[jt][3] + gx[jt][3]);
                c_state[jt] = cell_c(vf, c_state[jt], vi, vg);
                const float vh = vo * tanh_fast(c_state[jt]);
                unsigned char* o = ost + lr * ORS + ul[jt] * 2;
                *reinterpret_cast<unsigned short*>(o) = bf16_bits(vh);
                *reinterpret_cast<unsigned short*>(o + 1 * 16 * ORS) = bf16_bits(vi);
                *reinterpret_cast<unsigned short*>(o + 2 * 16 * ORS) = bf16_bits(vf);
                *reinterpret_cast<unsigned short*>(o + 3 * 16 * ORS) = bf16_bits(vg);
                *reinterpret_cast<unsigned short*>(o + 4 * 16 * ORS) = bf16_bits(vo);
                *reinterpret_cast<unsigned short*>(o + 5 * 16 * ORS) = bf16_bits(c_state[jt]);
            }
            // (behind the cell update: no branch between the MFMA chain and the reads of its accumulators -- DESIGN.md section 11)
            if (t == t_begin && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            // hand-off first (threads 0..63: 16 rows x four 16-byte pieces), then the five stash arrays (320 pieces)
            if (tid < 64) {
                const int row = tid >> 2, qt = tid & 3, rb = gs * g + row;
                if (row < gs && rb < Bp) {
                    const uint4 hv = *reinterpret_cast<const uint4*>(ost + row * ORS + qt * 16);
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
                    u32x4 d;
                    d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
                    const unsigned off = (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
            }
            asm volatile("" ::: "memory");   // keep the stash stores behind the hand-off
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i;                 // piece: array e / 64 (0..4 -> gates i f g o, c), row (e % 64) / 4, quarter e % 4
                if (e < 320) {
                    const int arr = e >> 6, row = (e & 63) >> 2, qt = e & 3, rb = gs * g + row;
                    if (row < gs && rb < Bp) {
                        const uint4 sv = *reinterpret_cast<const uint4*>(ost + (arr + 1) * 16 * ORS + row * ORS + qt * 16);
                        bf16_t* dst = arr < 4 ? G + (size_t)t * slabG + (size_t)rb * G4 + arr * Hp + 32 * p + 8 * qt
                                              : Cs + (size_t)t * slabH + (size_t)rb * Hp + 32 * p + 8 * qt;
                        *reinterpret_cast<uint4*>(dst) = sv;
                    }
                }
            }
            PL_ST(4);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the hand-off store is older than the (at most 2) stash stores
            PL_ST(5);
            publish<2>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);
        }
        if (t_end < T && ok) {
            a.carry[(size_t)b * Hp + 32 * p + ul[0]] = c_state[0];
            a.carry[(size_t)b * Hp + 32 * p + ul[1]] = c_state[1];
        }
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------
// backward (reduce-scatter of partial dh tiles; backward-DATA only)
// ---------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, PL16_OCC) void lstm_bwd16_rs_sweep_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS;
    constexpr int P = Hp / 32;
    constexpr int NTT = Hp / 16;                 // N tiles of 16 hidden units
    constexpr int NT = (NTT + 3) / 4;            // per wave (wave w: tiles w, w + 4, ...)
    constexpr int DRS = 128 * 2 + 16;            // dA image [16 batch rows][128 local gate rows] bf16
    constexpr int ORS = Hp * 2 + 16;             // partial image [16 batch rows][Hp] bf16
    constexpr int NST = (P * 64 + 255) / 256;    // hand-off stores (16 B) per thread per step
    __shared__ __attribute__((aligned(16))) unsigned char da_img[16 * DRS];
    // the wide ingest also uses this image for the four waves' f32 sums ([4][16 rows][36] floats = 9216 bytes): narrow models
    // (Hp < 288) need it larger than the partial image itself -- it was not, and their sums ran past the end of the array
    constexpr int RED_BYTES = 4 * 16 * 36 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char out_img[16 * ORS > RED_BYTES ? 16 * ORS : RED_BYTES];
    __shared__ int lds_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_res = gridDim.x / P;
    const int g_first = blockIdx.x % n_res, p = blockIdx.x / n_res;
    const int Bp = a.Bp, T = a.T, G4 = 4 * Hp;
    const int t_begin = a.t0, t_end = a.t1 > 0 ? a.t1 : T;   // time chunk of this launch
    const int gs = a.group_rows;
    const int n_groups = (Bp + gs - 1) / gs;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(a.W);   // Whh^T packed [Hp][4*Hp]
    const int lr = lane & 15, kq = lane >> 4;

    // weights -> registers: tile nt = wave + 4 i: A row lr = hidden column n = 16 nt + lr; k chunk kc (= gate kc): local k = 32 kc + 8 kq + jj
    // <-> gate row kc * Hp + 32 p + 8 kq + jj
    uint4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 16 * (nt < NTT ? nt : 0) + lr;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) wreg[i][kc] = *reinterpret_cast<const uint4*>(WT + (size_t)n * G4 + kc * Hp + 32 * p + 8 * kq);
    }

    // cell ownership: thread -> batch row tid >> 4, hidden units 32p + 2 (tid & 15), +1
    const int erow = tid >> 4, jq = tid & 15;
    const int j = 32 * p + 2 * jq;
    PL_ST_DECL
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(a.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(a.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(a.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(a.dh_last);
    // exchange [2 slots][groups][P destinations][P sources][16 rows][32 columns] bf16 (1-KB tiles)
    bf16_t* __restrict__ X = static_cast<bf16_t*>(a.xchg);
    constexpr size_t TILE = 16 * 32;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    auto ld2 = [](const bf16_t* q, float (&f)[2]) {
        const unsigned u = *reinterpret_cast<const unsigned*>(q);
        f[0] = bf16_to_f32((unsigned short)(u & 0xffffu));
        f[1] = bf16_to_f32((unsigned short)(u >> 16));
    };
    auto pk2 = [](float x, float y) -> unsigned { return (unsigned)bf16_bits(x) | ((unsigned)bf16_bits(y) << 16); };

    for (int g = g_first; g < n_groups; g += n_res) {
        const int b = gs * g + erow;
        const bool ok = erow < gs && b < Bp;
        const int bc = ok ? b : Bp - 1;
        float dc_next[2] = {0.f, 0.f};
        if (t_end < T) {
            dc_next[0] = a.carry[(size_t)bc * Hp + j];
            dc_next[1] = a.carry[(size_t)bc * Hp + j + 1];
        }
        int* xtab = a.xcc_tab + (size_t)g * 64;
        bool plain_handoff = false;
        int* cnt = a.counters + (size_t)g * T * a.flag_stride;

        for (int t = t_end - 1; t >= t_begin; --t) {
            const bf16_t* g_row = G + (size_t)t * slabG + (size_t)bc * G4 + j;
            float gi[2], gf[2], gg[2], go[2], c[2], cp[2] = {0.f, 0.f}, dh[2] = {0.f, 0.f};
            ld2(g_row, gi);
            ld2(g_row + Hp, gf);
            ld2(g_row + 2 * Hp, gg);
            ld2(g_row + 3 * Hp, go);
            ld2(Cs + (size_t)t * slabH + (size_t)bc * Hp + j, c);
            if (t > 0) ld2(Cs + (size_t)(t - 1) * slabH + (size_t)bc * Hp + j, cp);
            if (dhe) ld2(dhe + (size_t)t * slabH + (size_t)bc * Hp + j, dh);
            else if (dhl && t == T - 1) ld2(dhl + (size_t)bc * Hp + j, dh);
            PL_ST(0);
            if (t + 1 < T) {
                if (!wait_arrivals(cnt + (size_t)(t + 1) * a.flag_stride, P, plain_handoff, a.status, &lds_flag, a.spin_ticks, a.poll_mask)) return;
                if (t == t_end - 2 && a.xcd_fast) plain_handoff = group_on_one_xcd(xtab, P, &lds_flag);
                PL_ST(1);
                const bf16_t* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * P * TILE;
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 2));
                if (a.stash_via_lds & 4) {
                    // wide ingest: wave w sums the tiles of sources w * TPG .. with 16-byte loads (one wave instruction = one whole
                    // 1-KB tile: 23 instead of 92 load instructions per workgroup and step), the four waves' f32 sums meet in LDS
                    // (the partial image of the previous step is free by now).  Fixed order: deterministic.
                    constexpr int TPG = (P + 3) / 4;
                    uint4 pw[TPG];
#pragma unroll
                    for (int i = 0; i < TPG; ++i) {
                        const int src = wave * TPG + i;
                        pw[i] = src < P ? ld16_handoff(rx, (unsigned)(src * TILE * 2 + lane * 16), plain_handoff) : make_uint4(0, 0, 0, 0);
                    }
                    float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < TPG; ++i) {
                        float f[4];
                        unpack_bf16x4(make_uint2(pw[i].x, pw[i].y), f);
                        acc8[0] += f[0]; acc8[1] += f[1]; acc8[2] += f[2]; acc8[3] += f[3];
                        unpack_bf16x4(make_uint2(pw[i].z, pw[i].w), f);
                        acc8[4] += f[0]; acc8[5] += f[1]; acc8[6] += f[2]; acc8[7] += f[3];
                    }
                    float* redw = reinterpret_cast<float*>(out_img) + wave * (16 * 36);   // [wave][16 rows][32 + 4 pad] f32
                    const int row = lane >> 2, c8 = lane & 3;
                    *reinterpret_cast<float4*>(redw + row * 36 + c8 * 8) = make_float4(acc8[0], acc8[1], acc8[2], acc8[3]);
                    *reinterpret_cast<float4*>(redw + row * 36 + c8 * 8 + 4) = make_float4(acc8[4], acc8[5], acc8[6], acc8[7]);
                    __syncthreads();
                    const float* rd = reinterpret_cast<const float*>(out_img) + erow * 36 + 2 * jq;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        dh[0] += rd[w * (16 * 36)];
                        dh[1] += rd[w * (16 * 36) + 1];
                    }
                    __syncthreads();   // the image is written again by this step's epilogues
                } else {
                const unsigned o0 = (unsigned)((erow * 32 + 2 * jq) * 2);
                unsigned pv[P];
#pragma unroll
                for (int s = 0; s < P; ++s)
                    pv[s] = plain_handoff ? __builtin_amdgcn_raw_buffer_load_b32(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxNt)
                                          : __builtin_amdgcn_raw_buffer_load_b32(rx, o0 + (unsigned)(s * TILE * 2), 0, kAuxSc1);
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    dh[0] += bf16_to_f32((unsigned short)(pv[s] & 0xffffu));
                    dh[1] += bf16_to_f32((unsigned short)(pv[s] >> 16));
                }
                }
            }
            PL_ST(2);
            float dai[2], daf[2], dag[2], dao[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float tc = tanh_fast(c[u]);
                const float dc = dc_next[u] + dh[u] * go[u] * (1.f - tc * tc);
                dai[u] = dc * gg[u] * gi[u] * (1.f - gi[u]);
                daf[u] = dc * cp[u] * gf[u] * (1.f - gf[u]);
                dag[u] = dc * gi[u] * (1.f - gg[u] * gg[u]);
                dao[u] = dh[u] * tc * go[u] * (1.f - go[u]);
                dc_next[u] = dc * gf[u];
            }
            const unsigned pi = pk2(dai[0], dai[1]), pf = pk2(daf[0], daf[1]), pg = pk2(dag[0], dag[1]), po = pk2(dao[0], dao[1]);
            if (ok) {   // dA_t overwrites the gate stash in place (read later by the dX / dH GEMM launches)
                bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                *reinterpret_cast<unsigned*>(go_) = pi;
                *reinterpret_cast<unsigned*>(go_ + Hp) = pf;
                *reinterpret_cast<unsigned*>(go_ + 2 * Hp) = pg;
                *reinterpret_cast<unsigned*>(go_ + 3 * Hp) = po;
            }
            if (t == 0) break;   // nobody consumes the partials of step 0
            if (t == t_end - 1 && tid == 0) __hip_atomic_store(xtab + p, xcc_id_plus1(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 32 + unit]
                unsigned char* drow = da_img + erow * DRS + jq * 4;
                *reinterpret_cast<unsigned*>(drow) = pi;
                *reinterpret_cast<unsigned*>(drow + 64) = pf;
                *reinterpret_cast<unsigned*>(drow + 128) = pg;
                *reinterpret_cast<unsigned*>(drow + 192) = po;
            }
            __syncthreads();
            PL_ST(3);
            uint4 bfr[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) bfr[kc] = *reinterpret_cast<const uint4*>(da_img + lr * DRS + kc * 64 + kq * 16);
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            const bool own_store = (a.stash_via_lds & 2) != 0;   // every wave hands its own tiles over right behind their MFMAs
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int nt = wave + 4 * i;
                if (4 * i + 3 >= NTT && nt >= NTT) break;   // a compile-time fact for all but a wave's last tile
                // (the break sits IN FRONT of a tile's MFMAs, behind the previous tile's epilogue: no MFMA result is read across it --
                // tools/isa_mfma_hazard_scan.py, profiles/r04_isa_stale_accumulator.txt)
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[i][kc]), __builtin_bit_cast(bf16x8, bfr[kc]), acc, 0, 0, 0);
                // acc[r] = partial[n = 16 nt + 4 kq + r][batch lr] -> bf16 image [batch][n]
                *reinterpret_cast<uint2*>(out_img + lr * ORS + (16 * nt + 4 * kq) * 2) = pack_bf16x4(acc[0], acc[1], acc[2], acc[3]);
                if (own_store && lane < 32) {
                    // the 16 x 16 tile is this wave's alone: read it back by rows (a wave's LDS operations are ordered) and store the
                    // 32 16-byte chunks now; it is the (nt & 1) half of destination nt >> 1's [16 rows][32 columns] exchange tile
                    const int r = lane >> 1, hc = lane & 1;
                    const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (16 * nt + 8 * hc) * 2);
                    u32x4 d;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    const unsigned off = (unsigned)(((size_t)(nt >> 1) * P * TILE + r * 32 + (nt & 1) * 16 + hc * 8) * 2);
                    if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                }
            }
            PL_ST(4);
            if (!own_store) {   // hand-off after a barrier: P tiles of [16 rows][32 columns], whole 16-byte chunks
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    const int e = tid + 256 * i;      // chunk: destination e / 64, row (e % 64) / 4, quarter e % 4
                    if (e < P * 64) {
                        const int dst = e >> 6, r = (e & 63) >> 2, c4 = e & 3;
                        const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (32 * dst + 8 * c4) * 2);
                        u32x4 d;
                        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                        const unsigned off = (unsigned)(((size_t)dst * P * TILE + (e & 63) * 8) * 2);
                        if (plain_handoff) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                    }
                }
            }
            PL_ST(5);
            publish<0>(cnt + (size_t)t * a.flag_stride + p, plain_handoff);
            PL_ST(6);
        }
        if (t_begin > 0 && ok) {
            a.carry[(size_t)b * Hp + j] = dc_next[0];
            a.carry[(size_t)b * Hp + j + 1] = dc_next[1];
        }
    }
    PL_ST_DUMP(a.stamps);
}

#define PL_SWEEP16_KS_LIST(X) X(2) X(4) X(6) X(8) X(12) X(16) X(24) X(32) X(46) X(48)

// 16-row groups pay when all of them are resident at once: measured up to 128 rows at H = 720 (8 groups x 23 workgroups =
// 184 CUs, one group per XCD slot), and for narrow models at any batch that fills at most half the chip (H = 180: 6
// workgroups per group, 256 rows = 16 groups = 96 CUs: cfg3_setB 8.55 -> 8.08 ms per iteration)
bool lstm_sweep16_wanted(int Hp, int Bp, int n_cu) {
    const int P = Hp / 32;
    if (P < 1) return false;
    const int groups = (Bp + 15) / 16;
    if (Bp <= 16) return true;
    return groups * P <= PL16_OCC * n_cu && (Bp <= 128 * PL16_OCC || groups * P <= n_cu / 2);
}

int lstm_sweep16_grid(int Hp, int Bp, int n_cu, bool spread_small) {
    const int P = Hp / 32, groups = (Bp + 15) / 16;
    int res = PL16_OCC * n_cu / P;
    if (res < 1) return 0;
    if (spread_small && groups < 8 && res >= 8) return 8 * P;   // 8 group slots keep a group on one XCD (lstm_persist.hip)
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
    }
    return res * P;
}

void launch_lstm_sweep16(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a) {
    const int ksx = (!backward && a.x_in) ? a.in_p / 32 : 0;
#define PL_CASE(K)                                                                                                    \
    if (Hp == 16 * K) {                                                                                               \
        if (backward) hipLaunchKernelGGL(lstm_bwd16_rs_sweep_kernel<K>, dim3(grid), dim3(256), 0, stream, a);         \
        else if (ksx == 1) hipLaunchKernelGGL((lstm_fwd16_sweep_kernel<K, 1>), dim3(grid), dim3(256), 0, stream, a);  \
        else if (ksx == 2) hipLaunchKernelGGL((lstm_fwd16_sweep_kernel<K, 2>), dim3(grid), dim3(256), 0, stream, a);  \
        else hipLaunchKernelGGL((lstm_fwd16_sweep_kernel<K, 0>), dim3(grid), dim3(256), 0, stream, a);                \
        return;                                                                                                       \
    }
    PL_SWEEP16_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
