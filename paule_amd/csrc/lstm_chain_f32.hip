// f32 persistent LSTM sweeps for batches of MORE groups than the chip holds at once: chains.
//
// lstm_persist_f32.hip cuts the batch into groups of 16 rows served by P = Hp / 16 workgroups each (46 at H = 720); 5 groups
// are resident on 256 CUs, and a workgroup with more than one group sweeps them one AFTER the other (all T steps each):
// B = 256 = 16 groups = 4 rounds of ~5.5 / 7 us per step -- no better than the launch-per-step kernels, which is where such
// batches went.  Here a workgroup serves C groups ("chains") with its ONE copy of the W_hh slice in registers and takes a
// step of each in turn, as the roles of lstm_fused.hip do: while group A's hand-off is in flight, group B multiplies.  The
// flags of the next chain-step are looked at while this one's MFMAs run, its operands are loaded under this one's cell update
// and stores (forward) / under this one's MFMAs (backward), so a chain-step costs the workgroup's busy time -- the f32
// 16x16x4 MFMA chain, 2.6 / 3.2 us of it -- instead of a step's latency.
//
// Arithmetic, layouts, flags and exchange buffers are those of lstm_persist_f32.hip (same MFMA order per output element, same
// fixed-order sum of the partial tiles), so both produce the same bits; the hand-off is always write-through (46 workgroups
// of a group never sit on one XCD).  Whole sequences only (no time chunks: the layer wavefront is for batches that leave CUs
// free).  Bp is a multiple of 16: groups are whole, nothing is guarded.
#include "sweep_common.h"

namespace pl {

namespace {

constexpr int kChainMax = 8;   // chains per workgroup (LDS for the cell state of that many groups)

__device__ __forceinline__ f32x4 mfma4c(const float4& a, const float4& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void st16_sc1f(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
    u32x4 d;
    d[0] = __builtin_bit_cast(unsigned, v.x);
    d[1] = __builtin_bit_cast(unsigned, v.y);
    d[2] = __builtin_bit_cast(unsigned, v.z);
    d[3] = __builtin_bit_cast(unsigned, v.w);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, kAuxSc1);
}
// one look of wave 0 at the P arrival flags of a step (write-through: the memory side answers)
__device__ __forceinline__ int poll_flags(const int* flags, int P, int lane) {
    const __amdgpu_buffer_rsrc_t rf = make_rsrc(flags, (unsigned)(P * 4));
    return lane < P ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1) : 1;
}
// all storing waves have drained; one lane raises the workgroup's flag, write-through
__device__ __forceinline__ void raise(int* flag, int wave) {
    __syncthreads();
    if (wave == 0) {
        if ((threadIdx.x & 63) == 0) {
            const __amdgpu_buffer_rsrc_t rf = make_rsrc(flag, 4u);
            __builtin_amdgcn_raw_buffer_store_b32(1u, rf, 0u, 0, kAuxSc1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward (all-gather of h); KSX > 0: input projection fused (in_p = 16 * KSX)
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int KSX>
__global__ __launch_bounds__(256, 1) void lstm_fwd_chain_f32_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS, G4 = 4 * Hp;
    constexpr int ROWB = Hp * 4, RS = ROWB + 16;   // LDS row stride: odd number of 16-byte chunks
    constexpr int CPR = Hp / 4;
    constexpr int NLD = (16 * CPR + 255) / 256;
    constexpr int PF = 4;
    constexpr int HRS = 64 + 16;
    constexpr int XRS = KSX * 64 + 16;
    constexpr int P = KS;
    constexpr int PK = KS / 4;                     // k-chunk at which wave 0 looks at the next chain-step's flags
    constexpr int INP = KSX ? 16 * KSX : 16, XC = INP / 4;
    __shared__ __attribute__((aligned(16))) unsigned char himg[16 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char hst[6 * 16 * HRS];
    __shared__ __attribute__((aligned(16))) unsigned char ximg[KSX ? 16 * XRS : 16];
    __shared__ float cst[kChainMax * 256];
    __shared__ int lflag[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Bp = a.Bp, T = a.T, C = a.chains;
    const int n_groups = Bp / 16, n_sets = (n_groups + C - 1) / C;
    const int sets_res = gridDim.x / P;
    const int set_first = blockIdx.x % sets_res, p = blockIdx.x / sets_res;
    const float* __restrict__ W = static_cast<const float*>(a.W);
    const int kq = lane >> 4, bl = lane & 15;

    float4 wreg[KS];
    {
        const int r = lane & 15;
        const float* wrow = W + (size_t)((r & 3) * Hp + 16 * p + 4 * wave + (r >> 2)) * Hp + 4 * kq;
#pragma unroll
        for (int s = 0; s < KS; ++s) wreg[s] = *reinterpret_cast<const float4*>(wrow + 16 * s);
    }
    const int j = 16 * p + 4 * wave + kq;
    float4 wx[KSX ? KSX : 1];
    f32x4 bias_r = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KSX > 0) {
        const int r = lane & 15;
        const float* xrow = static_cast<const float*>(a.Wih) + (size_t)((r & 3) * Hp + 16 * p + 4 * wave + (r >> 2)) * INP + 4 * kq;
#pragma unroll
        for (int s = 0; s < KSX; ++s) wx[s] = *reinterpret_cast<const float4*>(xrow + 16 * s);
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) bias_r[gate] = a.bias[gate * Hp + j];
    }
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    float* __restrict__ G = static_cast<float*>(a.G);
    float* __restrict__ Hs = static_cast<float*>(a.h);
    float* __restrict__ Cs = static_cast<float*>(a.c);
    const size_t cnt_stride = (size_t)T * a.flag_stride;
    PL_ST_DECL

    for (int set = set_first; set < n_sets; set += sets_res) {
        const int g0 = set * C;
        int Ca = n_groups - g0;
        Ca = Ca < C ? Ca : C;

        // operands of the NEXT chain-step, in flight while the current one computes
        uint4 hv[NLD];
        uint4 xv = make_uint4(0, 0, 0, 0);
        float gxn[4] = {0.f, 0.f, 0.f, 0.f};
        auto issue_loads = [&](int g2, int t2) {
            // the small rows first: vmcnt retires in order, and the cell update waits for them, not for the h tile
            if constexpr (KSX > 0) {
                if (wave < XC / 4) {   // 16 x XC threads = XC / 4 whole waves
                    const int row = tid / XC, cc = tid % XC;
                    xv = *reinterpret_cast<const uint4*>(static_cast<const float*>(a.x_in) + ((size_t)t2 * Bp + 16 * g2 + row) * INP + cc * 4);
                }
            } else {
                const float* g_row = G + (size_t)t2 * slabG + (size_t)(16 * g2 + bl) * G4 + j;
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) gxn[gate] = g_row[gate * Hp];
            }
            if (t2 > 0) {
                const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t2 - 1) * slabH, (unsigned)(slabH * 4));
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + 256 * i;
                    hv[i] = (e < 16 * CPR) ? ld16_sc1(rh, (unsigned)((16 * g2 + e / CPR) * ROWB + (e % CPR) * 16)) : make_uint4(0, 0, 0, 0);
                }
            }
        };

        __syncthreads();   // the LDS images of the previous set are done with
        issue_loads(g0, 0);
        int c = 0, t = 0;
        for (;;) {
            const int g = g0 + c;
            int cn = c + 1, tn = t;
            if (cn == Ca) { cn = 0; tn = t + 1; }
            const bool has_next = tn < T;
            const int gn = g0 + cn;
            int* const cnt = a.counters + (size_t)g * cnt_stride;

            // A. the prefetched operands of this chain-step -> LDS
            float gx[4] = {gxn[0], gxn[1], gxn[2], gxn[3]};
            if (t > 0) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + 256 * i;
                    if (e < 16 * CPR) *reinterpret_cast<uint4*>(himg + (e / CPR) * RS + (e % CPR) * 16) = hv[i];
                }
            }
            if constexpr (KSX > 0) {
                if (wave < XC / 4) *reinterpret_cast<uint4*>(ximg + (tid / XC) * XRS + (tid % XC) * 16) = xv;
            }
            __syncthreads();
            PL_ST(0);

            // B / C. gates = W_hh h_{t-1} (+ W_ih x_t + b); wave 0 looks at the next chain-step's flags on the way
            const bool poll_here = wave == 0 && has_next && tn > 0;
            const int* const nflags = a.counters + (size_t)gn * cnt_stride + (size_t)(tn > 0 ? tn - 1 : 0) * a.flag_stride;
            int pv = 1;
            f32x4 acc = KSX > 0 ? bias_r : f32x4{0.f, 0.f, 0.f, 0.f};
            if (t > 0) {
                const unsigned char* bsrc = himg + bl * RS + kq * 16;
                float4 bq[PF];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < KS) bq[i] = *reinterpret_cast<const float4*>(bsrc + i * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc = mfma4c(wreg[s], bq[s % PF], acc);
                    if (s + PF < KS) bq[s % PF] = *reinterpret_cast<const float4*>(bsrc + (s + PF) * 64);
                    if (s == PK && poll_here) pv = poll_flags(nflags, P, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (poll_here) {
                pv = poll_flags(nflags, P, lane);
            }
            if constexpr (KSX > 0) {
#pragma unroll
                for (int s = 0; s < KSX; ++s)
                    acc = mfma4c(wx[s], *reinterpret_cast<const float4*>(ximg + bl * XRS + s * 64 + kq * 16), acc);
            }
            PL_ST(1);

            // D. has the next chain-step everything it waits for?  E. then its operands start now
            if (wave == 0) {
                const bool rdy = __all(pv != 0);
                if (lane == 0) lflag[0] = rdy ? 1 : 0;
            }
            __syncthreads();
            const bool ready = has_next && lflag[0] != 0;
            if (ready) issue_loads(gn, tn);
            PL_ST(2);

            // F. cell (libm forms: the f32 path carries the 1e-5 parity bar)
            float c_state = t > 0 ? cst[c * 256 + tid] : 0.f;
            const float vi = sigmoid_f(acc[0] + gx[0]), vf = sigmoid_f(acc[1] + gx[1]);
            const float vg = tanhf(acc[2] + gx[2]), vo = sigmoid_f(acc[3] + gx[3]);
            c_state = vf * c_state + vi * vg;
            const float vh = vo * tanhf(c_state);
            cst[c * 256 + tid] = c_state;
            {   // G. all six outputs leave through LDS as whole 64-byte row pieces
                unsigned char* o = hst + bl * HRS + (4 * wave + kq) * 4;
                *reinterpret_cast<float*>(o) = vh;
                *reinterpret_cast<float*>(o + 1 * 16 * HRS) = vi;
                *reinterpret_cast<float*>(o + 2 * 16 * HRS) = vf;
                *reinterpret_cast<float*>(o + 3 * 16 * HRS) = vg;
                *reinterpret_cast<float*>(o + 4 * 16 * HRS) = vo;
                *reinterpret_cast<float*>(o + 5 * 16 * HRS) = c_state;
            }
            __syncthreads();
            if (wave == 0) {
                const int row = tid >> 2, qt = tid & 3;
                const float4 hvv = *reinterpret_cast<const float4*>(hst + row * HRS + qt * 16);
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 4));
                st16_sc1f(ro, (unsigned)(((16 * g + row) * Hp + 16 * p + 4 * qt) * 4), hvv);
            }
            asm volatile("" ::: "memory");   // keep the stash stores behind it
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 0 || wave == 0) {   // 320 pieces: all threads once, wave 0 a second time
                    const int e = tid + 256 * i;
                    const int arr = e >> 6, row = (e & 63) >> 2, qt = e & 3, rb = 16 * g + row;
                    const float4 sv = *reinterpret_cast<const float4*>(hst + (arr + 1) * 16 * HRS + row * HRS + qt * 16);
                    float* dst = arr < 4 ? G + (size_t)t * slabG + (size_t)rb * G4 + arr * Hp + 16 * p + 4 * qt
                                         : Cs + (size_t)t * slabH + (size_t)rb * Hp + 16 * p + 4 * qt;
                    *reinterpret_cast<float4*>(dst) = sv;
                }
            }
            PL_ST(3);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the hand-off store is older than the (at most 2) stash stores
            raise(cnt + (size_t)t * a.flag_stride + p, wave);
            PL_ST(4);

            if (!has_next) break;
            // H. the next chain-step was not ready at the first look
            if (!ready) {
                if (tn > 0 && !wait_arrivals(nflags, P, false, a.status, &lflag[1], a.spin_ticks, a.poll_mask)) return;
                issue_loads(gn, tn);
            }
            PL_ST(5);
            c = cn;
            t = tn;
        }
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, reduce-scatter of f32 partial dh tiles (backward-DATA only)
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_chain_f32_kernel(LstmSweepArgs a) {
    constexpr int Hp = 16 * KS, G4 = 4 * Hp;
    constexpr int P = KS;
    constexpr int NT = (P + 3) / 4;              // N tiles per wave (wave w: tiles w, w + 4, ...)
    constexpr int TPG = (P + 3) / 4;             // partial tiles a wave sums
    constexpr int DRS = 64 * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char da_img[16 * DRS];
    __shared__ __attribute__((aligned(16))) float red[4][16][16];
    __shared__ float dcs[kChainMax * 256];
    __shared__ int lflag[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Bp = a.Bp, T = a.T, C = a.chains;
    const int n_groups = Bp / 16, n_sets = (n_groups + C - 1) / C;
    const int n_main = (int)gridDim.x - a.n_pf;
    const int sets_res = n_main / P;
    if ((int)blockIdx.x >= n_main) {   // prefetcher workgroups (sweep_common.h: speed only): a set's (step, chain) pairs in the sweep's order
        stash_prefetch_walk(a, Hp, 4, (int)blockIdx.x - n_main, sets_res, T - 1, 0, PfPaceCounters{a.counters, T, a.flag_stride}, C);
        return;
    }
    const int set_first = blockIdx.x % sets_res, p = blockIdx.x / sets_res;
    const float* __restrict__ WT = static_cast<const float*>(a.W);
    const int kq = lane >> 4;

    float4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 16 * (nt < P ? nt : 0) + (lane & 15);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) wreg[i][c4] = *reinterpret_cast<const float4*>(WT + (size_t)n * G4 + c4 * Hp + 16 * p + 4 * kq);
    }
    const int erow = tid >> 4, eu = tid & 15;
    const int j = 16 * p + eu;
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    float* __restrict__ G = static_cast<float*>(a.G);
    const float* __restrict__ Cs = static_cast<const float*>(a.c);
    const float* __restrict__ dhe = static_cast<const float*>(a.dh_ext);
    const float* __restrict__ dhl = static_cast<const float*>(a.dh_last);
    float* __restrict__ X = static_cast<float*>(a.xchg);
    constexpr size_t TILE = 16 * 16;
    const size_t grp_stride = (size_t)P * P * TILE;
    const size_t slot_stride = (size_t)n_groups * grp_stride;
    const size_t cnt_stride = (size_t)T * a.flag_stride;
    const int r16 = lane >> 2, quad = lane & 3;
    PL_ST_DECL

    for (int set = set_first; set < n_sets; set += sets_res) {
        const int g0 = set * C;
        int Ca = n_groups - g0;
        Ca = Ca < C ? Ca : C;

        // operands of the NEXT chain-step
        float sgi = 0.f, sgf = 0.f, sgg = 0.f, sgo = 0.f, sc = 0.f, scp = 0.f, sdh = 0.f;
        uint4 pw[TPG];
        auto issue_loads = [&](int g2, int t2) {
            const int b = 16 * g2 + erow;
            const float* g_row = G + (size_t)t2 * slabG + (size_t)b * G4 + j;
            sgi = g_row[0]; sgf = g_row[Hp]; sgg = g_row[2 * Hp]; sgo = g_row[3 * Hp];
            sc = Cs[(size_t)t2 * slabH + (size_t)b * Hp + j];
            scp = t2 > 0 ? Cs[(size_t)(t2 - 1) * slabH + (size_t)b * Hp + j] : 0.f;
            sdh = 0.f;
            if (dhe) sdh = dhe[(size_t)t2 * slabH + (size_t)b * Hp + j];
            else if (dhl && t2 == T - 1) sdh = dhl[(size_t)b * Hp + j];
        };
        // ... and the P partial tiles of its step t2 + 1.  Issued in one piece before this chain-step's MFMA tiles: that holds the
        // wave for ~1.1 us (the CU's memory pipe takes the 46 KB at about half its peak rate behind the tile stores), but issued
        // one by one BETWEEN the MFMA tiles the same time shows up inside the tile loop and more (5.36 -> 5.75 us per chain-step).
        auto issue_tile = [&](int g2, int t2, int i) {
            const float* xs = X + (size_t)((t2 + 1) & 1) * slot_stride + (size_t)g2 * grp_stride + (size_t)p * P * TILE;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 4));
            const int src = wave * TPG + i;
            pw[i] = (src < P && t2 + 1 < T) ? ld16_sc1(rx, (unsigned)(src * TILE * 4 + (r16 * 16 + quad * 4) * 4)) : make_uint4(0, 0, 0, 0);
        };
        auto issue_tiles = [&](int g2, int t2) {
#pragma unroll
            for (int i = 0; i < TPG; ++i) issue_tile(g2, t2, i);
        };

        __syncthreads();
        issue_loads(g0, T - 1);
#pragma unroll
        for (int i = 0; i < TPG; ++i) pw[i] = make_uint4(0, 0, 0, 0);
        int c = 0, t = T - 1;
        for (;;) {
            const int g = g0 + c;
            int cn = c + 1, tn = t;
            if (cn == Ca) { cn = 0; tn = t - 1; }
            const bool has_next = tn >= 0;
            const int gn = g0 + cn;
            int* const cnt = a.counters + (size_t)g * cnt_stride;
            const int b = 16 * g + erow;

            // the next chain-step's flags (those of step tn + 1 of its group): looked at now, answered before the MFMAs
            const bool poll_here = wave == 0 && has_next && tn + 1 < T;
            const int* const nflags = a.counters + (size_t)gn * cnt_stride + (size_t)(tn + 1 < T ? tn + 1 : 0) * a.flag_stride;
            int pv = 1;
            if (poll_here) pv = poll_flags(nflags, P, lane);

            // A. the prefetched operands: dh = external part + sum of the P partial tiles (fixed order)
            const float gi = sgi, gf = sgf, gg = sgg, go = sgo, cc = sc, cp = scp;
            float dh = sdh;
            if (t + 1 < T) {
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
            }
            PL_ST(0);

            const float dc_next = t + 1 < T ? dcs[c * 256 + tid] : 0.f;
            const float tc = tanhf(cc);
            const float dc = dc_next + dh * go * (1.f - tc * tc);
            const float dai = dc * gg * gi * (1.f - gi);
            const float daf = dc * cp * gf * (1.f - gf);
            const float dag = dc * gi * (1.f - gg * gg);
            const float dao = dh * tc * go * (1.f - go);
            dcs[c * 256 + tid] = dc * gf;
            {   // dA_t overwrites the gate stash in place
                float* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                go_[0] = dai;
                go_[Hp] = daf;
                go_[2 * Hp] = dag;
                go_[3 * Hp] = dao;
            }
            {
                float* drow = reinterpret_cast<float*>(da_img + erow * DRS) + eu;
                drow[0] = dai;
                drow[16] = daf;
                drow[32] = dag;
                drow[48] = dao;
            }
            if (wave == 0) {
                const bool rdy = __all(pv != 0);
                if (lane == 0) lflag[0] = rdy ? 1 : 0;
            }
            __syncthreads();
            PL_ST(1);
            // E. the next chain-step's operands fly under this one's MFMAs
            const bool ready = has_next && lflag[0] != 0;
            if (ready) {
                issue_loads(gn, tn);
                issue_tiles(gn, tn);
            }
            PL_ST(2);

            if (t > 0) {   // nobody consumes the partials of step 0
                float* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 4));
                float4 bfr[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) bfr[c4] = *reinterpret_cast<const float4*>(da_img + (lane & 15) * DRS + c4 * 64 + kq * 16);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int nt = wave + 4 * i;
                    if (nt < P) {
                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) acc = mfma4c(wreg[i][c4], bfr[c4], acc);
                        st16_sc1f(ro, (unsigned)(((size_t)nt * P * TILE + (lane & 15) * 16 + 4 * kq) * 4), make_float4(acc[0], acc[1], acc[2], acc[3]));
                    }
                }
                PL_ST(3);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                raise(cnt + (size_t)t * a.flag_stride + p, wave);
                PL_ST(4);
            } else {
                __syncthreads();   // da_img / red are rewritten by the next chain-step
            }
            if (!has_next) break;
            if (!ready) {
                if (tn + 1 < T && !wait_arrivals(nflags, P, false, a.status, &lflag[1], a.spin_ticks, a.poll_mask)) return;
                issue_loads(gn, tn);
                issue_tiles(gn, tn);
            }
            PL_ST(5);
            c = cn;
            t = tn;
        }
    }
    PL_ST_DUMP(a.stamps);
}

}  // namespace

#define PL_CHAIN_F32_KS_LIST(X) X(6) X(46)

bool lstm_chain_f32_supported(int Hp) {
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_CHAIN_F32_KS_LIST(PL_CASE)
#undef PL_CASE
    return false;
}

// chains per workgroup and the grid for a batch whose groups do not all fit the chip: the fewest chains that make every set
// resident (up to kChainMax: beyond that the sets take turns)
int lstm_chain_f32_plan(int Hp, int Bp, int n_cu, int forced_chains, int* grid) {
    const int P = Hp / 16, groups = (Bp + 15) / 16;
    const int res = n_cu / P;   // sets (or groups) resident at once
    *grid = 0;
    if (res < 1 || !lstm_chain_f32_supported(Hp)) return 0;
    int C = forced_chains > 0 ? forced_chains : (groups + res - 1) / res;
    if (C > kChainMax) C = kChainMax;
    if (C < 2 && forced_chains <= 0) return 0;
    if (C > groups) C = groups;
    if (C < 1) C = 1;
    const int sets = (groups + C - 1) / C;
    *grid = (sets < res ? sets : res) * P;
    return C;
}

void launch_lstm_chain_f32(hipStream_t stream, bool backward, int Hp, int grid, const LstmSweepArgs& a) {
    const int ksx = (!backward && a.x_in) ? a.in_p / 16 : 0;
#define PL_CASE(K)                                                                                                              \
    if (Hp == 16 * K) {                                                                                                         \
        if (backward) hipLaunchKernelGGL((lstm_bwd_chain_f32_kernel<K>), dim3(grid + a.n_pf), dim3(256), 0, stream, a);         \
        else if (ksx == 2) hipLaunchKernelGGL((lstm_fwd_chain_f32_kernel<K, 2>), dim3(grid), dim3(256), 0, stream, a);          \
        else if (ksx == 4) hipLaunchKernelGGL((lstm_fwd_chain_f32_kernel<K, 4>), dim3(grid), dim3(256), 0, stream, a);          \
        else hipLaunchKernelGGL((lstm_fwd_chain_f32_kernel<K, 0>), dim3(grid), dim3(256), 0, stream, a);                        \
        return;                                                                                                                 \
    }
    PL_CHAIN_F32_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
