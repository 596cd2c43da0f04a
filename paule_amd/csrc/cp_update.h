// The CP update of one planning iteration, per element: gradient of the smoothness terms (velocity, jerk, local linearity) from the
// correlations the loss reduction left in dwork, the total gradient, torch.optim.Adam (no amsgrad / weight decay) and the projection
// behind it (paule/paule.py:797, :1199-1211).  Used by cp_update_kernel (elementwise.hip); a header of its own because round 5 also ran
// it as a role of the predictor's backward sweep (measured, not kept: NOTEBOOK.md A.15) and both had to compute the same bits.
#pragma once
#include "kernels.h"

namespace pl {

// correlation taps: d[t] = sum_k taps[k] * x[t + k]
//   velocity     : five-point stencil, paule/util.py:600 (delta_t = 1, paule/paule.py:78)
//   jerk         : the same stencil applied three times (paule/util.py:633-636) = one 13-tap correlation
//   local linear : paule/util.py:614
static __constant__ double kVelTaps[5] = {1.0 / 12, -8.0 / 12, 0.0, 8.0 / 12, -1.0 / 12};
static __constant__ double kJerkTaps[13] = {1.0 / 1728,    -24.0 / 1728, 192.0 / 1728,  -488.0 / 1728, -387.0 / 1728,
                                            1584.0 / 1728, 0.0,          -1584.0 / 1728, 387.0 / 1728,  488.0 / 1728,
                                            -192.0 / 1728, 24.0 / 1728,  -1.0 / 1728};
static __constant__ double kLlTaps[3] = {-0.5, 1.0, -0.5};

// loss = w * mean(d^2), d[u] = sum_j taps[j] x[u+j], u in [0, n)  =>  dloss/dx[t] = w*2/(n*C) * sum_k taps[k] d[t-k]
// The correlations d[u] (computed by loss_reduce_kernel of the same iteration, dwork [B][3][T][C]) are read with raw buffer loads: rd covers
// the utterances from a wave-uniform first one on (dwork_rsrc), off0 is the byte offset of this thread's channel at u = 0, and an index
// outside [0, n) gets an offset beyond the buffer's range, which reads 0 -- no branch.  (As plain loads under `u >= 0 && u < n` every one of
// them sat in a block of its own and was waited for before the next was issued -- 21 memory latencies in a row per element, 35 us per pair
// of frames in the trailing update role; a clamped index with a select is turned back into that form by the compiler.)
// TG consecutive frames t0 .. t0 + TG - 1 of one channel: the K + TG - 1 correlations are loaded once.
constexpr unsigned kDworkRangeMax = 0x7ffffff0u, kDworkOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dwork_rsrc(const AdamArgs& a, int b_first) {
    const size_t per3 = (size_t)3 * a.T * a.C;
    const size_t bytes = (size_t)(a.B - b_first) * per3 * 8;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.dwork + (size_t)b_first * per3), 0,
                                             bytes < kDworkRangeMax ? (unsigned)bytes : kDworkRangeMax, 0x00020000);
}
template <int K, int TG>
__device__ __forceinline__ void corr_grad_run(__amdgpu_buffer_rsrc_t rd, unsigned off0, int T, int C, const double* taps, int t0, double w,
                                              double (&g)[TG]) {
    typedef __attribute__((ext_vector_type(2))) unsigned int cg_u32x2;
    const int n = T - K + 1;
    double d[K + TG - 1];                       // d[i] = correlation at u = t0 - (K - 1) + i, zero outside [0, n)
#pragma unroll
    for (int i = 0; i < K + TG - 1; ++i) {
        const int u = t0 - (K - 1) + i;
        const cg_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rd, (u >= 0 && u < n) ? off0 + (unsigned)(u * C * 8) : kDworkOob, 0, 0);
        d[i] = __builtin_bit_cast(double, v);
    }
    const double scale = w * 2.0 / ((double)n * C);
#pragma unroll
    for (int j = 0; j < TG; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += taps[k] * d[(K - 1) + j - k];   // u = t0 + j - k
        g[j] += acc * scale;
    }
}

// the three smoothness gradients of TG consecutive frames of channel c of utterance b; rd = dwork_rsrc(a, b_first) with b_first <= b
// wave-uniform and near (the byte offsets are 32-bit)
template <int TG>
__device__ __forceinline__ void smooth_grads(const AdamArgs& a, __amdgpu_buffer_rsrc_t rd, int b_first, int b, int c, int t0, double (&gv)[TG],
                                             double (&gj)[TG], double (&gl)[TG]) {
    const unsigned per = (unsigned)(a.T * a.C * 8);
    const unsigned off = (unsigned)(b - b_first) * 3u * per + (unsigned)(c * 8);
    corr_grad_run<5, TG>(rd, off, a.T, a.C, kVelTaps, t0, (double)a.w_vel, gv);
    corr_grad_run<13, TG>(rd, off + per, a.T, a.C, kJerkTaps, t0, (double)a.w_jerk, gj);
    corr_grad_run<3, TG>(rd, off + 2u * per, a.T, a.C, kLlTaps, t0, (double)a.w_ll, gl);
}

// total gradient of one element from the model gradient(s) and the three smoothness terms, in this order
__device__ __forceinline__ double total_grad_of(double dx, bool has2, double dx2, double gv, double gj, double gl) {
    double g = dx;
    if (has2) g += dx2;
    g += gv;
    g += gj;
    g += gl;
    return g;
}

// the step's scalars, k = *step_count + 1: lr / (1 - beta1^k) and sqrt(1 - beta2^k) (once per workgroup; the per-element expressions
// (lr / bc1) * (m / denom) and sqrt(v) / sqrt(bc2) keep their operands)
__device__ __forceinline__ void adam_step_scalars(const AdamArgs& a, double& lr_bc1, double& sqrt_bc2) {
    const int k = *a.step_count + 1;
    const double bc1 = 1.0 - pow(a.beta1, (double)k);
    const double bc2 = 1.0 - pow(a.beta2, (double)k);
    lr_bc1 = a.lr / bc1;
    sqrt_bc2 = sqrt(bc2);
}

// Adam on one element (b, t, c) of the CP tensor with total gradient g, then clamp / smiling / past_cp: (m, v, x) -> their new values
__device__ __forceinline__ void adam_value(const AdamArgs& a, int b, int t, int c, double g, double lr_bc1, double sqrt_bc2, double& m, double& v,
                                           double& x) {
    m = a.beta1 * m + (1.0 - a.beta1) * g;
    v = a.beta2 * v + (1.0 - a.beta2) * g * g;
    const double denom = sqrt(v) / sqrt_bc2 + a.eps;
    x = x - lr_bc1 * (m / denom);
    x = fmin(fmax(x, a.clamp_lo), a.clamp_hi);
    if (a.smiling) {
        if (c == 4) x = -1.0;   // "LP"
        if (c == 1) x = 1.0;    // "HY"
    }
    if (a.past && t < a.past_len)
        x = a.past[((size_t)(a.past_per_utt ? b : 0) * a.past_len + t) * a.C + c];
}

}  // namespace pl
