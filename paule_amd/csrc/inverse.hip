// Kernels of the inverse model's forward pass (InverseModelMelTimeSmoothResidual, paule/models.py:177-247): the
// initialisation of a plan from the target acoustics (paule/paule.py:550-556).  Everything except the LSTM stack
// (the planner's forward sweeps) and post_linear (gemm.hip) lives here: small grouped convolutions over time on
// [B][frames][channels] f32 arrays, one thread per output element -- HBM-trivial (a few MB once per plan), written
// for exactness (f32 FMA chains in the reference's summation order are not required: parity bar 2e-5).
#include "kernels.h"
#include "pl_types.h"

namespace pl {

namespace {

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

// MelChannelConv1D(M, 3) + residual (paule/models.py:142-169, :222-227).  Three Conv1d(M -> M/3, k = 5, pad 2,
// groups = M/3): group g of conv j reads channels 3g .. 3g+2 of the input shifted by (j - 1) channels (zero filled),
// and its output is channel 3g + j.  w [3][M/3][3][5], b [3][M/3].
__global__ void mel_block_kernel(const float* __restrict__ x, int B, int Tp, int M, const float* __restrict__ w,
                                 const float* __restrict__ b, float* __restrict__ y) {
    const int64_t n = (int64_t)B * Tp * M;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % M);
    const int t = (int)((idx / M) % Tp);
    const int bb = (int)(idx / ((int64_t)M * Tp));
    const int g = c / 3, j = c % 3, G = M / 3;
    const float* wj = w + ((size_t)j * G + g) * 15;
    const float* xb = x + (size_t)bb * Tp * M;
    float acc = b[j * G + g];
    for (int q = 0; q < 3; ++q) {
        const int ch = 3 * g + q + (j - 1);          // xs[0] = channels shifted down, xs[1] = x, xs[2] = shifted up
        if (ch < 0 || ch >= M) continue;
        for (int k = 0; k < 5; ++k) {
            const int tt = t + k - 2;
            if (tt >= 0 && tt < Tp) acc += wj[q * 5 + k] * xb[(size_t)tt * M + ch];
        }
    }
    y[idx] = x[idx] + acc;
}

// add_vel_and_acc_info (paule/models.py:47-61) + time-major packing: [B][Tp][M] -> [Tp][Bp][in_p] (x, velocity, acceleration)
template <typename AT>
__global__ void vel_acc_pack_kernel(const float* __restrict__ x, int B, int Tp, int M, AT* __restrict__ dst, int Bp, int in_p) {
    const int64_t n = (int64_t)Tp * Bp * in_p;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int f = (int)(idx % in_p);
    const int bb = (int)((idx / in_p) % Bp);
    const int t = (int)(idx / ((int64_t)in_p * Bp));
    float v = 0.f;
    if (bb < B && f < 3 * M) {
        const int c = f % M, kind = f / M;
        const float* xc = x + (size_t)bb * Tp * M + c;
        auto at = [&](int tt) { return xc[(size_t)tt * M]; };
        if (kind == 0) v = at(t);
        else if (kind == 1) v = (t + 1 < Tp) ? at(t + 1) - at(t) : 0.f;
        else v = (t >= 1 && t + 1 < Tp) ? (at(t + 1) - at(t)) - (at(t) - at(t - 1)) : 0.f;
    }
    dst[idx] = from_f32<AT>(v);
}

// double_sequence (paule/models.py:63-81) of the time-major post_linear output Y [Tp][Bp][Cp] -> z [B][2 Tp][C]
__global__ void double_seq_kernel(const float* __restrict__ Y, int B, int Tp, int C, int Bp, int Cp, float* __restrict__ z) {
    const int64_t n = (int64_t)B * 2 * Tp * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const int t2 = (int)((idx / C) % (2 * Tp));
    const int bb = (int)(idx / ((int64_t)C * 2 * Tp));
    const int t = t2 >> 1;
    const float a = Y[((size_t)t * Bp + bb) * Cp + c];
    float v = a;
    if ((t2 & 1) && t + 1 < Tp) v = (a + Y[((size_t)(t + 1) * Bp + bb) * Cp + c]) / 2.0f;
    z[idx] = v;
}

// channelwise Conv1d(C, C, 5, padding 2, groups = C) over time (time_conv_1x5, paule/models.py:36-45) (+ residual)
__global__ void time_conv5_kernel(const float* __restrict__ x, int B, int T, int C, const float* __restrict__ w,
                                  const float* __restrict__ b, const float* __restrict__ resid, float* __restrict__ y) {
    const int64_t n = (int64_t)B * T * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const int t = (int)((idx / C) % T);
    const int bb = (int)(idx / ((int64_t)C * T));
    const float* xc = x + (size_t)bb * T * C + c;
    float acc = b[c];
    for (int k = 0; k < 5; ++k) {
        const int tt = t + k - 2;
        if (tt >= 0 && tt < T) acc += w[c * 5 + k] * xc[(size_t)tt * C];
    }
    if (resid) acc += resid[idx];
    y[idx] = acc;
}

// resid_weighting: Conv1d(2C -> C, 5, padding 2, groups = C) on the channel-interleaved (smoothed, lstm_output) pair
// (paule/models.py:206-208, :240-243); w [C][2][5].  Optional clip (paule/paule.py:555).
__global__ void resid_weight_kernel(const float* __restrict__ zs, const float* __restrict__ zl, int B, int T, int C,
                                    const float* __restrict__ w, const float* __restrict__ b, int clip, float* __restrict__ y) {
    const int64_t n = (int64_t)B * T * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const int t = (int)((idx / C) % T);
    const int bb = (int)(idx / ((int64_t)C * T));
    const size_t base = (size_t)bb * T * C + c;
    float acc = b[c];
    for (int k = 0; k < 5; ++k) {
        const int tt = t + k - 2;
        if (tt >= 0 && tt < T) acc += w[c * 10 + k] * zs[base + (size_t)tt * C] + w[c * 10 + 5 + k] * zl[base + (size_t)tt * C];
    }
    if (clip) acc = fminf(1.0f, fmaxf(-1.0f, acc));
    y[idx] = acc;
}

// backward-data of mel_block_kernel (embedder variants with mel smoothing in the planning loop, paule/models.py:393-401):
// dx = dy + conv^T(dy).  Element (b, t, c) of the input / output sits at b * sb + t * st + c, so the same kernel reads the
// LSTM's time-major input gradient ([Tp][Bp][Mp]: sb = Mp, st = Bp * Mp) and the batch-major intermediates.
__global__ void mel_block_bwd_kernel(const float* __restrict__ dy, int64_t sb_in, int64_t st_in, int B, int Tp, int M,
                                     const float* __restrict__ w, float* __restrict__ dx, int64_t sb_out, int64_t st_out) {
    const int64_t n = (int64_t)B * Tp * M;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % M);
    const int t = (int)((idx / M) % Tp);
    const int bb = (int)(idx / ((int64_t)M * Tp));
    const int G = M / 3;
    const float* db = dy + (size_t)bb * sb_in;
    float acc = db[(size_t)t * st_in + c];
    for (int j = 0; j < 3; ++j) {                  // conv j read input channel c as its q-th tap of group g: 3g + q + (j - 1) = c
        const int u = c - (j - 1);
        if (u < 0 || u >= M) continue;
        const int g = u / 3, q = u % 3;
        const float* wj = w + ((size_t)j * G + g) * 15 + q * 5;
        for (int k = 0; k < 5; ++k) {
            const int ts = t - k + 2;              // output frame that read input frame t through tap k
            if (ts >= 0 && ts < Tp) acc += wj[k] * db[(size_t)ts * st_in + 3 * g + j];
        }
    }
    dx[(size_t)bb * sb_out + (size_t)t * st_out + c] = acc;
}

// LeakyReLU of the embedder head (post_activation, paule/models.py:374, :425): pre f32 -> activation type
template <typename AT>
__global__ void leaky_kernel(const float* __restrict__ pre, int64_t n, float slope, AT* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = pre[idx];
    out[idx] = from_f32<AT>(v > 0.f ? v : slope * v);
}
template <typename AT>
__global__ void leaky_bwd_kernel(const float* __restrict__ d, const float* __restrict__ pre, int64_t n, float slope, AT* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    out[idx] = from_f32<AT>(pre[idx] > 0.f ? d[idx] : slope * d[idx]);
}

__global__ void clip_copy_kernel(const float* __restrict__ x, int64_t n, int clip, float* __restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = x[idx];
    y[idx] = clip ? fminf(1.0f, fmaxf(-1.0f, v)) : v;
}

}  // namespace

void launch_mel_block(hipStream_t st, const float* x, int B, int Tp, int M, const float* w, const float* b, float* y) {
    hipLaunchKernelGGL(mel_block_kernel, dim3(blocks256((int64_t)B * Tp * M)), dim3(256), 0, st, x, B, Tp, M, w, b, y);
}

void launch_mel_block_bwd(hipStream_t st, const float* dy, int64_t sb_in, int64_t st_in, int B, int Tp, int M, const float* w, float* dx,
                          int64_t sb_out, int64_t st_out) {
    hipLaunchKernelGGL(mel_block_bwd_kernel, dim3(blocks256((int64_t)B * Tp * M)), dim3(256), 0, st, dy, sb_in, st_in, B, Tp, M, w, dx,
                       sb_out, st_out);
}

void launch_leaky(hipStream_t st, int dt, const float* pre, int64_t n, float slope, void* out) {
    if (dt == BF16) hipLaunchKernelGGL(leaky_kernel<bf16_t>, dim3(blocks256(n)), dim3(256), 0, st, pre, n, slope, static_cast<bf16_t*>(out));
    else hipLaunchKernelGGL(leaky_kernel<float>, dim3(blocks256(n)), dim3(256), 0, st, pre, n, slope, static_cast<float*>(out));
}

void launch_leaky_bwd(hipStream_t st, int dt, const float* d, const float* pre, int64_t n, float slope, void* out) {
    if (dt == BF16) hipLaunchKernelGGL(leaky_bwd_kernel<bf16_t>, dim3(blocks256(n)), dim3(256), 0, st, d, pre, n, slope, static_cast<bf16_t*>(out));
    else hipLaunchKernelGGL(leaky_bwd_kernel<float>, dim3(blocks256(n)), dim3(256), 0, st, d, pre, n, slope, static_cast<float*>(out));
}

void launch_vel_acc_pack(hipStream_t st, int dt, const float* x, int B, int Tp, int M, void* dst, int Bp, int in_p) {
    const int64_t n = (int64_t)Tp * Bp * in_p;
    if (dt == BF16)
        hipLaunchKernelGGL(vel_acc_pack_kernel<bf16_t>, dim3(blocks256(n)), dim3(256), 0, st, x, B, Tp, M, static_cast<bf16_t*>(dst), Bp, in_p);
    else
        hipLaunchKernelGGL(vel_acc_pack_kernel<float>, dim3(blocks256(n)), dim3(256), 0, st, x, B, Tp, M, static_cast<float*>(dst), Bp, in_p);
}

void launch_double_seq(hipStream_t st, const float* Y, int B, int Tp, int C, int Bp, int Cp, float* z) {
    hipLaunchKernelGGL(double_seq_kernel, dim3(blocks256((int64_t)B * 2 * Tp * C)), dim3(256), 0, st, Y, B, Tp, C, Bp, Cp, z);
}

void launch_time_conv5(hipStream_t st, const float* x, int B, int T, int C, const float* w, const float* b, const float* resid, float* y) {
    hipLaunchKernelGGL(time_conv5_kernel, dim3(blocks256((int64_t)B * T * C)), dim3(256), 0, st, x, B, T, C, w, b, resid, y);
}

void launch_resid_weight(hipStream_t st, const float* zs, const float* zl, int B, int T, int C, const float* w, const float* b, int clip,
                         float* y) {
    hipLaunchKernelGGL(resid_weight_kernel, dim3(blocks256((int64_t)B * T * C)), dim3(256), 0, st, zs, zl, B, T, C, w, b, clip, y);
}

void launch_clip_copy(hipStream_t st, const float* x, int64_t n, int clip, float* y) {
    hipLaunchKernelGGL(clip_copy_kernel, dim3(blocks256(n)), dim3(256), 0, st, x, n, clip, y);
}

}  // namespace pl
