// HBM-bound kernels of the planner: layout packing, pooling, the criterion (per-utterance
// reductions, paule/paule.py:647-662 / :705-717 / :760-773), its gradients, and the Adam update with
// the post-step projection (paule/paule.py:797, :1199-1211).
//
// The CP master copy, the Adam moments and all trajectory (smoothness) arithmetic are kept in f64:
// they are a few MB, cost nothing next to the LSTM products, and remove the f32 cancellation in
// the 1e5-weighted local-linear term (the reference is float64 end to end, paule/paule.py:124).
#include "cp_update.h"
#include "kernels.h"
#include "pl_types.h"

namespace pl {

static inline int blocks_for(int64_t n, int bs = 256) { return (int)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------
template <typename AT>
__global__ void pack_matrix_kernel(const float* __restrict__ src, int nblk, int R, int C, AT* __restrict__ dst, int Rp,
                                   int Cp, int transpose) {
    const int64_t n = (int64_t)nblk * Rp * Cp;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    int rowp, col;
    if (transpose) {
        col = (int)(idx / ((int64_t)nblk * Rp));
        rowp = (int)(idx % ((int64_t)nblk * Rp));
    } else {
        rowp = (int)(idx / Cp);
        col = (int)(idx % Cp);
    }
    const int blk = rowp / Rp, r = rowp % Rp;
    const float v = (r < R && col < C) ? src[((size_t)blk * R + r) * C + col] : 0.f;
    dst[idx] = from_f32<AT>(v);
}

void launch_pack_matrix(hipStream_t stream, int dt, const float* src, int nblk, int R, int C, void* dst, int Rp, int Cp,
                        bool transpose) {
    const int64_t n = (int64_t)nblk * Rp * Cp;
    if (dt == BF16)
        hipLaunchKernelGGL(pack_matrix_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream, src, nblk, R, C,
                           static_cast<bf16_t*>(dst), Rp, Cp, transpose ? 1 : 0);
    else
        hipLaunchKernelGGL(pack_matrix_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream, src, nblk, R, C,
                           static_cast<float*>(dst), Rp, Cp, transpose ? 1 : 0);
}

__global__ void pack_bias_kernel(const float* __restrict__ b0, const float* __restrict__ b1, int nblk, int R,
                                 float* __restrict__ dst, int Rp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nblk * Rp) return;
    const int blk = idx / Rp, r = idx % Rp;
    float v = 0.f;
    if (r < R) v = b0[blk * R + r] + (b1 ? b1[blk * R + r] : 0.f);
    dst[idx] = v;
}

void launch_pack_bias(hipStream_t stream, const float* b0, const float* b1, int nblk, int R, float* dst, int Rp) {
    hipLaunchKernelGGL(pack_bias_kernel, dim3(blocks_for(nblk * Rp)), dim3(256), 0, stream, b0, b1, nblk, R, dst, Rp);
}

// batch-major source [B][T][C] (f64 CP master or f32 mel) -> time-major activation [T][Bp][Cp]
template <typename ST, typename AT>
__global__ void pack_tm_kernel(const ST* __restrict__ src, int B, int T, int C, AT* __restrict__ dst, int Bp, int Cp) {
    const int64_t n = (int64_t)T * Bp * Cp;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % Cp);
    const int b = (int)((idx / Cp) % Bp);
    const int t = (int)(idx / ((int64_t)Cp * Bp));
    float v = 0.f;
    if (b < B && c < C) v = (float)src[((size_t)b * T + t) * C + c];
    dst[idx] = from_f32<AT>(v);
}

void launch_pack_cp(hipStream_t stream, int dt, const double* x, int B, int T, int C, void* dst, int Bp, int Cp) {
    const int64_t n = (int64_t)T * Bp * Cp;
    if (dt == BF16)
        hipLaunchKernelGGL((pack_tm_kernel<double, bf16_t>), dim3(blocks_for(n)), dim3(256), 0, stream, x, B, T, C,
                           static_cast<bf16_t*>(dst), Bp, Cp);
    else
        hipLaunchKernelGGL((pack_tm_kernel<double, float>), dim3(blocks_for(n)), dim3(256), 0, stream, x, B, T, C,
                           static_cast<float*>(dst), Bp, Cp);
}

void launch_pack_mel(hipStream_t stream, int dt, const float* mel, int B, int Tp, int C, void* dst, int Bp, int Cp) {
    const int64_t n = (int64_t)Tp * Bp * Cp;
    if (dt == BF16)
        hipLaunchKernelGGL((pack_tm_kernel<float, bf16_t>), dim3(blocks_for(n)), dim3(256), 0, stream, mel, B, Tp, C,
                           static_cast<bf16_t*>(dst), Bp, Cp);
    else
        hipLaunchKernelGGL((pack_tm_kernel<float, float>), dim3(blocks_for(n)), dim3(256), 0, stream, mel, B, Tp, C,
                           static_cast<float*>(dst), Bp, Cp);
}

// AvgPool1d(2, stride 2) over time (paule/models.py:351-354); an odd last frame is dropped.
template <typename AT>
__global__ void pool_mel_kernel(const float* __restrict__ Y, int B, int Tp, int C, int Bp, int Cp,
                                float* __restrict__ mel_bm, AT* __restrict__ mel_tm, int tp0, int n_tp) {
    const int64_t n = (int64_t)n_tp * Bp * Cp;   // pooled frames tp0 .. tp0 + n_tp - 1
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    idx += (int64_t)tp0 * Bp * Cp;
    const int c = (int)(idx % Cp);
    const int b = (int)((idx / Cp) % Bp);
    const int tp = (int)(idx / ((int64_t)Cp * Bp));
    float v = 0.f;
    if (b < B && c < C) {
        const size_t slab = (size_t)Bp * Cp;
        v = 0.5f * (Y[(size_t)(2 * tp) * slab + (size_t)b * Cp + c] + Y[(size_t)(2 * tp + 1) * slab + (size_t)b * Cp + c]);
        mel_bm[((size_t)b * Tp + tp) * C + c] = v;
    }
    mel_tm[idx] = from_f32<AT>(v);
}

void launch_pool_mel(hipStream_t stream, int dt, const float* Y, int B, int T, int C, int Bp, int Cp, float* mel_bm,
                     void* mel_tm, int tp0, int n_tp) {
    const int Tp = T / 2;
    if (n_tp < 0) { tp0 = 0; n_tp = Tp; }
    const int64_t n = (int64_t)n_tp * Bp * Cp;
    if (n <= 0) return;
    if (dt == BF16)
        hipLaunchKernelGGL(pool_mel_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream, Y, B, Tp, C, Bp, Cp, mel_bm,
                           static_cast<bf16_t*>(mel_tm), tp0, n_tp);
    else
        hipLaunchKernelGGL(pool_mel_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream, Y, B, Tp, C, Bp, Cp, mel_bm,
                           static_cast<float*>(mel_tm), tp0, n_tp);
}

// output[i, lens[i]-1, :] gather of EmbeddingModel.forward (paule/models.py:442)
template <typename AT>
__global__ void gather_last_kernel(const AT* __restrict__ h_tm, const int32_t* __restrict__ lens, int B, int Tl, int Bp,
                                   int Hp, AT* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bp * Hp) return;
    const int b = idx / Hp, k = idx % Hp;
    AT v = from_f32<AT>(0.f);
    if (b < B) {
        int t = (lens ? lens[b] : Tl) - 1;
        t = t < 0 ? 0 : (t >= Tl ? Tl - 1 : t);
        v = h_tm[((size_t)t * Bp + b) * Hp + k];
    }
    dst[idx] = v;
}

void launch_gather_last(hipStream_t stream, int dt, const void* h_tm, const int32_t* lens, int B, int Tl, int Bp, int Hp,
                        void* dst) {
    if (dt == BF16)
        hipLaunchKernelGGL(gather_last_kernel<bf16_t>, dim3(blocks_for(Bp * Hp)), dim3(256), 0, stream,
                           static_cast<const bf16_t*>(h_tm), lens, B, Tl, Bp, Hp, static_cast<bf16_t*>(dst));
    else
        hipLaunchKernelGGL(gather_last_kernel<float>, dim3(blocks_for(Bp * Hp)), dim3(256), 0, stream,
                           static_cast<const float*>(h_tm), lens, B, Tl, Bp, Hp, static_cast<float*>(dst));
}

// ---------------------------------------------------------------------------------------------
// criterion: per-utterance reductions (one workgroup per utterance, fixed order -> deterministic)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// sum of squares of the correlation d[t] = sum_k taps[k] x[t+k] over the frames t0 .. t1-1; d itself is kept (d_out [n][C]) for the
// gradient pass
template <int K>
__device__ __forceinline__ double corr_sumsq(const double* __restrict__ x, int T, int C, const double* taps, int tid,
                                             int nthreads, double* __restrict__ d_out, int t0, int t1) {
    const int n = T - K + 1;
    const int u1 = t1 < n ? t1 : n;
    double s = 0.0;
    for (int e = t0 * C + tid; e < u1 * C; e += nthreads) {
        const int t = e / C, c = e % C;
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) d += taps[k] * x[(size_t)(t + k) * C + c];
        d_out[e] = d;
        s += d * d;
    }
    return s;
}

// N sums at once (fixed order: deterministic): wave-level shuffle tree, then the waves' partials through LDS
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double (*sh)[N]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) sh[wave][k] = v[k];
    __syncthreads();
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += sh[w][k];
        v[k] = t;
    }
}
__device__ __forceinline__ void block_sum6(double (&v)[6], double (*sh)[6]) { block_sum_n<6>(v, sh); }

// One workgroup per (utterance, chunk of kLossChunkT CP frames): B = 16 x 2000 frames on 16 workgroups took 0.2 ms (cfg5).  The
// chunking is a function of T alone, so an utterance's sums are the same bits in whatever batch it is planned (the row-independence
// tests); the chunks' partial sums are added in order by loss_finalize_kernel.
constexpr int kLossChunkT = 256;
constexpr int kCorrRun = 10;   // frames per thread of the correlation window (T = 300 on 32 runs of a 1024-thread workgroup)
// up to 511 frames one workgroup per utterance (T = 300 at B = 256: two half-size workgroups per utterance measured no faster), then
// one per ~256 frames, of equal length
int loss_chunks(int T) { return T < 2 * kLossChunkT ? 1 : (T + kLossChunkT - 1) / kLossChunkT; }

__global__ __launch_bounds__(1024) void loss_reduce_kernel(LossArgs a, int nchunk) {
    __shared__ double sh[16][8];
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, tid = threadIdx.x, nt = blockDim.x;
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // mel SSE, semvec SSE, vel, jerk, ll, classifier logit sum, tube mel SSE, tube semvec SSE

    // RMSE over the utterance's T' x M mel frames (RMSELoss eps = 0, paule/util.py:570-572) and, in the same pass, the speech
    // classifier's logit: mean over time of Linear(60 -> 1) (paule/models.py:899-908); this chunk's share of the elements
    const int nm = a.Tp * a.M;
    const int e0 = (int)((long long)nm * ch / nchunk), e1 = (int)((long long)nm * (ch + 1) / nchunk);
    const float* mel = a.mel + (size_t)b * nm;
    const float* tgt = a.target_mel + (size_t)b * nm;
    for (int e = e0 + tid; e < e1; e += nt) {
        const double m = (double)mel[e], d = m - (double)tgt[e];
        s[0] += d * d;
        if (a.cls_wb) s[5] += (double)a.cls_wb[e % a.M] * m;
    }
    if (a.sem && ch == 0)
        for (int e = tid; e < a.S; e += nt) {
            const double d = (double)a.sem[(size_t)b * a.Sp + e] - (double)a.target_sem[(size_t)b * a.S + e];
            s[1] += d * d;
        }
    if (a.mel2) {
        const float* m2 = a.mel2 + (size_t)b * nm;
        for (int e = e0 + tid; e < e1; e += nt) {
            const double d = (double)m2[e] - (double)tgt[e];
            s[6] += d * d;
        }
    }
    if (a.sem2 && ch == 0)
        for (int e = tid; e < a.S; e += nt) {
            const double d = (double)a.sem2[(size_t)b * a.Sp + e] - (double)a.target_sem[(size_t)b * a.S + e];
            s[7] += d * d;
        }
    const double* x = a.x + (size_t)b * a.T * a.C;
    const size_t per = (size_t)a.T * a.C;
    double* dws = a.dwork + (size_t)b * 3 * per;     // [vel | jerk | ll] correlations of this utterance
    const int tc = (a.T + nchunk - 1) / nchunk;   // frames per chunk
    const int t0 = ch * tc, t1 = ch == nchunk - 1 ? a.T : (ch + 1) * tc;
    // The three correlations of this workgroup's frames from ONE window of CP rows per thread (round 5): a thread owns channel tid & 31 of a run of
    // kCorrRun frames, loads its kCorrRun + 12 rows once (all in flight; channel-fastest over the lanes: whole 240-byte rows) and forms every
    // d[t] = sum_k taps[k] x[t + k] in the taps' order -- the same bits per correlation as one load per tap (21 loads per frame and channel from L2:
    // 42 us at cfg3), only the order in which their squares are added differs (fixed: a function of T alone, so the row-independence rule holds).
    if (a.C <= 32) {
        const int c = tid & 31, nrun = nt >> 5;
        for (int ta = t0 + (tid >> 5) * kCorrRun; ta < t1; ta += nrun * kCorrRun) {
            if (c >= a.C) continue;
            double xw[kCorrRun + 12];
#pragma unroll
            for (int k = 0; k < kCorrRun + 12; ++k) xw[k] = ta + k < a.T ? x[(size_t)(ta + k) * a.C + c] : 0.0;
#pragma unroll
            for (int i = 0; i < kCorrRun; ++i) {
                const int t = ta + i;
                double dv = 0.0, dj = 0.0, dl = 0.0;
#pragma unroll
                for (int k = 0; k < 5; ++k) dv += kVelTaps[k] * xw[i + k];
#pragma unroll
                for (int k = 0; k < 13; ++k) dj += kJerkTaps[k] * xw[i + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) dl += kLlTaps[k] * xw[i + k];
                const size_t e = (size_t)t * a.C + c;
                if (t < t1 && t < a.T - 4) { dws[e] = dv; s[2] += dv * dv; }
                if (t < t1 && t < a.T - 12) { dws[per + e] = dj; s[3] += dj * dj; }
                if (t < t1 && t < a.T - 2) { dws[2 * per + e] = dl; s[4] += dl * dl; }
            }
        }
    } else {
        s[2] = corr_sumsq<5>(x, a.T, a.C, kVelTaps, tid, nt, dws, t0, t1);
        s[3] = corr_sumsq<13>(x, a.T, a.C, kJerkTaps, tid, nt, dws + per, t0, t1);
        s[4] = corr_sumsq<3>(x, a.T, a.C, kLlTaps, tid, nt, dws + 2 * per, t0, t1);
    }
    block_sum_n<8>(s, sh);
    if (tid == 0) {
        double* pr = a.part + ((size_t)b * nchunk + ch) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[k] = s[k];
    }
}

void launch_loss_reduce(hipStream_t stream, const LossArgs& a) {
    const int nchunk = loss_chunks(a.T);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(a.B * nchunk), dim3(1024), 0, stream, a, nchunk);   // 16 waves per workgroup
}

// weighted sub-losses, as the reference logs them (paule/paule.py:654-662, :942-945)
__global__ void loss_finalize_kernel(LossArgs a, int nchunk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    double* sc = a.scal + (size_t)b * 8;
    {   // the chunks' partial sums, in order -> the utterance's scalars (read by the gradient kernels of this iteration)
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int ch = 0; ch < nchunk; ++ch)
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += a.part[((size_t)b * nchunk + ch) * 8 + k];
        const int nm = a.Tp * a.M;
        sc[6] = a.mel2 ? sqrt(s[6] / nm) : 0.0;
        sc[7] = a.sem2 ? sqrt(s[7] / a.S) : 0.0;
        sc[0] = sqrt(s[0] / nm);
        sc[1] = a.sem ? sqrt(s[1] / a.S) : 0.0;
        sc[2] = s[2] / ((double)(a.T - 4) * a.C);
        sc[3] = s[3] / ((double)(a.T - 12) * a.C);
        sc[4] = s[4] / ((double)(a.T - 2) * a.C);
        sc[5] = a.cls_wb ? s[5] / a.Tp + (double)a.cls_wb[a.M] : 0.0;
    }
    const double mel = a.w_mel * sc[0], sem = a.sem ? a.w_sem * sc[1] : 0.0;
    const double vel = a.w_vel * sc[2], jerk = a.w_jerk * sc[3], ll = a.w_ll * sc[4];
    // BCEWithLogits(z, 0) = softplus(z) (paule/paule.py:610-612), numerically stable form
    const double z = sc[5];
    const double cls = a.cls_wb ? a.w_cls * (fmax(z, 0.0) + log1p(exp(-fabs(z)))) : 0.0;
    // somatosensory feedback: both tube terms enter whenever they are evaluated (paule/paule.py:642, :755)
    const double tmel = a.mel2 ? a.w_mel * sc[6] : 0.0, tsem = a.sem2 ? a.w_sem * sc[7] : 0.0;
    double total = vel + jerk + ll + cls + tmel + tsem;
    if (a.use_mel) total += mel;
    if (a.use_sem) total += sem;
    float* row = a.loss_rows + ((size_t)(*a.iter_slot) * a.B + b) * 8;
    row[0] = (float)total;
    row[1] = (float)mel;
    row[2] = (float)sem;
    row[3] = (float)vel;
    row[4] = (float)jerk;
    row[5] = (float)ll;
    row[6] = (float)(a.mel2 ? tmel : cls);   // speech classifier and somatosensory feedback exclude each other (paule/paule.py:117)
    row[7] = (float)tsem;
}

void launch_loss_finalize(hipStream_t stream, const LossArgs& a) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(blocks_for(a.B, 64)), dim3(64), 0, stream, a, loss_chunks(a.T));
}

// d(w_sem * rmse_sem)/d sem = w_sem (sem - tgt) / (S * rmse).  rmse == 0 (exact match) would be 0/0 in the
// reference (RMSELoss eps = 0); the gradient is defined as 0 there instead of NaN.
template <typename AT>
__global__ void dsem_kernel(LossArgs a, AT* __restrict__ dsem, int tube) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.Bp * a.Sp) return;
    const int b = idx / a.Sp, s = idx % a.Sp;
    const float* sem = tube ? a.sem2 : a.sem;
    float v = 0.f;
    if (b < a.B && s < a.S) {
        const double rm = a.scal[(size_t)b * 8 + (tube ? 7 : 1)];
        if (rm > 0.0)
            v = (float)((double)a.w_sem * ((double)sem[(size_t)b * a.Sp + s] - (double)a.target_sem[(size_t)b * a.S + s]) /
                        ((double)a.S * rm));
    }
    dsem[idx] = from_f32<AT>(v);
}

void launch_dsem(hipStream_t stream, int dt, const LossArgs& a, void* dsem, bool tube) {
    const int n = a.Bp * a.Sp;
    if (dt == BF16)
        hipLaunchKernelGGL(dsem_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream, a, static_cast<bf16_t*>(dsem), tube ? 1 : 0);
    else
        hipLaunchKernelGGL(dsem_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream, a, static_cast<float*>(dsem), tube ? 1 : 0);
}

// dL/dY (pre-pool linear output): each pooled frame's gradient goes half to each of its two frames.
template <typename AT>
__global__ void dy_kernel(LossArgs a, const float* __restrict__ dmel_e, AT* __restrict__ dY, int tube, int t0, int n_t) {
    // one thread per POOLED frame element (tp, b, m): both frames of the pair get the same value (round 5: one thread per frame element
    // computed everything twice -- an f64 division and three 64-bit index divisions each: 27 us for 20 MB at cfg3)
    const int tp0 = t0 >> 1, tp1 = (t0 + n_t + 1) >> 1;   // pooled frames touched by frames t0 .. t0 + n_t - 1
    const unsigned n = (unsigned)(tp1 - tp0) * (unsigned)a.Bp * (unsigned)a.Mp;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const unsigned m = idx % (unsigned)a.Mp, r = idx / (unsigned)a.Mp;
    const unsigned b = r % (unsigned)a.Bp;
    const int tp = tp0 + (int)(r / (unsigned)a.Bp);
    float v = 0.f;
    if ((int)b < a.B && (int)m < a.M && tp < a.Tp) {
        double g = 0.0;
        if (tube) {
            const double rm = a.scal[(size_t)b * 8 + 6];
            const size_t e = ((size_t)b * a.Tp + tp) * a.M + m;
            if (rm > 0.0) g = (double)a.w_mel * ((double)a.mel2[e] - (double)a.target_mel[e]) / ((double)a.Tp * a.M * rm);
        } else if (a.use_mel) {
            const double rm = a.scal[(size_t)b * 8 + 0];
            const size_t e = ((size_t)b * a.Tp + tp) * a.M + m;
            if (rm > 0.0) g = (double)a.w_mel * ((double)a.mel[e] - (double)a.target_mel[e]) / ((double)a.Tp * a.M * rm);
        }
        if (dmel_e) g += (double)dmel_e[((size_t)tp * a.Bp + b) * a.Mp + m];
        if (a.cls_wb && !tube) {   // d(w_cls softplus(z))/d mel = w_cls sigmoid(z) w[m] / T'
            const double z = a.scal[(size_t)b * 8 + 5];
            g += (double)a.w_cls / (1.0 + exp(-z)) * (double)a.cls_wb[m] / (double)a.Tp;
        }
        v = (float)(0.5 * g);
    }
    const AT o = from_f32<AT>(v);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = 2 * tp + h;
        if (t >= t0 && t < t0 + n_t) dY[((size_t)t * a.Bp + b) * a.Mp + m] = o;
    }
}

void launch_dy(hipStream_t stream, int dt, const LossArgs& a, const float* dmel_e, void* dY, bool tube, int t0, int n_t) {
    if (n_t < 0) { t0 = 0; n_t = a.T; }
    if (n_t <= 0) return;
    const int64_t n = (int64_t)(((t0 + n_t + 1) >> 1) - (t0 >> 1)) * a.Bp * a.Mp;   // (< 2^31: 2000 frames x 2048 rows x 64 = 1.3e8)
    if (n <= 0) return;
    if (dt == BF16)
        hipLaunchKernelGGL(dy_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream, a, dmel_e, static_cast<bf16_t*>(dY), tube ? 1 : 0, t0, n_t);
    else
        hipLaunchKernelGGL(dy_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream, a, dmel_e, static_cast<float*>(dY), tube ? 1 : 0, t0, n_t);
}

// ---------------------------------------------------------------------------------------------
// gradient of the smoothness terms + Adam
// ---------------------------------------------------------------------------------------------
// (the per-element arithmetic -- smoothness gradients, total gradient, Adam, projection -- is in cp_update.h)
constexpr int kGradRun = 4;   // frames per thread

// Total gradient (model gradient(s) + the three smoothness terms), torch.optim.Adam (no amsgrad / weight decay) and paule/paule.py:1201-1211 on the
// CP tensor, one thread per (utterance, channel, run of kGradRun frames).  One kernel since round 5 (two before: the total gradient went out to
// memory and came back, every workgroup of the second waited for its first thread's two pow() before it loaded anything, and a thread carried one
// element's dependent f64 chain -- 25 + 31 us at cfg3): the loads of a run go out first, the step's scalars are computed under them, and four
// independent chains of divisions and square roots fill each other's latencies.  Same operations per element in the same order (cp_update.h).
__global__ __launch_bounds__(256) void cp_update_kernel(AdamArgs a) {
    __shared__ double step_sc[2];   // lr / (1 - beta1^k), sqrt(1 - beta2^k)
    const int nrun = (a.T + kGradRun - 1) / kGradRun;
    const int64_t n = (int64_t)a.B * nrun * a.C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = idx < n;
    const int64_t id = on ? idx : 0;
    const int c = (int)(id % a.C);
    const int t0 = (int)((id / a.C) % nrun) * kGradRun;
    const int b = (int)(id / ((int64_t)a.C * nrun));
    const int b_first = (int)(((int64_t)blockIdx.x * blockDim.x) / ((int64_t)a.C * nrun));   // the workgroup's first utterance: the buffer's base
    double gv[kGradRun] = {}, gj[kGradRun] = {}, gl[kGradRun] = {};
    smooth_grads<kGradRun>(a, dwork_rsrc(a, b_first < a.B ? b_first : a.B - 1), b_first < a.B ? b_first : a.B - 1, b, c, t0, gv, gj, gl);
    double g[kGradRun], m[kGradRun], v[kGradRun], x[kGradRun];
#pragma unroll
    for (int j = 0; j < kGradRun; ++j) {
        const int t = t0 + j < a.T ? t0 + j : a.T - 1;   // (a frame beyond T: loaded from the last one, never stored)
        const size_t e = ((size_t)t * a.Bp + b) * a.Cp + c;
        const size_t k = ((size_t)b * a.T + t) * a.C + c;
        g[j] = total_grad_of((double)a.dX[e], a.dX2 != nullptr, a.dX2 ? (double)a.dX2[e] : 0.0, gv[j], gj[j], gl[j]);
        m[j] = a.m[k]; v[j] = a.v[k]; x[j] = a.x[k];
    }
    if (threadIdx.x == 0) adam_step_scalars(a, step_sc[0], step_sc[1]);
    __syncthreads();
    const double lr_bc1 = step_sc[0], sqrt_bc2 = step_sc[1];
#pragma unroll
    for (int j = 0; j < kGradRun; ++j) adam_value(a, b, t0 + j < a.T ? t0 + j : a.T - 1, c, g[j], lr_bc1, sqrt_bc2, m[j], v[j], x[j]);
    if (!on) return;
#pragma unroll
    for (int j = 0; j < kGradRun; ++j) {
        const int t = t0 + j;
        if (t < a.T) {
            const size_t k = ((size_t)b * a.T + t) * a.C + c;
            a.grad[k] = g[j]; a.m[k] = m[j]; a.v[k] = v[j]; a.x[k] = x[j];
        }
    }
}

__global__ void bump_counters_kernel(int* step_count, int* iter_slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *step_count += 1;
        *iter_slot += 1;
    }
}

void launch_cp_update(hipStream_t stream, const AdamArgs& a) {
    const int64_t n = (int64_t)a.B * ((a.T + kGradRun - 1) / kGradRun) * a.C;
    hipLaunchKernelGGL(cp_update_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(bump_counters_kernel, dim3(1), dim3(64), 0, stream, a.step_count, a.iter_slot);
}

// ---------------------------------------------------------------------------------------------
// conversions
// ---------------------------------------------------------------------------------------------
__global__ void f64_to_f32_kernel(const double* __restrict__ s, float* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}
__global__ void f32_to_f64_kernel(const float* __restrict__ s, double* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (double)s[i];
}
// time-major padded f32 [T][Bp][Cp] -> batch-major [B][T][C]
__global__ void tm_to_bm_kernel(const float* __restrict__ s, int B, int T, int C, int Bp, int Cp, float* __restrict__ d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * T * C) return;
    const int c = (int)(idx % C), t = (int)((idx / C) % T), b = (int)(idx / ((int64_t)C * T));
    d[idx] = s[((size_t)t * Bp + b) * Cp + c];
}

template <typename AT>
__global__ void add2_act_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, AT* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = from_f32<AT>(a[idx] + b[idx]);
}

template <typename AT>
__global__ void act_to_f32_kernel(const AT* __restrict__ s, float* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = to_f32<AT>(s[i]);
}
__global__ void unpad_rows_kernel(const float* __restrict__ s, int B, int S, int Sp, float* __restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * S) d[i] = s[(size_t)(i / S) * Sp + (i % S)];
}

void launch_f64_to_f32(hipStream_t stream, const double* src, float* dst, int64_t n) {
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, src, dst, n);
}
void launch_f32_to_f64(hipStream_t stream, const float* src, double* dst, int64_t n) {
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, src, dst, n);
}
void launch_act_to_f32(hipStream_t stream, int dt, const void* src, float* dst, int64_t n) {
    if (dt == BF16)
        hipLaunchKernelGGL(act_to_f32_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream,
                           static_cast<const bf16_t*>(src), dst, n);
    else
        hipLaunchKernelGGL(act_to_f32_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream,
                           static_cast<const float*>(src), dst, n);
}
void launch_tm_to_bm(hipStream_t stream, const float* src, int B, int T, int C, int Bp, int Cp, float* dst) {
    hipLaunchKernelGGL(tm_to_bm_kernel, dim3(blocks_for((int64_t)B * T * C)), dim3(256), 0, stream, src, B, T, C, Bp, Cp, dst);
}
void launch_add2_act(hipStream_t stream, int dt, const float* a, const float* b, int64_t n, void* out) {
    if (dt == BF16)
        hipLaunchKernelGGL(add2_act_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, stream, a, b, n, static_cast<bf16_t*>(out));
    else
        hipLaunchKernelGGL(add2_act_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, stream, a, b, n, static_cast<float*>(out));
}
void launch_unpad_rows(hipStream_t stream, const float* src, int B, int S, int Sp, float* dst) {
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(blocks_for(B * S)), dim3(256), 0, stream, src, B, S, Sp, dst);
}

}  // namespace pl
