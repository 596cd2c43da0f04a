// Kernels of the continued-learning step of the predictive model (paule/paule.py:1353-1379: forward, RMSE against the
// produced mel, backward INCLUDING the weight gradients, Adam on the parameters).  The forward pass and the
// backward-data recurrences are the planner's own (persistent sweeps / step kernels); this file adds what planning
// never needs: the weight-gradient products, bias gradients, the batch-wide RMSE, and the parameter update.
//
//   dW[m][n] = sum over (t, b) of A[(tA0 + t) * Bp + b][m] * B[(tB0 + t) * Bp + b][n]      ("TN": both operands k-major)
//
// A = dA stash (gate rows), B = h_{t-1} stash / layer input.  Only the first nb (multiple of 16) batch rows of every
// time slab carry training samples, so a K tile is 16 consecutive rows of one slab.  The products run on the exact
// f32 MFMA (v_mfma_f32_16x16x4_f32) whatever the activation type: the operand of that instruction is one element per
// lane per k, so a k-major LDS image feeds it without a transpose, and weight gradients summed over thousands of
// (t, b) terms want f32 products anyway.  bf16 activations are widened while they are staged into LDS.
#include <cstdlib>

#include "kernels.h"
#include "pl_types.h"

namespace pl {

namespace {

template <typename AT> __device__ __forceinline__ void widen8(const AT* p, float (&f)[8]);
template <> __device__ __forceinline__ void widen8<bf16_t>(const bf16_t* p, float (&f)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <> __device__ __forceinline__ void widen8<float>(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// Block tile 128 (m) x BN (n), K tile 16; 4 waves as 2 x 2, each 64 x BN/2 = 4 x (BN/32) MFMA tiles.
// LDS image [16 k][128 + 16] f32: the +16 makes rows k and k+1 land on disjoint bank halves for ds_read_b32.
template <typename AT, int BN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const AT* __restrict__ A, int lda, const AT* __restrict__ Bm, int ldb,
                                                      float* __restrict__ C, int ldc, int M, int N, int Bp, int nb, int Tk,
                                                      int tA0, int tB0) {
    constexpr int BM = 128, BK = 16, SA = BM + 16, SB = BN + 16;
    constexpr int TN = BN / 32;                  // MFMA tiles per wave along n
    __shared__ float sA[2][BK * SA];
    __shared__ float sB[2][BK * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tiles_per_t = nb / 16, tiles_all = Tk * tiles_per_t;
    // split-K: blockIdx.z takes a contiguous range of K tiles and writes its own partial C (summed by sum_splits_kernel)
    const int kt0 = (int)((long long)tiles_all * blockIdx.z / gridDim.z), kt1 = (int)((long long)tiles_all * (blockIdx.z + 1) / gridDim.z);
    const int n_tiles = kt1 - kt0;
    C += (size_t)blockIdx.z * M * ldc;

    // staging: thread -> (k row, 8 consecutive columns)
    const int ak = tid >> 4, ac = (tid & 15) * 8;            // A tile: 16 rows x 128 columns
    const int bk = (tid * 8) / BN, bc = (tid * 8) % BN;      // B tile: 16 rows x BN columns (BN = 64: threads 0..127)
    const bool b_on = tid * 8 < BK * BN;
    float ra[8], rb[8];
    auto load = [&](int kt) {
        const int t = (kt0 + kt) / tiles_per_t, b0 = ((kt0 + kt) % tiles_per_t) * 16;
        const size_t rowA = (size_t)(tA0 + t) * Bp + b0 + ak, rowB = (size_t)(tB0 + t) * Bp + b0 + bk;
        if (m0 + ac < M) widen8<AT>(A + rowA * lda + m0 + ac, ra);      // M, N are multiples of 8: a piece is whole or absent
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = 0.f;
        if (b_on) {
            if (n0 + bc < N) widen8<AT>(Bm + rowB * ldb + n0 + bc, rb);
            else
#pragma unroll
                for (int i = 0; i < 8; ++i) rb[i] = 0.f;
        }
    };
    auto stage = [&](int buf) {
        float* pa = sA[buf] + ak * SA + ac;
        *reinterpret_cast<float4*>(pa) = make_float4(ra[0], ra[1], ra[2], ra[3]);
        *reinterpret_cast<float4*>(pa + 4) = make_float4(ra[4], ra[5], ra[6], ra[7]);
        if (b_on) {
            float* pb = sB[buf] + bk * SB + bc;
            *reinterpret_cast<float4*>(pb) = make_float4(rb[0], rb[1], rb[2], rb[3]);
            *reinterpret_cast<float4*>(pb + 4) = make_float4(rb[4], rb[5], rb[6], rb[7]);
        }
    };

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (n_tiles > 0) {
        load(0);
        stage(0);
    }
    __syncthreads();
    const int lr = lane & 15, kq = lane >> 4;
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_tiles) load(kt + 1);
        const float* pa = sA[buf] + kq * SA + wm * 64 + lr;
        const float* pb = sB[buf] + kq * SB + wn * (BN / 2) + lr;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float fa[4], fb[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = pa[ks * 4 * SA + i * 16];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = pb[ks * 4 * SB + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < n_tiles) stage(buf ^ 1);   // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // C layout: column n = lane & 15, row m = 4 (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + kq * 4 + r;
                if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][r];
            }
        }
}

// column sums over the active rows: out[c] = sum_(t, b < nb) A[(t0 + t) * Bp + b][c].  Two fixed-order stages
// (deterministic): blockIdx.y sums a contiguous range of time slabs into part[y][c] (f64), then one thread per column
// adds the parts.
template <typename AT>
__global__ void colsum_part_kernel(const AT* __restrict__ A, int lda, int ncols, int Bp, int nb, int Tk, int t0, double* __restrict__ part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const int ta = (int)((long long)Tk * blockIdx.y / gridDim.y), tb = (int)((long long)Tk * (blockIdx.y + 1) / gridDim.y);
    double s = 0.0;
    for (int t = ta; t < tb; ++t) {
        const AT* row = A + ((size_t)(t0 + t) * Bp) * lda + c;
        for (int b = 0; b < nb; ++b) s += (double)to_f32<AT>(row[(size_t)b * lda]);
    }
    part[(size_t)blockIdx.y * ncols + c] = s;
}

__global__ void colsum_final_kernel(const double* __restrict__ part, int S, int ncols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    double s = 0.0;
    for (int i = 0; i < S; ++i) s += part[(size_t)i * ncols + c];
    out[c] = (float)s;
}

// C[i] = sum_s P[s][i], fixed order
__global__ void sum_splits_kernel(const float* __restrict__ P, int S, int64_t n, float* __restrict__ C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = P[i];
    for (int k = 1; k < S; ++k) s += P[(size_t)k * n + i];
    C[i] = s;
}

// batch-wide RMSE of the predicted mel against the produced mel (RMSELoss(eps=0), paule/util.py:564-572): one workgroup,
// fixed-order f64 tree -> deterministic.  out[0] = rmse, out[1] = sum of squares
__global__ __launch_bounds__(1024) void train_rmse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                          double* __restrict__ out, float* __restrict__ loss_out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double d = (double)pred[i] - (double)target[i];
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double rmse = sqrt(red[0] / (double)n);
        out[0] = rmse;
        out[1] = red[0];
        if (loss_out) *loss_out = (float)rmse;
    }
}

// dL/dY of rmse(avgpool2(Y), target): each of the two pooled frames gets half of (mel - target) / (N rmse)
template <typename AT>
__global__ void train_dy_kernel(const float* __restrict__ pred, const float* __restrict__ target, const double* __restrict__ scal,
                                int n_rows, int T, int Tp, int M, int Bp, int Mp, AT* __restrict__ dY, int pooled) {
    const int64_t n = (int64_t)T * Bp * Mp;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int m = (int)(idx % Mp);
    const int b = (int)((idx / Mp) % Bp);
    const int t = (int)(idx / ((int64_t)Mp * Bp));
    const int tp = pooled ? t >> 1 : t;     // pooled = 0: the model does not halve its sequence (Tp = T; cp -> tube model)
    float v = 0.f;
    const double rmse = scal[0];
    if (b < n_rows && m < M && tp < Tp && rmse > 0.0) {
        const size_t e = ((size_t)b * Tp + tp) * M + m;
        v = (float)((pooled ? 0.5 : 1.0) * ((double)pred[e] - (double)target[e]) / ((double)n_rows * Tp * M * rmse));
    }
    dY[idx] = from_f32<AT>(v);
}

// torch.optim.Adam on one parameter matrix (f64 master / moments, paule/paule.py:287) + refresh of the packed compute copies.
// grad is in the padded compute layout [nblk * Rp][Cp]; master / moments in torch layout [nblk * R][C].
template <typename AT>
__global__ void adam_matrix_kernel(const float* __restrict__ grad, int nblk, int R, int C, int Rp, int Cp, double* __restrict__ x,
                                   double* __restrict__ am, double* __restrict__ av, AT* __restrict__ W, AT* __restrict__ WT,
                                   double lr, double b1, double b2, double eps, double bc1, double bc2) {
    const int64_t n = (int64_t)nblk * R * C;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const int r = (int)((idx / C) % R);
    const int blk = (int)(idx / ((int64_t)C * R));
    const size_t prow = (size_t)blk * Rp + r;
    const double g = (double)grad[prow * Cp + c];
    const double m = b1 * am[idx] + (1.0 - b1) * g;
    const double v = b2 * av[idx] + (1.0 - b2) * g * g;
    am[idx] = m;
    av[idx] = v;
    const double xn = x[idx] - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps);
    x[idx] = xn;
    const AT w = from_f32<AT>((float)xn);
    W[prow * Cp + c] = w;
    if (WT) WT[(size_t)c * ((size_t)nblk * Rp) + prow] = w;
}

// bias vectors: b0 (and b1: bias_ih / bias_hh share one gradient) updated separately, packed = b0 + b1
__global__ void adam_bias_kernel(const float* __restrict__ grad, int nblk, int R, int Rp, double* __restrict__ x0, double* __restrict__ am0,
                                 double* __restrict__ av0, double* __restrict__ x1, double* __restrict__ am1, double* __restrict__ av1,
                                 float* __restrict__ packed, double lr, double b1, double b2, double eps, double bc1, double bc2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nblk * R) return;
    const int blk = idx / R, r = idx % R;
    const double g = (double)grad[blk * Rp + r];
    auto upd = [&](double* x, double* am, double* av) -> double {
        const double m = b1 * am[idx] + (1.0 - b1) * g;
        const double v = b2 * av[idx] + (1.0 - b2) * g * g;
        am[idx] = m;
        av[idx] = v;
        const double xn = x[idx] - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps);
        x[idx] = xn;
        return xn;
    };
    double s = upd(x0, am0, av0);
    if (x1) s += upd(x1, am1, av1);
    packed[blk * Rp + r] = (float)s;
}


// bf16 activations: the same product on the bf16 MFMA (v_mfma_f32_16x16x32_bf16, 16x the f32 instruction's rate).  Its operands
// want 8 consecutive k per lane while both matrices are k-major -- gfx950's transposed LDS read (ds_read_b64_tr_b16: a 16-lane
// group reads a block of 4 rows x 16 columns and every lane receives one COLUMN) delivers exactly that from a plain [k][m] image.
// K tile = 32 rows (two 16-row pieces of the (t, b) enumeration; an odd last piece is zero-filled).  Lane group g takes image
// rows 4g .. 4g+3 as its k elements 0..3 and rows 16+4g .. 16+4g+3 as elements 4..7 -- the same (arbitrary) assignment of rows
// to k slots for both operands, which is all a sum over k needs -- so that the two groups of a 32-lane half read 8 consecutive
// rows: with a row stride of 256 + 32 bytes they cover all 64 banks exactly once.  bf16 x bf16 products are exact in f32 and
// accumulate in f32, as before; only the summation order over k differs from the f32-MFMA kernel.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
template <int BN>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ Bm, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int N, int Bp, int nb, int Tk,
                                                           int tA0, int tB0) {
    constexpr int BM = 128, BK = 32;
    constexpr int SA = BM * 2 + 32, SB = BN * 2 + 32;   // bytes per image row
    constexpr int TN = BN / 32;                         // MFMA tiles per wave along n
    constexpr int CB = BN / 8;                          // 16-byte chunks per B row
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][BK * SA];
    __shared__ __attribute__((aligned(16))) unsigned char sB[2][BK * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tiles_per_t = nb / 16, tiles_all = Tk * tiles_per_t;
    const int kt0 = (int)((long long)tiles_all * blockIdx.z / gridDim.z), kt1 = (int)((long long)tiles_all * (blockIdx.z + 1) / gridDim.z);
    const int n_super = (kt1 - kt0 + 1) / 2;            // K tiles of 32 rows
    C += (size_t)blockIdx.z * M * ldc;

    // staging: A tile 32 rows x 16 chunks = 2 per thread (rows tid >> 4 and 16 + (tid >> 4)); B tile 32 rows x CB chunks
    const int ar = tid >> 4, ac = tid & 15;
    constexpr int BPT = (BK * CB + 255) / 256;          // B chunks per thread (BN = 128: 2, BN = 64: 1)
    uint4 ra[2], rb[BPT];
    auto load = [&](int st) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kt = kt0 + 2 * st + half;
            const int t = kt / tiles_per_t, b0 = (kt % tiles_per_t) * 16;
            const size_t rowA = (size_t)(tA0 + t) * Bp + b0 + ar;
            ra[half] = (kt < kt1 && m0 + ac * 8 < M) ? *reinterpret_cast<const uint4*>(A + rowA * lda + m0 + ac * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int e = tid + 256 * i, row = e / CB, ch = e % CB;
            const int kt = kt0 + 2 * st + (row >> 4);
            const int t = kt / tiles_per_t, b0 = (kt % tiles_per_t) * 16;
            const size_t rowB = (size_t)(tB0 + t) * Bp + b0 + (row & 15);
            rb[i] = (kt < kt1 && n0 + ch * 8 < N) ? *reinterpret_cast<const uint4*>(Bm + rowB * ldb + n0 + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<uint4*>(sA[buf] + ar * SA + ac * 16) = ra[0];
        *reinterpret_cast<uint4*>(sA[buf] + (16 + ar) * SA + ac * 16) = ra[1];
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int e = tid + 256 * i, row = e / CB, ch = e % CB;
            *reinterpret_cast<uint4*>(sB[buf] + row * SB + ch * 16) = rb[i];
        }
    };

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (n_super > 0) {
        load(0);
        stage(0);
    }
    __syncthreads();
    // transposed-read address of this lane: group g = lane >> 4, in-group lane 4 q + p -> image row 4 g + q, columns 4 p .. 4 p + 3
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int offA = (4 * g + q) * SA + (wm * 64 + 4 * pp) * 2;
    const int offB = (4 * g + q) * SB + (wn * (BN / 2) + 4 * pp) * 2;
    for (int st = 0; st < n_super; ++st) {
        const int buf = st & 1;
        if (st + 1 < n_super) load(st + 1);
        bf16x8 fa[4], fb[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(sA[buf] + offA + i * 32));
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(sA[buf] + offA + 16 * SA + i * 32));
            fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(sB[buf] + offB + j * 32));
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(sB[buf] + offB + 16 * SB + j * 32));
            fb[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (st + 1 < n_super) stage(buf ^ 1);   // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // C layout: column n = lane & 15, row m = 4 (lane >> 4) + reg
    const int lr = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + kq * 4 + r;
                if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][r];
            }
        }
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// scratch: at least train_scratch_bytes(M, N) for split-K partials (ldc == N is required when the split is used)
size_t train_scratch_bytes(int M, int N) { return (size_t)8 * M * N * sizeof(float); }

void launch_gemm_tn(hipStream_t stream, int dt, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N,
                    int Bp, int nb, int Tk, int tA0, int tB0, float* scratch, size_t scratch_bytes, int n_cu, bool tn_bf16) {
    if (M <= 0 || N <= 0) return;
    const bool narrow = N <= 64;
    const int gx = (N + (narrow ? 63 : 127)) / (narrow ? 64 : 128), gy = (M + 127) / 128;
    // split-K so that ~2 blocks per CU are in flight: a block's K loop is serial (Tk * nb / 16 tiles of 16 rows)
    const int tiles = Tk * (nb / 16);
    int S = (2 * n_cu) / (gx * gy);
    if (S > tiles / 8) S = tiles / 8;                                  // keep >= 8 K tiles per block
    if (S > 64) S = 64;
    while (S > 1 && (size_t)S * M * N * sizeof(float) > scratch_bytes) --S;
    if (S < 2 || !scratch || ldc != N) S = 1;
    float* dst = S > 1 ? scratch : C;
    const dim3 grid(gx, gy, S);
#define PL_TN(AT_)                                                                                                              \
    do {                                                                                                                        \
        if (narrow)                                                                                                             \
            hipLaunchKernelGGL((gemm_tn_kernel<AT_, 64>), grid, dim3(256), 0, stream, static_cast<const AT_*>(A), lda,          \
                               static_cast<const AT_*>(B), ldb, dst, ldc, M, N, Bp, nb, Tk, tA0, tB0);                          \
        else                                                                                                                    \
            hipLaunchKernelGGL((gemm_tn_kernel<AT_, 128>), grid, dim3(256), 0, stream, static_cast<const AT_*>(A), lda,         \
                               static_cast<const AT_*>(B), ldb, dst, ldc, M, N, Bp, nb, Tk, tA0, tB0);                          \
    } while (0)
    // tn_bf16 false (PAULE_HIP_TN_BF16=0, read once per handle): the f32-MFMA form for bf16 operands too (A/B, tests)
    if (dt == BF16 && tn_bf16) {
        if (narrow)
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<64>), grid, dim3(256), 0, stream, static_cast<const bf16_t*>(A), lda,
                               static_cast<const bf16_t*>(B), ldb, dst, ldc, M, N, Bp, nb, Tk, tA0, tB0);
        else
            hipLaunchKernelGGL((gemm_tn_bf16_kernel<128>), grid, dim3(256), 0, stream, static_cast<const bf16_t*>(A), lda,
                               static_cast<const bf16_t*>(B), ldb, dst, ldc, M, N, Bp, nb, Tk, tA0, tB0);
    } else if (dt == BF16) PL_TN(bf16_t);
    else PL_TN(float);
#undef PL_TN
    if (S > 1) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(sum_splits_kernel, dim3(blocks256(n)), dim3(256), 0, stream, scratch, S, n, C);
    }
}

// part: at least 64 * ncols doubles
void launch_colsum(hipStream_t stream, int dt, const void* A, int lda, int ncols, int Bp, int nb, int Tk, int t0, float* out, double* part) {
    const int S = Tk < 64 ? Tk : 64;
    const dim3 grid((ncols + 63) / 64, S);
    if (dt == BF16)
        hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, grid, dim3(64), 0, stream, static_cast<const bf16_t*>(A), lda, ncols, Bp, nb, Tk, t0,
                           part);
    else
        hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(64), 0, stream, static_cast<const float*>(A), lda, ncols, Bp, nb, Tk, t0,
                           part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((ncols + 63) / 64), dim3(64), 0, stream, part, S, ncols, out);
}

void launch_train_rmse(hipStream_t stream, const float* pred, const float* target, int64_t n, double* scal, float* loss_out) {
    hipLaunchKernelGGL(train_rmse_kernel, dim3(1), dim3(1024), 0, stream, pred, target, n, scal, loss_out);
}

void launch_train_dy(hipStream_t stream, int dt, const float* pred, const float* target, const double* scal, int n_rows, int T, int Tp,
                     int M, int Bp, int Mp, void* dY, bool pooled) {
    const int64_t n = (int64_t)T * Bp * Mp;
    if (dt == BF16)
        hipLaunchKernelGGL(train_dy_kernel<bf16_t>, dim3(blocks256(n)), dim3(256), 0, stream, pred, target, scal, n_rows, T, Tp, M, Bp, Mp,
                           static_cast<bf16_t*>(dY), pooled ? 1 : 0);
    else
        hipLaunchKernelGGL(train_dy_kernel<float>, dim3(blocks256(n)), dim3(256), 0, stream, pred, target, scal, n_rows, T, Tp, M, Bp, Mp,
                           static_cast<float*>(dY), pooled ? 1 : 0);
}

void launch_adam_matrix(hipStream_t stream, int dt, const float* grad, int nblk, int R, int C, int Rp, int Cp, double* x, double* am,
                        double* av, void* W, void* WT, const AdamHyper& hp) {
    const int64_t n = (int64_t)nblk * R * C;
    if (dt == BF16)
        hipLaunchKernelGGL(adam_matrix_kernel<bf16_t>, dim3(blocks256(n)), dim3(256), 0, stream, grad, nblk, R, C, Rp, Cp, x, am, av,
                           static_cast<bf16_t*>(W), static_cast<bf16_t*>(WT), hp.lr, hp.b1, hp.b2, hp.eps, hp.bc1, hp.bc2);
    else
        hipLaunchKernelGGL(adam_matrix_kernel<float>, dim3(blocks256(n)), dim3(256), 0, stream, grad, nblk, R, C, Rp, Cp, x, am, av,
                           static_cast<float*>(W), static_cast<float*>(WT), hp.lr, hp.b1, hp.b2, hp.eps, hp.bc1, hp.bc2);
}

void launch_adam_bias(hipStream_t stream, const float* grad, int nblk, int R, int Rp, double* x0, double* am0, double* av0, double* x1,
                      double* am1, double* av1, float* packed, const AdamHyper& hp) {
    hipLaunchKernelGGL(adam_bias_kernel, dim3(blocks256(nblk * R)), dim3(256), 0, stream, grad, nblk, R, Rp, x0, am0, av0, x1, am1, av1,
                       packed, hp.lr, hp.b1, hp.b2, hp.eps, hp.bc1, hp.bc2);
}

}  // namespace pl
