// Batched (non-recurrent) GEMMs of the planner:  C[M,N] = A[M,K] * W[N,K]^T (+ bias)
//
// Used for: LSTM input projections for all time steps (W_ih x_t + b), post_linear / linear_mapping,
// and the backward-data products dA * W_ih, dY * W_p, dsem * W_m (weights pre-transposed at upload,
// so every product is "NT").  M = T * Bp rows of a time-major activation slab.
#include <atomic>

#include "kernels.h"
#include "tile_gemm.h"

namespace pl {

template <typename AT, typename OT, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const AT* __restrict__ A, int lda, const AT* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias, OT* __restrict__ C, int ldc, int M,
                                                      int N, int K, int n_blocks_n) {
    using TG = TileGemm<AT, BM, BN, WM, WN>;
    constexpr int EPI_BYTES = (sizeof(OT) == 2 && WM == 64 && WN == 64) ? 4 * 64 * (64 * 2 + 16) : 0;   // transposed-epilogue tiles
    __shared__ __attribute__((aligned(16))) unsigned char lds[TG::LDS_BYTES > EPI_BYTES ? TG::LDS_BYTES : EPI_BYTES];
    const int bid = blockIdx.x;
    const int m0 = (bid / n_blocks_n) * BM;
    const int n0 = (bid % n_blocks_n) * BN;

    f32x4 acc[TG::TM][TG::TN];
#pragma unroll
    for (int i = 0; i < TG::TM; ++i)
#pragma unroll
        for (int j = 0; j < TG::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto arow = [&](int r) -> const AT* { return (m0 + r < M) ? A + (size_t)(m0 + r) * lda : nullptr; };
    auto wrow = [&](int r) -> const AT* { return (n0 + r < N) ? W + (size_t)(n0 + r) * ldw : nullptr; };
    TG::run(arow, wrow, K, acc, lds);

    const auto cd = TG::coord();
    if constexpr (sizeof(OT) == 2 && WM == 64 && WN == 64) {
        // bf16 output: the MFMA C layout gives a lane one column x 4 rows per tile (2-byte stores, 32-byte segments: the
        // output-heavy projections ran at 2 TB/s).  Transpose each wave's 64 x 64 tile through LDS (free after the main
        // loop) and store whole 128-byte rows as 16-byte pieces.
        constexpr int RS = 64 * 2 + 16;   // row stride: odd number of 16-byte chunks
        static_assert(4 * 64 * RS <= EPI_BYTES, "epilogue tiles fit the LDS array");
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        unsigned char* tile = lds + wave * 64 * RS;
#pragma unroll
        for (int j = 0; j < TG::TN; ++j) {
            const int n = n0 + cd.n(j);
            const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TG::TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<OT*>(tile + (i * 16 + cd.kq * 4 + r) * RS + (j * 16 + cd.lr) * 2) =
                        from_f32<OT>(acc[i][j][r] + bv);
        }
        __syncthreads();
        const int mw = m0 + cd.wm * WM, nw = n0 + cd.wn * WN;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int cid = lane + 64 * q, row = cid >> 3, c8 = cid & 7;
            const int m = mw + row, n = nw + c8 * 8;
            if (m < M && n < N)   // N is a multiple of 16: a chunk is whole or absent
                *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = *reinterpret_cast<const uint4*>(tile + row * RS + c8 * 16);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TG::TN; ++j) {
        const int n = n0 + cd.n(j);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TG::TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + cd.m(i, r);
                if (m < M) C[(size_t)m * ldc + n] = from_f32<OT>(acc[i][j][r] + bv);
            }
    }
}

template <typename AT, typename OT, int BM, int BN, int WM, int WN>
static void launch_cfg(hipStream_t stream, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                       int ldc, int M, int N, int K) {
    const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_nt_kernel<AT, OT, BM, BN, WM, WN>), dim3(nbm * nbn), dim3(256), 0, stream,
                       static_cast<const AT*>(A), lda, static_cast<const AT*>(W), ldw, bias, static_cast<OT*>(C), ldc, M, N,
                       K, nbn);
}

// Tile choice: the large tiles when they make enough workgroups, else smaller ones (the time chunks of the layer wavefront are
// products of a few hundred rows: 128-row tiles would leave them on a handful of CUs).  Every configuration consumes K in the
// same order, so an output element has the same bits whichever tile computes it.
template <typename AT, typename OT>
static void launch_typed(hipStream_t stream, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                         int ldc, int M, int N, int K) {
    auto blocks = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    constexpr long kEnough = 64;
    // one block column (N <= 64): the product streams A once and is bound by how many CUs pull on HBM, so the large tile needs most of the
    // chip's worth of workgroups (round 5: cfg2's dL/dCP product ran 256 x 32 tiles on 75 CUs at 1.6 TB/s)
    // ... and more: with one workgroup per CU at a time, 300 large tiles are two rounds, the second on 44 CUs (cfg3's dL/dmel product: 65 us for
    // 226 MB); from three rounds on the tail no longer matters
    constexpr long kEnoughNarrow = 768;
    if (N <= 32) {
        if (blocks(256, 32) >= kEnoughNarrow || M <= 64)
            launch_cfg<AT, OT, 256, 32, 64, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else
            launch_cfg<AT, OT, 64, 32, 16, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    } else if (N <= 64) {
        if (blocks(128, 64) >= kEnoughNarrow || M <= 32)
            launch_cfg<AT, OT, 128, 64, 64, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);   // 2 x 2 waves of 64 x 32
        else
            launch_cfg<AT, OT, 32, 64, 16, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    } else {
        if (blocks(128, 128) >= kEnough || M <= 32)
            launch_cfg<AT, OT, 128, 128, 64, 64>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else if (blocks(64, 64) >= kEnough)
            launch_cfg<AT, OT, 64, 64, 32, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else
            launch_cfg<AT, OT, 32, 64, 16, 32>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    }
}

// gemm_big.hip's 256 x 256 tiles for the large bf16 products: on by default; a handle created with PAULE_HIP_GEMM_BIG=0 switches them off for
// what it enqueues (enqueue_iteration sets this before it launches anything; the results are bit-identical either way)
static std::atomic<bool> g_gemm_big{true};
void gemm_set_big(bool on) { g_gemm_big.store(on, std::memory_order_relaxed); }

void launch_gemm_nt(hipStream_t stream, int dt, bool out_f32, const void* A, int lda, const void* W, int ldw,
                    const float* bias, void* C, int ldc, int M, int N, int K) {
    if (M <= 0 || N <= 0) return;
    if (dt == BF16 && g_gemm_big.load(std::memory_order_relaxed) && gemm_big_takes(M, N, K)) {
        launch_gemm_nt_big(stream, out_f32, A, lda, W, ldw, bias, C, ldc, M, N, K);
        return;
    }
    if (dt == BF16) {
        if (out_f32)
            launch_typed<bf16_t, float>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else
            launch_typed<bf16_t, bf16_t>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    } else {
        launch_typed<float, float>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    }
}

}  // namespace pl
