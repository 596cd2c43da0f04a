// Large bf16 products of the planner on 256 x 256 workgroup tiles (round 5):  C[M,N] = A[M,K] * W[N,K]^T (+ bias), both operands K-contiguous.
//
// The tile engine of gemm.hip (128 x 128 tiles, four waves -- one per SIMD -- of 64 x 64, operands staged through registers into ONE LDS
// buffer, two barriers per 32 MFMAs of a wave) ran the iteration's largest product -- dL/dh of embedder layer 1, 38400 x 736 x 2944 at cfg3
// -- at 0.73 PFLOP/s: nothing overlaps a wave's barrier and staging phases.  Here:
//   * 512 threads = 8 waves (two per SIMD: one's MFMAs run under the other's waits) as 2 x 4, each a 128 x 64 tile (8 x 4 MFMA tiles, 128
//     accumulator registers): 12 fragment reads per 32 MFMAs instead of 8 per 16;
//   * both operand tiles go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no registers), two stage buffers of 64 KB (two k-steps
//     each), ONE barrier per stage; the 16-byte-chunk XOR swizzle of tile_gemm.h is applied to the per-lane SOURCE address (the DMA writes
//     LDS lane-linearly).  A ring of four stages of ONE k-step (three in flight, counted vmcnt) was measured slower -- 187 against 170 us: the
//     kernel does not wait for memory, the barrier per 32 MFMAs costs more than the deeper ring gains (profiles/r05_ab_gemm_big.txt);
//   * workgroup ids are dealt so that the N tiles of one M tile run on one XCD at the same time (its A rows come out of that XCD's L2);
//   * bf16 output leaves through LDS as whole 16-byte pieces of 128-byte rows, as in gemm.hip.
// The MFMA (v_mfma_f32_16x16x32_bf16), the k order (ascending in steps of 32) and the epilogue arithmetic are those of gemm.hip: an output
// element has the same bits whichever kernel computes it (tests/test_hip_parity.py::test_big_gemm_matches_tile_gemm_bit_for_bit).
#include "kernels.h"
#include "ring_gemm.h"

namespace pl {

namespace {

constexpr int GB_BM = 256, GB_BN = 256, GB_ROWB = 128, GB_KT = 64;
constexpr int GB_STAGE = (GB_BM + GB_BN) * GB_ROWB;   // 64 KB: two MFMA k-steps of both tiles
constexpr int GB_LDS = 2 * GB_STAGE;                    // 128 KB
constexpr int GB_RS = 64 * 2 + 16;                      // epilogue row stride (odd number of 16-byte chunks)

template <typename OT>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, OT* __restrict__ C, int ldc, int M, int N, int K,
                                                          int nbn, int n_wg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    // ids of one XCD (id % 8 under the observed dealing) take consecutive tiles, N tile fastest
    const int per = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (logical >= n_wg) return;
    const int m0 = (logical / nbn) * GB_BM, n0 = (logical % nbn) * GB_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, lr = lane & 15, kq = lane >> 4;
    using lds_ptr_t = __attribute__((address_space(3))) unsigned char*;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    // A stage is 512 LDS rows of 128 bytes (rows 0 .. 255 the A tile, 256 .. 511 the W tile).  DMA piece q = 8 wave + i covers rows 8 q ..
    // 8 q + 7; this lane brings position lane & 7 of row 8 q + (lane >> 3), i.e. chunk (lane & 7) ^ (row & 7) of that row's 128 bytes of K:
    // a fragment read (ds_read_b128: 16 rows x one chunk per 16-lane group) then touches every one of the 64 banks once
    const unsigned char* src[8];
    const int rsub = lane >> 3, chunk = (lane & 7) ^ rsub;   // row & 7 == rsub
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + rsub;
        if (row < GB_BM) {
            int m = m0 + row;
            m = m < M ? m : M - 1;   // rows outside the problem read a valid row; their results are never stored
            src[i] = reinterpret_cast<const unsigned char*>(A + (size_t)m * lda);
        } else {
            int n = n0 + row - GB_BM;
            n = n < N ? n : N - 1;
            src[i] = reinterpret_cast<const unsigned char*>(W + (size_t)n * ldw);
        }
    }
    const int nk = (K + GB_KT - 1) / GB_KT;
    const bool k_tail = (K % GB_KT) != 0;   // K = 32 (mod 64): the last stage holds one k-step; its upper chunks re-read the lower ones
    auto issue = [&](int ks) {
        const unsigned dst = lds_base + (unsigned)((ks & 1) * GB_STAGE + wave * 8 * 1024);
        const int c = (k_tail && ks == nk - 1) ? (chunk & 3) : chunk;
        const size_t koff = (size_t)ks * GB_ROWB + (size_t)c * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(src[i] + koff, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + i * 1024)));
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    for (int ks = 0; ks < nk; ++ks) {
        wait_vmcnt<0>();     // this wave's pieces of stage ks
        block_barrier();     // everybody's; and everybody is through with the buffer stage ks + 1 goes into
        if (ks + 1 < nk) issue(ks + 1);
        const unsigned char* sA = lds + (ks & 1) * GB_STAGE;
        const unsigned char* sW = sA + GB_BM * GB_ROWB;
        const int nsteps = (k_tail && ks == nk - 1) ? 1 : 2;
        for (int s = 0; s < nsteps; ++s) {
            const int sw = ((4 * s + kq) ^ (lr & 7)) << 4;
            uint4 a[8], w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const uint4*>(sW + (wn * 64 + j * 16 + lr) * GB_ROWB + sw);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const uint4*>(sA + (wm * 128 + i * 16 + lr) * GB_ROWB + sw);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) MfmaStep<bf16_t>::run(a[i], w[j], acc[i][j]);
        }
    }

    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + lr;
        bv[j] = (bias && n < N) ? bias[n] : 0.f;
    }
    if constexpr (sizeof(OT) == 2) {
        unsigned char* tile = lds + wave * 64 * GB_RS;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();   // first: the last stage's reads; second: the first half's stores
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<OT*>(tile + (i * 16 + kq * 4 + r) * GB_RS + (j * 16 + lr) * 2) = from_f32<OT>(acc[4 * half + i][j][r] + bv[j]);
            __syncthreads();
            const int mw = m0 + wm * 128 + half * 64, nw = n0 + wn * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cid = lane + 64 * q, row = cid >> 3, c8 = cid & 7;
                const int m = mw + row, n = nw + c8 * 8;
                if (m < M && n < N)   // N is a multiple of 16: a chunk is whole or absent
                    *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = *reinterpret_cast<const uint4*>(tile + row * GB_RS + c8 * 16);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + lr;
            if (n >= N) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 128 + i * 16 + kq * 4 + r;
                    if (m < M) C[(size_t)m * ldc + n] = from_f32<OT>(acc[i][j][r] + bv[j]);
                }
        }
    }
}

template <typename OT>
void launch_big(hipStream_t stream, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K) {
    const int nbm = (M + GB_BM - 1) / GB_BM, nbn = (N + GB_BN - 1) / GB_BN, n_wg = nbm * nbn;
    const int grid = (n_wg + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_nt_big_kernel<OT>), dim3(grid), dim3(512), GB_LDS, stream, static_cast<const bf16_t*>(A), lda,
                       static_cast<const bf16_t*>(W), ldw, bias, static_cast<OT*>(C), ldc, M, N, K, nbn, n_wg);
}

}  // namespace

void gemm_big_init() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_big_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_big_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS);
}

// the shapes the 256 x 256 tiles are for: enough tiles to fill the chip, K in whole k-steps
bool gemm_big_takes(int M, int N, int K) {
    if (K < 256 || K % 32 != 0 || N < 384 || N % 16 != 0 || M < 2048) return false;
    const long tiles = (long)((M + GB_BM - 1) / GB_BM) * ((N + GB_BN - 1) / GB_BN);
    return tiles >= 192;
}

void launch_gemm_nt_big(hipStream_t stream, bool out_f32, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                        int M, int N, int K) {
    if (out_f32)
        launch_big<float>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    else
        launch_big<bf16_t>(stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
}

}  // namespace pl
