// Workgroup-level MFMA tile engine shared by the batched GEMM and the LSTM step kernels.
//
//   acc[BM x BN] += A[BM x K] * W[BN x K]^T        (both operands K-contiguous, "NT")
//
// 256 threads = 4 wave64s arranged (BM/WM) x (BN/WN); each wave owns a WM x WN block made of
// 16x16 MFMA tiles (TM x TN accumulators of 4 VGPRs).  K is consumed in stages of ROWB = 128 /
// 256 / 512 bytes per row: both tiles are staged through LDS with a 16-byte-chunk XOR swizzle
// (chunk c of row r is stored at chunk c ^ (r & SWZ)), which makes the ds_read_b128 fragment
// reads (16 rows x one k-chunk per 16-lane group) conflict free under the 256-byte LDS bank row.  The next stage's global loads are issued before the MFMAs of
// the current one (register-staged software pipeline, one LDS buffer, two barriers per stage).
//
// MFMA shapes (gfx950):
//   bf16: v_mfma_f32_16x16x32_bf16  -- lane l holds A[row l&15][k 8(l>>4)..+7], same for W
//   f32 : v_mfma_f32_16x16x4_f32    -- exact f32 FMA chain; a lane's 16-byte chunk (4 floats at
//         k 4(l>>4)..+3) feeds 4 consecutive MFMAs, element e of A and of W in MFMA e, so both
//         operands see the same (permuted) k order.
//   C/D : col = lane & 15, row = 4 * (lane >> 4) + reg.
#pragma once
#include "pl_types.h"

namespace pl {

template <typename AT> struct MfmaStep;

template <> struct MfmaStep<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& w, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, w),
                                                      acc, 0, 0, 0);
    }
};

template <> struct MfmaStep<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& w, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, w.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, w.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, w.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, w.w), acc, 0, 0, 0);
    }
};

template <typename AT, int BM_, int BN_, int WM_, int WN_, int ROWB_ = 128>
struct TileGemm {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NT = 256;
    static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile is made of 16x16 MFMA tiles");
    static constexpr int TM = WM / 16, TN = WN / 16;
    static constexpr int ROWB = ROWB_;                    // bytes of K per LDS row per stage (128 / 256 / 512)
    static_assert(ROWB == 128 || ROWB == 256 || ROWB == 512, "stage width");
    static constexpr int CPR = ROWB / 16;                 // 16-byte chunks per LDS row
    static constexpr int SWZ = (CPR < 16 ? CPR : 16) - 1; // chunk c of row r lives at chunk c ^ (r & SWZ)
    static constexpr int KSTEPS = ROWB / 64;              // MFMA k-steps (64 bytes of K per row) per stage
    static constexpr int KT = ROWB / (int)sizeof(AT);     // K elements per stage
    static constexpr int EPC = 16 / (int)sizeof(AT);      // elements per 16-byte chunk
    static constexpr int A_CHUNKS = BM * CPR, W_CHUNKS = BN * CPR;
    static constexpr int A_PER_T = (A_CHUNKS + NT - 1) / NT;
    static constexpr int W_PER_T = (W_CHUNKS + NT - 1) / NT;
    static constexpr int LDS_BYTES = (BM + BN) * ROWB;

    struct Coord {                      // coordinates of this lane inside the workgroup tile
        int wm, wn, lr, kq;
        __device__ __forceinline__ int m(int i, int r) const { return wm * WM + i * 16 + kq * 4 + r; }
        __device__ __forceinline__ int n(int j) const { return wn * WN + j * 16 + lr; }
    };

    __device__ static __forceinline__ Coord coord() {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        return Coord{wave / WAVES_N, wave % WAVES_N, lane & 15, lane >> 4};
    }

    // arow(r) / wrow(r): pointer to element k = 0 of tile row r (16-byte aligned), or nullptr for a
    // row outside the problem (it then contributes zeros).  K must be a multiple of 32 elements.
    template <class ARowFn, class WRowFn>
    __device__ static __forceinline__ void run(ARowFn arow, WRowFn wrow, int K, f32x4 (&acc)[TM][TN],
                                               unsigned char* lds) {
        if (K <= 0) return;
        const int tid = threadIdx.x;
        unsigned char* ldsA = lds;
        unsigned char* ldsW = lds + BM * ROWB;

        const AT* a_src[A_PER_T];
        const AT* w_src[W_PER_T];
        int a_dst[A_PER_T], w_dst[W_PER_T], a_k[A_PER_T], w_k[W_PER_T];
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            const AT* p = (q < A_CHUNKS) ? arow(row) : nullptr;
            a_src[i] = p ? p + c * EPC : nullptr;
            a_k[i] = c * EPC;
            a_dst[i] = (q < A_CHUNKS) ? row * ROWB + ((c ^ (row & SWZ)) << 4) : -1;
        }
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            const AT* p = (q < W_CHUNKS) ? wrow(row) : nullptr;
            w_src[i] = p ? p + c * EPC : nullptr;
            w_k[i] = c * EPC;
            w_dst[i] = (q < W_CHUNKS) ? row * ROWB + ((c ^ (row & SWZ)) << 4) : -1;
        }

        uint4 ra[A_PER_T], rw[W_PER_T];
        auto load_stage = [&](int k0) {
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                ra[i] = make_uint4(0, 0, 0, 0);
                if (a_src[i] && k0 + a_k[i] < K) ra[i] = *reinterpret_cast<const uint4*>(a_src[i] + k0);
            }
#pragma unroll
            for (int i = 0; i < W_PER_T; ++i) {
                rw[i] = make_uint4(0, 0, 0, 0);
                if (w_src[i] && k0 + w_k[i] < K) rw[i] = *reinterpret_cast<const uint4*>(w_src[i] + k0);
            }
        };

        const Coord cd = coord();
        load_stage(0);
        for (int k0 = 0; k0 < K; k0 += KT) {
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i)
                if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(ldsA + a_dst[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < W_PER_T; ++i)
                if (w_dst[i] >= 0) *reinterpret_cast<uint4*>(ldsW + w_dst[i]) = rw[i];
            __syncthreads();
            if (k0 + KT < K) load_stage(k0 + KT);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const int c = 4 * s + cd.kq;
                const int sw = (c ^ (cd.lr & SWZ)) << 4;
                uint4 a[TM], w[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const uint4*>(ldsA + (cd.wm * WM + i * 16 + cd.lr) * ROWB + sw);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    w[j] = *reinterpret_cast<const uint4*>(ldsW + (cd.wn * WN + j * 16 + cd.lr) * ROWB + sw);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) MfmaStep<AT>::run(a[i], w[j], acc[i][j]);
            }
            __syncthreads();
        }
    }
};

}  // namespace pl
