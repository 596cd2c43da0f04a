// LDS-DMA ring pipeline for the latency-critical LSTM step kernels.
//
//   acc[BM x BN] += A[BM x K] * W[BN x K]^T      (both operands K-contiguous)
//
// One launch of a step kernel has to pull its whole A and W tiles (100-400 KB per workgroup) out of
// L2 / Infinity Cache exactly once, with one workgroup per CU: what bounds it is how many bytes the CU
// keeps in flight, not the MFMA rate.  So K is cut into stages of RB bytes per row and up to D stages
// (D * (BM + BN) * RB bytes <= 128 KB) are kept in flight by `global_load_lds_dwordx4` (LDS-DMA: no VGPR
// staging, 1 KB per wave-instruction), retired by COUNTED `s_waitcnt vmcnt(N)` + raw `s_barrier`, and the
// ring slot of a consumed stage is refilled at once.  The DMA writes LDS lane-linearly (wave-uniform
// base + 16 * lane), so the bank swizzle is applied to the per-lane SOURCE address: the 16-byte chunk
// stored at position p of row r is chunk p ^ (r & 15) of that row's stage slice; fragment reads apply the
// same XOR (conflict-free ds_read_b128: 16 rows x one k-chunk per lane group).
//
// The DMA is issued from inline asm (the compiler would otherwise drain vmcnt(0) before every ds_read
// that may alias an in-flight LDS-DMA).  Rule kept by the callers: no other global load/store between the
// first issue and the last wait of run() (older loads are fine: vmcnt retires in order).
#pragma once
#include "pl_types.h"
#include "tile_gemm.h"

namespace pl {

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// SPLITK = false: wave w owns A rows [16w, 16w+16) and all BN columns (TN = BN/16 accumulators)  [BM = 64]
// SPLITK = true : every wave owns the whole BM x BN tile and every 4th k-step; the caller reduces the
//                 four partial accumulators.
template <typename AT, int BM, int BN, int RB, int D, bool SPLITK>
struct RingGemm {
    static constexpr int R = BM + BN;                  // LDS rows per stage
    static constexpr int SB = R * RB;                  // bytes per stage
    static constexpr int NP = SB / 1024;               // 1-KB DMA pieces per stage
    static constexpr int LPT = NP / 4;                 // DMA instructions per wave per stage
    static_assert(NP % 4 == 0 && SB % 1024 == 0, "stage must split evenly over 4 waves");
    static_assert(RB == 256 || RB == 512, "stage row width");
    static_assert(D >= 2 && D <= 4, "ring depth");
    static constexpr int LDS_BYTES = D * SB;
    static constexpr int KSTEPS = RB / 64;             // MFMA k-steps per stage (64 bytes of K per row each)
    static constexpr int TM = SPLITK ? BM / 16 : 1;
    static constexpr int TN = BN / 16;
    static_assert(SPLITK || BM == 64, "non-split form: 4 waves x 16 rows");

    template <int LATER> static __device__ __forceinline__ void wait_stage() { wait_vmcnt<LATER * LPT>(); }

    // arow(r), wrow(r): row base pointers (never null: callers clamp out-of-range rows to a valid row).
    // Kb = K in bytes (multiple of 64).
    // st: optional in-register stamp array (diagnostic builds only): [1] DMA prologue issued, [2] first stage
    // landed, [3] last stage computed
    template <class ARowFn, class WRowFn>
    __device__ static __forceinline__ void run(ARowFn arow, WRowFn wrow, int Kb, f32x4 (&acc)[TM][TN], unsigned char* lds,
                                               unsigned long long* st = nullptr) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lr = lane & 15, kq = lane >> 4;
        using lds_ptr_t = __attribute__((address_space(3))) unsigned char*;
        const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)lds;   // LDS byte address of the ring
        const int NS = (Kb + RB - 1) / RB;

        // per-lane source of each of this wave's pieces: row pointer + swizzled chunk offset (stage invariant)
        const unsigned char* src[LPT];
        int coff[LPT];
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int off = (i * 4 + wave) * 1024 + lane * 16;
            const int row = off / RB, pos = (off % RB) >> 4;
            const int c = pos ^ (row & 15);
            src[i] = reinterpret_cast<const unsigned char*>(row < BM ? (const void*)arow(row) : (const void*)wrow(row - BM));
            coff[i] = c << 4;
        }
        auto issue = [&](int s) {
            const unsigned dst = lds_base + (unsigned)((s % D) * SB + wave * 1024);
            const int k0 = s * RB;
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                int o = k0 + coff[i];
                o = o < Kb ? o : Kb - 16;   // K tail: keep the DMA count uniform, the bytes are never read
                glds16(src[i] + o, dst + (unsigned)(i * 4096));
            }
        };

        int issued = 0;
        for (; issued < D && issued < NS; ++issued) issue(issued);
        if (st) st[1] = __builtin_amdgcn_s_memrealtime();

        for (int s = 0; s < NS; ++s) {
            const int later = issued - (s + 1);
            if (later >= 3) wait_stage<3>();
            else if (later == 2) wait_stage<2>();
            else if (later == 1) wait_stage<1>();
            else wait_stage<0>();
            block_barrier();
            if (st && s == 0) st[2] = __builtin_amdgcn_s_memrealtime();

            const unsigned char* buf = lds + (s % D) * SB;
            const int kb_left = Kb - s * RB;
            const int ksteps = kb_left >= RB ? KSTEPS : kb_left / 64;
            if (SPLITK) {
                for (int ks = wave; ks < ksteps; ks += 4) {
                    const int sw = ((4 * ks + kq) ^ lr) << 4;
                    uint4 a[TM], w[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const uint4*>(buf + (i * 16 + lr) * RB + sw);
#pragma unroll
                    for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const uint4*>(buf + (BM + j * 16 + lr) * RB + sw);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) MfmaStep<AT>::run(a[i], w[j], acc[i][j]);
                }
            } else {
#pragma unroll 2
                for (int ks = 0; ks < ksteps; ++ks) {
                    const int sw = ((4 * ks + kq) ^ lr) << 4;
                    const uint4 a = *reinterpret_cast<const uint4*>(buf + (wave * 16 + lr) * RB + sw);
                    uint4 w[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) w[j] = *reinterpret_cast<const uint4*>(buf + (BM + j * 16 + lr) * RB + sw);
#pragma unroll
                    for (int j = 0; j < TN; ++j) MfmaStep<AT>::run(a, w[j], acc[0][j]);
                }
            }
            if (issued < NS) {          // refill the slot just consumed
                block_barrier();
                issue(issued);
                ++issued;
            }
        }
        block_barrier();   // every wave is done with the ring: callers may reuse the LDS
        if (st) st[3] = __builtin_amdgcn_s_memrealtime();
    }
};

}  // namespace pl
