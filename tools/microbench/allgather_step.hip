// Microbenchmark (round 5): what ONE step of the forward recurrence's exchange costs by hand-off path and tile layout, with the chip loaded the
// way the role-fused forward launch loads it.  A set of P = 23 workgroups (one batch group of 32 rows, Hp = 736) repeats per step:
//   1. every workgroup stores its own h slice (32 rows x 32 units bf16 = 2 KB)            -- write-through (sc1) or plain
//        layout ROW : row-major [32][Hp], the slice is 32 pieces of 64 B (half lines; what the stash layout gives the kernels today)
//        layout TILE: [slice][32 rows][64 B], the slice is 2 KB contiguous (16 whole lines)
//   2. drains, barrier, one lane raises the workgroup's flag (token = step + 1)           -- sc1 or plain
//   3. wave 0 polls the 23 flags of the set (sc1 or nt loads; optional s_sleep between polls), barrier
//   4. all four waves fetch the 23 slices into LDS by LDS-DMA, 46 pieces of 1 KB (sc1 or nt), wait, barrier
//   5. optional pause that stands for the MFMA chain and the cell update (s_sleep)
// Sets sit on one XCD each under the observed dealing (set = block % n_sets, n_sets a multiple of 8); the dynamic LDS size decides whether one
// or two workgroups share a CU.  Output: us per step (median over sets of the per-workgroup time / steps).
//   hipcc --offload-arch=gfx950 -O3 allgather_step.hip -o allgather_step && ./allgather_step
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int P = 23, HP = 736, ROWB = HP * 2;

__device__ __forceinline__ void glds16(const void* gsrc_uniform, unsigned lane_off, unsigned lds_dst_uniform, int nt) {
    unsigned keep;
    if (nt)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst_uniform) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst_uniform) : "memory");
}

// L2PATH: 0 write-through both sides (sc1 stores, sc1 flags, sc1 polls, sc1 DMA), 1 plain stores / plain flags / nt polls / nt DMA
// TILE: 0 row-major slices, 1 tile-major slices
template <int L2PATH, int TILE>
__global__ __launch_bounds__(256) void ag_kernel(unsigned char* xbuf, int* flags, int n_sets, int steps, int poll_sleep, int compute_sleep,
                                                 unsigned long long* out, int* xcc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = blockIdx.x % n_sets, p = blockIdx.x / n_sets;
    if (p >= P) return;
    if (tid == 0) xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(6164);
    const size_t slot_bytes = (size_t)32 * ROWB;                       // one set's h tile (both layouts: 47 KB)
    unsigned char* const base = xbuf + (size_t)set * 2 * slot_bytes;  // [2 slots]
    int* const fl = flags + (size_t)set * 2 * 32;                      // [2 slots][32]
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    __shared__ int lflag;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < steps; ++t) {
        unsigned char* const slot = base + (size_t)(t & 1) * slot_bytes;
        // 1. own slice: 128 threads x 16 B
        if (wave < 2) {
            const int row = tid >> 2, q = tid & 3;
            const size_t off = TILE ? (size_t)p * 2048 + row * 64 + q * 16 : (size_t)row * ROWB + p * 64 + q * 16;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, (unsigned)slot_bytes, 0x00020000);
            u32x4 v = {(unsigned)t, (unsigned)tid, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, (unsigned)off, 0, L2PATH ? 0 : 16);
        }
        // 2. drain, barrier, flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(fl + (t & 1) * 32 + p, 0, 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + 1), rf, 0, 0, L2PATH ? 0 : 16);
        }
        // 3. poll the set's flags
        if (wave == 0) {
            const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(fl + (t & 1) * 32, 0, P * 4, 0x00020000);
            int ok = 1;
            for (unsigned spin = 0;; ++spin) {
                int v = t + 1;
                if (lane < P) v = (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, L2PATH ? 2 : 16);
                if (__all(v == t + 1)) break;
                if (poll_sleep) __builtin_amdgcn_s_sleep(1);
                if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { ok = 0; break; }   // 2 s guard
            }
            if (lane == 0) lflag = ok;
        }
        __syncthreads();
        if (!lflag) { if (tid == 0) out[blockIdx.x] = ~0ull; return; }
        // 4. the 23 slices -> LDS, 46 pieces of 1 KB (16 rows x 64 B), wave w takes pieces w, w + 4, ...
        for (int pc = wave; pc < 2 * P; pc += 4) {
            const int kb = pc >> 1, half = pc & 1;
            const int row = 16 * half + (lane >> 2), q = lane & 3;
            const unsigned loff = TILE ? (unsigned)(row * 64 + q * 16) : (unsigned)(row * ROWB + q * 16);
            const unsigned char* src = slot + (TILE ? (size_t)kb * 2048 : (size_t)kb * 64);
            glds16(src, loff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(pc * 1024))), L2PATH);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // 5. stand-in for the MFMA chain + cell update
        if (compute_sleep) {
            const unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - c0 < (unsigned long long)compute_sleep) __builtin_amdgcn_s_sleep(2);
        }
    }
    if (tid == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

int main(int argc, char** argv) {
    const int steps = 300;
    unsigned char* xbuf; int* flags; unsigned long long* out; int* xcc;
    const int max_sets = 24;
    hipMalloc(&xbuf, (size_t)max_sets * 2 * 32 * ROWB);
    hipMalloc(&flags, (size_t)max_sets * 2 * 32 * 4);
    hipMalloc(&out, 1024 * 8);
    hipMalloc(&xcc, 1024 * 4);
    struct Cfg { const char* name; int n_sets; int lds; int compute; };
    const Cfg cfgs[] = {
        {"1 workgroup per CU,  8 sets (184 WGs), no compute ", 8, 120 * 1024, 0},
        {"1 workgroup per CU,  8 sets (184 WGs), 2.0 us busy", 8, 120 * 1024, 200},
        {"2 workgroups per CU, 16 sets (368 WGs), no compute ", 16, 64 * 1024, 0},
        {"2 workgroups per CU, 16 sets (368 WGs), 2.0 us busy", 16, 64 * 1024, 200},
        {"2 workgroups per CU, 16 sets (368 WGs), 3.0 us busy", 16, 64 * 1024, 300},
    };
    for (const Cfg& c : cfgs) {
        printf("== %s\n", c.name);
        for (int l2 = 0; l2 < 2; ++l2)
            for (int tile = 0; tile < 2; ++tile)
                for (int ps = 0; ps < 2; ++ps) {
                    double best = 1e30;
                    bool bad = false;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipMemset(flags, 0, (size_t)max_sets * 2 * 32 * 4);
                        hipMemset(out, 0, 1024 * 8);
                        const dim3 grid(c.n_sets * P), blk(256);
#define LAUNCH(L, T) hipFuncSetAttribute((const void*)ag_kernel<L, T>, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds); \
                     hipLaunchKernelGGL((ag_kernel<L, T>), grid, blk, c.lds, 0, xbuf, flags, c.n_sets, steps, ps, c.compute, out, xcc)
                        if (l2 == 0 && tile == 0) { LAUNCH(0, 0); }
                        if (l2 == 0 && tile == 1) { LAUNCH(0, 1); }
                        if (l2 == 1 && tile == 0) { LAUNCH(1, 0); }
                        if (l2 == 1 && tile == 1) { LAUNCH(1, 1); }
                        if (hipDeviceSynchronize() != hipSuccess) { printf("launch error\n"); return 1; }
                        std::vector<unsigned long long> h(c.n_sets * P);
                        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                        for (auto v : h) if (v == ~0ull) bad = true;
                        std::sort(h.begin(), h.end());
                        best = std::min(best, h[h.size() / 2] * 0.01 / steps);
                    }
                    std::vector<int> hx(c.n_sets * P);
                    hipMemcpy(hx.data(), xcc, hx.size() * 4, hipMemcpyDeviceToHost);
                    int one_xcd = 0;
                    for (int s = 0; s < c.n_sets; ++s) {
                        bool same = true;
                        for (int p = 1; p < P; ++p) same = same && hx[p * c.n_sets + s] == hx[s];
                        one_xcd += same;
                    }
                    printf("   %-14s %-9s poll %-6s : %6.2f us per step%s   (%d of %d sets on one XCD)\n", l2 ? "L2 (plain/nt)" : "write-through", tile ? "tile-major" : "row-major",
                           ps ? "sleep" : "tight", best, bad ? "  TIMEOUT" : "", one_xcd, c.n_sets);
                }
    }
    return 0;
}
