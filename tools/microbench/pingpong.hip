// Microbenchmark: one-way latency of a flag hand-off between two workgroups (ping-pong), by store / load flavour and by
// placement (same XCD or different XCDs).  hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int ST, int LD>   // ST: 0 plain, 1 sc1 ; LD: 0 sc1, 1 nt
__global__ void pingpong(int* flags, int* xcc, int a_blk, int b_blk, int iters, unsigned long long* out) {
    const int bid = blockIdx.x;
    if (threadIdx.x == 0) xcc[bid] = (int)__builtin_amdgcn_s_getreg(6164);
    if (bid != a_blk && bid != b_blk) return;
    if (threadIdx.x != 0) return;
    const bool is_a = bid == a_blk;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 1024, 0x00020000);
    const unsigned mine = is_a ? 0u : 256u, theirs = is_a ? 256u : 0u;   // separate 128-B lines
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 1; k <= iters; ++k) {
        if (is_a) {
            if (ST) __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, r, mine, 0, 16); else __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, r, mine, 0, 0);
        }
        for (;;) {
            asm volatile("" ::: "memory");   // the poll must be re-issued every time
            unsigned v = LD ? __builtin_amdgcn_raw_buffer_load_b32(r, theirs, 0, 2) : __builtin_amdgcn_raw_buffer_load_b32(r, theirs, 0, 16);
            if ((int)v == k) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { out[1] = 0xdead; return; }   // 3 s guard
        }
        if (!is_a) {
            if (ST) __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, r, mine, 0, 16); else __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, r, mine, 0, 0);
        }
    }
    if (is_a) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = 0; }
}

int main() {
    int *flags, *xcc; unsigned long long* out;
    hipMalloc(&flags, 4096); hipMalloc(&xcc, 4096); hipMalloc(&out, 64);
    const int nblk = 64, iters = 2000;
    // find placement: run once to read XCC ids
    hipMemset(flags, 0, 4096);
    hipLaunchKernelGGL((pingpong<1, 0>), dim3(nblk), dim3(64), 0, 0, flags, xcc, 0, 8, 1, out);
    hipDeviceSynchronize();
    std::vector<int> hx(nblk); hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost);
    printf("xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    struct { const char* name; int b; } place[] = {{"same-XCD (blocks 0, 8)", 8}, {"cross-XCD (blocks 0, 1)", 1}};
    for (auto& pl : place) {
        for (int st = 0; st < 2; ++st) for (int ld = 0; ld < 2; ++ld) {
            hipMemset(flags, 0, 4096);
            unsigned long long h = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(flags, 0, 4096);
                if (st == 0 && ld == 0) hipLaunchKernelGGL((pingpong<0, 0>), dim3(nblk), dim3(64), 0, 0, flags, xcc, 0, pl.b, iters, out);
                if (st == 0 && ld == 1) hipLaunchKernelGGL((pingpong<0, 1>), dim3(nblk), dim3(64), 0, 0, flags, xcc, 0, pl.b, iters, out);
                if (st == 1 && ld == 0) hipLaunchKernelGGL((pingpong<1, 0>), dim3(nblk), dim3(64), 0, 0, flags, xcc, 0, pl.b, iters, out);
                if (st == 1 && ld == 1) hipLaunchKernelGGL((pingpong<1, 1>), dim3(nblk), dim3(64), 0, 0, flags, xcc, 0, pl.b, iters, out);
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
                unsigned long long hh[2];
                hipMemcpy(hh, out, 16, hipMemcpyDeviceToHost);
                h = hh[0];
                if (hh[1]) { printf("TIMEOUT "); break; }
            }
            hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost);
            printf("%-26s store %-5s load %-3s : one-way %.3f us  (xcc %d -> %d)\n", pl.name, st ? "sc1" : "plain", ld ? "nt" : "sc1",
                   h * 0.01 / iters / 2, hx[0], hx[pl.b]);
        }
    }
    return 0;
}
