// Microbenchmark for DESIGN.md section 9, item 1: how much of a chain-step's phases would overlap if TWO workgroups shared a CU.
//
// A workgroup (4 waves, one per SIMD) repeats a synthetic chain-step with the resource mix of the fused forward recurrence role:
//   A  47 KB of global loads (its own region, L2-resident) -> registers -> an LDS image
//   C  46 chained v_mfma_f32_32x32x16_bf16 per wave, B operand read from the LDS image every k-step
//   F  ~250 VALU instructions per lane, 40 of them transcendental (the cell update's mix)
//   G  15 KB through an LDS staging area out to global memory
// There are no dependencies between workgroups: this measures pipes, not latencies.  Launched with one workgroup per CU and with
// two (the kernel needs 64 KB of LDS and < 128 registers, so two fit), the same number of steps per workgroup: if the time
// does not grow, a second resident workgroup is free -- the matrix core, the VALU and the memory pipe of a CU overlap across
// workgroups although one wave executes its phases in sequence.
// hipcc --offload-arch=gfx950 -O3 occupancy_overlap.hip -o occupancy_overlap && ./occupancy_overlap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KS = 46, ROWB = KS * 32, RS = ROWB + 16;   // image [32 rows][736 bf16], padded rows
constexpr int CH = ROWB / 16, NL = (32 * CH + 255) / 256;

template <int MODE>   // 0: all phases, 1: MFMA only, 2: loads + image only, 3: VALU only, 4: stores only
__global__ __launch_bounds__(256, 2) void chain_steps(const uint4* __restrict__ src, uint4* __restrict__ dst, int steps, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char himg[32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char hst[6 * 32 * 80];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4* my = src + (size_t)blockIdx.x * (32 * CH);
    uint4* out = dst + (size_t)blockIdx.x * 1024;
    bf16x8 w[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) w[i][j] = (__bf16)(0.001f * (float)(lane + i + j));
    float keep = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0 || MODE == 2) {   // A
            uint4 hv[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int q = tid + 256 * i;
                hv[i] = q < 32 * CH ? my[q] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int q = tid + 256 * i;
                if (q < 32 * CH) *reinterpret_cast<uint4*>(himg + (q / CH) * RS + (q % CH) * 16) = hv[i];
            }
            __syncthreads();
        }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (MODE == 0 || MODE == 1) {   // C
            const unsigned char* bsrc = himg + (lane & 31) * RS + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 b = *reinterpret_cast<const uint4*>(bsrc + ks * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks & 3], __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            }
        }
        float v[16];
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + (float)s;
        if (MODE == 0 || MODE == 3) {   // F: 4 units x (3 sigmoids + 2 tanh) + the cell arithmetic
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float i_ = __frcp_rn(1.f + __expf(-v[u])), f_ = __frcp_rn(1.f + __expf(-v[4 + u]));
                const float g_ = 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * v[8 + u])), o_ = __frcp_rn(1.f + __expf(-v[12 + u]));
                const float c_ = f_ * keep + i_ * g_;
                const float h_ = o_ * (1.f - 2.f * __frcp_rn(1.f + __expf(2.f * c_)));
                v[u] = h_; v[4 + u] = i_ + f_; v[8 + u] = g_ + o_; v[12 + u] = c_;
                keep = c_ * 0.5f;
            }
        }
        if (MODE == 0 || MODE == 4) {   // G
            float* o = reinterpret_cast<float*>(hst) + tid * 8;
            for (int r = 0; r < 8; ++r) o[r] = v[r] + v[8 + r];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 sv = *reinterpret_cast<const uint4*>(hst + ((tid + 256 * i) % 480) * 16);
                out[(tid + 256 * i) & 1023] = sv;
            }
        }
        keep += v[0];
    }
    if (keep == 1234.5f) sink[0] = keep;
}

// The half-slice variant a redesign could actually fit in 256 registers: the workgroup owns 16 hidden units (64 gate rows: one 16-row
// A tile per wave = 92 weight registers), the 32 batch rows are two B tiles: 46 v_mfma_f32_16x16x32_bf16 per wave, the SAME 47 KB of
// h to load (the all-gather does not shrink with the slice), half the cell work and half the outputs.  Two of them per CU do the
// work of one full-slice workgroup.
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256, 2) void chain_steps_half(const uint4* __restrict__ src, uint4* __restrict__ dst, int steps, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char himg[32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char hst[6 * 32 * 48];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint4* my = src + (size_t)blockIdx.x * (32 * CH);
    uint4* out = dst + (size_t)blockIdx.x * 1024;
    bf16x8 w[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) w[i][j] = (__bf16)(0.001f * (float)(lane + i + j));
    float keep = 0.f;
    for (int s = 0; s < steps; ++s) {
        uint4 hv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i;
            hv[i] = q < 32 * CH ? my[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i;
            if (q < 32 * CH) *reinterpret_cast<uint4*>(himg + (q / CH) * RS + (q % CH) * 16) = hv[i];
        }
        __syncthreads();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* b0 = himg + (lane & 15) * RS + (lane >> 4) * 16;
        const unsigned char* b1 = b0 + 16 * RS;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {   // 23 k-steps of 32
            const uint4 x0 = *reinterpret_cast<const uint4*>(b0 + ks * 64), x1 = *reinterpret_cast<const uint4*>(b1 + ks * 64);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks & 3], __builtin_bit_cast(bf16x8, x0), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks & 3], __builtin_bit_cast(bf16x8, x1), a1, 0, 0, 0);
        }
        float v[8] = {a0[0] + s, a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float i_ = __frcp_rn(1.f + __expf(-v[4 * u])), f_ = __frcp_rn(1.f + __expf(-v[4 * u + 1]));
            const float g_ = 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * v[4 * u + 2])), o_ = __frcp_rn(1.f + __expf(-v[4 * u + 3]));
            const float c_ = f_ * keep + i_ * g_;
            const float h_ = o_ * (1.f - 2.f * __frcp_rn(1.f + __expf(2.f * c_)));
            v[4 * u] = h_; v[4 * u + 1] = i_ + f_; v[4 * u + 2] = g_ + o_; v[4 * u + 3] = c_;
            keep = c_ * 0.5f;
        }
        float* o = reinterpret_cast<float*>(hst) + tid * 4;
        for (int r = 0; r < 4; ++r) o[r] = v[r] + v[4 + r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint4 sv = *reinterpret_cast<const uint4*>(hst + ((tid + 256 * i) % 240) * 16);
            out[(tid + 256 * i) & 1023] = sv;
        }
        keep += v[0];
    }
    if (keep == 1234.5f) sink[0] = keep;
}

static float run_half(int grid, int steps, const uint4* src, uint4* dst, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_steps_half, dim3(grid), dim3(256), 0, 0, src, dst, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_steps_half, dim3(grid), dim3(256), 0, 0, src, dst, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE>
static float run(int grid, int steps, const uint4* src, uint4* dst, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_steps<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_steps<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int steps = 2000, max_grid = 512;
    uint4 *src, *dst;
    float* sink;
    hipMalloc(&src, (size_t)max_grid * 32 * CH * sizeof(uint4));
    hipMalloc(&dst, (size_t)max_grid * 1024 * sizeof(uint4));
    hipMalloc(&sink, 16);
    hipMemset(src, 0, (size_t)max_grid * 32 * CH * sizeof(uint4));
    const char* names[5] = {"all phases", "MFMA chain only", "loads + LDS image only", "cell VALU only", "staging + stores only"};
    printf("# %d chain-steps per workgroup; us per chain-step with 1 workgroup per CU (grid 256) and 2 per CU (grid 512)\n", steps);
    float t1[5], t2[5];
    t1[0] = run<0>(256, steps, src, dst, sink); t2[0] = run<0>(512, steps, src, dst, sink);
    t1[1] = run<1>(256, steps, src, dst, sink); t2[1] = run<1>(512, steps, src, dst, sink);
    t1[2] = run<2>(256, steps, src, dst, sink); t2[2] = run<2>(512, steps, src, dst, sink);
    t1[3] = run<3>(256, steps, src, dst, sink); t2[3] = run<3>(512, steps, src, dst, sink);
    t1[4] = run<4>(256, steps, src, dst, sink); t2[4] = run<4>(512, steps, src, dst, sink);
    for (int m = 0; m < 5; ++m)
        printf("%-24s  1/CU %6.2f us   2/CU %6.2f us per step of each workgroup  -> %.2f us of CU time per chain-step (x%.2f throughput)\n", names[m],
               t1[m] * 1e3f / steps, t2[m] * 1e3f / steps, t2[m] * 1e3f / steps / 2.f, 2.f * t1[m] / t2[m]);
    const float h1 = run_half(256, steps, src, dst, sink), h2 = run_half(512, steps, src, dst, sink);
    printf("half-slice workgroups (16 units, 92 weight registers): 1/CU %6.2f us   2/CU %6.2f us per step of each workgroup  -> two per CU do one\n"
           "full slice's chain-step in %.2f us of CU time (full slice, one workgroup per CU: %.2f us)\n",
           h1 * 1e3f / steps, h2 * 1e3f / steps, h2 * 1e3f / steps, t1[0] * 1e3f / steps);
    return 0;
}
