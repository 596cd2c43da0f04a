// LSTM time-step kernels (torch.nn.LSTM semantics used at paule/models.py:344, :431:
// gate order i,f,g,o; i,f,o = sigmoid, g = tanh; c' = f*c + i*g; h' = o*tanh(c'); h0 = c0 = 0).
//
// One launch = one time step of one layer for the whole batch slab:
//   forward : a = Gx_t + h_{t-1} * Whh^T          -> gates (stash), c_t, h_t
//   backward: dh = dh_ext_t + dA_{t+1} * Whh       -> dA_t (backward-DATA only; no dW)
// The recurrent product runs on the LDS-DMA ring pipeline (ring_gemm.h) with the cell arithmetic fused
// into the epilogue, so the gate pre-activations never leave registers.  One workgroup per CU; the
// operands of the epilogue are fetched BEFORE the product so their latency hides under it.
//
// Data layout (time-major slabs, every feature dim padded to 32 with zeros):
//   G   [Bp][4*Hp]  (gate-blocked: column g*Hp + j),  h / c stash [Bp][Hp],  running c / dc f32 [Bp][Hp]
#include "kernels.h"
#include "ring_gemm.h"

namespace pl {

template <typename AT> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    static __device__ __forceinline__ float4 load(const bf16_t* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    static __device__ __forceinline__ void store(bf16_t* p, float4 v) {
        bf16x4 o;
        o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
        *reinterpret_cast<bf16x4*>(p) = o;
    }
};

// ------------------------------------------------------------------------------------------
// forward: workgroup = 64 batch rows x 16 hidden units x 4 gates (K = Hp); wave w owns batch rows
// 16w..16w+15 and all four gates of the 16 hidden units (accumulator tile g = gate g), so the cell
// update is lane-local.  grid = (Hp / 16, ceil(Bp / 64)).
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmStepArgs a) {
    using RG = RingGemm<AT, 64, 64, 256, 4, false>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RG::LDS_BYTES];
    const int Hp = a.Hp, Bp = a.Bp, G4 = 4 * Hp;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 64;
    const AT* __restrict__ W = static_cast<const AT*>(a.W);
    const AT* __restrict__ hp = static_cast<const AT*>(a.h_prev);
    AT* __restrict__ G = static_cast<AT*>(a.G_t);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kq = lane >> 4;
    const int j = j0 + lr;
#ifdef PL_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st[0] = __builtin_amdgcn_s_memrealtime();
#else
    unsigned long long* st = nullptr;
#endif

    // epilogue operands first (older than the DMA stream, so every counted wait covers them)
    float gx[4][4], cprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + wave * 16 + kq * 4 + r;
        const bool ok = b < Bp;
        const AT* g_row = G + (size_t)(ok ? b : 0) * G4 + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) gx[r][g] = to_f32<AT>(g_row[g * Hp]);
        cprev[r] = a.c_in ? a.c_in[(size_t)(ok ? b : 0) * Hp + j] : 0.f;
    }

    f32x4 acc[1][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[0][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hp) {
        auto arow = [&](int r) -> const AT* { return hp + (size_t)(b0 + r < Bp ? b0 + r : 0) * Hp; };
        auto wrow = [&](int r) -> const AT* { return W + (size_t)((r >> 4) * Hp + j0 + (r & 15)) * Hp; };
        RG::run(arow, wrow, Hp * (int)sizeof(AT), acc, lds, st);
    }

    AT* __restrict__ h_out = static_cast<AT*>(a.h_out);
    AT* __restrict__ c_st = static_cast<AT*>(a.c_stash_t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + wave * 16 + kq * 4 + r;
        if (b >= Bp) continue;
        AT* g_row = G + (size_t)b * G4 + j;
        const float gi = act_sigmoid<AT>(acc[0][0][r] + gx[r][0]);
        const float gf = act_sigmoid<AT>(acc[0][1][r] + gx[r][1]);
        const float gg = act_tanh<AT>(acc[0][2][r] + gx[r][2]);
        const float go = act_sigmoid<AT>(acc[0][3][r] + gx[r][3]);
        const float c = gf * cprev[r] + gi * gg;
        const float h = go * act_tanh<AT>(c);
        g_row[0] = from_f32<AT>(gi);
        g_row[Hp] = from_f32<AT>(gf);
        g_row[2 * Hp] = from_f32<AT>(gg);
        g_row[3 * Hp] = from_f32<AT>(go);
        a.c_out[(size_t)b * Hp + j] = c;
        c_st[(size_t)b * Hp + j] = from_f32<AT>(c);
        h_out[(size_t)b * Hp + j] = from_f32<AT>(h);
    }
#ifdef PL_STAMPS
    if (a.stamps && tid == 0) {
        st[4] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 5; ++i) o[i] = st[i];
        o[5] = __builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID, bits 0..3 */);   // which XCD ran this block
    }
#endif
}

// ------------------------------------------------------------------------------------------
// backward: workgroup = 32 batch rows x 32 hidden units, K = 4*Hp split over the four waves
// (each wave takes every 4th k-step of the whole 32 x 32 tile: half the LDS fragment traffic of a
// 2 x 2 wave grid), partial tiles reduced through LDS; thread (row, 4 columns) then runs the cell
// backward on 4 consecutive hidden units with vector loads / stores.  grid = (Hp / 32, ceil(Bp / 32)).
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmStepArgs a) {
    using RG = RingGemm<AT, 32, 32, 512, 4, true>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RG::LDS_BYTES];
    const int Hp = a.Hp, Bp = a.Bp, G4 = 4 * Hp;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const AT* __restrict__ WT = static_cast<const AT*>(a.W);
    const AT* __restrict__ dAn = static_cast<const AT*>(a.G_next);
    AT* __restrict__ G = static_cast<AT*>(a.G_t);
    const AT* __restrict__ c_st = static_cast<const AT*>(a.c_stash_t);
    const AT* __restrict__ c_pv = static_cast<const AT*>(a.c_stash_prev);
    const AT* __restrict__ dhe = static_cast<const AT*>(a.dh_ext);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PL_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st[0] = __builtin_amdgcn_s_memrealtime();
#else
    unsigned long long* st = nullptr;
#endif

    // this thread's cells: batch row b, hidden units j..j+3
    const int row = tid >> 3, col4 = (tid & 7) * 4;
    const int b = b0 + row, j = j0 + col4;
    const bool ok = b < Bp;
    const size_t bj = (size_t)(ok ? b : 0) * Hp + j;
    AT* g_row = G + (size_t)(ok ? b : 0) * G4 + j;
    const float4 gi = Vec4<AT>::load(g_row), gf = Vec4<AT>::load(g_row + Hp);
    const float4 gg = Vec4<AT>::load(g_row + 2 * Hp), go = Vec4<AT>::load(g_row + 3 * Hp);
    const float4 c = Vec4<AT>::load(c_st + bj);
    const float4 cprev = c_pv ? Vec4<AT>::load(c_pv + bj) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 dhx = dhe ? Vec4<AT>::load(dhe + bj) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 dcn = a.c_in ? *reinterpret_cast<const float4*>(a.c_in + bj) : make_float4(0.f, 0.f, 0.f, 0.f);

    float4 dh = dhx;
    if (dAn) {
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto arow = [&](int r) -> const AT* { return dAn + (size_t)(b0 + r < Bp ? b0 + r : 0) * G4; };
        auto wrow = [&](int r) -> const AT* { return WT + (size_t)(j0 + r) * G4; };
        RG::run(arow, wrow, G4 * (int)sizeof(AT), acc, lds, st);
        // reduce the four k-partials: red[wave][32][36] floats (row stride padded), ring memory is free now
        float* red = reinterpret_cast<float*>(lds);
        constexpr int LDR = 36;
        const int lr = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(wave * 32 + i * 16 + kq * 4 + r) * LDR + jj * 16 + lr] = acc[i][jj][r];
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 p = *reinterpret_cast<const float4*>(red + (w * 32 + row) * LDR + col4);
            dh.x += p.x; dh.y += p.y; dh.z += p.z; dh.w += p.w;
        }
    }
    float4 dai, daf, dag, dao, dco;
#define PL_CELL_BWD(X)                                                        \
    {                                                                         \
        const float tc = act_tanh<AT>(c.X);                                   \
        const float dc = dcn.X + dh.X * go.X * (1.f - tc * tc);               \
        dai.X = dc * gg.X * gi.X * (1.f - gi.X);                              \
        daf.X = dc * cprev.X * gf.X * (1.f - gf.X);                           \
        dag.X = dc * gi.X * (1.f - gg.X * gg.X);                              \
        dao.X = dh.X * tc * go.X * (1.f - go.X);                              \
        dco.X = dc * gf.X;                                                    \
    }
    PL_CELL_BWD(x) PL_CELL_BWD(y) PL_CELL_BWD(z) PL_CELL_BWD(w)
#undef PL_CELL_BWD
    if (ok) {
        Vec4<AT>::store(g_row, dai);
        Vec4<AT>::store(g_row + Hp, daf);
        Vec4<AT>::store(g_row + 2 * Hp, dag);
        Vec4<AT>::store(g_row + 3 * Hp, dao);
        *reinterpret_cast<float4*>(a.c_out + bj) = dco;
    }
#ifdef PL_STAMPS
    if (a.stamps && tid == 0) {
        st[4] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = a.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 5; ++i) o[i] = st[i];
        o[5] = __builtin_amdgcn_s_getreg(6164);
    }
#endif
}

void launch_lstm_fwd_step(hipStream_t stream, int dt, const LstmStepArgs& a) {
    dim3 grid(a.Hp / 16, (a.Bp + 63) / 64);
    if (dt == BF16)
        hipLaunchKernelGGL(lstm_fwd_step_kernel<bf16_t>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(lstm_fwd_step_kernel<float>, grid, dim3(256), 0, stream, a);
}

void launch_lstm_bwd_step(hipStream_t stream, int dt, const LstmStepArgs& a) {
    dim3 grid(a.Hp / 32, (a.Bp + 31) / 32);
    if (dt == BF16)
        hipLaunchKernelGGL(lstm_bwd_step_kernel<bf16_t>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(lstm_bwd_step_kernel<float>, grid, dim3(256), 0, stream, a);
}

}  // namespace pl
