// LSTM time-step kernels (torch.nn.LSTM semantics used at paule/models.py:344, :431:
// gate order i,f,g,o; i,f,o = sigmoid, g = tanh; c' = f*c + i*g; h' = o*tanh(c'); h0 = c0 = 0).
//
// One launch = one time step of one layer for the whole batch slab:
//   forward : a = Gx_t + h_{t-1} * Whh^T          -> gates (stash), c_t, h_t
//   backward: dh = dh_ext_t + dA_{t+1} * Whh       -> dA_t (backward-DATA only; no dW)
// The recurrent product runs on the MFMA tile engine with the cell arithmetic fused into the
// epilogue, so the gate pre-activations never leave registers.
//
// Data layout (time-major slabs, every feature dim padded to 32 with zeros):
//   G   [Bp][4*Hp]  (gate-blocked: column g*Hp + j),  h / c stash [Bp][Hp],  running c / dc f32 [Bp][Hp]
#include "kernels.h"
#include "tile_gemm.h"

namespace pl {

// ------------------------------------------------------------------------------------------
// forward: workgroup = 64 batch rows x 16 hidden units x 4 gates; wave w owns batch rows 16w..16w+15
// and all four gates of the 16 hidden units (accumulator tile j = gate j), so the cell update is
// lane-local.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmStepArgs a) {
    using TG = TileGemm<AT, 64, 64, 16, 64, 256>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[TG::LDS_BYTES];
    const int Hp = a.Hp, Bp = a.Bp, G4 = 4 * Hp;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 64;
    const AT* __restrict__ W = static_cast<const AT*>(a.W);
    const AT* __restrict__ hp = static_cast<const AT*>(a.h_prev);

    f32x4 acc[1][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[0][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto arow = [&](int r) -> const AT* { return (hp && b0 + r < Bp) ? hp + (size_t)(b0 + r) * Hp : nullptr; };
    auto wrow = [&](int r) -> const AT* { return W + (size_t)((r >> 4) * Hp + j0 + (r & 15)) * Hp; };
    TG::run(arow, wrow, hp ? Hp : 0, acc, lds);

    const auto cd = TG::coord();
    AT* __restrict__ G = static_cast<AT*>(a.G_t);
    AT* __restrict__ h_out = static_cast<AT*>(a.h_out);
    AT* __restrict__ c_st = static_cast<AT*>(a.c_stash_t);
    const int j = j0 + cd.lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + cd.m(0, r);
        if (b >= Bp) continue;
        AT* g_row = G + (size_t)b * G4 + j;
        const float ai = acc[0][0][r] + to_f32<AT>(g_row[0]);
        const float af = acc[0][1][r] + to_f32<AT>(g_row[Hp]);
        const float ag = acc[0][2][r] + to_f32<AT>(g_row[2 * Hp]);
        const float ao = acc[0][3][r] + to_f32<AT>(g_row[3 * Hp]);
        const float gi = sigmoid_f(ai), gf = sigmoid_f(af), gg = tanhf(ag), go = sigmoid_f(ao);
        const float cp = a.c_in ? a.c_in[(size_t)b * Hp + j] : 0.f;
        const float c = gf * cp + gi * gg;
        const float h = go * tanhf(c);
        g_row[0] = from_f32<AT>(gi);
        g_row[Hp] = from_f32<AT>(gf);
        g_row[2 * Hp] = from_f32<AT>(gg);
        g_row[3 * Hp] = from_f32<AT>(go);
        a.c_out[(size_t)b * Hp + j] = c;
        c_st[(size_t)b * Hp + j] = from_f32<AT>(c);
        h_out[(size_t)b * Hp + j] = from_f32<AT>(h);
    }
}

// ------------------------------------------------------------------------------------------
// backward: workgroup = 32 batch rows x 32 hidden units (2 x 2 waves of 16 x 16), K = 4*Hp.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmStepArgs a) {
    using TG = TileGemm<AT, 32, 32, 16, 16, 512>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[TG::LDS_BYTES];
    const int Hp = a.Hp, Bp = a.Bp, G4 = 4 * Hp;
    const int j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const AT* __restrict__ WT = static_cast<const AT*>(a.W);
    const AT* __restrict__ dAn = static_cast<const AT*>(a.G_next);

    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto arow = [&](int r) -> const AT* { return (dAn && b0 + r < Bp) ? dAn + (size_t)(b0 + r) * G4 : nullptr; };
    auto wrow = [&](int r) -> const AT* { return (j0 + r < Hp) ? WT + (size_t)(j0 + r) * G4 : nullptr; };
    TG::run(arow, wrow, dAn ? G4 : 0, acc, lds);

    const auto cd = TG::coord();
    AT* __restrict__ G = static_cast<AT*>(a.G_t);
    const AT* __restrict__ c_st = static_cast<const AT*>(a.c_stash_t);
    const AT* __restrict__ c_pv = static_cast<const AT*>(a.c_stash_prev);
    const AT* __restrict__ dhe = static_cast<const AT*>(a.dh_ext);
    const int j = j0 + cd.n(0);
    if (j >= Hp) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + cd.m(0, r);
        if (b >= Bp) continue;
        const size_t bj = (size_t)b * Hp + j;
        AT* g_row = G + (size_t)b * G4 + j;
        const float gi = to_f32<AT>(g_row[0]), gf = to_f32<AT>(g_row[Hp]);
        const float gg = to_f32<AT>(g_row[2 * Hp]), go = to_f32<AT>(g_row[3 * Hp]);
        const float c = to_f32<AT>(c_st[bj]);
        const float cprev = c_pv ? to_f32<AT>(c_pv[bj]) : 0.f;
        const float tc = tanhf(c);
        const float dh = acc[0][0][r] + (dhe ? to_f32<AT>(dhe[bj]) : 0.f);
        const float dc = (a.c_in ? a.c_in[bj] : 0.f) + dh * go * (1.f - tc * tc);
        g_row[0] = from_f32<AT>(dc * gg * gi * (1.f - gi));
        g_row[Hp] = from_f32<AT>(dc * cprev * gf * (1.f - gf));
        g_row[2 * Hp] = from_f32<AT>(dc * gi * (1.f - gg * gg));
        g_row[3 * Hp] = from_f32<AT>(dh * tc * go * (1.f - go));
        a.c_out[bj] = dc * gf;
    }
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
