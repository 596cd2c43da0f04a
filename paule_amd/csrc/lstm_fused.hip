// The acoustic path's LSTM sweeps of one direction as ONE persistent launch (bf16).
//
// Why.  A layer's recurrence is a chain of T dependent steps, and a step is mostly latency: of the 4.1 (forward) / 5.3 us
// (backward) a step of the per-layer sweeps (lstm_persist.hip) takes at B = 256, about 1.7 / 2.2 us are MFMA and cell work
// of the workgroup, the rest is the group's exchange (store drain, flag, poll, tile read-back).  Run one layer after the
// other, cfg3 is 1 200 such steps forward and 1 200 backward with the CUs idle more than half of the time.  Two changes:
//
//  * chains: a workgroup serves C batch groups of 32 rows with ONE copy of its W_hh slice in registers and works on them in
//    turn -- while one group's hand-off is in flight the next group's operands (prefetched into registers one chain-step
//    ahead) multiply.  A sweep of 8 groups then needs 4 x 23 instead of 8 x 23 CUs at the same time per step;
//  * roles: the CUs that frees carry the OTHER layers at the same time.  The workgroups of one launch take roles from a
//    host-built table (blockIdx -> role, set of groups, slice): the predictor's recurrence, the mel head (post_linear +
//    pooling), the embedder's first layer, the input projection of its second layer, the second layer.  A role runs all
//    its time steps; what it needs from another role it waits for per (group, step) on that role's arrival flags.  The
//    dependent chain of an iteration shrinks from the sum of the layers' steps to the longest layer's.
//
// Hand-off protocol: as in lstm_persist.hip (MI355X guide, hand-off table row 1): handed-off bytes are stored write-through
// (sc1), every storing wave drains them (s_waitcnt vmcnt), the workgroup barriers, ONE lane raises the flag (sc1); a
// consumer polls with sc1 loads from ONE wave, barriers, and every load of handed-off bytes is an sc1 load.  Flags only go
// 0 -> 1 inside a launch and are zeroed by a kernel of ours before it.  All waits are bounded (status word, then every
// workgroup leaves).  Nothing depends on placement or dispatch order: the role graph is acyclic and every role walks its
// steps in increasing time, so a resident grid (<= one workgroup per CU, checked by the host) always drains.
//
// Arithmetic is that of the per-layer path, instruction for instruction (same MFMA shapes and k order per output element):
// the forward results are bit-identical to it (tests/test_hip_parity.py::test_fused_forward_is_bit_identical).
#include "fused_common.h"
#include "lstm_fused16.h"

#ifndef FUSED_BWD_ANSWER_AT
#define FUSED_BWD_ANSWER_AT 1   // backward recurrence: next chain-step's flag answer + prefetch issue 0 before / 1 half way through / 2 after the tiles
#endif
#ifndef FUSED_POLL_NUM
#define FUSED_POLL_NUM 4   // wave 0's look at the next chain-step's flags after FUSED_POLL_NUM / 8 of the MFMA chain (answered after the chain)
#endif
#ifndef FUSED_POLL_EARLY
#define FUSED_POLL_EARLY -1   // wave 0's EARLY look after FUSED_POLL_EARLY / 8 of the MFMA chain: if it finds the flags up, the late look's answer is not waited for (-1: off, the default: measured slower)
#endif

#ifndef FUSED_PASSES
#define FUSED_PASSES 0   // 1: fused_fwd_kernel's pass loop is compiled in (PAULE_HIP_FUSED_GPP then switches it on).  Off by default (round 5): the launch in passes
                         // was measured slower than the per-layer sweeps where it applies (NOTEBOOK.md A.10), and the role descriptor held live across the pass
                         // loop costs every role of every launch scalar registers (B = 128 x 300: 3.47 -> 3.44 ms, 128 x 2000: 22.29 -> 22.17 without it)
#endif

namespace pl {

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// forward recurrence of one layer (arithmetic of lstm_fwd_sweep_kernel)
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int KSX>
struct LstmFwdLds {
    static constexpr int Hp = 16 * KS;
    static constexpr int RS = Hp * 2 + 16;        // h image row stride: odd number of 16-byte chunks -> conflict-free b128 reads
    static constexpr int HRS = 64 + 16;           // outgoing tiles [32 rows][32 units] bf16
    static constexpr int XRS = KSX * 32 + 16;
    static constexpr int O_HIMG = 0;
    static constexpr int O_HST = O_HIMG + 32 * RS;
    static constexpr int O_XIMG = O_HST + 6 * 32 * HRS;
    static constexpr int GRS = 4 * 64 + 16;       // KSX = 0: the workgroup's projection rows [32 rows][4 gates x 32 units] bf16
    static constexpr int O_CST = O_XIMG + (KSX ? 32 * XRS : 32 * GRS);
    static constexpr int O_FLAG = O_CST + kFusedMaxChains * 256 * 16;
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, int KSX>
__device__ __forceinline__ void fused_lstm_fwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmFwdLds<KS, KSX>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp;
    constexpr int ROWB = Hp * 2, RS = L::RS, HRS = L::HRS, XRS = L::XRS;
    constexpr int CH = Hp / 8;                        // 16-byte chunks per h row
    constexpr int NL = (32 * CH + 255) / 256;         // tile loads per thread
    constexpr int PF = 6;                             // B-fragment read-ahead
    constexpr int PK = KS * FUSED_POLL_NUM / 8;       // k-step at which wave 0 takes its look at the next chain-step's flags
    // a poll's answer takes ~1.2 us (write-through flags: the memory side answers), as long as the whole MFMA chain: the look at PK is
    // late enough to see flags raised at the top of this chain-step but is answered ~0.6 us after the chain.  An EARLY look at k-step
    // PK0 has its answer by the end of the chain; only if that one fails is the late one waited for.  (Issuing the operand loads
    // from inside the chain on an early answer was measured: the wait for the answer then sits in the chain, +0.75 us per chain-step.)
    constexpr int PK0 = FUSED_POLL_EARLY >= 0 && KS >= 16 ? KS * FUSED_POLL_EARLY / 8 : -1;
    constexpr int INP = KSX ? 16 * KSX : 16, XC = INP / 8;
    unsigned char* himg = lds + L::O_HIMG;
    unsigned char* hst = lds + L::O_HST;
    unsigned char* ximg = lds + L::O_XIMG;
    float4* cst = reinterpret_cast<float4*>(lds + L::O_CST);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);   // wave-uniform, and the compiler should know
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(R.W);

    // weights -> registers: A-operand row (lane & 31) = gate (row >> 3), unit 32p + 8 wave + (row & 7)
    uint4 wreg[KS];
    {
        const int ar = lane & 31;
        const bf16_t* wrow = W + (size_t)((ar >> 3) * Hp + 32 * p + 8 * wave + (ar & 7)) * Hp + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { wreg[ks] = gld<uint4>(wrow + 16 * ks); pin(wreg[ks]); }
    }
    const int bl = lane & 31, hh = lane >> 5;
    uint4 wx[KSX ? KSX : 1];
    float bias_r[16];
    if constexpr (KSX > 0) {
        const int ar = lane & 31;
        const bf16_t* xrow = static_cast<const bf16_t*>(R.Wih) + (size_t)((ar >> 3) * Hp + 32 * p + 8 * wave + (ar & 7)) * INP + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) { wx[ks] = gld<uint4>(xrow + 16 * ks); pin(wx[ks]); }
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = gld<float>(R.bias + (r >> 2) * Hp + 32 * p + 8 * wave + 4 * hh + (r & 3));
    }
    const int j = 32 * p + 8 * wave + 4 * hh;     // this lane's 4 hidden units
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G);
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(R.h);
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(R.c);
    const bool src_sc1 = R.src_sc1 != 0;   // x / G rows come from a role of this launch: write-through loads
    const bf16_t* const x_in = static_cast<const bf16_t*>(R.x_in);

    // operands of the NEXT chain-step, in flight while the current one computes
    uint4 hv[NL];
    uint4 xv = make_uint4(0, 0, 0, 0);
    uint2 gxn[4] = {};
    uint4 gv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};   // KSX = 0, rows from a role of this launch: 16-byte pieces, through LDS
    auto issue_loads = [&](int g2, int t2) {
        // the small x / projection rows first: vmcnt retires in order, and the cell update waits for them, not for the h tile
        if constexpr (KSX > 0) {
            if (wave < XC / 2) {   // 32 x XC threads = XC / 2 whole waves: a scalar branch
                const int row = tid / XC, cc = tid % XC;
                int rb = 32 * g2 + row;
                rb = rb < Bp ? rb : Bp - 1;
                const size_t eo = ((size_t)t2 * Bp + rb) * INP + cc * 8;
                if (src_sc1) {
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x_in + (size_t)t2 * Bp * INP, (unsigned)((size_t)Bp * INP * 2));
                    xv = ld16_sc1(rx, (unsigned)((rb * INP + cc * 8) * 2));
                } else {
                    xv = gld<uint4>(x_in + eo);
                }
            }
        } else {
            int b2 = 32 * g2 + bl;
            b2 = b2 < Bp ? b2 : Bp - 1;
            if (src_sc1) {
                // the slice's 32 rows x 4 gates x 64 bytes as 512 pieces of 16 bytes (4 lanes per 64-byte run): read lane by lane as
                // 8-byte cells they were 32 separate segments per wave instruction and cost as much memory-pipe time as the h tile
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t2 * slabG, (unsigned)(slabG * 2));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                    const int rb = 32 * g2 + row;
                    gv[q] = ld16_sc1(rg, rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2) : kOob);
                }
            } else {
                const bf16_t* g_row = G + (size_t)t2 * slabG + (size_t)b2 * G4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) gxn[q] = gld<uint2>(g_row + q * Hp);
            }
        }
        if (t2 > 0) {
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t2 - 1) * slabH, (unsigned)(slabH * 2));
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int q = tid + 256 * i;
                const int row = q / CH, ch = q % CH;
                const int rb = 32 * g2 + row;
                hv[i] = ld16_sc1(rh, (q < 32 * CH && rb < Bp) ? (unsigned)(rb * ROWB + ch * 16) : kOob);
            }
        }
    };

    PL_ST_DECL
    int c = 0, t = 0;
    {   // operands of chain-step (0, 0)
        const FlagPoll s0 = step_flags(a, WT_, set * RC, 0, p);
        if (!poll_empty(s0) && !flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        issue_loads(set * RC, 0);
    }
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t + 1; }
        const bool has_next = tn < T;
        const int gn = set * RC + cn;
        const int b = 32 * g + bl;
        const bool ok = b < Bp;

        // A. the prefetched operands of this chain-step -> LDS (the loads were issued one chain-step ago)
        // the projection rows of this chain-step, widened NOW: the registers they arrived in are free for the next prefetch, and the
        // cell update below does not have to wait for those loads (it did when it unpacked them after the prefetch issue)
        float gxi[4], gxf[4], gxg[4], gxo[4];
        if constexpr (KSX == 0) {
            if (src_sc1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                    *reinterpret_cast<uint4*>(ximg + row * L::GRS + gate * 64 + q4 * 16) = gv[q];
                }
            }
        }
        if (t > 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int q = tid + 256 * i;
                if (wave < 2 * KS - 4 * i) *reinterpret_cast<uint4*>(himg + (q / CH) * RS + (q % CH) * 16) = hv[i];   // 32 CH = 64 x 2 KS chunks: whole waves
            }
        }
        if constexpr (KSX > 0) {
            if (wave < XC / 2) *reinterpret_cast<uint4*>(ximg + (tid / XC) * XRS + (tid % XC) * 16) = xv;
        }
        __syncthreads();
        if constexpr (KSX == 0) {
            if (src_sc1) {
                const unsigned char* gsrc = ximg + bl * L::GRS + (8 * wave + 4 * hh) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) gxn[q] = *reinterpret_cast<const uint2*>(gsrc + q * 64);
            }
        }
        unpack_bf16x4(gxn[0], gxi);
        unpack_bf16x4(gxn[1], gxf);
        unpack_bf16x4(gxn[2], gxg);
        unpack_bf16x4(gxn[3], gxo);
        if constexpr (KSX == 0) {   // pin the widened values before the prefetch below reuses registers
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(gxi[u]), "+v"(gxf[u]), "+v"(gxg[u]), "+v"(gxo[u]));
        }
        PL_ST(0);   // operands landed + LDS image

        // B. first look at the next chain-step's flags, answered while the MFMAs run
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = step_flags(a, WT_, gn, tn, p);
        int pv = 1, pv0 = 0;
        const bool poll_here = wave == 0 && has_next;
        if (poll_here && t == 0) pv = poll_load(pn, lane);
        __builtin_amdgcn_sched_barrier(0);

        // C. gates = W_hh h_{t-1} (+ W_ih x_t + b)
        f32x16 acc;
        if constexpr (KSX > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias_r[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        if (t > 0) {
            const unsigned char* bsrc = himg + bl * RS + hh * 16;
            uint4 bq[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < KS) bq[i] = *reinterpret_cast<const uint4*>(bsrc + i * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks]), __builtin_bit_cast(bf16x8, bq[ks % PF]), acc, 0, 0, 0);
                if (ks + PF < KS) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 32);
                if (PK0 >= 0 && ks == PK0 && poll_here) pv0 = poll_load(pn, lane);
                if (ks == PK && poll_here) pv = poll_load(pn, lane);   // late enough for flags raised at the top of this chain-step
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (KSX > 0) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                const uint4 xb = *reinterpret_cast<const uint4*>(ximg + bl * XRS + ks * 32 + hh * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wx[ks]), __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
            }
        }
        PL_ST(1);   // MFMA chain

        // D. has the next chain-step everything it waits for?  (uniform answer through LDS)
        if (wave == 0) {
            bool rdy = PK0 >= 0 && __all(pv0 != 0);
            if (!rdy) rdy = __all(pv != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }
        __syncthreads();
        const bool ready = has_next && lflag[0] != 0;
        // E. then its operands start now and fly under the cell update and the stores of this one
        if (ready) issue_loads(gn, tn);
        PL_ST(2);   // flag answer + prefetch issue

        // F. cell update: acc[4 * gate + unit]
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t > 0) cs = cst[c * 256 + tid];
        float c_state[4] = {cs.x, cs.y, cs.z, cs.w};
        float vi[4], vf[4], vg[4], vo[4], vc[4], vh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            vi[u] = sigmoid_fast(acc[u] + gxi[u]);
            vf[u] = sigmoid_fast(acc[4 + u] + gxf[u]);
            vg[u] = tanh_fast(acc[8 + u] + gxg[u]);
            vo[u] = sigmoid_fast(acc[12 + u] + gxo[u]);
            c_state[u] = cell_c(vf[u], c_state[u], vi[u], vg[u]);
            vc[u] = c_state[u];
            vh[u] = vo[u] * tanh_fast(vc[u]);
        }
        cst[c * 256 + tid] = make_float4(c_state[0], c_state[1], c_state[2], c_state[3]);
        // G. the h tile (hand-off) and the five stash arrays leave through LDS as whole 64-byte row pieces
        {
            unsigned char* o = hst + bl * HRS + (8 * wave + 4 * hh) * 2;
            *reinterpret_cast<uint2*>(o) = pack_bf16x4(vh[0], vh[1], vh[2], vh[3]);
            *reinterpret_cast<uint2*>(o + 32 * HRS) = pack_bf16x4(vi[0], vi[1], vi[2], vi[3]);
            *reinterpret_cast<uint2*>(o + 2 * 32 * HRS) = pack_bf16x4(vf[0], vf[1], vf[2], vf[3]);
            *reinterpret_cast<uint2*>(o + 3 * 32 * HRS) = pack_bf16x4(vg[0], vg[1], vg[2], vg[3]);
            *reinterpret_cast<uint2*>(o + 4 * 32 * HRS) = pack_bf16x4(vo[0], vo[1], vo[2], vo[3]);
            *reinterpret_cast<uint2*>(o + 5 * 32 * HRS) = pack_bf16x4(vc[0], vc[1], vc[2], vc[3]);
        }
        (void)ok;
        __syncthreads();
        if (wave < 2) {
            const int row = tid >> 2, qt = tid & 3;
            const int rb = 32 * g + row;
            const uint4 hvv = *reinterpret_cast<const uint4*>(hst + row * HRS + qt * 16);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
            st16_sc1(ro, rb < Bp ? (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2) : kOob, hvv);
        }
        // One chain per workgroup: the five stash arrays leave BEHIND the flag (nobody inside the launch waits for them, and nothing else of
        // this workgroup hides their issue: B = 64 3.30 -> 3.25 ms per iteration, 128 x 2000 frames 22.49 -> 22.35); with more chains the
        // next chain's operands are already waiting, and stores in front of its MFMAs cost more than they save behind this chain's flag
        // (B = 192, chains 2 / 3: 4.65 -> 4.73): there they stay in front (profiles/r04_ab_fused_occ2.txt, item 11)
        const bool stash_late = RC == 1;
        auto stash_stores = [&]() {
            asm volatile("" ::: "memory");   // keep the stash stores behind the hand-off store
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = tid + 256 * i;   // piece: array e / 128, row (e % 128) / 4, quarter e % 4
                if (i < 2 || wave < 2) {   // 640 pieces: all threads twice, waves 0 and 1 a third time
                    const int arr = e >> 7, row = (e & 127) >> 2, qt = e & 3, rb = 32 * g + row;
                    const uint4 sv = *reinterpret_cast<const uint4*>(hst + (arr + 1) * 32 * HRS + row * HRS + qt * 16);
                    u32x4 d;
                    d[0] = sv.x; d[1] = sv.y; d[2] = sv.z; d[3] = sv.w;
                    if (i < 2) {   // pieces 0 .. 511: the four gate arrays (128 pieces = 2 waves each); 512 .. 639: c
                        const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
                        __builtin_amdgcn_raw_buffer_store_b128(d, rg, rb < Bp ? (unsigned)(((size_t)rb * G4 + arr * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
                    } else {
                        const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cs + (size_t)t * slabH, (unsigned)(slabH * 2));
                        __builtin_amdgcn_raw_buffer_store_b128(d, rc, rb < Bp ? (unsigned)(((size_t)rb * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
                    }
                }
            }
        };
        if (!stash_late) stash_stores();
        PL_ST(3);   // cell + store issue
        if (stash_late) raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);   // only the hand-off is in flight
        else raise_flag<3>(rflags + ((size_t)g * T + t) * a.flag_stride + p);   // the hand-off store is older than the (at most 3) stash stores
        PL_ST(4);   // drain + barrier + flag
        if (stash_late) stash_stores();

        if (!has_next) break;
        // H. the next chain-step was not ready at the first look (always so with one chain: it waits for the flag just raised)
        if (!ready) {
            if (!flags_wait(pn, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
            issue_loads(gn, tn);
        }
        PL_ST(5);   // blocking wait
        c = cn;
        t = tn;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// row-tile products on a producing layer's h: the input projection of the layer above (PROJ: G_t = h_t Wih^T + b, bf16)
// and the mel head with its pooling (HEAD: mel = avg-pool-2(h_t Wlin^T + b)).  Arithmetic of gemm_nt_kernel (gemm.hip):
// v_mfma_f32_16x16x32_bf16, A = activation rows, B = weight rows, k-step n covers k = 32 n .. 32 n + 31, accumulators start
// at zero, bias added at the end -- so the outputs carry the same bits as the batched GEMM's.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct GemmFwdLds {
    static constexpr int Hp = 16 * KS;
    static constexpr int CH = Hp / 8;
    static constexpr int IRS = (CH + ((CH % 4 == 0) ? 2 : 0)) * 16;   // chunk stride = 2 mod 4: conflict-free for (row lr, chunk 4n + kq) reads
    static constexpr int ORS = 128 * 2 + 16;
    static constexpr int O_IMG = 0;
    static constexpr int O_OST = O_IMG + 32 * IRS;
    static constexpr int O_YB = O_OST + 32 * ORS;
    static constexpr int O_FLAG = O_YB + kFusedMaxChains * 256 * 32;
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, bool HEAD, bool G16 = false>   // G16: the role it waits for is a 16-row LSTM role (one flag row per 16-row group)
__device__ __forceinline__ void fused_gemm_fwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = GemmFwdLds<KS>;
    constexpr int Hp = 16 * KS, KB = KS / 2, CH = L::CH, IRS = L::IRS, ORS = L::ORS, ROWB = Hp * 2;
    constexpr int NL = (32 * CH + 255) / 256;
    constexpr int NJ = HEAD ? 1 : 2;
    constexpr int PF = 4;
    unsigned char* img = lds + L::O_IMG;
    unsigned char* ost = lds + L::O_OST;
    float4* yb = reinterpret_cast<float4*>(lds + L::O_YB);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);   // wave-uniform, and the compiler should know
    const int lr = lane & 15, kq = lane >> 4;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const int G4 = 4 * Hp;   // PROJ: gate columns of the consuming layer (same hidden size)
    const bf16_t* __restrict__ Wg = static_cast<const bf16_t*>(R.Wg);
    uint4 wreg[NJ][KB];
    float bias_v[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int col = HEAD ? 16 * wave + lr : wave * Hp + 32 * p + 16 * jj + lr;
        const bf16_t* wrow = Wg + (size_t)col * Hp + 8 * kq;
#pragma unroll
        for (int n = 0; n < KB; ++n) { wreg[jj][n] = gld<uint4>(wrow + 32 * n); pin(wreg[jj][n]); }
        bias_v[jj] = R.bias ? gld<float>(R.bias + col) : 0.f;
    }
    const size_t slabH = (size_t)Bp * Hp;
    const bf16_t* __restrict__ Hsrc = static_cast<const bf16_t*>(R.src_h);
    const int out_dim = R.out_dim;
    float* const out_bm = R.out_bm;
    void* const out_ptr = R.out;

    uint4 hv[NL];
    auto issue_loads = [&](int g2, int t2) {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hsrc + (size_t)t2 * slabH, (unsigned)(slabH * 2));
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i;
            const int row = q / CH, ch = q % CH;
            const int rb = 32 * g2 + row;
            hv[i] = ld16_sc1(rh, (q < 32 * CH && rb < Bp) ? (unsigned)(rb * ROWB + ch * 16) : kOob);
        }
    };

    PL_ST_DECL
    int c = 0, t = 0;
    {
        const FlagPoll s0 = (G16 ? step_flags_lstm16_w0(a, WT_, set * RC, 0, p) : step_flags(a, WT_, set * RC, 0, p));
        if (!flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        issue_loads(set * RC, 0);
    }
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t + 1; }
        const bool has_next = tn < T;
        const int gn = set * RC + cn;

#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i;
            if (wave < 2 * KS - 4 * i) *reinterpret_cast<uint4*>(img + (q / CH) * IRS + (q % CH) * 16) = hv[i];   // whole waves
        }
        __syncthreads();
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = (G16 ? step_flags_lstm16_w0(a, WT_, gn, tn, p) : step_flags(a, WT_, gn, tn, p));
        int pv = 1;
        const bool poll_here = wave == 0 && has_next;
        __builtin_amdgcn_sched_barrier(0);
        PL_ST(0);

        f32x4 acc[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const unsigned char* a0 = img + lr * IRS + kq * 16;
            const unsigned char* a1 = img + (16 + lr) * IRS + kq * 16;
            uint4 f0[PF], f1[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < KB) {
                    f0[i] = *reinterpret_cast<const uint4*>(a0 + i * 64);
                    f1[i] = *reinterpret_cast<const uint4*>(a1 + i * 64);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < KB; ++n) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    acc[0][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[n % PF]), __builtin_bit_cast(bf16x8, wreg[jj][n]), acc[0][jj], 0, 0, 0);
                    acc[1][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[n % PF]), __builtin_bit_cast(bf16x8, wreg[jj][n]), acc[1][jj], 0, 0, 0);
                }
                if (n + PF < KB) {
                    f0[n % PF] = *reinterpret_cast<const uint4*>(a0 + (n + PF) * 64);
                    f1[n % PF] = *reinterpret_cast<const uint4*>(a1 + (n + PF) * 64);
                }
                if (n == KB * FUSED_POLL_NUM / 8 && poll_here) pv = poll_load(pn, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PL_ST(1);
        if (wave == 0) {
            const bool rdy = __all(pv != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }
        __syncthreads();
        const bool ready = has_next && lflag[0] != 0;
        if (ready) issue_loads(gn, tn);
        PL_ST(2);

        // epilogue: D[row 16 i + 4 kq + r][column 16 jj + lr (of this wave's columns)]
        if constexpr (!HEAD) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<bf16_t*>(ost + (16 * i + 4 * kq + r) * ORS + (32 * wave + 16 * jj + lr) * 2) = (bf16_t)(acc[i][jj][r] + bias_v[jj]);
            __syncthreads();
            bf16_t* Gout = static_cast<bf16_t*>(out_ptr);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(Gout + (size_t)t * Bp * G4, (unsigned)((size_t)Bp * G4 * 2));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                const int rb = 32 * g + row;
                const uint4 v = *reinterpret_cast<const uint4*>(ost + row * ORS + (32 * gate + 8 * q4) * 2);
                st16_sc1(ro, rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2) : kOob, v);
            }
            raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);
        } else {
            float y[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[4 * i + r] = acc[i][0][r] + bias_v[0];
            if ((t & 1) == 0) {   // even frame: kept for its partner
                yb[(c * 256 + tid) * 2] = make_float4(y[0], y[1], y[2], y[3]);
                yb[(c * 256 + tid) * 2 + 1] = make_float4(y[4], y[5], y[6], y[7]);
                __syncthreads();   // the image is rewritten at the top of the next chain-step
            } else {
                const float4 e0 = yb[(c * 256 + tid) * 2], e1 = yb[(c * 256 + tid) * 2 + 1];
                const float ye[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                const int tp = t >> 1, Tp = T >> 1, col = 16 * wave + lr;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * i + 4 * kq + r, bb = 32 * g + row;
                        const bool live = bb < a.B && col < out_dim;
                        const float v = live ? 0.5f * (ye[4 * i + r] + y[4 * i + r]) : 0.f;
                        const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(out_bm, (unsigned)((size_t)a.B * Tp * out_dim * 4));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rb_, live ? (unsigned)((((size_t)bb * Tp + tp) * out_dim + col) * 4) : kOob, 0, 0);
                        *reinterpret_cast<bf16_t*>(ost + row * ORS + col * 2) = (bf16_t)v;
                    }
                __syncthreads();
                {   // pooled frame, time-major activation [tp][Bp][64]: the input of the embedder's first layer (hand-off)
                    const int row = tid >> 3, q8 = tid & 7, rb = 32 * g + row;
                    bf16_t* Mout = static_cast<bf16_t*>(out_ptr);
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Mout + (size_t)tp * Bp * 64, (unsigned)((size_t)Bp * 64 * 2));
                    const uint4 v = *reinterpret_cast<const uint4*>(ost + row * ORS + q8 * 16);
                    st16_sc1(ro, rb < Bp ? (unsigned)((rb * 64 + 8 * q8) * 2) : kOob, v);
                }
                raise_flag<0>(rflags + ((size_t)g * Tp + tp) * a.flag_stride);
            }
        }
        PL_ST(3);
        if (!has_next) break;
        if (!ready) {
            if (!flags_wait(pn, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
            issue_loads(gn, tn);
        }
        PL_ST(5);
        c = cn;
        t = tn;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward recurrence of one layer, reduce-scatter form (arithmetic of lstm_bwd_rs_sweep_kernel): a workgroup multiplies the dA
// it produced itself with its 128 rows of W_hh^T and hands the P partial 32 x 32 tiles over; it sums the P tiles of its own
// hidden units -- and, below the top layer, the P tiles of dL/dh from the layer above (FR_DX_BWD role of this launch).
// The first layer of the embedder also multiplies its dA with its rows of W_ih^T (K = 128 -> out_p mel columns) and hands
// those partial tiles to the backward mel head.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFusedMaxChainsBwd = 4;   // the backward roles keep their partial tiles in LDS (2 x 46 KB): less room for per-chain state
template <int KS>
struct LstmBwdLds {
    static constexpr int Hp = 16 * KS, P = Hp / 32;
    static constexpr int DRS = 128 * 2 + 16;      // dA^T image: [32 batch rows][128 local gate rows] bf16, odd chunk stride
    static constexpr int TRS = 64 + 16;           // a wave's outgoing tile: [32 batch rows][32 columns] bf16
    static constexpr int MRS = 64 * 2 + 16;       // backward mel head: dL/dY image [32 batch rows][64] bf16
    static constexpr int O_DA = 0;
    static constexpr int O_TST = O_DA + 32 * DRS;                    // [4 waves][32][TRS]
    static constexpr int O_MEL = O_TST + 4 * 32 * TRS;
    static constexpr int O_DC = O_MEL + 32 * MRS;
    static constexpr int O_WM = O_DC + kFusedMaxChainsBwd * 256 * 16;   // W_ih^T rows of the slice for the input-gradient tiles: [64 columns][128 local gate rows]
    static constexpr int O_XR = O_WM + 64 * DRS;                     // the P partial tiles of the recurrence, as they lie in the exchange (LDS-DMA)
    static constexpr int O_XE = O_XR + P * 2048;                     // ... and of dL/dh from the layer above
    static constexpr int O_FLAG = O_XE + P * 2048;
    static constexpr int BYTES = O_FLAG + 64;
};


// one partial tile: 8 k-steps over the workgroup's 128 local gate rows; acc[r] = out[n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][batch lane & 31]
// -> bf16 image [batch][n], read back by rows by the same wave (a wave's LDS operations are ordered) and stored as 2 KB
// plain = true (a scalar): the tile is stored without write-through -- only for a role's OWN exchange once its workgroups have verified
// that they share one XCD (fused_lstm_bwd); everything another role may read stays sc1
__device__ __forceinline__ void partial_tile(const uint4 (&w)[8], const uint4 (&bfr)[8], unsigned char* img_row0, int rs, int col0,
                                             __amdgpu_buffer_rsrc_t ro, unsigned tile_off, int lane, bool plain = false) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ks]), __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
    unsigned char* orow = img_row0 + (lane & 31) * rs + (col0 + 4 * (lane >> 5)) * 2;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
        *reinterpret_cast<uint2*>(orow + rg * 16) = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(img_row0 + r * rs + (col0 + 8 * c4) * 2);
        if (plain) {
            u32x4 d;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            __builtin_amdgcn_raw_buffer_store_b128(d, ro, (unsigned)(cidx * 16), tile_off, 0);
        } else {
            st16_sc1_so(ro, (unsigned)(cidx * 16), tile_off, v);
        }
    }
}

template <int KS, bool EXT, bool MEL>   // EXT: partial tiles of dL/dh from the layer above; MEL: input-gradient tiles for the backward mel head
__device__ __forceinline__ void fused_lstm_bwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmBwdLds<KS>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, P = Hp / 32, NT = (P + 3) / 4;
    constexpr int DRS = L::DRS, TRS = L::TRS;
    constexpr size_t TILE = 32 * 32;
    unsigned char* da_img = lds + L::O_DA;
    unsigned char* xr_img = lds + L::O_XR;
    unsigned char* xe_img = lds + L::O_XE;
    float4* dcs = reinterpret_cast<float4*>(lds + L::O_DC);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);   // wave-uniform, and the compiler should know
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(R.W);   // Whh^T packed [Hp][4 Hp]
    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 32 * (nt < P ? nt : 0) + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
        {
            wreg[i][ks] = gld<uint4>(WT + (size_t)n * G4 + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
            pin(wreg[i][ks]);
        }
    }
    const int n_mel = MEL ? R.out_p / 32 : 0;   // input-gradient tiles (waves 0 .. n_mel-1 take one each)
    unsigned char* wm_img = lds + L::O_WM;
    if constexpr (MEL) {   // these weights live in LDS (16 KB): in registers, beside W_hh^T and two sets of partial tiles in flight, they spill
        const bf16_t* WM = static_cast<const bf16_t*>(R.Wg);   // Wih^T packed [out_p][4 Hp]
        for (int e = tid; e < 32 * n_mel * 16; e += 256) {
            const int n = e >> 4, ks = (e >> 1) & 7, hf = e & 1;
            *reinterpret_cast<uint4*>(wm_img + n * DRS + ks * 32 + hf * 16) =
                gld<uint4>(WM + (size_t)n * G4 + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * hf);
        }
        __syncthreads();
    }

    // cell ownership: thread -> batch row (tid >> 3), hidden units 32p + 4 (tid & 7) .. +3
    const int erow = tid >> 3, jq = tid & 7;
    const int j = 32 * p + 4 * jq;
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G);
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(R.c);
    const bf16_t* __restrict__ dhe = static_cast<const bf16_t*>(R.dh_ext);
    const bf16_t* __restrict__ dhl = static_cast<const bf16_t*>(R.dh_last);
    bf16_t* __restrict__ X = static_cast<bf16_t*>(R.xchg);          // [2][groups][P dest][P src][32][32]
    bf16_t* __restrict__ XE = static_cast<bf16_t*>(R.xchg_ext);     // [ring][groups][P dest][P src][32][32]
    bf16_t* __restrict__ XM = static_cast<bf16_t*>(R.xchg_mel);     // [ring][groups][n_mel][P src][32][32]
    const size_t grp_stride = (size_t)P * P * TILE, slot_stride = (size_t)a.n_groups * grp_stride;
    const size_t mgrp_stride = (size_t)n_mel * P * TILE, mslot_stride = (size_t)a.n_groups * mgrp_stride;
    const bool src_sc1 = R.src_sc1 != 0, dA_sc1 = R.dA_sc1 != 0;
    const int dh_ext_half = R.dh_ext_half, dh_ext_rows = R.dh_ext_rows;

    // The role's OWN exchange through the shared L2 once its P workgroups have found themselves on one XCD (round 4; the verified
    // same-XCD form of sweep_common.h, data path only).  Written through, the tiles drop out of the XCD's L2 and every consumer's
    // LDS-DMA re-reads them from the memory side: 27.9 GB fetched per launch at 128 x 2000 frames against 8 GB of algorithmic
    // reads (profiles/r03_pmc_summary_cfg5_128.txt).  Plain stores keep the lines in the L2 the consumers read through (nt).
    // The flags stay write-through: other roles poll them too, and a tile is acknowledged by that L2 before its flag is even
    // stored.  Placement is never assumed: every workgroup publishes its XCD with its first hand-off (drained with it), the members
    // compare after their first completed wait on each other, and only a set that is whole on one XCD switches -- for good.
    // Same tiles, same order of every sum: bit-identical to the write-through form.
    // The flags of the own exchange follow (a second, PLAIN set beside the write-through one, polled nt): other roles keep polling the
    // write-through set, which every hand-off raises as before.  The members switch at the same chain-step (c = 0, t = T - 2), so the
    // plain set holds the flags of steps <= T - 2 and a wait for step T - 1 stays on the write-through set.
    int* const xtab = R.xtab ? R.xtab + set * 64 : nullptr;
    const long fdelta = (R.fast_flags && R.xtab) ? (long)(R.fast_flags - R.flags) : 0;
    bool fast = false;
    unsigned char* tst = lds + L::O_TST + wave * 32 * TRS;   // this wave's outgoing tile
    const unsigned xr_lds = (unsigned)(uintptr_t)(lds_ptr_t)xr_img, xe_lds = (unsigned)(uintptr_t)(lds_ptr_t)xe_img;
    // operands of the NEXT chain-step: the stash rows in registers, the partial tiles by LDS-DMA (held in registers across the
    // tiles' MFMAs they cost 46 - 92 VGPRs, and the compiler parked them in AGPRs behind waits in the middle of the tiles)
    uint2 sg[4] = {}, sc = make_uint2(0u, 0u), scp = make_uint2(0u, 0u), sdh = make_uint2(0u, 0u);
    auto issue_loads = [&](int g2, int t2) {
        int b2 = 32 * g2 + erow;
        b2 = b2 < Bp ? b2 : Bp - 1;
        const bf16_t* g_row = G + (size_t)t2 * slabG + (size_t)b2 * G4 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) sg[q] = gld<uint2>(g_row + q * Hp);
        sc = gld<uint2>(Cs + (size_t)t2 * slabH + (size_t)b2 * Hp + j);
        scp = make_uint2(0u, 0u);
        if (t2 > 0) scp = gld<uint2>(Cs + (size_t)(t2 - 1) * slabH + (size_t)b2 * Hp + j);
        sdh = make_uint2(0u, 0u);
        if (dhe) {
            const int row = dh_ext_half ? (t2 >> 1) : t2;
            if (row < dh_ext_rows) {
                if (src_sc1) {
                    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dhe + (size_t)row * slabH, (unsigned)(slabH * 2));
                    sdh = ld8_sc1(rd, (unsigned)(((size_t)b2 * Hp + j) * 2));
                } else {
                    sdh = gld<uint2>(dhe + (size_t)row * slabH + (size_t)b2 * Hp + j);
                }
            }
        } else if (dhl && t2 == T - 1) {
            sdh = gld<uint2>(dhl + (size_t)b2 * Hp + j);
        }
        constexpr int NPC = P * 2;   // 1-KB pieces of a destination's P tiles (contiguous in the exchange)
        if (t2 + 1 < T) {   // the P partial tiles of step t2 + 1 that belong to this workgroup's cells
            const unsigned char* xs = reinterpret_cast<const unsigned char*>(X + (size_t)((t2 + 1) & 1) * slot_stride + (size_t)g2 * grp_stride + (size_t)p * P * TILE);
            if (fast) {
#pragma unroll
                for (int k = 0; k < (NPC + 3) / 4; ++k) {
                    const int pc = wave + 4 * k;
                    if (pc < NPC) glds16_nt(uni(xs + pc * 1024), (unsigned)(lane * 16), (unsigned)uni((int)(xr_lds + (unsigned)(pc * 1024))));
                }
            } else {
#pragma unroll
                for (int k = 0; k < (NPC + 3) / 4; ++k) {
                    const int pc = wave + 4 * k;
                    if (pc < NPC) glds16_sc1(uni(xs + pc * 1024), (unsigned)(lane * 16), (unsigned)uni((int)(xr_lds + (unsigned)(pc * 1024))));
                }
            }
        }
        if constexpr (EXT) {   // ... and the P partial tiles of dL/dh_t2 from the layer above
            const unsigned char* xs = reinterpret_cast<const unsigned char*>(XE + (size_t)(t2 % kFusedRing) * slot_stride + (size_t)g2 * grp_stride + (size_t)p * P * TILE);
#pragma unroll
            for (int k = 0; k < (NPC + 3) / 4; ++k) {
                const int pc = wave + 4 * k;
                if (pc < NPC) glds16_sc1(uni(xs + pc * 1024), (unsigned)(lane * 16), (unsigned)uni((int)(xe_lds + (unsigned)(pc * 1024))));
            }
        }
    };

    PL_ST_DECL
    int c = 0, t = T - 1;
    {
        const FlagPoll s0 = step_flags(a, WT_, set * RC, t, p);
        if (!poll_empty(s0) && !flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        issue_loads(set * RC, t);
    }
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t - 1; }
        const bool has_next = tn >= 0;
        const int gn = set * RC + cn;
        const int b = 32 * g + erow;
        const bool ok = b < Bp;

        // first look at the next chain-step's flags, taken where NO store of this wave is in flight: vmcnt counts loads and stores
        // together and in order, so a poll issued behind hand-off stores would only be answered after their write-through round trips
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = step_flags(a, WT_, gn, tn, p);
        if (fast && fdelta != 0 && has_next && tn + 1 <= T - 2 && pn.na > 0) { pn.fa += fdelta; pn.fa_nt = 1; }   // own flags of step tn + 1, plain set
        // A. dL/dh_t of this thread's cells: from above + the partial sums (fixed order).  Every wave's DMA pieces have landed:
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int pvl = 1;
        if (wave == 0 && has_next) pvl = poll_load(pn, lane);   // behind the wait above: that one must not wait for this round trip
        float dh[4];
        unpack_bf16x4(sdh, dh);
        if (t + 1 < T) {
            const unsigned char* src = xr_img + erow * 64 + jq * 8;
#pragma unroll
            for (int s = 0; s < P; ++s) {
                float f[4];
                unpack_bf16x4(*reinterpret_cast<const uint2*>(src + s * 2048), f);
                dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
            }
        }
        if constexpr (EXT) {
            const unsigned char* src = xe_img + erow * 64 + jq * 8;
#pragma unroll
            for (int s = 0; s < P; ++s) {
                float f[4];
                unpack_bf16x4(*reinterpret_cast<const uint2*>(src + s * 2048), f);
                dh[0] += f[0]; dh[1] += f[1]; dh[2] += f[2]; dh[3] += f[3];
            }
        }
        bool early = true;
        if (wave == 0 && has_next) {   // a second look if the first came too early (with two chains the flags were raised a moment ago)
            early = __all(pvl != 0);
            if (!early) pvl = poll_load(pn, lane);
        }
        PL_ST(0);   // operands landed + sums
        // B. cell backward
        float gi[4], gf[4], gg[4], go[4], cc[4], cp[4];
        unpack_bf16x4(sg[0], gi);
        unpack_bf16x4(sg[1], gf);
        unpack_bf16x4(sg[2], gg);
        unpack_bf16x4(sg[3], go);
        unpack_bf16x4(sc, cc);
        unpack_bf16x4(scp, cp);
        float4 dcv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t + 1 < T) dcv = dcs[c * 256 + tid];
        float dc_next[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
        float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], cc[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
        dcs[c * 256 + tid] = make_float4(dc_next[0], dc_next[1], dc_next[2], dc_next[3]);
        const uint2 pi = pack_bf16x4(dai[0], dai[1], dai[2], dai[3]), pf = pack_bf16x4(daf[0], daf[1], daf[2], daf[3]);
        const uint2 pg = pack_bf16x4(dag[0], dag[1], dag[2], dag[3]), po = pack_bf16x4(dao[0], dao[1], dao[2], dao[3]);
        if (wave == 0) {   // the answer, before this wave's first store
            const bool rdy = early || __all(pvl != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }
        if (ok) {   // dA_t overwrites the gate stash in place: read by the dX product after the launch, or by the FR_DX_BWD role in it
            if (dA_sc1) {
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
                const unsigned o = (unsigned)(((size_t)b * G4 + j) * 2);
                st8_sc1(rg, o, pi);
                st8_sc1(rg, o + Hp * 2, pf);
                st8_sc1(rg, o + 2 * Hp * 2, pg);
                st8_sc1(rg, o + 3 * Hp * 2, po);
            } else {
                bf16_t* go_ = G + (size_t)t * slabG + (size_t)b * G4 + j;
                gst<uint2>(go_, pi);
                gst<uint2>(go_ + Hp, pf);
                gst<uint2>(go_ + 2 * Hp, pg);
                gst<uint2>(go_ + 3 * Hp, po);
            }
        }
        {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 32 + unit]
            unsigned char* drow = da_img + erow * DRS + jq * 8;
            *reinterpret_cast<uint2*>(drow) = pi;
            *reinterpret_cast<uint2*>(drow + 64) = pf;
            *reinterpret_cast<uint2*>(drow + 128) = pg;
            *reinterpret_cast<uint2*>(drow + 192) = po;
        }
        const bool xcd_look = xtab && c == 0 && t == T - 2;   // the wait of this chain-step saw every member's first flag: their ids are in place
        if (xcd_look && wave == 1) {
            const int mine = xcc_id_plus1();
            int v = mine;
            if (lane < P) v = flag_load(xtab + lane);
            const bool same = __all(v == mine);
            if (lane == 0) lflag[2] = same ? 1 : 0;
        }
        __syncthreads();
        if (xcd_look) fast = uni(lflag[2]) != 0;
        PL_ST(1);   // cell + stash stores + dA image
        uint4 bfr[8];   // B fragments of the dA image, shared by all tiles of the wave
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bfr[ks] = *reinterpret_cast<const uint4*>(da_img + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
        // the next chain-step's operands start now (older than every hand-off store below: the drain before the flag covers them,
        // and by then the tiles' MFMAs have run beside their flight)
        const bool ready = has_next && lflag[0] != 0;
        if (ready) issue_loads(gn, tn);
        PL_ST(6);   // prefetch issue
        // C. partial tiles.  Nobody consumes the recurrence's partials of step 0.
        if (t > 0) {
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int nt = wave + 4 * i;
                if (4 * i + 3 < P || nt < P) partial_tile(wreg[i], bfr, tst, TRS, 0, ro, (unsigned)((size_t)nt * P * TILE * 2), lane, fast);
            }
        }
        if (xtab && c == 0 && t == T - 1 && tid == 0) flag_store(xtab + p, xcc_id_plus1());   // with the first hand-off: drained before its flag
        if constexpr (MEL) if (wave < n_mel) {
            bf16_t* xd = XM + (size_t)(t % kFusedRing) * mslot_stride + (size_t)g * mgrp_stride + (size_t)p * TILE;   // [tile][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(n_mel - 1) * P + 1) * TILE * 2));
            uint4 wmel[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wmel[ks] = *reinterpret_cast<const uint4*>(wm_img + (32 * wave + (lane & 31)) * DRS + ks * 32 + (lane >> 5) * 16);
            partial_tile(wmel, bfr, tst, TRS, 0, ro, (unsigned)((size_t)wave * P * TILE * 2), lane);
        }
        PL_ST(2);   // MFMA + hand-off stores (+ flag answer, prefetch issue)
        raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);
        if (fast && fdelta != 0 && wave == 0 && lane == 0) flag_store_plain(rflags + ((size_t)g * T + t) * a.flag_stride + p + fdelta, 1);
        PL_ST(3);   // drain + barrier + flag
        if (!has_next) break;
        if (!ready) {
            if (!flags_wait(pn, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
            issue_loads(gn, tn);
        }
        PL_ST(5);   // blocking wait
        c = cn;
        t = tn;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// dL/dh of the layer below = dA W_ih of the layer above, reduce-scatter form like the recurrence: workgroup p takes the
// dA slice its partner (slice p of the layer above's recurrence) has just written, multiplies it with its 128 rows of
// W_ih^T and hands the P partial tiles to the layer below (ring of kFusedRing steps; a slot is rewritten only after every
// consumer has raised its flag of the step that used it).
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, bool G16 = false>
__device__ __forceinline__ void fused_dx_bwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmBwdLds<KS>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, P = Hp / 32, NT = (P + 3) / 4;
    constexpr int DRS = L::DRS, TRS = L::TRS;
    constexpr size_t TILE = 32 * 32;
    unsigned char* da_img = lds + L::O_DA;
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);   // wave-uniform, and the compiler should know
    unsigned char* tst = lds + L::O_TST + wave * 32 * TRS;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(R.Wg);   // Wih^T of the layer above, packed [Hp][4 Hp]
    uint4 wreg[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 32 * (nt < P ? nt : 0) + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
        {
            wreg[i][ks] = gld<uint4>(WT + (size_t)n * G4 + (ks >> 1) * Hp + 32 * p + 16 * (ks & 1) + 8 * (lane >> 5));
            pin(wreg[i][ks]);
        }
    }
    const size_t slabG = (size_t)Bp * G4;
    const bf16_t* __restrict__ G = static_cast<const bf16_t*>(R.G);   // dA stash of the layer above
    bf16_t* __restrict__ XE = static_cast<bf16_t*>(R.xchg_ext);
    const size_t grp_stride = (size_t)P * P * TILE, slot_stride = (size_t)a.n_groups * grp_stride;

    // phase 2 of a chain-step, one chain-step later: once all P workgroups of the set have handed over their partial tiles of
    // (group, step), this one sums the P tiles of ITS 32 hidden units and writes the rows of dL/dh the layer below reads -- the
    // layer below then takes 2 KB per step from here instead of 46 KB of partial tiles (it is the role with the most to move)
    unsigned char* xr_img = lds + L::O_XR;
    const unsigned xr_lds = (unsigned)(uintptr_t)(lds_ptr_t)xr_img;
    int* const rflags2 = R.flags2;
    bf16_t* __restrict__ Dred = static_cast<bf16_t*>(R.out);   // [T][Bp][Hp]
    const size_t slabH = (size_t)Bp * Hp;
    auto reduce = [&](int g2, int t2) -> bool {
        const FlagPoll peers{rflags + ((size_t)g2 * T + t2) * a.flag_stride, P, nullptr, 0, nullptr};
        if (!flags_wait(peers, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return false;
        constexpr int NPC = P * 2;
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(XE + (size_t)(t2 % kFusedRing) * slot_stride + (size_t)g2 * grp_stride + (size_t)p * P * TILE);
#pragma unroll
        for (int k = 0; k < (NPC + 3) / 4; ++k) {
            const int pc = wave + 4 * k;
            if (pc < NPC) glds16_sc1(uni(xs + pc * 1024), (unsigned)(lane * 16), (unsigned)uni((int)(xr_lds + (unsigned)(pc * 1024))));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int erow = tid >> 3, jq = tid & 7;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* src = xr_img + erow * 64 + jq * 8;
#pragma unroll
        for (int s = 0; s < P; ++s) {
            float f[4];
            unpack_bf16x4(*reinterpret_cast<const uint2*>(src + s * 2048), f);
            s4[0] += f[0]; s4[1] += f[1]; s4[2] += f[2]; s4[3] += f[3];
        }
        const int rb = 32 * g2 + erow;
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(Dred + (size_t)t2 * slabH, (unsigned)(slabH * 2));
        st8_sc1(rd, rb < Bp ? (unsigned)(((size_t)rb * Hp + 32 * p + 4 * jq) * 2) : kOob, pack_bf16x4(s4[0], s4[1], s4[2], s4[3]));
        raise_flag<0>(rflags2 + ((size_t)g2 * T + t2) * a.flag_stride + p);
        return true;
    };

    uint4 dv[2];   // the partner's dA slice of the next chain-step: 32 rows x 4 gates x 64 bytes = 512 pieces
    auto issue_loads = [&](int g2, int t2) {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t2 * slabG, (unsigned)(slabG * 2));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
            const int rb = 32 * g2 + row;
            dv[q] = rb < Bp ? ld16_sc1(rg, (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2)) : make_uint4(0, 0, 0, 0);
        }
    };
    PL_ST_DECL
    int c = 0, t = T - 1, pg = -1, pt = -1;
    {
        const FlagPoll s0 = (G16 ? step_flags_lstm16_w2(a, WT_, set * RC, t, p) : step_flags(a, WT_, set * RC, t, p));
        if (!flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        issue_loads(set * RC, t);
    }
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t - 1; }
        const bool has_next = tn >= 0;
        const int gn = set * RC + cn;
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = (G16 ? step_flags_lstm16_w2(a, WT_, gn, tn, p) : step_flags(a, WT_, gn, tn, p));
        int pvl = 1;
        if (wave == 0 && has_next) pvl = poll_load(pn, lane);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
            *reinterpret_cast<uint4*>(da_img + row * DRS + gate * 64 + q4 * 16) = dv[q];
        }
        if (wave == 0) {   // answered before this wave's first store (see the recurrence role)
            const bool rdy = __all(pvl != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }
        __syncthreads();
        PL_ST(0);
        uint4 bfr[8];   // B fragments of the dA image, shared by all tiles of the wave
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) bfr[ks] = *reinterpret_cast<const uint4*>(da_img + (lane & 31) * DRS + ks * 32 + (lane >> 5) * 16);
        const bool ready = has_next && lflag[0] != 0;
        if (ready) issue_loads(gn, tn);
        bf16_t* xd = XE + (size_t)(t % kFusedRing) * slot_stride + (size_t)g * grp_stride + (size_t)p * TILE;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int nt = wave + 4 * i;
            if (4 * i + 3 < P || nt < P) partial_tile(wreg[i], bfr, tst, TRS, 0, ro, (unsigned)((size_t)nt * P * TILE * 2), lane);
        }
        PL_ST(2);
        raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);
        PL_ST(3);
        if (pg >= 0 && !reduce(pg, pt)) return;
        pg = g;
        pt = t;
        PL_ST(4);   // reduction of the chain-step before
        if (!has_next) break;
        if (!ready) {
            if (!flags_wait(pn, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
            issue_loads(gn, tn);
        }
        PL_ST(5);
        c = cn;
        t = tn;
    }
    if (!reduce(pg, pt)) return;
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward mel head: dL/dmel_t = sum of the embedder's input-gradient partial tiles; dL/dY of the two frames pooled into t
// = loss part (computed before the launch) + 0.5 dL/dmel_t; dL/dh of the predictor's top layer for those two frames =
// dL/dY W_p (one row per pooled frame: both frames get the same)
// ---------------------------------------------------------------------------------------------------------------------
// KS: the predictor's width (the dL/dh rows written), KSS: the embedder's (the sources of the input-gradient tiles; model set B: 12 and 46)
template <int KS, int KSS = KS, bool G16 = false>
__device__ __forceinline__ void fused_head_bwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmBwdLds<KS>;
    constexpr int Hp = 16 * KS, P = Hp / 32, NT = (P + 3) / 4, PS = KSS / 2;
    constexpr int TRS = L::TRS, MRS = L::MRS;
    constexpr size_t TILE = 32 * 32;
    unsigned char* dy_img = lds + L::O_MEL;     // [32 batch rows][64 mel columns] bf16
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);   // wave-uniform, and the compiler should know
    unsigned char* tst = lds + L::O_TST + wave * 32 * TRS;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(R.Wg);   // Wlin^T packed [Hp][64]
    uint4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 32 * (nt < P ? nt : 0) + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { wreg[i][ks] = gld<uint4>(WT + (size_t)n * 64 + 16 * ks + 8 * (lane >> 5)); pin(wreg[i][ks]); }
    }
    const bf16_t* __restrict__ XM = static_cast<const bf16_t*>(R.xchg_mel);   // [ring][groups][2][P src][32][32]
    const size_t mgrp_stride = (size_t)2 * PS * TILE, mslot_stride = (size_t)a.n_groups * mgrp_stride;
    const float* __restrict__ Yl = static_cast<const float*>(R.dh_ext);       // loss part of dL/dY, f32 [2 T][Bp][64]
    bf16_t* __restrict__ Out = static_cast<bf16_t*>(R.out);                   // [T][Bp][Hp]
    const size_t slabH = (size_t)Bp * Hp;
    const int erow = tid >> 3, jq = tid & 7;   // thread -> batch row, mel columns 8 jq .. 8 jq + 7 (tile jq >> 2, columns 8 (jq & 3))
    const int out_dim = R.out_dim;

    uint4 pm[PS];
    float4 y0 = make_float4(0.f, 0.f, 0.f, 0.f), y1 = y0;
    auto issue_loads = [&](int g2, int t2) {
        const bf16_t* xs = XM + (size_t)(t2 % kFusedRing) * mslot_stride + (size_t)g2 * mgrp_stride + (size_t)(jq >> 2) * PS * TILE;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(PS * TILE * 2));
        const unsigned o0 = (unsigned)((erow * 32 + 8 * (jq & 3)) * 2);
#pragma unroll
        for (int s = 0; s < PS; ++s) pm[s] = ld16_sc1_so(rx, o0, (unsigned)(s * TILE * 2));
        int b2 = 32 * g2 + erow;
        b2 = b2 < Bp ? b2 : Bp - 1;
        const float* yr = Yl + ((size_t)(2 * t2) * Bp + b2) * 64 + 8 * jq;
        y0 = gld<float4>(yr);
        y1 = gld<float4>(yr + 4);
    };
    PL_ST_DECL
    int c = 0, t = T - 1;
    {
        const FlagPoll s0 = (G16 ? step_flags_lstm16_w0(a, WT_, set * RC, t, p) : step_flags(a, WT_, set * RC, t, p));
        if (!flags_wait(s0, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
        issue_loads(set * RC, t);
    }
    for (;;) {
        const int g = set * RC + c;
        int cn = c + 1, tn = t;
        if (cn == Ca) { cn = 0; tn = t - 1; }
        const bool has_next = tn >= 0;
        const int gn = set * RC + cn;
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = (G16 ? step_flags_lstm16_w0(a, WT_, gn, tn, p) : step_flags(a, WT_, gn, tn, p));
        int pvl = 1;
        if (wave == 0 && has_next) pvl = poll_load(pn, lane);
        {
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < PS; ++s) {
                float f[4];
                unpack_bf16x4(make_uint2(pm[s].x, pm[s].y), f);
                sum[0] += f[0]; sum[1] += f[1]; sum[2] += f[2]; sum[3] += f[3];
                unpack_bf16x4(make_uint2(pm[s].z, pm[s].w), f);
                sum[4] += f[0]; sum[5] += f[1]; sum[6] += f[2]; sum[7] += f[3];
            }
            const float yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const bool live = 32 * g + erow < a.B;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (live && 8 * jq + k < out_dim) ? yl[k] + 0.5f * sum[k] : 0.f;
            unsigned char* d = dy_img + erow * MRS + jq * 16;
            *reinterpret_cast<uint2*>(d) = pack_bf16x4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint2*>(d + 8) = pack_bf16x4(v[4], v[5], v[6], v[7]);
        }
        if (wave == 0) {
            const bool rdy = __all(pvl != 0);
            if (lane == 0) lflag[0] = rdy ? 1 : 0;
        }
        __syncthreads();
        PL_ST(0);
        uint4 bfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bfr[ks] = *reinterpret_cast<const uint4*>(dy_img + (lane & 31) * MRS + ks * 32 + (lane >> 5) * 16);
        const bool ready = has_next && lflag[0] != 0;
        if (ready) issue_loads(gn, tn);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(Out + (size_t)t * slabH, (unsigned)(slabH * 2));
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int nt = wave + 4 * i;
            if (4 * i + 3 < P || nt < P) {   // a compile-time fact for all but a wave's last tile
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[i][ks]), __builtin_bit_cast(bf16x8, bfr[ks]), acc, 0, 0, 0);
                unsigned char* orow = tst + (lane & 31) * TRS + (4 * (lane >> 5)) * 2;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<uint2*>(orow + rg * 16) = pack_bf16x4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {   // the wave's own tile, by rows: 64 bytes of a dL/dh row per 4 lanes
                    const int cidx = lane + 64 * q, r = cidx >> 2, c4 = cidx & 3;
                    const int rb = 32 * g + r;
                    if (rb < Bp) {
                        const uint4 v = *reinterpret_cast<const uint4*>(tst + r * TRS + (8 * c4) * 2);
                        st16_sc1(ro, (unsigned)(((size_t)rb * Hp + 32 * nt + 8 * c4) * 2), v);
                    }
                }
            }
        }
        PL_ST(2);
        raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride);
        PL_ST(3);
        if (!has_next) break;
        if (!ready) {
            if (!flags_wait(pn, a.status, lflag + 1, a.spin_ticks, a.poll_mask)) return;
            issue_loads(gn, tn);
        }
        PL_ST(5);
        c = cn;
        t = tn;
    }
    PL_ST_DUMP(a.stamps);
}

template <int KS>
__global__ __launch_bounds__(256, 1) void fused_bwd_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[(LstmBwdLds<KS>::BYTES + 15) / 16 * 16];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    // readfirstlane: the table is read with vector loads, and a role index the compiler takes for divergent turns every access
    // to the role's descriptor into a vector load with an s_waitcnt vmcnt(0) -- which also waits for the prefetched tiles
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    // the role's descriptor stays in (constant) device memory: uniform address -> scalar loads; the role functions copy what
    // their loops use into locals (the whole struct held in SGPRs crowds them out, read through LDS it costs a round trip per use)
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    switch (R.type) {
        case FR_LSTM_BWD:
            if (R.xchg_mel && R.xchg_ext) fused_lstm_bwd<KS, true, true>(a, R, set, p, lds);
            else if (R.xchg_mel) fused_lstm_bwd<KS, false, true>(a, R, set, p, lds);
            else if (R.xchg_ext) fused_lstm_bwd<KS, true, false>(a, R, set, p, lds);
            else fused_lstm_bwd<KS, false, false>(a, R, set, p, lds);
            break;
        case FR_DX_BWD: fused_dx_bwd<KS>(a, R, set, p, lds); break;
        case FR_HEAD_BWD: fused_head_bwd<KS>(a, R, set, p, lds); break;
        default: break;
    }
}

template <int KS>
constexpr int fused_fwd_lds_bytes() {
    int m = LstmFwdLds<KS, 0>::BYTES;
    m = m > LstmFwdLds<KS, 2>::BYTES ? m : LstmFwdLds<KS, 2>::BYTES;
    m = m > LstmFwdLds<KS, 4>::BYTES ? m : LstmFwdLds<KS, 4>::BYTES;
    m = m > GemmFwdLds<KS>::BYTES ? m : GemmFwdLds<KS>::BYTES;
    return (m + 15) / 16 * 16;
}

// The roles of one forward launch may have TWO hidden sizes (round 3): the predictor's (KSP: its recurrences, the projections between
// its layers, the mel head) and the embedder's (KSE); a role's descriptor says which (`wide`).  Equal sizes instantiate one set.
template <int KS>
__device__ __forceinline__ void fused_fwd_role(const FusedArgs& a, const FusedRole& R, int set, int p, unsigned char* lds) {
    switch (R.type) {
        case FR_LSTM_FWD:
            if (R.ksx == 2) fused_lstm_fwd<KS, 2>(a, R, set, p, lds);
            else if (R.ksx == 4) fused_lstm_fwd<KS, 4>(a, R, set, p, lds);
            else fused_lstm_fwd<KS, 0>(a, R, set, p, lds);
            break;
        case FR_PROJ_FWD: fused_gemm_fwd<KS, false>(a, R, set, p, lds); break;
        case FR_HEAD_FWD: fused_gemm_fwd<KS, true>(a, R, set, p, lds); break;
        default: break;
    }
}

template <int KSP, int KSE>
__global__ __launch_bounds__(256, 1) void fused_fwd_kernel(FusedArgs a) {
    constexpr int kLds = fused_fwd_lds_bytes<KSP>() > fused_fwd_lds_bytes<KSE>() ? fused_fwd_lds_bytes<KSP>() : fused_fwd_lds_bytes<KSE>();
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    // readfirstlane: the table is read with vector loads, and a role index the compiler takes for divergent turns every access
    // to the role's descriptor into a vector load with an s_waitcnt vmcnt(0) -- which also waits for the prefetched tiles
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    // the role's descriptor stays in (constant) device memory: uniform address -> scalar loads; the role functions copy what
    // their loops use into locals (the whole struct held in SGPRs crowds them out, read through LDS it costs a round trip per use)
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    // Batches of more groups than the roles hold at once (round 4: cfg4's 2048 rows on one GPU are 64 groups) are taken in PASSES: the role
    // table is the one of a chip-load of a.gpp groups, and a workgroup walks its role over the sets s, s + gpp / C, s + 2 gpp / C, ... --
    // every role function starts from scratch (weights, cell state at t = 0), flags and stashes are indexed by the absolute group, so a
    // pass is exactly the launch a batch of those groups alone would run (tests: test_full_size_cfg4_one_gpu..., bit-equal rows).  Roles
    // drift apart by passes as they please: a wait is per (group, step) and every role walks the passes in the same order.
    const int set_step = (FUSED_PASSES && a.gpp > 0) ? uni(a.gpp / R.C) : 0;
    for (int s2 = set;; s2 += set_step) {
        if constexpr (KSP == KSE) {
            fused_fwd_role<KSE>(a, R, s2, p, lds);
        } else {
            if (R.wide) fused_fwd_role<KSE>(a, R, s2, p, lds);
            else fused_fwd_role<KSP>(a, R, s2, p, lds);
        }
        if (!FUSED_PASSES || set_step <= 0 || (s2 + set_step) * R.C >= a.n_groups) break;
        __syncthreads();   // nobody starts the next pass's LDS images while a wave still reads this one's
        if (uni(__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;   // a wait gave up: everybody leaves
    }
}

// ---- the same tables with the LSTM roles on 16 batch rows (lstm_fused16.h): batches of up to 16 rows --------------------------------
template <int KS>
constexpr int fused_fwd16_lds_bytes() {
    int m = LstmFwd16Lds<KS, 0>::BYTES;
    m = m > LstmFwd16Lds<KS, 1>::BYTES ? m : LstmFwd16Lds<KS, 1>::BYTES;
    m = m > LstmFwd16Lds<KS, 2>::BYTES ? m : LstmFwd16Lds<KS, 2>::BYTES;
    m = m > GemmFwdLds<KS>::BYTES ? m : GemmFwdLds<KS>::BYTES;
    return (m + 15) / 16 * 16;
}

template <int KS>
__device__ __forceinline__ void fused_fwd16_role(const FusedArgs& a, const FusedRole& R, int set, int p, unsigned char* lds) {
    switch (R.type) {
        case FR_LSTM_FWD:   // the descriptor counts the fused input projection in k-steps of 16 (in_p = 32 -> 2, 64 -> 4)
            if (R.ksx == 2) fused_lstm_fwd16<KS, 1>(a, R, set, p, lds);
            else if (R.ksx == 4) fused_lstm_fwd16<KS, 2>(a, R, set, p, lds);
            else fused_lstm_fwd16<KS, 0>(a, R, set, p, lds);
            break;
        case FR_PROJ_FWD: fused_gemm_fwd<KS, false, true>(a, R, set, p, lds); break;
        case FR_HEAD_FWD: fused_gemm_fwd<KS, true, true>(a, R, set, p, lds); break;
        default: break;
    }
}

template <int KSP, int KSE>
__global__ __launch_bounds__(256, 1) void fused_fwd16_kernel(FusedArgs a) {
    constexpr int kLds = fused_fwd16_lds_bytes<KSP>() > fused_fwd16_lds_bytes<KSE>() ? fused_fwd16_lds_bytes<KSP>() : fused_fwd16_lds_bytes<KSE>();
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    if constexpr (KSP == KSE) {
        fused_fwd16_role<KSE>(a, R, set, p, lds);
    } else {
        if (R.wide) fused_fwd16_role<KSE>(a, R, set, p, lds);
        else fused_fwd16_role<KSP>(a, R, set, p, lds);
    }
}

template <int KS>
__global__ __launch_bounds__(256, 1) void fused_bwd16_kernel(FusedArgs a) {
    constexpr int kLds = LstmBwdLds<KS>::BYTES > LstmBwd16Lds<KS>::BYTES ? LstmBwdLds<KS>::BYTES : LstmBwd16Lds<KS>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(kLds + 15) / 16 * 16];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    const int pfi = __builtin_amdgcn_readfirstlane((int)bt[3]);   // > 0: a stash prefetcher of this role's set: (count << 8) | (index + 1)
    if (pfi > 0) {
        if (R.type == FR_LSTM_BWD) fused_pf_bwd16<KS>(a, R, set, (pfi & 255) - 1, pfi >> 8, reinterpret_cast<int*>(lds));
        return;
    }
    switch (R.type) {
        case FR_LSTM_BWD:
            if (R.xchg_mel) fused_lstm_bwd16<KS, true>(a, R, set, p, lds);
            else fused_lstm_bwd16<KS, false>(a, R, set, p, lds);
            break;
        case FR_DX_BWD: fused_dx_bwd<KS, true>(a, R, set, p, lds); break;
        case FR_HEAD_BWD: fused_head_bwd<KS, KS, true>(a, R, set, p, lds); break;
        default: break;
    }
}

// the backward launch of a STACKED predictor in front of an embedder of another width (model set B; 16-row roles): every role
// runs at the width its descriptor names (`wide`: the embedder's), the backward mel head at both
template <int KSP, int KSE>
__global__ __launch_bounds__(256, 1) void fused_bwd16_kernel2(FusedArgs a) {
    constexpr int l1 = LstmBwdLds<KSP>::BYTES > LstmBwdLds<KSE>::BYTES ? LstmBwdLds<KSP>::BYTES : LstmBwdLds<KSE>::BYTES;
    constexpr int l2 = LstmBwd16Lds<KSP>::BYTES > LstmBwd16Lds<KSE>::BYTES ? LstmBwd16Lds<KSP>::BYTES : LstmBwd16Lds<KSE>::BYTES;
    constexpr int kLds = l1 > l2 ? l1 : l2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(kLds + 15) / 16 * 16];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    // (no stash prefetchers in the two-width launch: they change nothing there, and their code in this kernel cost its roles 23 more scalar spills)
    switch (R.type) {
        case FR_LSTM_BWD:
            if (R.wide) {
                if (R.xchg_mel) fused_lstm_bwd16<KSE, true, false>(a, R, set, p, lds);
                else fused_lstm_bwd16<KSE, false, false>(a, R, set, p, lds);
            } else {
                fused_lstm_bwd16<KSP, false, false>(a, R, set, p, lds);
            }
            break;
        case FR_DX_BWD:
            if (R.wide) fused_dx_bwd<KSE, true>(a, R, set, p, lds);
            else fused_dx_bwd<KSP, true>(a, R, set, p, lds);
            break;
        case FR_HEAD_BWD: fused_head_bwd<KSP, KSE, true>(a, R, set, p, lds); break;
        default: break;
    }
}

// ... and the same on 32-row tiles (49 ... 128 rows)
template <int KSP, int KSE>
__global__ __launch_bounds__(256, 1) void fused_bwd_kernel2(FusedArgs a) {
    constexpr int kLds = LstmBwdLds<KSP>::BYTES > LstmBwdLds<KSE>::BYTES ? LstmBwdLds<KSP>::BYTES : LstmBwdLds<KSE>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(kLds + 15) / 16 * 16];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    switch (R.type) {
        case FR_LSTM_BWD:
            if (R.wide) {
                if (R.xchg_mel) fused_lstm_bwd<KSE, false, true>(a, R, set, p, lds);
                else fused_lstm_bwd<KSE, false, false>(a, R, set, p, lds);
            } else {
                fused_lstm_bwd<KSP, false, false>(a, R, set, p, lds);
            }
            break;
        case FR_DX_BWD:
            if (R.wide) fused_dx_bwd<KSE>(a, R, set, p, lds);
            else fused_dx_bwd<KSP>(a, R, set, p, lds);
            break;
        case FR_HEAD_BWD: fused_head_bwd<KSP, KSE>(a, R, set, p, lds); break;
        default: break;
    }
}

}  // namespace

#define PL_FUSED_KS_LIST(X) X(6) X(46)
// (predictor, embedder) widths / 16 of the two-width backward launch (16-row roles)
#define PL_FUSED_BWD16_PAIRS(X) X(12, 46)
// (predictor, embedder) hidden sizes / 16 of the forward launch: equal widths, and the class-default stacked predictor (4 x 180) in
// front of a 720-wide embedder (model set B)
#define PL_FUSED_FWD_PAIRS(X) X(6, 6) X(46, 46) X(12, 46)

bool fused_supported(int Hp) {
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_FUSED_KS_LIST(PL_CASE)
#undef PL_CASE
    return false;
}

bool fused_fwd_supported(int Hp_pred, int Hp_emb) {
#define PL_CASE(KP, KE) if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) return true;
    PL_FUSED_FWD_PAIRS(PL_CASE)
#undef PL_CASE
    return false;
}

void launch_fused_fwd(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a) {
#define PL_CASE(KP, KE)                                                                               \
    if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) {                                                    \
        hipLaunchKernelGGL((fused_fwd_kernel<KP, KE>), dim3(a.grid), dim3(256), 0, stream, a);        \
        return;                                                                                       \
    }
    PL_FUSED_FWD_PAIRS(PL_CASE)
#undef PL_CASE
}

void launch_fused_fwd16(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a) {
#define PL_CASE(KP, KE)                                                                               \
    if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) {                                                    \
        hipLaunchKernelGGL((fused_fwd16_kernel<KP, KE>), dim3(a.grid), dim3(256), 0, stream, a);      \
        return;                                                                                       \
    }
    PL_FUSED_FWD_PAIRS(PL_CASE)
#undef PL_CASE
}

void launch_fused_bwd16(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a) {
    if (Hp_pred == Hp_emb) {
#define PL_CASE(K)                                                                              \
    if (Hp_pred == 16 * K) {                                                                    \
        hipLaunchKernelGGL(fused_bwd16_kernel<K>, dim3(a.grid), dim3(256), 0, stream, a);       \
        return;                                                                                 \
    }
    PL_FUSED_KS_LIST(PL_CASE)
#undef PL_CASE
    }
#define PL_CASE(KP, KE)                                                                            \
    if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) {                                                 \
        hipLaunchKernelGGL((fused_bwd16_kernel2<KP, KE>), dim3(a.grid), dim3(256), 0, stream, a);  \
        return;                                                                                    \
    }
    PL_FUSED_BWD16_PAIRS(PL_CASE)
#undef PL_CASE
}

bool fused_bwd16_supported(int Hp_pred, int Hp_emb) {
    if (Hp_pred == Hp_emb) return fused_supported(Hp_pred);
#define PL_CASE(KP, KE) if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) return true;
    PL_FUSED_BWD16_PAIRS(PL_CASE)
#undef PL_CASE
    return false;
}

void launch_fused_bwd(hipStream_t stream, int Hp_pred, int Hp_emb, const FusedArgs& a) {
    if (Hp_pred == Hp_emb) {
#define PL_CASE(K)                                                                              \
    if (Hp_pred == 16 * K) {                                                                    \
        hipLaunchKernelGGL(fused_bwd_kernel<K>, dim3(a.grid), dim3(256), 0, stream, a);         \
        return;                                                                                 \
    }
    PL_FUSED_KS_LIST(PL_CASE)
#undef PL_CASE
    }
#define PL_CASE(KP, KE)                                                                          \
    if (Hp_pred == 16 * KP && Hp_emb == 16 * KE) {                                               \
        hipLaunchKernelGGL((fused_bwd_kernel2<KP, KE>), dim3(a.grid), dim3(256), 0, stream, a);  \
        return;                                                                                  \
    }
    PL_FUSED_BWD16_PAIRS(PL_CASE)
#undef PL_CASE
}

bool fused_passes_compiled() { return FUSED_PASSES != 0; }

}  // namespace pl
