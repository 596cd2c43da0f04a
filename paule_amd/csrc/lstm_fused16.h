// LSTM roles of the fused launches for batches of UP TO 48 ROWS, on 16-row tiles (the reference plans one utterance, paule/paule.py:585-588;
// continued learning uses mini-batches of 8, :404; BASELINE configs[4] leaves 16 utterances per GPU).
//
// Why.  A recurrence of one 16-row group is pure latency: 2 x 23 MFMAs of 16 x 16 x 32 and a 23-KB operand per step.  The 32-row
// roles (lstm_fused.hip) spend a 32 x 32 x 16 chain and a 46-KB tile on it (3.8 us per forward step at B = 16 against 2.8 us of
// the 16-row per-layer kernel, lstm_persist16.hip), which is why batches of up to 48 rows stayed on the chunk pipelines over
// hardware queues (planner.hip, 4.1f) -- and those pay a pipeline fill of one chunk per stage (cfg5: 3.8 of 18.8 ms) that more
// chunks do not shrink: the runtime folds the branches of a captured graph onto four queues and the pipeline drains in the
// middle (profiles/r03_timeline_cfg5_pipelines8.txt).  Inside ONE launch a dependency costs a flag, not a queue:
//
//  * the arithmetic of lstm_persist16.hip (A tiles ordered [unit][gate], two 16-row tiles per wave forward; 46 N tiles of 16
//    hidden units backward, wide ingest, every wave hands its own tiles over), as roles of the fused launches' tables: same
//    flags, same stashes, same product roles (mel head, projections, dL/dh products, backward mel head: those keep their
//    32-row tiles and guard the rows that do not exist);
//  * the role's OWN exchange takes the verified same-XCD form of the per-layer kernels (sweep_common.h): the host table puts a
//    recurrence's 23 workgroups on one XCD slot, every workgroup publishes the XCD it runs on with its first hand-off, and a
//    role that finds itself on one XCD switches its own hand-off to plain stores / nt loads (served by that XCD's L2) and
//    its own flags to a second, plain flag set.  What OTHER roles read stays write-through: the forward role stores h twice
//    (a private 2-slot copy for itself, the stash for everybody else), and the write-through flag of a step is raised ONE STEP
//    LATE, together with the next step's plain flag: a wave's stores complete in order, so the drain the plain flag needs
//    anyway covers the write-through stores of the step before -- nothing is added to the recurrence's critical path;
//  * what a step needs from another role (`FusedWait` 1 and 2) is looked at AHEAD (one step forward, two steps backward) by a
//    wave that does not poll (wave 3): in steady state the producers are ahead, the blocking wait then polls the role's own
//    flags only (an L2 round trip), and the backward role fetches the write-through dL/dh row a step before it is used;
//  * operands nobody in the launch writes (CP frames forward, stash rows backward) are fetched one step ahead, BEHIND the
//    hand-off loads of the step before: a wave's loads return in order, a memory-latency load in front of the flag poll or of
//    the tile loads would hold them back by its whole latency.
//
// Measured (profiles/r03_ab_fused16_final.txt, r03_bench_cfg5.json): B = 1 .. 16 x 300 frames 3.07 -> 2.05 ms per iteration, cfg5
// (16 x 2000) 18.6 -> 13.8 ms; with the write-through form of the own exchange forced 2.33 ms at B = 16 x 300.  Phase stamps of the
// critical roles (tools/fused_stamps.py 16): forward 2.6 us per step (wait 0.7, h tile -> LDS 0.6, MFMA chain 0.8, cell + stores +
// flag 0.5), backward ~3.9 (wait 1.0, ingest 1.0, cell 0.45, tiles 1.1, drain + flag 0.3).  What was tried and dropped: DESIGN.md A.7.
//
// A stacked predictor of another width than the embedder (model set B) runs on the same roles through the two-width kernels of
// lstm_fused.hip (fused_bwd16_kernel2): the predictor's layers and their dL/dh product roles at the predictor's width.
//
// One 16-row group per set of workgroups, one chain: the host plans these roles for up to 48 rows (one to three groups; planner.hip:
// plan_fused).  The product roles keep one set per 32 rows and wait for the flag rows of both 16-row groups of their 32
// (step_flags_lstm16_w0 / _w2, fused_common.h).
#pragma once
#include "fused_common.h"

#ifndef F16_TG
#define F16_TG 3     // backward: partial tiles whose MFMAs interleave (independent accumulators)
#endif
#ifndef F16_PF
#define F16_PF 4     // forward: B-fragment read-ahead of the MFMA chain (k-steps of 32)
#endif
#ifndef F16_DIRECT_STORE
#define F16_DIRECT_STORE 1   // backward, same-XCD hand-off: partial tiles stored straight from the accumulator layout (8 bytes per lane)
#endif
// -DF16_NO_LOOKAHEAD (diagnostic builds only): the look-ahead on other roles' flags never answers "seen" -- every step's blocking wait
// polls them, every write-through row is fetched behind it.  Used to rule the look-ahead out while the reproducibility bug of the
// first backward role was bisected; same results, a little slower.
#ifndef F16_PIPE
#define F16_PIPE 0   // backward: 1 = the next group's MFMAs are issued before this group's epilogue
#endif

namespace pl {
namespace {

__device__ __forceinline__ unsigned short bf16_bits16(float v) { return __builtin_bit_cast(unsigned short, (bf16_t)v); }
__device__ __forceinline__ float bf16_val16(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// blocking, bounded wait of a 16-row role: its own P flags (lanes 0 .. P-1; nt loads once the role runs on one XCD) and, unless the
// look-ahead has seen them already, the flags of other roles (write-through).  Wave 0 polls; ends with a barrier.
__device__ __forceinline__ bool wait16(const int* own, int n_own, bool own_fast, const FlagPoll& ext, bool with_ext, int* status,
                                       int* lds_word, unsigned long long spin_ticks, unsigned poll_mask) {
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const __amdgpu_buffer_rsrc_t rf = make_rsrc(own ? own : status, (unsigned)((own ? n_own : 1) * 4));
        const int n = own ? n_own : 0;
        int ok = 1;
        for (unsigned spin = 1;; ++spin) {
            int v = 1;
            if (lane < n) {
                v = own_fast ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxNt)
                             : (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
            } else if (with_ext) {
                if (lane >= 32 && lane - 32 < ext.nb) v = flag_load(ext.fb + (lane - 32));
                else if (lane == 63 && ext.fc) v = flag_load(ext.fc);
            }
            if (__all(v != 0)) break;
            if ((spin & poll_mask) == 0 && (flag_load(status) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks)) {
                ok = 0;
                break;
            }
        }
        if (lane == 0) {
            if (!ok) {
                int expected = 0;
                __hip_atomic_compare_exchange_strong((PL_GLOBAL int*)status, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *lds_word = ok;
        }
    }
    __syncthreads();
    return *lds_word != 0;
}

// what a step waits for from OTHER roles (FusedWait 1 -> lanes 32.., FusedWait 2 -> lane 63)
__device__ __forceinline__ FlagPoll ext_flags(const FusedArgs& a, const FusedRole& R, int g32, int t, int p) {   // g32: the product roles' 32-row group
    FlagPoll s{nullptr, 0, nullptr, 0, nullptr};
    int n2 = 0;
    s.fb = wait_addr(a, R.wait[1], g32, t, p, s.nb);
    s.fc = wait_addr(a, R.wait[2], g32, t, p, n2);
    return s;
}
__device__ __forceinline__ bool ext_empty(const FlagPoll& s) { return s.nb == 0 && !s.fc; }
__device__ __forceinline__ int ext_poll(const FlagPoll& s, int lane) {
    int v = 1;
    if (lane >= 32 && lane - 32 < s.nb) v = flag_load(s.fb + (lane - 32));
    else if (lane == 63 && s.fc) v = flag_load(s.fc);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward recurrence (arithmetic of lstm_fwd16_sweep_kernel)
// ---------------------------------------------------------------------------------------------------------------------
template <int KS, int KSX>   // KSX = in_p / 32 (0: G holds the input projection, written by a projection role of this launch)
struct LstmFwd16Lds {
    static constexpr int Hp = 16 * KS;
    static constexpr int RS = Hp * 2 + 16;        // h image row stride: odd number of 16-byte chunks
    static constexpr int ORS = 64 + 16;           // staged outputs [6 arrays][16 rows][32 units] bf16
    static constexpr int XRS = KSX * 64 + 16;
    static constexpr int O_HIMG = 0;
    static constexpr int O_OST = O_HIMG + 16 * RS;
    static constexpr int O_XIMG = O_OST + 6 * 16 * ORS;
    static constexpr int O_FLAG = O_XIMG + (KSX ? 16 * XRS : 16);
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, int KSX>
__device__ __forceinline__ void fused_lstm_fwd16(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmFwd16Lds<KS, KSX>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, KS32 = KS / 2, P = Hp / 32;
    constexpr int ROWB = Hp * 2, RS = L::RS, ORS = L::ORS, XRS = L::XRS;
    constexpr int CPR = Hp / 8;                   // 16-byte chunks per h row
    constexpr int NLD = (16 * CPR + 255) / 256;
    constexpr int PF = F16_PF;                    // B-fragment read-ahead (k-steps of 32)
    constexpr int INP = KSX ? 32 * KSX : 32, XC = INP / 8;
    static_assert(KS % 2 == 0, "Hp is a multiple of 32");
    unsigned char* himg = lds + L::O_HIMG;
    unsigned char* ost = lds + L::O_OST;
    unsigned char* ximg = lds + L::O_XIMG;
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int Bp = a.Bp, T = R.T;
    // one 16-row group per set (one chain): set s serves batch rows 16 s .. 16 s + 15 (Bp is a multiple of 16: all of them exist).  Every
    // pointer below is moved to the group's first row / the group's own region once, so the step loop indexes rows 0 .. 15
    const int g = set, g32 = g >> 1, ng16 = Bp / 16;
    if (16 * g >= Bp) return;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(R.W);

    // weights -> registers: A row lr of tile j = unit 8 wave + 4 j + (lr >> 2), gate lr & 3; k = 32 ks + 8 kq .. + 7
    uint4 wreg[2][KS32];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const bf16_t* wrow = W + (size_t)((lr & 3) * Hp + 32 * p + 8 * wave + 4 * jt + (lr >> 2)) * Hp + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) { wreg[jt][ks] = gld<uint4>(wrow + 32 * ks); pin(wreg[jt][ks]); }
    }
    uint4 wx[2][KSX ? KSX : 1];
    float bias_r[2][4];
    if constexpr (KSX > 0) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const bf16_t* xrow = static_cast<const bf16_t*>(R.Wih) + (size_t)((lr & 3) * Hp + 32 * p + 8 * wave + 4 * jt + (lr >> 2)) * INP + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) { wx[jt][ks] = gld<uint4>(xrow + 32 * ks); pin(wx[jt][ks]); }
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) bias_r[jt][gate] = gld<float>(R.bias + gate * Hp + 32 * p + 8 * wave + 4 * jt + kq);
        }
    }
    // cell ownership (C layout): batch row lr, units 8 wave + kq (tile 0) and 8 wave + 4 + kq (tile 1); accumulator register = gate
    const int ul[2] = {8 * wave + kq, 8 * wave + 4 + kq};
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G) + (size_t)16 * g * G4;
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(R.h) + (size_t)16 * g * Hp;
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(R.c) + (size_t)16 * g * Hp;
    const size_t hx_slot = (size_t)ng16 * 16 * Hp;
    bf16_t* __restrict__ HX = R.hx ? static_cast<bf16_t*>(R.hx) + (size_t)g * 16 * Hp : nullptr;   // [2][groups][16][Hp], this role's own copy of the hand-off
    const bf16_t* const x_in = R.x_in ? static_cast<const bf16_t*>(R.x_in) + (size_t)16 * g * INP : nullptr;
    const bool src_sc1 = R.src_sc1 != 0;
    const int fs = a.flag_stride;
    int* const Fpub = R.flags + (size_t)g * T * fs;
    int* const Ffast = R.fast_flags ? R.fast_flags + (size_t)g * T * fs : nullptr;
    int* const xtab = R.xtab ? R.xtab + g * 64 : nullptr;

    float c_state[2] = {0.f, 0.f};
    bool fast = false;          // the role's workgroups share one XCD (verified at step 1): own hand-off through its L2
    bool pub_pending = false;   // the write-through flag of step t - 1 is still to be raised
    bool ext_known = false;     // the look-ahead saw everything step t needs from other roles
    int la_pv = 0;              // wave 3: the look-ahead's answer (flags of step t + 1), asked at the top of step t
    // input rows nobody in this launch writes (the predictor's CP frames) are fetched ONE STEP AHEAD, behind the h tile loads: a
    // wave's loads return in order, and a memory-latency load in front of wave 0's flag poll delays the poll by its whole latency
    uint4 xv_n = make_uint4(0, 0, 0, 0);
    auto fetch_x = [&](int t2) {
        if constexpr (KSX > 0) {
            if (!src_sc1 && wave < KSX) xv_n = gld<uint4>(x_in + ((size_t)t2 * Bp + tid / XC) * INP + (tid % XC) * 8);
        }
    };
    PL_ST_DECL
    fetch_x(0);

    for (int t = 0; t < T; ++t) {
        const FlagPoll ext = ext_flags(a, R, g32, t, p);
        const bool has_ext = !ext_empty(ext);
        // look-ahead at what step t + 1 needs from other roles: asked by a wave that does not poll, answered under this step
        FlagPoll ext_n{nullptr, 0, nullptr, 0, nullptr};
        if (t + 1 < T) ext_n = ext_flags(a, R, g32, t + 1, p);
        const bool la_here = !ext_empty(ext_n);
        if (wave == 3 && la_here) la_pv = ext_poll(ext_n, lane);
        uint4 xv = xv_n;   // fetched a step ago
        if (t > 0 || (has_ext && !ext_known)) {
            const bool of = fast && t >= 2;   // step t - 1 was handed over in the same-XCD form
            const int* own = t > 0 ? (of ? Ffast : Fpub) + (size_t)(t - 1) * fs : nullptr;
            if (!wait16(own, P, of, ext, has_ext && !ext_known, a.status, lflag, a.spin_ticks, a.poll_mask)) return;
        }
        if (t == 1 && Ffast && HX) fast = group_on_one_xcd(xtab, P, lflag + 2);
        PL_ST(0);   // wait
        if constexpr (KSX > 0) {
            if (src_sc1 && wave < KSX) {
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(x_in + (size_t)t * Bp * INP, (unsigned)((size_t)Bp * INP * 2));
                xv = ld16_sc1(rx, (unsigned)(((tid / XC) * INP + (tid % XC) * 8) * 2));
            }
        }
        f32x4 acc[2];
        if (t > 0) {
            const bool from_hx = fast && t >= 2;
            const bf16_t* hsrc = from_hx ? HX + (size_t)((t - 1) & 1) * hx_slot : Hs + (size_t)(t - 1) * slabH;
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(hsrc, (unsigned)(16 * ROWB));
            uint4 v[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int q = tid + 256 * i, row = q / CPR, c = q % CPR;
                v[i] = q < 16 * CPR ? ld16_handoff(rh, (unsigned)(row * ROWB + c * 16), from_hx) : make_uint4(0, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            if (t + 1 < T) fetch_x(t + 1);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int q = tid + 256 * i;
                if (q < 16 * CPR) *reinterpret_cast<uint4*>(himg + (q / CPR) * RS + (q % CPR) * 16) = v[i];
            }
        }
        else if (t + 1 < T) fetch_x(t + 1);
        if constexpr (KSX > 0) {
            if (wave < KSX) *reinterpret_cast<uint4*>(ximg + (tid / XC) * XRS + (tid % XC) * 16) = xv;
        }
        __syncthreads();
        PL_ST(1);   // operands -> LDS
        if (t == 0 && tid == 0 && xtab) flag_store(xtab + p, xcc_id_plus1());   // (not between the MFMA chain and the cell update: no branch there)
        // the projection rows of this step (written by a projection role: the blocking wait or the look-ahead has seen its flag)
        float gx[2][4];
        if constexpr (KSX == 0) {
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) {
                    const unsigned off = (unsigned)(((size_t)lr * G4 + gate * Hp + 32 * p + ul[jt]) * 2);
                    const unsigned short u = src_sc1 ? (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rg, off, 0, kAuxSc1)
                                                     : (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rg, off, 0, 0);
                    gx[jt][gate] = bf16_val16(u);
                }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                acc[jt] = f32x4{bias_r[jt][0], bias_r[jt][1], bias_r[jt][2], bias_r[jt][3]};
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) gx[jt][gate] = 0.f;
            }
        }
        if (t > 0) {
            const unsigned char* bsrc = himg + lr * RS + kq * 16;
            uint4 bq[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < KS32) bq[i] = *reinterpret_cast<const uint4*>(bsrc + i * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, bq[ks % PF]);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[0][ks]), bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[1][ks]), bf, acc[1], 0, 0, 0);
                if (ks + PF < KS32) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (KSX > 0) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ximg + lr * XRS + ks * 64 + kq * 16));
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[0][ks]), xb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[1][ks]), xb, acc[1], 0, 0, 0);
            }
        }
        PL_ST(2);   // MFMA
        // cell update (2 cells per lane) -> all six outputs into the staging image [array][row][unit]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const float vi = sigmoid_fast(acc[jt][0] + gx[jt][0]), vf = sigmoid_fast(acc[jt][1] + gx[jt][1]);
            const float vg = tanh_fast(acc[jt][2] + gx[jt][2]), vo = sigmoid_fast(acc[jt][3] + gx[jt][3]);
            c_state[jt] = cell_c(vf, c_state[jt], vi, vg);
            const float vh = vo * tanh_fast(c_state[jt]);
            unsigned char* o = ost + lr * ORS + ul[jt] * 2;
            *reinterpret_cast<unsigned short*>(o) = bf16_bits16(vh);
            *reinterpret_cast<unsigned short*>(o + 1 * 16 * ORS) = bf16_bits16(vi);
            *reinterpret_cast<unsigned short*>(o + 2 * 16 * ORS) = bf16_bits16(vf);
            *reinterpret_cast<unsigned short*>(o + 3 * 16 * ORS) = bf16_bits16(vg);
            *reinterpret_cast<unsigned short*>(o + 4 * 16 * ORS) = bf16_bits16(vo);
            *reinterpret_cast<unsigned short*>(o + 5 * 16 * ORS) = bf16_bits16(c_state[jt]);
        }
        // the look-ahead's answer, for everybody after the next barrier
        if (wave == 3 && lane == 0) lflag[3] = 0;
        if (wave == 3 && la_here) {
            const bool seen = __all(la_pv != 0);
            if (lane == 0) lflag[3] = seen ? 1 : 0;
        }
        __syncthreads();
        ext_known = lflag[3] != 0;
#ifdef F16_NO_LOOKAHEAD
        ext_known = false;
#endif
        PL_ST(3);   // cell update
        // hand-off first (wave 0: 16 rows x four 16-byte pieces), then the five stash arrays (320 pieces)
        if (wave == 0) {
            const int row = tid >> 2, qt = tid & 3;
            const uint4 hv = *reinterpret_cast<const uint4*>(ost + row * ORS + qt * 16);
            u32x4 d;
            d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
            if (fast) {   // the role's own copy: plain, stays in this XCD's L2
                const __amdgpu_buffer_rsrc_t rx = make_rsrc(HX + (size_t)(t & 1) * hx_slot, (unsigned)(16 * ROWB));
                __builtin_amdgcn_raw_buffer_store_b128(d, rx, (unsigned)((row * Hp + 32 * p + 8 * qt) * 2), 0, 0);
            }
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)t * slabH, (unsigned)(slabH * 2));
            __builtin_amdgcn_raw_buffer_store_b128(d, ro, (unsigned)((row * Hp + 32 * p + 8 * qt) * 2), 0, kAuxSc1);
        }
        asm volatile("" ::: "memory");   // keep the stash stores behind the hand-off
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 || wave == 0) {   // 320 pieces: array e / 64 (gates i f g o, c), row (e % 64) / 4, quarter e % 4
                const int e = tid + 256 * i;
                const int arr = e >> 6, row = (e & 63) >> 2, qt = e & 3;
                const uint4 sv = *reinterpret_cast<const uint4*>(ost + (arr + 1) * 16 * ORS + row * ORS + qt * 16);
                bf16_t* dst = arr < 4 ? G + (size_t)t * slabG + (size_t)row * G4 + arr * Hp + 32 * p + 8 * qt
                                      : Cs + (size_t)t * slabH + (size_t)row * Hp + 32 * p + 8 * qt;
                gst<uint4>(dst, sv);
            }
        }
        PL_ST(4);   // store issue
        if (fast) {
            // wave 0's own-copy store is older than its write-through store and its two stash stores -- and younger than the
            // write-through store of step t - 1 (a wave's stores complete in order): that step's write-through flag goes up here too
            if (wave == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const __amdgpu_buffer_rsrc_t rf = make_rsrc(Ffast + (size_t)t * fs + p, 4u);
                __builtin_amdgcn_raw_buffer_store_b32(1u, rf, 0u, 0, 0);
                if (pub_pending) flag_store(Fpub + (size_t)(t - 1) * fs + p, 1);
            }
            pub_pending = true;
        } else {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // write-through store older than the two stash stores
            __syncthreads();
            if (tid == 0) flag_store(Fpub + (size_t)t * fs + p, 1);
        }
        PL_ST(5);   // drain + barrier + flag
    }
    if (pub_pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) flag_store(Fpub + (size_t)(T - 1) * fs + p, 1);
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward recurrence, reduce-scatter of bf16 partial tiles (arithmetic of lstm_bwd16_rs_sweep_kernel: wide ingest, every wave
// hands its own tiles over).  MEL: the embedder's first layer also multiplies its dA with its rows of W_ih^T (four 16-column
// tiles, one per wave) and hands the partial tiles to the backward mel head in the head's [32][32] layout (rows 0 .. 15).
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct LstmBwd16Lds {
    static constexpr int Hp = 16 * KS;
    static constexpr int DRS = 128 * 2 + 16;      // dA image [16 batch rows][128 local gate rows] bf16
    static constexpr int ORS = (Hp + 64) * 2 + 16;   // partial image [16 batch rows][Hp + 64] bf16 (a wave's surplus tile lands in the 64 spare columns)
    static constexpr int MRS = 64 * 2 + 16;       // input-gradient image [16 batch rows][64 mel columns] bf16
    static constexpr int RED_BYTES = 4 * 16 * 36 * 4;   // the four waves' f32 sums of the wide ingest
    static constexpr int O_DA = 0;
    static constexpr int O_OUT = O_DA + 16 * DRS;
    static constexpr int O_RED = O_OUT + 16 * ORS;
    static constexpr int O_MEL = O_RED + RED_BYTES;
    static constexpr int O_FLAG = O_MEL + 16 * MRS;
    static constexpr int GRS = 4 * 64 + 16;       // gate rows of the step [16 batch rows][4 gates][32 units] bf16 (fetched as 16-byte pieces, read back as cells)
    static constexpr int O_GST = O_FLAG + 64;
    static constexpr int BYTES = O_GST + 16 * GRS;
};

// Stash prefetcher of a 16-row backward recurrence (round 5; block-table entries with a prefetcher index behind a role's set, on its XCD slot).
// The role fetches the six stash rows of a step (gates, c_{t-1}, dL/dh) one step ahead, from HBM, and everything its waves ask for next returns
// behind them (one row less was worth 4.3 % at cfg5).  Prefetcher q of n follows the role's write-through flags -- slice 0's flag of step t + D --
// and reads one dword of every 128-byte line of the group's gate rows of step t and of its c rows of step t - 1, so that they sit in the
// XCD's L2 when the role asks.  Speed only: nothing waits for a prefetcher, it waits for nobody longer than 20 us at a time and gives up
// after eight time-outs in a row (a role it cannot see), and a launch without them computes the same bits.
template <int KS>
__device__ __forceinline__ void fused_pf_bwd16(const FusedArgs& a, const FusedRole& R, const int set, const int q, const int n, int* lflag) {
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, D = 4;
    const int tid = threadIdx.x, Bp = a.Bp, T = R.T, g = set, fs = a.flag_stride;
    if (16 * g >= Bp || n < 1) return;
    const size_t slabG = (size_t)Bp * G4 * 2, slabH = (size_t)Bp * Hp * 2;   // bytes per step
    const unsigned char* G = static_cast<const unsigned char*>(R.G) + (size_t)16 * g * G4 * 2;
    const unsigned char* Cs = static_cast<const unsigned char*>(R.c) + (size_t)16 * g * Hp * 2;
    const int* Fpub = R.flags + (size_t)g * T * fs;
    constexpr unsigned bytesG = 16u * G4 * 2u, bytesC = 16u * Hp * 2u, linesG = (bytesG + 127u) / 128u, linesC = (bytesC + 127u) / 128u;
    unsigned acc = 0;
    int misses = 0;
    for (int t = T - 1; t >= 0; --t) {
        if (t + D <= T - 1) {   // pace: step t + D of the role has been handed over (its write-through flag trails by one step)
            if (tid < 64) {
                int ok = 1;
                if (tid == 0) {
                    const __amdgpu_buffer_rsrc_t rf = make_rsrc(Fpub + (size_t)(t + D) * fs, 4u);
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (__builtin_amdgcn_raw_buffer_load_b32(rf, 0u, 0, kAuxSc1) == 0) {
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000ull) { ok = 0; break; }
                        __builtin_amdgcn_s_sleep(48);
                    }
                    *lflag = ok;
                }
            }
            __syncthreads();
            misses = *lflag ? 0 : misses + 1;
            __syncthreads();
            if (misses >= 8 || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        }
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, bytesG);
        for (unsigned l = (unsigned)(q * 256 + tid); l < linesG; l += (unsigned)(n * 256)) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rg, l * 128u, 0, 0);
        if (t > 0) {
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cs + (size_t)(t - 1) * slabH, bytesC);
            for (unsigned l = (unsigned)(q * 256 + tid); l < linesC; l += (unsigned)(n * 256)) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rc, l * 128u, 0, 0);
        }
    }
    if (acc == 0x7fc01234u) __hip_atomic_fetch_or(a.status, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // keeps the loads alive, changes nothing
}

template <int KS, bool MEL, bool OPT = true>   // OPT: the round-5 stash forms (gate image, carried c row); off in the two-width launch, which they slow down
__device__ __forceinline__ void fused_lstm_bwd16(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = LstmBwd16Lds<KS>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, P = Hp / 32;
    constexpr int NTT = Hp / 16;                  // N tiles of 16 hidden units
    constexpr int NT = (NTT + 3) / 4;             // per wave (wave w: tiles w, w + 4, ...)
    constexpr int TPG = (P + 3) / 4;              // sources a wave sums in the ingest
    constexpr int DRS = L::DRS, ORS = L::ORS, MRS = L::MRS;
    constexpr size_t TILE = 16 * 32;              // own exchange: [16 rows][32 columns] bf16
    constexpr size_t TILE32 = 32 * 32;            // the product roles' tiles
    unsigned char* da_img = lds + L::O_DA;
    unsigned char* out_img = lds + L::O_OUT;
    float* red = reinterpret_cast<float*>(lds + L::O_RED);
    unsigned char* mel_img = lds + L::O_MEL;
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int Bp = a.Bp, T = R.T;
    const int g = set, g32 = g >> 1, ng16 = Bp / 16;   // one 16-row group per set; pointers moved to the group once (see the forward role)
    if (16 * g >= Bp) return;
    const bf16_t* __restrict__ WT = static_cast<const bf16_t*>(R.W);   // Whh^T packed [Hp][4 Hp]
    // tile nt = wave + 4 i: A row lr = hidden column 16 nt + lr; k chunk kc (= gate kc): gate row kc * Hp + 32 p + 8 kq + jj
    uint4 wreg[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nt = wave + 4 * i;
        const int n = 16 * (nt < NTT ? nt : 0) + lr;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) { wreg[i][kc] = gld<uint4>(WT + (size_t)n * G4 + kc * Hp + 32 * p + 8 * kq); pin(wreg[i][kc]); }
    }
    uint4 wmel[4] = {};
    if constexpr (MEL) {   // wave w: mel columns 16 w .. 16 w + 15 (out_p = 64: the host checks)
        const bf16_t* WM = static_cast<const bf16_t*>(R.Wg);   // Wih^T packed [out_p][4 Hp]
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) { wmel[kc] = gld<uint4>(WM + (size_t)(16 * wave + lr) * G4 + kc * Hp + 32 * p + 8 * kq); pin(wmel[kc]); }
    }
    // cell ownership: thread -> batch row tid >> 4, hidden units 32 p + 2 (tid & 15), + 1
    const int erow = tid >> 4, jq = tid & 15;
    const int j = 32 * p + 2 * jq;
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G) + (size_t)16 * g * G4;
    const bf16_t* __restrict__ Cs = static_cast<const bf16_t*>(R.c) + (size_t)16 * g * Hp;
    const bf16_t* __restrict__ dhe = R.dh_ext ? static_cast<const bf16_t*>(R.dh_ext) + (size_t)16 * g * Hp : nullptr;
    const bf16_t* __restrict__ dhl = R.dh_last ? static_cast<const bf16_t*>(R.dh_last) + (size_t)16 * g * Hp : nullptr;
    // own exchange [2 slots][groups][P destinations][P sources][16][32]
    bf16_t* __restrict__ X = static_cast<bf16_t*>(R.xchg) + (size_t)g * P * P * TILE;
    // input-gradient tiles in the backward mel head's layout [ring][32-row groups][2 tiles of 32 columns][P sources][32][32]: this group's
    // rows are rows 16 (g & 1) .. of its 32-row group's tiles
    bf16_t* __restrict__ XM = R.xchg_mel ? static_cast<bf16_t*>(R.xchg_mel) + (size_t)g32 * 2 * P * TILE32 + (size_t)(g & 1) * 16 * 32 : nullptr;
    const size_t slot_stride = (size_t)ng16 * P * P * TILE;
    const size_t mslot_stride = (size_t)a.n_groups * 2 * P * TILE32;
    const bool src_sc1 = R.src_sc1 != 0, dA_sc1 = R.dA_sc1 != 0;
    const int dh_ext_half = R.dh_ext_half, dh_ext_rows = R.dh_ext_rows;
    const int fs = a.flag_stride;
    int* const Fpub = R.flags + (size_t)g * T * fs;
    int* const Ffast = R.fast_flags ? R.fast_flags + (size_t)g * T * fs : nullptr;
    int* const xtab = R.xtab ? R.xtab + g * 64 : nullptr;
    auto pk2 = [](float x, float y) -> unsigned { return (unsigned)bf16_bits16(x) | ((unsigned)bf16_bits16(y) << 16); };

    float dc_next[2] = {0.f, 0.f};
    bool fast = false;
    bool pub_pending = false;   // the write-through flag of step t + 1 is still to be raised
    // What a step needs from other roles is looked at TWO steps ahead (wave 3, at the top of a step; answered by its end): known_cur
    // says the flags of step t are known to be up, known_nxt those of step t - 1 -- early enough to fetch step t - 1's write-through
    // dL/dh row one step ahead, right behind this step's tile loads.
    bool known_cur = false, known_nxt = false;
    int la_pv = 0;
    // Stash rows (and dL/dh from above) of the NEXT step are fetched one step ahead, BEHIND this step's tile loads: a wave's loads
    // return in order, so a memory-latency load issued in front of the flag poll (wave 0) or of the tile loads delays them by its
    // whole latency -- 0.7 us per step when the stash rows were fetched at the top of their own step.
    // The slice's gate rows of a step are 16 rows x 4 gates x 64 bytes = 256 pieces of 16 bytes: ONE load per lane (piece tid: row tid >> 4, gate
    // (tid >> 2) & 3, quarter tid & 3), handed to the cell threads through an LDS image at the top of the step that uses them -- as four 4-byte
    // loads per lane they were four of the seven stash rows in the waves' in-order queues (round 5: one row less had been worth 4.3 % at cfg5)
    // (the equal-width launch only: in the two-width launch of model set B -- narrow predictor roles, a wide embedder role with slack -- cfg5_setB
    // went 12.86 -> 13.06 ms with these forms, whichever of its roles had them)
    constexpr bool GIMG = OPT && KS >= 24;
    uint4 n_gv = make_uint4(0u, 0u, 0u, 0u);
    unsigned n_g[4] = {0u, 0u, 0u, 0u};
    unsigned n_c = 0u, n_cp = 0u, n_dh = 0u;
    unsigned char* const gst = lds + L::O_GST;
    bool n_dh_valid = false;
    auto dh_row_of = [&](int t2) { return dh_ext_half ? (t2 >> 1) : t2; };
    auto fetch_stash = [&](int t2) {
        if constexpr (GIMG) {
            n_gv = gld<uint4>(G + (size_t)t2 * slabG + (size_t)(tid >> 4) * G4 + ((tid >> 2) & 3) * Hp + 32 * p + 8 * (tid & 3));
        } else {
            const bf16_t* g_row = G + (size_t)t2 * slabG + (size_t)erow * G4 + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) n_g[q] = *(const PL_GLOBAL unsigned*)(g_row + q * Hp);
        }
        // c_t of step t2 is the c_{t-1} fetched for step t2 + 1 (one chain: consecutive calls are consecutive steps): loaded once
        // (wide roles, as the image above: cfg5_setB 12.85 -> 13.05 ms with the carry in its narrow roles.  The test on t2 stays a run-time one: with the
        // first call peeled off -- no c load in the loop at all -- set A's gain shrank from 7 % to 2.4 %: what the compiler makes of the merge matters
        // more here than the load it saves, see NOTEBOOK.md A.18)
        n_c = (!GIMG || t2 == T - 1) ? *(const PL_GLOBAL unsigned*)(Cs + (size_t)t2 * slabH + (size_t)erow * Hp + j) : n_cp;
        n_cp = t2 > 0 ? *(const PL_GLOBAL unsigned*)(Cs + (size_t)(t2 - 1) * slabH + (size_t)erow * Hp + j) : 0u;
    };
    auto fetch_dh = [&](int t2) -> unsigned {
        const int row = dh_row_of(t2);
        if (!dhe || row >= dh_ext_rows) return 0u;
        if (src_sc1) {
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(dhe + (size_t)row * slabH, (unsigned)(slabH * 2));
            return __builtin_amdgcn_raw_buffer_load_b32(rd, (unsigned)(((size_t)erow * Hp + j) * 2), 0, kAuxSc1);
        }
        return *(const PL_GLOBAL unsigned*)(dhe + (size_t)row * slabH + (size_t)erow * Hp + j);
    };
    auto unpack2 = [](unsigned u, float (&f)[2]) {
        f[0] = bf16_val16((unsigned short)(u & 0xffffu));
        f[1] = bf16_val16((unsigned short)(u >> 16));
    };
    PL_ST_DECL
    fetch_stash(T - 1);
    if (!dhe && dhl) { n_dh = *(const PL_GLOBAL unsigned*)(dhl + (size_t)erow * Hp + j); n_dh_valid = true; }
    else if (!dhe) n_dh_valid = true;

    for (int t = T - 1; t >= 0; --t) {
        const FlagPoll ext = ext_flags(a, R, g32, t, p);
        const bool has_ext = !ext_empty(ext);
        FlagPoll ext_2{nullptr, 0, nullptr, 0, nullptr};
        if (t > 1) ext_2 = ext_flags(a, R, g32, t - 2, p);
        const bool la_here = !ext_empty(ext_2);
        if (wave == 3 && la_here) la_pv = ext_poll(ext_2, lane);
        // this step's operands: fetched one step ago; the gate pieces go through the image (read behind the barrier of the ingest, below)
        if constexpr (GIMG) *reinterpret_cast<uint4*>(gst + (tid >> 4) * L::GRS + ((tid >> 2) & 3) * 64 + (tid & 3) * 16) = n_gv;
        const unsigned d_g0 = n_g[0], d_g1 = n_g[1], d_g2 = n_g[2], d_g3 = n_g[3];   // (the direct form's operands)
        const unsigned u_c = n_c, u_cp = n_cp;
        unsigned dh_bits = n_dh;
        const bool dh_have = n_dh_valid;
        const bool wait_ext = has_ext && !known_cur;
        if (t + 1 < T || wait_ext) {
            const bool of = fast && t + 1 <= T - 2;   // step t + 1 was handed over in the same-XCD form
            const int* own = t + 1 < T ? (of ? Ffast : Fpub) + (size_t)(t + 1) * fs : nullptr;
            if (!wait16(own, P, of, ext, wait_ext, a.status, lflag, a.spin_ticks, a.poll_mask)) return;
        }
        if (t == T - 2 && Ffast) fast = group_on_one_xcd(xtab, P, lflag + 2);
        PL_ST(0);   // wait
        if (!dh_have) dh_bits = fetch_dh(t);   // its producer's flag was not known a step ago: the blocking wait has seen it now
        if (t + 1 < T) {
            // wide ingest: wave w sums the tiles of sources w * TPG .. (one wave instruction = one whole 1-KB tile), the four waves'
            // f32 sums meet in LDS; fixed order
            const bf16_t* xs = X + (size_t)((t + 1) & 1) * slot_stride + (size_t)p * P * TILE;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xs, (unsigned)(P * TILE * 2));
            uint4 pw[TPG];
#pragma unroll
            for (int i = 0; i < TPG; ++i) {
                const int src = wave * TPG + i;
                pw[i] = src < P ? ld16_handoff(rx, (unsigned)(src * TILE * 2 + lane * 16), fast) : make_uint4(0, 0, 0, 0);
            }
            asm volatile("" ::: "memory");   // the next step's operands behind the tile loads
            if (t > 0) {
                fetch_stash(t - 1);
                const bool ext1 = !ext_empty(ext_flags(a, R, g32, t - 1, p));
                n_dh_valid = !dhe || !ext1 || known_nxt;
                n_dh = (dhe && n_dh_valid) ? fetch_dh(t - 1) : 0u;
            }
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TPG; ++i) {
                float f[4];
                unpack_bf16x4(make_uint2(pw[i].x, pw[i].y), f);
                acc8[0] += f[0]; acc8[1] += f[1]; acc8[2] += f[2]; acc8[3] += f[3];
                unpack_bf16x4(make_uint2(pw[i].z, pw[i].w), f);
                acc8[4] += f[0]; acc8[5] += f[1]; acc8[6] += f[2]; acc8[7] += f[3];
            }
            float* redw = red + wave * (16 * 36);   // [wave][16 rows][32 + 4 pad] f32
            const int row = lane >> 2, c8 = lane & 3;
            *reinterpret_cast<float4*>(redw + row * 36 + c8 * 8) = make_float4(acc8[0], acc8[1], acc8[2], acc8[3]);
            *reinterpret_cast<float4*>(redw + row * 36 + c8 * 8 + 4) = make_float4(acc8[4], acc8[5], acc8[6], acc8[7]);
            __syncthreads();
        } else {
            if (t > 0) {
                fetch_stash(t - 1);
                const bool ext1 = !ext_empty(ext_flags(a, R, g32, t - 1, p));
                n_dh_valid = !dhe || !ext1 || known_nxt;
                n_dh = (dhe && n_dh_valid) ? fetch_dh(t - 1) : 0u;
            }
            if constexpr (GIMG) __syncthreads();   // (the gate image: every other step has the ingest's barrier)
        }
        PL_ST(1);   // ingest: tile loads, wave sums, barrier
        const unsigned char* gcell = gst + erow * L::GRS + jq * 4;
        unsigned u_g0 = d_g0, u_g1 = d_g1, u_g2 = d_g2, u_g3 = d_g3;
        if constexpr (GIMG) {
            u_g0 = *reinterpret_cast<const unsigned*>(gcell); u_g1 = *reinterpret_cast<const unsigned*>(gcell + 64);
            u_g2 = *reinterpret_cast<const unsigned*>(gcell + 128); u_g3 = *reinterpret_cast<const unsigned*>(gcell + 192);
        }
        float gi[2], gf[2], gg[2], go[2], c[2], cp[2], dh[2];
        unpack2(u_g0, gi);
        unpack2(u_g1, gf);
        unpack2(u_g2, gg);
        unpack2(u_g3, go);
        unpack2(u_c, c);
        unpack2(u_cp, cp);
        unpack2(dh_bits, dh);
        if (t + 1 < T) {
            const float* rd = red + erow * 36 + 2 * jq;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                dh[0] += rd[w * (16 * 36)];
                dh[1] += rd[w * (16 * 36) + 1];
            }
        }
        float dai[2], daf[2], dag[2], dao[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) cell_bwd(dh[u], dc_next[u], gi[u], gf[u], gg[u], go[u], c[u], cp[u], dai[u], daf[u], dag[u], dao[u], dc_next[u]);
        const unsigned pi = pk2(dai[0], dai[1]), pf = pk2(daf[0], daf[1]), pg = pk2(dag[0], dag[1]), po = pk2(dao[0], dao[1]);
        if (t == T - 1 && tid == 0 && xtab) flag_store(xtab + p, xcc_id_plus1());
        {   // dA_t overwrites the gate stash in place: read by a role of this launch (write-through) or by the dL/dCP product after it.
            // Issued FIRST: by the time of the flag these stores are a tile phase old and do not sit in front of the next poll
            // (round 5: as ONE 16-byte piece per lane from the image behind the barrier below: neutral, 13.43 / 13.34 against 13.36 / 13.37 ms at cfg5)
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG, (unsigned)(slabG * 2));
            const unsigned o = (unsigned)(((size_t)erow * G4 + j) * 2);
            if (dA_sc1) {
                __builtin_amdgcn_raw_buffer_store_b32(pi, rg, o, 0, kAuxSc1);
                __builtin_amdgcn_raw_buffer_store_b32(pf, rg, o + Hp * 2, 0, kAuxSc1);
                __builtin_amdgcn_raw_buffer_store_b32(pg, rg, o + 2 * Hp * 2, 0, kAuxSc1);
                __builtin_amdgcn_raw_buffer_store_b32(po, rg, o + 3 * Hp * 2, 0, kAuxSc1);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(pi, rg, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(pf, rg, o + Hp * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(pg, rg, o + 2 * Hp * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(po, rg, o + 3 * Hp * 2, 0, 0);
            }
        }
        {   // dA_t of this slice as the MFMA B operand: image [batch row][gate * 32 + unit]
            unsigned char* drow = da_img + erow * DRS + jq * 4;
            *reinterpret_cast<unsigned*>(drow) = pi;
            *reinterpret_cast<unsigned*>(drow + 64) = pf;
            *reinterpret_cast<unsigned*>(drow + 128) = pg;
            *reinterpret_cast<unsigned*>(drow + 192) = po;
        }
        // the look-ahead's answer, for everybody after the next barrier
        if (wave == 3 && lane == 0) lflag[3] = 0;
        if (wave == 3 && la_here) {
            const bool seen = __all(la_pv != 0);
            if (lane == 0) lflag[3] = seen ? 1 : 0;
        }
        __syncthreads();
        known_cur = known_nxt;          // for step t - 1
        known_nxt = lflag[3] != 0;      // for step t - 2
#ifdef F16_NO_LOOKAHEAD
        known_cur = known_nxt = false;
#endif
        PL_ST(2);   // cell + dA stores + dA image
        uint4 bfr[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) bfr[kc] = *reinterpret_cast<const uint4*>(da_img + lr * DRS + kc * 64 + kq * 16);
        const bool hand_fast = fast;   // this step's own hand-off form
        if constexpr (MEL) {   // before the recurrence's tiles: nothing but those then sits between the hand-off and the flag
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wmel[kc]), __builtin_bit_cast(bf16x8, bfr[kc]), acc, 0, 0, 0);
            // acc[r] = dmel partial[column 16 wave + 4 kq + r][batch lr]
            *reinterpret_cast<uint2*>(mel_img + lr * MRS + (16 * wave + 4 * kq) * 2) = pack_bf16x4(acc[0], acc[1], acc[2], acc[3]);
            bf16_t* xd = XM + (size_t)(t % kFusedRing) * mslot_stride + (size_t)p * TILE32;   // [tile of 32 columns][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)P + 1) * TILE32 * 2));
            if (lane < 32) {
                const int r = lane >> 1, hc = lane & 1;
                const uint4 v = *reinterpret_cast<const uint4*>(mel_img + r * MRS + (16 * wave + 8 * hc) * 2);
                st16_sc1(ro, (unsigned)(((size_t)(wave >> 1) * P * TILE32 + r * 32 + (wave & 1) * 16 + hc * 8) * 2), v);
            }
        }
        if (t > 0) {   // nobody consumes the recurrence's partials of step 0
            bf16_t* xd = X + (size_t)(t & 1) * slot_stride + (size_t)p * TILE;   // [dest][this source]
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(xd, (unsigned)(((size_t)(P - 1) * P + 1) * TILE * 2));
            // tiles in groups of TG: a group's 4 x TG MFMAs interleave over independent accumulators (one tile's four are a dependent
            // chain).  acc[r] = partial[n = 16 nt + 4 kq + r][batch lr] -> bf16 image [batch][n]; a 16 x 16 tile is its wave's alone:
            // read back by rows (a wave's LDS operations are ordered), 32 chunks of 16 bytes = the (nt & 1) half of destination
            // nt >> 1's [16][32] tile.  NO run-time branch anywhere between a tile's MFMAs and the reads of its accumulators: a wave's
            // surplus tile (46 tiles over 4 waves: the twelfth of waves 2 and 3) is computed like the others (its weight rows are
            // tile 0's) and only its store is dropped through the buffer range check.  With a branch around it the tile BEFORE it came
            // out with a stale accumulator register in one run of four (the plan was not reproducible run to run).  The cause,
            // from the ISA (profiles/r04_isa_stale_accumulator.txt): hipcc put the conditional branch directly behind the previous tile's last
            // MFMA and its target opened with the v_accvgpr_read of that MFMA's result -- one wait state on the taken path where gfx950 needs
            // eight (the compiler pads only what it sees in straight-line code).  tools/isa_mfma_hazard_scan.py checks every kernel for it
            // (tests/test_isa_hazards.py)
            constexpr int TG = F16_TG, NGRP = (NT + TG - 1) / TG;
            f32x4 accg[2][TG];
            auto tiles_mfma = [&](int grp, f32x4 (&acc)[TG]) {
#pragma unroll
                for (int u = 0; u < TG; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                    for (int u = 0; u < TG; ++u) {
                        const int i = grp * TG + u;
                        if (i < NT)   // compile time.  A wave's surplus tile (waves 2, 3 of 46 tiles) is COMPUTED too and only its store dropped
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[i < NT ? i : 0][kc]), __builtin_bit_cast(bf16x8, bfr[kc]), acc[u], 0, 0, 0);
                    }
            };
            auto tiles_out = [&](int grp, const f32x4 (&acc)[TG]) {
#if F16_DIRECT_STORE
                if (hand_fast) {   // same-XCD hand-off: the accumulator layout goes out as it is, 8 bytes per lane (a row's 32 bytes per tile contiguous)
#pragma unroll
                    for (int u = 0; u < TG; ++u) {
                        const int i = grp * TG + u, nt = wave + 4 * i;
                        if (i < NT) {
                            const uint2 v = pack_bf16x4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
                            u32x2 d;
                            d[0] = v.x; d[1] = v.y;
                            const bool valid = 4 * i + 3 < NTT || nt < NTT;
                            const unsigned off = valid ? (unsigned)(((size_t)(nt >> 1) * P * TILE + lr * 32 + (nt & 1) * 16 + 4 * kq) * 2) : kOob;
                            __builtin_amdgcn_raw_buffer_store_b64(d, ro, off, 0, 0);
                        }
                    }
                    return;
                }
#endif
#pragma unroll
                for (int u = 0; u < TG; ++u) {
                    const int i = grp * TG + u, nt = wave + 4 * i;
                    if (i < NT)
                        *reinterpret_cast<uint2*>(out_img + lr * ORS + (16 * nt + 4 * kq) * 2) = pack_bf16x4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
                }
#pragma unroll
                for (int u = 0; u < TG; ++u) {
                    const int i = grp * TG + u, nt = wave + 4 * i;
                    if (i < NT && lane < 32) {
                        const int r = lane >> 1, hc = lane & 1;
                        const uint4 v = *reinterpret_cast<const uint4*>(out_img + r * ORS + (16 * nt + 8 * hc) * 2);
                        u32x4 d;
                        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                        const bool valid = 4 * i + 3 < NTT || nt < NTT;
                        const unsigned off = valid ? (unsigned)(((size_t)(nt >> 1) * P * TILE + r * 32 + (nt & 1) * 16 + hc * 8) * 2) : kOob;
                        if (hand_fast) __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(d, ro, off, 0, kAuxSc1);
                    }
                }
            };
            if (F16_PIPE) tiles_mfma(0, accg[0]);
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                if (F16_PIPE) {
                    if (grp + 1 < NGRP) tiles_mfma(grp + 1, accg[(grp + 1) & 1]);
                    tiles_out(grp, accg[grp & 1]);
                } else {
                    tiles_mfma(grp, accg[0]);
                    tiles_out(grp, accg[0]);
                }
            }
        }
        PL_ST(3);   // tiles: MFMA + stores issued
        // Every store of this step (dA, input-gradient tiles, partial tiles) is older than nothing but itself at this point, and a
        // wave's stores complete in order: once they have drained, so have the write-through stores of step t + 1 -- the
        // write-through flag of step t + 1 goes up with the plain flag of step t, at a barrier the hand-off needs anyway.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (pub_pending) flag_store(Fpub + (size_t)(t + 1) * fs + p, 1);
            if (hand_fast && t > 0) {
                const __amdgpu_buffer_rsrc_t rf = make_rsrc(Ffast + (size_t)t * fs + p, 4u);
                __builtin_amdgcn_raw_buffer_store_b32(1u, rf, 0u, 0, 0);
            } else {
                flag_store(Fpub + (size_t)t * fs + p, 1);
            }
        }
        pub_pending = hand_fast && t > 0;
        PL_ST(4);   // drain + barrier + flag
    }
    PL_ST_DUMP(a.stamps);
}

}  // namespace
}  // namespace pl
