// EXPERIMENT OF ROUND 3 -- NOT PART OF THE LIBRARY (not in paule_amd/csrc/Makefile; kept as the record of a measured dead end).
//
// The forward launch of lstm_fused.hip with 512-thread workgroups, two waves per SIMD: the K loop of a recurrence role cut in two and
// given to the two waves of a SIMD as the stages of a pipeline over chain-steps (front waves 0 .. 3: k-steps 0 .. KS/2-1 of chain-step m
// and its half of the h tile; back waves 4 .. 7: k-steps KS/2 .. KS-1 of chain-step m - 1, input columns, cell update, stores).  The
// accumulators cross from the front to the back wave through LDS as f32, so an output element still sees ONE accumulation chain in
// the order of lstm_fwd_sweep_kernel: the launch was BIT-IDENTICAL to the shipped forward (test_fused_forward_is_bit_identical passed on
// all four shapes with it) -- and slower everywhere (profiles/r03_fwd_pipeline_probe.txt): B = 256: 4.1 - 4.2 ms per launch against
// 2.2 ms (13.7 us per predictor step with four chains, 27 us with eight: ~3.4 us per pipeline iteration, as much as the one-wave-per-SIMD
// chain-step it was to undercut); B = 32 / 64 / 128: 7.4 / 10.0 / 13.7 us per step against 4.1.  Why: the stages advance in lock-step
// between two workgroup barriers per iteration, so (1) a group's step takes five to six iterations from tile to tile (load issue, front,
// back, stores, flag on its way, poll answer) where the old kernel overlaps the poll with its MFMAs and the tile load with its cell
// update; (2) the back stage carries everything but half of the MFMAs (its tile half, the input columns, the cell update, the output
// stores, the drain before the flag) and is the iteration's long pole while the front waves idle.  A workable form needs stages that
// hand over through LDS words instead of workgroup barriers and a finer split of the memory work; not attempted in round 3 (DESIGN.md,
// appendix A.4).  Builds against paule_amd/csrc/fused_common.h; the planner hooks that drove it were removed with it.
//
#include "fused_common.h"

namespace pl {

namespace {

constexpr int kMaxChains8 = 8;

// bounded blocking poll by ONE wave (no barrier inside); false: timed out / aborted (status word set)
__device__ __forceinline__ bool flags_spin(const FlagPoll& s, int lane, int* status, unsigned long long spin_ticks, unsigned poll_mask) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 1;; ++spin) {
        const int v = poll_load(s, lane);
        if (__all(v != 0)) return true;
        if ((spin & poll_mask) == 0 && (flag_load(status) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks)) {
            if (lane == 0) {
                int expected = 0;   // keep the first cause (a census failure stores 2)
                __hip_atomic_compare_exchange_strong((PL_GLOBAL int*)status, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
        }
    }
}

template <int KS, int KSX>
struct Lstm8Lds {
    static_assert(KS % 2 == 0, "the K loop is cut into two equal halves");
    static constexpr int KH = KS / 2;
    static constexpr int RS = KH * 32 + 16;        // half-image row stride: odd number of 16-byte chunks -> conflict-free b128 reads
    static constexpr int HRS = 64 + 16;            // outgoing tiles [32 rows][32 units] bf16
    static constexpr int GRS = 64 + 16;            // KSX = 0: a back wave's projection rows [32 rows][4 gates x 8 units] bf16
    static constexpr int O_A = 0;                  // front half of the h tile (columns 0 .. Hp/2-1)
    static constexpr int O_B = O_A + 32 * RS;      // back half
    static constexpr int O_ACC = O_B + 32 * RS;    // accumulators front -> back: [2 slots][4 waves][4][64 lanes] x 16 bytes
    static constexpr int O_HST = O_ACC + 2 * 4 * 4 * 64 * 16;
    static constexpr int O_GP = O_HST + 6 * 32 * HRS;
    static constexpr int O_CST = O_GP + (KSX ? 0 : 4 * 32 * GRS);
    static constexpr int O_FLAG = O_CST + kMaxChains8 * 256 * 16;
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, int KSX>
__device__ __forceinline__ void fused8_lstm_fwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = Lstm8Lds<KS, KSX>;
    constexpr int Hp = 16 * KS, G4 = 4 * Hp, ROWB = Hp * 2;
    constexpr int KH = L::KH, RS = L::RS, HRS = L::HRS, GRS = L::GRS;
    constexpr int CH = KH * 2;                        // 16-byte chunks per half row
    constexpr int NL = (32 * CH + 255) / 256;         // half-tile loads per thread
    constexpr int PF = KH < 4 ? KH : 4;               // B-fragment read-ahead (the SIMD's other wave covers the rest of the LDS latency)
    constexpr int DK = KH - 2 > 0 ? KH - 2 : 0;       // k-step at which the back waves that stored the h tile drain it (by then the store of P1 is acknowledged)
    constexpr int INP = KSX ? 16 * KSX : 16;
    unsigned char* imgA = lds + L::O_A;
    unsigned char* imgB = lds + L::O_B;
    unsigned char* accb = lds + L::O_ACC;
    unsigned char* hst = lds + L::O_HST;
    float4* cst = reinterpret_cast<float4*>(lds + L::O_CST);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);   // [0] abort, [1] arrivals of the draining waves

    const int tid = threadIdx.x, lane0 = tid & 63, wave = uni(tid >> 6);
    const bool front = wave < 4;
    const int w4 = wave & 3, htid0 = tid & 255;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const bf16_t* __restrict__ W = static_cast<const bf16_t*>(R.W);

    // weights -> registers: A-operand row (lane & 31) = gate (row >> 3), unit 32p + 8 w4 + (row & 7); this wave's half of the k-steps
    uint4 wreg[KH];
    const int bl0 = lane0 & 31, hh0 = lane0 >> 5;
    {
        const int bl = bl0, hh = hh0;
        const bf16_t* wrow = W + (size_t)((bl >> 3) * Hp + 32 * p + 8 * w4 + (bl & 7)) * Hp + 8 * hh + (front ? 0 : 16 * KH);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) { wreg[ks] = gld<uint4>(wrow + 16 * ks); pin(wreg[ks]); }
    }
    // front waves: the bias the accumulators start from (16 floats); back waves: their W_ih columns (KSX x 16 bytes) -- one array,
    // so that neither half pays registers for what only the other one uses
    uint4 aux[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if constexpr (KSX > 0) {
        if (!front) {
            const bf16_t* xrow = static_cast<const bf16_t*>(R.Wih) + (size_t)((bl0 >> 3) * Hp + 32 * p + 8 * w4 + (bl0 & 7)) * INP + 8 * hh0;
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) aux[ks] = gld<uint4>(xrow + 16 * ks);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float b4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = gld<float>(R.bias + q * Hp + 32 * p + 8 * w4 + 4 * hh0 + r);
                aux[q] = make_uint4(__builtin_bit_cast(unsigned, b4[0]), __builtin_bit_cast(unsigned, b4[1]), __builtin_bit_cast(unsigned, b4[2]), __builtin_bit_cast(unsigned, b4[3]));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) pin(aux[q]);
    }
    const size_t slabG = (size_t)Bp * G4, slabH = (size_t)Bp * Hp;
    bf16_t* __restrict__ G = static_cast<bf16_t*>(R.G);
    bf16_t* __restrict__ Hs = static_cast<bf16_t*>(R.h);
    bf16_t* __restrict__ Cs = static_cast<bf16_t*>(R.c);
    const bool src_sc1 = R.src_sc1 != 0;   // x / G rows come from a role of this launch: write-through loads
    const bf16_t* const x_in = static_cast<const bf16_t*>(R.x_in);
    const int g0 = set * RC;
    const int coff = front ? 0 : CH * 16;  // byte offset of this wave's half inside an h row
    unsigned char* const img = front ? imgA : imgB;

    // this wave's half of the h_{t2-1} tile of group g2: 32 rows x CH pieces of 16 bytes over the 256 threads of the stage
    uint4 hv[NL];
    auto issue_tile = [&](int g2, int t2, int htid) {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(Hs + (size_t)(t2 - 1) * slabH, (unsigned)(slabH * 2));
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = htid + 256 * i;
            const int row = q / CH, ch = q % CH;
            const int rb = 32 * g2 + row;
            hv[i] = ld16_sc1(rh, (q < 32 * CH && rb < Bp) ? (unsigned)(rb * ROWB + coff + ch * 16) : kOob);
        }
    };
    auto land_tile = [&](int htid) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = htid + 256 * i;
            if (w4 + 4 * i < KH) *reinterpret_cast<uint4*>(img + (q / CH) * RS + (q % CH) * 16) = hv[i];   // 32 CH = 64 KH pieces: whole waves
        }
    };
    // The pipeline: chain-steps m = t * Ca + c enter in order.  Stage L (front waves): the flags chain-step mL waits for are up ->
    // issue its front half-tile; F: front k-steps; K: back k-steps + cell; S: stores + flag.  A chain-step moves one stage per
    // iteration; when mL is not ready an empty slot moves instead (with fewer chains than stages a group's own previous step is
    // still in the pipeline: the slots stay empty until its flag is up -- nothing ever blocks inside an iteration).  All waves run
    // the same state machine; the one dynamic input, "mL is ready", is decided by wave 0 and handed round through LDS.
    auto next_ct = [&](int& c, int& t) { if (++c == Ca) { c = 0; ++t; } };
    PL_ST_DECL
    int cL = 0, tL = 0;                       // chain-step mL
    int cF = 0, tF = 0, cK = 0, tK = 0, cS = 0, tS = 0;
    bool vF = false, vK = false, vS = false;
    int pv_a = 0, pv_b = 0;                   // wave 0: answers of the two looks at mL's flags taken in the iteration before
    bool have_poll = false;
    unsigned idle = 0;
    unsigned long long idle_t0 = 0;
    if (tid == 0) { lflag[0] = 0; lflag[1] = 0; lflag[2] = 0; }
    __syncthreads();

    while (tL < T || vF || vK || vS) {
        const bool l_on = tL < T;
        // per-iteration copies of the thread coordinates the compiler cannot see through: left alone it hoists every address that
        // depends only on them out of the loop (~50 registers), and the weights spill to scratch instead
        int lane = lane0, htid = htid0;
        asm volatile("" : "+v"(lane), "+v"(htid));
        const int bl = lane & 31, hh = lane >> 5;
        // ---------------- P1 ----------------
        if (front) {
            if (vF && tF > 0) land_tile(htid);
            if (wave == 0) {
                int rdy = 0, abort_ = 0;
                if (l_on) {
                    const FlagPoll s1 = step_flags(a, WT_, g0 + cL, tL, p);
                    if (poll_empty(s1)) rdy = 1;
                    else if (have_poll) rdy = (__all(pv_a != 0) || __all(pv_b != 0)) ? 1 : 0;
                }
                if (!rdy && !vF && !vK && !vS) {   // nothing in flight: this is a wait for another workgroup -- bounded
                    if (idle == 0) idle_t0 = __builtin_amdgcn_s_memrealtime();
                    if ((++idle & a.poll_mask) == 0 && (uni(flag_load(a.status)) != 0 || __builtin_amdgcn_s_memrealtime() - idle_t0 > a.spin_ticks)) {
                        abort_ = 1;
                        if (lane == 0) {
                            int expected = 0;
                            __hip_atomic_compare_exchange_strong((PL_GLOBAL int*)a.status, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                } else {
                    idle = 0;
                }
                if (lane == 0) { lflag[2] = rdy; if (abort_) lflag[0] = 1; }
            }
        } else {
            if (vS) {   // outputs of the chain-step in stage S: the h tile (hand-off, write-through) first, then the five stash arrays
                const int g = g0 + cS;
                if (w4 < 2) {
                    const int row = htid >> 2, qt = htid & 3;
                    const int rb = 32 * g + row;
                    const uint4 hvv = *reinterpret_cast<const uint4*>(hst + row * HRS + qt * 16);
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(Hs + (size_t)tS * slabH, (unsigned)(slabH * 2));
                    st16_sc1(ro, rb < Bp ? (unsigned)((rb * Hp + 32 * p + 8 * qt) * 2) : kOob, hvv);
                }
                asm volatile("" ::: "memory");   // keep the stash stores behind it
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int e = htid + 256 * k;   // piece: array e / 128, row (e % 128) / 4, quarter e % 4
                    if (k < 2 || w4 < 2) {          // 640 pieces: all back threads twice, two waves a third time
                        const int arr = e >> 7, row = (e & 127) >> 2, qt = e & 3, rb = 32 * g + row;
                        const uint4 sv = *reinterpret_cast<const uint4*>(hst + (arr + 1) * 32 * HRS + row * HRS + qt * 16);
                        u32x4 d;
                        d[0] = sv.x; d[1] = sv.y; d[2] = sv.z; d[3] = sv.w;
                        if (k < 2) {
                            const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)tS * slabG, (unsigned)(slabG * 2));
                            __builtin_amdgcn_raw_buffer_store_b128(d, rg, rb < Bp ? (unsigned)(((size_t)rb * G4 + arr * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
                        } else {
                            const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cs + (size_t)tS * slabH, (unsigned)(slabH * 2));
                            __builtin_amdgcn_raw_buffer_store_b128(d, rc, rb < Bp ? (unsigned)(((size_t)rb * Hp + 32 * p + 8 * qt) * 2) : kOob, 0, 0);
                        }
                    }
                }
            }
            if (vK && tK > 0) land_tile(htid);
        }
        __syncthreads();
        if (uni(lflag[0]) != 0) return;       // (uniform values, and the compiler should know: scalar branches)
        const bool rdy = uni(lflag[2]) != 0;
        PL_ST(0);   // P1

        // ---------------- P2 ----------------
        const int cN = cL, tN = tL;   // the chain-step that enters stage F next iteration (if rdy)
        if (rdy) next_ct(cL, tL);
        if (front) {
            if (rdy && tN > 0) issue_tile(g0 + cN, tN, htid);
            FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
            const bool poll_here = wave == 0 && tL < T;   // first looks at the flags of the (new) chain-step mL
            if (poll_here) pn = step_flags(a, WT_, g0 + cL, tL, p);
            have_poll = poll_here;
            pv_a = pv_b = 0;
            if (poll_here) pv_a = poll_load(pn, lane);   // first look now, second one behind the front's work of this iteration
            if (vF) {
                f32x16 acc;
                if constexpr (KSX > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[4 * q] = __builtin_bit_cast(float, aux[q].x); acc[4 * q + 1] = __builtin_bit_cast(float, aux[q].y);
                        acc[4 * q + 2] = __builtin_bit_cast(float, aux[q].z); acc[4 * q + 3] = __builtin_bit_cast(float, aux[q].w);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tF > 0) {
                    const unsigned char* bsrc = imgA + bl * RS + hh * 16;
                    uint4 bq[PF];
#pragma unroll
                    for (int k = 0; k < PF; ++k) bq[k] = *reinterpret_cast<const uint4*>(bsrc + k * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks]), __builtin_bit_cast(bf16x8, bq[ks % PF]), acc, 0, 0, 0);
                        if (ks + PF < KH) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // accumulators -> LDS (lane-contiguous 16-byte pieces: conflict-free); the slot alternates with the parity of the chain-step
                unsigned char* ad = accb + ((((cF + tF * Ca) & 1) * 4 + w4) * 4) * 1024 + lane * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(ad + q * 1024) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
            if (poll_here) pv_b = poll_load(pn, lane);   // second look, behind the front's work of this iteration
            PL_ST(1);   // front P2
        } else {
            // loads: the back half-tile of the chain-step now in stage F (it is in stage K next iteration), then this stage's input
            // columns / projection rows
            if (vF && tF > 0) issue_tile(g0 + cF, tF, htid);
            uint4 xb[KSX ? KSX : 1];
            uint4 gv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            const int g = g0 + cK;
            if (vK) {
                if constexpr (KSX > 0) {
                    // B fragments straight from the x row: lane (bl, hh) holds elements 16 ks + 8 hh .. + 7 of row bl
                    const int rb = 32 * g + bl;
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x_in + (size_t)tK * Bp * INP, (unsigned)((size_t)Bp * INP * 2));
#pragma unroll
                    for (int ks = 0; ks < KSX; ++ks) {
                        const unsigned off = rb < Bp ? (unsigned)((rb * INP + 16 * ks + 8 * hh) * 2) : kOob;
                        if (src_sc1) xb[ks] = ld16_sc1(rx, off);
                        else { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0); xb[ks] = make_uint4(v[0], v[1], v[2], v[3]); }
                    }
                } else {
                    // this wave's units of the projection rows: 32 rows x 4 gates x 16 bytes = 128 pieces, two per lane
                    const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)tK * slabG, (unsigned)(slabG * 2));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = lane + 64 * q, row = e >> 2, gate = e & 3;
                        const int rb = 32 * g + row;
                        const unsigned off = rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * w4) * 2) : kOob;
                        if (src_sc1) gv[q] = ld16_sc1(rg, off);
                        else { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0); gv[q] = make_uint4(v[0], v[1], v[2], v[3]); }
                    }
                }
            }
            // the two waves that stored the h tile of the chain-step in stage S (P1 above) drain it; the later one raises its flag
            // (vmcnt retires in order: wait for all but the operations issued after that store -- 3 stash stores, the loads above)
            const bool drainer = vS && w4 < 2;
            constexpr int NX = KSX ? KSX : 2;
            const int younger = 3 + ((vF && tF > 0) ? NL : 0) + (vK ? NX : 0);
            auto drain_and_raise = [&]() {
                if (younger == 3 + NL + NX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + NL + NX) : "memory");
                else if (younger == 3 + NL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + NL) : "memory");
                else if (younger == 3 + NX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + NX) : "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                if (lane == 0) {
                    const int old = uni(__hip_atomic_fetch_add(lflag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));   // a scalar branch below
                    if (old & 1) flag_store(rflags + ((size_t)(g0 + cS) * T + tS) * a.flag_stride + p, 1);
                }
            };
            if (vK) {
                f32x16 acc;
                {
                    const unsigned char* as = accb + ((((cK + tK * Ca) & 1) * 4 + w4) * 4) * 1024 + lane * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(as + q * 1024);
                        acc[4 * q] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
                    }
                }
                bool drained = false;
                if (tK > 0) {
                    const unsigned char* bsrc = imgB + bl * RS + hh * 16;
                    uint4 bq[PF];
#pragma unroll
                    for (int k = 0; k < PF; ++k) bq[k] = *reinterpret_cast<const uint4*>(bsrc + k * 32);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks]), __builtin_bit_cast(bf16x8, bq[ks % PF]), acc, 0, 0, 0);
                        if (ks + PF < KH) bq[ks % PF] = *reinterpret_cast<const uint4*>(bsrc + (ks + PF) * 32);
                        if (ks == DK && drainer) drain_and_raise();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    drained = true;
                }
                if (drainer && !drained) drain_and_raise();
                if constexpr (KSX > 0) {
#pragma unroll
                    for (int ks = 0; ks < KSX; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aux[ks]), __builtin_bit_cast(bf16x8, xb[ks]), acc, 0, 0, 0);
                }
                float gxi[4] = {0.f, 0.f, 0.f, 0.f}, gxf[4] = {0.f, 0.f, 0.f, 0.f}, gxg[4] = {0.f, 0.f, 0.f, 0.f}, gxo[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (KSX == 0) {
                    // redistribute inside the wave: pieces -> [row][gate][8 units], each lane takes the 4 units of its cells
                    unsigned char* gp = lds + L::O_GP + w4 * 32 * GRS;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = lane + 64 * q, row = e >> 2, gate = e & 3;
                        *reinterpret_cast<uint4*>(gp + row * GRS + gate * 16) = gv[q];
                    }
                    const unsigned char* gsrc = gp + bl * GRS + hh * 8;
                    unpack_bf16x4(*reinterpret_cast<const uint2*>(gsrc), gxi);
                    unpack_bf16x4(*reinterpret_cast<const uint2*>(gsrc + 16), gxf);
                    unpack_bf16x4(*reinterpret_cast<const uint2*>(gsrc + 32), gxg);
                    unpack_bf16x4(*reinterpret_cast<const uint2*>(gsrc + 48), gxo);
                }
                // cell update: acc[4 * gate + unit]
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tK > 0) cs = cst[cK * 256 + htid];
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
                cst[cK * 256 + htid] = make_float4(c_state[0], c_state[1], c_state[2], c_state[3]);
                unsigned char* o = hst + bl * HRS + (8 * w4 + 4 * hh) * 2;
                *reinterpret_cast<uint2*>(o) = pack_bf16x4(vh[0], vh[1], vh[2], vh[3]);
                *reinterpret_cast<uint2*>(o + 32 * HRS) = pack_bf16x4(vi[0], vi[1], vi[2], vi[3]);
                *reinterpret_cast<uint2*>(o + 2 * 32 * HRS) = pack_bf16x4(vf[0], vf[1], vf[2], vf[3]);
                *reinterpret_cast<uint2*>(o + 3 * 32 * HRS) = pack_bf16x4(vg[0], vg[1], vg[2], vg[3]);
                *reinterpret_cast<uint2*>(o + 4 * 32 * HRS) = pack_bf16x4(vo[0], vo[1], vo[2], vo[3]);
                *reinterpret_cast<uint2*>(o + 5 * 32 * HRS) = pack_bf16x4(vc[0], vc[1], vc[2], vc[3]);
            } else if (drainer) {
                drain_and_raise();
            }
        }
        __syncthreads();
        // every chain-step in flight moves one stage on
        vS = vK; cS = cK; tS = tK;
        vK = vF; cK = cF; tK = tF;
        vF = rdy; cF = cN; tF = tN;
    }
    PL_ST_DUMP(a.stamps);
}

// ---------------------------------------------------------------------------------------------------------------------
// row-tile products on a producing layer's h (arithmetic of gemm_nt_kernel, as fused_gemm_fwd of lstm_fused.hip): the input
// projection of the layer above (PROJ: 128 gate columns of slice p, one 16-column tile per wave) and the mel head with its
// pooling (HEAD: 64 mel columns x 32 rows, one 16 x 16 tile per wave).  An output element is computed by one wave with the same
// sequence of v_mfma_f32_16x16x32_bf16 as before: same bits.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
struct Gemm8Lds {
    static constexpr int Hp = 16 * KS;
    static constexpr int CH = Hp / 8;
    static constexpr int IRS = (CH + ((CH % 4 == 0) ? 2 : 0)) * 16;   // chunk stride = 2 mod 4: conflict-free for (row lr, chunk 4n + kq) reads
    static constexpr int ORS = 128 * 2 + 16;
    static constexpr int O_IMG = 0;
    static constexpr int O_OST = O_IMG + 32 * IRS;
    static constexpr int O_YB = O_OST + 32 * ORS;
    static constexpr int O_FLAG = O_YB + kMaxChains8 * 512 * 16;
    static constexpr int BYTES = O_FLAG + 64;
};

template <int KS, bool HEAD>
__device__ __forceinline__ void fused8_gemm_fwd(const FusedArgs& a, const FusedRole& R, const int set, const int p, unsigned char* lds) {
    using L = Gemm8Lds<KS>;
    constexpr int Hp = 16 * KS, KB = KS / 2, CH = L::CH, IRS = L::IRS, ORS = L::ORS, ROWB = Hp * 2;
    constexpr int NL = (32 * CH + 511) / 512;
    constexpr int NI = HEAD ? 1 : 2;                  // row tiles of 16 per wave
    constexpr int PF = KB < 4 ? KB : 4;
    unsigned char* img = lds + L::O_IMG;
    unsigned char* ost = lds + L::O_OST;
    float4* yb = reinterpret_cast<float4*>(lds + L::O_YB);
    int* lflag = reinterpret_cast<int*>(lds + L::O_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int Bp = a.Bp, T = R.T, RC = R.C;
    int* const rflags = R.flags;
    const Waits WT_{R.wait[0], R.wait[1], R.wait[2]};
    int Ca = a.n_groups - set * RC;
    Ca = Ca < RC ? Ca : RC;
    if (Ca <= 0) return;
    const int G4 = 4 * Hp;   // PROJ: gate columns of the consuming layer (same hidden size)
    const bf16_t* __restrict__ Wg = static_cast<const bf16_t*>(R.Wg);
    // PROJ: wave -> gate (wave & 3), 16-column half (wave >> 2) of the slice's 32 units; HEAD: wave -> mel columns 16 (wave & 3), rows 16 (wave >> 2)
    const int wq = wave & 3, wh = wave >> 2;
    const int col = HEAD ? 16 * wq + lr : wq * Hp + 32 * p + 16 * wh + lr;
    uint4 wreg[KB];
    {
        const bf16_t* wrow = Wg + (size_t)col * Hp + 8 * kq;
#pragma unroll
        for (int n = 0; n < KB; ++n) { wreg[n] = gld<uint4>(wrow + 32 * n); pin(wreg[n]); }
    }
    const float bias_v = R.bias ? gld<float>(R.bias + col) : 0.f;
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
            const int q = tid + 512 * i;
            const int row = q / CH, ch = q % CH;
            const int rb = 32 * g2 + row;
            hv[i] = ld16_sc1(rh, (q < 32 * CH && rb < Bp) ? (unsigned)(rb * ROWB + ch * 16) : kOob);
        }
    };

    PL_ST_DECL
    int c = 0, t = 0;
    {
        const FlagPoll s0 = step_flags(a, WT_, set * RC, 0, p);
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
            const int q = tid + 512 * i;
            if (q < 32 * CH) *reinterpret_cast<uint4*>(img + (q / CH) * IRS + (q % CH) * 16) = hv[i];
        }
        __syncthreads();
        FlagPoll pn{nullptr, 0, nullptr, 0, nullptr};
        if (has_next) pn = step_flags(a, WT_, gn, tn, p);
        int pv = 1;
        const bool poll_here = wave == 0 && has_next;
        __builtin_amdgcn_sched_barrier(0);
        PL_ST(0);

        f32x4 acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // HEAD: this wave's row tile wh; PROJ: both row tiles
            const unsigned char* a0 = img + ((HEAD ? 16 * wh : 0) + lr) * IRS + kq * 16;
            const unsigned char* a1 = img + (16 + lr) * IRS + kq * 16;
            uint4 f0[PF], f1[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                f0[i] = *reinterpret_cast<const uint4*>(a0 + i * 64);
                if constexpr (!HEAD) f1[i] = *reinterpret_cast<const uint4*>(a1 + i * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < KB; ++n) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0[n % PF]), __builtin_bit_cast(bf16x8, wreg[n]), acc[0], 0, 0, 0);
                if constexpr (!HEAD)
                    acc[NI - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1[n % PF]), __builtin_bit_cast(bf16x8, wreg[n]), acc[NI - 1], 0, 0, 0);
                if (n + PF < KB) {
                    f0[n % PF] = *reinterpret_cast<const uint4*>(a0 + (n + PF) * 64);
                    if constexpr (!HEAD) f1[n % PF] = *reinterpret_cast<const uint4*>(a1 + (n + PF) * 64);
                }
                if (n == KB / 2 && poll_here) pv = poll_load(pn, lane);
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

        // epilogue: D[row 16 i + 4 kq + r][column of this wave's tile + lr]
        if constexpr (!HEAD) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<bf16_t*>(ost + (16 * i + 4 * kq + r) * ORS + (32 * wq + 16 * wh + lr) * 2) = (bf16_t)(acc[i][r] + bias_v);
            __syncthreads();
            bf16_t* Gout = static_cast<bf16_t*>(out_ptr);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(Gout + (size_t)t * Bp * G4, (unsigned)((size_t)Bp * G4 * 2));
            {
                const int e = tid, row = e >> 4, gate = (e >> 2) & 3, q4 = e & 3;
                const int rb = 32 * g + row;
                const uint4 v = *reinterpret_cast<const uint4*>(ost + row * ORS + (32 * gate + 8 * q4) * 2);
                st16_sc1(ro, rb < Bp ? (unsigned)(((size_t)rb * G4 + gate * Hp + 32 * p + 8 * q4) * 2) : kOob, v);
            }
            raise_flag<0>(rflags + ((size_t)g * T + t) * a.flag_stride + p);
        } else {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc[0][r] + bias_v;
            if ((t & 1) == 0) {   // even frame: kept for its partner
                yb[c * 512 + tid] = make_float4(y[0], y[1], y[2], y[3]);
                __syncthreads();   // the image is rewritten at the top of the next chain-step
            } else {
                const float4 e0 = yb[c * 512 + tid];
                const float ye[4] = {e0.x, e0.y, e0.z, e0.w};
                const int tp = t >> 1, Tp = T >> 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * wh + 4 * kq + r, bb = 32 * g + row;
                    const bool live = bb < a.B && col < out_dim;
                    const float v = live ? 0.5f * (ye[r] + y[r]) : 0.f;
                    const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(out_bm, (unsigned)((size_t)a.B * Tp * out_dim * 4));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rb_, live ? (unsigned)((((size_t)bb * Tp + tp) * out_dim + col) * 4) : kOob, 0, 0);
                    *reinterpret_cast<bf16_t*>(ost + row * ORS + col * 2) = (bf16_t)v;
                }
                __syncthreads();
                if (wave < 4) {   // pooled frame, time-major activation [tp][Bp][64]: the input of the embedder's first layer (hand-off)
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

template <int KS>
constexpr int fused8_lds_bytes() {
    int m = Lstm8Lds<KS, 0>::BYTES;
    m = m > Lstm8Lds<KS, 2>::BYTES ? m : Lstm8Lds<KS, 2>::BYTES;
    m = m > Lstm8Lds<KS, 4>::BYTES ? m : Lstm8Lds<KS, 4>::BYTES;
    m = m > Gemm8Lds<KS>::BYTES ? m : Gemm8Lds<KS>::BYTES;
    return (m + 15) / 16 * 16;
}

template <int KS>
__global__ __launch_bounds__(512, 1) void fused_fwd8_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[fused8_lds_bytes<KS>()];
    if ((int)blockIdx.x >= a.grid) return;
    const PL_GLOBAL short* bt = (const PL_GLOBAL short*)(a.block_tab + 4 * blockIdx.x);
    const int role = __builtin_amdgcn_readfirstlane((int)bt[0]), set = __builtin_amdgcn_readfirstlane((int)bt[1]),
              p = __builtin_amdgcn_readfirstlane((int)bt[2]);
    if (role < 0 || role >= a.n_roles) return;
    if (a.census && !census_ok(a, reinterpret_cast<int*>(lds))) return;
    __syncthreads();
    const FusedRole R = uniform_role(a.roles[role]);
    switch (R.type) {
        case FR_LSTM_FWD:
            if (R.ksx == 2) fused8_lstm_fwd<KS, 2>(a, R, set, p, lds);
            else if (R.ksx == 4) fused8_lstm_fwd<KS, 4>(a, R, set, p, lds);
            else fused8_lstm_fwd<KS, 0>(a, R, set, p, lds);
            break;
        case FR_PROJ_FWD: fused8_gemm_fwd<KS, false>(a, R, set, p, lds); break;
        case FR_HEAD_FWD: fused8_gemm_fwd<KS, true>(a, R, set, p, lds); break;
        default: break;
    }
}

}  // namespace

#define PL_FUSED8_KS_LIST(X) X(6) X(46)

bool fused8_supported(int Hp) {
#define PL_CASE(K) if (Hp == 16 * K) return true;
    PL_FUSED8_KS_LIST(PL_CASE)
#undef PL_CASE
    return false;
}

int fused8_max_chains() { return kMaxChains8; }

void launch_fused_fwd8(hipStream_t stream, int Hp, const FusedArgs& a) {
#define PL_CASE(K)                                                                              \
    if (Hp == 16 * K) {                                                                         \
        hipLaunchKernelGGL(fused_fwd8_kernel<K>, dim3(a.grid), dim3(512), 0, stream, a);        \
        return;                                                                                 \
    }
    PL_FUSED8_KS_LIST(PL_CASE)
#undef PL_CASE
}

}  // namespace pl
