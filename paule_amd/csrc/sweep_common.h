// Device helpers shared by the persistent LSTM sweep kernels (lstm_persist.hip, lstm_persist_rs.hip):
// sc1 (write-through / L1-bypass) buffer accesses, bf16 packing, the bounded arrival wait and the publish step
// of the in-launch exchange, and the diagnostic stamp macros.
#pragma once
#include <type_traits>

#include "kernels.h"
#include "pl_types.h"

namespace pl {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kAuxSc1 = 16;                       // cache-policy bit sc1 of raw buffer loads / stores on gfx950

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxSc1);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st8_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, uint2 v) {
    u32x2 d;
    d[0] = v.x;
    d[1] = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, kAuxSc1);
}
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
    bf16x4 o;
    o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
    return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ void unpack_bf16x4(uint2 u, float (&f)[4]) {
    const bf16x4 v = __builtin_bit_cast(bf16x4, u);
    f[0] = (float)v[0]; f[1] = (float)v[1]; f[2] = (float)v[2]; f[3] = (float)v[3];
}

// ---- same-XCD fast path of the hand-off (a SPEED choice that is verified at run time) -----------------------------
// sc1 stores write through and DROP the line from the XCD's L2, so every consumer re-reads the bytes from the memory
// side (Infinity Cache): that fabric rate (~10 TB/s chip-wide) bounds the backward ingest, and every flag costs two
// memory-side round trips.  When all P workgroups of a batch group run on ONE XCD they share its L2: plain stores
// (which keep the line in that L2) and nt loads (which bypass only the reader's L1) are then coherent through the L2
// and served at L2 latency.  Placement is never assumed: every workgroup publishes the XCD it runs on
// (HW_REG_XCC_ID) with its first hand-off, the members compare after the first arrival wait, and only a group that
// found itself on one XCD switches; the first step and every other group keep the write-through form.
__device__ __forceinline__ int xcc_id_plus1() { return (int)__builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID[3:0] */) + 1; }

// call after the group's FIRST completed arrival wait (all members have stored their id); ends with a barrier
__device__ __forceinline__ bool group_on_one_xcd(const int* tab, int P, int* lds_flag) {
    if (threadIdx.x < 64) {
        const int mine = xcc_id_plus1();
        int v = mine;
        if ((int)threadIdx.x < P) v = __hip_atomic_load(tab + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool same = __all(v == mine) && P <= 64;
        if (threadIdx.x == 0) *lds_flag = same ? 1 : 0;
    }
    __syncthreads();
    const bool r = *lds_flag != 0;
    __syncthreads();
    return r;
}

constexpr int kAuxNt = 2;
__device__ __forceinline__ uint4 ld16_handoff(__amdgpu_buffer_rsrc_t r, unsigned off, bool same_xcd) {
    const u32x4 v = same_xcd ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxNt)
                             : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxSc1);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void st8_handoff(__amdgpu_buffer_rsrc_t r, unsigned off, uint2 v, bool plain) {
    u32x2 d;
    d[0] = v.x;
    d[1] = v.y;
    if (plain) __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, kAuxSc1);
}

// ---- arrival flags -------------------------------------------------------------------------------------------------
// One 4-byte flag per (group, step, producing workgroup), zeroed before every launch.  A producer sets its flag to 1
// after its hand-off stores were drained by every storing wave and the workgroup passed a barrier; a consumer polls
// all P flags of the step with ONE wave instruction (lane i reads flag i).  Write-through mode: sc1 flag store, sc1
// poll (hand-off table row 1 of the MI355X guide).  Same-XCD mode: plain store / nt poll, both served by the shared L2
// (an L2 round trip instead of two memory-side ones: the wait drops from ~1.2 to ~0.5 us per step).
// Bounded: on a timeout the status word is set and every workgroup leaves.  Ends with a barrier.
__device__ __forceinline__ bool wait_arrivals(const int* flags, int P, bool same_xcd, int* status, int* lds_flag,
                                              unsigned long long spin_ticks, unsigned poll_mask = 63u) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const __amdgpu_buffer_rsrc_t rf = make_rsrc(flags, (unsigned)(P * 4));
        int ok = 1;
        for (unsigned spin = 1;; ++spin) {
            int v = 1;
            if (lane < P)
                v = same_xcd ? (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, 2 /* nt */)
                             : (int)__builtin_amdgcn_raw_buffer_load_b32(rf, (unsigned)(lane * 4), 0, kAuxSc1);
            if (__all(v != 0)) break;
            // the abort / timeout check costs a second memory round trip: only every 64th poll
            if ((spin & poll_mask) == 0 &&
                (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                 __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks)) {
                ok = 0;
                break;
            }
        }
        if (lane == 0) {
            if (!ok) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *lds_flag = ok;
        }
    }
    __syncthreads();
    return *lds_flag != 0;
}

// every storing wave has drained its hand-off stores (all but its N youngest memory operations, which are not part
// of the hand-off); one lane raises the workgroup's flag
template <int N>
__device__ __forceinline__ void publish(int* flag, bool same_xcd) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const __amdgpu_buffer_rsrc_t rf = make_rsrc(flag, 4u);
        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b32(1u, rf, 0u, 0, 0);          // plain: stays in the shared L2
        else __builtin_amdgcn_raw_buffer_store_b32(1u, rf, 0u, 0, kAuxSc1);             // write-through
    }
}

// backward of one LSTM cell (backward-data): dA of the four gates and the running dL/dc handed to step t - 1.  The roundings
// are spelled out (which product the compiler fuses differs from kernel to kernel; lstm_persist_rs.hip and lstm_fused.hip are
// compared bit for bit).
__device__ __forceinline__ void cell_bwd(float dh, float dc_in, float gi, float gf, float gg, float go, float c, float cp,
                                         float& dai, float& daf, float& dag, float& dao, float& dc_out) {
    const float tc = tanh_fast(c);
    const float dc = __builtin_fmaf(dh * go, __builtin_fmaf(-tc, tc, 1.f), dc_in);
    dai = dc * gg * gi * (1.f - gi);
    daf = dc * cp * gf * (1.f - gf);
    dag = dc * gi * __builtin_fmaf(-gg, gg, 1.f);
    dao = dh * tc * go * (1.f - go);
    dc_out = dc * gf;
}

// ---- stash prefetchers (round 5) -----------------------------------------------------------------------------------------
// A cell-owning wave's vector-memory operations return in order, and the first flag poll of a backward step is issued behind the step's
// stash loads (gates, c_t, c_{t-1}, dL/dh from above: written by the forward pass milliseconds ago, so they come from HBM): the poll's
// answer -- and with it every load of handed-over bytes -- waits ~2 us for them however early the hand-off was there
// (profiles/r05_chain_stamps.txt: a workgroup that serves two groups in turn still spends 2.6 us per chain-step in "polls + tile loads").
// Moving the stash loads to other waves of the workgroup moves the delay to those waves' hand-off flags (round 4, A.9; round 5's
// in-workgroup touches: profiles/r05_ab_prefetchers.txt).  PREFETCHER workgroups on the CUs a sweep leaves idle do it instead: appended
// to the sweep's grid, workgroup q follows the groups of slot q % n_res -- the groups its blockIdx % 8 shares an XCD with under the
// observed dealing -- `dist` steps ahead of their hand-off flags and reads one dword of every 128-byte line of the stash rows of that
// step, so that the lines sit in that XCD's L2 when the cell waves ask for them.  Speed only: nothing waits for a prefetcher, a
// prefetcher waits for nobody longer than 20 us, and a launch without them computes the same bits.
// pace(g, u): returns once group g has handed over step u (bounded).  elt: bytes per stash element (2 bf16, 4 f32).
// chains: groups a workgroup of the sweep serves in turn per step (1: a workgroup takes its groups one after the other); a slot's
// prefetchers walk the (step, chain) pairs in the sweep's order.  When the slots of one group sit on 8 / n_res XCDs (n_res divides 8:
// blocks b, b + n_res, ... of the slot), the prefetchers of each of those XCDs share ALL lines among themselves.
template <typename Pace>
__device__ __forceinline__ void stash_prefetch_walk(const LstmSweepArgs& a, int Hp, int elt, int q, int n_res, int t_hi, int t_lo, Pace pace,
                                                    int chains = 1) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const int Bp = a.Bp, gs = a.group_rows, n_groups = (Bp + gs - 1) / gs;
    const int C = chains < 1 ? 1 : chains, n_sets = (n_groups + C - 1) / C;
    int per = a.n_pf / n_res;                 // prefetchers per slot
    const int s_first = q % n_res;
    int part = q / n_res;
    if (per < 1 || part >= per) return;
    const int xs = (n_res <= 8 && 8 % n_res == 0 && per % (8 / n_res) == 0) ? 8 / n_res : 1;   // XCDs a slot's workgroups sit on (observed dealing)
    part /= xs;
    per /= xs;
    const int D = a.pf_dist > 0 ? a.pf_dist : 4;
    const size_t rowG = (size_t)4 * Hp * elt, rowH = (size_t)Hp * elt;
    const size_t slabG = (size_t)Bp * rowG, slabH = (size_t)Bp * rowH;
    const unsigned char* G = static_cast<const unsigned char*>(a.G);
    const unsigned char* Cs = static_cast<const unsigned char*>(a.c);
    const unsigned char* dhe = static_cast<const unsigned char*>(a.dh_ext);
    unsigned acc = 0;
    for (int set = s_first; set < n_sets; set += n_res) {
        for (int t = t_hi; t >= t_lo; --t) {
            for (int c = 0; c < C; ++c) {
                const int g = set * C + c;
                if (g >= n_groups) break;
                const int r0 = gs * g, nr = (r0 + gs <= Bp ? gs : Bp - r0);
                const unsigned bytesG = (unsigned)(nr * rowG), bytesH = (unsigned)(nr * rowH);
                const unsigned linesG = (bytesG + 127) / 128, linesH = (bytesH + 127) / 128;
                const unsigned n_lines = linesG + linesH + (dhe ? linesH : 0u);
                if (t + D <= t_hi) pace(g, t + D);
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(G + (size_t)t * slabG + (size_t)r0 * rowG, bytesG);
                const __amdgpu_buffer_rsrc_t rc = make_rsrc(Cs + (size_t)t * slabH + (size_t)r0 * rowH, bytesH);
                const __amdgpu_buffer_rsrc_t rd = make_rsrc(dhe ? dhe + (size_t)t * slabH + (size_t)r0 * rowH : Cs, dhe ? bytesH : 0u);
                for (unsigned l = (unsigned)(part * nth + tid); l < n_lines; l += (unsigned)(per * nth)) {
                    if (l < linesG) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rg, l * 128u, 0, 0);
                    else if (l < linesG + linesH) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rc, (l - linesG) * 128u, 0, 0);
                    else acc ^= __builtin_amdgcn_raw_buffer_load_b32(rd, (l - linesG - linesH) * 128u, 0, 0);
                }
            }
        }
    }
    if (acc == 0x7fc01234u && a.status) __hip_atomic_fetch_or(a.status, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // keeps the loads alive, changes nothing
}
// pace on the per-workgroup arrival flags of the whole-workgroup hand-off (counters [group][step][flag_stride]): slice 0's flag of step u
struct PfPaceCounters {
    const int* counters; int T, flag_stride;
    __device__ __forceinline__ void operator()(int g, int u) const {
        const int* f = counters + ((size_t)g * T + u) * flag_stride;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000ull) break;   // 20 us without news (plain flags of a group on another XCD, an abandoned sweep): unpaced
            __builtin_amdgcn_s_sleep(16);
        }
    }
};

#ifdef PL_STAMPS
#define PL_ST(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); st_acc[i] += now_ - st_prev; st_prev = now_; } while (0)
#define PL_ST_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memrealtime();
#define PL_ST_DUMP(ptr) do { if ((ptr) && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) (ptr)[(size_t)blockIdx.x * 8 + i_] = st_acc[i_]; } } while (0)
#else
#define PL_ST(i) do { } while (0)
#define PL_ST_DECL
#define PL_ST_DUMP(ptr) do { } while (0)
#endif

}  // namespace pl
