// Device helpers of the role-fused launches (lstm_fused.hip; also included by the experiments under tools/experiments/):
// global-address-space accessors, flag polls, bounded waits, the residency census and the uniform copy of a role descriptor.
#pragma once
#include "sweep_common.h"

namespace pl {
namespace {

// Wave-granular conditions are written as SCALAR branches on the readfirstlane'd wave index, never as `tid < 128`: the kernels
// spill SGPRs (to VGPR lanes), and the compiler placed reloads inside `tid < N` blocks that whole waves skip (s_cbranch_execz)
// -- those waves then ran on with stale SGPRs (wrong LDS addresses: gates saturated in the columns of waves 2 and 3).  Row
// guards of ragged groups go through the buffer range check instead of a branch (offset out of range: loads return 0, stores
// are dropped).
constexpr unsigned kOob = 0x80000000u;

// Every pointer of a role comes out of the descriptor table, so the compiler cannot tell its address space and would use FLAT
// instructions -- and while a FLAT access is pending, every LDS wait (each __syncthreads) becomes s_waitcnt vmcnt(0): the
// prefetched tiles and the hand-off stores in flight would be drained at every barrier.  All global accesses of the roles
// therefore go through these global-address-space casts (or raw buffer operations).
#define PL_GLOBAL __attribute__((address_space(1)))
// keeps a loaded value in its register: under pressure the compiler otherwise REMATERIALIZES loop-invariant loads -- it would
// fetch the weights again in every chain-step instead of holding them
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <typename V> struct GAcc;   // HIP's vector classes have no address-space-qualified copies: go through the native vectors
template <> struct GAcc<uint4> {
    template <typename T> static __device__ __forceinline__ uint4 ld(const T* q) { const u32x4 v = *(const PL_GLOBAL u32x4*)(q); return make_uint4(v[0], v[1], v[2], v[3]); }
    template <typename T> static __device__ __forceinline__ void st(T* q, uint4 v) { u32x4 d; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; *(PL_GLOBAL u32x4*)(q) = d; }
};
template <> struct GAcc<uint2> {
    template <typename T> static __device__ __forceinline__ uint2 ld(const T* q) { const u32x2 v = *(const PL_GLOBAL u32x2*)(q); return make_uint2(v[0], v[1]); }
    template <typename T> static __device__ __forceinline__ void st(T* q, uint2 v) { u32x2 d; d[0] = v.x; d[1] = v.y; *(PL_GLOBAL u32x2*)(q) = d; }
};
template <> struct GAcc<float4> {
    template <typename T> static __device__ __forceinline__ float4 ld(const T* q) { const f32x4v v = *(const PL_GLOBAL f32x4v*)(q); return make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct GAcc<float> {
    template <typename T> static __device__ __forceinline__ float ld(const T* q) { return *(const PL_GLOBAL float*)(q); }
    template <typename T> static __device__ __forceinline__ void st(T* q, float v) { *(PL_GLOBAL float*)(q) = v; }
};
template <typename V, typename T>
__device__ __forceinline__ V gld(const T* ptr) { return GAcc<V>::ld(ptr); }
template <typename V, typename T>
__device__ __forceinline__ void gst(T* ptr, V v) { GAcc<V>::st(ptr, v); }
__device__ __forceinline__ int flag_load(const int* ptr) {
    return __hip_atomic_load((const PL_GLOBAL int*)ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(int* ptr, int v) {
    __hip_atomic_store((PL_GLOBAL int*)ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// flags that stay inside ONE XCD (a role's own exchange once its workgroups have verified that they share one): stored plainly, they
// sit in that XCD's L2; polled with nt loads (bypass only the polling CU's L1) they are answered by that L2 instead of the memory side
__device__ __forceinline__ int flag_load_nt(const int* ptr) {
    asm volatile("" ::: "memory");   // a poll loop must issue the load every time round
    return __builtin_nontemporal_load((const PL_GLOBAL int*)ptr);
}
__device__ __forceinline__ void flag_store_plain(int* ptr, int v) {   // a buffer store without cache-policy bits (a volatile store would be written through)
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, make_rsrc(ptr, 4u), 0u, 0, 0);
}

// what one chain-step waits for, resolved to addresses: lanes 0 .. na-1 read fa[lane], lanes 32 .. 32+nb-1 read fb[lane - 32],
// lane 63 reads fc
struct FlagPoll {
    const int* fa;
    int na;
    const int* fb;
    int nb;
    const int* fc;
    int fa_nt;   // 1: fa is a same-XCD (plain) flag set: nt loads
};

__device__ __forceinline__ int poll_load(const FlagPoll& s, int lane) {
    int v = 1;
    if (lane < s.na) v = s.fa_nt ? flag_load_nt(s.fa + lane) : flag_load(s.fa + lane);
    else if (lane >= 32 && lane - 32 < s.nb) v = flag_load(s.fb + (lane - 32));
    else if (lane == 63 && s.fc) v = flag_load(s.fc);
    return v;
}

// blocking, bounded; wave 0 polls, everybody meets at the barrier.  false: timed out / aborted (uniform over the workgroup)
__device__ __forceinline__ bool flags_wait(const FlagPoll& s, int* status, int* lds_word, unsigned long long spin_ticks, unsigned poll_mask) {
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {   // wave 0, as a scalar branch
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        for (unsigned spin = 1;; ++spin) {
            const int v = poll_load(s, lane);
            if (__all(v != 0)) break;
            if ((spin & poll_mask) == 0 && (flag_load(status) != 0 ||
                                            __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks)) {
                ok = 0;
                break;
            }
        }
        if (lane == 0) {
            if (!ok) {   // 0 -> 1 only: an earlier cause (the census's 2) stays
                int expected = 0;
                __hip_atomic_compare_exchange_strong((PL_GLOBAL int*)status, &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *lds_word = ok;
        }
    }
    __syncthreads();
    return *lds_word != 0;
}

// every storing wave has drained (all but its N youngest memory operations); one lane raises the flag, write-through
template <int N>
__device__ __forceinline__ void raise_flag(int* flag) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {   // wave 0 by a scalar branch (see the note on wave-granular conditions)
        if ((threadIdx.x & 63) == 0) flag_store(flag, 1);
    }
}

// Residency census at the top of a fused launch: the roles wait for each other inside the launch, so all role-bearing workgroups
// have to be resident together.  Every one signs in and waits for the others with a SHORT bound; if they do not all show up
// (a second process on the GPU holds CUs: the launches of one process are chained) the launch gives up at once with status 2.
__device__ __forceinline__ bool census_ok(const FusedArgs& a, int* lds_word) {
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        const int lane = threadIdx.x;
        int ok = 1;
        if (a.census_late_ticks && blockIdx.x == 0) {   // test hook: this workgroup shows up only after the others have given up
            const unsigned long long tl = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - tl < a.census_late_ticks) __builtin_amdgcn_s_sleep(32);
        }
        if (lane == 0) __hip_atomic_fetch_add((PL_GLOBAL int*)a.census, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 1;; ++spin) {
            const int v = flag_load(a.census);
            // the others may have given up already (status set) while the count still completes with this late sign-in: the
            // launch is void either way, and the FIRST cause must stay in the status word
            if (v >= a.n_active) {
                if (flag_load(a.status) != 0) ok = 0;
                break;
            }
            if ((spin & 15u) == 0 && (flag_load(a.status) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > a.census_ticks)) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) {
            if (!ok) {
                int expected = 0;
                __hip_atomic_compare_exchange_strong((PL_GLOBAL int*)a.status, &expected, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *lds_word = ok;
        }
    }
    __syncthreads();
    return *lds_word != 0;
}

// The role descriptors live in device memory and are read with vector loads, so the compiler takes every value in them for
// divergent: buffer descriptors built from such pointers get a waterfall loop per access, and loop bounds land in VGPRs.
// One readfirstlane per field, once per role, puts them where kernel arguments would be: in SGPRs.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uni(T* ptr) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ FusedWait uni(const FusedWait& w) {
    return FusedWait{uni(w.flags), uni(w.T), uni(w.n), uni(w.per_p), uni(w.t_shr), uni(w.t_add)};
}
__device__ __forceinline__ FusedRole uniform_role(const FusedRole& g) {
    FusedRole r;
    r.type = uni(g.type); r.ksx = uni(g.ksx); r.wide = uni(g.wide); r.C = uni(g.C); r.T = uni(g.T); r.flags = uni(g.flags); r.flags2 = uni(g.flags2);
    r.wait[0] = uni(g.wait[0]); r.wait[1] = uni(g.wait[1]); r.wait[2] = uni(g.wait[2]);
    r.src_sc1 = uni(g.src_sc1);
    r.G = uni(g.G); r.W = uni(g.W); r.h = uni(g.h); r.c = uni(g.c); r.x_in = uni(g.x_in); r.Wih = uni(g.Wih); r.bias = uni(g.bias);
    r.src_h = uni(g.src_h); r.Wg = uni(g.Wg); r.out = uni(g.out); r.out_bm = uni(g.out_bm); r.out_dim = uni(g.out_dim); r.out_p = uni(g.out_p);
    r.dh_ext = uni(g.dh_ext); r.dh_ext_half = uni(g.dh_ext_half); r.dh_ext_rows = uni(g.dh_ext_rows); r.dh_last = uni(g.dh_last);
    r.dA_sc1 = uni(g.dA_sc1); r.xchg = uni(g.xchg); r.xchg_ext = uni(g.xchg_ext); r.xchg_mel = uni(g.xchg_mel);
    r.fast_flags = uni(g.fast_flags); r.xtab = uni(g.xtab); r.hx = uni(g.hx);
    return r;
}

__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, uint4 v) {
    u32x4 d;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, kAuxSc1);
}
__device__ __forceinline__ void st16_sc1_so(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, uint4 v) {   // soff: wave-uniform part
    u32x4 d;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, kAuxSc1);
}
__device__ __forceinline__ uint4 ld16_sc1_so(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kAuxSc1);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 ld8_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, kAuxSc1);
    return make_uint2(v[0], v[1]);
}

// LDS-DMA, write-through read (sc1): 64 lanes x 16 bytes land at lds_dst_uniform + 16 * lane.  Issued from inline asm: the
// compiler does not see the LDS write (callers wait with s_waitcnt vmcnt and a barrier before reading the region) and does not
// count the operation (its own counted waits only get more conservative: vmcnt retires in order).
__device__ __forceinline__ void glds16_sc1(const void* gsrc_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 sc1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst_uniform)
        : "memory");
}
// the same through the XCD's own L2 (nt: bypasses only this CU's L1): for tiles that workgroups of the SAME XCD stored plainly
__device__ __forceinline__ void glds16_nt(const void* gsrc_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst_uniform)
        : "memory");
}
typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

// the flags chain-step (group g, step t) of role R waits for (FusedWait, kernels.h)
__device__ __forceinline__ const int* wait_addr(const FusedArgs& a, const FusedWait& w, int g, int t, int p, int& n) {
    n = 0;
    if (!w.flags) return nullptr;
    const int tt = (t >> w.t_shr) + w.t_add;
    if (tt < 0 || tt >= w.T) return nullptr;
    n = w.per_p ? 1 : w.n;
    return w.flags + ((size_t)g * w.T + tt) * a.flag_stride + (w.per_p ? p : 0);
}
struct Waits { FusedWait w0, w1, w2; };
__device__ __forceinline__ FlagPoll step_flags(const FusedArgs& a, const Waits& W, int g, int t, int p) {
    FlagPoll s{nullptr, 0, nullptr, 0, nullptr};
    int n2 = 0;
    s.fa = wait_addr(a, W.w0, g, t, p, s.na);
    s.fb = wait_addr(a, W.w1, g, t, p, s.nb);
    s.fc = wait_addr(a, W.w2, g, t, p, n2);
    return s;
}
__device__ __forceinline__ bool poll_empty(const FlagPoll& s) { return s.na == 0 && s.nb == 0 && !s.fc; }
// 16-row LSTM roles (lstm_fused16.h) keep one row of flags per 16-ROW group; a product role working on the 32-row group g waits for
// the rows of the 16-row groups 2 g and 2 g + 1 (the second only if the batch has it).  _w0: the role's wait 0 names an LSTM role
// (mel head, projections, backward mel head); _w2: its wait 2 does (dL/dh product role: one flag per group, its partner's).
__device__ __forceinline__ FlagPoll step_flags_lstm16_w0(const FusedArgs& a, const Waits& W, int g, int t, int p) {
    FlagPoll s{nullptr, 0, nullptr, 0, nullptr};
    int n2 = 0;
    s.fa = wait_addr(a, W.w0, 2 * g, t, p, s.na);
    if (16 * (2 * g + 1) < a.Bp) s.fb = wait_addr(a, W.w0, 2 * g + 1, t, p, s.nb);
    s.fc = wait_addr(a, W.w2, g, t, p, n2);
    return s;
}
__device__ __forceinline__ FlagPoll step_flags_lstm16_w2(const FusedArgs& a, const Waits& W, int g, int t, int p) {
    FlagPoll s{nullptr, 0, nullptr, 0, nullptr};
    int n2 = 0;
    s.fa = wait_addr(a, W.w0, g, t, p, s.na);
    if (16 * (2 * g + 1) < a.Bp) s.fb = wait_addr(a, W.w2, 2 * g + 1, t, p, s.nb);
    s.fc = wait_addr(a, W.w2, 2 * g, t, p, n2);
    return s;
}

}  // namespace
}  // namespace pl
