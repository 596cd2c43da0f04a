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

// One lane of wave 0 waits until *cnt == target (relaxed agent-scope = sc1 loads), bounded; the verdict is
// shared through LDS so that the whole workgroup leaves together on a timeout.  Ends with a barrier.
__device__ __forceinline__ bool wait_arrivals(const int* cnt, int target, int* status, int* lds_flag, unsigned long long spin_ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                __builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) {
                __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// every storing wave has drained its hand-off stores (all but its N youngest memory operations, which are not
// part of the hand-off); one lane signals for the workgroup
template <int N>
__device__ __forceinline__ void publish(int* cnt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
