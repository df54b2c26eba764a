// handoff.hpp -- inter-workgroup hand-offs inside one launch / between concurrently running kernels on gfx950
// (8 XCDs with private L2s, per-CU L1s that other CUs' stores never refresh).  The forms are those of
// cdna_hip_programming.md, Guideline 16:
//   producer: plain stores -> every wave drains them -> workgroup barrier -> ONE lane: agent-scope release fence
//             (buffer_wbl2) -> drain -> relaxed agent-scope store of the flag;
//   consumer: ONE lane polls the flag with relaxed agent-scope loads (+ s_sleep), then ONE agent-scope acquire fence
//             (buffer_inv: this CU's L1), workgroup barrier, plain loads.
// Every spin is bounded by a wall-clock timeout that raises the context's host-visible status word, so a scheduling
// accident turns into FR_HIP_ERROR instead of a hung GPU.
#pragma once

#include <hip/hip_runtime.h>

namespace fr {

typedef __attribute__((address_space(1))) unsigned hgu32;
typedef __attribute__((address_space(1))) int hgi32;

constexpr unsigned long long HANDOFF_TIMEOUT_TICKS = 500000000ull;  // 5 s of the 100 MHz wall clock

// All threads of the workgroup call this; returns false (uniformly) after a timeout.  ACQUIRE = false: the payload was
// stored write-through (sc1) and will be read with sc1 loads, which need no fence (Guideline 16, R1).
template <bool ACQUIRE = true>
__device__ __forceinline__ bool handoff_wait_ge(const int* flag, int target, unsigned* status)
{
    int ok = 1;
    if (threadIdx.x == 0) {
        int v = __hip_atomic_load((hgi32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < target) {
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load((hgi32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= target) break;
                if ((++spins & 127u) == 0) {
                    const bool dead = __hip_atomic_load((hgu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    if (dead || wall_clock64() - t0 > HANDOFF_TIMEOUT_TICKS) {
                        __hip_atomic_store((hgu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        ok = 0;
                        break;
                    }
                }
            }
        }
        if (ACQUIRE && ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return __builtin_amdgcn_readfirstlane(__syncthreads_and(ok)) != 0;  // uniform by construction: say so (scalar branch)
}

// `count` (<= 64) flags, one per lane of the first wave, must all reach `target`
__device__ __forceinline__ bool handoff_wait_all_ge(const int* flags, int count, int target, unsigned* status)
{
    int ok = 1;
    if ((int)threadIdx.x < count) {
        const int* flag = flags + threadIdx.x;
        int v = __hip_atomic_load((hgi32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < target) {
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load((hgi32*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= target) break;
                if ((++spins & 127u) == 0) {
                    const bool dead = __hip_atomic_load((hgu32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    if (dead || wall_clock64() - t0 > HANDOFF_TIMEOUT_TICKS) {
                        __hip_atomic_store((hgu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        ok = 0;
                        break;
                    }
                }
            }
        }
    }
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the polling lanes have reconverged)
    return __builtin_amdgcn_readfirstlane(__syncthreads_and(ok)) != 0;
}

// All threads call this after their last store of the payload.
__device__ __forceinline__ void handoff_publish(int* flag, int value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the wait behind buffer_wbl2 (Guideline 16, pitfall 12)
        __hip_atomic_store((hgi32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace fr
