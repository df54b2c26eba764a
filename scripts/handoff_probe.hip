// What does a hand-off between two workgroups cost, by placement (same XCD / different XCDs) and by the cache-policy
// bits of the store and of the polling load?  Two one-wave workgroups bounce a counter ROUNDS times (A writes 2i+1 to
// word 0, B answers 2i+2 in word 1 -- separate 128-byte lines); the round trip / 2 is one hand-off.  A poll that never
// sees the value (a stale L1 line) is cut off after SPIN_MAX polls and reported as "stale".
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32;
constexpr int ROUNDS = 2000;
constexpr int SPIN_MAX = 200000;

enum { LD_SC1, LD_SC0, LD_SC01, LD_PLAIN, LD_NT, LD_ATOMIC_OR_WAVE, LD_ATOMIC_OR_SC0, LD_ATOMIC_OR_SC1, LD_N };
enum { ST_SC1, ST_SC0, ST_SC01, ST_PLAIN, ST_NT, ST_ATOMIC_WAVE, ST_ATOMIC_SC1, ST_N };
static const char* LD_NAME[] = {"load sc1", "load sc0", "load sc0 sc1", "load plain", "load nt", "atomic_or rtn", "atomic_or rtn sc0", "atomic_or rtn sc1"};
static const char* ST_NAME[] = {"store sc1", "store sc0", "store sc0 sc1", "store plain", "store nt", "atomic_swap", "atomic_swap sc1"};

template <int LD>
__device__ __forceinline__ u32 poll(u32* p)
{
    u32 v;
    if constexpr (LD == LD_SC1) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == LD_SC0) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == LD_SC01) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == LD_PLAIN) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == LD_NT) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == LD_ATOMIC_OR_WAVE) {
        u32 z = 0;
        asm volatile("global_atomic_or %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");  // (sc0 on an atomic = return the old value)
    }
    if constexpr (LD == LD_ATOMIC_OR_SC0) {
        u32 z = 0;
        asm volatile("global_atomic_or %0, %1, %2, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    }
    if constexpr (LD == LD_ATOMIC_OR_SC1) {
        u32 z = 0;
        asm volatile("global_atomic_or %0, %1, %2, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    }
    return v;
}

template <int ST>
__device__ __forceinline__ void put(u32* p, u32 v)
{
    if constexpr (ST == ST_SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_SC0) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_SC01) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_PLAIN) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_NT) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_ATOMIC_WAVE) asm volatile("global_atomic_swap %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == ST_ATOMIC_SC1) asm volatile("global_atomic_swap %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// blocks `a` and `b` of the grid play; everybody else leaves.  out[0] = cycles (s_memrealtime, 100 MHz), out[1] = stale flag,
// out[2], out[3] = the XCDs of the two players
template <int LD, int ST>
__global__ void bounce(u32* words, long long* out, int a, int b)
{
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    if (threadIdx.x != 0) return;
    const bool first = (int)blockIdx.x == a;
    u32 id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[first ? 2 : 3] = id & 0xf;
    u32* mine = words + (first ? 0 : 32);
    u32* theirs = words + (first ? 32 : 0);
    long long t0 = __builtin_readcyclecounter();
    t0 = wall_clock64();
    bool stale = false;
    for (int i = 0; i < ROUNDS && !stale; ++i) {
        const u32 want = first ? 2u * i + 2u : 2u * i + 1u;
        if (first) put<ST>(mine, 2u * i + 1u);
        int spins = 0;
        while (poll<LD>(theirs) != want) {
            if (++spins > SPIN_MAX) { stale = true; break; }
        }
        if (!first) put<ST>(mine, 2u * i + 2u);
    }
    const long long t1 = wall_clock64();
    if (first) out[0] = t1 - t0;
    if (stale) out[1] = 1;
}

template <int LD, int ST>
static void run(u32* words, long long* out, int a, int b, const char* where)
{
    (void)hipMemset(words, 0, 256 * sizeof(u32));
    (void)hipMemset(out, 0, 4 * sizeof(long long));
    hipLaunchKernelGGL((bounce<LD, ST>), dim3(64), dim3(64), 0, 0, words, out, a, b);
    (void)hipDeviceSynchronize();
    long long h[4];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    if (h[1])
        printf("  %-18s + %-16s %-9s (xcd %lld -> %lld): STALE (poll never saw the value)\n", LD_NAME[LD], ST_NAME[ST], where, h[2], h[3]);
    else
        printf("  %-18s + %-16s %-9s (xcd %lld -> %lld): %7.0f ns per hand-off\n", LD_NAME[LD], ST_NAME[ST], where, h[2], h[3],
               (double)h[0] * 10.0 / (2.0 * ROUNDS));
}

template <int LD, int ST>
static void both(u32* words, long long* out)
{
    run<LD, ST>(words, out, 0, 8, "same XCD");
    run<LD, ST>(words, out, 0, 1, "other XCD");
}

template <int LD>
static void stores(u32* words, long long* out)
{
    both<LD, ST_SC1>(words, out);
    both<LD, ST_SC0>(words, out);
    both<LD, ST_SC01>(words, out);
    both<LD, ST_PLAIN>(words, out);
    both<LD, ST_NT>(words, out);
    both<LD, ST_ATOMIC_WAVE>(words, out);
    both<LD, ST_ATOMIC_SC1>(words, out);
}

// ---- second experiment: the same bounce while (a) every other workgroup of a 512-block launch streams from HBM, and/or (b) the
// polling wave itself has 16 streaming 16-byte loads in flight in front of every poll (vector loads return in order)
template <int ST, bool OWN_STREAM>
__global__ __launch_bounds__(256) void bounce_loaded(u32* words, long long* out, int a, int b, const double2* big, size_t big_elems, int bg_iters,
                                                     double* sink)
{
    const bool player = (int)blockIdx.x == a || (int)blockIdx.x == b;
    if (!player) {
        // background: stream bg_iters x 64 KiB per workgroup
        double acc = 0.0;
        size_t pos = ((size_t)blockIdx.x * 977u * 4096u + threadIdx.x) % big_elems;
        for (int i = 0; i < bg_iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double2 v = big[(pos + (size_t)k * 256u) % big_elems];
                acc += v.x + v.y;
            }
            pos = (pos + 4096u * 613u) % big_elems;
        }
        if (acc == 1.2345) sink[blockIdx.x] = acc;
        return;
    }
    if (threadIdx.x >= 64) return;
    const bool first = (int)blockIdx.x == a;
    u32 id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[first ? 2 : 3] = id & 0xf;
    u32* mine = words + (first ? 0 : 32);
    u32* theirs = words + (first ? 32 : 0);
    const long long t0 = wall_clock64();
    bool stale = false;
    double acc = 0.0;
    size_t pos = ((size_t)blockIdx.x * 7919u * 4096u + threadIdx.x) % big_elems;
    for (int i = 0; i < ROUNDS && !stale; ++i) {
        const u32 want = first ? 2u * i + 2u : 2u * i + 1u;
        if (first && threadIdx.x == 0) put<ST>(mine, 2u * i + 1u);
        double2 v[16];
        if (OWN_STREAM) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = big[(pos + (size_t)k * 64u) % big_elems];
            pos = (pos + 4096u * 331u) % big_elems;
        }
        int spins = 0;
        for (;;) {
            u32 got = poll<LD_SC1>(theirs);
            got = __builtin_amdgcn_readfirstlane(got);
            if (got == want) break;
            if (++spins > SPIN_MAX) { stale = true; break; }
        }
        if (OWN_STREAM) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y;
        }
        if (!first && threadIdx.x == 0) put<ST>(mine, 2u * i + 2u);
    }
    const long long t1 = wall_clock64();
    if (acc == 1.2345) sink[blockIdx.x] = acc;
    if (first && threadIdx.x == 0) out[0] = t1 - t0;
    if (stale) out[1] = 1;
}

template <int ST, bool OWN_STREAM>
static void run_loaded(u32* words, long long* out, int a, int b, const char* where, const double2* big, size_t big_elems, int bg_iters, double* sink)
{
    (void)hipMemset(words, 0, 256 * sizeof(u32));
    (void)hipMemset(out, 0, 4 * sizeof(long long));
    hipLaunchKernelGGL((bounce_loaded<ST, OWN_STREAM>), dim3(512), dim3(256), 0, 0, words, out, a, b, big, big_elems, bg_iters, sink);
    (void)hipDeviceSynchronize();
    long long h[4];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("  load sc1 + %-14s %-9s (xcd %lld -> %lld) background %-4s own loads in front %-3s: ", ST_NAME[ST], where, h[2], h[3], bg_iters ? "HBM" : "idle",
           OWN_STREAM ? "yes" : "no");
    if (h[1]) printf("STALE\n");
    else printf("%7.0f ns per hand-off\n", (double)h[0] * 10.0 / (2.0 * ROUNDS));
}

int main()
{
    {
        u32* words;
        long long* out;
        double2* big;
        double* sink;
        const size_t big_elems = (size_t)1 << 28;  // 4 GiB of double2
        (void)hipMalloc(&words, 256 * sizeof(u32));
        (void)hipMalloc(&out, 4 * sizeof(long long));
        (void)hipMalloc(&big, big_elems * sizeof(double2));
        (void)hipMalloc(&sink, 4096 * sizeof(double));
        (void)hipMemset(big, 0, big_elems * sizeof(double2));
        for (int bg = 0; bg < 2; ++bg) {
            const int it = bg ? 6000 : 0;
            run_loaded<ST_SC1, false>(words, out, 0, 8, "same XCD", big, big_elems, it, sink);
            run_loaded<ST_SC1, false>(words, out, 0, 1, "other XCD", big, big_elems, it, sink);
            run_loaded<ST_PLAIN, false>(words, out, 0, 8, "same XCD", big, big_elems, it, sink);
            run_loaded<ST_SC1, true>(words, out, 0, 8, "same XCD", big, big_elems, it, sink);
            run_loaded<ST_SC1, true>(words, out, 0, 1, "other XCD", big, big_elems, it, sink);
            run_loaded<ST_PLAIN, true>(words, out, 0, 8, "same XCD", big, big_elems, it, sink);
        }
        (void)hipFree(big);
        if (getenv("HANDOFF_LOADED_ONLY")) return 0;
    }
    u32* words;
    long long* out;
    (void)hipMalloc(&words, 256 * sizeof(u32));
    (void)hipMalloc(&out, 4 * sizeof(long long));
    stores<LD_SC1>(words, out);
    stores<LD_SC0>(words, out);
    stores<LD_SC01>(words, out);
    stores<LD_PLAIN>(words, out);
    stores<LD_NT>(words, out);
    stores<LD_ATOMIC_OR_WAVE>(words, out);
    stores<LD_ATOMIC_OR_SC0>(words, out);
    stores<LD_ATOMIC_OR_SC1>(words, out);
    return 0;
}
