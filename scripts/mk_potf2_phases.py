"""Developer tool: per-phase s_memtime instrumentation of the diagonal-block kernel (writes scripts/potf2_bench_phases.hip).
Wave 0 stamps: start | block loaded | after each F_b | after each stage's last barrier | inverse stored."""
src = open('friedrich_amd/csrc/potf2.hip').read()
kern = src[src.index("constexpr int PB = 128;"):src.index("int launch_potf2(")]
# potf2_block: one more argument, stamps by wave 0 and by update wave 0
kern = kern.replace("int64_t* __restrict__ info, double* __restrict__ cest = nullptr)\n{",
                    "int64_t* __restrict__ info, double* __restrict__ cest = nullptr, long long* ts = nullptr)\n{\n    if (ts && threadIdx.x == 0) ts[0] = __builtin_amdgcn_s_memtime();", 1)
assert "long long* ts = nullptr" in kern
k0 = "    if (w == 0) {\n"
assert kern.count(k0) == 1
kern = kern.replace(k0, k0 + "        if (ts && lane == 0) ts[1] = __builtin_amdgcn_s_memtime();\n", 1)
three = "            lds_barrier();\n            lds_barrier();\n            lds_barrier();\n"
assert kern.count(three) == 1
kern = kern.replace(three, "            if (ts && lane == 0) ts[2 + 2 * b] = __builtin_amdgcn_s_memtime();\n" + three +
                    "            if (ts && lane == 0) ts[3 + 2 * b] = __builtin_amdgcn_s_memtime();\n")
# update-wave stamps: after each of the 3 barriers of a stage (8-space indent inside the update branch, before the inverse's store)
head, tail = kern.split("    const int u = w - 1;\n", 1)
body, rest = tail.split("    // ---- store the inverse", 1)
parts = body.split("        lds_barrier();\n")
assert len(parts) == 4, len(parts)
body = parts[0]
for idx in range(3):
    body += "        lds_barrier();\n        if (ts && w == 1 && lane == 0) ts[16 + 8 * b + %d] = __builtin_amdgcn_s_memtime();\n" % idx + parts[idx + 1]
kern = head + "    const int u = w - 1;\n" + body + "    // ---- store the inverse" + rest
# the kernel wrapper passes the stamp buffer on and stamps the end
kern = kern.replace("unsigned* __restrict__ xcc_word)\n{", "unsigned* __restrict__ xcc_word, long long* ts)\n{", 2)
assert "long long* ts)" in kern
import re
n_calls = 0
def _stamp(m):
    global n_calls
    n_calls += 1
    return ('    { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); if ((threadIdx.x & 63) == 0) ts[40 + (threadIdx.x >> 6)] = hwid; }\n'
            + m.group(0).replace("cest);", "cest, ts);") + "    if (threadIdx.x == 0) ts[10] = __builtin_amdgcn_s_memtime();\n")
kern = re.sub(r"    potf2_block<[^>]*>\(lds, A, lda, n, col0, mode, sub, inv, ldinv, info, cest\);\n", _stamp, kern)
assert n_calls == 2, n_calls
kern = kern.replace('potf2_flat_kernel', 'potf2_flat_kernel_h')  # (the library exports a kernel of the same name and signature)
prog = '''#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <unistd.h>
#include "friedrich_amd.h"
#include "fr_internal.hpp"
namespace fr {
''' + kern + '''}
#ifdef FR_K4_TS
__global__ void copy_k4ts(long long* out) { if (threadIdx.x < 64) out[threadIdx.x] = fr::flat::k4ts[threadIdx.x]; }
#endif
int main(int argc, char** argv){
  const bool noise = argc > 1 && argv[1][0] == 'n'; const bool capped = argc > 2 && argv[2][0] == 'c'; const bool flatk = argc > 2 && argv[2][0] == 'f';
  fr_ctx* ctx = nullptr; double *NA = nullptr, *NC = nullptr; const int64_t NM = 16384, NK = 512;
  if (noise) {
    if (fr_ctx_create(&ctx, 0) != FR_OK) { printf("ctx failed\\n"); return 1; }
    (void)hipMalloc(&NA, NM*NK*8); (void)hipMalloc(&NC, NM*NM*8); (void)hipMemset(NA, 0, NM*NK*8); (void)hipMemset(NC, 0, NM*NM*8);
  }
  hipStream_t hs; int lo_p, hi_p; (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p); (void)hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, hi_p);
  const int n=128; std::vector<double> h(n*n);
  for(int c=0;c<n;++c) for(int r=0;r<n;++r) h[r+c*n]= (r==c? n+1.0 : 1.0/(1.0+abs(r-c)));
  double *A,*inv; int64_t* info; long long* ts; (void)hipMalloc(&A,n*n*8); (void)hipMalloc(&inv,n*n*8); (void)hipMalloc(&info,8*(3+n)); (void)hipMalloc(&ts,8*64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fr::potf2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fr::POTF2_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fr::potf2_uncapped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fr::POTF2_LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fr::potf2_flat_kernel_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fr::flat::LDS_BYTES);
  if (flatk) {
    std::vector<double> L0(n*n), W0(n*n), L1(n*n), W1(n*n);
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemcpy(A,h.data(),n*n*8,hipMemcpyHostToDevice); (void)hipMemset(info,0,8*(3+n)); (void)hipMemset(inv,0,n*n*8);
        if (noise) { for (int g = 0; g < 6; ++g) fr_gemm(ctx, 0, 1, NM, NM, NK, -1.0, NA, NM, NA, NM, 1.0, NC, NM); usleep(6000); }
        hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0,hs);
        if (which == 0) hipLaunchKernelGGL(fr::potf2_uncapped_kernel,dim3(1),dim3(512),fr::POTF2_LDS,hs,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info,(double*)nullptr,(unsigned*)nullptr,ts);
        else hipLaunchKernelGGL(fr::potf2_flat_kernel_h,dim3(1),dim3(512),fr::flat::LDS_BYTES,hs,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info,(double*)nullptr,(unsigned*)nullptr);
        (void)hipEventRecord(e1,hs); hipError_t er = hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms,e0,e1);
        printf("%s rep %d: %.1f us (%s)\\n", which ? "flat" : "staged", rep, ms*1e3, hipGetErrorString(er));
      }
      (void)hipMemcpy(which ? L1.data() : L0.data(), A, n*n*8, hipMemcpyDeviceToHost); (void)hipMemcpy(which ? W1.data() : W0.data(), inv, n*n*8, hipMemcpyDeviceToHost);
    }
#ifdef FR_K4_TS
    { long long kt[128]; (void)hipDeviceSynchronize(); hipLaunchKernelGGL(copy_k4ts, dim3(1), dim3(128), 0, 0, ts); (void)hipDeviceSynchronize(); (void)hipMemcpy(kt, ts, sizeof(kt) < 8*64 ? sizeof(kt) : 8*64, hipMemcpyDeviceToHost); const char* nm[8] = {"PA","PB","XA","XB","U0","U1","U2","U3"};
      for (int w = 0; w < 8; ++w) printf("  %s (SIMD %lld): role %lld cycles, waiting %lld in %lld blocked waits; pivot panels: all waits %lld load %lld steps %lld whole pivot phase %lld\\n", nm[w], kt[8*w+3], kt[8*w], kt[8*w+1], kt[8*w+2], kt[8*w+6], kt[8*w+4], kt[8*w+5], kt[8*w+7]); }
#endif
    double el = 0, ew = 0, ml = 0, mw = 0; for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) { double a = L0[r+c*n], b = L1[r+c*n]; if (r < c) { a = 0; } el = fmax(el, fabs(a-b)); ml = fmax(ml, fabs(a)); ew = fmax(ew, fabs(W0[r+c*n]-W1[r+c*n])); mw = fmax(mw, fabs(W0[r+c*n])); if (!(fabs(a-b) < 1e-9) && r >= c && c < 3 && r < 6) printf("  L[%d,%d] staged %.12g flat %.12g\\n", r, c, a, b); }
    printf("flat vs staged: max |dL| %.3e (max |L| %.3e)  max |dW| %.3e (max |W| %.3e)\\n", el, ml, ew, mw);
    return 0;
  }
  for(int rep=0;rep<3;++rep){
    (void)hipMemcpy(A,h.data(),n*n*8,hipMemcpyHostToDevice); (void)hipMemset(info,0,8*(3+n));
    if (noise) { for (int g = 0; g < 6; ++g) fr_gemm(ctx, 0, 1, NM, NM, NK, -1.0, NA, NM, NA, NM, 1.0, NC, NM); usleep(6000); }
    hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0,hs);
    hipLaunchKernelGGL(capped ? fr::potf2_kernel : fr::potf2_uncapped_kernel,dim3(1),dim3(512),fr::POTF2_LDS,hs,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info,(double*)nullptr,(unsigned*)nullptr,ts); (void)hipEventRecord(e1,hs); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms,e0,e1);
    long long t[64]; (void)hipMemcpy(t,ts,8*64,hipMemcpyDeviceToHost);
    if (rep==2) { printf("event %.1f us; shader-clock cycles: load %lld", ms*1e3, t[1]-t[0]);
      long long prev=t[1]; for(int b=0;b<4;++b){ printf(" | F%d %lld upd %lld", b, t[2+2*b]-prev, t[3+2*b]-t[2+2*b]); prev=t[3+2*b]; }
      printf(" | store %lld | total %lld\\n", t[10]-prev, t[10]-t[0]);
      printf("  wave -> SIMD:"); for (int w = 0; w < 8; ++w) printf(" %d:%lld", w, (t[40+w] >> 4) & 3); printf("  (cu %lld)\\n", (t[40] >> 8) & 15);
      for(int b=0;b<4;++b){ printf("  stage %d (update wave 0): P2 products %lld | P3 %lld\\n", b, t[16+8*b+1]-t[16+8*b], t[16+8*b+2]-t[16+8*b+1]); } }
  }
  return 0; }
'''
open('scripts/potf2_bench_phases.hip', 'w').write(prog)
