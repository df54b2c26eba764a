"""Developer tool: per-phase s_memtime instrumentation of the potf2 kernel (writes scripts/potf2_bench_phases.hip)."""
src = open('friedrich_amd/csrc/potf2.hip').read()
kern = src[src.index("constexpr int PB = 128;"):src.index("int launch_potf2(")]
kern = kern.replace("    for (int j = 0; j < n; ++j) {\n        const int jcg = j & 7, jk = j >> 3;", "    long long s1=0,s2=0,s3=0,s4=0;\n    for (int j = 0; j < n; ++j) {\n        const long long q0 = __builtin_amdgcn_s_memtime();\n        const int jcg = j & 7, jk = j >> 3;")
i1 = kern.index("        lds_barrier();\n        // ---- phase 2")
kern = kern[:i1] + "        const long long q1 = __builtin_amdgcn_s_memtime();\n        lds_barrier();\n        const long long q2 = __builtin_amdgcn_s_memtime();\n" + kern[i1 + len("        lds_barrier();\n"):]
i2 = kern.index("        lds_barrier();\n    }\n")
kern = kern[:i2] + "        const long long q3 = __builtin_amdgcn_s_memtime();\n        lds_barrier();\n        const long long q4 = __builtin_amdgcn_s_memtime();\n        s1+=q1-q0; s2+=q2-q1; s3+=q3-q2; s4+=q4-q3;\n    }\n" + kern[i2 + len("        lds_barrier();\n    }\n"):]
kern = kern.replace("#pragma unroll\n    for (int k = 0; k < PE; ++k) {\n        const int c = cg + 8 * k;\n        if (row_ok && c < n && i >= c) {", "    if ((t & 63) == 0 && info) { int w = t >> 6; info[8+4*w]=s1; info[9+4*w]=s2; info[10+4*w]=s3; info[11+4*w]=s4; }\n#pragma unroll\n    for (int k = 0; k < PE; ++k) {\n        const int c = cg + 8 * k;\n        if (row_ok && c < n && i >= c) {", 1)
prog = '''#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
namespace fr {
''' + kern + '''}
int main(){
  const int n=128; std::vector<double> h(n*n);
  for(int c=0;c<n;++c) for(int r=0;r<n;++r) h[r+c*n]= (r==c? n+1.0 : 1.0/(1.0+abs(r-c)));
  double *A,*inv; int64_t* info; (void)hipMalloc(&A,n*n*8); (void)hipMalloc(&inv,n*n*8); (void)hipMalloc(&info,8*(3+n));
  for(int rep=0;rep<2;++rep){
    (void)hipMemcpy(A,h.data(),n*n*8,hipMemcpyHostToDevice); (void)hipMemset(info,0,8*(3+n));
    hipLaunchKernelGGL(fr::potf2_kernel,dim3(1),dim3(512),0,0,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info); (void)hipDeviceSynchronize();
    int64_t hi[80]; (void)hipMemcpy(hi,info,8*80,hipMemcpyDeviceToHost);
    if (rep) for (int w=0; w<16; w+=3) printf("wave %2d: per-step ticks P1 %5.0f  bar1 %5.0f  P2 %5.0f  bar2 %5.0f\\n",w,hi[8+4*w]/128.0,hi[9+4*w]/128.0,hi[10+4*w]/128.0,hi[11+4*w]/128.0);
  }
  return 0; }
'''
open('scripts/potf2_bench_phases.hip', 'w').write(prog)
