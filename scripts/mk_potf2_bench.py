"""Developer tool: wrap friedrich_amd/csrc/potf2.hip's kernel into a self-contained timing harness (scripts/potf2_bench.hip)."""
src = open('friedrich_amd/csrc/potf2.hip').read()
kern = src[src.index("constexpr int PB = 128;"):src.index("int launch_potf2(")]
kern = kern.replace("    lds_barrier();\n\n    for (int j = 0; j < n; ++j) {", "    lds_barrier();\n    const long long tc0 = __builtin_amdgcn_s_memtime();\n    for (int j = 0; j < n; ++j) {")
kern = kern.replace("#pragma unroll\n    for (int k = 0; k < PE; ++k) {\n        const int c = cg + 8 * k;\n        if (row_ok && c < n && i >= c) {", "    if (t == 0 && info) info[2] = __builtin_amdgcn_s_memtime() - tc0;\n#pragma unroll\n    for (int k = 0; k < PE; ++k) {\n        const int c = cg + 8 * k;\n        if (row_ok && c < n && i >= c) {")
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
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for(int rep=0;rep<4;++rep){
    (void)hipMemcpy(A,h.data(),n*n*8,hipMemcpyHostToDevice); (void)hipMemset(info,0,8*(3+n));
    (void)hipEventRecord(e0); hipLaunchKernelGGL(fr::potf2_kernel,dim3(1),dim3(512),0,0,A,(int64_t)n,n,(int64_t)0,0,0.0,inv,(int64_t)n,info); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); int64_t hi[3]; (void)hipMemcpy(hi,info,24,hipMemcpyDeviceToHost);
    printf("rep %d: %.1f us, loop ticks %lld (%.0f per step), fail=%lld\\n",rep,ms*1e3,(long long)hi[2],hi[2]/128.0,(long long)hi[0]);
  }
  std::vector<double> L(n*n); (void)hipMemcpy(L.data(),A,n*n*8,hipMemcpyDeviceToHost); printf("L00=%.6f L10=%.6f L[127,126]=%.6f\\n",L[0],L[1],L[127+126*n]);
  return 0; }
'''
open('scripts/potf2_bench.hip', 'w').write(prog)
