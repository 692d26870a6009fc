// Microbenchmark (round 5): how fast does the dispatcher hand out workgroups?  2048 workgroups of 256 threads that note their start (s_memrealtime), sleep for
// `life` us and end; with and without the row pass's resources (18.9 KB of LDS, 80 VGPRs: six workgroups per CU).  Start-time percentiles over the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
__device__ __forceinline__ unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template<int REGS>
__global__ void __launch_bounds__(1024) probe(unsigned long long *starts, int life_ticks) {
  extern __shared__ unsigned char lds[];
  unsigned long long const t0 = realtime();
  if (REGS > 0) { asm volatile("v_mov_b32 v79, 0" ::: "v79"); }
  if (REGS > 100) { asm volatile("v_mov_b32 v123, 0" ::: "v123"); }
  if (threadIdx.x == 0) starts[blockIdx.x] = t0;
  while (realtime() < t0 + life_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 1023 && blockIdx.x == 1u << 30) lds[threadIdx.x] = 1;
}
int main() {
  int const W = 8192; unsigned long long *starts; CK(hipMalloc(&starts, W * 8));
  std::vector<unsigned long long> h(W);
  auto run = [&](char const *name, auto kernel, int groups, int threads, size_t lds, float life_us) {
    hipFuncSetAttribute(reinterpret_cast<void const*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kernel, dim3(groups), dim3(threads), lds, 0, starts, (int)(life_us * 100)); }
    hipDeviceSynchronize(); hipMemcpy(h.data(), starts, groups * 8, hipMemcpyDeviceToHost);
    std::vector<double> s(groups); unsigned long long m = *std::min_element(h.begin(), h.begin() + groups); for (int i = 0; i < groups; ++i) s[i] = (h[i] - m) * 0.01; std::sort(s.begin(), s.end());
    printf("%-86s starts: p10 %5.2f p25 %5.2f p50 %5.2f p75 %5.2f p90 %5.2f max %5.2f us\n", name, s[groups/10], s[groups/4], s[groups/2], s[groups*3/4], s[groups*9/10], s.back());
  };
  run("2048 x 256 threads, no LDS, few registers, 8 us of life (8 per CU fit)", probe<0>, 2048, 256, 0, 8);
  run("2048 x 256 threads, 18.9 KB LDS, few registers", probe<0>, 2048, 256, 18944, 8);
  run("2048 x 256 threads, no LDS, 80 registers (6 per CU)", probe<80>, 2048, 256, 0, 8);
  run("2048 x 256 threads, 18.9 KB LDS, 80 registers (the row pass's shape)", probe<80>, 2048, 256, 18944, 8);
  run("1024 x 256 threads, 37 KB LDS, 124 registers (the column pass's shape)", probe<124>, 1024, 256, 37376, 12);
  run("2048 x 256 threads, no LDS, few registers, 1 us of life", probe<0>, 2048, 256, 0, 1);
  run("4096 x 128 threads, no LDS, few registers, 8 us of life", probe<0>, 4096, 128, 0, 8);
  run("1024 x 512 threads, no LDS, few registers, 8 us of life", probe<0>, 1024, 512, 0, 8);
  run("8192 x 64 threads, no LDS, few registers, 8 us of life", probe<0>, 8192, 64, 0, 8);
  return 0;
}
