#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
template<int N> void probe() {
  hipFuncAttributes fa;
  hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<void const*>(&ocean_rowpass_kernel<N>));
  printf("N=%d getattr=%d regs=%d shared=%zu local=%zu maxthreads=%d maxdyn=%d LDSreq=%zu\n", N, (int)e, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock, fa.maxDynamicSharedSizeBytes, RowCfg<N>::LDS);
  for (int sz : {32768, 65536, 66000, 98304, 104960, 131072, 163840}) {
    e = hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, sz);
    printf("   set %d -> %d (%s)\n", sz, (int)e, hipGetErrorString(e));
  }
  (void)hipGetLastError();
}
int main(){ probe<2048>(); probe<4096>(); return 0; }
