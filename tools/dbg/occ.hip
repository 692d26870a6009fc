#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
template<int N> void probe() {
  int nb = -1;
  hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N>::LDS);
  hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_colpass_kernel<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ColCfg<N>::LDS);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, false>), RowCfg<N>::THREADS, RowCfg<N>::LDS);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, false>));
  printf("N=%d rowpass: threads %d lds %zu regs %d -> blocks/CU %d\n", N, RowCfg<N>::THREADS, RowCfg<N>::LDS, fa.numRegs, nb);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<void const*>(&ocean_colpass_kernel<N, false>), ColCfg<N>::THREADS, ColCfg<N>::LDS);
  hipFuncGetAttributes(&fa, reinterpret_cast<void const*>(&ocean_colpass_kernel<N, false>));
  printf("N=%d colpass: threads %d lds %zu regs %d -> blocks/CU %d\n", N, ColCfg<N>::THREADS, ColCfg<N>::LDS, fa.numRegs, nb);
}
int main(){ probe<512>(); probe<1024>(); probe<2048>(); return 0; }
