// Microbenchmark: map-store patterns of a 4-column tile when both layers of a texel are adjacent (32 B per texel) --
// does it matter whether one store instruction writes 64 contiguous bytes, or 16-byte pieces at a 32-byte stride that the
// next instruction of the same wave completes?  (1024^2 x 4 cascades, 134 MB per launch, as tools/dbg/width.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 1024, C = 4;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// MODE 0: quad layout (64 B of layer 0 of four texels, then 64 B of layer 1): each instruction writes a 64-byte run
// MODE 1: texel layout, natural lanes: instruction A writes bytes [32 cp, 32 cp + 16), instruction B [32 cp + 16, 32 cp + 32)
// MODE 2: texel layout, lanes transposed: instruction A writes bytes [16 cp, 16 cp + 16) (texels 0, 1), B the next 64 (address pattern == MODE 0)
template<int MODE, int AUX, int RUN>
__global__ void __launch_bounds__(512) mapstore(float4* __restrict__ maps, float v) {
  constexpr int TT = 512 / RUN, E2 = N / TT;
  int c = blockIdx.y, tile = blockIdx.x; int cp = threadIdx.x % RUN, t = threadIdx.x / RUN;
  size_t plane = (size_t)N * N; float4* l0 = maps + (size_t)c * 2 * plane;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(l0, 0, (int)(2 * plane * 16), 0x00020000);
  u4 d = { __float_as_uint(v), __float_as_uint(v + 1), __float_as_uint(v + 2), 0u };
  #pragma unroll
  for (int s = 0; s < E2; ++s) {
    int y = t + TT * s;
    int line = (y * 2 * N + tile * RUN * 2) * 16;    // byte offset of the tile's RUN texels (both layers) in row y
    int oa, ob;
    if (MODE == 0) { oa = line + cp * 16; ob = line + RUN * 16 + cp * 16; }
    else if (MODE == 1) { oa = line + cp * 32; ob = oa + 16; }
    else { oa = line + cp * 16; ob = oa + RUN * 16; }
    __builtin_amdgcn_raw_buffer_store_b128(d, r, oa, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, ob, 0, AUX);
  }
}
// 24-byte texels (dx, dy, dz, nx, ny, nz: the always-zero w components are not stored): a 4-column tile's row is 96 contiguous
// bytes.  MODE 0: per lane 16 B (displacement + nx) then 8 B (ny, nz).  MODE 1: per lane 12 B then 12 B.
typedef unsigned int u3 __attribute__((ext_vector_type(3)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template<int MODE, int AUX, int RUN>
__global__ void __launch_bounds__(512) mapstore24(float* __restrict__ maps, float v) {
  constexpr int TT = 512 / RUN, E2 = N / TT;
  int c = blockIdx.y, tile = blockIdx.x; int cp = threadIdx.x % RUN, t = threadIdx.x / RUN;
  size_t plane = (size_t)N * N; float* l0 = maps + (size_t)c * 6 * plane;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(l0, 0, (int)(6 * plane * 4), 0x00020000);
  u4 d4 = { __float_as_uint(v), __float_as_uint(v + 1), __float_as_uint(v + 2), __float_as_uint(v + 3) };
  u3 d3 = { d4.x, d4.y, d4.z }; u2 d2 = { d4.x, d4.y };
  #pragma unroll
  for (int s = 0; s < E2; ++s) {
    int y = t + TT * s;
    int o = (y * N + tile * RUN + cp) * 24;
    if (MODE == 0) { __builtin_amdgcn_raw_buffer_store_b128(d4, r, o, 0, AUX); __builtin_amdgcn_raw_buffer_store_b64(d2, r, o + 16, 0, AUX); }
    else { __builtin_amdgcn_raw_buffer_store_b96(d3, r, o, 0, AUX); __builtin_amdgcn_raw_buffer_store_b96(d3, r, o + 12, 0, AUX); }
  }
}
int main() {
  size_t plane = (size_t)N*N;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<5;++i) fn(); hipEventRecord(e0); for (int i=0;i<50;++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=50; printf("%-72s %8.1f us  %7.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); };
  float4* maps; CK(hipMalloc(&maps, C*2*plane*16)); double mb = 32.0*C*plane;
  timeit("4-col tiles, quad layout: 64-B run per instruction, plain", mb, [&]{ hipLaunchKernelGGL((mapstore<0, 0, 4>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("4-col tiles, quad layout: 64-B run per instruction, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<0, 17, 4>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("4-col tiles, texel layout: 16-B pieces at 32-B stride, plain", mb, [&]{ hipLaunchKernelGGL((mapstore<1, 0, 4>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("4-col tiles, texel layout: 16-B pieces at 32-B stride, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<1, 17, 4>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("2-col tiles, quad-of-2 layout: 32-B run per instruction, plain", mb, [&]{ hipLaunchKernelGGL((mapstore<0, 0, 2>), dim3(N/2, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("2-col tiles, quad-of-2 layout: 32-B run per instruction, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<0, 17, 2>), dim3(N/2, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("2-col tiles, texel layout: 16-B pieces at 32-B stride, plain", mb, [&]{ hipLaunchKernelGGL((mapstore<1, 0, 2>), dim3(N/2, C), dim3(512), 0, 0, maps, 1.0f); });
  timeit("8-col tiles, quad layout: 128-B run per instruction, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<0, 17, 8>), dim3(N/8, C), dim3(512), 0, 0, maps, 1.0f); });
  { float* m24; CK(hipMalloc(&m24, C*6*plane*4)); double mb24 = 24.0*C*plane;
    timeit("24-B texels, 4-col tiles: 16 B + 8 B per lane, plain", mb24, [&]{ hipLaunchKernelGGL((mapstore24<0, 0, 4>), dim3(N/4, C), dim3(512), 0, 0, m24, 1.0f); });
    timeit("24-B texels, 4-col tiles: 16 B + 8 B per lane, sc0 sc1", mb24, [&]{ hipLaunchKernelGGL((mapstore24<0, 17, 4>), dim3(N/4, C), dim3(512), 0, 0, m24, 1.0f); });
    timeit("24-B texels, 4-col tiles: 12 B + 12 B per lane, plain", mb24, [&]{ hipLaunchKernelGGL((mapstore24<1, 0, 4>), dim3(N/4, C), dim3(512), 0, 0, m24, 1.0f); });
    timeit("24-B texels, 8-col tiles: 16 B + 8 B per lane, plain", mb24, [&]{ hipLaunchKernelGGL((mapstore24<0, 0, 8>), dim3(N/8, C), dim3(512), 0, 0, m24, 1.0f); });
    timeit("24-B texels, 2-col tiles: 16 B + 8 B per lane, plain", mb24, [&]{ hipLaunchKernelGGL((mapstore24<0, 0, 2>), dim3(N/2, C), dim3(512), 0, 0, m24, 1.0f); }); }
  return 0;
}
