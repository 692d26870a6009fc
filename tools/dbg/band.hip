// Microbenchmark: the column pass's map stores and spectrum loads at 4096^2 (2-column tiles, 1024 threads, one workgroup per CU,
// 8 tiles per workgroup one after the other) with the row-major layouts the kernels use, against "banded" layouts in which the
// columns the chip works on at the same time (256 CUs x 2 columns = 512) are contiguous in memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 4096, W = 2, T = 512, E = 8, NT = N / W;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// map float4 index of texel (x, y, layer); G = 4 quad layout.  BAND = 0: rows of the whole image; BAND = B: [x / B][y][x % B]
template<int BAND> __device__ __forceinline__ size_t map_at(int y, int x, int layer) {
  if (BAND < 0) return (size_t)y * 2 * N + (size_t)(x >> 1) * 4 + layer * 2 + (x & 1);    // pairs: 32 B of layer 0 of two texels, 32 B of layer 1
  if (BAND == 0) return (size_t)y * 2 * N + (size_t)(x >> 2) * 8 + layer * 4 + (x & 3);
  int b = x / BAND, xi = x % BAND;
  return (size_t)b * 2 * N * BAND + (size_t)y * 2 * BAND + (size_t)(xi >> 2) * 8 + layer * 4 + (xi & 3);
}
// spectrum element index (16-byte values), 8 x 8 blocks; BAND as above
template<int BAND> __device__ __forceinline__ size_t spec_at(int y, int x) {
  if (BAND <= 0) return ((size_t)((y >> 3) * (N / 8) + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7);
  int b = x / BAND, xi = x % BAND;
  return (size_t)b * N * BAND + (((size_t)((y >> 3) * (BAND / 8) + (xi >> 3))) << 6) + ((y & 7) << 3) + (xi & 7);
}
// 2 x 2 texel patches: one 128-byte line = layer 0 of (y, x), (y, x+1), (y+1, x), (y+1, x+1), then layer 1 of the same four; bands of BAND columns
template<int BAND> __device__ __forceinline__ size_t patch_at(int y, int x, int layer) {
  int b = x / BAND, xi = x % BAND;
  return (size_t)b * 2 * N * BAND + ((size_t)(y >> 1) * (BAND / 2) + (xi >> 1)) * 8 + layer * 4 + (y & 1) * 2 + (xi & 1);
}
template<int BAND, bool LOADS, bool STORES, int PATCH = 0>
__global__ void __launch_bounds__(1024) colmem(float4 const* __restrict__ spec, float4* __restrict__ maps, float* sink, int groups) {
  int cp = threadIdx.x % W, t = threadIdx.x / W;
  float acc = 0;
  for (int item = blockIdx.x; item < NT; item += groups) {
    int q = item % NT; int tile = (q & 7) * (NT / 8) + (q >> 3);
    int x = tile * W + cp;
    float4 v[E];
    if (LOADS) {
      #pragma unroll
      for (int s = 0; s < E; ++s) v[s] = spec[spec_at<BAND>(t + T * s, x)];
    } else {
      #pragma unroll
      for (int s = 0; s < E; ++s) v[s] = make_float4(t, s, cp, 1.0f);
    }
    if (STORES) {
      #pragma unroll
      for (int s = 0; s < E; ++s) {
        if (PATCH == 1) {
          maps[patch_at<BAND>(t + T * s, x, 0)] = v[s];
          maps[patch_at<BAND>(t + T * s, x, 1)] = make_float4(v[s].y, v[s].x, v[s].w, 0.0f);
        } else if (PATCH == 2) {
          // four lanes = one patch; the even quad of lanes writes the displacement half, the odd quad the normal half: one instruction = whole lines
          maps[patch_at<BAND>(((t & ~3) | (t & 1)) + T * s, x, (t >> 1) & 1)] = v[s];
          maps[patch_at<BAND>(((t & ~3) | 2 | (t & 1)) + T * s, x, (t >> 1) & 1)] = make_float4(v[s].y, v[s].x, v[s].w, 0.0f);
        } else if (BAND == -3) {
          // one instruction = the 64-byte run of ONE row: lanes of the even row write its displacement pair, lanes of the odd row its normal pair
          maps[map_at<BAND>((t & ~1) + T * s, x, t & 1)] = v[s];
          maps[map_at<BAND>((t | 1) + T * s, x, t & 1)] = make_float4(v[s].y, v[s].x, v[s].w, 0.0f);
        } else {
          maps[map_at<BAND>(t + T * s, x, 0)] = v[s];
          maps[map_at<BAND>(t + T * s, x, 1)] = make_float4(v[s].y, v[s].x, v[s].w, 0.0f);
        }
      }
    } else {
      #pragma unroll
      for (int s = 0; s < E; ++s) acc += v[s].x + v[s].w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}
template<bool LOADS, bool STORES>
__global__ void __launch_bounds__(1024) dense(float4 const* __restrict__ spec, float4* __restrict__ maps, float* sink) {
  // the same bytes as one column-pass launch, as plain streams: 16 B per point in, 32 B per point out
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x, n = (size_t)N * N;
  float acc = 0;
  for (; i < n; i += st) {
    float4 v = LOADS ? spec[i] : make_float4(i, 1, 2, 3);
    if (STORES) { maps[i] = v; maps[n + i] = make_float4(v.y, v.x, v.w, 0.0f); } else acc += v.x + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}
int main() {
  size_t plane = (size_t)N * N;
  float4 *spec, *maps; float* sink;
  CK(hipMalloc(&spec, plane * 16)); CK(hipMalloc(&maps, 2 * plane * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(spec, 0, plane * 16)); CK(hipMemset(maps, 0, 2 * plane * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<3;++i) fn(); (void)hipEventRecord(e0); for (int i=0;i<20;++i) fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); ms/=20; printf("%-86s %8.1f us  %7.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); };
  double lb = 16.0 * plane, sb = 32.0 * plane;
  timeit("dense streams: loads only (268 MB)", lb, [&]{ hipLaunchKernelGGL((dense<true, false>), dim3(2048), dim3(1024), 0, 0, spec, maps, sink); });
  timeit("dense streams: stores only (537 MB)", sb, [&]{ hipLaunchKernelGGL((dense<false, true>), dim3(2048), dim3(1024), 0, 0, spec, maps, sink); });
  timeit("dense streams: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((dense<true, true>), dim3(2048), dim3(1024), 0, 0, spec, maps, sink); });
  for (int groups : {256, 2048}) {
    printf("-- %d workgroups of 1024 threads (%s)\n", groups, groups == 256 ? "persistent, 8 tiles each" : "one tile each");
    timeit("row-major image: loads only", lb, [&]{ hipLaunchKernelGGL((colmem<0, true, false>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("row-major image: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<0, false, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("row-major image: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<0, true, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("512-column bands: loads only", lb, [&]{ hipLaunchKernelGGL((colmem<512, true, false>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("512-column bands: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<512, false, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("512-column bands: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<512, true, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("64-column bands: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<64, true, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("64-column bands: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<64, false, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("2x2 patches, 64-column bands, half lines per instruction: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<64, false, true, 1>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("2x2 patches, 64-column bands, whole lines per instruction: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<64, false, true, 2>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("2x2 patches, 64-column bands, whole lines per instruction: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<64, true, true, 2>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("2x2 patches, 128-column bands, whole lines per instruction: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<128, true, true, 2>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("2x2 patches, 32-column bands, whole lines per instruction: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<32, true, true, 2>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("texel pairs (64 B per tile row, two 32-B instructions): stores only", sb, [&]{ hipLaunchKernelGGL((colmem<-2, false, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("texel pairs, one 64-B run per instruction: stores only", sb, [&]{ hipLaunchKernelGGL((colmem<-3, false, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
    timeit("texel pairs, one 64-B run per instruction: loads + stores", lb + sb, [&]{ hipLaunchKernelGGL((colmem<-3, true, true>), dim3(groups), dim3(1024), 0, 0, spec, maps, sink, groups); });
  }
  return 0;
}
