// Diagnostic (never shipped): the vertex stream of ocean.gen alone -- 1024 x 1024 vertices x 48 bytes = 50.3 MB per launch, written as the kernel
// writes it (2048 workgroups of 4 waves, every wave 6 store instructions of 16 bytes per lane into four 1.5 KB row segments), with no arithmetic,
// no fetches and no LDS; back-to-back launches on one stream, the same buffer every time (as bench.py and a renderer do).  Also: half the mesh,
// and the same bytes as one dense stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)

__global__ void __launch_bounds__(256) tile_stores(float4 *out, int sizex, int sizey, int tilesx, float seed)
{
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int const tilex = blockIdx.x % tilesx, tiley = blockIdx.x / tilesx;
  int const x0 = tilex * 32, y0 = tiley * 16 + 4 * wave;
  float4 *base = out + ((size_t)y0 * sizex + x0) * 3;
  float4 const v = make_float4(seed + lane, seed, wave, -1.0f);
  #pragma unroll
  for(int k = 0; k < 6; ++k)
  {
    int const j = 64 * k + lane, r = j / 96, c = j % 96;
    base[(unsigned)(r * sizex * 3 + c)] = v;
  }
}

__global__ void __launch_bounds__(256) dense_stores(float4 *out, size_t n, float seed)
{
  float4 const v = make_float4(seed, seed, seed, -1.0f);
  for(size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = v;
}

int main()
{
  int const sx = 1024, sy = 1024;
  float4 *buf; CK(hipMalloc(&buf, (size_t)sx * sy * 48));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](char const *name, auto launch, double bytes) {
    for(int i = 0; i < 20; ++i) launch(i);
    hipEventRecord(e0);
    for(int i = 0; i < 200; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %7.2f us per launch  %5.2f TB/s\n", name, ms / 200 * 1e3, bytes / (ms / 200 * 1e-3) * 1e-12);
  };
  timed("1024 x 1024 vertices, gen's tiles and store pattern (50.3 MB)", [&](int i) { hipLaunchKernelGGL(tile_stores, dim3(32 * 64), dim3(256), 0, 0, buf, sx, sy, 32, (float)i); }, 50331648.0);
  timed("1024 x 512 vertices, same pattern (25.2 MB)", [&](int i) { hipLaunchKernelGGL(tile_stores, dim3(32 * 32), dim3(256), 0, 0, buf, sx, sy / 2, 32, (float)i); }, 25165824.0);
  timed("50.3 MB as one dense stream, 2048 workgroups", [&](int i) { hipLaunchKernelGGL(dense_stores, dim3(2048), dim3(256), 0, 0, buf, (size_t)sx * sy * 3, (float)i); }, 50331648.0);
  timed("an empty launch (2048 workgroups, nothing stored)", [&](int i) { hipLaunchKernelGGL(dense_stores, dim3(2048), dim3(256), 0, 0, buf, (size_t)0, (float)i); }, 0.0);
  return 0;
}
