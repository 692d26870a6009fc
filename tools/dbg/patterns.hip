// Microbenchmark of the memory access patterns of the two ocean kernels (no arithmetic), 4 x 1024^2 points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 1024, C = 4, T = 128, E = 8;
__device__ __forceinline__ size_t blocked(int y, int x) { return ((size_t)((y >> 3) * (N / 8) + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7); }

__global__ void copy4(float4 const* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// row-pass traffic: 2 rows per 256-thread WG; loads phase(4) h0(8) h0 mirror(8); stores phase(4) + 3 fields (8) ; MODE 0 blocked, 1 row-major
template<int MODE>
__global__ void __launch_bounds__(256) rowlike(float2 const* __restrict__ h0, float* __restrict__ phase, float2* __restrict__ spec) {
  int c = blockIdx.y; int r = threadIdx.x / T, t = threadIdx.x % T; int y = blockIdx.x * 2 + r;
  size_t plane = (size_t)N * N;
  h0 += c * plane; phase += c * plane; spec += c * 3 * plane;
  float ph[E]; float2 a[E], b[E];
  #pragma unroll
  for (int s = 0; s < E; ++s) { int x = t + T * s; ph[s] = phase[(size_t)y*N+x]; a[s] = h0[(size_t)y*N+x]; b[s] = h0[(size_t)(N-1-y)*N + (N-1-x)]; }
  #pragma unroll
  for (int s = 0; s < E; ++s) { int x = t + T * s; phase[(size_t)y*N+x] = ph[s] + 1.0f;
    float2 v0 = make_float2(a[s].x + b[s].x, a[s].y - b[s].y), v1 = make_float2(a[s].y, b[s].x), v2 = make_float2(ph[s], a[s].x);
    size_t o = MODE == 0 ? blocked(y, x) : (size_t)y * N + x;
    spec[o] = v0; spec[plane + o] = v1; spec[2*plane + o] = v2; }
}
// column-pass traffic: tile of 8 columns, 512 threads (cp 4 x t 128); loads 3 x 16 B blocked; stores 2 layers x 2 texels x 16 B. MODE 0 pairs, 1 full-line lanes
template<int MODE>
__global__ void __launch_bounds__(512) collike(float2 const* __restrict__ spec, float4* __restrict__ maps) {
  int c = blockIdx.y; int x0 = blockIdx.x * 8; int cp = threadIdx.x % 4, t = threadIdx.x / 4; int xa = x0 + 2*cp;
  size_t plane = (size_t)N * N;
  spec += c * 3 * plane; float4* l0 = maps + (size_t)c * 2 * plane; float4* l1 = l0 + plane;
  float4 q[3][E];
  #pragma unroll
  for (int f = 0; f < 3; ++f)
  #pragma unroll
  for (int s = 0; s < E; ++s) q[f][s] = *reinterpret_cast<float4 const*>(spec + f*plane + blocked(t + T*s, xa));
  #pragma unroll
  for (int s = 0; s < E; ++s) {
    float4 A = make_float4(q[0][s].x, q[1][s].x, q[2][s].x, 0.f), B = make_float4(q[0][s].z, q[1][s].z, q[2][s].z, 0.f);
    float4 Cn = make_float4(q[0][s].y, q[1][s].y, q[2][s].y, 0.f), D = make_float4(q[0][s].w, q[1][s].w, q[2][s].w, 0.f);
    if (MODE == 0) { size_t o = (size_t)(t + T*s) * N + xa; l0[o] = A; l0[o+1] = B; l1[o] = Cn; l1[o+1] = D; }
    else { // same bytes, but each instruction covers full 128-B lines: lane -> (x = l%8, row = l/8) within the wave's 16 rows
      int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; int xx = lane & 7, rr = lane >> 3;
      size_t o = (size_t)(wave*16 + rr + T*s) * N + x0 + xx; size_t o2 = (size_t)(wave*16 + 8 + rr + T*s) * N + x0 + xx;
      l0[o] = A; l0[o2] = B; l1[o] = Cn; l1[o2] = D; }
  }
}
int main() {
  size_t plane = (size_t)N*N; float2 *h0, *spec; float *phase; float4 *maps, *cpy;
  CK(hipMalloc(&h0, C*plane*8)); CK(hipMalloc(&phase, C*plane*4)); CK(hipMalloc(&spec, C*3*plane*8)); CK(hipMalloc(&maps, C*2*plane*16)); CK(hipMalloc(&cpy, C*2*plane*16));
  CK(hipMemset(h0, 0, C*plane*8)); CK(hipMemset(phase, 0, C*plane*4)); CK(hipMemset(spec, 0, C*3*plane*8)); CK(hipMemset(maps,0,C*2*plane*16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<5;++i) fn(); hipEventRecord(e0); for (int i=0;i<50;++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=50; printf("%-34s %8.1f us  %7.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); };
  size_t n4 = C*2*plane; // 134 MB in + 134 MB out
  timeit("copy float4 (268 MB moved)", 2.0*n4*16, [&]{ hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, maps, cpy, n4); });
  timeit("copy float4 grid=8192", 2.0*n4*16, [&]{ hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, 0, maps, cpy, n4); });
  double rowb = 40.0*C*plane, colb = 56.0*C*plane;
  timeit("rowlike blocked stores", rowb, [&]{ hipLaunchKernelGGL(rowlike<0>, dim3(N/2, C), dim3(256), 0, 0, h0, phase, spec); });
  timeit("rowlike row-major stores", rowb, [&]{ hipLaunchKernelGGL(rowlike<1>, dim3(N/2, C), dim3(256), 0, 0, h0, phase, spec); });
  timeit("collike pair stores", colb, [&]{ hipLaunchKernelGGL(collike<0>, dim3(N/8, C), dim3(512), 0, 0, spec, maps); });
  timeit("collike full-line stores", colb, [&]{ hipLaunchKernelGGL(collike<1>, dim3(N/8, C), dim3(512), 0, 0, spec, maps); });
  timeit("row+col back to back (403 MB)", rowb+colb, [&]{ hipLaunchKernelGGL(rowlike<0>, dim3(N/2, C), dim3(256), 0, 0, h0, phase, spec); hipLaunchKernelGGL(collike<0>, dim3(N/8, C), dim3(512), 0, 0, spec, maps); });
  return 0;
}
