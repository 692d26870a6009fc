// Microbenchmark: does the width of a per-lane access matter at equal bytes?  Streams of 4/8/16 B per lane, reads and writes,
// plus the row-pass traffic with strided 8-byte accesses (as the kernel does) vs contiguous 16-byte accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 1024, C = 4, T = 128, E = 8;
__device__ __forceinline__ size_t blocked(int y, int x) { return ((size_t)((y >> 3) * (N / 8) + (x >> 3)) << 6) + ((y & 7) << 3) + (x & 7); }

template<typename V> __global__ void __launch_bounds__(256) rd(V const* __restrict__ in, float* out, size_t n) {
  V acc = {}; float s = 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i + 7 * st < n; i += 8 * st) {
    V a[8];
    #pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = in[i + k * st];
    #pragma unroll
    for (int k = 0; k < 8; ++k) s += *reinterpret_cast<float*>(&a[k]);
  }
  if (s == 123.456f) out[0] = s;
}
template<typename V> __global__ void __launch_bounds__(256) wr(V* __restrict__ out, size_t n, float v) {
  V a; float* pa = reinterpret_cast<float*>(&a);
  for (unsigned k = 0; k < sizeof(V)/4; ++k) pa[k] = v;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i + 7 * st < n; i += 8 * st) {
    #pragma unroll
    for (int k = 0; k < 8; ++k) out[i + k * st] = a;
  }
}
// MODE 0: the kernel's accesses (8 points x = t + 128 s per thread, 4/8-byte accesses).  MODE 1: 8 contiguous points per thread, 16-byte accesses.
template<int MODE>
__global__ void __launch_bounds__(256) rowlike(float2 const* __restrict__ h0, float* __restrict__ phase, float2* __restrict__ spec) {
  int c = blockIdx.y; int r = threadIdx.x / T, t = threadIdx.x % T; int y = blockIdx.x * 2 + r;
  size_t plane = (size_t)N * N;
  h0 += c * plane; phase += c * plane; spec += c * 3 * plane;
  float ph[E]; float2 a[E], b[E];
  if (MODE == 0) {
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; ph[s] = phase[(size_t)y*N+x]; a[s] = h0[(size_t)y*N+x]; b[s] = h0[(size_t)(N-1-y)*N + (N-1-x)]; }
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; phase[(size_t)y*N+x] = ph[s] + 1.0f;
      float2 v0 = make_float2(a[s].x + b[s].x, a[s].y - b[s].y), v1 = make_float2(a[s].y, b[s].x), v2 = make_float2(ph[s], a[s].x);
      size_t o = blocked(y, x); spec[o] = v0; spec[plane + o] = v1; spec[2*plane + o] = v2; }
  } else {
    int x0 = 8 * t;
    float4 const* p4 = reinterpret_cast<float4 const*>(phase + (size_t)y*N + x0);
    float4 const* a4 = reinterpret_cast<float4 const*>(h0 + (size_t)y*N + x0);
    float4 const* b4 = reinterpret_cast<float4 const*>(h0 + (size_t)(N-1-y)*N + (N-8-x0));
    float4 P[2], A[4], B[4];
    #pragma unroll
    for (int k = 0; k < 2; ++k) P[k] = p4[k];
    #pragma unroll
    for (int k = 0; k < 4; ++k) { A[k] = a4[k]; B[k] = b4[k]; }
    float4* q4 = reinterpret_cast<float4*>(phase + (size_t)y*N + x0);
    #pragma unroll
    for (int k = 0; k < 2; ++k) q4[k] = make_float4(P[k].x + 1.0f, P[k].y + 1.0f, P[k].z + 1.0f, P[k].w + 1.0f);
    size_t o = blocked(y, x0);
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v0 = make_float4(A[k].x + B[k].x, A[k].y - B[k].y, A[k].z + B[k].z, A[k].w - B[k].w), v1 = make_float4(A[k].y, B[k].x, A[k].w, B[k].z), v2 = make_float4(P[k/2].x, A[k].x, P[k/2].y, A[k].z);
      *reinterpret_cast<float4*>(spec + o + 2*k) = v0; *reinterpret_cast<float4*>(spec + plane + o + 2*k) = v1; *reinterpret_cast<float4*>(spec + 2*plane + o + 2*k) = v2; }
  }
}
// map-store pattern of the column pass: a wave stores float4 to RUN consecutive texels of each of 64*4/RUN... rows (row pitch N*16 B),
// 2 layers; AUX = cache policy of the stores (17 = sc0 sc1)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template<int RUN, int AUX>
__global__ void __launch_bounds__(512) mapstore(float4* __restrict__ maps, float v) {
  // tile of RUN columns; 512 threads: cp = tid % RUN, t = tid / RUN; rows y = t + (512/RUN) * s, s < E2 so that the tile covers 1024 rows
  constexpr int TT = 512 / RUN, E2 = 1024 / TT;
  int c = blockIdx.y, tile = blockIdx.x; int cp = threadIdx.x % RUN, t = threadIdx.x / RUN; int x = tile * RUN + cp;
  size_t plane = (size_t)N * N; float4* l0 = maps + (size_t)c * 2 * plane;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(l0, 0, (int)(2 * plane * 16), 0x00020000);
  u4 d = { __float_as_uint(v), __float_as_uint(v + 1), __float_as_uint(v + 2), 0u };
  #pragma unroll
  for (int s = 0; s < E2; ++s) {
    int y = t + TT * s; int o = (y * N + x) * 16;
    __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, o + (int)(plane * 16), 0, AUX);
  }
}
// the same map-store pattern writing only x, y, z of every texel (12 of 16 bytes: the w components stay as they are)
typedef unsigned int u3 __attribute__((ext_vector_type(3)));
template<int RUN, int AUX>
__global__ void __launch_bounds__(512) mapstore12(float4* __restrict__ maps, float v) {
  constexpr int TT = 512 / RUN, E2 = 1024 / TT;
  int c = blockIdx.y, tile = blockIdx.x; int cp = threadIdx.x % RUN, t = threadIdx.x / RUN; int x = tile * RUN + cp;
  size_t plane = (size_t)N * N; float4* l0 = maps + (size_t)c * 2 * plane;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(l0, 0, (int)(2 * plane * 16), 0x00020000);
  u3 d = { __float_as_uint(v), __float_as_uint(v + 1), __float_as_uint(v + 2) };
  #pragma unroll
  for (int s = 0; s < E2; ++s) {
    int y = t + TT * s; int o = (y * N + x) * 16;
    __builtin_amdgcn_raw_buffer_store_b96(d, r, o, 0, AUX);
    __builtin_amdgcn_raw_buffer_store_b96(d, r, o + (int)(plane * 16), 0, AUX);
  }
}
int main() {
  size_t plane = (size_t)N*N; float2 *h0, *spec; float *phase; float *buf;
  size_t nb = (size_t)168 << 20;
  CK(hipMalloc(&h0, C*plane*8)); CK(hipMalloc(&phase, C*plane*4)); CK(hipMalloc(&spec, C*3*plane*8)); CK(hipMalloc(&buf, nb));
  CK(hipMemset(h0, 0, C*plane*8)); CK(hipMemset(phase, 0, C*plane*4)); CK(hipMemset(spec, 0, C*3*plane*8)); CK(hipMemset(buf,0,nb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<5;++i) fn(); hipEventRecord(e0); for (int i=0;i<50;++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=50; printf("%-40s %8.1f us  %7.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); };
  for (int g : {2048, 8192}) {
    printf("grid %d x 256 threads, 168 MiB\n", g);
    timeit("  read  4 B/lane", nb, [&]{ hipLaunchKernelGGL(rd<float>,  dim3(g), dim3(256), 0, 0, (float const*)buf,  buf, nb/4); });
    timeit("  read  8 B/lane", nb, [&]{ hipLaunchKernelGGL(rd<float2>, dim3(g), dim3(256), 0, 0, (float2 const*)buf, buf, nb/8); });
    timeit("  read 16 B/lane", nb, [&]{ hipLaunchKernelGGL(rd<float4>, dim3(g), dim3(256), 0, 0, (float4 const*)buf, buf, nb/16); });
    timeit("  write 4 B/lane", nb, [&]{ hipLaunchKernelGGL(wr<float>,  dim3(g), dim3(256), 0, 0, (float*)buf,  nb/4, 1.0f); });
    timeit("  write 8 B/lane", nb, [&]{ hipLaunchKernelGGL(wr<float2>, dim3(g), dim3(256), 0, 0, (float2*)buf, nb/8, 1.0f); });
    timeit("  write 16 B/lane", nb, [&]{ hipLaunchKernelGGL(wr<float4>, dim3(g), dim3(256), 0, 0, (float4*)buf, nb/16, 1.0f); });
  }
  { float4* maps; CK(hipMalloc(&maps, C*2*plane*16)); double mb = 32.0*C*plane;
    timeit("map stores, 64-B runs (4-column tiles), plain", mb, [&]{ hipLaunchKernelGGL((mapstore<4, 0>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 64-B runs, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<4, 17>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 12 of 16 B per texel, 64-B runs, plain", mb, [&]{ hipLaunchKernelGGL((mapstore12<4, 0>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 12 of 16 B per texel, 64-B runs, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore12<4, 17>), dim3(N/4, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 128-B runs (8-column tiles), plain", mb, [&]{ hipLaunchKernelGGL((mapstore<8, 0>), dim3(N/8, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 128-B runs, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<8, 17>), dim3(N/8, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 256-B runs (16-column tiles), sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<16, 17>), dim3(N/16, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 32-B runs (2-column tiles), plain", mb, [&]{ hipLaunchKernelGGL((mapstore<2, 0>), dim3(N/2, C), dim3(512), 0, 0, maps, 1.0f); });
    timeit("map stores, 32-B runs, sc0 sc1", mb, [&]{ hipLaunchKernelGGL((mapstore<2, 17>), dim3(N/2, C), dim3(512), 0, 0, maps, 1.0f); });
  }
  double rowb = 40.0*C*plane;
  timeit("rowlike strided 4/8-byte accesses", rowb, [&]{ hipLaunchKernelGGL(rowlike<0>, dim3(N/2, C), dim3(256), 0, 0, h0, phase, spec); });
  timeit("rowlike contiguous 16-byte accesses", rowb, [&]{ hipLaunchKernelGGL(rowlike<1>, dim3(N/2, C), dim3(256), 0, 0, h0, phase, spec); });
  return 0;
}
