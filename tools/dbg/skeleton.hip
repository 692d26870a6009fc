// Microbenchmark (round 5): the MEMORY SKELETONS of the two step kernels at 1024^2 x 4 -- every load and store the kernels make, in their layouts,
// thread shapes and cache policies, and nothing else (no arithmetic, no LDS, no barriers).  What the kernels take beyond these times is what their
// arithmetic, LDS exchanges and barriers fail to hide.  Row pass: also with 16-byte loads and phase stores in place of the kernel's 4- and 8-byte ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 1024, C = 4, Q = N / 2 + 1;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// row pass: 256 threads = rows p and N - p, thread t of 128 per row holds x = t + 128 s (MODE 0) or four consecutive x in each half of the row (MODE 1)
template<int MODE>
__global__ void __launch_bounds__(1024) rowskel(float2 const* __restrict__ h0, float* __restrict__ phase, float const* __restrict__ omega, float4* __restrict__ spec, float const* __restrict__ omegawide, int delay = 0, int per = 1) {
  extern __shared__ unsigned char occupancy_cap[];      // (dynamic LDS of the launch: caps the workgroups per CU, nothing is stored there)
  constexpr int T = 128, E = 8;
  float2 const *h0_ = h0; float *phase_ = phase; float4 *spec_ = spec; float const *omega_ = omega, *omegawide_ = omegawide;
  for (int rep = 0; rep < per; ++rep) {
  h0 = h0_; phase = phase_; spec = spec_; omega = omega_; omegawide = omegawide_;
  int const side = blockDim.x >= 256 ? blockDim.x / 256 : 1;     // row pairs side by side in one workgroup (256 threads each)
  int const sub = blockDim.x >= 256 ? 1 : 256 / blockDim.x;       // or workgroups per pair (128 threads: one row each; 64: half a row)
  int const tid = blockDim.x >= 256 ? threadIdx.x % 256 : (blockIdx.x % sub) * blockDim.x + threadIdx.x;
  int const item = ((blockIdx.x / sub) * side + threadIdx.x / 256) * per + rep; int const c = item / (N / 2), q = item % (N / 2); int const p = (q & 7) * (N / 16) + (q >> 3);       // the kernel's XCD bands
  int const half = tid / T, t = tid % T; int const y = half ? (p == 0 ? N / 2 : N - p) : p;
  size_t const plane = (size_t)N * N;
  h0 += c * plane; phase += c * plane; spec += c * plane; omega += (size_t)c * Q * Q; omegawide += (size_t)c * Q * N;
  __amdgpu_buffer_rsrc_t rph = make_rsrc(phase, plane * 4), rsp = make_rsrc(spec, plane * 16);
  int const i = abs(y - N / 2);
  float ph[E], om[E]; float2 a[E], b[E];
  if (MODE == 0) {
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; ph[s] = phase[(size_t)y*N+x]; a[s] = h0[(size_t)y*N+x]; b[s] = h0[(size_t)(N-1-y)*N + (N-1-x)]; om[s] = omega[i * Q + abs(x - N / 2)]; }
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ph[s] + om[s]), rph, (y * N + x) * 4, 0, PHASE_STORE_AUX); }
  } else {
    #pragma unroll
    for (int j = 0; j < 2; ++j) {
      int x0 = 512 * j + 4 * t;
      float4 P = *reinterpret_cast<float4 const*>(phase + (size_t)y*N + x0);
      float4 O = *reinterpret_cast<float4 const*>(omegawide + (size_t)i*N + x0);
      float4 A0 = *reinterpret_cast<float4 const*>(h0 + (size_t)y*N + x0), A1 = *reinterpret_cast<float4 const*>(h0 + (size_t)y*N + x0 + 2);
      float4 B0 = *reinterpret_cast<float4 const*>(h0 + (size_t)(N-1-y)*N + (N-4-x0)), B1 = *reinterpret_cast<float4 const*>(h0 + (size_t)(N-1-y)*N + (N-4-x0) + 2);
      ph[4*j+0] = P.x; ph[4*j+1] = P.y; ph[4*j+2] = P.z; ph[4*j+3] = P.w; om[4*j+0] = O.x; om[4*j+1] = O.y; om[4*j+2] = O.z; om[4*j+3] = O.w;
      a[4*j+0] = make_float2(A0.x, A0.y); a[4*j+1] = make_float2(A0.z, A0.w); a[4*j+2] = make_float2(A1.x, A1.y); a[4*j+3] = make_float2(A1.z, A1.w);
      b[4*j+3] = make_float2(B0.x, B0.y); b[4*j+2] = make_float2(B0.z, B0.w); b[4*j+1] = make_float2(B1.x, B1.y); b[4*j+0] = make_float2(B1.z, B1.w);
    }
    #pragma unroll
    for (int j = 0; j < 2; ++j) { int x0 = 512 * j + 4 * t;
      u4 d = { __float_as_uint(ph[4*j] + om[4*j]), __float_as_uint(ph[4*j+1] + om[4*j+1]), __float_as_uint(ph[4*j+2] + om[4*j+2]), __float_as_uint(ph[4*j+3] + om[4*j+3]) };
      __builtin_amdgcn_raw_buffer_store_b128(d, rph, (y * N + x0) * 4, 0, PHASE_STORE_AUX); }
  }
  // the kernel's arithmetic between its loads and its spectrum stores, as idle time: `delay` x 0.43 us (s_sleep 16 = 1024 clocks) once the inputs are there
  if (delay > 0) { float keep = ph[0] + ph[E - 1] + a[0].x + b[E - 1].y + om[E - 1]; asm volatile("s_waitcnt vmcnt(0)" :: "v"(keep) : "memory"); for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16); }
  #pragma unroll
  for (int s = 0; s < E; ++s) { int x = t + T * s;
    u4 d = { __float_as_uint(a[s].x + b[s].x), __float_as_uint(a[s].y - b[s].y), __float_as_uint(ph[s]), __float_as_uint(om[s]) };
    __builtin_amdgcn_raw_buffer_store_b128(d, rsp, (int)blocked<N>(y, x) * 16, 0, SPEC_STORE_AUX); }
  }
}

// column pass: a tile of four columns per 256-thread workgroup, threads column-fastest, 16 rows y = t + 64 s per thread: one 16-byte load per point from
// the blocked spectrum, one 16-byte and one 8-byte written-through store per texel into the patch layout
__global__ void __launch_bounds__(256) colskel(float4 const* __restrict__ spec, char* __restrict__ maps) {
  constexpr int W = 4, T = 64, E = 16, NT = N / W;
  int const item = blockIdx.x; int const c = item / NT, q = item % NT; int const tile = (q & 7) * (NT / 8) + (q >> 3);
  int const cp = threadIdx.x % W, t = threadIdx.x / W; int const x = tile * W + cp;
  size_t const plane = (size_t)N * N;
  __amdgpu_buffer_rsrc_t rsp = make_rsrc(spec + c * plane, plane * 16), rmp = make_rsrc(maps + (size_t)c * map_cascade_bytes(N), map_cascade_bytes(N));
  float4 v[E];
  #pragma unroll
  for (int s = 0; s < E; ++s) v[s] = buf_load_f32x4_aux<0>(rsp, (int)blocked<N>(t + T * s, x) * 16, 0);
  #pragma unroll
  for (int s = 0; s < E; ++s) { int y = t + T * s;
    buf_store_f32x4_aux<MAP_STORE_AUX>(make_float4(v[s].x, v[s].y, v[s].z, v[s].w + 1.0f), rmp, (int)map_compact_a(N, y, x), 0);
    buf_store_cf_aux<MAP_STORE_AUX>(cf{ v[s].x + v[s].z, v[s].y }, rmp, (int)map_compact_b(N, y, x), 0); }
}
int main() {
  size_t plane = (size_t)N*N; float2 *h0; float4 *spec; float *phase, *omega, *omegawide; char *maps;
  CK(hipMalloc(&h0, C*plane*8)); CK(hipMalloc(&phase, C*plane*4)); CK(hipMalloc(&spec, C*plane*16)); CK(hipMalloc(&omega, (size_t)C*Q*Q*4)); CK(hipMalloc(&omegawide, (size_t)C*Q*N*4)); CK(hipMalloc(&maps, C*map_cascade_bytes(N)));
  CK(hipMemset(h0, 0, C*plane*8)); CK(hipMemset(phase, 0, C*plane*4)); CK(hipMemset(spec, 0, C*plane*16)); CK(hipMemset(omega, 0, (size_t)C*Q*Q*4)); CK(hipMemset(omegawide, 0, (size_t)C*Q*N*4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<20;++i) fn(); hipEventRecord(e0); for (int i=0;i<200;++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=200; printf("%-78s %8.2f us  %7.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); };
  for (int rep = 0; rep < 3; ++rep) {
    timeit("row pass skeleton (32 B/pt by design), the kernel's 4- and 8-byte loads", 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C), dim3(256), 0, 0, h0, phase, omega, spec, omegawide); });
    timeit("row pass skeleton, 16-byte loads and phase stores instead", 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<1>, dim3(N/2*C), dim3(256), 0, 0, h0, phase, omega, spec, omegawide); });
    timeit("column pass skeleton (40 B/pt)", 40.0*C*plane, [&]{ hipLaunchKernelGGL(colskel, dim3(N/4*C), dim3(256), 0, 0, spec, maps); });
    for (int wgs : {8, 6, 4}) for (int delay : {0, 6, 12}) { char name[160]; snprintf(name, sizeof(name), "row pass skeleton, %d workgroups per CU, %.1f us idle between loads and spectrum stores", wgs, delay * 0.427);
      size_t lds = wgs == 8 ? 0 : (size_t)(160 * 1024 / wgs) - 512; hipFuncSetAttribute(reinterpret_cast<void const*>(&rowskel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      timeit(name, 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C), dim3(256), lds, 0, h0, phase, omega, spec, omegawide, delay); }); }
    for (int per : {1, 2, 4, 8}) { char name[160]; snprintf(name, sizeof(name), "row pass skeleton, %d pairs per workgroup one after the other (%d workgroups)", per, N / 2 * C / per);
      timeit(name, 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C/per), dim3(256), 0, 0, h0, phase, omega, spec, omegawide, 0, per); }); }
    for (int threads : {128, 64}) { char name[160]; snprintf(name, sizeof(name), "row pass skeleton, %d workgroups of %d threads (a pair split over %d workgroups)", N / 2 * C * 256 / threads, threads, 256 / threads);
      timeit(name, 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C*256/threads), dim3(threads), 0, 0, h0, phase, omega, spec, omegawide, 0, 1); }); }
    for (int side : {1, 2, 4}) { char name[160]; snprintf(name, sizeof(name), "row pass skeleton, %d pairs SIDE BY SIDE per workgroup (%d workgroups of %d threads)", side, N / 2 * C / side, 256 * side);
      timeit(name, 32.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C/side), dim3(256 * side), 0, 0, h0, phase, omega, spec, omegawide, 0, 1); }); }
    timeit("both, back to back (72 B/pt: the step)", 72.0*C*plane, [&]{ hipLaunchKernelGGL(rowskel<0>, dim3(N/2*C), dim3(256), 0, 0, h0, phase, omega, spec, omegawide); hipLaunchKernelGGL(colskel, dim3(N/4*C), dim3(256), 0, 0, spec, maps); });
  }
  return 0;
}
