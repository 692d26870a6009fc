// Microbenchmark (round 6): the MEMORY SKELETON of the 4096^2 row pass with the fp16-stored spectrum (BASELINE configs[4]) -- every load and store of the
// kernel in its layouts, thread shapes, XCD bands and cache policies, nothing else -- with the workgroups per CU capped through unused LDS and an IDLE gap
// (s_sleep) between the arrival of a workgroup's inputs and its spectrum stores, the place of the kernel's arithmetic ("skeleton first").
// Forms:
//   K  the kernel's shape: one 512-thread workgroup per row pair, 16 points per thread, 4- and 8-byte loads, at 2 / 3 / 4 workgroups per CU
//      (3 and 4 only with h0 as halves: 96 registers of inputs in flight do not fit under 80 / 64)
//   Q  256-thread workgroups, a pair as two bursts of 16 points per thread (row p, then row N - p), 4 per CU
//   P  one PERSISTENT 1024-thread workgroup per CU that walks its pairs: the next pair's HBM inputs (phase, own h0 rows) by LDS-DMA
//      (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers) while the current pair "computes", mirror rows and dispersion
//      (L2 hits) into registers at the head of a pair
// H0B = bytes per h0 value as stored (8: float2; 4: two halves -- SURVEY.md 8d's own count for configs[4] has 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
constexpr int N = 4096, Q = N / 2 + 1, T = 256, E = 16;
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pair_of(int q) { return (q & 7) * (N / 16) + (q >> 3); }     // the kernel's XCD bands (G = 2048 groups)

template<int H0B> struct H0T { typedef float2 type; };
template<> struct H0T<4> { typedef float type; };

__device__ __forceinline__ void idle(int delay) { for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16); }

// K: THREADS = 512 (both rows at once) or Q: THREADS = 256 (row p, then row N - p)
template<int THREADS, int H0B, int MINWAVES>
__global__ void __launch_bounds__(THREADS, MINWAVES) rowskel(void const* __restrict__ h0v, float* __restrict__ phase, float const* __restrict__ omega, u2* __restrict__ spec, int delay) {
  extern __shared__ unsigned char occupancy_cap[];
  typedef typename H0T<H0B>::type HV;
  HV const *h0 = static_cast<HV const*>(h0v);
  int const p = pair_of(blockIdx.x);
  constexpr int ROUNDS = 512 / THREADS;
  size_t const plane = (size_t)N * N;
  __amdgpu_buffer_rsrc_t rph = make_rsrc(phase, plane * 4), rsp = make_rsrc(spec, plane * 8);
  #pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    int const half = (THREADS == 512) ? threadIdx.x / T : r, t = threadIdx.x % T;
    int const y = half ? (p == 0 ? N / 2 : N - p) : p;
    int const i = abs(y - N / 2);
    float ph[E], om[E]; HV a[E], b[E];
    // (the kernel's addressing: one register offset per stream, compile-time steps)
    __amdgpu_buffer_rsrc_t rh0 = make_rsrc(h0, plane * H0B), rom = make_rsrc(omega, (size_t)Q * Q * 4);
    int const e0 = y * N + t, m0 = (N - 1 - y) * N + (N - 1 - t - T * (E - 1));
    int const lower = (i * Q + N / 2 - t - T * (E / 2 - 1)) * 4, upper = (i * Q + t) * 4;
    #pragma unroll
    for (int s = 0; s < E; ++s) {
      ph[s] = buf_load_f32(rph, e0 * 4, T * s * 4);
      if constexpr (H0B == 8) { a[s] = buf_load_f32x2(rh0, e0 * 8, T * s * 8); b[s] = buf_load_f32x2(rh0, m0 * 8, T * (E - 1 - s) * 8); }
      else { a[s] = buf_load_f32(rh0, e0 * 4, T * s * 4); b[s] = buf_load_f32(rh0, m0 * 4, T * (E - 1 - s) * 4); }
      om[s] = (s < E / 2) ? buf_load_f32(rom, lower, T * (E / 2 - 1 - s) * 4) : buf_load_f32(rom, upper, T * (s - E / 2) * 4);
    }
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ph[s] + om[s]), rph, (y * N + x) * 4, 0, PHASE_STORE_AUX); }
    if (delay > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); idle(delay / ROUNDS); }
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s;
      float ax, bx; if constexpr (H0B == 8) { ax = a[s].x + a[s].y; bx = b[s].x - b[s].y; } else { ax = a[s]; bx = b[s]; }
      u2 d = { __float_as_uint(ax + bx), __float_as_uint(ph[s] * om[s]) };
      __builtin_amdgcn_raw_buffer_store_b64(d, rsp, (int)blocked<N>(y, x) * 8, 0, SPEC_STORE_AUX); }
  }
}


// S: the kernel's shape once more (512-thread pair, fp32 h0, 2 workgroups per CU) with ONE thing changed at a time: which of its six streams is the slow one?
//   DROP bit 0 phase loads, 1 own h0 loads, 2 mirror h0 loads, 3 dispersion loads, 4 phase stores, 5 spectrum stores
//   PAUX / SAUX: cache policy of the phase / spectrum stores (17 = sc0 sc1 written through: what ships; 0 plain; 2 nt)
//   LAYOUT 0: 8 x 8 blocks (64-byte block rows with the 8-byte fp16 value: what ships), 1: 8 rows x 16 columns (128-byte block rows), 2: row-major
template<int DROP, int PAUX, int SAUX, int LAYOUT>
__global__ void __launch_bounds__(512, 4) rowvar(float2 const* __restrict__ h0, float* __restrict__ phase, float const* __restrict__ omega, u2* __restrict__ spec, int delay) {
  extern __shared__ unsigned char occupancy_cap[];
  int const p = pair_of(blockIdx.x);
  size_t const plane = (size_t)N * N;
  __amdgpu_buffer_rsrc_t rph = make_rsrc(phase, plane * 4), rsp = make_rsrc(spec, plane * 8);
  int const half = threadIdx.x / T, t = threadIdx.x % T;
  int const y = half ? (p == 0 ? N / 2 : N - p) : p;
  int const i = abs(y - N / 2);
  float ph[E], om[E]; float2 a[E], b[E];
  __amdgpu_buffer_rsrc_t rh0 = make_rsrc(h0, plane * 8), rom = make_rsrc(omega, (size_t)Q * Q * 4);
  int const e0 = y * N + t, m0 = (N - 1 - y) * N + (N - 1 - t - T * (E - 1));
  int const lower = (i * Q + N / 2 - t - T * (E / 2 - 1)) * 4, upper = (i * Q + t) * 4;
  #pragma unroll
  for (int s = 0; s < E; ++s) {
    ph[s] = (DROP & 1) ? (float)t : buf_load_f32(rph, e0 * 4, T * s * 4);
    a[s] = (DROP & 2) ? make_float2(t, s) : buf_load_f32x2(rh0, e0 * 8, T * s * 8);
    b[s] = (DROP & 4) ? make_float2(s, t) : buf_load_f32x2(rh0, m0 * 8, T * (E - 1 - s) * 8);
    om[s] = (DROP & 8) ? (float)s : ((s < E / 2) ? buf_load_f32(rom, lower, T * (E / 2 - 1 - s) * 4) : buf_load_f32(rom, upper, T * (s - E / 2) * 4));
  }
  if (!(DROP & 16)) {
    #pragma unroll
    for (int s = 0; s < E; ++s) { int x = t + T * s; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ph[s] + om[s]), rph, (y * N + x) * 4, 0, PAUX); }
  }
  if (delay > 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); idle(delay); }
  float keep = 0;
  #pragma unroll
  for (int s = 0; s < E; ++s) { int x = t + T * s;
    u2 d = { __float_as_uint(a[s].x + a[s].y + b[s].x - b[s].y), __float_as_uint(ph[s] * om[s]) };
    int at;
    if (LAYOUT == 0) at = (int)blocked<N>(y, x);
    else if (LAYOUT == 1) { int const B = 128; at = (x / B) * N * B + ((y / 8) * (B / 16) + (x % B) / 16) * 128 + (y % 8) * 16 + (x % 16); }
    else at = y * N + x;
    if (!(DROP & 32)) __builtin_amdgcn_raw_buffer_store_b64(d, rsp, at * 8, 0, SAUX); else keep += __uint_as_float(d.x) + __uint_as_float(d.y); }
  if ((DROP & 32) && keep == 12345.678f) phase[0] = keep;
}

// P: persistent, 1024 threads (16 waves) per CU, pairs b, b + gridDim.x, ...  LDS: phase rows 2 x 16 KB + own h0 rows 2 x (4096 x H0B)
template<int H0B>
__global__ void __launch_bounds__(1024, 4) rowwalk(void const* __restrict__ h0v, float* __restrict__ phase, float const* __restrict__ omega, u2* __restrict__ spec, int delay, int pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename H0T<H0B>::type HV;
  HV const *h0 = static_cast<HV const*>(h0v);
  constexpr int PE = 8;                                  // points per thread and pair: 1024 threads, 8192 points
  constexpr int PHB = 2 * N * 4, HB = 2 * N * H0B;       // bytes per pair
  constexpr int PIECES = (PHB + HB) / 1024;              // 1 KiB LDS-DMA pieces per pair: 96 (fp32 h0) or 64
  int const wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  int const half = threadIdx.x / 512, t = threadIdx.x % 512;
  size_t const plane = (size_t)N * N;
  __amdgpu_buffer_rsrc_t rph = make_rsrc(phase, plane * 4), rsp = make_rsrc(spec, plane * 8);

  unsigned const ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the dynamic segment
  auto rows = [&](int item, int h) { int const p = pair_of(item); return h ? (p == 0 ? N / 2 : N - p) : p; };
  // piece k of a pair: phase row 0 (16 pieces), phase row 1, h0 row 0, h0 row 1; a piece is 1 KiB of one row
  auto dma = [&](int item) {
    #pragma unroll
    for (int j = 0; j < PIECES / 16; ++j) {
      int const k = wave + 16 * j;
      char const *src; int off;
      if (k < PHB / 1024) { int h = k / (N * 4 / 1024), c = k % (N * 4 / 1024); src = reinterpret_cast<char const*>(phase + (size_t)rows(item, h) * N) + c * 1024; off = k * 1024; }
      else { int kk = k - PHB / 1024; int h = kk / (N * H0B / 1024), c = kk % (N * H0B / 1024); src = reinterpret_cast<char const*>(h0 + (size_t)rows(item, h) * N) + c * 1024; off = PHB + kk * 1024; }
      // (inline asm: hipcc drains every vector-memory operation in flight -- vmcnt(0) -- in front of an LDS-DMA it issues itself; this one it does not
      // count, the waits are the kernel's own.  M0 = the wave-uniform LDS byte address of the piece, written in the statement that uses it)
      unsigned keep;
      unsigned const dst = __builtin_amdgcn_readfirstlane((unsigned)off) + ldsbase;
      char const *gsrc = src + lane * 16;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    }
  };

  int item = blockIdx.x;
  float om[PE], om2[PE]; HV b[PE], b2[PE];
  auto regloads = [&](int it, float (&o)[PE], HV (&m)[PE]) {
    int const y = rows(it, half); int const i = abs(y - N / 2);
    #pragma unroll
    for (int s = 0; s < PE; ++s) { int x = t + 512 * s; m[s] = h0[(size_t)(N-1-y)*N + (N-1-x)]; o[s] = omega[i * Q + abs(x - N / 2)]; }
  };
  // (raw barriers with an LDS-only wait: __syncthreads() would drain the stores and the LDS-DMA in flight with vmcnt(0); the first pair is peeled off the loop
  // so that every path into the counted wait has the same history -- the walking column pass's recipe)
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
  auto body = [&](int item, auto firsttag, float (&om)[PE], HV (&b)[PE], float (&om2)[PE], HV (&b2)[PE]) {
    constexpr bool FIRST = decltype(firsttag)::value;
    int const y = rows(item, half);
    // this pair's DMA and register loads are older than the previous pair's 16 stores per thread: a counted wait leaves those in flight
    if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    barrier();
    float ph[PE]; HV a[PE];
    #pragma unroll
    for (int s = 0; s < PE; ++s) { int x = t + 512 * s; ph[s] = reinterpret_cast<float const*>(smem)[half * N + x]; a[s] = reinterpret_cast<HV const*>(smem + PHB)[half * N + x]; }
    barrier();                                             // the staging buffer is free: the next pair's inputs
    bool const more = item + (int)gridDim.x < pairs;
    if (more) { dma(item + gridDim.x); regloads(item + gridDim.x, om2, b2); }
    #pragma unroll
    for (int s = 0; s < PE; ++s) { int x = t + 512 * s; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ph[s] + om[s]), rph, (y * N + x) * 4, 0, PHASE_STORE_AUX); }
    idle(delay);
    #pragma unroll
    for (int s = 0; s < PE; ++s) { int x = t + 512 * s;
      float ax, bx; if constexpr (H0B == 8) { ax = a[s].x + a[s].y; bx = b[s].x - b[s].y; } else { ax = a[s]; bx = b[s]; }
      u2 d = { __float_as_uint(ax + bx), __float_as_uint(ph[s] * om[s]) };
      __builtin_amdgcn_raw_buffer_store_b64(d, rsp, (int)blocked<N>(y, x) * 8, 0, SPEC_STORE_AUX); }
  };
  if (item >= pairs) return;
  // (two pairs per trip with the register sets in swapped roles: no copies between them)
  dma(item); regloads(item, om, b);
  body(item, std::true_type(), om, b, om2, b2);
  for (item += gridDim.x; item < pairs; item += 2 * gridDim.x)
  {
    body(item, std::false_type(), om2, b2, om, b);
    if (item + (int)gridDim.x < pairs)
      body(item + gridDim.x, std::false_type(), om, b, om2, b2);
  }
}

int main() {
  size_t plane = (size_t)N*N; void *h0; u2 *spec; float *phase, *omega;
  CK(hipMalloc(&h0, plane*8)); CK(hipMalloc(&phase, plane*4)); CK(hipMalloc(&spec, plane*8)); CK(hipMalloc(&omega, (size_t)Q*Q*4));
  CK(hipMemset(h0, 0, plane*8)); CK(hipMemset(phase, 0, plane*4)); CK(hipMemset(spec, 0, plane*8)); CK(hipMemset(omega, 0, (size_t)Q*Q*4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const* name, double bytes, auto fn) { for (int i=0;i<5;++i) fn(); hipEventRecord(e0); for (int i=0;i<40;++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=40; printf("%-118s %8.2f us  %6.0f GB/s\n", name, ms*1e3, bytes/ms/1e6); fflush(stdout); };
  auto cap = [&](int wgs, size_t own) { size_t lds = (size_t)(160 * 1024 / wgs) - 1024; return lds > own ? lds - own : 0; };
  #define ATTR(k) hipFuncSetAttribute(reinterpret_cast<void const*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)
  ATTR((rowskel<512, 8, 4>)); ATTR((rowskel<512, 4, 4>)); ATTR((rowskel<512, 4, 6>)); ATTR((rowskel<256, 8, 4>)); ATTR((rowskel<256, 4, 4>)); ATTR((rowwalk<8>)); ATTR((rowwalk<4>));
  double const B8 = 24.0 * plane, B4 = 20.0 * plane;     // HBM bytes by design: h0 8 (4) + phase 4 + 4 + spectrum 8
  char name[240];
  #define SVAR(DROP, PAUX, SAUX, LAYOUT, what) do { ATTR((rowvar<DROP, PAUX, SAUX, LAYOUT>)); for (int delay : {0, 30}) { \
      snprintf(name, sizeof(name), "S: %s, %4.1f us idle", what, delay * 0.427); \
      timeit(name, B8, [&]{ hipLaunchKernelGGL((rowvar<DROP, PAUX, SAUX, LAYOUT>), dim3(N/2), dim3(512), cap(2, 0), 0, (float2 const*)h0, phase, omega, spec, delay); }); } } while(0)
  if (getenv("SKEL_STREAMS")) for (int rep = 0; rep < 2; ++rep) {
    SVAR(0, 17, 17, 0, "what ships (phase and spectrum stores written through, 8 x 8 blocks)");
    SVAR(1, 17, 17, 0, "without the phase loads");
    SVAR(2, 17, 17, 0, "without the own h0 loads");
    SVAR(4, 17, 17, 0, "without the mirror h0 loads");
    SVAR(8, 17, 17, 0, "without the dispersion loads");
    SVAR(16, 17, 17, 0, "without the phase stores");
    SVAR(32, 17, 17, 0, "without the spectrum stores");
    SVAR(48, 17, 17, 0, "without any store");
    SVAR(15, 17, 17, 0, "without any load");
    SVAR(0, 0, 17, 0, "phase stores plain");
    SVAR(0, 17, 0, 0, "spectrum stores plain");
    SVAR(0, 0, 0, 0, "both store streams plain");
    SVAR(0, 2, 2, 0, "both store streams nt");
    SVAR(0, 17, 17, 1, "spectrum in 8 x 16 blocks (128-byte block rows), written through");
    SVAR(0, 17, 0, 1, "spectrum in 8 x 16 blocks, plain");
    SVAR(0, 0, 0, 1, "spectrum in 8 x 16 blocks, both store streams plain");
    SVAR(0, 17, 17, 2, "spectrum row-major, written through");
    SVAR(0, 0, 0, 2, "spectrum row-major, both store streams plain");
  }
  if (!getenv("SKEL_STREAMS")) for (int rep = 0; rep < 2; ++rep) {
    for (int delay : {0, 15, 30}) {
      snprintf(name, sizeof(name), "K: 512-thread pair, fp32 h0, 2 workgroups per CU (the kernel's shape), %4.1f us idle between inputs and spectrum stores", delay * 0.427);
      timeit(name, B8, [&]{ hipLaunchKernelGGL((rowskel<512, 8, 4>), dim3(N/2), dim3(512), cap(2, 0), 0, h0, phase, omega, spec, delay); });
      snprintf(name, sizeof(name), "K: 512-thread pair, h0 as halves, 2 workgroups per CU, %4.1f us idle", delay * 0.427);
      timeit(name, B4, [&]{ hipLaunchKernelGGL((rowskel<512, 4, 4>), dim3(N/2), dim3(512), cap(2, 0), 0, h0, phase, omega, spec, delay); });
      snprintf(name, sizeof(name), "K: 512-thread pair, h0 as halves, 3 workgroups per CU (80 registers), %4.1f us idle", delay * 0.427);
      timeit(name, B4, [&]{ hipLaunchKernelGGL((rowskel<512, 4, 6>), dim3(N/2), dim3(512), cap(3, 0), 0, h0, phase, omega, spec, delay); });
      snprintf(name, sizeof(name), "Q: 256-thread pair in two bursts, fp32 h0, 4 workgroups per CU, %4.1f us idle per pair", delay * 0.427);
      timeit(name, B8, [&]{ hipLaunchKernelGGL((rowskel<256, 8, 4>), dim3(N/2), dim3(256), cap(4, 0), 0, h0, phase, omega, spec, delay); });
      snprintf(name, sizeof(name), "Q: 256-thread pair in two bursts, h0 as halves, 4 workgroups per CU, %4.1f us idle per pair", delay * 0.427);
      timeit(name, B4, [&]{ hipLaunchKernelGGL((rowskel<256, 4, 4>), dim3(N/2), dim3(256), cap(4, 0), 0, h0, phase, omega, spec, delay); });
    }
    for (int delay : {0, 9, 14, 19}) {
      snprintf(name, sizeof(name), "P: persistent 1024-thread workgroup per CU, next pair's phase + own h0 by LDS-DMA, fp32 h0, %4.1f us idle per pair", delay * 0.427);
      timeit(name, B8, [&]{ hipLaunchKernelGGL((rowwalk<8>), dim3(256), dim3(1024), 2 * N * 4 + 2 * N * 8, 0, h0, phase, omega, spec, delay, N / 2); });
      snprintf(name, sizeof(name), "P: persistent 1024-thread workgroup per CU, next pair's phase + own h0 by LDS-DMA, h0 as halves, %4.1f us idle per pair", delay * 0.427);
      timeit(name, B4, [&]{ hipLaunchKernelGGL((rowwalk<4>), dim3(256), dim3(1024), 2 * N * 4 + 2 * N * 4, 0, h0, phase, omega, spec, delay, N / 2); });
    }
  }
  return 0;
}
