// Diagnostic build (never shipped): s_memrealtime stamps per phase of the two ocean kernels and where each workgroup
// ran (XCC, SE, CU), 1024^2 x 4, random inputs.  Prints phase durations and, per CU, how phases of co-resident
// workgroups overlap in time.
#define OCEAN_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#include <cmath>
#include <random>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
int main() {
  
#ifndef STAMP_N
#define STAMP_N 1024
#endif
#ifndef STAMP_C
#define STAMP_C 4
#endif
#ifndef STAMP_H16
#define STAMP_H16 false
#endif
  constexpr int N = STAMP_N, C = STAMP_C; constexpr bool H = STAMP_H16; typedef RowCfg<N, H> RC; constexpr bool RW = false, CW = col_walks<N, H>(); size_t P = (size_t)N*N;
  StepArgs a{};
  float2 *h0; float *phase; cd *spec; cf *tw; float4 *maps; float *omega; unsigned long long *stamps;
  CK(hipMalloc(&h0, C*P*8)); CK(hipMalloc(&phase, C*P*4)); CK(hipMalloc(&spec, C*P*16)); CK(hipMalloc(&maps, C*2*P*16));
  CK(hipMalloc(&tw, N*8)); CK(hipMalloc(&omega, (size_t)C*(N/2+1)*(N/2+1)*4));
  size_t nst = (size_t)2*65536*16; CK(hipMalloc(&stamps, nst*8)); CK(hipMemset(stamps, 0, nst*8));
  std::mt19937 rng(1); std::normal_distribution<float> nd;
  { std::vector<float> h(C*P*2); for (auto &v : h) v = 0.01f*nd(rng); CK(hipMemcpy(h0, h.data(), h.size()*4, hipMemcpyHostToDevice));
    std::vector<float> ph(C*P); for (auto &v : ph) v = 3.0f + nd(rng)*0.5f; CK(hipMemcpy(phase, ph.data(), ph.size()*4, hipMemcpyHostToDevice));
    std::vector<float> om((size_t)C*(N/2+1)*(N/2+1), 1.0f); CK(hipMemcpy(omega, om.data(), om.size()*4, hipMemcpyHostToDevice));
    std::vector<cf> t(N); for (int k=0;k<N;++k) t[k] = cf{(float)cos(2*M_PI*k/N),(float)sin(2*M_PI*k/N)}; CK(hipMemcpy(tw, t.data(), N*8, hipMemcpyHostToDevice)); }
  a.h0=h0; a.phase=phase; a.spec=spec; a.maps=maps; a.tw=tw; a.omega=omega; a.ndt=1; a.cascades=C; a.dt[0]=1.f/60; a.stamps=stamps;
  for (int c=0;c<DATUM_OCEAN_MAX_CASCADES;++c) a.casc[c] = CascadeConst{22.f, 1/22.f, 1.35f, 4/(N/22.f), 1.f, 1.f};
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC::LDS));
  CK(hipFuncSetAttribute(colpass_entry<N, H>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ColCfg<N>::LDS));
  for (int it = 0; it < 5; ++it) {
    hipLaunchKernelGGL((ocean_rowpass_kernel<N, H>), dim3(RW ? 256 : RC::GROUPS * C), dim3(RC::THREADS), RC::LDS, 0, a);
    { void *args[] = { &a }; CK(hipLaunchKernel(colpass_entry<N, H>(), dim3(CW ? 256 : ColCfg<N>::TILES * C), dim3(ColCfg<N>::THREADS), args, ColCfg<N>::LDS, 0)); }
  }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(nst); CK(hipMemcpy(st.data(), stamps, nst*8, hipMemcpyDeviceToHost));
  auto analyse = [&](char const *name, size_t base, int nwg, int nph, std::vector<char const*> labels) {
    printf("== %s: %d workgroups; times in us (s_memrealtime, 10 ns ticks)\n", name, nwg);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nwg; ++b) { auto *s = &st[(base + b) * 16]; t0 = std::min(t0, s[0]); t1 = std::max(t1, s[nph]); }
    printf("   first start -> last end: %.2f us\n", (t1 - t0) * 0.01);
    for (int k = 0; k < nph; ++k) {
      std::vector<double> d; for (int b = 0; b < nwg; ++b) { auto *s = &st[(base + b) * 16]; d.push_back((s[k+1] - s[k]) * 0.01); }
      std::sort(d.begin(), d.end());
      printf("   %-44s median %6.2f  p10 %6.2f  p90 %6.2f\n", labels[k], d[d.size()/2], d[d.size()/10], d[d.size()*9/10]);
    }
    { std::vector<double> d; for (int b = 0; b < nwg; ++b) { auto *s = &st[(base + b) * 16]; d.push_back((s[nph] - s[0]) * 0.01); } std::sort(d.begin(), d.end());
      printf("   %-44s median %6.2f  p10 %6.2f  p90 %6.2f\n", "whole workgroup (to last instruction)", d[d.size()/2], d[d.size()/10], d[d.size()*9/10]); }
    // placement
    std::map<unsigned, std::vector<int>> bycu;
    for (int b = 0; b < nwg; ++b) { auto *s = &st[(base + b) * 16]; unsigned hw = (unsigned)s[15], xcc = (unsigned)s[14] & 15; unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; bycu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b); }
    printf("   distinct CUs seen: %zu; workgroups per CU: ", bycu.size());
    { std::map<size_t,int> hist; for (auto &kv : bycu) hist[kv.second.size()]++; for (auto &h : hist) printf("%zu:%d ", h.first, h.second); printf("\n"); }
    // start-time histogram (when do workgroups start, relative to the first)
    { std::vector<double> d; for (int b = 0; b < nwg; ++b) d.push_back((st[(base + b) * 16] - t0) * 0.01); std::sort(d.begin(), d.end());
      printf("   start times: p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f\n", d[d.size()/10], d[d.size()/4], d[d.size()/2], d[d.size()*3/4], d[d.size()*9/10], d.back()); }
    // chip-wide occupancy of each phase over time (1 us bins)
    int nb = (int)((t1 - t0) / 100) + 1; std::vector<std::vector<double>> occ(nph, std::vector<double>(nb, 0.0));
    for (int b = 0; b < nwg; ++b) { auto *s = &st[(base + b) * 16]; for (int k = 0; k < nph; ++k) for (unsigned long long t = s[k]; t < s[k+1]; ++t) occ[k][(t - t0) / 100] += 0.01; }
    printf("   workgroups in each phase, per 1-us bin (chip-wide):\n");
    for (int k = 0; k < nph; ++k) { printf("   %-30.30s", labels[k]); for (int i = 0; i < nb && i < 48; ++i) printf("%4.0f", occ[k][i]); printf("\n"); }
    // timeline of the first two CUs
    int shown = 0;
    for (auto &kv : bycu) { if (shown++ >= 2) break; printf("   CU %06x:", kv.first); auto v = kv.second; std::sort(v.begin(), v.end(), [&](int x, int y){ return st[(base+x)*16] < st[(base+y)*16]; });
      for (int b : v) { auto *s = &st[(base + b) * 16]; printf("  [wg %d:", b); for (int k = 0; k <= nph; ++k) printf(" %.1f", (s[k] - t0) * 0.01); printf("]"); } printf("\n"); }
  };
  analyse("rowpass", 0, RW ? 256 : RC::GROUPS * C, 5, {"start -> inputs arrived", "advance + phase stores + sim + swap barrier", "build C, D + barrier", "2-field transform (6 barriers)", "spectrum stores issued"});
  analyse("colpass", 65536, CW ? 256 : ColCfg<N>::TILES * C, 4, {"start -> inputs arrived", "2-field transform (6 barriers)", "height exchange + barrier", "normals + map stores issued"});
  return 0;
}
