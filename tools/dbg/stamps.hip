// Diagnostic build (never shipped): per-phase s_memtime stamps of the two ocean kernels, 1024^2 x 4, random inputs.
#define OCEAN_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cmath>
#include <random>
#include "../../datum_amd/csrc/ocean_kernels.hip"
using namespace ocean;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
int main() {
  constexpr int N = 1024, C = 4; size_t P = (size_t)N*N; constexpr int PAIRBLOCKS = 683;
  StepArgs a{};
  float2 *h0; float *phase; cf *spec, *halo, *tw; float4 *maps; float *omega; unsigned long long *stamps;
  CK(hipMalloc(&h0, C*P*8)); CK(hipMalloc(&phase, C*P*4)); CK(hipMalloc(&spec, C*3*P*8)); CK(hipMalloc(&maps, C*2*P*16));
  CK(hipMalloc(&halo, (size_t)C*TileCfg<N>::TILES*2*N*8)); CK(hipMalloc(&tw, N*8)); CK(hipMalloc(&omega, (size_t)C*(N/2+1)*(N/2+1)*4));
  size_t nst = (size_t)2*C*8192*16; CK(hipMalloc(&stamps, nst*8)); CK(hipMemset(stamps, 0, nst*8));
  std::mt19937 rng(1); std::normal_distribution<float> nd;
  { std::vector<float> h(C*P*2); for (auto &v : h) v = 0.01f*nd(rng); CK(hipMemcpy(h0, h.data(), h.size()*4, hipMemcpyHostToDevice));
    std::vector<float> ph(C*P); for (auto &v : ph) v = 3.0f + nd(rng)*0.5f; CK(hipMemcpy(phase, ph.data(), ph.size()*4, hipMemcpyHostToDevice));
    std::vector<float> om((size_t)C*(N/2+1)*(N/2+1), 1.0f); CK(hipMemcpy(omega, om.data(), om.size()*4, hipMemcpyHostToDevice));
    std::vector<cf> t(N); for (int k=0;k<N;++k) t[k] = cf{(float)cos(2*M_PI*k/N),(float)sin(2*M_PI*k/N)}; CK(hipMemcpy(tw, t.data(), N*8, hipMemcpyHostToDevice)); }
  a.h0=h0; a.phase=phase; a.spec=spec; a.maps=maps; a.tw=tw; a.omega=omega; a.halo=halo; a.ndt=1; a.cascades=C; a.dt[0]=1.f/60; a.stamps=stamps;
  for (int c=0;c<DATUM_OCEAN_MAX_CASCADES;++c) a.casc[c] = CascadeConst{22.f, 1/22.f, 1.35f, 4/(N/22.f)};
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N>::LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpair_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PairCfg<N>::LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_colpass_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ColCfg<N>::LDS));
  for (int it = 0; it < 5; ++it) {
#if OCEAN_ROW_PAIRED
    hipLaunchKernelGGL(ocean_rowpair_kernel<N>, dim3(PAIRBLOCKS), dim3(PairCfg<N>::THREADS), PairCfg<N>::LDS, 0, a);
#else
    hipLaunchKernelGGL(ocean_rowpass_kernel<N>, dim3(RowCfg<N>::BLOCKS, C), dim3(RowCfg<N>::THREADS), RowCfg<N>::LDS, 0, a);
#endif
    hipLaunchKernelGGL(ocean_colpass_kernel<N>, dim3(N/ColCfg<N>::W, C), dim3(ColCfg<N>::THREADS), ColCfg<N>::LDS, 0, a);
  }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(nst); CK(hipMemcpy(st.data(), stamps, nst*8, hipMemcpyDeviceToHost));
  auto report = [&](char const* name, size_t base, int nwg_x, int ny, std::vector<std::pair<int,int>> segs, std::vector<char const*> labels) {
    printf("%s (s_memtime ticks = shader cycles; median over workgroups)\n", name);
    for (size_t k = 0; k < segs.size(); ++k) {
      std::vector<double> d;
      for (int y = 0; y < ny; ++y) for (int x = 0; x < nwg_x; ++x) {
        unsigned long long *s = &st[(base + (size_t)y * (base ? 8192 : nwg_x) + x) * 16];
        if (s[segs[k].first] && s[segs[k].second]) d.push_back((double)(s[segs[k].second] - s[segs[k].first]));
      }
      if (d.empty()) continue; std::sort(d.begin(), d.end());
      printf("   %-34s median %8.0f  p10 %8.0f  p90 %8.0f cycles  (%.2f us @2.4GHz)\n", labels[k], d[d.size()/2], d[d.size()/10], d[d.size()*9/10], d[d.size()/2]/2400.0);
    }
  };
#if OCEAN_ROW_PAIRED
  {
    printf("rowpair, %d workgroups x 3 pairs (s_memtime ticks; median over workgroups)\n", PAIRBLOCKS);
    char const *lab[6] = {"-> phase advanced + stored (inputs waited for)", "sim (bpermute, sincos)", "-> next inputs requested, fields derived", "3-field transform (6 barriers)", "spectrum stores issued", "-> next iteration top"};
    for (int it = 0; it < 3; ++it)
      for (int k = 0; k < 6; ++k) {
        std::vector<double> d;
        for (int b = 0; b < PAIRBLOCKS; ++b) {
          unsigned long long *s = &st[(size_t)b * 32];
          int i0 = 1 + 6*it + k, i1 = i0 + 1;
          if (k == 5 && it == 2) continue;
          if (s[i0] && s[i1]) d.push_back((double)(s[i1] - s[i0]));
        }
        if (d.empty()) continue; std::sort(d.begin(), d.end());
        printf("   it %d %-46s median %8.0f  p10 %8.0f  p90 %8.0f\n", it, lab[k], d[d.size()/2], d[d.size()/10], d[d.size()*9/10]);
      }
    std::vector<double> d, d0;
    for (int b = 0; b < PAIRBLOCKS; ++b) { unsigned long long *s = &st[(size_t)b * 32]; if (s[0] && s[18]) d.push_back((double)(s[18]-s[0])); if (s[0] && s[1]) d0.push_back((double)(s[1]-s[0])); }
    std::sort(d.begin(), d.end()); std::sort(d0.begin(), d0.end());
    if (!d.empty()) printf("   whole workgroup (3 pairs)  median %8.0f p10 %8.0f p90 %8.0f ; start -> first top %8.0f\n", d[d.size()/2], d[d.size()/10], d[d.size()*9/10], d0[d0.size()/2]);
    { std::vector<double> r; for (int b = 0; b < PAIRBLOCKS; ++b) { unsigned long long *s = &st[(size_t)b * 32]; if (s[31] > s[30] && s[29] > s[0]) r.push_back((double)(s[29]-s[0]) / (double)(s[31]-s[30])); }
      std::sort(r.begin(), r.end()); if (!r.empty()) printf("   s_memtime ticks per s_memrealtime tick (100 MHz): median %.2f  p10 %.2f p90 %.2f\n", r[r.size()/2], r[r.size()/10], r[r.size()*9/10]); }
    unsigned long long lo = ~0ull, hi = 0; for (int b = 0; b < PAIRBLOCKS; ++b) { unsigned long long *s = &st[(size_t)b * 32]; if (s[0]) lo = std::min(lo, s[0]); for (int k = 0; k < 19; ++k) hi = std::max(hi, s[k]); }
    printf("   first stamp -> last stamp over the launch: %llu ticks\n", hi - lo);
  }
#else
  report("rowpass", 0, RowCfg<N>::BLOCKS, C, {{0,1},{1,2},{2,3},{3,4},{4,5},{0,5}}, {"launch -> inputs arrived+advanced", "sim (sincos, kinv)", "prefetch issue + derive fields", "3-field transform (6 barriers)", "spectrum + halo stores issued", "whole workgroup"});
#endif
  report("colpass", (size_t)C*8192, N/ColCfg<N>::W, C, {{0,1},{1,2},{2,3},{3,4},{4,5},{5,6},{6,7},{7,8},{8,9},{9,10},{0,10}}, {"launch -> halo data arrived", "halo transform", "-> field 0 ready", "field 0 transform (pair)", "-> field 1 ready", "field 1 transform", "-> field 2 ready", "field 2 transform", "height exchange + barrier", "normals + stores issued", "whole workgroup"});
  return 0;
}
