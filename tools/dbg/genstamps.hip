// Diagnostic build (never shipped): s_memrealtime stamps per phase of ocean_gen_kernel and where each workgroup ran,
// 1024 x 1024 mesh from STAMP_N^2 maps filled with smooth values.  Phases: ray + swell arithmetic | fetches issued ->
// arrived | shading | LDS staging + stores issued.
#define OCEAN_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#include <cmath>
#include <cstring>
#include "../../datum_amd/csrc/ocean_gen.hip"
using namespace ocean;
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)
#ifndef STAMP_N
#define STAMP_N 64
#endif
int main() {
  constexpr int N = STAMP_N; int const sx = 1024, sy = 1024; size_t P = (size_t)N * N;
  float4 *maps; float *verts; unsigned long long *stamps;
  CK(hipMalloc(&maps, 2 * P * 16)); CK(hipMalloc(&verts, (size_t)sx * sy * 48));
  int const tiles = ((sx + GEN_TILE_X - 1) / GEN_TILE_X) * ((sy + GEN_TILE_Y - 1) / GEN_TILE_Y);
  CK(hipMalloc(&stamps, (size_t)tiles * 16 * 8)); CK(hipMemset(stamps, 0, (size_t)tiles * 16 * 8));
  { std::vector<float4> h(2 * P); char *hb = reinterpret_cast<char*>(h.data());
    for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) { float a[4] = { 0.1f * sinf(0.3f * x), 0.1f * cosf(0.2f * y), 0.2f * sinf(0.1f * (x + y)), 0.05f }, b[2] = { 0.02f, 0.998f };
      memcpy(hb + map_compact_a(N, y, x), a, 16); memcpy(hb + map_compact_b(N, y, x), b, 8); }
    CK(hipMemcpy(maps, h.data(), h.size() * 16, hipMemcpyHostToDevice)); }
  // the example camera's OceanSet (examples/ocean/ocean.cpp:33,63): position (0,0,8) looking along +x, fov 60 deg, 16:9; identity-free closed form
  datum_ocean_set set = {};
  float halftan = tanf(30.0f * 3.14159265f / 180.0f), aspect = 1920.0f / 1080.0f, zn = 0.1f, zf = 24000.0f, depth = zf - zn;
  float proj[16] = {}; proj[0] = 1 / (aspect * halftan); proj[5] = -1 / halftan; proj[10] = zf / depth - 1; proj[11] = zf * zn / depth; proj[14] = -1;
  memcpy(set.proj, proj, 64);
  // inverse of that sparse matrix
  float inv[16] = {}; inv[0] = 1 / proj[0]; inv[5] = 1 / proj[5]; inv[11] = -1; inv[14] = 1 / proj[11]; inv[15] = proj[10] / proj[11];
  memcpy(set.invproj, inv, 64);
  // lookat from (0,0,8) towards +x with up +z: camera looks down -z in view space; rotation = (w,x,y,z) = (0.5, 0.5, -0.5, -0.5)
  float real[4] = { 0.5f, 0.5f, -0.5f, -0.5f };
  memcpy(set.camera_real, real, 16);
  // dual = 0.5 * (0, p) * real
  float px = 0, py = 0, pz = 8; float w = real[0], x = real[1], y = real[2], z = real[3];
  float dual[4] = { 0.5f * (-px * x - py * y - pz * z), 0.5f * (px * w + py * z - pz * y), 0.5f * (py * w + pz * x - px * z), 0.5f * (pz * w + px * y - py * x) };
  memcpy(set.camera_dual, dual, 16);
  set.plane[2] = 1; set.swelllength = 40; set.swellamplitude = 0.8f; set.swelldirection[0] = 0.780869f; set.swelldirection[1] = 0.624695f;
  set.scale = 1 / 22.0f; set.choppiness = 1.35f; set.smoothing = 1 / 320.0f; set.size = N;
  GenArgs g; g.set = set; g.map = maps; g.vertices = verts; g.stamps = stamps; gen_shape(g, N, sx, sy);
  if (g.tiles != tiles) { printf("tile count\n"); return 1; }
  for (int it = 0; it < 5; ++it) CK(launch_gen(g, 0));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st((size_t)tiles * 16); CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  char const *labels[4] = { "ray, plane hit, swell (arithmetic)", "fetches issued -> arrived", "shading", "staging + stores issued" };
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < tiles; ++b) { t0 = std::min(t0, st[b * 16]); t1 = std::max(t1, st[b * 16 + 4]); }
  printf("== ocean_gen, %d^2 maps: %d workgroups; first start -> last end %.2f us\n", N, tiles, (t1 - t0) * 0.01);
  for (int half = 0; half < 2; ++half) {
    printf("  %s half of the mesh (tiles %d..%d)\n", half ? "lower (water)" : "upper (sky)", half * tiles / 2, (half + 1) * tiles / 2 - 1);
    for (int k = 0; k < 4; ++k) { std::vector<double> d; for (int b = half * tiles / 2; b < (half + 1) * tiles / 2; ++b) d.push_back((st[b * 16 + k + 1] - st[b * 16 + k]) * 0.01); std::sort(d.begin(), d.end());
      printf("   %-40s median %6.2f  p10 %6.2f  p90 %6.2f\n", labels[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]); }
    std::vector<double> d; for (int b = half * tiles / 2; b < (half + 1) * tiles / 2; ++b) d.push_back((st[b * 16 + 4] - st[b * 16]) * 0.01); std::sort(d.begin(), d.end());
    printf("   %-40s median %6.2f  p10 %6.2f  p90 %6.2f\n", "whole workgroup", d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
  }
  { std::vector<double> d; for (int b = 0; b < tiles; ++b) d.push_back((st[b * 16] - t0) * 0.01); std::sort(d.begin(), d.end());
    printf("   start times: p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f\n", d[d.size()/10], d[d.size()/4], d[d.size()/2], d[d.size()*3/4], d[d.size()*9/10], d.back()); }
  int nb = (int)((t1 - t0) / 100) + 1; std::vector<std::vector<double>> occ(4, std::vector<double>(nb, 0.0));
  for (int b = 0; b < tiles; ++b) for (int k = 0; k < 4; ++k) for (unsigned long long t = st[b * 16 + k]; t < st[b * 16 + k + 1]; ++t) occ[k][(t - t0) / 100] += 0.01;
  printf("   workgroups in each phase, per 1-us bin (chip-wide):\n");
  for (int k = 0; k < 4; ++k) { printf("   %-28.28s", labels[k]); for (int i = 0; i < nb && i < 44; ++i) printf("%5.0f", occ[k][i]); printf("\n"); }
  std::map<unsigned, std::vector<int>> bycu;
  for (int b = 0; b < tiles; ++b) { unsigned hw = (unsigned)st[b * 16 + 15], xcc = (unsigned)st[b * 16 + 14] & 15; bycu[(xcc << 16) | (hw & 0xff00 & ~0u)].push_back(b); }
  printf("   distinct (XCC, SE, CU) seen: %zu\n", bycu.size());
  printf("   where workgroups ran (b: xcc/se.sh.cu start-us):");
  for (int b = 0; b < 48; ++b) { unsigned hw = (unsigned)st[b * 16 + 15], xcc = (unsigned)st[b * 16 + 14] & 15; if (b % 8 == 0) printf("\n    "); printf(" %d:%u/%u.%u.%u@%.2f", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (st[b * 16] - t0) * 0.01); }
  printf("\n   workgroups sharing the CU of workgroup 0:");
  { unsigned key0 = (((unsigned)st[14] & 15) << 16) | ((unsigned)st[15] & 0xff00); for (int b : bycu[key0]) printf(" %d@%.2f", b, (st[(size_t)b * 16] - t0) * 0.01); }
  printf("\n");
  return 0;
}
