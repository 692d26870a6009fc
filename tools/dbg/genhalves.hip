// Microbenchmark (round 5): ocean.gen of a 1024 x 1024 mesh as ONE launch against TWO half launches back to back on the
// same stream (the second half's workgroups would have to overlap the first half's stores for this to pay: consecutive dispatches of one
// stream do not overlap), and against the two halves on two streams.
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
  GenArgs g; g.set = set; g.map = maps; g.vertices = verts; gen_shape(g, N, sx, sy);
  int const groups = gen_groups(g); int const first = ((groups / 2 + 7) / 8) * 8;
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](char const *name, auto fn) { for (int i = 0; i < 20; ++i) fn(); CK(hipDeviceSynchronize()); hipEventRecord(e0, s0); for (int i = 0; i < 200; ++i) fn(); hipEventRecord(e1, s0); hipEventSynchronize(e1); CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-72s %7.2f us per mesh\n", name, ms / 200 * 1e3); return 0; };
  for (int rep = 0; rep < 3; ++rep) {
    timeit("one launch", [&]{ launch_gen_part(g, 0, groups, s0); });
    timeit("two half launches back to back on the same stream", [&]{ launch_gen_part(g, 0, first, s0); launch_gen_part(g, first, groups - first, s0); });
    timeit("four quarter launches back to back on the same stream", [&]{ int q = ((groups / 4 + 7) / 8) * 8; launch_gen_part(g, 0, q, s0); launch_gen_part(g, q, q, s0); launch_gen_part(g, 2 * q, q, s0); launch_gen_part(g, 3 * q, groups - 3 * q, s0); });
  }
  return 0;
}
