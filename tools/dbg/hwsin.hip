// Diagnostic: accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) for phases in [0, 2 pi) -- the domain
// update_ocean keeps OceanParams::phase in -- against double precision, next to the row pass's own sincos_phase.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/dbg/bin/hwsin tools/dbg/hwsin.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../datum_amd/csrc/ocean_kernels.hip"

__global__ void k(float const *x, float *s_hw, float *c_hw, float *s_sw, float *c_sw, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float rev = x[i] * 0.15915494309189535f;
  s_hw[i] = __builtin_amdgcn_sinf(rev);
  c_hw[i] = __builtin_amdgcn_cosf(rev);
  ocean::sincos_phase(x[i], &s_sw[i], &c_sw[i]);
}

int main()
{
  size_t const n = (size_t)1 << 24;
  std::vector<float> x(n), r[4];
  for(size_t i = 0; i < n; ++i) x[i] = (float)((double)i * 6.283185307179586 / n);
  float *d[5];
  for(int j = 0; j < 5; ++j) hipMalloc(&d[j], n * 4);
  hipMemcpy(d[0], x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(d[0], d[1], d[2], d[3], d[4], n);
  for(int j = 0; j < 4; ++j) { r[j].resize(n); hipMemcpy(r[j].data(), d[j + 1], n * 4, hipMemcpyDeviceToHost); }
  double e[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for(size_t i = 0; i < n; ++i)
  {
    double s = sin((double)x[i]), c = cos((double)x[i]);
    double v[4] = { fabs(r[0][i] - s), fabs(r[1][i] - c), fabs(r[2][i] - s), fabs(r[3][i] - c) };
    for(int j = 0; j < 4; ++j) { if (v[j] > e[j]) e[j] = v[j]; q[j] += v[j] * v[j]; }
  }
  printf("2^24 phases in [0, 2 pi): max |error| (rms)  v_sin_f32 %.3e (%.3e)  v_cos_f32 %.3e (%.3e)  sincos_phase sin %.3e (%.3e) cos %.3e (%.3e)\n",
         e[0], sqrt(q[0] / n), e[1], sqrt(q[1] / n), e[2], sqrt(q[2] / n), e[3], sqrt(q[3] / n));
  return 0;
}
