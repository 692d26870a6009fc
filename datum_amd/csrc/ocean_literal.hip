// ocean_literal.hip -- the displacement step through the reference's OWN algorithm, on the GPU: a validation mode, not the fast path.
//
// The fused kernels of ocean_kernels.hip compute what the reference's five dispatches compute, but with two packed mixed-radix
// transforms and correctly rounded twiddles; against the reference's literal arithmetic -- radix-2 Stockham stages that read
// cos / sin of UNREDUCED fp32 angles from a table (src/renderer/ocean.cpp:686-700) -- they differ by that table's own error, which
// grows with N (1.4e-5 RMSE at 1024^2, 7e-5 at 4096^2; the fast path is the one closer to a float64 transform).  An
// integrator who wants to compare a HIP frame with a frame of the Vulkan build texel for texel, or to bisect a difference, switches
// a handle to this path (datum_ocean_set_literal_transform): one thread per point / texel, one workgroup per line, three separate
// complex fields in HBM, the same operations in the same order as the shaders:
//
//     literal_sim_kernel      data/ocean.sim.comp:44-79        h~, -i k^x h~, -i k^y h~ into three row-major planes
//     literal_fft_kernel      data/ocean.fftx.comp:49-100 / ocean.ffty.comp:49-100   conj, log2 N radix-2 Stockham stages with the
//                             literal table (second operand at + N/2, twiddle of the FULL lane index), conj; rows, then columns
//     literal_map_kernel      data/ocean.map.comp:51-82        sign, choppiness, central-difference normal -> the module's map layout
//
// The phase advance is the general fmodf kernel (ocean_advance_kernel: update_ocean, ocean.cpp:217-236).  It moves 196 B/pt like the
// reference and takes 4-17 x the fused step (tools/literal_bench.py: 44 us at 64^2, 211 us at 1024^2, 4.7 ms at 4096^2); nothing in bench.py runs it.

#pragma once

#include "ocean_kernels.hip"

namespace ocean
{
  // the reference's Spectrum buffer (ocean.cpp:61-68): h, hx, hy, one row-major plane of complex values each
  struct LiteralArgs
  {
    float2 const *h0;      // [N*N]
    float const *phase;    // [N*N]
    float2 *h, *hx, *hy;   // [N*N] each
    float const *weights;  // [N][2 log2 N]: cos, sin of -2 pi i / 2^(s+1) for lane i and stage s, as ocean.cpp:694-695 evaluates them
    char *maps;            // the cascade's map block (map_compact_a / map_compact_b)
    int N;
    float scale, choppiness;
  };

  __global__ void __launch_bounds__(256) literal_sim_kernel(LiteralArgs a)
  {
    int const N = a.N;
    size_t const plane = (size_t)N * N;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int const y = (int)(i / N), x = (int)(i % N);

      // k = 2 pi (xy - N/2) scale and its direction, zero at k = 0 (sim.comp:52-54)
      float const kx = (6.2831855f * ((float)x - 0.5f * (float)N)) * a.scale;
      float const ky = (6.2831855f * ((float)y - 0.5f * (float)N)) * a.scale;

      float knx = 0.0f, kny = 0.0f;

      if (kx != 0.0f || ky != 0.0f)
      {
        float const len = sqrtf(kx * kx + ky * ky);

        knx = kx / len;
        kny = ky / len;
      }

      float2 const hk = a.h0[i];
      float2 const hm = a.h0[(size_t)(N - 1 - y) * N + (N - 1 - x)];        // sim.comp:59

      float const ph = a.phase[i];
      float const c = cosf(ph), s = sinf(ph);

      // the expanded form of sim.comp:65-66
      float const re = (hk.x + hm.x) * c - (hk.y + hm.y) * s;
      float const im = (hk.x - hm.x) * s + (hk.y - hm.y) * c;

      a.h[i] = make_float2(re, im);
      a.hx[i] = make_float2(im * knx, -re * knx);
      a.hy[i] = make_float2(im * kny, -re * kny);
    }
  }

  // one workgroup per line (row y, or column x when COLUMNS), the three fields one after the other through an LDS ping-pong of
  // 2 x N complex values; thread t takes lanes t, t + 256, ... of every stage
  template<bool COLUMNS>
  __global__ void __launch_bounds__(256) literal_fft_kernel(LiteralArgs a)
  {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int const N = a.N;
    int const line = blockIdx.x;

    int stages = 0;
    while ((1 << stages) < N)
      ++stages;

    float2 *ping = reinterpret_cast<float2*>(smem);
    float2 *pong = ping + N;

    size_t const first = COLUMNS ? (size_t)line : (size_t)line * N;
    size_t const step = COLUMNS ? (size_t)N : 1;

    float2 *fields[3] = { a.h, a.hx, a.hy };

    for(int f = 0; f < 3; ++f)
    {
      float2 *data = fields[f];

      // load + conjugate (fftx.comp:55-62)
      for(int i = threadIdx.x; i < N; i += blockDim.x)
      {
        float2 const v = data[first + i * step];

        ping[i] = make_float2(v.x, -v.y);
      }

      __syncthreads();

      float2 *src = ping, *dst = pong;

      int s = 0;

      for(int n = 2; n <= N; n *= 2, ++s)
      {
        for(int i = threadIdx.x; i < N; i += blockDim.x)
        {
          int const i0 = (i / n) * (n / 2) + i % (n / 2);
          int const i1 = i0 + N / 2;

          float2 const u = src[i0], v = src[i1];

          float const t0 = a.weights[(size_t)i * 2 * stages + 2 * s + 0];
          float const t1 = a.weights[(size_t)i * 2 * stages + 2 * s + 1];

          // h0 + vec2(dot(vec2(t0, -t1), h1), dot(vec2(t1, t0), h1))   (fftx.comp:88)
          dst[i] = make_float2(u.x + (t0 * v.x + (-t1) * v.y), u.y + (t1 * v.x + t0 * v.y));
        }

        __syncthreads();

        float2 *const t = src;
        src = dst;
        dst = t;
      }

      // conjugate + store (fftx.comp:97-99)
      for(int i = threadIdx.x; i < N; i += blockDim.x)
      {
        float2 const v = src[i];

        data[first + i * step] = make_float2(v.x, -v.y);
      }

      __syncthreads();
    }
  }

  __global__ void __launch_bounds__(256) literal_map_kernel(LiteralArgs a)
  {
    int const N = a.N;
    size_t const plane = (size_t)N * N;

    float const nz = 4 / (a.scale * N);                                             // map.comp:77

    // Re h with the sign (-1)^(x+y) of map.comp:60, periodic (map.comp:58)
    auto dzat = [&](int x, int y)
    {
      int const xx = (x + N) % N, yy = (y + N) % N;

      return a.h[(size_t)yy * N + xx].x * (((x + y) & 1) ? -1.0f : 1.0f);
    };

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int const y = (int)(i / N), x = (int)(i % N);

      float const sigma = ((x + y) & 1) ? -1.0f : 1.0f;

      float const dz = a.h[i].x * sigma;
      float const dx = a.hx[i].x * sigma * a.choppiness;
      float const dy = a.hy[i].x * sigma * a.choppiness;

      float const nx = dzat(x - 1, y) - dzat(x + 1, y);
      float const ny = dzat(x, y + 1) - dzat(x, y - 1);
      float const len = sqrtf(nx * nx + ny * ny + nz * nz);

      *reinterpret_cast<float4*>(a.maps + map_compact_a(N, y, x)) = make_float4(dx, dy, dz, nx / len);
      *reinterpret_cast<float2*>(a.maps + map_compact_b(N, y, x)) = make_float2(ny / len, nz / len);
    }
  }

  // the five dispatches of ocean.cpp:769-789 for one cascade, in their order, on `stream`
  inline hipError_t launch_literal(LiteralArgs const &a, hipStream_t stream)
  {
    size_t const lds = (size_t)2 * a.N * sizeof(float2);
    int const blocks = (int)std::min<size_t>(((size_t)a.N * a.N + 255) / 256, 4096);

    hipLaunchKernelGGL(literal_sim_kernel, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(literal_fft_kernel<false>, dim3(a.N), dim3(256), lds, stream, a);
    hipLaunchKernelGGL(literal_fft_kernel<true>, dim3(a.N), dim3(256), lds, stream, a);
    hipLaunchKernelGGL(literal_map_kernel, dim3(blocks), dim3(256), 0, stream, a);

    return hipGetLastError();
  }
}
