// Diagnostic (never shipped): what a 16-byte-per-lane buffer gather costs the texture path by address pattern, with the
// table resident in L2 (128 KB, the 64^2 map's size).  One wave per SIMD x 8 resident waves, 64 fetches per lane per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template<int PATTERN>
__global__ void __launch_bounds__(256) gather(float4 const *table, int entries, float4 *out, int rounds)
{
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, entries * 16, 0x00020000);
  int const lane = threadIdx.x & 63;
  unsigned seed = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u;
  float4 acc = make_float4(0, 0, 0, 0);
  for(int k = 0; k < rounds; ++k)
  {
    int off[8];
    #pragma unroll
    for(int j = 0; j < 8; ++j)
    {
      unsigned h = hash(seed + k * 8 + j);          // wave-uniform random base
      unsigned hl = hash(h + lane * 2654435761u);   // per-lane random
      unsigned hq = hash(h + (lane >> 2) * 40503u); // per-quad random
      int e;
      if (PATTERN == 0) e = (h % (entries - 64)) + lane;                       // coalesced: lane k -> base + k
      else if (PATTERN == 1) e = h % entries;                                  // all lanes the same address
      else if (PATTERN == 2) e = hl % entries;                                 // every lane random
      else if (PATTERN == 3) e = hq % entries;                                 // the four lanes of a quad the same address, quads random
      else if (PATTERN == 4) e = ((hq % (entries / 4)) * 4) + (lane & 3);       // a quad reads one aligned 64-byte block in order
      else if (PATTERN == 5) e = ((hq % (entries / 4)) * 4) + ((lane * 3) & 3); // ... permuted inside the block
      else if (PATTERN == 6) e = ((lane & 3) == 0) ? (int)(hq % entries) : -16; // one lane of four fetches, three out of range
      else if (PATTERN == 7) e = ((hq % (entries / 8)) * 8) + (lane & 1) + 4 * ((lane >> 1) & 1);   // two 32-byte runs 64 bytes apart per quad
      else e = ((hash(h + (lane >> 4) * 7919u) % (entries / 16)) * 16) + (lane & 15); // 16 lanes read 256 contiguous bytes, four such runs
      off[j] = e * 16;
    }
    float4 v[8];
    #pragma unroll
    for(int j = 0; j < 8; ++j)
      v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off[j], 0, 0));
    #pragma unroll
    for(int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  if (acc.x == 123456.789f) out[threadIdx.x] = acc;
}

template<int P> int run(char const *name, float4 *table, int entries, float4 *out)
{
  int const rounds = 64, groups = 256 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int it = 0; it < 3; ++it) hipLaunchKernelGGL(gather<P>, dim3(groups), dim3(256), 0, 0, table, entries, out, rounds);
  CK(hipEventRecord(e0));
  for(int it = 0; it < 10; ++it) hipLaunchKernelGGL(gather<P>, dim3(groups), dim3(256), 0, 0, table, entries, out, rounds);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  double const instr_per_cu = (double)groups * 4 * rounds * 8 / 256;     // wave-level fetch instructions per CU
  printf("%-72s %8.1f us   %6.1f cycles per fetch instruction per CU (2.4 GHz)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / instr_per_cu);
  return 0;
}

int main()
{
  int const entries = 8192;          // 128 KB
  float4 *table, *out; CK(hipMalloc(&table, entries * 16)); CK(hipMalloc(&out, 4096)); CK(hipMemset(table, 0, entries * 16));
  run<0>("0 coalesced (lane k -> base + 16 k)", table, entries, out);
  run<1>("1 every lane the same address", table, entries, out);
  run<2>("2 every lane a random address", table, entries, out);
  run<3>("3 quads: four lanes one address, quads random", table, entries, out);
  run<4>("4 quads: one aligned 64-byte block, in order", table, entries, out);
  run<5>("5 quads: one aligned 64-byte block, permuted", table, entries, out);
  run<6>("6 one lane of four fetches (random), three out of range", table, entries, out);
  run<7>("7 quads: two 32-byte runs, 64 bytes apart", table, entries, out);
  run<8>("8 four runs of 16 lanes x 16 bytes (256 contiguous bytes each)", table, entries, out);
  return 0;
}
