// ocean_kernels.hip -- gfx950 kernels of the ocean displacement step (mesh generation: ocean_gen.hip).
//
// What the reference does in five Vulkan dispatches plus a host loop (src/renderer/ocean.cpp:217-236,
// :769-793) is done here in two fused kernels (+ gen):
//
//   row pass    update_ocean phase advance (ocean.cpp:223-233) + ocean.sim (data/ocean.sim.comp:44-79)
//               + ocean.fftx (data/ocean.fftx.comp:49-100) of TWO packed fields (see "packed step" below),
//               N/8 (N/16 from 2048^2 up) threads per row, rows y and N-y per workgroup
//               reads  h0 (8 B/pt, plus its mirror row through L2), phase (4)   writes phase (4), spectrum (16)
//   column pass ocean.ffty (data/ocean.ffty.comp:49-100) + ocean.map (data/ocean.map.comp:51-82),
//               one workgroup per tile of W columns, one column per thread group
//               reads spectrum (16)    writes the two map layers without their zero .w channels (24: map_compact_a)
//
// = 72 bytes of HBM traffic per grid point; the algorithm as the reference states it (three transforms) has
// 96 algorithmic bytes per point (bench.py's roofline.frac_on_survey_bytes) and the reference as
// written moves 196.
//
// Work spectrum layout (private to these kernels): per cascade, 8 x 8 blocks of 16-byte values (C, D),
// [y/8][x/8][y%8][x%8]: a row of a block is one 128-byte line; a column-pass wave reads whole blocks.  The largest grids
// keep the columns one XCD works on at a time contiguous ([x/B][...]: blocked_at, band_cols), for the maps too (map_compact_patch).
// At 4096^2 the column pass's workgroups are persistent and walk their tiles; everything else is one work item per workgroup.
//
// Built with -ffp-contract=off: products and sums are rounded as written (the phase state is
// bit-identical to the host formula); the FFT butterflies ask for their FMAs explicitly.

#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ocean_fft_core.h"
#include "../../include/datum_ocean_hip.h"

namespace ocean
{
  constexpr int MAX_PENDING = 8;

  typedef float4 cd;    // (C.re, C.im, D.re, D.im) of one grid point: the two packed fields of the work spectrum

  // the same four numbers as IEEE halves, 8 bytes (BASELINE.json configs[4]: spectrum stored fp16, arithmetic fp32)
  typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
  struct ch { half4_ v; };

  template<bool H16> struct SpecValue { typedef cd type; };
  template<> struct SpecValue<true> { typedef ch type; };

  struct CascadeConst
  {
    float wavescale;     // OceanParams::wavescale                 (update_ocean, ocean.cpp:225-229)
    float scale;         // OceanSet::scale = 1 / wavescale        (ocean.cpp:743)
    float choppiness;    // OceanSet::choppiness                   (ocean.cpp:744)
    float nz;            // 4 / (scale * N)                        (ocean.map.comp:77)
    float specscale;     // fp16 work spectrum only: power of two the row pass multiplies (C, D) by before rounding to half
    float specinv;       // ... and its reciprocal, folded into the column pass's sign factor (1 for the fp32 spectrum)
    float rowscale;      // h0 as halves only (DATUM_OCEAN_SPECTRUM_FP16_H0): specscale over the power of two the stored h0 carries (size_spectrum_scale)
  };

  struct StepArgs
  {
    float2 const *h0;    // [cascade][N*N]       OceanSet::h0
    unsigned int const *h0h;   // [cascade][N*N]   the same as two IEEE halves times a power of two per cascade (DATUM_OCEAN_SPECTRUM_FP16_H0 only, else null)
    float *phase;        // [cascade][N*N]       OceanSet::phase
    void *spec;          // [cascade - first][N*N] work spectrum of THIS launch (replaces Spectrum::h, hx, hy), blocked layout: cd, or ch (fp16 variant)
    float4 *maps;        // [cascade][2*N*N]     displacementmap, 2 layers, 24-byte texels in patches (map_compact_a / map_compact_b)
    cf const *tw;        // [N]                  exp(+2 pi i k / N)
    float const *omega;  // [cascade][(N/2+1)^2] dispersion(k) by (|m - N/2|, |n - N/2|)
    int ndt;
    int cascades;        // cascades THIS launch works on ...
    int first;           // ... starting with this one (the two passes are launched per group of cascades: ocean_capi, cascade_group)
    float dt[MAX_PENDING];
    CascadeConst casc[DATUM_OCEAN_MAX_CASCADES];
#ifdef OCEAN_STAMPS
    unsigned long long *stamps;   // diagnostic builds only (tools/dbg/stamps.hip): [kernel][workgroup][16] timestamps
#endif
  };

#ifdef OCEAN_STAMPS
  // diagnostic builds only: s_memrealtime (100 MHz, one clock for the whole device) per phase, plus where the workgroup ran
  #define OCEAN_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) stampbase[slot] = t_; } while(0)
  #define OCEAN_STAMP_WHERE() do { unsigned int hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); if (threadIdx.x == 0) { stampbase[14] = xcc_; stampbase[15] = hw_; } } while(0)
  #define OCEAN_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
  #define OCEAN_STAMP(slot) do { } while(0)
  #define OCEAN_STAMP_WHERE() do { } while(0)
  #define OCEAN_WAIT_LOADS() do { } while(0)
#endif

  constexpr int SBR = 8, SBC = 8;

  // columns per block, by the stored value's size.  The 8-byte values of the fp16-stored spectrum keep 8 columns (64-byte block rows): with 16 -- whole
  // lines per row-pass store -- the row pass gains 6-14 us at 4096^2 and 2048^2 x 4 and the column pass, whose narrow tiles then take 16 bytes of
  // every line they touch, loses as much or more (profiles/r06_spectrum_blocks.txt: 4096^2 with h0 as halves 5.2 -> 5.1 k grids/s)
  __host__ __device__ __forceinline__ constexpr int spec_block_cols(bool half) { return half ? 8 : SBC; }

  // Bands (large grids): the columns one XCD's column-pass workgroups work on at the same time are made contiguous in
  // memory -- [x / B][rows][x % B] -- for the work spectrum and for the maps alike, so that what is read and written
  // concurrently is a dense region instead of 2 KB pieces of rows 128 KB apart (4096^2).  B = band_cols(N), 0 = whole rows.
  // measured (profiles/r02_large_grids.txt): 4096^2 B = 64 (32 CUs x 2-column tiles): column pass 240 -> 226 us, with the
  // fp16-stored spectrum 202 -> 164 us; 2048^2 x 4 B = 128 (32 CUs x 4-column tiles): 181 -> 167 us; B = 512 at 4096^2: 270 us.
  // With the maps in 2 x 2 patches at 4096^2 (round 3's layout): B = 64 184-195 us, B = 128 176-183 us, B = 256 192 us, B = 512 / none 220 us
  __host__ __device__ __forceinline__ constexpr int band_cols(int N) { return (N >= 2048) ? 128 : N; }

  // element index of grid point (y, x) in the blocked work spectrum: per band, blocks of SBR rows x SBC columns, row-major inside
  __host__ __device__ __forceinline__ constexpr size_t blocked_at(int N, int y, int x, bool half = false)
  {
    int const B = band_cols(N);
    int const BC = spec_block_cols(half);

    return (size_t)(x / B) * N * B + ((size_t)(y / SBR) * (B / BC) + (x % B) / BC) * (SBR * BC) + (y % SBR) * BC + (x % BC);
  }

  template<int N, bool H16 = false>
  __host__ __device__ __forceinline__ constexpr size_t blocked(int y, int x)
  {
    return blocked_at(N, y, x, H16);
  }

  // Displacement map layout (private to this module: in the reference the map is a VK_IMAGE_TILING_OPTIMAL 2-layer image whose
  // only reader is ocean.gen's sampler, ocean.cpp:706, gen.comp:113-114; datum_ocean_read_maps / datum_ocean_export_maps hand out
  // the logical [layer][y][x] RGBA32F image).
  // 24 bytes per texel instead of 32 -- the two RGBA32F layers' .w channels are
  // constant zero (map.comp:79-80) and nothing reads them (gen.comp:113-114 takes .xyz), yet they were a quarter of what the
  // write-bound column pass stores.  Per cascade, bands as above; inside a band PATCHES of PW x PH = 16 texels, patch rows
  // one after the other; a patch is 384 bytes = three 128-byte lines:
  //     part A, 256 bytes: texel j = (y % PH) * PW + x % PW  ->  float4 (dx, dy, dz, nx)   at 16 j
  //     part B, 128 bytes: texel j                            ->  float2 (ny, nz)           at 256 + 8 j
  // PW = the column pass's tile width at that resolution (8 up to 256^2, 2 at 512^2, 4 at 1024^2 and 2048^2, 2 at 4096^2), so that the 16 texels of
  // a patch are 16 neighbouring lanes of a column-pass wave: one 16-byte and one 8-byte store instruction per thread and slot
  // write two whole lines and one whole line per patch -- no lane trades, no partial lines.  For ocean.gen a 4 x 4 patch holds
  // the four corners of a bilinear fetch more often than a 4 x 1 group did.
  constexpr int MAP_PATCH = 16;                   // texels per patch
  constexpr int MAP_PATCH_BYTES = 384;

  __host__ __device__ __forceinline__ constexpr int map_patch_cols(int N) { return N <= 256 ? 8 : (N == 512 ? 2 : (N <= 2048 ? 4 : 2)); }     // == ColCfg<N>::W (asserted there)
  __host__ __device__ __forceinline__ constexpr int map_patch_rows(int N) { return MAP_PATCH / map_patch_cols(N); }

  // bytes of one cascade's maps
  __host__ __device__ __forceinline__ constexpr size_t map_cascade_bytes(int N) { return (size_t)N * N * 24; }

  // byte offset of texel (x, y)'s part A inside its cascade's block; part B is map_compact_b(...)
  __host__ __device__ __forceinline__ constexpr size_t map_compact_patch(int N, int y, int x)
  {
    int const B = band_cols(N);
    int const PW = map_patch_cols(N), PH = map_patch_rows(N);

    return (size_t)(x / B) * 24 * N * B + ((size_t)(y / PH) * (B / PW) + (x % B) / PW) * MAP_PATCH_BYTES;
  }

  __host__ __device__ __forceinline__ constexpr int map_compact_j(int N, int y, int x) { return (y % map_patch_rows(N)) * map_patch_cols(N) + x % map_patch_cols(N); }

  __host__ __device__ __forceinline__ constexpr size_t map_compact_a(int N, int y, int x) { return map_compact_patch(N, y, x) + 16 * map_compact_j(N, y, x); }
  __host__ __device__ __forceinline__ constexpr size_t map_compact_b(int N, int y, int x) { return map_compact_patch(N, y, x) + 256 + 8 * map_compact_j(N, y, x); }

  // bytes from a texel's patch to the patch of the same column k * PH rows on
  __host__ __device__ __forceinline__ constexpr int map_compact_patchrow_bytes(int N) { return (band_cols(N) / map_patch_cols(N)) * MAP_PATCH_BYTES; }

  //|---------------------- buffer addressing ----------------------------------
  // Global accesses whose addresses differ between a thread's slots only by a wave-uniform amount go through
  // buffer instructions: one 32-bit VGPR offset per thread plus an SGPR offset per access, instead of a 64-bit
  // VGPR address computed per access (which was ~18 % of the VALU instructions of these kernels).

  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

  __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void const *base, size_t bytes)
  {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
  }

  template<int AUX = 0>
  __device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, AUX));
  }

  __device__ __forceinline__ unsigned int buf_load_u32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 0);
  }

  template<int AUX = 0>
  __device__ __forceinline__ float2 buf_load_f32x2(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voffset, soffset, AUX));
  }

  __device__ __forceinline__ cf buf_load_cf(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    return __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(r, voffset, soffset, 0));
  }

  // with a cache-policy operand (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
  template<int AUX>
  __device__ __forceinline__ float4 buf_load_f32x4_aux(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, AUX));
  }

  // Stores: the per-access part goes into the VGPR offset (one v_add_u32), not into an SGPR.  A 128-bit buffer
  // store with an SGPR offset was observed on gfx950 / ROCm 7.2 to store stale x, y components in the last lanes of
  // each 16-lane group when the next VALU instruction rewrote its data registers: hipcc enforces the "VALU write of
  // VMEM store data" wait states only when soffset is not a register (tools/dbg/colerr.py found it).
  template<int AUX>
  __device__ __forceinline__ void buf_store_f32_aux(float v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voffset + soffset, 0, AUX);
  }

  template<int AUX>
  __device__ __forceinline__ void buf_store_cf_aux(cf v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, v), r, voffset + soffset, 0, AUX);
  }

  template<int AUX>
  __device__ __forceinline__ void buf_store_f32x4_aux(float4 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
  {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, voffset + soffset, 0, AUX);
  }

  // cache policies (the aux operand of the buffer instructions: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1 = written through), measured:
  //   map stores written through leave the XCD's L2 to the spectrum lines that neighbouring tiles share: column pass 36.7 -> 35.1 us at
  //   1024^2 x 4 (nt on the same stores 44.5 us, nt on the spectrum loads 44.4 us); the row pass with both of its store streams written
  //   through: 29.4 -> 28.4 us (either one alone: no change); loads: nothing gains, nt on h0 loses 6 us (the mirror read no longer
  //   finds the row in L2): profiles/r04_store_policy_sweep.txt, r04_load_policy_sweep.txt.
  //   Round 6 (profiles/r06_store_policies.txt): where the handle's working set does NOT fit the 256 MiB Infinity Cache the maps are
  //   STREAMED (nt): they are the one stream nobody reads again before it has left the cache anyway, and stored any other way -- plain
  //   through round 5, written through -- their 24 bytes per point push the next ROW pass's inputs and the exchange spectrum out of it:
  //   4096^2 fp16 row pass 118-120 -> 100-104 us and column pass 133 -> 115-122 us (the row pass gains although only the column pass's
  //   instruction changed), 2048^2 x 4 15.9 -> 17.1 k grids/s, 1024^2 x 8 68.8 -> 77.9-78.4 k (its row pass 63 -> 48 us: the 164 MB of h0,
  //   phase and one group's spectrum then STAY in the cache), x 6 68 -> 75 k, x 16 +1.5 %.  nt on the row pass's own stores loses at every
  //   size (4096^2: 120 -> 128 us; 1024^2 x 16: 68.5 -> 60.5 k), sc1 alone and plain equal each other, written through + nt equals nt.
  constexpr int MAP_STORE_AUX = 17, MAP_STORE_AUX_STREAM = 2, PHASE_STORE_AUX = 17, SPEC_STORE_AUX = 17;

  //|---------------------- update_ocean --------------------------------------

  // dispersion(k) of ocean.cpp:82-87 at grid point (n, m), k as update_ocean forms it (ocean.cpp:225,229)
  __device__ __forceinline__ float dispersion_at(int n, int m, int N, float wavescale)
  {
    float x = (6.2831855f * ((float)n - 0.5f * (float)N)) / wavescale;
    float y = (6.2831855f * ((float)m - 0.5f * (float)N)) / wavescale;

    float k2 = x * x + y * y;

    return sqrtf((9.81f * sqrtf(k2)) * (1.0f + k2 / 136900.0f));
  }

  // dispersion(k) depends on |kx|, |ky| only and x -> -x is exact in fp32, so one quadrant of it, built once per
  // wavescale with the formula above, serves every step: omega[i * (N/2+1) + j] = dispersion at |m - N/2| = i,
  // |n - N/2| = j.  The row pass then pays one L2-resident load instead of two IEEE divides and two IEEE
  // square roots per point, and the phase it produces is still bit-identical to update_ocean's.
  struct WaveScales { float v[DATUM_OCEAN_MAX_CASCADES]; };

  // (the wave scales travel by value in the kernel arguments and only the cascades in `dirty` are rebuilt: a blend of
  // the wave parameters changes one cascade's scale every frame, lerp_ocean_waves, ocean.cpp:185-192)
  __global__ void ocean_omega_kernel(float *omega, int N, int cascades, WaveScales wavescales, unsigned int dirty)
  {
    int const Q = N / 2 + 1;

    for(int cascade = 0; cascade < cascades; ++cascade)
    {
      if (!((dirty >> cascade) & 1))
        continue;

      float const wavescale = wavescales.v[cascade];
      float *table = omega + (size_t)cascade * Q * Q;

      for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < (size_t)Q * Q; k += (size_t)gridDim.x * blockDim.x)
      {
        int i = (int)(k / Q), j = (int)(k % Q);

        table[k] = dispersion_at(N / 2 + j, N / 2 + i, N, wavescale);
      }
    }
  }

  // does an uploaded phase array leave [0, 2 pi)?  (the fused row pass's one-subtraction fmod is exact only inside)
  __global__ void ocean_phaserange_kernel(float const *phase, size_t count, unsigned int *wild)
  {
    bool bad = false;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    {
      float const p = phase[i];

      bad |= !(p >= 0.0f && p < 6.2831855f);
    }

    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0)
      atomicOr(wild, 1u);
  }

  __device__ __forceinline__ float dispersion_lookup(float const *table, int n, int m, int N)
  {
    int j = n - N / 2, i = m - N / 2;

    j = j < 0 ? -j : j;
    i = i < 0 ? -i : i;

    return table[i * (N / 2 + 1) + j];
  }

  // fmod(phase + w*dt, 2 pi) of ocean.cpp:231, any operands (phase-only kernel).  fmod is exact.
  __device__ __forceinline__ float advance_phase(float phase, float wdt)
  {
    return fmodf(phase + wdt, 6.2831855f);
  }

  // (the fused row pass is given 0 <= phase < 2 pi and 0 <= w*dt < 2 pi -- the host checks both and otherwise runs the phase-only
  // kernel first: then 0 <= a < 4 pi and fmod(a, 2 pi) is a or a - 2 pi, the subtraction being exact (Sterbenz), bit-identical to fmod's:
  // the row pass's packed select)

  //|---------------------- ocean.sim -----------------------------------------

  // sin and cos of the phase (sim.comp:61-62).  update_ocean keeps the phase in [0, 2 pi), so the argument
  // reduction is a two-constant Cody-Waite step to [-pi/4, pi/4] followed by the Cephes single-precision
  // minimax polynomials: about 1 ulp there (measured against float64 in tests), at a quarter of the
  // instructions and registers of the all-range libm path.  Arguments far outside (|x| >> 1e4) lose accuracy
  // gradually, as GLSL's own sin/cos do.
  __device__ __forceinline__ void sincos_phase(float x, float *sin_out, float *cos_out)
  {
    float k = rintf(x * 0.636619772367581343f);                 // x * 2/pi

    float r = fmaf(k, -1.57079637050628662109375f, x);          // pi/2 head
    r = fmaf(k, 4.37113900018624283e-8f, r);                    // pi/2 tail

    float z = r * r;

    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(z, -0.5f, 1.0f));

    int q = (int)k;

    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;

    *sin_out = (q & 2) ? -s : s;
    *cos_out = ((q + 1) & 2) ? -c : c;
  }

  // The row pass's sin / cos of ONE phase: the hardware's v_sin_f32 / v_cos_f32 of phase / 2 pi (what a
  // Vulkan driver makes of the shader's sin() and cos(), sim.comp:61-62, on this GPU: |error| <= 4.8e-7 for phases in [0, 2 pi),
  // tools/dbg/hwsin.hip) or sincos_phase above
  template<bool WILD> __device__ __forceinline__ void sincos_row(float x, float *sin_out, float *cos_out);

  // one point of data/ocean.sim.comp:52-66: h~ from h0(k), h0 at the mirror index and the phase, evaluated as
  // h0(k) e^{i phase} + conj(h0(mirror) e^{i phase})  (two complex products and a conjugating add: 5 packed
  // instructions; equal to the shader's expanded form, sim.comp:65-66, up to rounding)
  template<bool WILD>
  __device__ __forceinline__ cf sim_height_products(float2 h0k, float2 h0mk, float phase)
  {
    float sin_v, cos_v;
    sincos_row<WILD>(phase, &sin_v, &cos_v);        // (the row pass's own sin / cos: the stage test pins the product's arithmetic)

    cf const e = cf{ cos_v, sin_v };
    cf const u = cmul(cf{ h0k.x, h0k.y }, e);
    cf const m = cmul(cf{ h0mk.x, h0mk.y }, e);

    return add_conj(u, m);
  }

  // k of sim.comp:52 and its normalisation (sim.comp:54)
  __device__ __forceinline__ float wavevector(int i, int N, float scale)
  {
    return (6.2831855f * ((float)i - 0.5f * (float)N)) * scale;
  }

  // 1 / |k|, 0 at k = 0: normalize(k) = k * inversesqrt(dot(k, k)) with the k = 0 guard of sim.comp:54
  __device__ __forceinline__ float kinv_of(float kx, float ky)
  {
    float k2 = kx * kx + ky * ky;

    return (k2 != 0.0f) ? rsqrtf(k2) : 0.0f;
  }

  __device__ __forceinline__ float2 knorm_of(float kx, float ky)
  {
    float inv = kinv_of(kx, ky);

    return make_float2(kx * inv, ky * inv);
  }

  //|---------------------- two slots per instruction (round 4) ----------------
  // The row pass's prologue -- update_ocean, sin / cos of the phase, k and 1 / |k| -- works on one float per grid point, and
  // gfx950's packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) do two per lane per issue: a thread's
  // slots are taken in pairs (s, s + 1).  With -ffp-contract=off every product and sum is rounded as the scalar form rounds
  // it (the phase state stays bit-identical to update_ocean's).  A real factor that belongs to ONE slot of a pair enters
  // the complex operations through the half-select modifiers (broadcast of one half), not through a splat.
  typedef float f2_ __attribute__((ext_vector_type(2)));

  __device__ __forceinline__ f2_ pfma2(f2_ a, f2_ b, f2_ c) { return __builtin_elementwise_fma(a, b, c); }

  // sincos_phase of two arguments: the same reduction and polynomials, packed
  __device__ __forceinline__ void sincos_phase_pair_poly(f2_ x, f2_ &sn, f2_ &cs)
  {
    f2_ const t = x * 0.636619772367581343f;                            // x * 2/pi
    f2_ const k = { __builtin_rintf(t.x), __builtin_rintf(t.y) };

    f2_ r = pfma2(k, f2_{ -1.57079637050628662109375f, -1.57079637050628662109375f }, x);       // pi/2 head
    r = pfma2(k, f2_{ 4.37113900018624283e-8f, 4.37113900018624283e-8f }, r);                    // pi/2 tail

    f2_ const z = r * r;

    f2_ const sp = pfma2(pfma2(pfma2(f2_{ -1.9515295891e-4f, -1.9515295891e-4f }, z, f2_{ 8.3321608736e-3f, 8.3321608736e-3f }), z, f2_{ -1.6666654611e-1f, -1.6666654611e-1f }), z * r, r);
    f2_ const cp = pfma2(pfma2(pfma2(f2_{ 2.443315711809948e-5f, 2.443315711809948e-5f }, z, f2_{ -1.388731625493765e-3f, -1.388731625493765e-3f }), z, f2_{ 4.166664568298827e-2f, 4.166664568298827e-2f }), z * z, pfma2(z, f2_{ -0.5f, -0.5f }, f2_{ 1.0f, 1.0f }));

    #pragma unroll
    for(int i = 0; i < 2; ++i)
    {
      int const q = (int)k[i];

      float const s_ = (q & 1) ? cp[i] : sp[i];
      float const c_ = (q & 1) ? sp[i] : cp[i];

      sn[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s_) ^ (((unsigned)q << 30) & 0x80000000u));
      cs[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c_) ^ (((unsigned)(q + 1) << 30) & 0x80000000u));
    }
  }

  // sin / cos of the phases of two slots.  WILD (the row pass's template flag, chosen per launch by the host): a phase may lie
  // outside [0, 2 pi) -- uploaded so, or left there by a negative dt -- and v_sin_f32 / v_cos_f32 end at 256 turns: the reduction +
  // polynomials, which take any argument
  template<bool WILD>
  __device__ __forceinline__ void sincos_phase_pair(f2_ x, f2_ &sn, f2_ &cs)
  {
    if constexpr (WILD)
    {
      sincos_phase_pair_poly(x, sn, cs);
      return;
    }

    f2_ const rev = x * 0.15915494309189535f;

    sn = f2_{ __builtin_amdgcn_sinf(rev.x), __builtin_amdgcn_sinf(rev.y) };
    cs = f2_{ __builtin_amdgcn_cosf(rev.x), __builtin_amdgcn_cosf(rev.y) };

    // gfx940+: a VALU instruction that reads the result of a transcendental one needs two wait states in between.  hipcc
    // inserts them for the instructions it emits, but the consumers here are the inline-asm packed complex products, which
    // its hazard recognizer does not look into (measured: RMSE 1e-3..6e-2 at the resolutions where the scheduler happened to
    // put a product right behind a v_cos_f32).  The results pass through this statement, so every reader comes after it.
    asm volatile("s_nop 1" : "+v"(sn), "+v"(cs));
  }

  template<bool WILD>
  __device__ __forceinline__ void sincos_row(float x, float *sin_out, float *cos_out)
  {
    if constexpr (WILD)
    {
      sincos_phase(x, sin_out, cos_out);
      return;
    }

    float const rev = x * 0.15915494309189535f;

    float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);

    asm volatile("s_nop 1" : "+v"(sn), "+v"(cs));       // (see sincos_phase_pair)

    *sin_out = sn;
    *cos_out = cs;
  }

#if defined(__HIP_DEVICE_COMPILE__)
  // a + c[H] * b
  template<int H> __device__ __forceinline__ cf fma_real_h(cf a, cf b, f2_ c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), r;

    if (H == 0)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(vb), "v"(c), "v"(va));
    else
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(vb), "v"(c), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  // a + c[H] * (-i b) = (a.x + c b.y, a.y - c b.x)
  template<int H> __device__ __forceinline__ cf fma_negi_h(cf a, cf b, f2_ c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), vb = __builtin_bit_cast(v2f_, b), r;

    if (H == 0)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(vb), "v"(c), "v"(va));
    else
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(vb), "v"(c), "v"(va));

    return __builtin_bit_cast(cf, r);
  }

  // c[H] * a
  template<int H> __device__ __forceinline__ cf scale_real_h(cf a, f2_ c)
  {
    v2f_ va = __builtin_bit_cast(v2f_, a), r;

    if (H == 0)
      asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(va), "v"(c));
    else
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(va), "v"(c));

    return __builtin_bit_cast(cf, r);
  }
#else
  template<int H> __host__ __device__ __forceinline__ cf fma_real_h(cf a, cf b, f2_ c) { return fma_real(a, b, c[H]); }
  template<int H> __host__ __device__ __forceinline__ cf fma_negi_h(cf a, cf b, f2_ c) { return fma_negi(a, b, c[H]); }
  template<int H> __host__ __device__ __forceinline__ cf scale_real_h(cf a, f2_ c) { return scale_real(a, c[H]); }
#endif

  //|---------------------- per-thread twiddles of a line transform ------------

  template<int N, int E_> struct LineTw
  {
    typedef typename LineFFT<N, 1, E_>::Twiddles type;
    static __device__ __forceinline__ void load(cf const *tw, int t, type &w) { LineFFT<N, 1, E_>::load_twiddles(tw, t, w); }
  };

  //|---------------------- line FFTs with workgroup barriers ------------------
  // K independent lines per thread go through the exchange phases together, so the number of barriers per
  // workgroup does not grow with K (one line per barrier phase is what the reference's per-field loop does).

  // the step kernels keep their LDS accesses unpaired (ocean_fft_core.h, "LDS layout of the exchanges")
#if defined(__HIP_DEVICE_COMPILE__)
#define OCEAN_LDS_UNPAIRED __attribute__((target("no-load-store-opt")))
#else
#define OCEAN_LDS_UNPAIRED
#endif

  struct NoHook { __device__ __forceinline__ void operator()() const { } };

  // `before_last` runs between the last exchange and the last pass: every value of the lines is in LDS then and the
  // threads' value registers are free (the walking column pass requests its next tile there)
  // W: lines interleaved element by element (`line` points at this thread's line, element positions W apart): LineFFT
  // between the two sides of an exchange: a workgroup barrier, or -- where the line belongs to ONE wave -- a wave-level fence
  template<bool WAVE>
  __device__ __forceinline__ void exchange_sync()
  {
    if constexpr (WAVE)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    else
      __syncthreads();
  }

  template<int N, int K, int W, int E_, typename Hook = NoHook>
  __device__ __forceinline__ void fft_lines(cf (&v)[K][E_], int t, cf *line, int linestride, cf const *midtab, typename LineTw<N, E_>::type const &w, bool active, Hook before_last = Hook())
  {
    typedef LineFFT<N, W, E_> L;

    // a line whose T threads are exactly one wave (row pass, W = 1, T = 64: 512^2) is exchanged without workgroup barriers: a wave's LDS
    // operations execute in order, a fence that completes its stores is all an exchange needs (512^2 x 1 11.6-11.7 -> 11.5 us, x 4 row pass
    // 11.4 -> 11.0 us; at 1024^2 with 16 points per thread -- one wave per row too -- it changes nothing: profiles/r05_rowpass_forms.txt)
    constexpr bool WAVE = (W == 1) && (Plan<N, E_>::T == 64);

    if (active)
    {
      #pragma unroll
      for(int k = 0; k < K; ++k)
        L::pass0(v[k], t, line + k * linestride);
    }

    exchange_sync<WAVE>();

    if (Plan<N, E_>::NP >= 3)
    {
      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_load<1>(v[k], t, line + k * linestride, midtab, w);
      }

      exchange_sync<WAVE>();

      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_store<1>(v[k], t, line + k * linestride);
      }

      exchange_sync<WAVE>();
    }

    if (Plan<N, E_>::NP >= 4)
    {
      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_load<2>(v[k], t, line + k * linestride, midtab, w);
      }

      exchange_sync<WAVE>();

      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_store<2>(v[k], t, line + k * linestride);
      }

      exchange_sync<WAVE>();
    }

    if (Plan<N, E_>::NP >= 5)
    {
      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_load<3>(v[k], t, line + k * linestride, midtab, w);
      }

      exchange_sync<WAVE>();

      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_store<3>(v[k], t, line + k * linestride);
      }

      exchange_sync<WAVE>();
    }

    if (Plan<N, E_>::NP >= 6)
    {
      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_load<4>(v[k], t, line + k * linestride, midtab, w);
      }

      exchange_sync<WAVE>();

      if (active)
      {
        #pragma unroll
        for(int k = 0; k < K; ++k)
          L::template mid_store<4>(v[k], t, line + k * linestride);
      }

      exchange_sync<WAVE>();
    }

    before_last();

    if (active)
    {
      #pragma unroll
      for(int k = 0; k < K; ++k)
        L::last(v[k], t, line + k * linestride, w);
    }

    exchange_sync<WAVE>();
  }

  //|---------------------- packed step: two complex transforms instead of three --
  // ocean.map keeps only the REAL part of each of the three inverse transforms (map.comp:62-64).  The real part
  // of the 2-D inverse transform of F is the transform of F's Hermitian part  F_H[k] = (F[k] + conj(F[-k])) / 2
  // (-k = index negation modulo N), and two Hermitian spectra travel through one complex transform as A + i B:
  //
  //     C = h_H + i hx_H                        -> Re = height,          Im = choppy x
  //     D = hy_H + i (-2 i sin(2 pi n / N)) h_H -> Re = choppy y,        Im = height[x-1] - height[x+1]
  //
  // (stored as 2 C and 2 D: F[k] + conj(F[-k]) without the halving, which the column pass folds into its sign factor)
  //
  // (the second term of D is the transfer function of the central difference of map.comp:72-75, periodic wrap
  // included: the x slope needs no neighbouring columns any more).  So the row pass transforms and writes two
  // fields instead of three, the column pass reads and transforms two instead of three plus a halo, and one
  // 16-byte value per point (C, D) crosses between them: 80 B/pt of HBM traffic instead of 96 (72 with the 24-byte texels).
  //
  // The Hermitian part pairs index (y, x) with ((N-y) % N, (N-x) % N) -- not ocean.sim's own mirror (N-1-y, N-1-x)
  // -- so a row-pass workgroup takes rows y and N-y together (rows 0 and N/2 pair with themselves and share a
  // workgroup), evaluates ocean.sim once per point and swaps the values through LDS.
  // k^ at the negated index is -k^, except where the index is its own negative (x = 0 or y = 0: the grid's k there
  // is -pi N scale at both): the general form  F_H = (F[k] + conj(F[-k])) / 2  is evaluated with the sign that
  // applies, so the result equals the reference's three transforms to rounding (tests/test_oracle_pins.py).

  // 2048^2 (round 4, profiles/r04_structure_variants.txt): the 1024^2 column pass's recipe in the row pass -- 16 points per thread (2048 = 16 x 16 x 8:
  // three passes, two exchanges instead of four and three), the two fields one after the other through one LDS line per row (35 KB
  // per 256-thread pair), three workgroups per CU at 144 registers, no spill: row pass 31.2 -> 28.6 us (x 1), 118.9 -> 112.2 us (x 4),
  // with the fp16-stored spectrum 29.1 -> 25.7 us.  Not at 1024^2 (28.9 against 25.2 us), not with four per CU (128 registers, 72 bytes
  // of spill: 33.8 us), not without the sequential fields (65 KB, two 4-wave workgroups per CU: 31.2 us).
  template<int N, bool H16 = false>
  struct RowCfg
  {
    static constexpr int E = (N >= 2048) ? 16 : default_radix(N);         // points per thread
    static constexpr int T = Plan<N, E>::T;
    // threads of a row-pass workgroup when one row pair needs fewer (small grids are latency-bound: more, smaller workgroups;
    // 512^2 x 1: 9.2 us with 128 against 9.9 us with 256)
    static constexpr int PAIR_THREADS = 128;
    static constexpr int PAIRS = (2 * T >= PAIR_THREADS) ? 1 : PAIR_THREADS / (2 * T);   // row pairs per workgroup
    static constexpr int THREADS = 2 * T * PAIRS;
    // SEQ (from 1024^2 up): the two packed fields one after the other through ONE LDS line per row (half the LDS: two 512-thread
    // row pairs per CU at 4096^2, three or four 256-thread ones at 2048^2, SIX at 1024^2), the other field waits in registers.
    // Up to round 4 the fp32 spectrum at 4096^2 spilled in this form (C's results wait in 32 registers: 155 us) and ran as one
    // persistent 1024-thread workgroup per CU that walked its pairs with the next pair's inputs requested early (131-132 us); with
    // round 5's registers (122, no spill) the sequential form is the faster one there too: 120.4-121.8 against 123.0-123.6 us, and the
    // walking form is gone.  At 1024^2 the form lost in round 4 (28.9 against 25.2 us, 100 registers compiled for four per CU); with
    // round 5's layouts it needs 80 registers and 18.9 KB, six workgroups per CU instead of four: 24.1 against 25.2 us
    // (profiles/r05_rowpass_forms.txt).  Not below: 512^2 is latency-bound and the form doubles the barrier phases.
    static constexpr bool SEQ = N >= 1024;
    static constexpr int K = SEQ ? 1 : 2;                                  // LDS lines per row
    static constexpr int LINE = LineFFT<N, 1, E>::LINE;                    // >= N + 2: element 0 once more at index N (the Hermitian swap)
    static constexpr int GROUPS = (N / 2) / PAIRS;                      // workgroups per cascade
    static constexpr size_t LDS = ((size_t)LineFFT<N, 1, E>::MIDTAB + (size_t)PAIRS * 2 * K * LINE) * sizeof(cf);

    static constexpr int FIT = (int)(((size_t)160 * 1024) / LDS);                              // workgroups per CU the LDS allows
    static constexpr int SEQ_PER_CU = (N == 2048) ? 3 : (N == 1024 ? 6 : 4);                                     // sequential form: workgroups per CU the registers are budgeted for
    static constexpr int MIN_WAVES = SEQ ? ((THREADS / 64) * (FIT > SEQ_PER_CU ? SEQ_PER_CU : FIT) + 3) / 4 : 1;   // per SIMD, for __launch_bounds__

    static_assert((N / 2) % PAIRS == 0, "row pairs per workgroup must divide N / 2");
  };

  // sin(alpha + 2 pi s / E) for the E slots of a thread from (cos, sin)(alpha)
  template<int E>
  __device__ __forceinline__ float slot_sine(cf ca, int s)
  {
    constexpr float R = 0.70710678118654752440f;

    if (E == 8)
    {
      switch(s & 7)
      {
        case 0: return ca.y;
        case 1: return R * (ca.y + ca.x);
        case 2: return ca.x;
        case 3: return R * (ca.x - ca.y);
        case 4: return -ca.y;
        case 5: return -R * (ca.y + ca.x);
        case 6: return -ca.x;
        default: return R * (ca.y - ca.x);
      }
    }
    else if (E == 16)
    {
      // sin(alpha + s pi / 8) = sin(alpha) cos(s pi / 8) + cos(alpha) sin(s pi / 8)
      constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
      constexpr float cs[16] = { 1.0f, C1, R, S1, 0.0f, -S1, -R, -C1, -1.0f, -C1, -R, -S1, 0.0f, S1, R, C1 };
      constexpr float sn[16] = { 0.0f, S1, R, C1, 1.0f, C1, R, S1, 0.0f, -S1, -R, -C1, -1.0f, -C1, -R, -S1 };

      return fmaf(ca.y, cs[s & 15], ca.x * sn[s & 15]);
    }
    else
    {
      switch(s & 3)
      {
        case 0: return ca.y;
        case 1: return ca.x;
        case 2: return -ca.y;
        default: return -ca.x;
      }
    }
  }

  // One group of row pairs per workgroup.
  // WILD: a phase may lie outside [0, 2 pi) (sincos_phase_pair); its own instantiation rather than a branch in the kernel: the
  // registers of the second path cost the 4096^2 fp16 form 12 more bytes of spill and 6 us (profiles/r04_rowpass_packed.txt)
  // H0H (with H16 only; DATUM_OCEAN_SPECTRUM_FP16_H0, SURVEY.md 8d's own count for BASELINE configs[4]: "spectrum + intermediates stored fp16"):
  // h0 is read as two halves per point -- 4 instead of 8 bytes for the row itself and for ocean.sim's mirror row, 32 input registers
  // fewer in flight at 16 points per thread -- from the copy size_spectrum_scale keeps (h0 times a power of two per cascade that brings
  // max |h0| just under 2^15).  Everything up to the store is linear in h0, so that power of two is taken out again together with the
  // work spectrum's own scale: one factor, CascadeConst::rowscale, exact.
  template<int N, bool H16, bool WILD = false, bool H0H = false>
  __global__ void OCEAN_LDS_UNPAIRED __launch_bounds__((RowCfg<N, H16>::THREADS), (RowCfg<N, H16>::MIN_WAVES)) ocean_rowpass_kernel(StepArgs a)
  {
    static_assert(H16 || !H0H, "h0 as halves goes with the fp16-stored work spectrum");

    typedef RowCfg<N, H16> C;
    typedef Plan<N, C::E> P;
    typedef LineFFT<N, 1, C::E> L;

    constexpr int E = P::E;
    constexpr int T = P::T;
    constexpr int K = C::K;
    static_assert(E == 16 || E == 8 || E == 4, "slot_sine covers E = 4, 8 and 16");
    static_assert(elem_in<N, E>(0, 1) == T && elem_out<N, E>(0, 1) == T, "slots are T columns apart");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    cf *midtab = reinterpret_cast<cf*>(smem);

    for(int i = threadIdx.x; i < L::MIDTAB; i += C::THREADS)
      midtab[i] = L::midtab_entry(a.tw, i);

    // work items = (cascade, group of PAIRS row pairs), cascade-major; workgroup b takes items b, b + gridDim.x, ...
    // Neighbouring pairs read each other's rows as ocean.sim's mirror rows: item -> group deals contiguous bands of
    // groups to the workgroups of one XCD (b % 8)
    constexpr int G = C::GROUPS;

    auto group_of = [&](int item) { int const q = item % G; return (G % 8 == 0) ? (q & 7) * (G / 8) + (q >> 3) : q; };

    int const pr = threadIdx.x / (2 * T);
    int const half = (threadIdx.x % (2 * T)) / T;
    int const t_ = threadIdx.x % T;

    size_t const plane = (size_t)N * N;

    constexpr int DBO = (int)(blocked<N, H16>(0, T) - blocked<N, H16>(0, 0));

    static_assert(T % spec_block_cols(H16) == 0, "slots must be whole blocks apart");
    static_assert(C::LINE >= N + 2 && C::LINE % 2 == 0, "the Hermitian swap fits the line; lines stay 16-byte aligned");

    typedef typename SpecValue<H16>::type SV;

    bool const advance = a.ndt > 0;

    // rows p and N - p of the item; p = 0: rows 0 and N/2, each its own partner
    auto row_of = [&](int item) { int const p = group_of(item) * C::PAIRS + pr; return half ? (p == 0 ? N / 2 : N - p) : p; };

    // inputs of ocean.sim: this row of phase and h0, the mirror row (sim.comp:59) backwards, the dispersion of the row
    typedef typename std::conditional<H0H, unsigned int, float2>::type H0V;      // a point of h0 as it is loaded

    struct Inputs
    {
      float ph[E], om[E];
      H0V hk[E], hm[E];
    };

    // The halves are widened once, right behind the loads, by two instructions written out (the high half through SDWA): with the C++
    // casts hipcc sinks the conversions into the sim loop and the 1024^2 form spills 28 bytes under its 80 registers; widened where
    // they are used the row pass is 4-10 % slower at every size from 512^2 up (profiles/r06_h0_halves.txt).  Exact: every half is a float.
    auto height = [](H0V const &v) -> cf
    {
      if constexpr (H0H)
      {
#if defined(__HIP_DEVICE_COMPILE__)
        float lo, hi;
        asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(lo) : "v"(v));
        asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(hi) : "v"(v));
        return cf{ lo, hi };
#else
        typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
        half2_ const h = __builtin_bit_cast(half2_, v);
        return cf{ (float)h.x, (float)h.y };
#endif
      }
      else
        return cf{ v.x, v.y };
    };

    auto request = [&](int item, int t, Inputs &in)
    {
      int const cascade = a.first + item / G;
      int const y = row_of(item);

      constexpr size_t QUAD = (size_t)(N / 2 + 1) * (N / 2 + 1);

      __amdgpu_buffer_rsrc_t romega = make_rsrc(a.omega + cascade * QUAD, QUAD * sizeof(float));
      __amdgpu_buffer_rsrc_t rphase = make_rsrc(a.phase + cascade * plane, plane * sizeof(float));
      __amdgpu_buffer_rsrc_t rh0 = H0H ? make_rsrc(a.h0h + cascade * plane, plane * sizeof(unsigned int)) : make_rsrc(a.h0 + cascade * plane, plane * sizeof(float2));

      int const e0 = y * N + t;
      int const m0 = (N - 1 - y) * N + (N - 1 - t - T * (E - 1));

      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        in.ph[s] = buf_load_f32(rphase, e0 * 4, T * s * 4);

        if constexpr (H0H)
        {
          in.hk[s] = buf_load_u32(rh0, e0 * 4, T * s * 4);
          in.hm[s] = buf_load_u32(rh0, m0 * 4, T * (E - 1 - s) * 4);
        }
        else
        {
          in.hk[s] = buf_load_f32x2(rh0, e0 * 8, T * s * 8);
          in.hm[s] = buf_load_f32x2(rh0, m0 * 8, T * (E - 1 - s) * 8);
        }
      }

      if (advance)
      {
        // dispersion_lookup (ocean_omega_kernel's quadrant table) through the buffer path.  |x - N/2| of slot s is N/2 - t - T s in the
        // lower half of the slots and t + T (s - E/2) in the upper: two per-thread offsets and compile-time steps, not one address
        // register per slot
        int const i = abs(y - N / 2);
        int const lower = (i * (N / 2 + 1) + N / 2 - t - T * (E / 2 - 1)) * 4;       // slot E/2 - 1, the lowest address of the lower half
        int const upper = (i * (N / 2 + 1) + t) * 4;                                 // slot E/2

        #pragma unroll
        for(int s = 0; s < E; ++s)
          in.om[s] = (s < E / 2) ? buf_load_f32(romega, lower, T * (E / 2 - 1 - s) * 4) : buf_load_f32(romega, upper, T * (s - E / 2) * 4);
      }
    };

    cf const ca_ = a.tw[t_];                  // exp(2 pi i t / N)

    typename LineTw<N, E>::type w_;
    LineTw<N, E>::load(a.tw, t_, w_);

#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = a.stamps + (size_t)blockIdx.x * 16;
#endif
    OCEAN_STAMP_WHERE();
    OCEAN_STAMP(0);

    auto one_item = [&](int item, Inputs &in)
    {
      int const t = t_;
      cf const ca = ca_;
      typename LineTw<N, E>::type const &w = w_;

      OCEAN_STAMP(1);

      int const cascade = a.first + item / G;
      int const p = group_of(item) * C::PAIRS + pr;
      int const y = row_of(item);
      int const otherhalf = (p == 0) ? half : 1 - half;

      // (SEQ: the row's one line is the swap buffer first and the transforms' line afterwards)
      cf *line = midtab + L::MIDTAB + (pr * 2 + half) * K * C::LINE;
      cf *swap_out = line + (K - 1) * C::LINE;
      cf const *swap_in = midtab + L::MIDTAB + (pr * 2 + otherhalf) * K * C::LINE + (K - 1) * C::LINE;

      CascadeConst const cc = a.casc[cascade];

      __amdgpu_buffer_rsrc_t rphase = make_rsrc(a.phase + cascade * plane, plane * sizeof(float));
      // (the work spectrum is per LAUNCH: slot cascade - first.  Every cascade group writes and reads the same slots, which therefore stay in the
      // Infinity Cache from one group to the next instead of going out to HBM and back once per cascade)
      __amdgpu_buffer_rsrc_t rspec = make_rsrc(static_cast<SV*>(a.spec) + (cascade - a.first) * plane, plane * sizeof(SV));

      float ph[E];
      cf hk[E], hm[E];

      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        ph[s] = in.ph[s];
        hk[s] = height(in.hk[s]);
        hm[s] = height(in.hm[s]);
      }

      // update_ocean (ocean.cpp:223-233), each pending dt in turn
      if (advance)
      {
        for(int k = 0; k < a.ndt; ++k)
        {
          float const dt = a.dt[k];

          // two slots per instruction: w = omega dt, a = phase + w, b = a - 2 pi (the roundings of phase + w dt, see advance_phase), then the select
          f2_ const dt2 = { dt, dt };

          #pragma unroll
          for(int s = 0; s < E; s += 2)
          {
            f2_ const w = f2_{ in.om[s], in.om[s + 1] } * dt2;
            f2_ const sum = f2_{ ph[s], ph[s + 1] } + w;
            f2_ const wrapped = sum - f2_{ 6.2831855f, 6.2831855f };

            ph[s] = (sum.x >= 6.2831855f) ? wrapped.x : sum.x;
            ph[s + 1] = (sum.y >= 6.2831855f) ? wrapped.y : sum.y;
          }
        }

        #pragma unroll
        for(int s = 0; s < E; ++s)
        {
          buf_store_f32_aux<PHASE_STORE_AUX>(ph[s], rphase, (y * N + t) * 4, T * s * 4);
        }
      }

      // ocean.sim once per point; the value goes to the thread that holds the negated index
      cf h[E];

      // (two slots per instruction throughout the prologue: since round 5's register savings at every size -- 2048^2 x 4 row pass 110 -> 106-107 us)
      #pragma unroll
      for(int s = 0; s < E; s += 2)
      {
        f2_ sn, cs;
        sincos_phase_pair<WILD>(f2_{ ph[s], ph[s + 1] }, sn, cs);

        #pragma unroll
        for(int i = 0; i < 2; ++i)
        {
          // sim_height_products with e^{i phase} from the pair
          cf const e = cf{ cs[i], sn[i] };

          h[s + i] = add_conj(cmul(hk[s + i], e), cmul(hm[s + i], e));

          swap_out[t + T * (s + i)] = h[s + i];
        }
      }

      // element 0 once more at index N: the partner of x is N - x for every x, without a wrap
      if (t == 0)
        swap_out[N] = h[0];

      __syncthreads();

      OCEAN_STAMP(2);

      float const ky = wavevector(y, N, cc.scale);

      // k^ keeps its sign where the index is its own negative (x = 0, y = 0): there the partner enters hx / hy negated
      float const cy = (y == 0) ? -2.0f : 0.0f;
      float const cx = (t == 0) ? -2.0f : 0.0f;

      cf v[2][E];

      // k of sim.comp:52 for two slots at a time: (float)x - N/2 is an exact integer, so xf0 + T s equals it bit for bit, and the
      // products keep wavevector()'s order; 1 / |k| by the bare reciprocal square root of max(k^2, smallest normal) -- at the one
      // point with k = 0 both components of k are 0 and k^ comes out 0 as sim.comp:54's guard has it
      float const xf0 = (float)t - 0.5f * (float)N;
      f2_ const ky2 = { ky * ky, ky * ky };
      f2_ const kyp = { ky, ky };

      #pragma unroll
      for(int s = 0; s < E; s += 2)
      {
        f2_ const xf = { xf0 + (float)(T * s), xf0 + (float)(T * (s + 1)) };
        f2_ const kx = (f2_{ 6.2831855f, 6.2831855f } * xf) * f2_{ cc.scale, cc.scale };
        f2_ const k2 = kx * kx + ky2;
        f2_ const kinv = { __builtin_amdgcn_rsqf(__builtin_fmaxf(k2.x, 1.17549435e-38f)), __builtin_amdgcn_rsqf(__builtin_fmaxf(k2.y, 1.17549435e-38f)) };
        f2_ const khx = kx * kinv, khy = kyp * kinv;
        f2_ const s2 = { 2.0f * slot_sine<E>(ca, s), 2.0f * slot_sine<E>(ca, s + 1) };       // 2 sin(2 pi x / N)

        #pragma unroll
        for(int i = 0; i < 2; ++i)
        {
          cf const n = swap_in[(N - t) - (s + i) * T];

          cf const hh = add_conj(h[s + i], n);                                // h~[k] + conj(h~[-k])
          cf const hhx = (s + i == 0) ? fma_conj(hh, n, cx) : hh;
          cf const hhy = fma_conj(hh, n, cy);

          if (i == 0)
          {
            v[0][s + i] = fma_real_h<0>(hh, hhx, khx);
            v[1][s + i] = fma_negi_h<0>(scale_real_h<0>(hh, s2), hhy, khy);
          }
          else
          {
            v[0][s + i] = fma_real_h<1>(hh, hhx, khx);
            v[1][s + i] = fma_negi_h<1>(scale_real_h<1>(hh, s2), hhy, khy);
          }
        }
      }

      // every thread has fetched its partner values before pass 0 overwrites the lines
      __syncthreads();

      OCEAN_STAMP(3);

      if constexpr (C::SEQ)
      {
        // C through the row's line, then D through the same line; C's results wait in registers for the one store per point
        fft_lines<N, 1, 1, E>(reinterpret_cast<cf (&)[1][E]>(v[0]), t, line, C::LINE, midtab, w, true);
        fft_lines<N, 1, 1, E>(reinterpret_cast<cf (&)[1][E]>(v[1]), t, line, C::LINE, midtab, w, true);
      }
      else
        fft_lines<N, K, 1, E>(v, t, line, C::LINE, midtab, w, true);

      OCEAN_STAMP(4);

      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        if constexpr (H16)
        {
          // round to nearest even; the host picks specscale so that no row sum can overflow (ocean_capi: spectrum_scale)
          float const sc = H0H ? cc.rowscale : cc.specscale;
          half4_ const hv = { (_Float16)(v[0][s].x * sc), (_Float16)(v[0][s].y * sc), (_Float16)(v[1][s].x * sc), (_Float16)(v[1][s].y * sc) };

          buf_store_cf_aux<SPEC_STORE_AUX>(__builtin_bit_cast(cf, hv), rspec, (int)blocked<N, true>(y, t) * 8, DBO * s * 8);
        }
        else
        {
          buf_store_f32x4_aux<SPEC_STORE_AUX>(make_float4(v[0][s].x, v[0][s].y, v[1][s].x, v[1][s].y), rspec, (int)blocked<N>(y, t) * 16, DBO * s * 16);
        }
      }

      OCEAN_STAMP(5);
    };

    int item = (int)blockIdx.x;

    Inputs in;

    request(item, t_, in);

    OCEAN_WAIT_LOADS();

    one_item(item, in);
  }

  template<int N>
  struct ColCfg
  {
    // 16 points per thread at 1024^2 (1024 = 16 x 16 x 4: three passes, two exchanges, 256-thread tiles of four columns),
    // with the two fields one after the other (half the LDS: 35 KB, four workgroups per CU at 124 registers): column pass
    // 34.2 -> 31.3 us in round 2 (profiles/r02_col_radix16.txt).  Round 5: the same at 2048^2 (2048 = 16 x 16 x 8: 512-thread tiles, 66 KB,
    // TWO per CU, one tile per workgroup instead of one persistent 1024-thread workgroup per CU that walks its tiles): 31.5 -> 26.5 us
    // at 2048^2 x 1, 144.7 -> 137-143 us at x 4 -- in round 4's layouts the form was a loss there (168 against 160 us).  Not at 4096^2,
    // whose maps are beyond the Infinity Cache: 173 against 147 us (the walking form's prefetch overlaps HBM-bound stores), nor at
    // 512^2 (8.2 against 5.5 us).  profiles/r05_colpass_forms.txt
    static constexpr int E = (N == 1024 || N == 2048) ? 16 : default_radix(N);         // points per thread
    static constexpr int T = Plan<N, E>::T;
    static constexpr int K = (E == 16) ? 1 : 2;                            // fields per set of barrier phases: 2 = together, 1 = one after the other (half the LDS)
    // LDS: the W columns of a tile element by element in one array per field (position * W + column: threads are
    // column-fastest, so consecutive lanes touch consecutive addresses); the exchange layouts depend on W (LineFFT).  The
    // line length does too, hence the two steps: the widest tile the threads allow, narrowed until its lines fit
    // 512^2: two-column tiles of 128 threads -- a single cascade is then 256 workgroups, one per CU, instead of 128 on half the chip
    // (round 5: column pass 3.69 -> 3.39 us inside the kernel, step 11.9 -> 11.6 us; ocean.gen from 512^2 maps 18.1 -> 16.6 us with
    // the 2 x 8 patches that come with it; four cascades: no change.  profiles/r05_sizes.txt)
    static constexpr int WRAW = (N == 512 ? 128 : ((T < 128) ? 256 : (T == 128) ? 512 : 1024)) / T;
    template<int W_> static constexpr bool fits() { return (size_t)LineFFT<N, 1, E>::MIDTAB * sizeof(cf) + (size_t)K * W_ * LineFFT<N, W_, E>::LINE * sizeof(cf) <= (size_t)160 * 1024; }
    static constexpr int WFIT = fits<8>() ? 8 : fits<4>() ? 4 : fits<2>() ? 2 : 1;
    static constexpr int WCAP = WRAW > 8 ? 8 : (WRAW < 1 ? 1 : WRAW);
    static constexpr int W = WCAP < WFIT ? WCAP : WFIT;                 // columns per workgroup, one per thread group
    static constexpr int THREADS = W * T;
    static constexpr int MIN_WAVES = (N == 2048) ? 4 : 1;        // per SIMD, for __launch_bounds__: two 512-thread tiles per CU at 2048^2 (128 registers, 20 bytes of spill)
    static constexpr int CS = LineFFT<N, W, E>::LINE;                      // elements per column line
    static constexpr int TILES = N / W;

    static constexpr size_t OFF_MAIN = (size_t)LineFFT<N, 1, E>::MIDTAB * sizeof(cf);
    static constexpr size_t MAIN_FFT = (size_t)W * K * CS * sizeof(cf);
    static constexpr size_t MAIN_DZ = (size_t)W * N * sizeof(float);
    static constexpr size_t LDS = OFF_MAIN + (MAIN_FFT > MAIN_DZ ? MAIN_FFT : MAIN_DZ);

    static_assert(N % W == 0 && 8 % W == 0, "a tile must sit inside one 8-column block");
    static_assert(W == map_patch_cols(N) && T % map_patch_rows(N) == 0, "a patch of the map layout is 16 neighbouring lanes of a column-pass wave");
    static_assert(OFF_MAIN % 16 == 0, "LDS carve must stay 16-byte aligned");
  };

  // N <= 2048: one tile per workgroup (2-4 workgroups per CU overlap each other's memory and arithmetic phases; a
  // persistent variant was measured 15 % slower at 1024^2).  N = 4096: one 1024-thread workgroup fills a CU (LDS) and its
  // phases -- 8 loads per thread, the transforms, 16 stores per thread -- ran one after the other (4096^2: 64 us of
  // transforms + 72 us of loads + 110 us of stores = the 247 us measured).  There the workgroups are persistent: each
  // walks a run of tiles, requests the next tile's values before transforming the current one and lets the current
  // tile's stores drain under the next tile's transforms.
  // measured (profiles/r02_large_grids.txt), with the band layouts: 2048^2 x 4 182 -> 168 us, 4096^2 219 -> 210 us; without
  // them 208 -> 187 us and 248 -> 259 us (not every step of this was a gain on its own)
  template<int N, bool H16> constexpr bool col_walks() { return N >= 4096; }

  // STREAM: the maps' stores are streamed (nt) instead of written through -- where the handle's working set (h0 8 + phase 4 + work spectrum
  // 16 or 8 + maps 24 bytes per point and cascade) is beyond the Infinity Cache; the policy is part of the instruction, hence a template
  // flag.  4096^2 always streams (one cascade is 0.9 GB); grids below 1024^2 never do (sixteen cascades of 512^2 still fit).
  template<int N> constexpr bool col_has_stream_variant() { return N == 1024 || N == 2048; }

  // the working set up to which writing the maps through wins: 1024^2 x 4 (218 MB) 82.4 k grids/s written through against 78.9 k streamed, x 5
  // (272 MB) 72.4-74.0 against 72.1-72.5 k, x 6 (327 MB) 68.8 against 76.1 k; 2048^2 x 1 and the fp16 spectrum's x 4 / x 6 (185 / 277 MB)
  // written through by 0-3 % (profiles/r06_store_policies.txt)
  constexpr double MAPS_RESIDENT_BYTES = 300.0e6;

  inline bool maps_stream(int N, int cascades, bool half)
  {
    return N >= 4096 || (N >= 1024 && (double)cascades * N * N * (half ? 44.0 : 52.0) > MAPS_RESIDENT_BYTES);
  }

  template<int N, bool H16, bool STREAM>
  __device__ __forceinline__ void colpass_body(StepArgs const &a)
  {
    typedef ColCfg<N> C;
    typedef Plan<N, C::E> P;
    typedef LineFFT<N, C::W, C::E> L;

    constexpr int E = P::E;
    constexpr int T = P::T;
    constexpr int W = C::W;
    constexpr int K = C::K;
    constexpr bool WALK = col_walks<N, H16>();

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    cf *midtab = reinterpret_cast<cf*>(smem);
    cf *lines = reinterpret_cast<cf*>(smem + C::OFF_MAIN);            // [K][CS][W]
    float *dzmain = reinterpret_cast<float*>(smem + C::OFF_MAIN);     // [N][W], after the transforms

    for(int i = threadIdx.x; i < L::MIDTAB; i += C::THREADS)
      midtab[i] = L::midtab_entry(a.tw, i);

    constexpr int NT = C::TILES;

    size_t const plane = (size_t)N * N;

    int const cp_ = threadIdx.x % W;
    int const t_ = threadIdx.x / W;

    constexpr int DBI = (int)(blocked<N, H16>(T, 0) - blocked<N, H16>(0, 0));      // blocked spectrum, slot to slot (elements)

    static_assert(T % SBR == 0, "slots must be whole blocks apart");

    typedef typename SpecValue<H16>::type SV;
    typedef typename std::conditional<H16, cf, float4>::type Raw;        // one point of the work spectrum as loaded

    // work items = (cascade, tile), cascade-major; workgroup b takes items b, b + gridDim.x, ...  Tiles narrower than a
    // block of the spectrum share its cache lines with their neighbours: item -> tile deals contiguous bands of tiles to
    // the workgroups of one XCD (b % 8), so that neighbours run there at the same time
    int const items = NT * a.cascades;

    auto tile_of = [&](int item) { int const q = item % NT; return (NT % 8 == 0) ? (q & 7) * (NT / 8) + (q >> 3) : q; };

    auto request = [&](int item, int t, int cp, Raw (&q)[E])
    {
      int const cascade = a.first + item / NT;
      int const x = tile_of(item) * W + cp;

      __amdgpu_buffer_rsrc_t rspec = make_rsrc(static_cast<SV const*>(a.spec) + (cascade - a.first) * plane, plane * sizeof(SV));

      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        if constexpr (H16)
          q[s] = buf_load_cf(rspec, (int)blocked<N, true>(t, x) * 8, DBI * s * 8);
        else
          q[s] = buf_load_f32x4_aux<0>(rspec, (int)blocked<N>(t, x) * 16, DBI * s * 16);
      }
    };

    auto unpack = [&](Raw const (&q)[E], cf (&v)[2][E])
    {
      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        if constexpr (H16)
        {
          half4_ const hv = __builtin_bit_cast(half4_, q[s]);

          v[0][s] = cf{ (float)hv.x, (float)hv.y };
          v[1][s] = cf{ (float)hv.z, (float)hv.w };
        }
        else
        {
          v[0][s] = cf{ q[s].x, q[s].y };
          v[1][s] = cf{ q[s].z, q[s].w };
        }
      }
    };

#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = a.stamps + ((size_t)65536 + blockIdx.x) * 16;
#endif
    OCEAN_STAMP_WHERE();
    OCEAN_STAMP(0);

    typename LineTw<N, E>::type w_;
    LineTw<N, E>::load(a.tw, t_, w_);

    // One tile: `q` holds its values as loaded; when `more`, the next tile's values are requested into `q` again between
    // the last exchange and the last pass (registers are free there) and are in flight during the last pass, the map
    // arithmetic and this tile's 16 stores, which then drain under the next tile's transforms.  The wait for those loads
    // stands at the top of the next tile: the loads are OLDER than the stores, so a counted wait (vmcnt = the stores
    // issued since) does not wait for the stores.  hipcc counts such a wait only if every path into it has the same
    // history, hence the first tile is peeled off the loop below.
    auto one_tile = [&](int item, Raw (&q)[E], bool more, int next)
    {
      // Inside the walking loop everything derived from the thread's coordinates -- LDS addresses of every pass, store
      // offsets, twiddle powers -- is loop invariant, and hipcc hoists it all out of the loop and keeps it in registers
      // (128 VGPRs and spills against 70 for one tile; a spill's reload waits for vmcnt(0), i.e. for the very stores
      // that should drain under the next tile).  Opaque copies of the coordinates make that work belong to the tile.
      int t = t_, cp = cp_;

      if constexpr (WALK)
        asm volatile("" : "+v"(t), "+v"(cp));

      // (the per-thread twiddles stay in registers across tiles: a load issued here would be younger than the previous
      // tile's stores, and waiting for it would wait for them)
      typename LineTw<N, E>::type w = w_;

      if constexpr (WALK)
      {
        #pragma unroll
        for(int k = 0; k < LineTw<N, E>::type::NMIDREG; ++k)
          asm volatile("" : "+v"(w.mid[k].x), "+v"(w.mid[k].y));

        #pragma unroll
        for(int m = 0; m < P::M; ++m)
          asm volatile("" : "+v"(w.last[m].x), "+v"(w.last[m].y));
      }

      cf v[2][E];

      unpack(q, v);

      auto prefetch = [&]()
      {
        if constexpr (WALK)
        {
          if (more)
            request(next, t, cp, q);
        }
      };

      if constexpr (K == 2)
        fft_lines<N, 2, W, E>(v, t, lines + cp, W * C::CS, midtab, w, true, prefetch);    // lines [K][CS][W]
      else
      {
        #pragma unroll
        for(int f = 0; f < 2; ++f)
        {
          cf u[1][E];

          #pragma unroll
          for(int s = 0; s < E; ++s)
            u[0][s] = v[f][s];

          if (f == 1)
            fft_lines<N, 1, W, E>(u, t, lines + cp, W * C::CS, midtab, w, true, prefetch);
          else
            fft_lines<N, 1, W, E>(u, t, lines + cp, W * C::CS, midtab, w, true);

          #pragma unroll
          for(int s = 0; s < E; ++s)
            v[f][s] = u[0][s];
        }
      }

      OCEAN_STAMP(2);

      int const cascade = a.first + item / NT;
      int const x = tile_of(item) * W + cp;

      CascadeConst const cc = a.casc[cascade];

      __amdgpu_buffer_rsrc_t rmaps = make_rsrc(reinterpret_cast<char const*>(a.maps) + (size_t)cascade * map_cascade_bytes(N), map_cascade_bytes(N));

      // (-1)^(x+y) of map.comp:60; a thread's rows differ by even amounts
      // times the 1/2 of the Hermitian parts, which the row pass leaves out
      float const sig = (((x + t) & 1) ? -0.5f : 0.5f) * cc.specinv;
      float const sigchop = sig * cc.choppiness;

      // the transform lines are free after the last barrier of fft_lines: heights of this column for the y slope
      float *own = dzmain + cp;

      #pragma unroll
      for(int s = 0; s < E; ++s)
        own[(t + T * s) * W] = v[0][s].x * sig;

      __syncthreads();

      OCEAN_STAMP(3);

      float const nz = cc.nz;

      // this thread's texel of its patch, part A and part B; slot to slot T / PH patch rows on
      int const oa = (int)map_compact_a(N, t, x);
      int const ob = (int)map_compact_b(N, t, x);

      constexpr int SLOTBYTES = (T / map_patch_rows(N)) * map_compact_patchrow_bytes(N);

      // (written through while the handle's working set is resident in the Infinity Cache, streamed beyond it: MAP_STORE_AUX above)
      constexpr int MAPAUX = (N <= 2048 && !STREAM) ? MAP_STORE_AUX : MAP_STORE_AUX_STREAM;

      #pragma unroll
      for(int s = 0; s < E; ++s)
      {
        int const y = t + T * s;

        // displacement (map.comp:62-64) and central-difference normal (map.comp:72-77)
        float const dz = v[0][s].x * sig;
        float const dx = v[0][s].y * sigchop;
        float const dy = v[1][s].x * sigchop;

        float const nx = -(v[1][s].y * sig);
        float const ny = own[((y + 1) & (N - 1)) * W] - own[((y + N - 1) & (N - 1)) * W];
        float const inv = rsqrtf(nx * nx + ny * ny + nz * nz);

        // 16 neighbouring lanes = one patch: 256 contiguous bytes (two lines) by the first instruction, 128 (one line) by the second
        buf_store_f32x4_aux<MAPAUX>(make_float4(dx, dy, dz, nx * inv), rmaps, oa, SLOTBYTES * s);
        buf_store_cf_aux<MAPAUX>(cf{ ny * inv, nz * inv }, rmaps, ob, SLOTBYTES * s);
      }

      OCEAN_STAMP(4);
    };

    int item = (int)blockIdx.x;

    Raw q[E];

    request(item, t_, cp_, q);

    OCEAN_WAIT_LOADS();
    OCEAN_STAMP(1);

    if constexpr (!WALK)
      one_tile(item, q, false, 0);
    else
    {
      int const stride = (int)gridDim.x;

      // first tile (peeled), then the rest
      one_tile(item, q, item + stride < items, item + stride);

      for(item += stride; item < items; item += stride)
      {
        // the heights of the previous tile were read from the LDS this tile's first pass writes
        __syncthreads();

        one_tile(item, q, item + stride < items, item + stride);
      }
    }
  }

  template<int N, bool H16, bool STREAM = false>
  __global__ void OCEAN_LDS_UNPAIRED __launch_bounds__(ColCfg<N>::THREADS, ColCfg<N>::MIN_WAVES) ocean_colpass_kernel(StepArgs a)
  {
    colpass_body<N, H16, STREAM>(a);
  }

  // The same kernel with hipcc's pairing of LDS accesses left on.  1024^2 only: there the column pass (16 points per thread, 37 KB of
  // LDS per 256-thread tile) is worth more with FOUR tiles per CU than with 64-bank reads, and the register allocation fits four
  // (124 registers) only in the paired form (unpaired: 138, or 128 with 44 bytes of spill): 24.8-25.1 against 25.7-26.2 us at
  // 1024^2 x 4, 20.7 against 23.5 us with the fp16-stored spectrum (profiles/r05_lds_conflicts.txt).
  template<int N> constexpr bool col_pairs_lds() { return N == 1024; }

  namespace paired
  {
    template<int N, bool H16, bool STREAM = false>
    __global__ void __launch_bounds__(ColCfg<N>::THREADS, ColCfg<N>::MIN_WAVES) ocean_colpass_kernel(StepArgs a)
    {
      colpass_body<N, H16, STREAM>(a);
    }
  }

  // the column-pass kernel the module launches at this resolution
  template<int N, bool H16, bool STREAM = false>
  inline void const *colpass_entry()
  {
    if constexpr (col_pairs_lds<N>())
      return reinterpret_cast<void const*>(&paired::ocean_colpass_kernel<N, H16, STREAM>);
    else
      return reinterpret_cast<void const*>(&ocean_colpass_kernel<N, H16, STREAM>);
  }

  // blocked packed spectrum -> two row-major complex planes (datum_ocean_debug_rowpass)
  template<bool H16>
  __global__ void ocean_unpack_kernel(void const *spec, int N, float specinv, cf *c, cf *d)
  {
    size_t const plane = (size_t)N * N;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int y = (int)(i / N), x = (int)(i % N);

      size_t const at = blocked_at(N, y, x, H16);

      float4 v;

      if constexpr (H16)
      {
        half4_ hv = static_cast<ch const*>(spec)[at].v;
        v = make_float4((float)hv.x * specinv, (float)hv.y * specinv, (float)hv.z * specinv, (float)hv.w * specinv);
      }
      else
        v = static_cast<cd const*>(spec)[at];

      c[i] = cf{ v.x, v.y };
      d[i] = cf{ v.z, v.w };
    }
  }

  // h0 as two halves per point, times a power of two (DATUM_OCEAN_SPECTRUM_FP16_H0; re-run when h0 changes, like the reduction below)
  __global__ void ocean_h0half_kernel(float2 const *h0, unsigned int *h0h, size_t count, float scale)
  {
    typedef _Float16 half2_ __attribute__((ext_vector_type(2)));

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    {
      float2 const v = h0[i];
      half2_ const h = { (_Float16)(v.x * scale), (_Float16)(v.y * scale) };      // round to nearest even

      h0h[i] = __builtin_bit_cast(unsigned int, h);
    }
  }

  // largest |component| of a cascade's h0 (bit pattern of a non-negative float orders like the float): sizes the
  // fp16 spectrum's scale
  __global__ void ocean_absmax_kernel(float2 const *h0, size_t count, unsigned int *result)
  {
    float m = 0.0f;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    {
      float2 v = h0[i];
      float a = fmaxf(fabsf(v.x), fabsf(v.y));

      // a NaN must not hide -- and fmaxf drops one: each component is looked at by itself (round 6: a NaN in ONE component of a point went
      // through the fp16 formats unnoticed until then, tests/test_gpu_parity.py::test_fp16_formats_refuse_an_h0_that_is_not_finite)
      m = (v.x == v.x && v.y == v.y) ? fmaxf(m, a) : __builtin_inff();
    }

    for(int o = 32; o > 0; o >>= 1)
      m = fmaxf(m, __shfl_xor(m, o));

    if ((threadIdx.x & 63) == 0)
      atomicMax(result, __float_as_uint(m));
  }

  //|---------------------- all-gather payload ---------------------------------
  // What a rank sends to its peers when the tiles of a farm are reassembled: the displacement layer
  // (dx, dy, dz) of every cascade, row-major [cascade][y][x], as three floats (12 B per point) or four halves
  // (dx, dy, dz, 0: 8 B per point).  One thread per texel IN THE ORDER OF THE MAP LAYOUT: consecutive lanes read consecutive 16-byte
  // parts A (patch after patch, the 128-byte parts B skipped: whole lines, every byte of them used by the wave that fetched them) and
  // write 12 or 8 bytes each into runs of PW x (lanes / 16) texels per row.  Round 5: 39.7 -> 18.5 us for xyz32 at 1024^2 x 4 (6.35 TB/s), 169 -> 97 us at x 16
  // (up to round 4 a thread read the four texels of a patch row -- every line fetched by two far-apart waves -- and wrote 48 bytes
  // at a 48-byte stride: tools/dbg/pack_time.py).
  struct PackShape
  {
    int n2;                 // log2 N
    int pw2;                // log2 of a patch's width
    int bp2;                // log2 of the patches in a row of patches of one band (B / PW)
    int bandpatches2;       // log2 of the patches per band ((N / PH) * (B / PW))
    int b2;                 // log2 of the band's columns
  };

  template<bool HALF>
  __global__ void __launch_bounds__(256) ocean_pack_kernel(float4 const *maps, int N, int cascades, void *payload, PackShape sh)
  {
    size_t const plane = (size_t)N * N;
    size_t const total = (size_t)cascades * plane;
    size_t const stride = (size_t)gridDim.x * blockDim.x;

    constexpr int U = 4;                           // texels in flight per thread

    for(size_t q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < total; q0 += U * stride)
    {
      float4 v[U];

      #pragma unroll
      for(int k = 0; k < U; ++k)
      {
        size_t const q = q0 + k * stride;

        if (q < total)
        {
          size_t const c = q >> (2 * sh.n2), r = q & (plane - 1);

          v[k] = *reinterpret_cast<float4 const*>(reinterpret_cast<char const*>(maps) + c * map_cascade_bytes(N) + (r >> 4) * MAP_PATCH_BYTES + (r & 15) * 16);
        }
      }

      #pragma unroll
      for(int k = 0; k < U; ++k)
      {
        size_t const q = q0 + k * stride;

        if (q < total)
        {
          size_t const c = q >> (2 * sh.n2), r = q & (plane - 1);

          // the texel of part A number j of patch P (map_compact_patch / map_compact_j read backwards)
          int const P = (int)(r >> 4), j = (int)(r & 15);
          int const band = P >> sh.bandpatches2, pp = P & ((1 << sh.bandpatches2) - 1);
          int const y = ((pp >> sh.bp2) << (4 - sh.pw2)) + (j >> sh.pw2);
          int const x = (band << sh.b2) + ((pp & ((1 << sh.bp2) - 1)) << sh.pw2) + (j & ((1 << sh.pw2) - 1));

          size_t const at = c * plane + ((size_t)y << sh.n2) + x;

          if constexpr (HALF)
            static_cast<half4_*>(payload)[at] = half4_{ (_Float16)v[k].x, (_Float16)v[k].y, (_Float16)v[k].z, (_Float16)0.0f };
          else
          {
            struct xyz { float x, y, z; };

            static_cast<xyz*>(payload)[at] = xyz{ v[k].x, v[k].y, v[k].z };
          }
        }
      }
    }
  }

  //|---------------------- spectrum rebuild (lerp_ocean_waves) ----------------

  // phillips(k, a, v, w) of ocean.cpp:89-107, same operation order
  __device__ __forceinline__ float phillips_at(float kx, float ky, float a, float v, float wx, float wy)
  {
    if (kx == 0.0f && ky == 0.0f)
      return 0.0f;

    float kdotw = kx * wx + ky * wy;
    float d = (kdotw < 0.0f) ? 0.2f : 1.0f;

    float L = v * v / 9.81f;
    float L2 = L * L;
    float l2 = L2 * 0.001f * 0.001f;

    float k2 = kx * kx + ky * ky;

    return a * d * expf(-1.0f / (k2 * L2)) / (k2 * k2 * k2) * (kdotw * kdotw) * expf(-k2 * l2);
  }

  // h0 = seed * dk * sqrt(phillips / 2)   (ocean.cpp:196-209)
  __global__ void ocean_height_kernel(float2 const *seed, float2 *h0, int N, float wavescale, float waveamplitude, float windspeed, float windx, float windy)
  {
    size_t const plane = (size_t)N * N;

    float const dk = 6.2831855f / wavescale;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int m = (int)(i / N), n = (int)(i % N);

      float y = dk * ((float)m - 0.5f * (float)N);
      float x = dk * ((float)n - 0.5f * (float)N);

      float amp = dk * sqrtf(phillips_at(x, y, waveamplitude, windspeed, windx, windy) / 2.0f);

      float2 s = seed[i];

      h0[i] = make_float2(s.x * amp, s.y * amp);
    }
  }

  //|---------------------- phase-only advance --------------------------------
  // used when more than MAX_PENDING updates are queued between two displace calls

  __global__ void ocean_advance_kernel(StepArgs a, int N)
  {
    int const cascade = blockIdx.y;
    size_t const plane = (size_t)N * N;

    float *phase = a.phase + cascade * plane;
    float const *table = a.omega + (size_t)cascade * (N / 2 + 1) * (N / 2 + 1);

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int m = (int)(i / N), n = (int)(i % N);

      float omega = dispersion_lookup(table, n, m, N);

      float p = phase[i];
      for(int k = 0; k < a.ndt; ++k)
        p = advance_phase(p, omega * a.dt[k]);

      phase[i] = p;
    }
  }

  //|---------------------- diagnostics ---------------------------------------

  // ocean.sim alone, row-major output (datum_ocean_debug_sim): h~ by the same function the row pass uses
  // (sim_height_products), so that the stage test pins the product's arithmetic, not a sibling of it
  template<bool WILD>
  __global__ void ocean_sim_kernel(StepArgs a, int N, int cascade, cf *h, cf *hx, cf *hy)
  {
    size_t const plane = (size_t)N * N;

    float2 const *h0 = a.h0 + cascade * plane;
    float const *phase = a.phase + cascade * plane;
    float const scale = a.casc[cascade].scale;

    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
    {
      int y = (int)(i / N), x = (int)(i % N);

      cf hh = sim_height_products<WILD>(h0[i], h0[(size_t)(N - 1 - y) * N + (N - 1 - x)], phase[i]);
      float2 kn = knorm_of(wavevector(x, N, scale), wavevector(y, N, scale));

      h[i] = hh;
      hx[i] = cf{ hh.y * kn.x, -hh.x * kn.x };
      hy[i] = cf{ hh.y * kn.y, -hh.x * kn.y };
    }
  }

}
