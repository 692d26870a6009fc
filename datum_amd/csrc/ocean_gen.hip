// ocean_gen.hip -- ocean.gen (data/ocean.gen.comp:67-137): the projected-grid mesh from the displacement map.
//
// The kernel is bound by the vector ALU (profiles/r02_gen_experiments.txt: 419 instructions per vertex, the units busy
// for 11 of 18 us), so it is built around gfx950's PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32:
// two floats per lane per issue): every thread owns TWO vertices of a mesh row, 16 columns apart, and carries them
// through the shader as the two halves of 64-bit register pairs (`v2`).  What cannot be packed (reciprocals, square roots,
// exp / log, floor, conversions, the texel addressing, the fetches) is issued per vertex.
//
//   * the view ray, the plane hit and the swell phase (gen.comp:81-99) keep the shader's operation order with IEEE-exact
//     results -- near the horizon the plane hit amplifies one ulp of the ray by dist / costheta and the swell phase turns
//     metres into radians, so the oracle is only reproducible there with the same roundings.  Exact does not mean the
//     compiler's division sequence: the operands are in the normal range, so the range scaling and the fix-up are left
//     out, and the reciprocal of a shared denominator (|viewvec|) is refined once (div_exact, sqrt_exact: the same
//     refinement steps as the IEEE expansion, packed);
//   * what does not depend on the vertex is evaluated once on the host in the shader's operation order (GenFrame) --
//     gfx950 has no scalar float unit, a uniform product costs a vector instruction in every wave;
//   * the shading frame (gen.comp:101-120) uses v_rsq_f32 / v_exp_f32 / v_log_f32 and FMAs: errors are not amplified there;
//   * sin / cos of the swell phase (up to 1e5..1e6 at the horizon) by the two-constant Cody-Waite step of sincos_phase;
//   * texture(sampler2DArray) (gen.comp:113-114) is a manual bilinear REPEAT fetch from the module's own map layout
//     (ocean_kernels.hip: map_compact_a / map_compact_b) in fp32 with float weights (lavapipe-style exact bilinear).  The texel index wraps
//     through v_fract_f32 of coordinate / N (exact: N is a power of two), right for every float the oracle's 64-bit wrap
//     is right for -- an int32 conversion saturates from |coordinate| = 2^31 (dist = 1e6 at wavescale < 2, N = 4096);
//   * where a bilinear weight is exactly 0 along an axis (beyond |coordinate| = 2^23 texels, i.e. every ray above the
//     horizon) the second texel of that axis is not fetched: its offset is pushed out of the buffer's range (zeros come
//     back without a memory access), so no branch -- and no wait -- separates the fetches;
//   * the normal layer is fetched only by waves with a vertex whose distance smoothing (gen.comp:116) is not exactly 1
//     (elsewhere the blended normal is 0 * finite + plane normal), the Gerstner frame likewise;
//   * the 48-byte Mesh::Vertex'es of a wave (4 rows x 32 vertices) go through LDS so that every store instruction writes 16
//     contiguous bytes per lane, 1.5 KB per mesh row.
// Measured and not kept (profiles/r02_gen_experiments.txt): maps of N <= 64 copied into LDS, a persistent loop over the
// tiles with the next tile's ray arithmetic under the current tile's fetches, alternating tile order, priority classes.

#pragma once

#include "ocean_kernels.hip"

namespace ocean
{
  typedef float v2 __attribute__((ext_vector_type(2)));      // one value of a thread's two vertices

  // per-launch constants of ocean.gen that do not depend on the vertex (gen.comp:75-79,93-105), evaluated once on the
  // host in the shader's operation order instead of once per thread
  struct GenFrame
  {
    float camerapos[3];
    float cameraheight;
    float margin;
    float frequency;
    float qi;
    float phi;
    float sxm1, sym1;        // float(sizex - 1), float(sizey - 1)
    float viewz[3];          // invproj[2] * 0.0, [6] * 0.0, [10] * 0.0   (the third term of each viewvec row, gen.comp:84)
    float vieww[3];          // invproj[3] * 1.0, [7] * 1.0, [11] * 1.0
    float negplane[3];       // -plane.xyz
    float basez;             // -plane.w
    float gx, gy;            // (qi * amplitude) * direction   (gen.comp:101)
    float nx, ny, nz;        // Gerstner normal  = (nx ct, ny ct, nz st)   (gen.comp:103)
    float tx, ty, tz;        // Gerstner tangent = (tx st, ty st, tz ct)   (gen.comp:104)
    float fn, rfn;           // float(N), 1 / float(N)
  };

  struct GenArgs
  {
    datum_ocean_set set;
    GenFrame frame;
    float4 const *map;     // the cascade's displacement map, map_cascade_bytes(N) bytes (ocean_kernels.hip: map_compact_a / map_compact_b)
    int N;
    int sizex;
    int sizey;
    int tilesx;
    int tiles;
    int chunk;             // tiles per XCD chunk, 0: tiles in launch order
    int block0;            // index of this launch's first workgroup in the whole mesh's list (a mesh may be launched in two parts)
    float *vertices;
#ifdef OCEAN_STAMPS
    unsigned long long *stamps;   // diagnostic builds only (tools/dbg/genstamps.hip): [workgroup][16] timestamps
#endif
  };

  // a workgroup's tile is 32 x 16 vertices, a wave owns 32 x 4 of them (a "wave tile"), a thread (x, y) and (x + 16, y).
  // Measured and not kept (profiles/r03_gen_experiments.txt, r04_gen_levers.txt): 128 ... 1024 threads per workgroup, two sets of
  // rows per wave as a software pipeline, forced occupancies through unused LDS.
  constexpr int GEN_TILE_X = 32;
  constexpr int GEN_THREADS = 256;
  constexpr int GEN_TILE_Y = 4 * (GEN_THREADS / 64);
  constexpr size_t GEN_LDS = (size_t)GEN_THREADS * 2 * 3 * sizeof(float4);

  inline GenFrame make_gen_frame(datum_ocean_set const &p, int N, int sizex, int sizey)
  {
    GenFrame f;

    // camerapos = 2 * (dual * conjugate(real)).yzw   (gen.comp:75, transform.inc:13-28)
    float rw = p.camera_real[0], ri = -p.camera_real[1], rj = -p.camera_real[2], rk = -p.camera_real[3];
    float dw = p.camera_dual[0], di = p.camera_dual[1], dj = p.camera_dual[2], dk = p.camera_dual[3];

    f.camerapos[0] = 2 * (dw * ri + di * rw + dj * rk - dk * rj);
    f.camerapos[1] = 2 * (dw * rj + dj * rw + dk * ri - di * rk);
    f.camerapos[2] = 2 * (dw * rk + dk * rw + di * rj - dj * ri);

    f.cameraheight = (p.plane[0] * f.camerapos[0] + p.plane[1] * f.camerapos[1] + p.plane[2] * f.camerapos[2]) + p.plane[3];
    f.margin = 1 + sqrtf((2 * p.swellamplitude + 0.5f) / f.cameraheight);

    f.sxm1 = (float)(sizex - 1);
    f.sym1 = (float)(sizey - 1);

    for(int r = 0; r < 3; ++r)
    {
      f.viewz[r] = p.invproj[4 * r + 2] * 0.0f;
      f.vieww[r] = p.invproj[4 * r + 3] * 1.0f;
      f.negplane[r] = -p.plane[r];
    }

    f.basez = -p.plane[3];

    // Gerstner swell constants (gen.comp:93-105)
    f.frequency = 2 * 3.14159265358979323846f / p.swelllength;
    f.qi = p.swellsteepness / (f.frequency * p.swellamplitude * 4 + 1e-6f);
    f.phi = f.frequency * p.swellamplitude;

    float const dirx = p.swelldirection[0], diry = p.swelldirection[1];

    f.gx = f.qi * p.swellamplitude * dirx;
    f.gy = f.qi * p.swellamplitude * diry;

    f.nx = f.phi * dirx / 6;
    f.ny = f.phi * diry / 6;
    f.nz = f.qi * f.phi;
    f.tx = f.qi * f.phi * dirx * dirx;
    f.ty = f.qi * f.phi * diry * dirx;
    f.tz = f.phi * dirx / 6;

    f.fn = (float)N;
    f.rfn = 1.0f / (float)N;

    return f;
  }

  //|---------------------- packed helpers --------------------------------------

  struct p3 { v2 x, y, z; };                      // a 3-vector of each of the two vertices

  __device__ __forceinline__ v2 splat(float a) { return v2{ a, a }; }
  __device__ __forceinline__ v2 pfma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
  __device__ __forceinline__ v2 pfma(float a, v2 b, v2 c) { return __builtin_elementwise_fma(splat(a), b, c); }
  __device__ __forceinline__ v2 pfma(v2 a, v2 b, float c) { return __builtin_elementwise_fma(a, b, splat(c)); }
  __device__ __forceinline__ v2 pfma(float a, v2 b, float c) { return __builtin_elementwise_fma(splat(a), b, splat(c)); }
  __device__ __forceinline__ v2 prcp(v2 a) { return v2{ __builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y) }; }
  __device__ __forceinline__ v2 prsq(v2 a) { return v2{ __builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y) }; }
  __device__ __forceinline__ v2 pfloor(v2 a) { return v2{ __builtin_floorf(a.x), __builtin_floorf(a.y) }; }

  // (with -ffp-contract=off a product and a sum stay v_pk_mul_f32 + v_pk_add_f32: two IEEE roundings, the oracle's)
  __device__ __forceinline__ v2 dot3(p3 a, p3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

  // IEEE-exact a / b for operands in the normal range: the refinement steps of the compiler's own fp32 division
  // (v_rcp_f32, two Newton steps on the reciprocal, quotient + two residual corrections) without v_div_scale / v_div_fixup.
  // `y` is the refined reciprocal of b: shared by the three components of a normalisation.
  __device__ __forceinline__ v2 refined_rcp(v2 b)
  {
    v2 y = prcp(b);
    v2 e = pfma(-b, y, 1.0f);

    return pfma(e, y, y);
  }

  __device__ __forceinline__ v2 div_exact(v2 a, v2 b, v2 y)
  {
    v2 q = a * y;
    v2 r = pfma(-b, q, a);
    q = pfma(r, y, q);
    r = pfma(-b, q, a);

    return pfma(r, y, q);
  }

  __device__ __forceinline__ v2 div_exact(v2 a, v2 b) { return div_exact(a, b, refined_rcp(b)); }

  // IEEE-exact sqrt(a) for a in the normal range: v_rsq_f32 and the Goldschmidt steps of the compiler's fp32 square root
  __device__ __forceinline__ v2 sqrt_exact(v2 a)
  {
    v2 rs = prsq(a);
    v2 s = a * rs;
    v2 h = rs * 0.5f;
    v2 e = pfma(-h, s, 0.5f);

    h = pfma(h, e, h);
    s = pfma(s, e, s);

    v2 d = pfma(-s, s, a);

    return pfma(d, h, s);
  }

  // a / |a| with the hardware reciprocal square root (1 ulp): the shading frame, where errors are not amplified
  __device__ __forceinline__ p3 normalize3(p3 a)
  {
    v2 inv = prsq(pfma(a.z, a.z, pfma(a.y, a.y, a.x * a.x)));

    return { a.x * inv, a.y * inv, a.z * inv };
  }

  // sincos_phase (ocean_kernels.hip) of two arguments: the same reduction and polynomials, packed.  Always the software form
  // here: the swell phase reaches 1e5..1e6 at the horizon, far outside v_sin_f32's domain
  __device__ __forceinline__ void sincos_phase2(v2 x, v2 &sn, v2 &cs)
  {
    v2 t = x * 0.636619772367581343f;                            // x * 2/pi
    v2 k = { __builtin_rintf(t.x), __builtin_rintf(t.y) };

    v2 r = pfma(k, splat(-1.57079637050628662109375f), x);       // pi/2 head
    r = pfma(k, splat(4.37113900018624283e-8f), r);              // pi/2 tail

    v2 z = r * r;

    v2 sp = pfma(pfma(pfma(splat(-1.9515295891e-4f), z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    v2 cp = pfma(pfma(pfma(splat(2.443315711809948e-5f), z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, pfma(z, splat(-0.5f), 1.0f));

    #pragma unroll
    for(int i = 0; i < 2; ++i)
    {
      int q = (int)k[i];

      float s = (q & 1) ? cp[i] : sp[i];
      float c = (q & 1) ? sp[i] : cp[i];

      sn[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) ^ (((unsigned)q << 30) & 0x80000000u));
      cs[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) ^ (((unsigned)(q + 1) << 30) & 0x80000000u));
    }
  }

  // rotate v by the unit quaternion q = (w, x, y, z)   (data/transform.inc:32-37), the shader's operation order
  __device__ __forceinline__ p3 rotate(float const (&q)[4], p3 v)
  {
    float const ux = q[1], uy = q[2], uz = q[3];

    p3 tt = { 2.0f * (uy * v.z - uz * v.y), 2.0f * (uz * v.x - ux * v.z), 2.0f * (ux * v.y - uy * v.x) };

    return { (v.x + q[0] * tt.x) + (uy * tt.z - uz * tt.y), (v.y + q[0] * tt.y) + (uz * tt.x - ux * tt.z), (v.z + q[0] * tt.z) + (ux * tt.y - uy * tt.x) };
  }

  // BYTE offset of a texel column's / row's part of the map layout (ocean_kernels.hip); the two add up to the texel's
  // displacement: part A of its patch, (dx, dy, dz, nx) -- whose part B, (ny, nz), lies at A + 256 - bcolumn(i) - brow(j).
  //   PLAIN   N <= 1024: whole rows
  //   BANDED  2048 and 4096: bands of band_cols(N) columns
  enum GenLayout { GEN_PLAIN = 0, GEN_BANDED = 1 };

  template<int LAYOUT> struct TexelIndex
  {
    int ln, lb, bmask;
    int lpw, lph;            // log2 of the patch's columns and rows

    __device__ __forceinline__ TexelIndex(int N) : ln(31 - __builtin_clz(N)), lb(31 - __builtin_clz(band_cols(N))), bmask(band_cols(N) - 1),
                                                   lpw(31 - __builtin_clz(map_patch_cols(N))), lph(31 - __builtin_clz(map_patch_rows(N))) { }

    __device__ __forceinline__ int column(int i) const
    {
      int const inband = ((i & bmask) >> lpw) * MAP_PATCH_BYTES + ((i & ((1 << lpw) - 1)) << 4);

      if constexpr (LAYOUT == GEN_PLAIN)
        return inband;
      else
        return (i >> lb) * (3 << (3 + ln + lb)) + inband;           // 24 N B bytes per band
    }

    __device__ __forceinline__ int row(int j) const
    {
      return (j >> lph) * (3 << (7 + lb - lpw)) + ((j & ((1 << lph) - 1)) << (4 + lpw));     // 384 B / PW bytes per patch row
    }

    // 8 * (the texel's index in its patch), column and row part
    __device__ __forceinline__ int bcolumn(int i) const { return (i & ((1 << lpw) - 1)) << 3; }
    __device__ __forceinline__ int brow(int j) const { return (j & ((1 << lph) - 1)) << (3 + lpw); }
  };

  //|---------------------- the kernel ------------------------------------------
  // (One function, local arrays: with the three stages as functions over structs, or inside a loop over tiles, hipcc keeps
  // 134-154 registers instead of 86 and spills -- measured 17.7 against 16.3 us; a persistent loop, 3 or 4 workgroups
  // per CU: 17.1 / 18.1 us; a 1024-thread workgroup per CU sampling a 64^2 map from LDS: 24.5 us.  profiles/r03_gen_experiments.txt)

  // The vertex stream (50 MB per 1024 x 1024 mesh, written once, read by the graphics queue) is stored non-temporally: 26.1 ->
  // 25.5 us from 1024^2 maps, nothing lost from 64^2 maps; the other policies measured in profiles/r04_gen_levers.txt.  (The
  // compiler's own builtin: as inline asm the store needs the wait states of "VALU write of a VGPR that holds the data of a VMEM
  // store wider than 64 bits" written out -- hipcc's hazard recognizer does not look into inline asm, and without them 3 in 10 000
  // floats of a 1024 x 1024 mesh came out as the NEXT store's address arithmetic: tools/dbg/gen_compare.py.)
  __device__ __forceinline__ void store_vertex_float4(float4 *at, float4 v)
  {
    typedef float f4_ __attribute__((ext_vector_type(4)));

    f4_ const d = { v.x, v.y, v.z, v.w };

    __builtin_nontemporal_store(d, reinterpret_cast<f4_*>(at));
  }

  // a corner's second fetch, (ny, nz); the normal's x rides in the first one's .w
  struct GenNormal
  {
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t rmap, int compact_offset) { return buf_load_f32x2(rmap, compact_offset, 0); }
    static __device__ __forceinline__ float x(float4 a, float2) { return a.w; }
    static __device__ __forceinline__ float y(float4, float2 b) { return b.x; }
    static __device__ __forceinline__ float z(float4, float2 b) { return b.y; }
  };

  typedef float2 GenNormalFetch;

  // (compiled for five waves per SIMD, 96 registers: 101 without the bound, one wave per SIMD fewer)
  template<int LAYOUT>
  __attribute__((amdgpu_waves_per_eu(5, 5)))
  __global__ void __launch_bounds__(GEN_THREADS) ocean_gen_kernel(GenArgs g)
  {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = g.stamps + (size_t)blockIdx.x * 16;
#endif
    OCEAN_STAMP(0);

    constexpr int PH = 1;       // sets of four rows per wave (two, software-pipelined, measured slower)

    datum_ocean_set const &p = g.set;
    GenFrame const &f = g.frame;

    int const tid = threadIdx.x;
    int const lane = tid & 63;
    int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Workgroup b runs on XCD b % 8.  A map that does not fit an XCD's 4 MB L2 is sampled in chunks of whole tile rows
    // dealt to the XCDs in turn: neighbouring tiles share an L2 (1024^2 maps: 36.4 -> 31.0 us), and every XCD gets its
    // share of the cheap rows above the horizon (one contiguous run of tiles per XCD: 43 us).  Small maps sit in every
    // L2 anyway; there the launch order is kept (64^2 maps: 16.5 against 17.1 us).
    int tile = (int)blockIdx.x + g.block0;

    if (g.chunk)
    {
      int const slot = tile >> 3;

      tile = ((slot / g.chunk) * 8 + (tile & 7)) * g.chunk + slot % g.chunk;

      if (tile >= g.tiles)
        return;
    }

    int const tilex = tile % g.tilesx, tiley = tile / g.tilesx;

    int const x0 = tilex * GEN_TILE_X;
    int const ywave = tiley * GEN_TILE_Y + 4 * PH * wave;          // first of this wave's 4 * PH rows

    int const xa = x0 + (lane & 15);

    float const *ip = p.invproj;

    TexelIndex<LAYOUT> const texel(g.N);

    int const nmask = g.N - 1;

    __amdgpu_buffer_rsrc_t const rmap = make_rsrc(g.map, map_cascade_bytes(g.N));

    // per phase (PH = 2: two sets of four rows per wave; the second set's ray arithmetic runs under the first set's fetches,
    // the first set's shading under the second set's fetches)
    p3 position[PH];
    v2 w00[PH], w10[PH], w01[PH], w11[PH], smoothing[PH], st[PH], ct[PH];
    int o00[PH][2], o10[PH][2], o01[PH][2], o11[PH][2];       // byte offsets of the four corners' displacement texels
    bool shaded[PH], near[PH];
    float4 a00[PH][2], a10[PH][2], a01[PH][2], a11[PH][2];      // displacement layer
    GenNormalFetch b00[PH][2], b10[PH][2], b01[PH][2], b11[PH][2];      // normal layer (compact: its y and z; x came with the displacement)
    int q00[PH][2], q10[PH][2], q01[PH][2], q11[PH][2];       // compact: byte offsets of the four corners' (ny, nz)

    #pragma unroll
    for(int ph = 0; ph < PH; ++ph)
    {
      int const yy = ywave + 4 * ph + (lane >> 4);

      //-- view ray, plane hit, swell phase: the shader's expressions and roundings (gen.comp:81-99) ----------------

      v2 const xx = { (float)xa, (float)(xa + 16) };

      v2 const u = (div_exact(2.0f * xx, splat(f.sxm1)) - 1.0f) * f.margin;

      // (one row per thread: its v is the first half of a packed division whose second half repeats it)
      float const v = ((1.0f - div_exact(splat(2.0f * (float)yy), splat(f.sym1))) * f.margin).x;

      p3 viewvec = { ((ip[0] * u + ip[1] * v) + f.viewz[0]) + f.vieww[0],
                     ((ip[4] * u + ip[5] * v) + f.viewz[1]) + f.vieww[1],
                     ((ip[8] * u + ip[9] * v) + f.viewz[2]) + f.vieww[2] };

      v2 const len = sqrt_exact(dot3(viewvec, viewvec));
      v2 const rlen = refined_rcp(len);

      p3 const worlddir = rotate(p.camera_real, p3{ div_exact(viewvec.x, len, rlen), div_exact(viewvec.y, len, rlen), div_exact(viewvec.z, len, rlen) });

      v2 const costheta = worlddir.x * f.negplane[0] + worlddir.y * f.negplane[1] + worlddir.z * f.negplane[2];

      v2 const hit = div_exact(splat(f.cameraheight), costheta);

      v2 const dist = { (costheta.x > 0) ? hit.x : 1e6f, (costheta.y > 0) ? hit.y : 1e6f };

      v2 const basex = f.camerapos[0] + dist * worlddir.x;
      v2 const basey = f.camerapos[1] + dist * worlddir.y;

      v2 const theta = f.frequency * (p.swelldirection[0] * basex + p.swelldirection[1] * basey) + p.swellphase;

      sincos_phase2(theta, st[ph], ct[ph]);

      position[ph] = { basex + f.gx * ct[ph], basey + f.gy * ct[ph], f.basez + p.swellamplitude * st[ph] };

      v2 cl = dist * p.smoothing - 0.35f;

      #pragma unroll
      for(int i = 0; i < 2; ++i)       // pow(clamp(cl, 0, 1), 0.2): 0 -> 0, 1 -> 1 exactly
        smoothing[ph][i] = __builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf(__builtin_amdgcn_fmed3f(cl[i], 0.0f, 1.0f)));

      //-- texture(sampler2DArray, REPEAT, linear, lod 0) of both layers at (position.xy * scale): texel centres at (i + 0.5) / N ----

      v2 const fx = (position[ph].x * p.scale) * f.fn - 0.5f;
      v2 const fy = (position[ph].y * p.scale) * f.fn - 0.5f;

      v2 const flx = pfloor(fx), fly = pfloor(fy);

      v2 const ax = fx - flx, ay = fy - fly;

      // floor(coordinate) mod N: N is a power of two, so coordinate / N, its fractional part and the product with N are exact
      v2 const wx = flx * f.rfn, wy = fly * f.rfn;
      v2 const mx = v2{ __builtin_amdgcn_fractf(wx.x), __builtin_amdgcn_fractf(wx.y) } * f.fn;
      v2 const my = v2{ __builtin_amdgcn_fractf(wy.x), __builtin_amdgcn_fractf(wy.y) } * f.fn;

      v2 const bx = 1.0f - ax, by = 1.0f - ay;

      w00[ph] = bx * by; w10[ph] = ax * by; w01[ph] = bx * ay; w11[ph] = ax * ay;

      near[ph] = false;                        // some weight other than w00 is not zero

      #pragma unroll
      for(int i = 0; i < 2; ++i)
      {
        int const i0 = (int)mx[i], j0 = (int)my[i];

        int const i1 = (i0 + 1) & nmask, j1 = (j0 + 1) & nmask;

        int const c0 = texel.column(i0), c1 = texel.column(i1);
        int const r0 = texel.row(j0), r1 = texel.row(j1);

        // A zero weight along an axis (beyond |coordinate| = 2^23 texels: every ray above the horizon): the second texel of
        // that axis is not needed (0 * finite adds nothing).  Its offset is pushed out of the buffer's range: zeros come
        // back without a memory access, and no branch -- hence no wait -- separates the fetches.
        bool const wantx = ax[i] != 0.0f, wanty = ay[i] != 0.0f;

        o00[ph][i] = r0 + c0; o10[ph][i] = wantx ? r0 + c1 : -256; o01[ph][i] = wanty ? r1 + c0 : -256; o11[ph][i] = (wantx && wanty) ? r1 + c1 : -256;

        int const bc0 = texel.bcolumn(i0), bc1 = texel.bcolumn(i1);
        int const br0 = 256 - texel.brow(j0), br1 = 256 - texel.brow(j1);

        q00[ph][i] = o00[ph][i] + br0 - bc0; q10[ph][i] = wantx ? o10[ph][i] + br0 - bc1 : -256; q01[ph][i] = wanty ? o01[ph][i] + br1 - bc0 : -256; q11[ph][i] = (wantx && wanty) ? o11[ph][i] + br1 - bc1 : -256;

        near[ph] = near[ph] || wantx || wanty;
      }

      near[ph] = __builtin_amdgcn_ballot_w64(near[ph]) != 0;

      OCEAN_STAMP(1);

      // wave-uniform: only a wave with a vertex inside the smoothing distance needs the normal layer and the Gerstner frame
      shaded[ph] = __builtin_amdgcn_ballot_w64(smoothing[ph].x != 1.0f || smoothing[ph].y != 1.0f) != 0;

      // (a corner's normal sits in the 128-byte line of its displacement: fetched right behind it, it finds the line in L1 --
      // eight displacement fetches of 64 lanes later the line may have left the cache again)
      #define OCEAN_GEN_FETCH_A(C) a##C[ph][i] = buf_load_f32x4_aux<0>(rmap, o##C[ph][i], 0)
      #define OCEAN_GEN_FETCH_B(C) b##C[ph][i] = GenNormal::load(rmap, q##C[ph][i])

      if (shaded[ph])
      {
        #pragma unroll
        for(int i = 0; i < 2; ++i)
        {
          OCEAN_GEN_FETCH_A(00); OCEAN_GEN_FETCH_B(00);
          OCEAN_GEN_FETCH_A(10); OCEAN_GEN_FETCH_B(10);
          OCEAN_GEN_FETCH_A(01); OCEAN_GEN_FETCH_B(01);
          OCEAN_GEN_FETCH_A(11); OCEAN_GEN_FETCH_B(11);
        }
      }
      else if (near[ph])
      {
        #pragma unroll
        for(int i = 0; i < 2; ++i)
        {
          OCEAN_GEN_FETCH_A(00);
          OCEAN_GEN_FETCH_A(10);
          OCEAN_GEN_FETCH_A(01);
          OCEAN_GEN_FETCH_A(11);
        }
      }
      else
      {
        // every ray of the wave lands beyond |coordinate| = 2^23 texels along both axes (the rays above the horizon): one
        // texel per vertex, its weight (1 - 0) * (1 - 0); the other three fetches would only occupy the texture path
        #pragma unroll
        for(int i = 0; i < 2; ++i)
          OCEAN_GEN_FETCH_A(00);
      }

      #undef OCEAN_GEN_FETCH_A
      #undef OCEAN_GEN_FETCH_B

      if (PH > 1)
        __builtin_amdgcn_sched_barrier(0);       // the next set's arithmetic stays BEHIND these fetches
    }

    OCEAN_WAIT_LOADS();
    OCEAN_STAMP(2);

    float4 *mine = reinterpret_cast<float4*>(smem) + 384 * wave;

    #pragma unroll
    for(int ph = 0; ph < PH; ++ph)
    {
      //-- bilinear blend and shading frame (gen.comp:101-120), FMAs ------------------------------------------------

      #define OCEAN_GEN_BLEND(T, C) pfma(w11[ph], v2{ T##11[ph][0].C, T##11[ph][1].C }, pfma(w01[ph], v2{ T##01[ph][0].C, T##01[ph][1].C }, pfma(w10[ph], v2{ T##10[ph][0].C, T##10[ph][1].C }, w00[ph] * v2{ T##00[ph][0].C, T##00[ph][1].C })))

      p3 displacement;

      if (shaded[ph] || near[ph])
        displacement = { OCEAN_GEN_BLEND(a, x), OCEAN_GEN_BLEND(a, y), OCEAN_GEN_BLEND(a, z) };
      else
        displacement = { w00[ph] * v2{ a00[ph][0].x, a00[ph][1].x }, w00[ph] * v2{ a00[ph][0].y, a00[ph][1].y }, w00[ph] * v2{ a00[ph][0].z, a00[ph][1].z } };

      p3 const planen = { splat(p.plane[0]), splat(p.plane[1]), splat(p.plane[2]) };

      p3 tbn2;

      if (shaded[ph])
      {
        #define OCEAN_GEN_BLENDN(F) pfma(w11[ph], v2{ F(a11[ph][0], b11[ph][0]), F(a11[ph][1], b11[ph][1]) }, pfma(w01[ph], v2{ F(a01[ph][0], b01[ph][0]), F(a01[ph][1], b01[ph][1]) }, pfma(w10[ph], v2{ F(a10[ph][0], b10[ph][0]), F(a10[ph][1], b10[ph][1]) }, w00[ph] * v2{ F(a00[ph][0], b00[ph][0]), F(a00[ph][1], b00[ph][1]) })))

        p3 const dn = { OCEAN_GEN_BLENDN(GenNormal::x), OCEAN_GEN_BLENDN(GenNormal::y), OCEAN_GEN_BLENDN(GenNormal::z) };

        #undef OCEAN_GEN_BLENDN

        // tbn[2] = normalize(-normal.xy, 1 - normal.z), tbn[0] = normalize(1 - tangent.x, -tangent.y, tangent.z), tbn[1] = tbn[0] x tbn[2]
        p3 const t2 = normalize3(p3{ -f.nx * ct[ph], -f.ny * ct[ph], pfma(-f.nz, st[ph], 1.0f) });
        p3 const t0 = normalize3(p3{ pfma(-f.tx, st[ph], 1.0f), -f.ty * st[ph], f.tz * ct[ph] });
        p3 const t1 = { t0.y * t2.z - t0.z * t2.y, t0.z * t2.x - t0.x * t2.z, t0.x * t2.y - t0.y * t2.x };

        // tbn * displacementnormal, mixed towards the plane normal with the distance smoothing
        p3 const tn = { pfma(dn.z, t2.x, pfma(dn.y, t1.x, dn.x * t0.x)), pfma(dn.z, t2.y, pfma(dn.y, t1.y, dn.x * t0.y)), pfma(dn.z, t2.z, pfma(dn.y, t1.z, dn.x * t0.z)) };

        v2 const keep = 1.0f - smoothing[ph];

        tbn2 = normalize3(p3{ pfma(keep, tn.x, smoothing[ph] * planen.x), pfma(keep, tn.y, smoothing[ph] * planen.y), pfma(keep, tn.z, smoothing[ph] * planen.z) });
      }
      else
        tbn2 = normalize3(planen);

      #undef OCEAN_GEN_BLEND

      // tbn[0] = normalize((1, 0, 0) - tbn[2].x * tbn[2])
      p3 const tbn0 = normalize3(p3{ pfma(-tbn2.x, tbn2.x, 1.0f), -tbn2.x * tbn2.y, -tbn2.x * tbn2.z });

      OCEAN_STAMP(3);

      //-- Mesh::Vertex { position3, texcoord2, normal3, tangent4 } = 48 bytes (src/renderer/mesh.h:20-26) -----------
      // The wave's 4 rows x 32 vertices = 4 x 96 float4 go through its 6 KB of LDS: lane i then stores float4 number
      // i, 64 + i, ... 320 + i of the wave's 384 (three 16-byte stores per vertex at a 48-byte stride touch every line three times).

      v2 const px = position[ph].x - displacement.x, py = position[ph].y - displacement.y, pz = position[ph].z + displacement.z;
      v2 const tu = 0.1f * position[ph].x, tv = 0.1f * position[ph].y;

      #pragma unroll
      for(int i = 0; i < 2; ++i)
      {
        float4 *vtx = mine + 3 * ((lane >> 4) * 32 + 16 * i + (lane & 15));

        vtx[0] = make_float4(px[i], py[i], pz[i], tu[i]);
        vtx[1] = make_float4(tv[i], tbn2.x[i], tbn2.y[i], tbn2.z[i]);
        vtx[2] = make_float4(tbn0.x[i], tbn0.y[i], tbn0.z[i], -1.0f);
      }

      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      int const y0 = ywave + 4 * ph;
      int const rowlen = min(GEN_TILE_X, g.sizex - x0) * 3;                                // float4 of this tile in one mesh row

      // (the wave's part of the address is uniform: a scalar base and a 32-bit offset per lane)
      float4 *out = reinterpret_cast<float4*>(g.vertices) + ((size_t)y0 * g.sizex + x0) * 3;

      #pragma unroll
      for(int k = 0; k < 6; ++k)
      {
        int const j = 64 * k + lane;
        int const r = j / 96, c = j % 96;

        if (c < rowlen && y0 + r < g.sizey)
          store_vertex_float4(out + (unsigned)(r * g.sizex * 3 + c), mine[j]);
      }

    }

    OCEAN_STAMP_WHERE();
    OCEAN_STAMP(4);
  }

  inline GenLayout gen_layout(int N)
  {
    return (band_cols(N) != N) ? GEN_BANDED : GEN_PLAIN;
  }

  // everything but the set header, the map and the vertex buffer
  inline void gen_shape(GenArgs &g, int N, int sizex, int sizey)
  {
    g.frame = make_gen_frame(g.set, N, sizex, sizey);
    g.N = N;
    g.sizex = sizex;
    g.sizey = sizey;
    g.tilesx = (sizex + GEN_TILE_X - 1) / GEN_TILE_X;
    g.tiles = g.tilesx * ((sizey + GEN_TILE_Y - 1) / GEN_TILE_Y);
    g.chunk = (map_cascade_bytes(N) > ((size_t)4 << 20)) ? g.tilesx : 0;       // maps beyond an XCD's L2: chunks of one tile row are dealt to the XCDs in turn
    g.block0 = 0;
  }

  inline void const *gen_kernel_for(int N)
  {
    switch(gen_layout(N))
    {
      case GEN_PLAIN: return reinterpret_cast<void const*>(&ocean_gen_kernel<GEN_PLAIN>);
      default: return reinterpret_cast<void const*>(&ocean_gen_kernel<GEN_BANDED>);
    }
  }

  // workgroups of the whole mesh (with XCD chunks: rounded up to whole sets of eight chunks)
  inline int gen_groups(GenArgs const &g)
  {
    return g.chunk ? ((g.tiles + 8 * g.chunk - 1) / (8 * g.chunk)) * 8 * g.chunk : g.tiles;
  }

  // workgroups [first, first + count) of the mesh's list on `stream`
  inline hipError_t launch_gen_part(GenArgs &g, int first, int count, hipStream_t stream)
  {
    void *args[] = { &g };
    void const *kernel = gen_kernel_for(g.N);

    g.block0 = first;

    return hipLaunchKernel(kernel, dim3(count), dim3(GEN_THREADS), args, GEN_LDS, stream);
  }

  inline hipError_t launch_gen(GenArgs &g, hipStream_t stream)
  {
    return launch_gen_part(g, 0, gen_groups(g), stream);
  }
}
