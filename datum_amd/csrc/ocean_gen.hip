// ocean_gen.hip -- ocean.gen (data/ocean.gen.comp:67-137): the projected-grid mesh from the displacement map.
//
// One thread per mesh vertex, 16 x 16 vertices per tile.  What the reference does with two texture() fetches of a
// sampler2DArray (gen.comp:113-114) is a manual bilinear REPEAT fetch from the module's own map layout
// (ocean_kernels.hip: map_index), in fp32 with float weights (lavapipe-style exact bilinear).  Per vertex:
//   * a corner whose bilinear weight is exactly 0 is not fetched (0 * finite = 0 adds nothing): beyond |coordinate| =
//     2^23 texels -- every ray above the horizon, where dist = 1e6 -- the fractional parts vanish and one corner is left;
//   * the normal layer is not fetched where the distance smoothing (gen.comp:116) is exactly 1: the blended normal is
//     then the plane's (0 * finite + n), which is the whole upper half of the projected grid and the horizon band;
//   * the eight fetches of a vertex are buffer loads issued back to back; an unwanted corner's offset is pushed out of
//     the buffer's range (zeros come back without a memory access), so no branch -- and no wait -- separates them.
// (Measured and not kept, profiles/r02_gen_experiments.txt: maps of N <= 64 copied into LDS by persistent 512-thread
// workgroups -- 22 us against 18.5 us, one workgroup per CU cannot hide the arithmetic's latencies; a persistent loop over
// the tiles -- no gain and 100 instead of 68 VGPRs; alternating workgroups between the top and the bottom of the mesh --
// 2 us slower.)
// The 48-byte Mesh::Vertex'es of a wave go through LDS so that every store instruction writes 16 contiguous bytes per lane.

#pragma once

#include "ocean_kernels.hip"

namespace ocean
{
  // per-launch constants of ocean.gen that do not depend on the vertex (gen.comp:75-79,93-99), evaluated once on the
  // host in the shader's operation order instead of once per thread (gfx950 has no scalar float unit)
  struct GenFrame
  {
    float camerapos[3];
    float cameraheight;
    float margin;
    float frequency;
    float qi;
    float phi;
  };

  struct GenArgs
  {
    datum_ocean_set set;
    GenFrame frame;
    float4 const *map;     // the cascade's displacement map, 2 * N * N float4 (map_index)
    int N;
    int sizex;
    int sizey;
    int tilesx;
    int tiles;
    float *vertices;
#ifdef OCEAN_STAMPS
    unsigned long long *stamps;   // diagnostic builds only (tools/dbg/genstamps.hip): [workgroup][16] timestamps
#endif
  };

  struct f3 { float x, y, z; };

  __host__ __device__ __forceinline__ f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
  __host__ __device__ __forceinline__ f3 operator*(float s, f3 a) { return { s * a.x, s * a.y, s * a.z }; }
  __host__ __device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
  __host__ __device__ __forceinline__ f3 cross3(f3 a, f3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

  // a / |a|: IEEE square root and divisions, as the oracle does it -- for the view ray, whose direction decides where a
  // grazing ray meets the plane (one ulp there moves horizon vertices by whole texels)
  __device__ __forceinline__ f3 normalize3_exact(f3 a) { float l = sqrtf(dot3(a, a)); return { a.x / l, a.y / l, a.z / l }; }

  // a / |a| with the hardware reciprocal square root (1 ulp): the shading frame, where errors are not amplified
  __device__ __forceinline__ f3 normalize3(f3 a) { float inv = __builtin_amdgcn_rsqf(dot3(a, a)); return { a.x * inv, a.y * inv, a.z * inv }; }

  // rotate v by the unit quaternion q = (w, x, y, z)   (data/transform.inc:32-37)
  __device__ __forceinline__ f3 rotate(float const (&q)[4], f3 v)
  {
    f3 u = { q[1], q[2], q[3] };
    f3 tt = 2.0f * cross3(u, v);

    return v + q[0] * tt + cross3(u, tt);
  }

  inline GenFrame make_gen_frame(datum_ocean_set const &p)
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

    // Gerstner swell constants (gen.comp:93-99)
    f.frequency = 2 * 3.14159265358979323846f / p.swelllength;
    f.qi = p.swellsteepness / (f.frequency * p.swellamplitude * 4 + 1e-6f);
    f.phi = f.frequency * p.swellamplitude;

    return f;
  }

  constexpr int GEN_TILE = 16;                  // vertices per tile side
  constexpr int GEN_THREADS = GEN_TILE * GEN_TILE;
  constexpr size_t GEN_LDS = (size_t)GEN_THREADS * 3 * sizeof(float4);

  // Where a bilinear corner comes from.  `want` = the corner's weight is not zero, `shaded` = the normal layer matters.
  //
  // Global map: buffer loads whose offset is pushed out of the buffer's range for a corner that is not wanted -- the
  // hardware then returns zeros without a memory access, and no branch (hence no wait) separates the fetches.
  struct GlobalMap
  {
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void fetch(int texel, bool want, bool shaded, float4 &a, float4 &b) const
    {
      int const off = texel * 16;

      a = buf_load_f32x4_aux<0>(rsrc, want ? off : -16, 0);
      b = buf_load_f32x4_aux<0>(rsrc, (want && shaded) ? off + MAP_GROUP * 16 : -16, 0);
    }
  };

  // One vertex (xx, yy) of data/ocean.gen.comp:67-137 in three stages, so that the kernel can put another vertex's
  // arithmetic between the fetches of one and their use:
  //   gen_ray    view ray, plane hit, Gerstner swell position, distance smoothing, bilinear corners and weights (gen.comp:75-112)
  //   gen_fetch  the (up to) eight texel fetches (gen.comp:113-114)
  //   gen_shade  bilinear blend, shading frame, the vertex as three float4 (gen.comp:113-137)
  struct GenRay
  {
    f3 position;
    float w00, w10, w01, w11;       // bilinear weights
    int t00, t10, t01, t11;         // float4 index of the corners' displacement texels (map_index)
    float smoothing;
    float st, ct;                   // swell phase
  };

  struct GenTexels
  {
    float4 a00, a10, a01, a11;      // displacement layer
    float4 b00, b10, b01, b11;      // normal layer
  };

  __device__ __forceinline__ void gen_ray(GenArgs const &g, int xx, int yy, GenRay &ray)
  {
#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = g.stamps + (size_t)blockIdx.x * 16;
#endif
    OCEAN_STAMP(0);

    datum_ocean_set const &p = g.set;
    GenFrame const &f = g.frame;

    int const N = g.N;

    f3 const camerapos = { f.camerapos[0], f.camerapos[1], f.camerapos[2] };
    f3 const planen = { p.plane[0], p.plane[1], p.plane[2] };

    // exactly the shader's expressions up to the base position: near the horizon the plane hit is ill-conditioned
    float u = (2 * (float)xx / (float)(g.sizex - 1) - 1) * f.margin;
    float v = (1 - 2 * (float)yy / (float)(g.sizey - 1)) * f.margin;

    float const *ip = p.invproj;

    f3 viewvec = { ip[0] * u + ip[1] * v + ip[2] * 0.0f + ip[3] * 1.0f,
                   ip[4] * u + ip[5] * v + ip[6] * 0.0f + ip[7] * 1.0f,
                   ip[8] * u + ip[9] * v + ip[10] * 0.0f + ip[11] * 1.0f };

    f3 worlddir = rotate(p.camera_real, normalize3_exact(viewvec));

    float costheta = dot3(worlddir, f3{ -planen.x, -planen.y, -planen.z });

    float dist = (costheta > 0) ? f.cameraheight / costheta : 1e6f;

    f3 baseposition = { camerapos.x + dist * worlddir.x, camerapos.y + dist * worlddir.y, -p.plane[3] };

    // Gerstner swell (gen.comp:93-109)
    float const amplitude = p.swellamplitude;
    float const dirx = p.swelldirection[0], diry = p.swelldirection[1];
    float const qi = f.qi;

    float theta = f.frequency * (dirx * baseposition.x + diry * baseposition.y) + p.swellphase;

    // theta reaches 1e5..1e6 at the horizon.  The two-constant Cody-Waite step of sincos_phase rounds once, relative to
    // the REDUCED argument (the products k * c are exact inside the FMAs), and what it leaves out is k * 1e-15: good to
    // 1e-7 absolute up to |theta| ~ 1e7, without libm's Payne-Hanek branch.
    float st, ct;
    sincos_phase(theta, &st, &ct);

    f3 position = { baseposition.x + qi * amplitude * dirx * ct, baseposition.y + qi * amplitude * diry * ct, baseposition.z + amplitude * st };

    float cl = dist * p.smoothing - 0.35f;
    cl = fminf(fmaxf(cl, 0.0f), 1.0f);
    float smoothing = __builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf(cl));   // pow(cl, 0.2): 0 -> 0, 1 -> 1 exactly

    // texture(sampler2DArray, REPEAT, linear, lod 0) of both layers at normalised (tu, tv): texel centres at (i + 0.5) / N.
    // N is a power of two: REPEAT is a mask (two's complement makes it right for negative texel indices too).
    float tu = position.x * p.scale;
    float tv = position.y * p.scale;

    float fx = tu * (float)N - 0.5f;
    float fy = tv * (float)N - 0.5f;

    float flx = floorf(fx);
    float fly = floorf(fy);

    float ax = fx - flx;
    float ay = fy - fly;

    int i0 = (int)flx & (N - 1);
    int j0 = (int)fly & (N - 1);
    int i1 = (i0 + 1) & (N - 1);
    int j1 = (j0 + 1) & (N - 1);

    float const w00 = (1 - ax) * (1 - ay), w10 = ax * (1 - ay), w01 = (1 - ax) * ay, w11 = ax * ay;

    // map_index, rows and columns apart (large maps are stored in bands of columns; groups of GX x GY texels).  Every
    // extent is a power of two: shifts by wave-uniform amounts (divisions by run-time values cost some 30 instructions each)
    int const lb = 31 - __builtin_clz(band_cols(N)), ln = 31 - __builtin_clz(N);
    int const lgx = 31 - __builtin_clz(map_group_cols(N)), lgy = 31 - __builtin_clz(map_group_rows(N));
    int const bmask = (1 << lb) - 1, xmask = (1 << lgx) - 1, ymask = (1 << lgy) - 1;

    int const r0 = ((j0 >> lgy) << (1 + lgy + lb)) + ((j0 & ymask) << lgx), r1 = ((j1 >> lgy) << (1 + lgy + lb)) + ((j1 & ymask) << lgx);
    int const c0 = ((i0 >> lb) << (1 + ln + lb)) + (((i0 & bmask) >> lgx) << 3) + (i0 & xmask);
    int const c1 = ((i1 >> lb) << (1 + ln + lb)) + (((i1 & bmask) >> lgx) << 3) + (i1 & xmask);

    static_assert(MAP_GROUP == 4, "a group is one 128-byte line: 8 float4");

    ray.position = position;
    ray.w00 = w00; ray.w10 = w10; ray.w01 = w01; ray.w11 = w11;
    ray.t00 = r0 + c0; ray.t10 = r0 + c1; ray.t01 = r1 + c0; ray.t11 = r1 + c1;
    ray.smoothing = smoothing;
    ray.st = st; ray.ct = ct;

    OCEAN_STAMP(1);
  }

  __device__ __forceinline__ void gen_fetch(GlobalMap const &map, GenRay const &ray, GenTexels &t)
  {
    bool const shaded = ray.smoothing != 1.0f;       // otherwise the sampled normal is multiplied by an exact 0

#ifdef OCEAN_GEN_ABLATE_LOADS      // timing-only builds (tools/): no map fetches
    t.a00 = t.a10 = t.a01 = t.a11 = make_float4(0.01f * (float)(ray.t00 & 7), 0.02f, 0.03f * (float)(ray.t01 & 3), 0.0f);
    t.b00 = t.b10 = t.b01 = t.b11 = make_float4(0.0f, 0.1f, 0.9f, 0.0f);
#else
    // all eight fetches are issued back to back and waited for once (a fetch inside a branch is waited for where the
    // branch rejoins: four round trips to the Infinity Cache per vertex, measured)
    map.fetch(ray.t00, ray.w00 != 0.0f, shaded, t.a00, t.b00);
    map.fetch(ray.t10, ray.w10 != 0.0f, shaded, t.a10, t.b10);
    map.fetch(ray.t01, ray.w01 != 0.0f, shaded, t.a01, t.b01);
    map.fetch(ray.t11, ray.w11 != 0.0f, shaded, t.a11, t.b11);
#endif
  }

  __device__ __forceinline__ void gen_shade(GenArgs const &g, GenRay const &ray, GenTexels const &t, float4 (&out)[3])
  {
#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = g.stamps + (size_t)blockIdx.x * 16;
#endif
    OCEAN_WAIT_LOADS();
    OCEAN_STAMP(2);

    datum_ocean_set const &p = g.set;
    GenFrame const &f = g.frame;

    f3 const planen = { p.plane[0], p.plane[1], p.plane[2] };
    float const dirx = p.swelldirection[0], diry = p.swelldirection[1];
    float const qi = f.qi, phi = f.phi;

    f3 const position = ray.position;
    float const w00 = ray.w00, w10 = ray.w10, w01 = ray.w01, w11 = ray.w11;
    float const smoothing = ray.smoothing, st = ray.st, ct = ray.ct;
    bool const shaded = smoothing != 1.0f;

    float4 const a00 = t.a00, a10 = t.a10, a01 = t.a01, a11 = t.a11;
    float4 const b00 = t.b00, b10 = t.b10, b01 = t.b01, b11 = t.b11;

    // (the sums below the base position are contracted into FMAs: the shading frame and the bilinear blend do not feed
    // an ill-conditioned step, and ocean.gen's tolerance is stated separately from the maps')
    f3 const displacement = { fmaf(w11, a11.x, fmaf(w01, a01.x, fmaf(w10, a10.x, w00 * a00.x))),
                              fmaf(w11, a11.y, fmaf(w01, a01.y, fmaf(w10, a10.y, w00 * a00.y))),
                              fmaf(w11, a11.z, fmaf(w01, a01.z, fmaf(w10, a10.z, w00 * a00.z))) };

    f3 tbn2, tbn0;

    // wave-uniform: a wave whose vertices all lie beyond the smoothing distance (the upper half of the projected grid
    // and the horizon band) skips the Gerstner frame and the normal blend: with smoothing == 1 the blend below is
    // 0 * tn + planen = planen exactly
    if (__builtin_amdgcn_ballot_w64(shaded) != 0)
    {
      float const sixth = 1.0f / 6;

      f3 normal = { phi * dirx * ct * sixth, phi * diry * ct * sixth, qi * phi * st };
      f3 tangent = { qi * phi * dirx * dirx * st, qi * phi * diry * dirx * st, phi * dirx * ct * sixth };

      tbn2 = normalize3(f3{ -normal.x, -normal.y, 1 - normal.z });
      tbn0 = normalize3(f3{ 1 - tangent.x, -tangent.y, tangent.z });

      f3 tbn1 = cross3(tbn0, tbn2);

      f3 const dn = { fmaf(w11, b11.x, fmaf(w01, b01.x, fmaf(w10, b10.x, w00 * b00.x))),
                      fmaf(w11, b11.y, fmaf(w01, b01.y, fmaf(w10, b10.y, w00 * b00.y))),
                      fmaf(w11, b11.z, fmaf(w01, b01.z, fmaf(w10, b10.z, w00 * b00.z))) };

      f3 tn = { fmaf(dn.z, tbn2.x, fmaf(dn.y, tbn1.x, dn.x * tbn0.x)),
                fmaf(dn.z, tbn2.y, fmaf(dn.y, tbn1.y, dn.x * tbn0.y)),
                fmaf(dn.z, tbn2.z, fmaf(dn.y, tbn1.z, dn.x * tbn0.z)) };

      float const keep = 1 - smoothing;

      tbn2 = normalize3(f3{ fmaf(keep, tn.x, smoothing * planen.x), fmaf(keep, tn.y, smoothing * planen.y), fmaf(keep, tn.z, smoothing * planen.z) });
    }
    else
      tbn2 = normalize3(planen);

    float d0 = tbn2.x;
    tbn0 = normalize3(f3{ 1 - d0 * tbn2.x, 0 - d0 * tbn2.y, 0 - d0 * tbn2.z });

    OCEAN_STAMP(3);

    // Mesh::Vertex { position3, texcoord2, normal3, tangent4 } = 48 bytes (src/renderer/mesh.h:20-26)
    out[0] = make_float4(position.x - displacement.x, position.y - displacement.y, position.z + displacement.z, 0.1f * position.x);
    out[1] = make_float4(0.1f * position.y, tbn2.x, tbn2.y, tbn2.z);
    out[2] = make_float4(tbn0.x, tbn0.y, tbn0.z, -1.0f);
  }

  __device__ __forceinline__ void gen_vertex(GenArgs const &g, GlobalMap const &map, int xx, int yy, float4 (&out)[3])
  {
    GenRay ray;
    GenTexels texels;

    gen_ray(g, xx, yy, ray);
    gen_fetch(map, ray, texels);
    gen_shade(g, ray, texels, out);
  }

  // The wave's 4 rows x 16 vertices = 4 x 48 float4 go through its 3 KB of LDS: thread i then stores float4 number
  // i, 64 + i, 128 + i of the wave's 192 (three 16-byte stores per thread at a 48-byte stride touched every line three
  // times: 26 -> 21 us per 1024^2 mesh from 64^2 maps).
  __device__ __forceinline__ void gen_store_tile(GenArgs const &g, float4 *stage, int tilex, int tiley, int tid, float4 const (&vtx)[3])
  {
    int const lane = tid & 63;
    float4 *mine = stage + 3 * (tid - lane);       // this wave's 192 float4

    mine[3 * lane + 0] = vtx[0];
    mine[3 * lane + 1] = vtx[1];
    mine[3 * lane + 2] = vtx[2];

    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int const x0 = tilex * GEN_TILE;
    int const y0 = tiley * GEN_TILE + 4 * (tid >> 6);
    int const rowlen = min(GEN_TILE, g.sizex - x0) * 3;                                // float4 of this tile in one mesh row

    #pragma unroll
    for(int k = 0; k < 3; ++k)
    {
      int const j = 64 * k + lane;
      int const r = j / 48, c = j % 48;

#ifdef OCEAN_GEN_ABLATE_STORES     // timing-only builds: the values stay live, nothing is written
      if (mine[j].w == 123456.789f)
#endif
      if (c < rowlen && y0 + r < g.sizey)
        reinterpret_cast<float4*>(g.vertices)[((size_t)(y0 + r) * g.sizex + x0) * 3 + c] = mine[j];
    }

  }

#ifndef OCEAN_GEN_PIPELINED
#define OCEAN_GEN_PIPELINED 0
#endif
#ifndef OCEAN_GEN_GROUPS_PER_CU
#define OCEAN_GEN_GROUPS_PER_CU 5
#endif

  // One tile per workgroup, tiles in row-major order (68 VGPRs = 7 workgroups per CU; forced to 64 for 8: 18.7 against 17.7 us).
  //
  // OCEAN_GEN_PIPELINED (measured, off): persistent workgroups walk the tiles (workgroup b takes tiles b, b + gridDim.x, ...)
  // and put the ray and swell arithmetic of the NEXT tile's vertex between the fetches of the current one and their use
  // (the request after the last tile repeats it: no branch around loads, see GlobalMap).  96 VGPRs, five workgroups per
  // CU: 35.7-37.3 us against 34.3 us from 1024^2 maps, 20.1-20.8 against 18.1 us from 64^2 maps, with 3 / 4 / 5 / 6 / 8
  // workgroups per CU alike; every second workgroup starting 3.4 us late in the one-tile kernel: + 2.5-4.5 us
  // (profiles/r02_gen_experiments.txt).
  __global__ void __launch_bounds__(GEN_THREADS) ocean_gen_kernel(GenArgs g)
  {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    float4 *stage = reinterpret_cast<float4*>(smem);

    int const tid = threadIdx.x;

    GlobalMap const map = { make_rsrc(g.map, (size_t)2 * g.N * g.N * sizeof(float4)) };

#if OCEAN_GEN_PIPELINED
    int const stride = (int)gridDim.x;
    int tile = (int)blockIdx.x;

    GenRay ray;
    GenTexels texels;

    gen_ray(g, (tile % g.tilesx) * GEN_TILE + (tid & 15), (tile / g.tilesx) * GEN_TILE + (tid >> 4), ray);
    gen_fetch(map, ray, texels);

    for(; tile < g.tiles; tile += stride)
    {
      int const next = (tile + stride < g.tiles) ? tile + stride : tile;

      GenRay ahead;
      gen_ray(g, (next % g.tilesx) * GEN_TILE + (tid & 15), (next / g.tilesx) * GEN_TILE + (tid >> 4), ahead);

      // hipcc otherwise sinks this arithmetic below the shading (nothing there depends on it) and the fetches are waited
      // for as soon as they are issued: pin `ahead` here and keep the scheduler from moving anything across
      asm volatile("" :: "v"(ahead.position.x), "v"(ahead.position.y), "v"(ahead.position.z), "v"(ahead.w00), "v"(ahead.w10), "v"(ahead.w01), "v"(ahead.w11),
                         "v"(ahead.t00), "v"(ahead.t10), "v"(ahead.t01), "v"(ahead.t11), "v"(ahead.smoothing), "v"(ahead.st), "v"(ahead.ct));
      __builtin_amdgcn_sched_barrier(0);

      float4 vtx[3];
      gen_shade(g, ray, texels, vtx);
      gen_store_tile(g, stage, tile % g.tilesx, tile / g.tilesx, tid, vtx);

      ray = ahead;
      gen_fetch(map, ray, texels);
    }
#else
    int const tile = (int)blockIdx.x;
    int const tilex = tile % g.tilesx, tiley = tile / g.tilesx;

    float4 vtx[3];
    gen_vertex(g, map, tilex * GEN_TILE + (tid & 15), tiley * GEN_TILE + (tid >> 4), vtx);
    gen_store_tile(g, stage, tilex, tiley, tid, vtx);
#endif

#ifdef OCEAN_STAMPS
    unsigned long long *stampbase = g.stamps + (size_t)blockIdx.x * 16;
#endif
    OCEAN_STAMP_WHERE();
    OCEAN_STAMP(4);
  }
}
