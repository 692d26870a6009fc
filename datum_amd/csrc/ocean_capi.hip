// ocean_capi.hip -- the extern "C" module of include/datum_ocean_hip.h (host side of the HIP path).
//
// Owns the device buffers the reference keeps in OceanContext (src/renderer/ocean.h:12-46:
// oceanset, spectrum, displacementmap) and enqueues the kernels of ocean_kernels.hip where the
// reference records its five dispatches (src/renderer/ocean.cpp:769-793).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "ocean_kernels.hip"
#include "ocean_gen.hip"
#include "ocean_literal.hip"
#include "ocean_farm.hip"

using namespace ocean;

struct datum_ocean_ctx
{
  int device = 0;
  int N = 0;
  int cascades = 0;

  int cus = 0;                        // compute units of the device

  hipStream_t stream = nullptr;       // the one in use
  hipStream_t ownstream = nullptr;

  float2 *h0 = nullptr;
  float2 *seed = nullptr;             // [cascade][N*N] OceanParams::seed, only when the caller uploads it
  float *phase = nullptr;
  void *spec = nullptr;               // cd[cascades][P], or ch[...] with the fp16 spectrum
  bool half = false;                  // DATUM_OCEAN_SPECTRUM_FP16 or _FP16_H0
  bool h0half = false;                // DATUM_OCEAN_SPECTRUM_FP16_H0: the row pass reads h0 as halves ...
  unsigned int *h0h = nullptr;        // ... from this copy, [cascades][P] x two halves, rebuilt with the scale when h0 changes (size_spectrum_scale)
  int cascadegroup = 0;               // cascades per launch of the two passes, 0 = sized to the Infinity Cache (cascade_group)
  int mappolicy = DATUM_OCEAN_MAPS_AUTO;   // datum_ocean_set_map_store_policy

  // validation mode (datum_ocean_set_literal_transform): the reference's own radix-2 transforms with its literal twiddle table
  bool literal = false;
  float2 *litfields = nullptr;        // [3][N*N]: h, hx, hy of the cascade being displaced (the reference's Spectrum buffer, ocean.cpp:61-68)
  float *litweights = nullptr;        // [N][2 log2 N] (ocean.cpp:686-700)
  bool scaledirty[DATUM_OCEAN_MAX_CASCADES] = {};   // fp16 only: h0 changed since specscale was sized
  unsigned int *absmax = nullptr;     // device word for ocean_absmax_kernel
  float4 *maps = nullptr;             // the one in use
  float4 *ownmaps = nullptr;
  cf *tw = nullptr;
  float *omega = nullptr;             // [cascade][(N/2+1)^2] dispersion quadrant, rebuilt when a wavescale changes
  unsigned int omegadirty = ~0u;      // cascades whose wave scale changed since their table was built
  float omegamax[DATUM_OCEAN_MAX_CASCADES] = {};   // largest dispersion of each cascade (table corner)
  bool wildphase[DATUM_OCEAN_MAX_CASCADES] = {};   // an uploaded phase lies outside [0, 2 pi)
  cf *scratch = nullptr;              // 3 row-major planes for the debug read-backs (lazy)

  CascadeConst casc[DATUM_OCEAN_MAX_CASCADES];
  bool uploaded[DATUM_OCEAN_MAX_CASCADES] = {};

  std::vector<float> pending;         // queued update_ocean dt's

  hipEvent_t complete = nullptr;      // "rendercomplete"

  // imported from the renderer (Vulkan external memory / semaphores)
  struct ImportedMemory { hipExternalMemory_t memory; void *ptr; size_t bytes; };
  std::vector<ImportedMemory> importedmemory;
  std::vector<hipExternalSemaphore_t> importedsemaphores;

  ocean::Farm *farm = nullptr;        // the tile farm's communicator, stream and double-buffered payload (datum_ocean_farm_init)

  // profiling
  bool profiling = false;
  int profmax = 0;
  int profsteps = 0;
  int profstride = 1;
  long profcalls = 0;
  std::vector<hipEvent_t> events;     // 4 per sampled step

  std::string error;
};

namespace
{
  std::string g_error;   // errors without a handle

  int fail(datum_ocean_ctx *ctx, int code, char const *what)
  {
    char buf[512];

    if (code > 0)
      snprintf(buf, sizeof(buf), "%s: %s (hipError %d)", what, hipGetErrorString((hipError_t)code), code);
    else
      snprintf(buf, sizeof(buf), "%s (code %d)", what, code);

    (ctx ? ctx->error : g_error) = buf;

    (void)hipGetLastError();   // the runtime's sticky last-error must not leak into a later, unrelated call

    return code;
  }

  #define HIPCHECK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, (int)e_, #call); } while(0)

  bool supported(int n)
  {
    return n == 64 || n == 128 || n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096;
  }

  size_t plane(datum_ocean_ctx const *ctx) { return (size_t)ctx->N * ctx->N; }

  StepArgs make_args(datum_ocean_ctx *ctx, int ndt, float const *dt)
  {
    StepArgs a;
    a.h0 = ctx->h0;
    a.h0h = ctx->h0half ? ctx->h0h : nullptr;
    a.phase = ctx->phase;
    a.spec = ctx->spec;
    a.maps = ctx->maps;
    a.tw = ctx->tw;
    a.omega = ctx->omega;
    a.ndt = ndt;
    a.cascades = ctx->cascades;
    a.first = 0;

    for(int i = 0; i < MAX_PENDING; ++i)
      a.dt[i] = (i < ndt) ? dt[i] : 0.0f;
    memcpy(a.casc, ctx->casc, sizeof(a.casc));
    return a;
  }

  // Cascades per launch of the two passes (replaces the one dispatch per shader of ocean.cpp:769-789).  A handle whose working set is resident
  // in the Infinity Cache takes every cascade in one launch per pass.  Beyond it the maps are streamed (ocean_kernels.hip: MAP_STORE_AUX) and
  // what can stay in the cache from the row pass to the column pass -- and for h0 and the phase from step to step -- is h0 8 + phase 4 + work
  // spectrum 16 (8: fp16) bytes per point: the passes are launched group by group -- row(g), column(g), row(g + 1), ... on the same stream,
  // the work spectrum's slots reused from group to group -- with the largest group whose share of that fits: 8 cascades of 1024^2, 2 of 2048^2,
  // 1 of 4096^2; groups of equal size where the cascades allow it (twelve as 6 + 6).  Measured (profiles/r06_cascade_groups.txt): 1024^2 x 16
  // as 2 x 8 77.8 k grids/s against 66.7 k in one launch and 71.1 k as 4 x 4; x 12 as 2 x 6 80.6 k against 74.2 k; x 8 in one launch 82.6 k
  // against 78.3 k as 2 x 4; 2048^2 x 4 as 2 x 2 18.2 k against 16.0 k in one launch and 17.2 k as 4 x 1.
  constexpr double CASCADE_GROUP_BYTES = 240.0e6;

  // The maps streamed past the Infinity Cache instead of written through (ocean_kernels.hip: MAP_STORE_AUX_STREAM; 1024^2 and 2048^2 have both forms):
  // where the handle's own working set is beyond the cache, and -- round 6, profiles/r06_farm_standin.txt -- while a farm of several ranks is
  // initialised: the collective's gathered buffer competes for the same cache, and the step loses less under it with the maps out of the way.
  bool handle_streams_maps(datum_ocean_ctx const *ctx);

  int cascade_group(datum_ocean_ctx const *ctx)
  {
    int g = ctx->cascadegroup;

    if (g > 0)
      return g > ctx->cascades ? ctx->cascades : g;

    if (!handle_streams_maps(ctx) || !maps_stream(ctx->N, ctx->cascades, ctx->half))
      return ctx->cascades;

    g = (int)(CASCADE_GROUP_BYTES / ((double)plane(ctx) * ((ctx->h0half ? 8.0 : 12.0) + (ctx->half ? 8.0 : 16.0))));
    g = g < 1 ? 1 : (g > ctx->cascades ? ctx->cascades : g);

    int const groups = (ctx->cascades + g - 1) / g;

    return (ctx->cascades + groups - 1) / groups;
  }

  template<int N, bool H16>
  hipError_t configure_one(char const **what)
  {
    *what = "hipFuncSetAttribute(ocean_rowpass_kernel, MaxDynamicSharedMemorySize)";
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, H16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N, H16>::LDS);
    if (e != hipSuccess)
      return e;

    if constexpr (H16)
    {
      *what = "hipFuncSetAttribute(ocean_rowpass_kernel, the instantiations with h0 as halves, MaxDynamicSharedMemorySize)";
      e = hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N, true>::LDS);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N, true>::LDS);
      if (e != hipSuccess)
        return e;
    }

    *what = "hipFuncSetAttribute(ocean_rowpass_kernel, the instantiation for phases outside [0, 2 pi), MaxDynamicSharedMemorySize)";
    e = hipFuncSetAttribute(reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, H16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowCfg<N, H16>::LDS);
    if (e != hipSuccess)
      return e;

    *what = "hipFuncSetAttribute(ocean_colpass_kernel, MaxDynamicSharedMemorySize)";
    e = hipFuncSetAttribute(colpass_entry<N, H16>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ColCfg<N>::LDS);
    if (e != hipSuccess)
      return e;

    if constexpr (col_has_stream_variant<N>())
      e = hipFuncSetAttribute(colpass_entry<N, H16, true>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ColCfg<N>::LDS);

    return e;
  }

  template<int N>
  hipError_t configure(datum_ocean_ctx *ctx, char const **what)
  {
    (void)ctx;

    hipError_t e = configure_one<N, false>(what);

    if (e == hipSuccess)
      e = configure_one<N, true>(what);

    return e;
  }

  // ev != nullptr: the dispatch itself carries a start and a stop event (hipExtLaunchKernel), so a sampled kernel is
  // timed from its first to its last workgroup without event packets between the kernels -- an hipEventRecord between
  // the two passes held the second one back until the first had drained (+4 us per kernel at 1024^2 x 4).
  hipError_t launch(void const *kernel, dim3 grid, dim3 block, void **args, size_t lds, hipStream_t stream, hipEvent_t *ev)
  {
    if (ev)
      return hipExtLaunchKernel(kernel, grid, block, args, lds, stream, ev[0], ev[1], 0);

    return hipLaunchKernel(kernel, grid, block, args, lds, stream);
  }

  template<int N, bool H16, bool H0H = false>
  hipError_t launch_rowpass_as(datum_ocean_ctx *ctx, StepArgs &a, hipEvent_t *ev)
  {
    typedef RowCfg<N, H16> C;

    // a phase outside [0, 2 pi) in some cascade (uploaded so, or left there by a negative dt): the instantiation whose sin / cos
    // take any argument
    bool wild = false;

    for(int c = 0; c < ctx->cascades; ++c)
      wild = wild || ctx->wildphase[c];

    void *args[] = { &a };
    void const *kernel = wild ? reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, H16, true, H0H>) : reinterpret_cast<void const*>(&ocean_rowpass_kernel<N, H16, false, H0H>);

    // work items = groups of row pairs x the cascades of this launch: one workgroup each
    int const items = C::GROUPS * a.cascades;

    return launch(kernel, dim3(items), dim3(C::THREADS), args, C::LDS, ctx->stream, ev);
  }

  template<int N>
  hipError_t launch_rowpass(datum_ocean_ctx *ctx, StepArgs &a, hipEvent_t *ev)
  {
    if (ctx->h0half)
      return launch_rowpass_as<N, true, true>(ctx, a, ev);

    return ctx->half ? launch_rowpass_as<N, true>(ctx, a, ev) : launch_rowpass_as<N, false>(ctx, a, ev);
  }

  template<int N>
  hipError_t launch_colpass(datum_ocean_ctx *ctx, StepArgs &a, hipEvent_t *ev)
  {
    void *args[] = { &a };
    void const *kernel = ctx->half ? colpass_entry<N, true>() : colpass_entry<N, false>();

    // the maps streamed instead of written through where the HANDLE's working set is beyond the Infinity Cache (whatever this launch's share of it)
    if constexpr (col_has_stream_variant<N>())
    {
      if (handle_streams_maps(ctx))
        kernel = ctx->half ? colpass_entry<N, true, true>() : colpass_entry<N, false, true>();
    }

    // work items = tiles x cascades; the large grids' workgroups are persistent, one per compute unit (the LDS of a
    // 1024-thread tile fills a CU), and walk their share of the items
    int const items = ColCfg<N>::TILES * a.cascades;
    bool const walks = ctx->half ? col_walks<N, true>() : col_walks<N, false>();
    int const groups = walks ? std::min(items, ctx->cus) : items;

    return launch(kernel, dim3(groups), dim3(ColCfg<N>::THREADS), args, ColCfg<N>::LDS, ctx->stream, ev);
  }

  #define DISPATCH_N(n, expr) \
    switch(n) { \
      case 64: { constexpr int NN = 64; expr; } break; \
      case 128: { constexpr int NN = 128; expr; } break; \
      case 256: { constexpr int NN = 256; expr; } break; \
      case 512: { constexpr int NN = 512; expr; } break; \
      case 1024: { constexpr int NN = 1024; expr; } break; \
      case 2048: { constexpr int NN = 2048; expr; } break; \
      case 4096: { constexpr int NN = 4096; expr; } break; \
    }

  // (re)build the dispersion quadrant tables of the cascades whose wave scale changed
  int ensure_omega(datum_ocean_ctx *ctx)
  {
    unsigned int const dirty = ctx->omegadirty & ((ctx->cascades >= 32) ? ~0u : ((1u << ctx->cascades) - 1));

    if (!dirty)
      return DATUM_OCEAN_OK;

    WaveScales ws;

    for(int c = 0; c < DATUM_OCEAN_MAX_CASCADES; ++c)
    {
      ws.v[c] = ctx->casc[c].wavescale;

      // dispersion grows with |k|: its maximum is the table corner |m - N/2| = |n - N/2| = N/2 (same fp32 formula)
      float kc = (6.2831855f * (0.5f * (float)ctx->N)) / ws.v[c];
      float k2 = kc * kc + kc * kc;
      ctx->omegamax[c] = sqrtf((9.81f * sqrtf(k2)) * (1.0f + k2 / 136900.0f));
    }

    hipLaunchKernelGGL(ocean_omega_kernel, dim3(512), dim3(256), 0, ctx->stream, ctx->omega, ctx->N, ctx->cascades, ws, dirty);
    HIPCHECK(ctx, hipGetLastError());

    ctx->omegadirty = 0;

    return DATUM_OCEAN_OK;
  }

  // The fused row pass advances the phase with a single conditional subtraction, exact only while
  // 0 <= phase < 2 pi and 0 <= dispersion * dt < 2 pi.  Anything else goes through the phase-only kernel.
  bool fusable(datum_ocean_ctx *ctx)
  {
    for(int c = 0; c < ctx->cascades; ++c)
    {
      if (ctx->wildphase[c])
        return false;

      for(float dt : ctx->pending)
      {
        if (!(dt >= 0.0f) || !(ctx->omegamax[c] * dt < 6.0f))
          return false;
      }
    }

    return true;
  }

  // flush queued updates that do not fit into one displace call
  int flush_pending(datum_ocean_ctx *ctx, size_t keep)
  {
    if (ctx->pending.size() > keep)
    {
      int rc = ensure_omega(ctx);
      if (rc != DATUM_OCEAN_OK)
        return rc;
    }

    while (ctx->pending.size() > keep)
    {
      int n = (int)std::min<size_t>(MAX_PENDING, ctx->pending.size() - keep);

      StepArgs a = make_args(ctx, n, ctx->pending.data());

      // fmod keeps the sign of its first operand: a negative dt can leave phases below zero, which the
      // fused fast path must never see
      for(int i = 0; i < n; ++i)
      {
        if (!(ctx->pending[i] >= 0.0f))
        {
          for(int c = 0; c < ctx->cascades; ++c)
            ctx->wildphase[c] = true;
        }
      }

      dim3 grid(1024, ctx->cascades);
      hipLaunchKernelGGL(ocean_advance_kernel, grid, dim3(256), 0, ctx->stream, a, ctx->N);
      HIPCHECK(ctx, hipGetLastError());

      ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + n);
    }

    return DATUM_OCEAN_OK;
  }

  // fp16 work spectrum: the power of two that brings the largest possible row sum under the largest half.
  // With mu = max |h0| (complex modulus): |h~| <= 2 mu, |h~[k] + conj(h~[-k])| <= 4 mu, |C| <= 8 mu, |D| <= 12 mu,
  // so no component after the row transform exceeds 12 N mu.  Typical values are ~ sqrt(N) times smaller and
  // still far above half's denormal range.
  int size_spectrum_scale(datum_ocean_ctx *ctx)
  {
    if (!ctx->half)
      return DATUM_OCEAN_OK;

    size_t const P = plane(ctx);

    for(int c = 0; c < ctx->cascades; ++c)
    {
      if (!ctx->scaledirty[c])
        continue;

      unsigned int bits = 0;

      HIPCHECK(ctx, hipMemsetAsync(ctx->absmax, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(ocean_absmax_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->h0 + c * P, P, ctx->absmax);
      HIPCHECK(ctx, hipGetLastError());
      HIPCHECK(ctx, hipMemcpyAsync(&bits, ctx->absmax, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

      float m;
      memcpy(&m, &bits, sizeof(m));

      if (!(m < 3.0e38f))
        return fail(ctx, DATUM_OCEAN_EINVAL, "fp16 spectrum: h0 holds a NaN or an infinity");

      double const bound = 12.0 * ctx->N * (1.41421356 * (double)m);
      int e = (bound > 0) ? (int)std::floor(std::log2(60000.0 / bound)) : 0;

      // (the exponent follows max |h0| over the whole range in which 2^e and 2^-e are normal floats with room to spare: a
      // sea 2^-40 times smaller than the example's must not land in half's denormals.  Round 3 clamped at +-24.)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);

      ctx->casc[c].specscale = std::ldexp(1.0f, e);
      ctx->casc[c].specinv = std::ldexp(1.0f, -e);
      ctx->casc[c].rowscale = ctx->casc[c].specscale;

      if (ctx->h0half)
      {
        // h0 as halves: the power of two that brings the largest component just under 2^15 (the whole of half's range below it is
        // h0's: 2^-29 of the largest value is still a normal half, 2^-39 a denormal one); the row pass takes it out again with the
        // work spectrum's scale in one exact factor
        int eh = (m > 0) ? (int)std::floor(std::log2(32768.0 / (double)m)) : 0;

        if (m > 0 && std::ldexp((double)m, eh) >= 32768.0)
          eh -= 1;

        eh = eh > 120 ? 120 : (eh < -120 ? -120 : eh);

        hipLaunchKernelGGL(ocean_h0half_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->h0 + c * P, ctx->h0h + c * P, P, std::ldexp(1.0f, eh));
        HIPCHECK(ctx, hipGetLastError());

        ctx->casc[c].rowscale = std::ldexp(1.0f, e - eh);
      }

      ctx->scaledirty[c] = false;
    }

    return DATUM_OCEAN_OK;
  }

  int ensure_scratch(datum_ocean_ctx *ctx)
  {
    if (!ctx->scratch)
      HIPCHECK(ctx, hipMalloc(&ctx->scratch, 3 * plane(ctx) * sizeof(cf)));

    return DATUM_OCEAN_OK;
  }

  // the map layout's shape as shifts (ocean_pack_kernel, ocean_export_kernel: texel <- part number)
  PackShape pack_shape(int N)
  {
    auto log2of = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };

    int const B = band_cols(N), PW = map_patch_cols(N), PH = map_patch_rows(N);

    PackShape sh;
    sh.n2 = log2of(N);
    sh.pw2 = log2of(PW);
    sh.bp2 = log2of(B / PW);
    sh.bandpatches2 = log2of((N / PH) * (B / PW));
    sh.b2 = log2of(B);

    return sh;
  }

  // the pack of datum_ocean_pack_displacement / datum_ocean_farm_gather, on the handle's stream (arguments checked by the callers)
  int pack_into(datum_ocean_ctx *ctx, int format, void *payload_device, size_t need)
  {
    if (format == DATUM_OCEAN_PAYLOAD_MAPS)
    {
      // the map block as it lies in memory (device layout), so that the producer may go on writing its own buffer
      HIPCHECK(ctx, hipMemcpyAsync(payload_device, ctx->maps, need, hipMemcpyDeviceToDevice, ctx->stream));
    }
    else
    {
      int const blocks = std::min<size_t>((size_t)ctx->cus * 8, ((size_t)ctx->cascades * plane(ctx) / 4 + 255) / 256);

      PackShape const sh = pack_shape(ctx->N);

      if (format == DATUM_OCEAN_PAYLOAD_XYZ16)
        hipLaunchKernelGGL(ocean_pack_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->maps, ctx->N, ctx->cascades, payload_device, sh);
      else
        hipLaunchKernelGGL(ocean_pack_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->maps, ctx->N, ctx->cascades, payload_device, sh);

      HIPCHECK(ctx, hipGetLastError());
    }

    return DATUM_OCEAN_OK;
  }

  // an ncclResult_t never leaves the module
  int fail_comm(datum_ocean_ctx *ctx, ocean::RcclApi *api, ncclComm_t comm, ncclResult_t r, char const *what)
  {
    char buf[768];

    char const *last = (api && api->GetLastError) ? api->GetLastError(comm) : "";

    snprintf(buf, sizeof(buf), "%s: %s (ncclResult %d)%s%s", what, (api && api->GetErrorString) ? api->GetErrorString(r) : "?", (int)r, (last && *last) ? ": " : "", (last && *last) ? last : "");

    (ctx ? ctx->error : g_error) = buf;

    // a communicator that failed is not destroyed (ncclCommDestroy waits for its outstanding operations) but aborted: farm_teardown
    if (ctx && ctx->farm && comm && comm == ctx->farm->comm)
      ctx->farm->failed = true;

    return DATUM_OCEAN_ECOMM;
  }

  #define RCCLCHECK(ctx, api, comm, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail_comm(ctx, api, comm, r_, #call); } while(0)

  void farm_teardown(datum_ocean_ctx *ctx)
  {
    ocean::Farm *f = ctx->farm;

    if (!f)
      return;

    if (f->stream && !f->failed)
      (void)hipStreamSynchronize(f->stream);

    if (f->comm && f->api)
    {
      if (f->failed && f->api->CommAbort)
        (void)f->api->CommAbort(f->comm);
      else
        (void)f->api->CommDestroy(f->comm);
    }

    if (f->stream && f->failed)
      (void)hipStreamSynchronize(f->stream);

    for(auto &sl : f->slots)
    {
      (void)hipFree(sl.payload);
      (void)hipFree(sl.gathered);

      for(hipEvent_t e : { sl.packed, sl.start, sl.done, sl.consumed })
        if (e)
          (void)hipEventDestroy(e);
    }

    if (f->stream)
      (void)hipStreamDestroy(f->stream);

    // datum_ocean_farm_partition had confined the handle's own stream to the compute units the collective left it: all of them again
    if (f->commcus && ctx->ownstream)
    {
      hipStream_t whole = nullptr;

      (void)hipStreamSynchronize(ctx->ownstream);

      if (hipStreamCreateWithFlags(&whole, hipStreamNonBlocking) == hipSuccess)
      {
        bool const onown = ctx->stream == ctx->ownstream;

        (void)hipStreamDestroy(ctx->ownstream);
        ctx->ownstream = whole;

        if (onown)
          ctx->stream = whole;
      }
    }

    delete f;

    ctx->farm = nullptr;
  }
}

extern "C"
{

int datum_ocean_create(datum_ocean_t *out, int device, int resolution, int cascades)
{
  if (!out)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_create: null out pointer");

  *out = nullptr;

  if (!supported(resolution))
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_create: resolution must be a power of two in [64, 4096]");

  if (cascades < 1 || cascades > DATUM_OCEAN_MAX_CASCADES)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_create: cascades out of range");

  HIPCHECK(nullptr, hipSetDevice(device));

  datum_ocean_ctx *ctx = new (std::nothrow) datum_ocean_ctx;
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_ENOMEM, "datum_ocean_create: out of host memory");

  ctx->device = device;
  ctx->N = resolution;
  ctx->cascades = cascades;

  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1)
      cus = 256;
    ctx->cus = cus;
  }

  size_t const P = plane(ctx);

  #define CREATECHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int rc_ = fail(nullptr, (int)e_, #call); datum_ocean_destroy(ctx); return rc_; } } while(0)

  CREATECHECK(hipStreamCreateWithFlags(&ctx->ownstream, hipStreamNonBlocking));
  ctx->stream = ctx->ownstream;

  CREATECHECK(hipMalloc(&ctx->h0, cascades * P * sizeof(float2)));
  CREATECHECK(hipMalloc(&ctx->phase, cascades * P * sizeof(float)));
  CREATECHECK(hipMalloc(&ctx->spec, cascades * P * sizeof(cd)));
  CREATECHECK(hipMalloc(&ctx->absmax, sizeof(unsigned int)));
  CREATECHECK(hipMalloc(&ctx->ownmaps, cascades * map_cascade_bytes(resolution)));
  CREATECHECK(hipMalloc(&ctx->tw, resolution * sizeof(cf)));
  CREATECHECK(hipMalloc(&ctx->omega, (size_t)cascades * (resolution / 2 + 1) * (resolution / 2 + 1) * sizeof(float)));
  ctx->maps = ctx->ownmaps;

  CREATECHECK(hipMemsetAsync(ctx->h0, 0, cascades * P * sizeof(float2), ctx->stream));
  CREATECHECK(hipMemsetAsync(ctx->phase, 0, cascades * P * sizeof(float), ctx->stream));
  CREATECHECK(hipMemsetAsync(ctx->ownmaps, 0, cascades * map_cascade_bytes(resolution), ctx->stream));

  // exp(+2 pi i k / N), rounded once from double (the reference's table -- ocean.cpp:686-700 -- is per
  // lane and stage and evaluated at unreduced fp32 angles; see datum_ocean_reference_weights)
  std::vector<cf> tw(resolution);
  for(int k = 0; k < resolution; ++k)
  {
    double ang = 2.0 * 3.14159265358979323846 * k / resolution;
    tw[k] = cf{ (float)std::cos(ang), (float)std::sin(ang) };
  }
  // exact values on the axes and diagonals
  tw[0] = cf{ 1.0f, 0.0f };
  tw[resolution/4] = cf{ 0.0f, 1.0f };
  tw[resolution/2] = cf{ -1.0f, 0.0f };
  tw[3*resolution/4] = cf{ 0.0f, -1.0f };

  CREATECHECK(hipMemcpy(ctx->tw, tw.data(), resolution * sizeof(cf), hipMemcpyHostToDevice));

  for(int c = 0; c < DATUM_OCEAN_MAX_CASCADES; ++c)
  {
    // OceanParams defaults (ocean.h:60,64)
    ctx->casc[c].wavescale = 64.0f;
    ctx->casc[c].scale = 1 / 64.0f;
    ctx->casc[c].choppiness = 1.35f;
    ctx->casc[c].nz = 4 / (ctx->casc[c].scale * resolution);
    ctx->casc[c].specscale = 1.0f;
    ctx->casc[c].specinv = 1.0f;
    ctx->casc[c].rowscale = 1.0f;
  }

  {
    hipError_t ce = hipSuccess;
    char const *what = "";
    DISPATCH_N(resolution, ce = configure<NN>(ctx, &what));
    if (ce != hipSuccess)
    {
      int rc = fail(nullptr, (int)ce, what);
      datum_ocean_destroy(ctx);
      return rc;
    }
  }

  CREATECHECK(hipStreamSynchronize(ctx->stream));

  #undef CREATECHECK

  *out = ctx;

  return DATUM_OCEAN_OK;
}

int datum_ocean_destroy(datum_ocean_t ctx)
{
  if (!ctx)
    return DATUM_OCEAN_OK;

  (void)hipSetDevice(ctx->device);

  if (ctx->stream)
    (void)hipStreamSynchronize(ctx->stream);

  farm_teardown(ctx);

  for(hipEvent_t e : ctx->events)
    (void)hipEventDestroy(e);

  if (ctx->complete)
    (void)hipEventDestroy(ctx->complete);

  for(auto &im : ctx->importedmemory)
    (void)hipDestroyExternalMemory(im.memory);

  for(hipExternalSemaphore_t sem : ctx->importedsemaphores)
    (void)hipDestroyExternalSemaphore(sem);

  (void)hipFree(ctx->h0);
  (void)hipFree(ctx->seed);
  (void)hipFree(ctx->phase);
  (void)hipFree(ctx->spec);
  (void)hipFree(ctx->absmax);
  (void)hipFree(ctx->h0h);
  (void)hipFree(ctx->ownmaps);
  (void)hipFree(ctx->tw);
  (void)hipFree(ctx->omega);
  (void)hipFree(ctx->litfields);
  (void)hipFree(ctx->litweights);
  (void)hipFree(ctx->scratch);

  if (ctx->ownstream)
    (void)hipStreamDestroy(ctx->ownstream);

  delete ctx;

  return DATUM_OCEAN_OK;
}

int datum_ocean_set_stream(datum_ocean_t ctx, void *hip_stream, int use_own)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_stream: null handle");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  ctx->stream = use_own ? ctx->ownstream : (hipStream_t)hip_stream;

  return DATUM_OCEAN_OK;
}

int datum_ocean_bind_maps(datum_ocean_t ctx, void *device_ptr, size_t bytes)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_bind_maps: null handle");

  size_t need = ctx->cascades * map_cascade_bytes(ctx->N);

  if (device_ptr && bytes < need)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_bind_maps: buffer smaller than cascades * N * N * texel_bytes (datum_ocean_map_layout)");

  if (device_ptr && ((uintptr_t)device_ptr & 15))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_bind_maps: buffer must be 16-byte aligned");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  ctx->maps = device_ptr ? (float4*)device_ptr : ctx->ownmaps;

  return DATUM_OCEAN_OK;
}

int datum_ocean_maps_device(datum_ocean_t ctx, void **device_ptr, size_t *bytes)
{
  if (!ctx || !device_ptr)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_maps_device: null argument");

  *device_ptr = ctx->maps;

  if (bytes)
    *bytes = ctx->cascades * map_cascade_bytes(ctx->N);

  return DATUM_OCEAN_OK;
}

int datum_ocean_map_layout(int resolution, int *group_cols, int *group_rows, int *band, int *texel_bytes)
{
  if (!supported(resolution) || !group_cols || !group_rows || !band || !texel_bytes)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_map_layout: bad argument");

  *group_cols = map_patch_cols(resolution);
  *group_rows = map_patch_rows(resolution);
  *band = band_cols(resolution);
  *texel_bytes = 24;

  return DATUM_OCEAN_OK;
}

int datum_ocean_set_cascade(datum_ocean_t ctx, int cascade, float wavescale, float choppiness)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_cascade: null handle");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_set_cascade: cascade out of range");

  if (!(wavescale > 0))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_set_cascade: wavescale must be positive");

  // queued updates were issued under the old wavescale: apply them first
  if (wavescale != ctx->casc[cascade].wavescale && !ctx->pending.empty())
  {
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    int rc = flush_pending(ctx, 0);
    if (rc != DATUM_OCEAN_OK)
      return rc;
  }

  CascadeConst &cc = ctx->casc[cascade];

  if (cc.wavescale != wavescale)
    ctx->omegadirty |= 1u << cascade;

  cc.wavescale = wavescale;
  cc.scale = 1 / wavescale;                      // ocean.cpp:743
  cc.choppiness = choppiness;                    // ocean.cpp:744
  cc.nz = 4 / (cc.scale * ctx->N);               // ocean.map.comp:77

  return DATUM_OCEAN_OK;
}

int datum_ocean_set_spectrum_format(datum_ocean_t ctx, int format)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_spectrum_format: null handle");

  if (format != DATUM_OCEAN_SPECTRUM_FP32 && format != DATUM_OCEAN_SPECTRUM_FP16 && format != DATUM_OCEAN_SPECTRUM_FP16_H0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_set_spectrum_format: unknown format");

  if (format != DATUM_OCEAN_SPECTRUM_FP32 && ctx->literal)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_set_spectrum_format: the handle is in the literal mode (the reference's fp32 arithmetic); switch it off first");

  if (format == DATUM_OCEAN_SPECTRUM_FP16_H0 && !ctx->h0h)
  {
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    HIPCHECK(ctx, hipMalloc(&ctx->h0h, (size_t)ctx->cascades * plane(ctx) * sizeof(unsigned int)));
  }

  ctx->half = (format != DATUM_OCEAN_SPECTRUM_FP32);
  ctx->h0half = (format == DATUM_OCEAN_SPECTRUM_FP16_H0);

  for(int c = 0; c < ctx->cascades; ++c)
  {
    ctx->scaledirty[c] = ctx->half;

    if (!ctx->half)
      ctx->casc[c].specscale = ctx->casc[c].specinv = ctx->casc[c].rowscale = 1.0f;
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_upload_state(datum_ocean_t ctx, int cascade, float const *h0, float const *phase)
{
  if (!ctx || !h0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_upload_state: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_upload_state: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  // the new state replaces the old one: updates queued against the old state are applied first so that
  // the other cascades keep them
  int rc = flush_pending(ctx, 0);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  HIPCHECK(ctx, hipMemcpyAsync(ctx->h0 + cascade * P, h0, P * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));

  unsigned int wild = 0;

  if (phase)
  {
    HIPCHECK(ctx, hipMemcpyAsync(ctx->phase + cascade * P, phase, P * sizeof(float), hipMemcpyHostToDevice, ctx->stream));

    // phases outside [0, 2 pi) take the general fmod kernel: looked for on the device, where the array now is
    HIPCHECK(ctx, hipMemsetAsync(ctx->absmax, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(ocean_phaserange_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->phase + cascade * P, P, ctx->absmax);
    HIPCHECK(ctx, hipGetLastError());
    HIPCHECK(ctx, hipMemcpyAsync(&wild, ctx->absmax, sizeof(wild), hipMemcpyDeviceToHost, ctx->stream));
  }
  else
    HIPCHECK(ctx, hipMemsetAsync(ctx->phase + cascade * P, 0, P * sizeof(float), ctx->stream));

  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));   // the host buffers are the caller's again

  ctx->wildphase[cascade] = wild != 0;

  ctx->uploaded[cascade] = true;
  ctx->scaledirty[cascade] = true;

  return DATUM_OCEAN_OK;
}

int datum_ocean_upload_seed(datum_ocean_t ctx, int cascade, float const *seed)
{
  if (!ctx || !seed)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_upload_seed: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_upload_seed: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  size_t const P = plane(ctx);

  if (!ctx->seed)
  {
    HIPCHECK(ctx, hipMalloc(&ctx->seed, ctx->cascades * P * sizeof(float2)));
    HIPCHECK(ctx, hipMemsetAsync(ctx->seed, 0, ctx->cascades * P * sizeof(float2), ctx->stream));
  }

  HIPCHECK(ctx, hipMemcpyAsync(ctx->seed + cascade * P, seed, P * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_rebuild_height(datum_ocean_t ctx, int cascade, float wavescale, float waveamplitude, float windspeed, float windx, float windy)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_rebuild_height: null handle");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_rebuild_height: cascade out of range");

  if (!ctx->seed)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_rebuild_height: no seed on the device (datum_ocean_upload_seed)");

  if (!(wavescale > 0))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_rebuild_height: wavescale must be positive");

  // queued updates belong to the old wave scale
  int rc = datum_ocean_set_cascade(ctx, cascade, wavescale, ctx->casc[cascade].choppiness);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  size_t const P = plane(ctx);

  hipLaunchKernelGGL(ocean_height_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->seed + cascade * P, ctx->h0 + cascade * P, ctx->N, wavescale, waveamplitude, windspeed, windx, windy);
  ctx->scaledirty[cascade] = true;
  HIPCHECK(ctx, hipGetLastError());

  if (!ctx->uploaded[cascade])
  {
    HIPCHECK(ctx, hipMemsetAsync(ctx->phase + cascade * P, 0, P * sizeof(float), ctx->stream));
    ctx->wildphase[cascade] = false;
    ctx->uploaded[cascade] = true;
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_read_height(datum_ocean_t ctx, int cascade, float *h0)
{
  if (!ctx || !h0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_height: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_height: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  size_t const P = plane(ctx);

  HIPCHECK(ctx, hipMemcpyAsync(h0, ctx->h0 + cascade * P, P * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_read_state(datum_ocean_t ctx, int cascade, float *phase)
{
  if (!ctx || !phase)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_state: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_state: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int rc = flush_pending(ctx, 0);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  HIPCHECK(ctx, hipMemcpyAsync(phase, ctx->phase + cascade * P, P * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

size_t datum_ocean_state_bytes(int resolution)
{
  return (size_t)resolution * resolution * (sizeof(float2) + sizeof(float));
}

int datum_ocean_park_state(datum_ocean_t ctx, int cascade, void *device_dst, size_t bytes, int *flags)
{
  if (!ctx || !device_dst || !flags)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_park_state: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_park_state: cascade out of range");

  if (bytes != datum_ocean_state_bytes(ctx->N))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_park_state: buffer must be datum_ocean_state_bytes()");

  if (!ctx->uploaded[cascade])
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_park_state: the cascade holds no state");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int rc = flush_pending(ctx, 0);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  HIPCHECK(ctx, hipMemcpyAsync(device_dst, ctx->h0 + cascade * P, P * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHECK(ctx, hipMemcpyAsync(static_cast<char*>(device_dst) + P * sizeof(float2), ctx->phase + cascade * P, P * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));

  *flags = ctx->wildphase[cascade] ? 1 : 0;

  return DATUM_OCEAN_OK;
}

int datum_ocean_resume_state(datum_ocean_t ctx, int cascade, void const *device_src, size_t bytes, int flags)
{
  if (!ctx || !device_src)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_resume_state: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_resume_state: cascade out of range");

  if (bytes != datum_ocean_state_bytes(ctx->N))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_resume_state: buffer must be datum_ocean_state_bytes()");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  // updates queued against the state that is being replaced are applied first (the other cascades keep them)
  int rc = flush_pending(ctx, 0);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  HIPCHECK(ctx, hipMemcpyAsync(ctx->h0 + cascade * P, device_src, P * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHECK(ctx, hipMemcpyAsync(ctx->phase + cascade * P, static_cast<char const*>(device_src) + P * sizeof(float2), P * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));

  ctx->wildphase[cascade] = (flags & 1) != 0;
  ctx->uploaded[cascade] = true;
  ctx->scaledirty[cascade] = true;

  return DATUM_OCEAN_OK;
}

int datum_ocean_update(datum_ocean_t ctx, float dt)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_update: null handle");

  ctx->pending.push_back(dt);

  if (ctx->pending.size() > 4 * MAX_PENDING)
  {
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    return flush_pending(ctx, MAX_PENDING);
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_displace(datum_ocean_t ctx)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_displace: null handle");

  for(int c = 0; c < ctx->cascades; ++c)
    if (!ctx->uploaded[c])
      return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_displace: a cascade has no state (datum_ocean_upload_state)");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int rc = DATUM_OCEAN_OK;

  if (!ctx->pending.empty())
  {
    rc = ensure_omega(ctx);

    if (rc == DATUM_OCEAN_OK)
      rc = flush_pending(ctx, (fusable(ctx) && !ctx->literal) ? MAX_PENDING : 0);
  }

  if (rc != DATUM_OCEAN_OK)
    return rc;

  if (ctx->literal)
  {
    // the reference's five dispatches, cascade by cascade (ocean.cpp:769-789); the phase was advanced by the general kernel above
    size_t const P = plane(ctx);

    for(int c = 0; c < ctx->cascades; ++c)
    {
      LiteralArgs a;
      a.h0 = ctx->h0 + c * P;
      a.phase = ctx->phase + c * P;
      a.h = ctx->litfields;
      a.hx = ctx->litfields + P;
      a.hy = ctx->litfields + 2 * P;
      a.weights = ctx->litweights;
      a.maps = reinterpret_cast<char*>(ctx->maps) + (size_t)c * map_cascade_bytes(ctx->N);
      a.N = ctx->N;
      a.scale = ctx->casc[c].scale;
      a.choppiness = ctx->casc[c].choppiness;

      HIPCHECK(ctx, launch_literal(a, ctx->stream));
    }

    return DATUM_OCEAN_OK;
  }

  rc = size_spectrum_scale(ctx);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  StepArgs a = make_args(ctx, (int)ctx->pending.size(), ctx->pending.data());
  ctx->pending.clear();

  bool const prof = ctx->profiling && ctx->profsteps < ctx->profmax && (ctx->profcalls++ % ctx->profstride) == 0;

  // the two passes group by group (cascade_group): row(g), column(g), row(g + 1), ...
  int const group = cascade_group(ctx);
  int const groups = (ctx->cascades + group - 1) / group;

  if (prof && ctx->events.size() < (size_t)4 * ctx->profmax * groups)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_displace: the cascade group changed while profiling");

  for(int g = 0; g < groups; ++g)
  {
    a.first = g * group;
    a.cascades = std::min(group, ctx->cascades - a.first);

    hipEvent_t *ev = prof ? &ctx->events[4 * ((size_t)ctx->profsteps * groups + g)] : nullptr;   // row start, row stop, column start, column stop

    hipError_t le = hipSuccess;

    DISPATCH_N(ctx->N, le = launch_rowpass<NN>(ctx, a, ev));
    HIPCHECK(ctx, le);

    DISPATCH_N(ctx->N, le = launch_colpass<NN>(ctx, a, ev ? ev + 2 : nullptr));
    HIPCHECK(ctx, le);
  }

  if (prof)
    ctx->profsteps += 1;

  return DATUM_OCEAN_OK;
}

int datum_ocean_gen(datum_ocean_t ctx, int cascade, datum_ocean_set const *set, int sizex, int sizey, void *vertices_device)
{
  if (!ctx || !set || !vertices_device)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_gen: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_gen: cascade out of range");

  if (sizex < 2 || sizey < 2)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_gen: mesh must be at least 2 x 2");

  if ((uintptr_t)vertices_device & 15)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_gen: vertex buffer must be 16-byte aligned");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  GenArgs g;
  g.set = *set;
  g.map = reinterpret_cast<float4 const*>(reinterpret_cast<char const*>(ctx->maps) + (size_t)cascade * map_cascade_bytes(ctx->N));
  g.vertices = (float*)vertices_device;
  gen_shape(g, ctx->N, sizex, sizey);

  // (one launch: the mesh as two half launches on two streams at once, joined by events, was measured and removed -- the fork and
  // join cost about 20 us per call on this runtime, 36 against 16 us: profiles/r04_gen_levers.txt)
  HIPCHECK(ctx, launch_gen(g, ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_payload_bytes(datum_ocean_t ctx, int format, size_t *bytes)
{
  if (!ctx || !bytes)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_payload_bytes: null argument");

  size_t const points = (size_t)ctx->cascades * plane(ctx);

  switch(format)
  {
    case DATUM_OCEAN_PAYLOAD_MAPS: *bytes = (size_t)ctx->cascades * map_cascade_bytes(ctx->N); break;
    case DATUM_OCEAN_PAYLOAD_XYZ32: *bytes = points * 12; break;
    case DATUM_OCEAN_PAYLOAD_XYZ16: *bytes = points * 8; break;
    default: return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_payload_bytes: unknown format");
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_pack_displacement(datum_ocean_t ctx, int format, void *payload_device, size_t bytes)
{
  if (!ctx || !payload_device)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_pack_displacement: null argument");

  size_t need = 0;

  int rc = datum_ocean_payload_bytes(ctx, format, &need);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  if (bytes < need)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_pack_displacement: payload buffer too small (datum_ocean_payload_bytes)");

  if ((uintptr_t)payload_device & 15)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_pack_displacement: payload buffer must be 16-byte aligned");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  return pack_into(ctx, format, payload_device, need);
}

/* -- the tile farm ------------------------------------------------------------------------------------------------------ */

int datum_ocean_farm_unique_id(void *id, size_t bytes)
{
  if (!id || bytes != DATUM_OCEAN_FARM_ID_BYTES)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_unique_id: the id is DATUM_OCEAN_FARM_ID_BYTES bytes");

  static_assert(sizeof(ncclUniqueId) == DATUM_OCEAN_FARM_ID_BYTES, "ncclUniqueId");

  std::string why;
  RcclApi *api = rccl_api(&why);

  if (!api)
    return fail(nullptr, DATUM_OCEAN_EUNSUPPORTED, ("datum_ocean_farm_unique_id: no RCCL library could be opened: " + why).c_str());

  ncclUniqueId uid;

  RCCLCHECK(nullptr, api, nullptr, api->GetUniqueId(&uid));

  memcpy(id, &uid, sizeof(uid));

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_init(datum_ocean_t ctx, void const *id, size_t idbytes, int rank, int world, int format, int slots)
{
  if (!ctx || !id || idbytes != DATUM_OCEAN_FARM_ID_BYTES)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_init: null argument, or an id that is not DATUM_OCEAN_FARM_ID_BYTES bytes");

  if (world < 1 || rank < 0 || rank >= world)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_init: rank outside [0, world)");

  if (slots < 1 || slots > 8)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_init: 1 to 8 slots (2 = double-buffered)");

  if (ctx->farm)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_init: the handle already farms (datum_ocean_farm_shutdown first)");

  size_t bytes = 0;

  int rc = datum_ocean_payload_bytes(ctx, format, &bytes);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  std::string why;
  RcclApi *api = rccl_api(&why);

  if (!api)
    return fail(ctx, DATUM_OCEAN_EUNSUPPORTED, ("datum_ocean_farm_init: no RCCL library could be opened: " + why).c_str());

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  Farm *f = new (std::nothrow) Farm;
  if (!f)
    return fail(ctx, DATUM_OCEAN_ENOMEM, "datum_ocean_farm_init: out of host memory");

  ctx->farm = f;

  f->api = api;
  f->rank = rank;
  f->world = world;
  f->format = format;
  f->bytes = bytes;

  #define FARMCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int rc_ = fail(ctx, (int)e_, #call); farm_teardown(ctx); return rc_; } } while(0)

  FARMCHECK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));

  try { f->slots.resize(slots); } catch (...) { farm_teardown(ctx); return fail(ctx, DATUM_OCEAN_ENOMEM, "datum_ocean_farm_init: out of host memory"); }

  for(auto &sl : f->slots)
  {
    FARMCHECK(hipMalloc(&sl.payload, bytes));
    FARMCHECK(hipMalloc(&sl.gathered, bytes * world));
    FARMCHECK(hipEventCreateWithFlags(&sl.packed, hipEventDisableTiming));
    FARMCHECK(hipEventCreate(&sl.start));
    FARMCHECK(hipEventCreate(&sl.done));
    FARMCHECK(hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));
  }

  #undef FARMCHECK

  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));

  // (collective over the ranks: returns when every rank of the farm has called it)
  ncclResult_t r = api->CommInitRank(&f->comm, world, uid, rank);

  if (r != ncclSuccess)
  {
    int rc_ = fail_comm(ctx, api, nullptr, r, "ncclCommInitRank");
    f->comm = nullptr;
    farm_teardown(ctx);
    return rc_;
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_shutdown(datum_ocean_t ctx)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_shutdown: null handle");

  (void)hipSetDevice(ctx->device);

  farm_teardown(ctx);

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_info(datum_ocean_t ctx, int *rank, int *world, int *format, size_t *payload_bytes, int *slots, int *rccl_version)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_info: null handle");

  if (!ctx->farm)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_info: datum_ocean_farm_init first");

  Farm const *f = ctx->farm;

  if (rank) *rank = f->rank;
  if (world) *world = f->world;
  if (format) *format = f->format;
  if (payload_bytes) *payload_bytes = f->bytes;
  if (slots) *slots = (int)f->slots.size();

  if (rccl_version)
  {
    *rccl_version = 0;
    (void)f->api->GetVersion(rccl_version);
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_gather(datum_ocean_t ctx, int *slot)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_gather: null handle");

  if (!ctx->farm)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_gather: datum_ocean_farm_init first");

  Farm *f = ctx->farm;

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int const s = f->head;
  FarmSlot &sl = f->slots[s];

  // a consumer on a stream of its own took this slot's result and never said when it was done reading: the collective below
  // would overwrite what it may still be reading, and nothing orders the two
  if (sl.held)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_gather: the next slot's result was handed to another stream and not released (datum_ocean_farm_release)");

  // WAR on the payload: the collective that last READ this slot's payload must have finished before the pack overwrites it
  if (sl.launched)
    HIPCHECK(ctx, hipStreamWaitEvent(ctx->stream, sl.done, 0));

  // the pack, on the handle's stream behind the last displace
  int rc = pack_into(ctx, f->format, sl.payload, f->bytes);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  HIPCHECK(ctx, hipEventRecord(sl.packed, ctx->stream));

  // the collective, on the communication stream: behind the pack, and behind the consumer that may still read gathered[s]
  HIPCHECK(ctx, hipStreamWaitEvent(f->stream, sl.packed, 0));

  if (sl.busy)
  {
    HIPCHECK(ctx, hipStreamWaitEvent(f->stream, sl.consumed, 0));
    sl.busy = false;
  }

  HIPCHECK(ctx, hipEventRecord(sl.start, f->stream));

  RCCLCHECK(ctx, f->api, f->comm, f->api->AllGather(sl.payload, sl.gathered, f->bytes, ncclInt8, f->comm, f->stream));

  HIPCHECK(ctx, hipEventRecord(sl.done, f->stream));

  sl.launched = true;

  f->head = (s + 1) % (int)f->slots.size();
  f->gathers += 1;

  if (slot)
    *slot = s;

  return DATUM_OCEAN_OK;
}

namespace
{
  static int farm_slot(datum_ocean_ctx *ctx, int slot, char const *who, FarmSlot **out)
  {
    if (!ctx)
      return fail(nullptr, DATUM_OCEAN_EINVAL, who);

    if (!ctx->farm)
      return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_*: datum_ocean_farm_init first");

    if (slot < 0 || slot >= (int)ctx->farm->slots.size())
      return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_*: no such slot");

    if (!ctx->farm->slots[slot].launched)
      return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_*: nothing was gathered into this slot yet");

    *out = &ctx->farm->slots[slot];

    return DATUM_OCEAN_OK;
  }
}

int datum_ocean_farm_result(datum_ocean_t ctx, int slot, void *hip_stream, int on_handle_stream, void **gathered_device, size_t *bytes)
{
  FarmSlot *sl = nullptr;

  int rc = farm_slot(ctx, slot, "datum_ocean_farm_result: null handle", &sl);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  if (!gathered_device)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_result: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamWaitEvent(on_handle_stream ? ctx->stream : (hipStream_t)hip_stream, sl->done, 0));

  // (a reader on the handle's stream is ordered before the slot's next pack, and with it before its next collective; any
  // other stream has to release)
  if (!on_handle_stream && (hipStream_t)hip_stream != ctx->stream)
    sl->held = true;

  *gathered_device = sl->gathered;

  if (bytes)
    *bytes = ctx->farm->bytes * ctx->farm->world;

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_release(datum_ocean_t ctx, int slot, void *hip_stream, int on_handle_stream)
{
  FarmSlot *sl = nullptr;

  int rc = farm_slot(ctx, slot, "datum_ocean_farm_release: null handle", &sl);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipEventRecord(sl->consumed, on_handle_stream ? ctx->stream : (hipStream_t)hip_stream));

  sl->busy = true;
  sl->held = false;

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_query(datum_ocean_t ctx, int slot)
{
  FarmSlot *sl = nullptr;

  int rc = farm_slot(ctx, slot, "datum_ocean_farm_query: null handle", &sl);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipError_t const e = hipEventQuery(sl->done);

  if (e == hipErrorNotReady)
  {
    (void)hipGetLastError();
    return DATUM_OCEAN_ENOTREADY;
  }

  HIPCHECK(ctx, e);

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_wait(datum_ocean_t ctx, int slot, float *collective_ms)
{
  FarmSlot *sl = nullptr;

  int rc = farm_slot(ctx, slot, "datum_ocean_farm_wait: null handle", &sl);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipEventSynchronize(sl->done));

  if (collective_ms)
    HIPCHECK(ctx, hipEventElapsedTime(collective_ms, sl->start, sl->done));

  return DATUM_OCEAN_OK;
}

namespace
{
  // a stream on the compute units [first, first + count) of the device's CU-mask order (bit i: CU i / 8 of XCD i % 8 on an MI355X,
  // tools/cumask_probe.py); count == 0: an ordinary stream
  hipError_t make_stream(hipStream_t *stream, int cus, int first, int count)
  {
    if (count == 0)
      return hipStreamCreateWithFlags(stream, hipStreamNonBlocking);

    std::vector<uint32_t> words((cus + 31) / 32, 0u);

    for(int i = first; i < first + count; ++i)
      words[i / 32] |= 1u << (i % 32);

    return hipExtStreamCreateWithCUMask(stream, (uint32_t)words.size(), words.data());
  }
}

int datum_ocean_farm_partition(datum_ocean_t ctx, int comm_cus)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_partition: null handle");

  Farm *f = ctx->farm;

  if (!f)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_partition: the handle does not farm (datum_ocean_farm_init first)");

  // DATUM_OCEAN_FARM_PARTITION_AUTO: an eighth of the device in whole shares of 8 (one compute unit per XCD and share: 32 of an MI355X's
  // 256; none on a device with fewer than 64 compute units, where the call then leaves both streams on the whole device)
  if (comm_cus == DATUM_OCEAN_FARM_PARTITION_AUTO)
    comm_cus = (ctx->cus / 64) * 8;

  if (comm_cus < 0 || comm_cus % 8 != 0 || comm_cus > ctx->cus / 2)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_farm_partition: comm_cus must be 0, DATUM_OCEAN_FARM_PARTITION_AUTO or a multiple of 8 (one share per XCD), at most half the device");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  // nothing in flight on either stream while they are replaced (the slots' events stay valid: they have completed)
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->ownstream));
  HIPCHECK(ctx, hipStreamSynchronize(f->stream));

  hipStream_t comm = nullptr, own = nullptr;

  HIPCHECK(ctx, make_stream(&comm, ctx->cus, 0, comm_cus));

  hipError_t const e = make_stream(&own, ctx->cus, comm_cus, comm_cus ? ctx->cus - comm_cus : 0);

  if (e != hipSuccess)
  {
    (void)hipStreamDestroy(comm);
    return fail(ctx, (int)e, "datum_ocean_farm_partition: hipExtStreamCreateWithCUMask");
  }

  bool const onown = ctx->stream == ctx->ownstream;

  (void)hipStreamDestroy(f->stream);
  (void)hipStreamDestroy(ctx->ownstream);

  f->stream = comm;
  f->commcus = comm_cus;
  ctx->ownstream = own;

  if (onown)
    ctx->stream = own;

  return DATUM_OCEAN_OK;
}

int datum_ocean_farm_stream_flags(datum_ocean_t ctx, unsigned int *communication_stream_flags, unsigned int *own_stream_flags)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_farm_stream_flags: null handle");

  if (!ctx->farm)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_farm_stream_flags: the handle does not farm (datum_ocean_farm_init first)");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  // (CU-masked streams come without a flags argument: what the runtime made of them is what decides whether a null-stream operation of the
  // caller's serialises the two streams -- include/datum_ocean_hip.h, datum_ocean_farm_partition)
  if (communication_stream_flags)
    HIPCHECK(ctx, hipStreamGetFlags(ctx->farm->stream, communication_stream_flags));

  if (own_stream_flags)
    HIPCHECK(ctx, hipStreamGetFlags(ctx->ownstream, own_stream_flags));

  return DATUM_OCEAN_OK;
}

int datum_ocean_own_stream(datum_ocean_t ctx, void **hip_stream)
{
  if (!ctx || !hip_stream)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_own_stream: null argument");

  *hip_stream = ctx->ownstream;

  return DATUM_OCEAN_OK;
}

int datum_ocean_read_maps(datum_ocean_t ctx, int cascade, float *maps)
{
  if (!ctx || !maps)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_maps: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_read_maps: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  size_t const P = plane(ctx);

  // the device layout is the module's own (ocean_kernels.hip: map_compact_a / map_compact_b); hand out the reference's logical
  // image, [layer][y][x] RGBA32F with the .w channels zero (map.comp:79-80)
  size_t const bytes = map_cascade_bytes(ctx->N);

  std::vector<float> raw;
  try { raw.resize(bytes / sizeof(float)); } catch (...) { return fail(ctx, DATUM_OCEAN_ENOMEM, "datum_ocean_read_maps: out of host memory"); }

  HIPCHECK(ctx, hipMemcpyAsync(raw.data(), reinterpret_cast<char const*>(ctx->maps) + (size_t)cascade * bytes, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  int const N = ctx->N;
  float4 *out = reinterpret_cast<float4*>(maps);

  for(int y = 0; y < N; ++y)
  {
    for(int x = 0; x < N; ++x)
    {
      float const *a = raw.data() + map_compact_a(N, y, x) / sizeof(float);
      float const *b = raw.data() + map_compact_b(N, y, x) / sizeof(float);

      out[(size_t)y * N + x] = make_float4(a[0], a[1], a[2], 0.0f);
      out[P + (size_t)y * N + x] = make_float4(a[3], b[0], b[1], 0.0f);
    }
  }

  return DATUM_OCEAN_OK;
}

}   // extern "C"

// a cascade's maps as the reference's 2-layer RGBA32F image (datum_ocean_export_maps); one thread per texel.  LAYOUT (up to 1024^2): in the
// order of the map layout like the pack kernel -- consecutive lanes read consecutive parts A and B of a patch, 16-byte stores in runs of
// a patch row (1024^2: 13.0 -> 9.3 us); otherwise in the order of the image -- from 2048^2 up, beyond the Infinity Cache, whole lines
// written count for more than whole lines read (2048^2: 35.4 against 38.0 us, 4096^2: 228 against 240 us; profiles/r05_pack.txt)
template<bool LAYOUT>
__global__ void __launch_bounds__(256) ocean_export_kernel(char const *maps, int N, float4 *dst, ocean::PackShape sh)
{
  size_t const P = (size_t)N * N;

  for(size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < P; r += (size_t)gridDim.x * blockDim.x)
  {
    size_t i, oa, ob;

    if constexpr (LAYOUT)
    {
      int const patch = (int)(r >> 4), j = (int)(r & 15);

      oa = (size_t)patch * ocean::MAP_PATCH_BYTES + j * 16;
      ob = (size_t)patch * ocean::MAP_PATCH_BYTES + 256 + j * 8;

      int const band = patch >> sh.bandpatches2, pp = patch & ((1 << sh.bandpatches2) - 1);
      int const y = ((pp >> sh.bp2) << (4 - sh.pw2)) + (j >> sh.pw2);
      int const x = (band << sh.b2) + ((pp & ((1 << sh.bp2) - 1)) << sh.pw2) + (j & ((1 << sh.pw2) - 1));

      i = ((size_t)y << sh.n2) + x;
    }
    else
    {
      int const y = (int)(r >> sh.n2), x = (int)(r & (N - 1));

      oa = ocean::map_compact_a(N, y, x);
      ob = ocean::map_compact_b(N, y, x);
      i = r;
    }

    float4 const a = *reinterpret_cast<float4 const*>(maps + oa);
    float2 const b = *reinterpret_cast<float2 const*>(maps + ob);

    dst[i] = make_float4(a.x, a.y, a.z, 0.0f);
    dst[P + i] = make_float4(a.w, b.x, b.y, 0.0f);
  }
}

extern "C"
{

int datum_ocean_export_maps(datum_ocean_t ctx, int cascade, void *device_dst, size_t bytes)
{
  if (!ctx || !device_dst)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_export_maps: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_export_maps: cascade out of range");

  if (bytes < 2 * plane(ctx) * sizeof(float4))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_export_maps: destination smaller than 2 * N * N * 16 bytes");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  size_t const P = plane(ctx);
  int const blocks = (int)((P + 255) / 256 < 4096 ? (P + 255) / 256 : 4096);

  char const *block = reinterpret_cast<char const*>(ctx->maps) + (size_t)cascade * map_cascade_bytes(ctx->N);

  if (ctx->N <= 1024)
    hipLaunchKernelGGL(ocean_export_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, block, ctx->N, static_cast<float4*>(device_dst), pack_shape(ctx->N));
  else
    hipLaunchKernelGGL(ocean_export_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, block, ctx->N, static_cast<float4*>(device_dst), pack_shape(ctx->N));

  HIPCHECK(ctx, hipGetLastError());

  return DATUM_OCEAN_OK;
}

int datum_ocean_set_literal_transform(datum_ocean_t ctx, int on)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_literal_transform: null handle");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  // (profiling samples and the fp16 spectrum format belong to the fused kernels: refused together with the literal mode rather than ignored)
  if (on && ctx->profiling)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_set_literal_transform: a profile is open (datum_ocean_profile_end first): the literal mode's dispatches are not sampled");

  if (on && ctx->half)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_set_literal_transform: the handle stores an fp16 spectrum (datum_ocean_set_spectrum_format): the literal mode is the reference's fp32 arithmetic");

  if (on && !ctx->litfields)
  {
    size_t const P = plane(ctx);

    int stages = 0;
    while ((1 << stages) < ctx->N)
      ++stages;

    std::vector<float> weights;
    try { weights.resize((size_t)ctx->N * 2 * stages); } catch (...) { return fail(ctx, DATUM_OCEAN_ENOMEM, "datum_ocean_set_literal_transform: out of host memory"); }

    int rc = datum_ocean_reference_weights(ctx->N, weights.data());
    if (rc != DATUM_OCEAN_OK)
      return rc;

    // both buffers or neither: a failure between the two must not leave one behind for the next call to allocate over
    float *lw = nullptr;
    float2 *lf = nullptr;

    hipError_t e = hipMalloc(&lw, weights.size() * sizeof(float));

    if (e == hipSuccess)
      e = hipMalloc(&lf, 3 * P * sizeof(float2));

    if (e == hipSuccess)
      e = hipMemcpyAsync(lw, weights.data(), weights.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);

    if (e == hipSuccess)
      e = hipStreamSynchronize(ctx->stream);       // (the host vector goes out of scope)

    if (e != hipSuccess)
    {
      (void)hipFree(lw);
      (void)hipFree(lf);

      return fail(ctx, (int)e, "datum_ocean_set_literal_transform: buffers of the literal mode");
    }

    ctx->litweights = lw;
    ctx->litfields = lf;
  }

  ctx->literal = on != 0;

  return DATUM_OCEAN_OK;
}

int datum_ocean_set_cascade_group(datum_ocean_t ctx, int cascades_per_launch)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_cascade_group: null handle");

  if (cascades_per_launch < 0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_set_cascade_group: negative group");

  if (ctx->profiling)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_set_cascade_group: a profile is open (its samples are per group)");

  ctx->cascadegroup = cascades_per_launch;

  return DATUM_OCEAN_OK;
}

int datum_ocean_cascade_group(datum_ocean_t ctx, int *cascades_per_launch, int *launches_per_pass)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_cascade_group: null handle");

  int const group = cascade_group(ctx);

  if (cascades_per_launch) *cascades_per_launch = group;
  if (launches_per_pass) *launches_per_pass = (ctx->cascades + group - 1) / group;

  return DATUM_OCEAN_OK;
}

namespace
{
  bool handle_streams_maps(datum_ocean_ctx const *ctx)
  {
    if (ctx->N >= 4096)
      return true;

    if (ctx->N < 1024)
      return false;

    if (ctx->mappolicy != DATUM_OCEAN_MAPS_AUTO)
      return ctx->mappolicy == DATUM_OCEAN_MAPS_STREAMED;

    return maps_stream(ctx->N, ctx->cascades, ctx->half) || (ctx->farm && ctx->farm->world > 1);
  }
}

int datum_ocean_set_map_store_policy(datum_ocean_t ctx, int policy)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_set_map_store_policy: null handle");

  if (policy != DATUM_OCEAN_MAPS_AUTO && policy != DATUM_OCEAN_MAPS_WRITTEN_THROUGH && policy != DATUM_OCEAN_MAPS_STREAMED)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_set_map_store_policy: unknown policy");

  if (ctx->profiling)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_set_map_store_policy: a profile is open (the policy can change the cascade groups its samples are per)");

  ctx->mappolicy = policy;

  return DATUM_OCEAN_OK;
}

int datum_ocean_map_store_policy(datum_ocean_t ctx, int *policy, int *streamed)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_map_store_policy: null handle");

  if (policy) *policy = ctx->mappolicy;
  if (streamed) *streamed = handle_streams_maps(ctx) ? 1 : 0;

  return DATUM_OCEAN_OK;
}

int datum_ocean_abi_version(void)
{
  return DATUM_OCEAN_ABI_VERSION;
}

int datum_ocean_sync(datum_ocean_t ctx)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_sync: null handle");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_wait_event(datum_ocean_t ctx, void *hip_event)
{
  if (!ctx || !hip_event)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_wait_event: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));

  return DATUM_OCEAN_OK;
}

int datum_ocean_signal(datum_ocean_t ctx, void **hip_event)
{
  if (!ctx || !hip_event)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_signal: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  if (!ctx->complete)
    HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->complete, hipEventDisableTiming));

  HIPCHECK(ctx, hipEventRecord(ctx->complete, ctx->stream));

  *hip_event = ctx->complete;

  return DATUM_OCEAN_OK;
}

int datum_ocean_on_complete(datum_ocean_t ctx, void (*callback)(void *user), void *user)
{
  if (!ctx || !callback)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_on_complete: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipLaunchHostFunc(ctx->stream, callback, user));

  return DATUM_OCEAN_OK;
}

int datum_ocean_query(datum_ocean_t ctx)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_query: null handle");

  if (!ctx->complete)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_query: datum_ocean_signal first");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipError_t const e = hipEventQuery(ctx->complete);

  if (e == hipErrorNotReady)
  {
    (void)hipGetLastError();
    return DATUM_OCEAN_ENOTREADY;
  }

  HIPCHECK(ctx, e);

  return DATUM_OCEAN_OK;
}

int datum_ocean_import_memory_fd(datum_ocean_t ctx, int fd, size_t bytes, void **device_ptr)
{
  if (!ctx || !device_ptr)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_import_memory_fd: null argument");

  *device_ptr = nullptr;

  if (fd < 0 || bytes == 0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_import_memory_fd: bad descriptor or size");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipExternalMemoryHandleDesc desc = {};
  desc.type = hipExternalMemoryHandleTypeOpaqueFd;       // VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT
  desc.handle.fd = fd;
  desc.size = bytes;

  hipExternalMemory_t memory = nullptr;

  HIPCHECK(ctx, hipImportExternalMemory(&memory, &desc));

  hipExternalMemoryBufferDesc buffer = {};
  buffer.offset = 0;
  buffer.size = bytes;

  void *ptr = nullptr;

  hipError_t e = hipExternalMemoryGetMappedBuffer(&ptr, memory, &buffer);

  if (e != hipSuccess || !ptr)
  {
    (void)hipDestroyExternalMemory(memory);
    return fail(ctx, e != hipSuccess ? (int)e : DATUM_OCEAN_ENOMEM, "hipExternalMemoryGetMappedBuffer");
  }

  ctx->importedmemory.push_back({ memory, ptr, bytes });

  *device_ptr = ptr;

  return DATUM_OCEAN_OK;
}

int datum_ocean_release_memory(datum_ocean_t ctx, void *device_ptr)
{
  if (!ctx || !device_ptr)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_release_memory: null argument");

  for(size_t i = 0; i < ctx->importedmemory.size(); ++i)
  {
    if (ctx->importedmemory[i].ptr == device_ptr)
    {
      HIPCHECK(ctx, hipSetDevice(ctx->device));
      HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));    // nothing enqueued may still write it

      // (a VkBuffer commonly sits at an offset inside its VkDeviceMemory: the maps may be bound anywhere inside the block)
      char const *lo = static_cast<char const*>(device_ptr), *hi = lo + ctx->importedmemory[i].bytes;

      if (reinterpret_cast<char const*>(ctx->maps) >= lo && reinterpret_cast<char const*>(ctx->maps) < hi)
        ctx->maps = ctx->ownmaps;

      hipError_t e = hipDestroyExternalMemory(ctx->importedmemory[i].memory);

      ctx->importedmemory.erase(ctx->importedmemory.begin() + i);

      if (e != hipSuccess)
        return fail(ctx, (int)e, "hipDestroyExternalMemory");

      return DATUM_OCEAN_OK;
    }
  }

  return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_release_memory: not a pointer this handle imported");
}

int datum_ocean_import_semaphore_fd(datum_ocean_t ctx, int fd, void **semaphore)
{
  if (!ctx || !semaphore)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_import_semaphore_fd: null argument");

  *semaphore = nullptr;

  if (fd < 0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_import_semaphore_fd: bad descriptor");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipExternalSemaphoreHandleDesc desc = {};
  desc.type = hipExternalSemaphoreHandleTypeOpaqueFd;    // VK_EXTERNAL_SEMAPHORE_HANDLE_TYPE_OPAQUE_FD_BIT
  desc.handle.fd = fd;

  hipExternalSemaphore_t sem = nullptr;

  hipError_t const e = hipImportExternalSemaphore(&sem, &desc);

  if (e == hipErrorNotSupported)
  {
    (void)hipGetLastError();
    return fail(ctx, DATUM_OCEAN_EUNSUPPORTED, "datum_ocean_import_semaphore_fd: this HIP runtime has no external semaphores (hipImportExternalSemaphore: not supported); "
                                               "bridge rendercomplete on the host: datum_ocean_on_complete or datum_ocean_signal + datum_ocean_query");
  }

  HIPCHECK(ctx, e);

  ctx->importedsemaphores.push_back(sem);

  *semaphore = sem;

  return DATUM_OCEAN_OK;
}

namespace
{
  static bool owns_semaphore(datum_ocean_ctx *ctx, void *semaphore)
  {
    for(hipExternalSemaphore_t s : ctx->importedsemaphores)
      if (s == semaphore)
        return true;

    return false;
  }
}

int datum_ocean_release_semaphore(datum_ocean_t ctx, void *semaphore)
{
  if (!ctx || !semaphore || !owns_semaphore(ctx, semaphore))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_release_semaphore: not a semaphore this handle imported");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  for(size_t i = 0; i < ctx->importedsemaphores.size(); ++i)
    if (ctx->importedsemaphores[i] == semaphore)
      ctx->importedsemaphores.erase(ctx->importedsemaphores.begin() + i--);

  HIPCHECK(ctx, hipDestroyExternalSemaphore((hipExternalSemaphore_t)semaphore));

  return DATUM_OCEAN_OK;
}

int datum_ocean_signal_external(datum_ocean_t ctx, void *semaphore)
{
  if (!ctx || !semaphore || !owns_semaphore(ctx, semaphore))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_signal_external: not a semaphore this handle imported");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipExternalSemaphore_t sem = (hipExternalSemaphore_t)semaphore;
  hipExternalSemaphoreSignalParams params = {};

  HIPCHECK(ctx, hipSignalExternalSemaphoresAsync(&sem, &params, 1, ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_wait_external(datum_ocean_t ctx, void *semaphore)
{
  if (!ctx || !semaphore || !owns_semaphore(ctx, semaphore))
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_wait_external: not a semaphore this handle imported");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  hipExternalSemaphore_t sem = (hipExternalSemaphore_t)semaphore;
  hipExternalSemaphoreWaitParams params = {};

  HIPCHECK(ctx, hipWaitExternalSemaphoresAsync(&sem, &params, 1, ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_device_alloc(datum_ocean_t ctx, size_t bytes, void **device_ptr)
{
  if (!ctx || !device_ptr || bytes == 0)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_device_alloc: bad argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipMalloc(device_ptr, bytes));

  return DATUM_OCEAN_OK;
}

int datum_ocean_device_free(datum_ocean_t ctx, void *device_ptr)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_device_free: null handle");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHECK(ctx, hipFree(device_ptr));

  return DATUM_OCEAN_OK;
}

int datum_ocean_device_write(datum_ocean_t ctx, void *device_dst, void const *host_src, size_t bytes)
{
  if (!ctx || !device_dst || !host_src)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_device_write: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_device_read(datum_ocean_t ctx, void *host_dst, void const *device_src, size_t bytes)
{
  if (!ctx || !host_dst || !device_src)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_device_read: null argument");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

char const *datum_ocean_last_error(datum_ocean_t ctx)
{
  return ctx ? ctx->error.c_str() : g_error.c_str();
}

int datum_ocean_reference_weights(int resolution, float *weights)
{
  if (!supported(resolution) || !weights)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_reference_weights: bad argument");

  int stages = 0;
  while ((1 << stages) < resolution)
    ++stages;

  float const pi = 3.14159265358979323846f;

  // lane i, stage n: angle = -2 pi i / 2^(n+1), all in fp32 as the reference evaluates it
  for(int i = 0; i < resolution; ++i)
  {
    float *row = weights + (size_t)i * 2 * stages;

    for(int n = 0; n < stages; ++n)
    {
      float angle = -2 * pi * i / (2 * powf(2.0f, (float)n));

      row[2*n+0] = cosf(angle);
      row[2*n+1] = sinf(angle);
    }
  }

  return DATUM_OCEAN_OK;
}

int datum_ocean_debug_sim(datum_ocean_t ctx, int cascade, float *h, float *hx, float *hy)
{
  if (!ctx || !h || !hx || !hy)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_debug_sim: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_debug_sim: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int rc = flush_pending(ctx, 0);
  if (rc == DATUM_OCEAN_OK)
    rc = ensure_scratch(ctx);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  StepArgs a = make_args(ctx, 0, nullptr);

  if (ctx->wildphase[cascade])
    hipLaunchKernelGGL(ocean_sim_kernel<true>, dim3(1024), dim3(256), 0, ctx->stream, a, ctx->N, cascade, ctx->scratch, ctx->scratch + P, ctx->scratch + 2 * P);
  else
    hipLaunchKernelGGL(ocean_sim_kernel<false>, dim3(1024), dim3(256), 0, ctx->stream, a, ctx->N, cascade, ctx->scratch, ctx->scratch + P, ctx->scratch + 2 * P);
  HIPCHECK(ctx, hipGetLastError());

  HIPCHECK(ctx, hipMemcpyAsync(h, ctx->scratch, P * sizeof(cf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipMemcpyAsync(hx, ctx->scratch + P, P * sizeof(cf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipMemcpyAsync(hy, ctx->scratch + 2 * P, P * sizeof(cf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_debug_rowpass(datum_ocean_t ctx, int cascade, float *c, float *d)
{
  if (!ctx || !c || !d)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_debug_rowpass: null argument");

  if (cascade < 0 || cascade >= ctx->cascades)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_debug_rowpass: cascade out of range");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int rc = ensure_scratch(ctx);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  size_t const P = plane(ctx);

  // The work spectrum belongs to a launch, not to a cascade (slot cascade - first of the group: ocean_kernels.hip), and the last displace may
  // have left another cascade's values in the slot: the row pass of this one cascade once more -- no update pending, so the phase is neither
  // advanced nor stored and the values are those of the last displace -- into slot 0
  rc = size_spectrum_scale(ctx);
  if (rc != DATUM_OCEAN_OK)
    return rc;

  {
    StepArgs a = make_args(ctx, 0, nullptr);
    a.first = cascade;
    a.cascades = 1;

    hipError_t le = hipSuccess;
    DISPATCH_N(ctx->N, le = launch_rowpass<NN>(ctx, a, nullptr));
    HIPCHECK(ctx, le);
  }

  if (ctx->half)
    hipLaunchKernelGGL(ocean_unpack_kernel<true>, dim3(1024), dim3(256), 0, ctx->stream, static_cast<ch const*>(ctx->spec), ctx->N, ctx->casc[cascade].specinv, ctx->scratch, ctx->scratch + P);
  else
    hipLaunchKernelGGL(ocean_unpack_kernel<false>, dim3(1024), dim3(256), 0, ctx->stream, static_cast<cd const*>(ctx->spec), ctx->N, 1.0f, ctx->scratch, ctx->scratch + P);
  HIPCHECK(ctx, hipGetLastError());
  HIPCHECK(ctx, hipMemcpyAsync(c, ctx->scratch, P * sizeof(cf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipMemcpyAsync(d, ctx->scratch + P, P * sizeof(cf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  return DATUM_OCEAN_OK;
}

int datum_ocean_profile_begin(datum_ocean_t ctx, int max_steps, int stride)
{
  if (!ctx || max_steps < 1 || stride < 1)
    return fail(ctx, DATUM_OCEAN_EINVAL, "datum_ocean_profile_begin: bad argument");

  if (ctx->literal)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_profile_begin: the handle is in the literal mode, whose dispatches are not sampled");

  HIPCHECK(ctx, hipSetDevice(ctx->device));

  int const groups = (ctx->cascades + cascade_group(ctx) - 1) / cascade_group(ctx);

  while (ctx->events.size() < (size_t)4 * max_steps * groups)
  {
    hipEvent_t e;
    HIPCHECK(ctx, hipEventCreate(&e));
    ctx->events.push_back(e);
  }

  ctx->profiling = true;
  ctx->profmax = max_steps;
  ctx->profsteps = 0;
  ctx->profstride = stride;
  ctx->profcalls = 0;

  return DATUM_OCEAN_OK;
}

int datum_ocean_profile_end(datum_ocean_t ctx, double *rowpass_ms, double *colpass_ms, int *steps)
{
  if (!ctx || !ctx->profiling)
    return fail(ctx, DATUM_OCEAN_ESTATE, "datum_ocean_profile_end: profiling was not started");

  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));

  double row = 0, col = 0;

  // a sampled step = one launch of either pass per cascade group: the sums over a step's launches
  int const groups = (ctx->cascades + cascade_group(ctx) - 1) / cascade_group(ctx);

  for(int i = 0; i < ctx->profsteps * groups; ++i)
  {
    float ms;
    HIPCHECK(ctx, hipEventElapsedTime(&ms, ctx->events[4*i+0], ctx->events[4*i+1]));
    row += ms;
    HIPCHECK(ctx, hipEventElapsedTime(&ms, ctx->events[4*i+2], ctx->events[4*i+3]));
    col += ms;
  }

  int n = ctx->profsteps;

  if (rowpass_ms) *rowpass_ms = n ? row / n : 0.0;
  if (colpass_ms) *colpass_ms = n ? col / n : 0.0;
  if (steps) *steps = n;

  ctx->profiling = false;

  return DATUM_OCEAN_OK;
}

int datum_ocean_algorithmic_bytes(datum_ocean_t ctx, double *rowpass_bytes, double *colpass_bytes)
{
  if (!ctx)
    return fail(nullptr, DATUM_OCEAN_EINVAL, "datum_ocean_algorithmic_bytes: null handle");

  double pts = (double)plane(ctx) * ctx->cascades;

  // the ALGORITHM's bytes as the reference states it, three transforms (bench.py's roofline):
  // h0 8 + phase in 4 + phase out 4 + spectrum out 24 | spectrum in 24 + two RGBA32F layers 32.
  // The packed step moves 16 instead of 24 spectrum bytes each way and 24-byte texels: 32 + 40 = 72 B/pt of HBM traffic.
  // fp16-stored spectrum: 4 + 4 + 4 + 12 | 12 + 32 = 68 B/pt; this build keeps h0 in fp32 and
  // moves 8 + 4 + 4 + 8 | 8 + 24 = 56 B/pt in the format FP16, and with h0 as halves too (FP16_H0) 4 + 4 + 4 + 8 | 8 + 24 = 52 B/pt.
  if (rowpass_bytes) *rowpass_bytes = (ctx->half ? 24.0 : 40.0) * pts;
  if (colpass_bytes) *colpass_bytes = (ctx->half ? 44.0 : 56.0) * pts;

  return DATUM_OCEAN_OK;
}

}
