/* datum_ocean_hip.h -- C ABI of the MI355X (gfx950) ocean compute module.
 *
 * This is the drop-in boundary: the entry points below replace what
 * datum's src/renderer/ocean.cpp does through Vulkan compute between ":757" and ":803"
 * (five bind_pipeline + dispatch pairs: ocean.sim, ocean.fftx, ocean.ffty, ocean.map,
 * ocean.gen) plus the per-tick CPU phase advance of update_ocean (":217-236"), the
 * twiddle/spectrum/displacement-map allocations of prepare_ocean_context (":686-711") and the
 * per-frame OceanSet upload (":729-749").  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success; a negative DATUM_OCEAN_E* code of this module (misuse, or ENOTREADY from the
 *     two polling calls); a positive value is a hipError_t (an ncclResult_t never leaves the module: the farm entry points
 *     map it to DATUM_OCEAN_ECOMM).  Nothing throws.  datum_ocean_last_error() gives the text.
 *     (reference: throw std::runtime_error on device failure, ocean.cpp:271 / vulkan.cpp:550;
 *     the C++ shim in datum_amd/host re-throws.)
 *   - one HIP stream per handle; calls enqueue and return; datum_ocean_sync() is the fence wait of
 *     ocean.cpp:725.  A handle is not re-entrant; distinct handles (GPUs) are independent.
 *   - h0 / phase stay device resident (the reference re-uploads 12*N*N bytes per frame).
 *   - "cascade" = one independent (OceanParams, N x N grid) problem; the reference has exactly one.
 *   - all arrays are fp32, row-major [m = y][n = x] like OceanParams (ocean.h:69-71).
 */

#ifndef DATUM_OCEAN_HIP_H
#define DATUM_OCEAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DATUM_OCEAN_MAX_CASCADES 16

/* Version of THIS header's contract: bumped whenever a signature, an error code's value or the device layout of a bound map
 * buffer changes (the shared object carries no soname).  A consumer compares it with what the library it loaded reports before its
 * first call -- datum_amd/capi.py, the C++ host shim (initialise_ocean_context) and both examples do -- so that a stale
 * libdatum_ocean_hip.so is refused instead of misread.
 *   3  round 3: ENOTREADY = +1, datum_ocean_map_layout with four arguments, 32-byte texels
 *   4  round 4: ENOTREADY = -5, datum_ocean_map_layout gained texel_bytes, 24-byte texels in bound map buffers, farm entry points
 *   5  round 5: datum_ocean_abi_version, datum_ocean_export_maps, datum_ocean_set_literal_transform
 *   6  round 5: datum_ocean_farm_partition, datum_ocean_own_stream
 *   7  round 6: datum_ocean_set_cascade_group / datum_ocean_cascade_group (DATUM_OCEAN_SPECTRUM_FP16_H0, a further value of an existing
 *      argument, came later in the round without a bump)
 *   8  round 6: datum_ocean_set_map_store_policy / datum_ocean_map_store_policy */
#define DATUM_OCEAN_ABI_VERSION 8
int datum_ocean_abi_version(void);

enum
{
  DATUM_OCEAN_OK = 0,
  DATUM_OCEAN_EINVAL = -1,    /* bad argument (null pointer, unsupported resolution, cascade out of range) */
  DATUM_OCEAN_ESTATE = -2,    /* call order misuse (e.g. displace before upload_state)                     */
  DATUM_OCEAN_ENOMEM = -3,
  DATUM_OCEAN_EUNSUPPORTED = -4,  /* the HIP runtime on this machine lacks the feature (external semaphores on ROCm 7.0.x: use the
                                   host bridge, datum_ocean_on_complete / datum_ocean_query)                                  */
  DATUM_OCEAN_ECOMM = -6,         /* an RCCL call of the tile farm failed (datum_ocean_last_error has RCCL's own text)          */
  DATUM_OCEAN_ENOTREADY = -5      /* datum_ocean_query / datum_ocean_farm_query only: the work is still running (not a failure).
                                   A module code, because every positive return value is a hipError_t (hipErrorInvalidValue
                                   is 1) and a poller must be able to tell "not yet" from "the query itself failed"         */
};

typedef struct datum_ocean_ctx *datum_ocean_t;

/* The head of the reference's OceanSet SSBO (src/renderer/ocean.cpp:33-50; every shader mirrors it,
 * e.g. data/ocean.gen.comp:15-36): std430, row_major, same byte offsets, sizeof == 216.
 * Quaternions are (w, x, y, z) (data/transform.inc:13-28).  h0[] / phase[] that follow in the
 * reference struct live on the device here (datum_ocean_upload_state). */
typedef struct datum_ocean_set
{
  float proj[16];          /*   0  Camera::proj(), camera.cpp:77-89                        */
  float invproj[16];       /*  64  inverse(proj), ocean.cpp:732                            */
  float camera_real[4];    /* 128  camera.transform().real                                 */
  float camera_dual[4];    /* 144  camera.transform().dual                                 */
  float plane[4];          /* 160  (normal, distance), ocean.cpp:735                       */
  float swelllength;       /* 176 */
  float swellamplitude;    /* 180 */
  float swellsteepness;    /* 184 */
  float swellphase;        /* 188 */
  float swelldirection[2]; /* 192 */
  float scale;             /* 200  1 / wavescale, ocean.cpp:743                            */
  float choppiness;        /* 204 */
  float smoothing;         /* 208  1 / params.smoothing, ocean.cpp:745                     */
  uint32_t size;           /* 212  WaveResolution                                          */
} datum_ocean_set;

/* -- lifetime (replaces initialise_ocean_context / prepare_ocean_context, ocean.cpp:325-716) ------------ */

/* resolution: 64 (the reference's WaveResolution, ocean.h:16), 128, 256, 512, 1024, 2048 or 4096.
 * Allocates h0, phase, the work spectrum (ocean.cpp:61-68) and the 2-layer displacement map (ocean.cpp:706; stored as
 * 24-byte texels, datum_ocean_bind_maps) for `cascades` grids on HIP device `device`, and builds the twiddle table. */
int datum_ocean_create(datum_ocean_t *out, int device, int resolution, int cascades);
int datum_ocean_destroy(datum_ocean_t ctx);

/* use_own == 0: enqueue on the caller's hipStream_t (passed as void*; NULL is HIP's default stream).
 * use_own != 0: back to the handle's own stream.  Drains the stream in use before switching. */
int datum_ocean_set_stream(datum_ocean_t ctx, void *hip_stream, int use_own);

/* Let the displacement maps be written into caller-owned DEVICE memory of cascades * N * N * texel_bytes bytes (e.g. a buffer
 * that is later all-gathered); NULL restores the handle's own buffer.
 * DEVICE LAYOUT of a cascade's map block (the reference's displacementmap is a VK_IMAGE_TILING_OPTIMAL image,
 * ocean.cpp:706, whose physical layout is the driver's; its only reader is ocean.gen's sampler, ocean.cpp:759): 24 bytes per
 * texel -- the .w channels of the two RGBA32F layers are constant zero (map.comp:79-80) and are not stored.  Bands of B
 * columns (B = N except for the largest grids); inside a band PATCHES of PW x PH = 16 texels, patch rows one after the other;
 * a patch is 384 bytes = three 128-byte lines: 16 x float4 (dx, dy, dz, nx), then 16 x float2 (ny, nz), texel
 * j = (y % PH) * PW + x % PW of the patch at 16 j and 256 + 8 j.
 *   byte offset of texel (x, y)'s patch = (x / B) * 24 N B + ((y / PH) * (B / PW) + (x % B) / PW) * 384
 * with (PW, PH, B, texel_bytes) from datum_ocean_map_layout (texel_bytes = 24).
 * datum_ocean_read_maps returns the reference's logical image [layer][y][x][4] with .w = 0 on the host,
 * datum_ocean_export_maps writes the same image into device memory (a Vulkan-visible RGBA32F image). */
int datum_ocean_bind_maps(datum_ocean_t ctx, void *device_ptr, size_t bytes);
int datum_ocean_map_layout(int resolution, int *group_cols, int *group_rows, int *band, int *texel_bytes);
int datum_ocean_maps_device(datum_ocean_t ctx, void **device_ptr, size_t *bytes);

/* -- state (OceanParams arrays, ocean.h:67-72) ------------------------------------------------------- */

/* per-cascade wave constants used by sim / map: OceanSet.scale = 1/wavescale, OceanSet.choppiness */
int datum_ocean_set_cascade(datum_ocean_t ctx, int cascade, float wavescale, float choppiness);

/* h0 = OceanParams::height, N*N*2 floats (ocean.cpp:748); phase = OceanParams::phase, N*N floats
 * (ocean.cpp:749) or NULL for all-zero (seed_ocean, ocean.cpp:144).  Host pointers. */
/* Extension (BASELINE.json configs[4]): how the work spectrum between the two passes is stored.  FP32 (default): 16 B
 * per point.  FP16: 8 B per point, arithmetic stays fp32; the values are scaled by a power of two sized from max |h0|
 * so that no row sum can overflow a half.  Displacement error then ~5e-4 relative to the largest displacement
 * (tests state 2e-3).  Takes effect at the next datum_ocean_displace.
 * FP16_H0 (round 6; SURVEY.md 8d's own byte count for that config -- "spectrum + intermediates stored fp16": h0 4 B/pt): FP16, and the
 * row pass reads h0 as two halves per point as well, from a copy the module keeps beside the fp32 h0 (4 more bytes per point of
 * device memory; rebuilt on the device whenever h0 changes: upload, rebuild from the seed, resume): h0 times the power of two that
 * brings its largest component just under 2^15, rounded to nearest even.  What the caller uploads and fetches stays fp32; the phase
 * state never passes through a half and stays bit-exact.  Same stated tolerance (2e-3 of the largest displacement; measured with
 * the example's parameters: tests/test_gpu_parity.py). */
#define DATUM_OCEAN_SPECTRUM_FP32 0
#define DATUM_OCEAN_SPECTRUM_FP16 1
#define DATUM_OCEAN_SPECTRUM_FP16_H0 2
int datum_ocean_set_spectrum_format(datum_ocean_t ctx, int format);

/* VALIDATION MODE (round 5): displace through the reference's own algorithm instead of the fused kernels -- ocean.sim, log2 N
 * radix-2 Stockham stages along rows and along columns with the LITERAL twiddle table of ocean.cpp:686-700 (cos / sin of unreduced
 * fp32 angles, datum_ocean_reference_weights), ocean.map -- the same operations in the same order as data/ocean.{sim,fftx,ffty,map}.comp,
 * one thread per point.  For comparing a HIP frame with a frame of the Vulkan build texel for texel: the fused path differs from the
 * literal arithmetic by that table's own error (RMSE 1.4e-5 at 1024^2, 7e-5 at 4096^2; it is the one closer to a float64 transform).
 * 4-17 x slower than the fused step (tools/literal_bench.py), 24 * N * N bytes of extra device memory once switched on; phase, maps, gen, read_maps,
 * export_maps and the farm work as before.  Takes effect at the next datum_ocean_displace. */
int datum_ocean_set_literal_transform(datum_ocean_t ctx, int on);   /* DATUM_OCEAN_ESTATE while a profile is open or the spectrum format is FP16 (and those two refuse while the mode is on) */

/* Cascades per launch of the two kernels (ABI 7).  The reference records one dispatch per shader for its one grid (ocean.cpp:769-789).
 * A handle whose working set (52 bytes per point and cascade, 44 with the fp16 spectrum) is resident in the 256 MiB Infinity Cache takes
 * every cascade in one launch per kernel.  Beyond that the maps are streamed past the cache and row pass and column pass are launched
 * group by group -- row(g), column(g), row(g + 1), ... on the handle's stream -- so that what a group's row pass leaves for its column pass
 * (and h0 and the phase from step to step) stays in the cache: 0 (default) = the module's choice, the largest group whose 28 (20) bytes
 * per point fit -- 8 cascades of 1024^2, 2 of 2048^2, 1 of 4096^2 -- in groups of equal size; n > 0: n cascades per launch (n >= cascades:
 * one launch per kernel).  Results do not depend on the group.  The getter reports the group in use and the launches per kernel and
 * displace call. */
int datum_ocean_set_cascade_group(datum_ocean_t ctx, int cascades_per_launch);
int datum_ocean_cascade_group(datum_ocean_t ctx, int *cascades_per_launch, int *launches_per_pass);

/* How the column pass stores the maps (ABI 8): written through (sc0 sc1: the lines stay in the Infinity Cache for ocean.gen and the next
 * step) or streamed (nt: past the cache).  AUTO (default): written through while the handle's working set is resident in the cache AND no
 * multi-rank farm is initialised; streamed otherwise -- a collective's gathered buffer (7 peers' payloads at 8 ranks: 352 MB per batch at
 * 1024^2 x 4) competes for the same cache, and with the maps kept out of it the step loses less under the collective (one-GPU stand-in,
 * profiles/r06_farm_standin.txt: 58.5-61.7 k -> 63.2-64.1 k grids/s at 1024^2 x 4, 13.9-14.1 k -> 14.7-15.4 k at 2048^2 x 1; without a
 * collective in flight streaming costs 2-4 % there).  The explicit values override that.  Grids below 1024^2 are always written through
 * and 4096^2 always streams (there is one form of their kernels).  Results do not depend on the policy.  The getter reports the policy
 * set and whether the next displace will stream. */
#define DATUM_OCEAN_MAPS_AUTO 0
#define DATUM_OCEAN_MAPS_WRITTEN_THROUGH 1
#define DATUM_OCEAN_MAPS_STREAMED 2
int datum_ocean_set_map_store_policy(datum_ocean_t ctx, int policy);
int datum_ocean_map_store_policy(datum_ocean_t ctx, int *policy, int *streamed);

int datum_ocean_upload_state(datum_ocean_t ctx, int cascade, float const *h0, float const *phase);
int datum_ocean_read_state(datum_ocean_t ctx, int cascade, float *phase);

/* Park a cascade's state (h0 and the phase as advanced so far: 12 * N * N bytes, h0 first) in caller-owned DEVICE memory,
 * and bring a parked state back -- device to device on the handle's stream, no host round trip.  For a host object that
 * renders more states than the handle has cascades (the reference keeps every OceanParams' phase on the host and uploads
 * it per frame, ocean.cpp:748-749, so any context can render any params at any time; here the phase lives on the device).
 * Updates queued by datum_ocean_update are applied before either call.  `flags` carries what the module knows about
 * the parked phase (pass back what park returned).  bytes must be datum_ocean_state_bytes(). */
size_t datum_ocean_state_bytes(int resolution);
int datum_ocean_park_state(datum_ocean_t ctx, int cascade, void *device_dst, size_t bytes, int *flags);
int datum_ocean_resume_state(datum_ocean_t ctx, int cascade, void const *device_src, size_t bytes, int flags);

/* Device-side spectrum rebuild (what lerp_ocean_waves does on the host, ocean.cpp:194-211):
 * keep OceanParams::seed (N*N*2 floats, ocean.cpp:140-141) resident and recompute
 * h0 = seed * dk * sqrt(phillips(k, waveamplitude, windspeed, winddirection) / 2), dk = 2 pi / wavescale,
 * on the device when the wind changes, instead of an O(N^2) host loop plus an 8*N*N-byte upload.
 * rebuild_height also installs `wavescale` as the cascade's wave scale (choppiness is kept). */
int datum_ocean_upload_seed(datum_ocean_t ctx, int cascade, float const *seed);
int datum_ocean_rebuild_height(datum_ocean_t ctx, int cascade, float wavescale, float waveamplitude, float windspeed, float windx, float windy);
int datum_ocean_read_height(datum_ocean_t ctx, int cascade, float *h0);

/* -- the per-frame path ---------------------------------------------------------------------------------- */

/* update_ocean's phase advance (ocean.cpp:223-233) for every cascade:
 * phase = fmod(phase + dispersion(k) * dt, 2 pi), in fp32, in call order.  It is applied on the device,
 * fused into the next datum_ocean_displace (bit-identical to applying each dt in turn). */
int datum_ocean_update(datum_ocean_t ctx, float dt);

/* ocean.sim -> ocean.fftx -> ocean.ffty -> ocean.map for every cascade (ocean.cpp:769-789), as two fused
 * kernels.  Result: per cascade [layer][y][x][4] floats, layer 0 = (dx, dy, dz, 0), layer 1 = (normal, 0)
 * (data/ocean.map.comp:79-80). */
int datum_ocean_displace(datum_ocean_t ctx);

/* ocean.gen (data/ocean.gen.comp, ocean.cpp:791-793): fills sizex*sizey Mesh::Vertex (48 bytes:
 * position3, texcoord2, normal3, tangent4 -- src/renderer/mesh.h:20-26) in DEVICE memory from the
 * cascade's displacement map.  sizex, sizey: multiples of 16 in the reference; any size >= 2 here. */
int datum_ocean_gen(datum_ocean_t ctx, int cascade, datum_ocean_set const *set, int sizex, int sizey, void *vertices_device);

/* What a rank contributes when the tiles of a multi-GPU farm are reassembled with one all-gather (SURVEY.md 8e;
 * nothing in the reference: it has one device).  Packs the displacement of every cascade of the handle into
 * caller-owned DEVICE memory, enqueued on the handle's stream behind the last displace, so that the collective can
 * read it on another stream while the next displace overwrites the maps:
 *   MAPS   the whole map block as it lies in memory, both layers (24 B per point, device layout of bind_maps)
 *   XYZ32  layer 0 only, [cascade][y][x] (dx, dy, dz) as three floats   (12 B per point, exact)
 *   XYZ16  layer 0 only, [cascade][y][x] (dx, dy, dz, 0) as four halves  (8 B per point, round to nearest) */
#define DATUM_OCEAN_PAYLOAD_MAPS 0
#define DATUM_OCEAN_PAYLOAD_XYZ32 1
#define DATUM_OCEAN_PAYLOAD_XYZ16 2
int datum_ocean_payload_bytes(datum_ocean_t ctx, int format, size_t *bytes);
int datum_ocean_pack_displacement(datum_ocean_t ctx, int format, void *payload_device, size_t bytes);

/* -- the tile farm: N processes, one GPU each, independent tiles / cascades, ONE all-gather per batch ------------------
 * (nothing in the reference, which has one device.)  The displacement step needs no exchange; what
 * north_star's "single RCCL all-gather over xGMI" reassembles is the displacement field of every rank's grids.  These
 * entry points own that step, so a C++ renderer farms tiles with this module alone: the RCCL communicator, a communication
 * stream, `slots` payload buffers (this rank's packed displacement, datum_ocean_pack_displacement's formats) and `slots`
 * gathered buffers (world x payload, ordered by rank), and the event choreography between them.
 *
 *   unique_id   rank 0 only: the 128-byte id of a new communicator; hand it to every rank (a pipe, a file, MPI, a socket)
 *   init        every rank, same id / world / format / slots: creates the communicator (blocks until all ranks have called)
 *   gather      enqueue and return: on the handle's stream, behind the last displace, the pack into the next slot's payload
 *               (first waiting, on the device, for the collective that last read that payload); on the communication stream,
 *               behind the pack and behind the slot's last release, ncclAllGather into the slot's gathered buffer.  The
 *               following displace calls overlap the collective.  *slot = which slot.
 *   result      `hip_stream` (or the handle's own stream when on_handle_stream != 0) waits for the slot's collective;
 *               *gathered_device = world x payload_bytes bytes, rank r's payload at r * payload_bytes.  A slot is gathered into
 *               again `slots` gathers later: take its result (and release it) before that
 *   release     `hip_stream` (or the handle's) has finished reading the gathered buffer up to this point: the slot's next
 *               collective waits for it.  Needed whenever a consumer reads on a stream of its own: a gather that comes round to
 *               a slot whose result went to another stream and was not released fails with DATUM_OCEAN_ESTATE.
 *   query       DATUM_OCEAN_OK once the slot's collective has finished, DATUM_OCEAN_ENOTREADY before; never blocks
 *   wait        blocks the host until it has; *collective_ms (may be NULL) = its duration on the communication stream
 *   shutdown    destroys communicator, stream and buffers (datum_ocean_destroy does it too)
 * RCCL is opened with dlopen at the first farm call (DATUM_OCEAN_RCCL_LIB, librccl.so.1): DATUM_OCEAN_EUNSUPPORTED when no
 * library is found; an ncclResult_t is reported as DATUM_OCEAN_ECOMM with RCCL's text in datum_ocean_last_error.
 * datum_ocean_farm_init is a COLLECTIVE over the ranks (ncclCommInitRank) and RCCL gives it no timeout: it returns when every rank of
 * the farm has called it, and not at all when one never does.  A launcher must therefore end the remaining ranks when one rank dies
 * before its farm_init (examples/ocean_farm.cpp reaps its ranks in the order they end and does so; bench.py's launcher likewise). */
#define DATUM_OCEAN_FARM_ID_BYTES 128
int datum_ocean_farm_unique_id(void *id, size_t bytes);
int datum_ocean_farm_init(datum_ocean_t ctx, void const *id, size_t bytes, int rank, int world, int format, int slots);
int datum_ocean_farm_shutdown(datum_ocean_t ctx);
int datum_ocean_farm_info(datum_ocean_t ctx, int *rank, int *world, int *format, size_t *payload_bytes, int *slots, int *rccl_version);
int datum_ocean_farm_gather(datum_ocean_t ctx, int *slot);
int datum_ocean_farm_result(datum_ocean_t ctx, int slot, void *hip_stream, int on_handle_stream, void **gathered_device, size_t *bytes);
int datum_ocean_farm_release(datum_ocean_t ctx, int slot, void *hip_stream, int on_handle_stream);
int datum_ocean_farm_query(datum_ocean_t ctx, int slot);
int datum_ocean_farm_wait(datum_ocean_t ctx, int slot, float *collective_ms);

/* The farm's two streams on DISJOINT compute units (hipExtStreamCreateWithCUMask; ABI 6).  RCCL's channels are workgroups that copy; while
 * they share compute units with the step's workgroups their bursts sit in the same in-order memory queues and the step kernels take 46-54 %
 * longer under a collective-sized copy (one-GPU stand-in, profiles/r05_gather_overhead.txt); with the copy on 32 compute units of its own and
 * the step on the other 224 it is 27-28 % (7 % of it the 32 CUs the step gives up).
 *   partition   after farm_init, nothing in flight: comm_cus (0 = undo, else a multiple of 8, at most half the device) compute units --
 *               comm_cus / 8 of each XCD -- for the communication stream, the others for the handle's OWN stream; both streams are
 *               recreated.  A handle that runs on a caller's stream (datum_ocean_set_stream) keeps it: hand it the own stream instead
 *   own_stream  the handle's own hipStream_t (e.g. to record events on it or to make it the current stream of a framework), valid until
 *               the next partition / shutdown / destroy
 * farm_shutdown undoes the partition.
 * NULL STREAM: hipExtStreamCreateWithCUMask takes no flags; the streams it makes report hipStreamDefault where this runtime says anything
 * (farm_stream_flags below: 0 = hipStreamDefault = BLOCKING, 1 = hipStreamNonBlocking), i.e. unlike the hipStreamNonBlocking streams they
 * replace they synchronise with the legacy null stream: an operation a caller puts on the null stream (a synchronous hipMemcpy, a
 * framework's default stream) while the farm runs serialises the communication and the compute stream and costs the overlap the
 * partition exists for -- results stay right.  The module itself never touches the null stream after datum_ocean_create (every copy,
 * memset and kernel of every entry point is on the handle's stream or the communication stream: tests/test_golden_and_abi.py holds
 * the sources to that); keep the caller's own work off it too while a partitioned farm is gathering. */
#define DATUM_OCEAN_FARM_PARTITION_AUTO (-1)   /* comm_cus: an eighth of the device in whole shares of 8 (32 of 256); 0 on a device with fewer than 64 compute units */
int datum_ocean_farm_partition(datum_ocean_t ctx, int comm_cus);
int datum_ocean_own_stream(datum_ocean_t ctx, void **hip_stream);
int datum_ocean_farm_stream_flags(datum_ocean_t ctx, unsigned int *communication_stream_flags, unsigned int *own_stream_flags);   /* hipStreamGetFlags of the two (ABI 7) */

/* blocking read-backs (host pointers).  maps: 2*N*N*4 floats. */
int datum_ocean_read_maps(datum_ocean_t ctx, int cascade, float *maps);

/* The same image on the DEVICE: a cascade's displacement map as the reference's shaders see it (ocean.cpp:706, map.comp:79-80:
 * N x N x 2 layers RGBA32F, [layer][y][x][4], .w = 0) written into caller-owned device memory of 2 * N * N * 16 bytes -- e.g. the
 * imported memory of a linear VkImage / VkBuffer the renderer samples itself (datum_ocean_import_memory_fd).  One kernel on
 * the handle's stream behind the last displace; the module's own 24-byte layout is untouched. */
int datum_ocean_export_maps(datum_ocean_t ctx, int cascade, void *device_dst, size_t bytes);

/* wait_fence (ocean.cpp:725) */
int datum_ocean_sync(datum_ocean_t ctx);

/* Cross-queue ordering (the reference's VkSemaphores: up to 8 wait dependencies, vulkan.cpp:1308-1328, and the
 * `rendercomplete` semaphore signalled by the submit, ocean.cpp:803, that render() waits on, renderer.cpp:6848).
 * wait_event: work enqueued afterwards waits for the caller's hipEvent_t.  signal: records the handle's own
 * completion hipEvent_t behind everything enqueued so far and returns it (owned by the handle). */
int datum_ocean_wait_event(datum_ocean_t ctx, void *hip_event);
int datum_ocean_signal(datum_ocean_t ctx, void **hip_event);

/* The host bridge for `rendercomplete` (ocean.cpp:341,803; waited on at renderer.cpp:6848) where the runtime cannot
 * import a VkSemaphore (hipImportExternalSemaphore returns "not supported" on ROCm 7.0.x: import_semaphore_fd then fails with
 * DATUM_OCEAN_EUNSUPPORTED):
 *   on_complete  `callback(user)` runs on a runtime thread once everything enqueued on the handle's stream so far has
 *                finished -- the integrator signals the renderer from it (vkSignalSemaphore on a timeline semaphore, or an
 *                empty vkQueueSubmit that signals the binary `rendercomplete`).  No HIP call may be made from the callback.
 *   query        DATUM_OCEAN_OK when everything enqueued before the last datum_ocean_signal has finished,
 *                DATUM_OCEAN_ENOTREADY while it has not; never blocks (a renderer that polls once per frame).
 * INTEGRATION.md 3a has the call sequence. */
int datum_ocean_on_complete(datum_ocean_t ctx, void (*callback)(void *user), void *user);
int datum_ocean_query(datum_ocean_t ctx);

/* -- Vulkan <-> HIP interop, the HIP half --------------------------------------------------
 * In datum the Ocean mesh's vertex buffer is a VkBuffer the graphics queue draws from (ocean.cpp:270, bound at
 * geometrylist.cpp:463,513) and the frame's submit waits on the ocean's `rendercomplete` VkSemaphore (ocean.cpp:803,
 * renderer.cpp:6848).  For the renderer to consume what this module writes without a copy, the renderer exports the
 * buffer's VkDeviceMemory and its semaphores as POSIX file descriptors (VK_KHR_external_memory_fd /
 * VK_KHR_external_semaphore_fd, handle type OPAQUE_FD) and this module imports them:
 *   import_memory_fd     hipImportExternalMemory + hipExternalMemoryGetMappedBuffer: a device pointer over the same
 *                        memory, valid for datum_ocean_gen (vertices) and datum_ocean_bind_maps.  On success the
 *                        descriptor belongs to the handle (do not close it); released by release_memory or destroy.
 *   import_semaphore_fd  hipImportExternalSemaphore; signal_external / wait_external enqueue a signal / a wait on the
 *                        handle's stream: the rendercomplete semaphore and the up to 8 wait dependencies of the
 *                        reference's submit (vulkan.cpp:1308-1328).  Returns DATUM_OCEAN_EUNSUPPORTED where the runtime
 *                        has no external semaphores (ROCm 7.0.x on the MI355X boxes: only the MEMORY import has ever
 *                        succeeded there, profiles/r02_external_memory_probe.txt); the host bridge above is what works.
 * INTEGRATION.md has the Vulkan side of the handshake. */
int datum_ocean_import_memory_fd(datum_ocean_t ctx, int fd, size_t bytes, void **device_ptr);
int datum_ocean_release_memory(datum_ocean_t ctx, void *device_ptr);
int datum_ocean_import_semaphore_fd(datum_ocean_t ctx, int fd, void **semaphore);
int datum_ocean_release_semaphore(datum_ocean_t ctx, void *semaphore);
int datum_ocean_signal_external(datum_ocean_t ctx, void *semaphore);
int datum_ocean_wait_external(datum_ocean_t ctx, void *semaphore);

/* Device memory for the Ocean mesh (vertex buffer with compute-writable usage + index buffer,
 * ResourceManager::create<Ocean>, ocean.cpp:262-288) for hosts that do not link HIP themselves.
 * write/read are ordered on the handle's stream; read blocks until the data is on the host. */
int datum_ocean_device_alloc(datum_ocean_t ctx, size_t bytes, void **device_ptr);
int datum_ocean_device_free(datum_ocean_t ctx, void *device_ptr);
int datum_ocean_device_write(datum_ocean_t ctx, void *device_dst, void const *host_src, size_t bytes);
int datum_ocean_device_read(datum_ocean_t ctx, void *host_dst, void const *device_src, size_t bytes);

char const *datum_ocean_last_error(datum_ocean_t ctx);

/* -- the reference's own host-side table, kept for API parity (ocean.cpp:686-700) ------------------------- */

/* weights[i * 2*log2(N) + 2*s + {0,1}] = cos / sin(-2 pi i / 2^(s+1)) evaluated exactly as the reference
 * does (fp32, unreduced angle).  The HIP kernels do not consume it. */
int datum_ocean_reference_weights(int resolution, float *weights);

/* -- diagnostics ------------------------------------------------------------------------------------------- */

/* ocean.sim alone from the current device state (no phase advance): N*N*2 floats each, host pointers */
int datum_ocean_debug_sim(datum_ocean_t ctx, int cascade, float *h, float *hx, float *hy);

/* the work spectrum after the row pass of the last datum_ocean_displace, converted back to row-major: N*N*2 floats
 * each, host pointers.  The module transforms two packed fields instead of the reference's three (what
 * ocean.map keeps is the real part of each transform): with F_S[y][x] = F[y][x] + conj(F[(N-y)%N][(N-x)%N])
 * (twice the Hermitian part; the halving is folded into the column pass),
 *   c = rows of ocean.fftx applied to  C = h_S + i hx_S
 *   d = rows of ocean.fftx applied to  D = hy_S + 2 sin(2 pi x / N) h_S
 * where h, hx, hy are ocean.sim's outputs (datum_ocean_debug_sim). */
int datum_ocean_debug_rowpass(datum_ocean_t ctx, int cascade, float *c, float *d);

/* hipEvent timing of the two kernels of datum_ocean_displace on the handle's stream.
 * begin: every `stride`-th displace call (at most max_samples of them) launches its two kernels with a start and a
 * stop event attached to the dispatch itself (hipExtLaunchKernel), i.e. each sampled kernel is timed from its first
 * to its last workgroup and nothing is inserted between the kernels; end: sync and return the mean kernel durations
 * in milliseconds and the number of displace calls sampled.
 * With several cascade groups per displace call (datum_ocean_set_cascade_group) a sample is the SUM over the call's launches of either kernel. */
int datum_ocean_profile_begin(datum_ocean_t ctx, int max_samples, int stride);
int datum_ocean_profile_end(datum_ocean_t ctx, double *rowpass_ms, double *colpass_ms, int *steps);

/* algorithmic bytes moved by one displace call of the handle (96 B per point per cascade;
 * split: row pass 16 in + 24 out, column pass 24 in + 32 out) */
int datum_ocean_algorithmic_bytes(datum_ocean_t ctx, double *rowpass_bytes, double *colpass_bytes);

#ifdef __cplusplus
}
#endif

#endif
