//
// ocean.h -- host-side mirror of datum's ocean API (src/renderer/ocean.h:12-99) over the HIP module
//
// Same type names, field names, defaults and free-function signatures as the reference, so code written
// against datum's ocean API (examples/ocean/ocean.cpp:31,46-52,59,135,165,179) compiles against this header
// with three documented differences:
//   1. WaveResolution is a run-time value (OceanContext::resolution / OceanParams::resolution, default 64 =
//      ocean.h:16); the [64][64] state arrays of OceanParams (ocean.h:69-71) are sized by it.  The reference's POD itself
//      (memcpy / serialise by value) is OceanParamsPod below, with to_pod / from_pod.
//   2. The Vulkan objects of OceanContext are replaced by one opaque HIP-module handle
//      (include/datum_ocean_hip.h); VkSemaphore dependencies become hipEvent_t handles (void*).
//   3. update_ocean's phase loop (ocean.cpp:223-233) runs on the device: update_ocean() queues dt,
//      render_ocean_surface() applies the queue in order, bit-identically.  At the reference's own resolution
//      (N <= 64) update_ocean ALSO runs the loop on the host, as the reference does (OceanParams::hostphase, on by
//      default there: OceanParams::phase is current after every call); above it OceanParams::phase on the host
//      is the state as of the last fetch_ocean_state().
// Device failures throw std::runtime_error as the reference does (ocean.cpp:271, vulkan.cpp:550);
// misuse trips assert (ocean.cpp:351,722-723); prepare_ocean_context returns bool (ocean.cpp:493-501).
//

#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "lml.h"
#include "../../include/datum_ocean_hip.h"

//|---------------------- stand-ins for the surrounding engine ---------------
// only what the ocean call sites name

namespace DatumPlatform
{
  // src/platform.h: render_device() -> VkDevice + queues.  Here: which HIP device the ocean runs on.
  struct PlatformInterface
  {
    int hipdevice = 0;
  };
}

class AssetManager   // src/asset.h: the HIP module embeds its kernels, no asset pack is consulted
{
};

class Camera   // src/renderer/camera.h:16-107, the subset the ocean path reads
{
  public:
    Camera();

    lml::Vec3 position() const { return m_position; }
    lml::Quaternion3 rotation() const { return m_rotation; }

    float fov() const { return m_fov; }
    float aspect() const { return m_aspect; }
    float znear() const { return m_znear; }
    float zfar() const { return m_zfar; }

    lml::Matrix4f proj() const;                                                                       // camera.cpp:77-91
    lml::Transform transform() const { return lml::Transform::lookat(m_position, m_rotation); }       // camera.h:48

    void set_projection(float fov, float aspect, float znear = 0.1f, float zfar = 24000.0f);          // camera.h:52
    void set_position(lml::Vec3 const &position) { m_position = position; }
    void set_rotation(lml::Quaternion3 const &rotation) { m_rotation = rotation; }
    void lookat(lml::Vec3 const &position, lml::Vec3 const &target, lml::Vec3 const &up);             // camera.cpp:151-155

  private:

    float m_fov, m_aspect, m_znear, m_zfar;
    lml::Vec3 m_position;
    lml::Quaternion3 m_rotation;
};

class Mesh   // src/renderer/mesh.h:16-80, the parts Ocean uses
{
  public:

    struct Vertex
    {
      lml::Vec3 position;
      lml::Vec2 texcoord;
      lml::Vec3 normal;
      lml::Vec4 tangent;
    };

    enum class State { Empty, Loading, Waiting, Testing, Ready };

    bool ready() const { return state == State::Ready; }

    // vertex buffer in DEVICE memory (compute-writable, ocean.cpp:270) + static index buffer
    struct VertexBuffer
    {
      uint32_t vertexcount = 0;
      uint32_t vertexsize = 0;
      uint32_t indexcount = 0;
      uint32_t indexsize = 0;
      void *vertices = nullptr;
      void *indices = nullptr;
    } vertexbuffer;

    State state = State::Empty;
};

static_assert(sizeof(Mesh::Vertex) == 48, "Mesh::Vertex must be 48 bytes (mesh.h:20-26)");

class Ocean;
struct OceanContext;

class ResourceManager   // src/renderer/resource.h: create / release / destroy for the Ocean mesh only
{
  public:
    explicit ResourceManager(OceanContext &context) : m_context(&context) { }

    template<typename Resource, typename ...Args>
    Resource const *create(Args... args);

    template<typename Resource>
    void release(Resource const *resource);

    template<typename Resource>
    void destroy(Resource const *resource);

  private:
    OceanContext *m_context;
};

//|---------------------- Ocean ---------------------------------------------
//|--------------------------------------------------------------------------

struct OceanContext
{
  bool ready = false;

  static const int WaveResolution = 64;   // ocean.h:16: the default; `resolution` is what is used

  int resolution = WaveResolution;

  int device = 0;

  bool spectrumfp16 = false;              // extension: store the module's work spectrum as halves (set before prepare_ocean_context)
  bool heightfp16 = false;                // with spectrumfp16: the module reads OceanParams::height as halves too, from a device copy of its own
                                          // (DATUM_OCEAN_SPECTRUM_FP16_H0; what the caller holds and fetches stays fp32)
  bool literaltransform = false;          // validation: displace through the reference's own radix-2 transforms and literal twiddle table
                                          // (datum_ocean_set_literal_transform; set before prepare_ocean_context) -- for texel-for-texel A/B with the Vulkan build;
                                          // not together with spectrumfp16 (the mode is the reference's fp32 arithmetic: prepare_ocean_context throws)

  datum_ocean_t hip = nullptr;            // replaces vulkan / pipelines / oceanset / spectrum / displacementmap

  void *rendercomplete = nullptr;         // hipEvent_t recorded behind the last render (ocean.h:45)

  std::uint64_t boundstate = 0;           // which OceanParams state is resident on the device (0: none; state ids start at 1)
  std::uint64_t boundheight = 0;
  std::uint64_t boundseed = 0;            // which OceanParams seed is resident on the device (deviceheight only)
  std::uint64_t appliedupdates = 0;       // how many of that state's update_ocean calls the device has applied (OceanParams::updates)
  std::uint64_t appliedlineage = 0;       // OceanParams::lineage_at(appliedupdates) of the history those updates came from

  // States this context has rendered before and had to make room for: h0 and the phase as advanced so far, parked in
  // device memory (datum_ocean_park_state), so that alternating between several OceanParams on one context costs two
  // device-to-device copies per switch and replays only the updates issued since -- the reference keeps every params'
  // phase on the host and any context renders any params at any time (ocean.cpp:217-236,748-749).
  struct Parked
  {
    std::uint64_t stateid = 0, heightid = 0, appliedupdates = 0, appliedlineage = 0, lastuse = 0;
    void *device = nullptr;
    int flags = 0;
  };

  // Capacity and footprint: up to MaxParkedStates parked copies + one spare buffer of datum_ocean_state_bytes(resolution) =
  // 12 N^2 bytes each (64^2: 48 KB; 1024^2: 12.6 MB; 4096^2: 201 MB, i.e. up to 1 GB of HBM per context), allocated on first
  // use and held until release_parked_states() or the context's destruction -- a slot whose OceanParams was re-seeded or
  // destroyed is not noticed by the context.  The bound state and the slot being resumed trade places through the spare, so
  // MaxParkedStates + 1 states alternate without an eviction; with MORE, the least recently used slot is given up and its state
  // falls back to the host copy + recorded history (OceanParams::hostphase, or a fetch_ocean_state at least every
  // MaxRecordedUpdates / 2 steps, keeps that possible).
  static const std::size_t MaxParkedStates = 4;

  std::vector<Parked> parked;
  std::uint64_t useclock = 0;

  void *spare = nullptr;                  // one more buffer of the same size: the bound state and the slot being resumed trade places through it
  int spareflags = 0;

  OceanContext() = default;
  OceanContext(OceanContext const &) = delete;
  OceanContext &operator=(OceanContext const &) = delete;
  ~OceanContext();
};

struct OceanParams
{
  lml::Plane plane = { { 0.0f, 0.0f, 1.0f }, 0.0f };

  // Swell
  float swelllength = 40.0f;
  float swellamplitude = 0.8f;
  float swellsteepness = 0.0f;
  float swellspeed = 1.25f;
  lml::Vec2 swelldirection = { 0.780869f, 0.624695f };

  // Waves
  float wavescale = 64.0f;
  float waveamplitude = 0.00002f;
  float windspeed = 30.0f;
  lml::Vec2 winddirection = { 0.780869f, 0.624695f };
  float choppiness = 1.35f;
  float smoothing = 280.0f;

  // State
  float swellphase = 0.0f;
  int resolution;                 // N (the reference: OceanContext::WaveResolution)
  std::vector<float> seed;        // [N][N][2]
  std::vector<float> height;      // [N][N][2]   h0
  std::vector<float> phase;       // [N][N]      as of seed_ocean / the last fetch_ocean_state
  lml::Vec2 flow = { 0.0f, 0.0f };

  // extension: let lerp_ocean_waves leave the h0 rebuild (ocean.cpp:194-211) to the device: the seed is uploaded once
  // and render_ocean_surface runs the rebuild kernel when the wave parameters have changed.  `height` on the host is
  // then stale until fetch_ocean_state().  Off by default (the reference recomputes on the host).
  bool deviceheight = false;

  // Also advance `phase` on the host in every update_ocean call, as the reference does (ocean.cpp:223-233; N * N fmod per call).
  // ON by default at the reference's own resolution (N <= WaveResolution = 64: 33 us per call) -- there an OceanParams behaves
  // exactly like the reference's: `phase` is current after every update_ocean, any context renders any params at any time,
  // to_pod() always succeeds -- and OFF above it (the constructor decides), where the N * N host loop is what had to leave the
  // CPU: the phase lives on the device (and in the contexts' parked copies), the host copy is the state as of seed_ocean /
  // the last fetch_ocean_state and is only needed by a context that holds no copy.  Either way the device advances its own
  // copy (the row-pass kernel, bit for bit the same values).
  bool hostphase = false;

  // device residency bookkeeping (not in the reference)
  //
  // update_ocean's phase loop runs on the device, so update_ocean() only RECORDS the step: entry number
  // `firstupdate + i` of this state's history is updates[i] = (dt, the wavescale in force at that call).  The history
  // belongs to the state (stateid), not to one OceanParams object: OceanParams stays freely copyable like the reference's
  // POD -- a per-frame copy handed to the render thread carries the same stateid and a prefix-consistent history -- and
  // rendering is const: render_ocean_surface applies the entries the CONTEXT has not applied yet
  // (OceanContext::appliedupdates) and never touches the params.  The history keeps the last MaxRecordedUpdates entries.  A
  // context that alternates between several params parks the one it is not rendering (OceanContext::Parked) and continues
  // from there; without hostphase (N > 64, or switched off) a context that holds NO copy of a state whose host phase is older
  // than the recorded history throws (render or fetch a state every MaxRecordedUpdates / 2 steps, or set hostphase).
  // (lineage: a running hash over the history up to and including this entry.  A copy of an OceanParams that is then
  // advanced on its own -- update_ocean(P, a), update_ocean(Q, b) -- shares P's stateid and history NUMBERS but not their
  // contents; a context tells the two apart by the lineage of the last entry it applied.)
  struct Update { float dt, wavescale; std::uint64_t lineage; };

  static const std::size_t MaxRecordedUpdates = 4096;

  std::uint64_t stateid;                // a fresh id per constructed / seeded state (never 0)
  std::uint64_t heightid = 0;           // changes when `height` is recomputed (lerp_ocean_waves)
  std::uint64_t firstupdate = 0;        // history number of updates[0]
  std::vector<Update> updates;          // update_ocean calls since firstupdate
  std::uint64_t phaseupdates = 0;       // how many entries of the history `phase` (on the host) already contains
  std::uint64_t baselineage = 0;        // lineage of history entry firstupdate - 1 (0 at the start of a history)

  // lineage of the history's first `count` entries (count in [firstupdate, firstupdate + updates.size()])
  std::uint64_t lineage_at(std::uint64_t count) const { return count == firstupdate ? baselineage : updates[count - firstupdate - 1].lineage; }
  int rejectedseeds = 0;                // seed pairs whose 8 polar draws were all rejected (see seed_ocean)

  explicit OceanParams(int resolution = OceanContext::WaveResolution);
};

// The reference's OceanParams byte for byte (src/renderer/ocean.h:48-73 with WaveResolution = 64, ocean.h:16): for callers that
// memcpy, serialise or load an OceanParams by value -- OceanParams above carries run-time sized arrays and device bookkeeping and
// is not that POD.  to_pod() needs the host phase to be current (fetch_ocean_state, or hostphase):
// it returns false, and leaves `pod` alone, when update_ocean calls are recorded that `params.phase` does not contain yet.
struct OceanParamsPod
{
  lml::Plane plane;

  float swelllength, swellamplitude, swellsteepness, swellspeed;
  lml::Vec2 swelldirection;

  float wavescale, waveamplitude, windspeed;
  lml::Vec2 winddirection;
  float choppiness, smoothing;

  float swellphase;
  float seed[OceanContext::WaveResolution][OceanContext::WaveResolution][2];
  float height[OceanContext::WaveResolution][OceanContext::WaveResolution][2];
  float phase[OceanContext::WaveResolution][OceanContext::WaveResolution];
  lml::Vec2 flow;
};

static_assert(sizeof(OceanParamsPod) == 82000, "OceanParamsPod must be the reference's OceanParams (ocean.h:48-73): 80 bytes of tunables + swellphase, 3 arrays, flow");

bool to_pod(OceanParams const &params, OceanParamsPod &pod);         // resolution must be 64
OceanParams from_pod(OceanParamsPod const &pod);                      // a fresh state (new state id, empty history) holding the pod's arrays

class Ocean : public Mesh
{
  public:
    friend class ResourceManager;

    int sizex;
    int sizey;

  protected:
    Ocean() = default;

    OceanContext *context = nullptr;
};

template<> Ocean const *ResourceManager::create<Ocean>(int sizex, int sizey);
template<> void ResourceManager::release<Ocean>(Ocean const *ocean);
template<> void ResourceManager::destroy<Ocean>(Ocean const *ocean);

void seed_ocean(OceanParams &params);
void seed_ocean(OceanParams &params, std::uint32_t rngseed);   // extension: reproducible entropy (the reference uses random_device, ocean.cpp:132)
void lerp_ocean_swell(OceanParams &params, float swelllength, float swellamplitude, float swellspeed, lml::Vec2 swelldirection, float t);
void lerp_ocean_waves(OceanParams &params, float wavescale, float waveamplitude, float windspeed, lml::Vec2 winddirection, float t);
void update_ocean(OceanParams &params, float dt);

// Initialise
void initialise_ocean_context(DatumPlatform::PlatformInterface &platform, OceanContext &context, uint32_t queueindex);

// Prepare
bool prepare_ocean_context(DatumPlatform::PlatformInterface &platform, OceanContext &context, AssetManager &assets);

// Render
void render_ocean_surface(OceanContext &context, Ocean const *target, Camera const &camera, OceanParams const &params, void *const (&dependancies)[8] = {});

// Extensions (not in the reference)

// displacement maps only (ocean.sim .. ocean.map), no mesh: what the bench times
void displace_ocean_surface(OceanContext &context, OceanParams const &params);

// free the context's parked device copies (all of them, or all but `keep`'s): returns the bytes of HBM given back.  The states
// themselves are not lost while their OceanParams can still be uploaded (host phase + recorded history)
std::size_t release_parked_states(OceanContext &context, OceanParams const *keep = nullptr);

// copy the device-resident phase (and, with deviceheight, h0) back into params (applies any queued update first)
void fetch_ocean_state(OceanContext &context, OceanParams &params);

// blocking read-backs for tools and tests
void read_ocean_displacement(OceanContext &context, float *maps /* [2][N][N][4] */);
void read_ocean_vertices(OceanContext &context, Ocean const *ocean, Mesh::Vertex *vertices);

// the OceanSet header render_ocean_surface uploads (ocean.cpp:731-747)
datum_ocean_set make_oceanset(Camera const &camera, OceanParams const &params);

// the reference's twiddle table (ocean.cpp:686-700), N x 2 log2(N) floats; kept for API parity
std::vector<float> ocean_twiddle_table(int resolution);
