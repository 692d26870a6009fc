//
// ocean.cpp -- host side of the ocean path over the HIP module (see ocean.h)
//
// What stays on the host is what the reference runs once or rarely on the host: Phillips-spectrum
// seeding (src/renderer/ocean.cpp:82-168), the parameter blends (:172-213), the scalar part of
// update_ocean (:221,:235), the OceanSet header (:731-747) and the mesh index buffer (:275-286).
// Everything per-point per-frame is enqueued on the device through include/datum_ocean_hip.h.
//

#include "ocean.h"

#include <atomic>
#include <cassert>
#include <cmath>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>

using namespace std;
using namespace lml;

namespace
{
  atomic<uint64_t> g_stateids{1};

  void check(datum_ocean_t hip, int rc, char const *what)
  {
    if (rc != DATUM_OCEAN_OK)
      throw runtime_error(string(what) + ": " + datum_ocean_last_error(hip));
  }

  // omega(k) = sqrt(g |k| (1 + k^2 / 370^2))   (ocean.cpp:82-87)
  float dispersion(Vec2 const &k)
  {
    float const wm = 370.0f;
    float const knorm = sqrt(normsqr(k));

    return sqrt(9.81f * knorm * (1.0f + normsqr(k) / (wm * wm)));
  }

  // Phillips spectrum P(k) for wind speed v along w, amplitude a (ocean.cpp:89-107):
  // a * d * exp(-1/(k^2 L^2)) / k^6 * (k.w)^2 * exp(-k^2 l^2), L = v^2/g, l = L/1000,
  // waves running against the wind damped by d = 0.2; no energy at k = 0.
  float phillips(Vec2 const &k, float a, float v, Vec2 const &w)
  {
    if (k.x == 0 && k.y == 0)
      return 0.0f;

    float const against = 0.2f;
    float const damping = 0.001f;

    float kdotw = dot(k, w);
    float d = (kdotw < 0) ? against : 1.0f;

    float L = v * v / 9.81f;
    float L2 = L * L;
    float l2 = L2 * damping * damping;

    float k2 = normsqr(k);

    return a * d * exp(-1.0f / (k2 * L2)) / (k2*k2*k2) * (kdotw*kdotw) * exp(-k2 * l2);
  }

  // one complex Gaussian sample by the polar method, at most 8 rejections (ocean.cpp:109-123)
  bool gauss_pair(mt19937 &entropy, float &re, float &im)
  {
    uniform_real_distribution<float> real11{-1.0f, 1.0f};

    float x = 0, y = 0, w = 1;

    for(int tries = 0; tries < 8 && !(0 < w && w < 1); ++tries)
    {
      x = real11(entropy);
      y = real11(entropy);
      w = x*x + y*y;
    }

    float s = sqrt((-2*log(w)) / w);

    re = x * s;
    im = y * s;

    return (0 < w && w < 1);
  }

  // h0(k) = seed(k) * dk * sqrt(P(k) / 2), k = dk * (n - N/2, m - N/2), dk = 2 pi / wavescale
  // (ocean.cpp:148-165; the same loop again at :194-211)
  void rebuild_height(OceanParams &params)
  {
    int const size = params.resolution;

    float dk = 2*pi<float>() / params.wavescale;

    for(int m = 0; m < size; ++m)
    {
      float ky = dk * (m - 0.5f*size);

      for(int n = 0; n < size; ++n)
      {
        float kx = dk * (n - 0.5f*size);

        float h0 = dk * sqrt(phillips(Vec2(kx, ky), params.waveamplitude, params.windspeed, params.winddirection) / 2.0f);

        size_t i = 2 * ((size_t)m * size + n);

        params.height[i+0] = params.seed[i+0] * h0;
        params.height[i+1] = params.seed[i+1] * h0;
      }
    }

    params.heightid = g_stateids++;
  }

  void seed_with(OceanParams &params, mt19937 &entropy)
  {
    size_t const points = (size_t)params.resolution * params.resolution;

    params.swellphase = 0.0f;
    params.rejectedseeds = 0;

    // row-major (m outer, n inner): the order fixes which draw lands on which wave vector (ocean.cpp:134-146)
    for(size_t i = 0; i < points; ++i)
    {
      if (!gauss_pair(entropy, params.seed[2*i+0], params.seed[2*i+1]))
      {
        // The reference evaluates sqrt(-2 log(w) / w) on a rejected w here and seeds a NaN (probability
        // 4.5e-6 per point: harmless odds at 64 x 64, near-certain from 512 x 512 up, and one NaN poisons
        // the whole transform).  Such a point gets no energy instead; the generator is consumed identically.
        params.seed[2*i+0] = 0.0f;
        params.seed[2*i+1] = 0.0f;
        params.rejectedseeds += 1;
      }
    }

    fill(params.phase.begin(), params.phase.end(), 0.0f);

    rebuild_height(params);

    params.flow = Vec2(0);
    params.updates.clear();
    params.firstupdate = 0;
    params.phaseupdates = 0;
    params.baselineage = 0;
    params.stateid = g_stateids++;
  }

  // update_ocean's phase loop (ocean.cpp:223-233) on the host, for one recorded step (OceanParams::hostphase only)
  void advance_host_phase(OceanParams &params, OceanParams::Update const &u)
  {
    int const N = params.resolution;

    for(int m = 0; m < N; ++m)
    {
      float const y = 2*pi<float>() * (m - 0.5f*N) / u.wavescale;

      for(int n = 0; n < N; ++n)
      {
        float const x = 2*pi<float>() * (n - 0.5f*N) / u.wavescale;

        float &ph = params.phase[(size_t)m*N + n];

        ph = fmod(ph + dispersion(Vec2(x, y)) * u.dt, 2*pi<float>());
      }
    }
  }

  // Park the state that is resident on the context's device (h0 + phase, device to device) under its ids.
  // `resume`: the slot about to be resumed (or null).  Where the state goes, in this order:
  //   1. its own slot, if it has one (an older copy of itself)
  //   2. with a resume slot: the context's SPARE buffer, which then trades places with the resume slot's buffer -- the slot
  //      being vacated becomes this state's slot, nobody is evicted (N states alternate on N - 1 slots + the spare)
  //   3. a new slot while there are fewer than MaxParkedStates
  //   4. the least recently used slot (its state goes back to its host copy + recorded history)
  // Returns true when case 2 applies: the caller swaps the buffers once it has resumed from the slot.
  bool park_bound_state(OceanContext &context, OceanContext::Parked *resume = nullptr)
  {
    if (context.boundstate == 0)
      return false;

    size_t const bytes = datum_ocean_state_bytes(context.resolution);

    OceanContext::Parked *slot = nullptr;

    for(auto &p : context.parked)
      if (p.stateid == context.boundstate)
        slot = &p;

    if (!slot && resume)
    {
      if (!context.spare)
        check(context.hip, datum_ocean_device_alloc(context.hip, bytes, &context.spare), "datum_ocean_device_alloc");

      check(context.hip, datum_ocean_park_state(context.hip, 0, context.spare, bytes, &context.spareflags), "datum_ocean_park_state");

      return true;
    }

    if (!slot && context.parked.size() < OceanContext::MaxParkedStates)
    {
      void *device = nullptr;

      // (allocated before the entry exists: a failed allocation leaves no half-made slot behind)
      check(context.hip, datum_ocean_device_alloc(context.hip, bytes, &device), "datum_ocean_device_alloc");

      context.parked.emplace_back();
      context.parked.back().device = device;

      slot = &context.parked.back();
    }

    if (!slot)
    {
      for(auto &p : context.parked)      // least recently used
        if (!slot || p.lastuse < slot->lastuse)
          slot = &p;
    }

    check(context.hip, datum_ocean_park_state(context.hip, 0, slot->device, bytes, &slot->flags), "datum_ocean_park_state");

    slot->stateid = context.boundstate;
    slot->heightid = context.boundheight;
    slot->appliedupdates = context.appliedupdates;
    slot->appliedlineage = context.appliedlineage;
    slot->lastuse = ++context.useclock;

    return false;
  }

  // make the device hold this params' state: h0 (and phase when the whole state was replaced), then every update_ocean
  // step of the state's history that this context has not applied yet, each under the wavescale it was issued with
  void bind_state(OceanContext &context, OceanParams const &params)
  {
    assert(params.resolution == context.resolution);

    uint64_t const last = params.firstupdate + params.updates.size();

    // what the device holds can be continued with this params' history if it is the same state, not further along than the
    // history goes, not behind what the history still records, and came out of this very history (lineage)
    auto continues = [&](uint64_t stateid, uint64_t applied, uint64_t lineage) {
      return stateid == params.stateid && applied >= params.firstupdate && applied <= last && params.lineage_at(applied) == lineage;
    };

    if (!continues(context.boundstate, context.appliedupdates, context.appliedlineage))
    {
      // The slot to resume is looked up FIRST: with more states than slots rendered round-robin the least recently used slot
      // is exactly the state that comes next, and the slot it vacates is where the bound state should go (park_bound_state).
      OceanContext::Parked *slot = nullptr;

      for(auto &p : context.parked)
        if (continues(p.stateid, p.appliedupdates, p.appliedlineage))
          slot = &p;

      // (with a resume slot park_bound_state never appends to the vector: `slot` stays valid)
      bool const trade = (context.boundstate != params.stateid) ? park_bound_state(context, slot) : false;

      if (slot)
      {
        // rendered here before: the parked phase is ahead of (or level with) params.phase
        check(context.hip, datum_ocean_resume_state(context.hip, 0, slot->device, datum_ocean_state_bytes(context.resolution), slot->flags), "datum_ocean_resume_state");

        OceanContext::Parked const resumed = *slot;

        if (trade)
        {
          // the vacated slot takes the spare buffer with the state that was bound; its old buffer is the new spare
          slot->device = context.spare;
          slot->flags = context.spareflags;
          slot->stateid = context.boundstate;
          slot->heightid = context.boundheight;
          slot->appliedupdates = context.appliedupdates;
          slot->appliedlineage = context.appliedlineage;

          context.spare = resumed.device;
        }

        slot->lastuse = ++context.useclock;

        context.boundstate = params.stateid;
        context.boundheight = resumed.heightid;
        context.appliedupdates = resumed.appliedupdates;
        context.appliedlineage = resumed.appliedlineage;
      }
      else
      {
        // params.phase is the state as of seed_ocean / the last fetch_ocean_state (or as far as update_ocean had to carry it
        // when the history was trimmed): the recorded history from there on comes on top
        if (params.phaseupdates < params.firstupdate)
          throw runtime_error("ocean: this context holds no copy of the state and the update history behind OceanParams::phase is no longer recorded: render or fetch_ocean_state a state at least every OceanParams::MaxRecordedUpdates / 2 update_ocean calls, or set OceanParams::hostphase");

        check(context.hip, datum_ocean_set_cascade(context.hip, 0, params.wavescale, params.choppiness), "datum_ocean_set_cascade");
        check(context.hip, datum_ocean_upload_state(context.hip, 0, params.height.data(), params.phase.data()), "datum_ocean_upload_state");

        context.boundstate = params.stateid;
        context.boundheight = params.heightid;
        context.appliedupdates = params.phaseupdates;
        context.appliedlineage = params.lineage_at(params.phaseupdates);
      }
    }

    // the steps issued before a lerp_ocean_waves advanced the phase with the dispersion of the OLD wave scale
    // (ocean.cpp:225-231 uses params.wavescale as it is at the call): replay each step under its own
    for(uint64_t i = context.appliedupdates; i < last; ++i)
    {
      OceanParams::Update const &u = params.updates[i - params.firstupdate];

      check(context.hip, datum_ocean_set_cascade(context.hip, 0, u.wavescale, params.choppiness), "datum_ocean_set_cascade");
      check(context.hip, datum_ocean_update(context.hip, u.dt), "datum_ocean_update");
    }

    context.appliedupdates = last;
    context.appliedlineage = params.lineage_at(last);

    // (a change of wave scale applies the updates queued on the device under the old one first: datum_ocean_set_cascade)
    check(context.hip, datum_ocean_set_cascade(context.hip, 0, params.wavescale, params.choppiness), "datum_ocean_set_cascade");

    if (context.boundheight != params.heightid && params.deviceheight)
    {
      // lerp_ocean_waves changed the wave parameters: rebuild h0 on the device from the resident seed
      if (context.boundseed != params.stateid)
      {
        check(context.hip, datum_ocean_upload_seed(context.hip, 0, params.seed.data()), "datum_ocean_upload_seed");

        context.boundseed = params.stateid;
      }

      check(context.hip, datum_ocean_rebuild_height(context.hip, 0, params.wavescale, params.waveamplitude, params.windspeed, params.winddirection.x, params.winddirection.y), "datum_ocean_rebuild_height");

      context.boundheight = params.heightid;
    }
    else if (context.boundheight != params.heightid)
    {
      // lerp_ocean_waves changed h0 only: keep the device phase, which is ahead of params.phase
      vector<float> phase(params.phase.size());

      check(context.hip, datum_ocean_read_state(context.hip, 0, phase.data()), "datum_ocean_read_state");
      check(context.hip, datum_ocean_upload_state(context.hip, 0, params.height.data(), phase.data()), "datum_ocean_upload_state");

      context.boundheight = params.heightid;
    }
  }
}


//|---------------------- Camera --------------------------------------------
//|--------------------------------------------------------------------------

///////////////////////// Camera::Constructor ///////////////////////////////
Camera::Camera()
{
  // camera.cpp:20-31
  m_fov = 60.0f*pi<float>()/180.0f;
  m_aspect = 1.7777f;
  m_znear = 0.1f;
  m_zfar = 1000.0f;
  m_position = Vec3(0);
  m_rotation = Quaternion3(1, 0, 0, 0);
}


///////////////////////// Camera::set_projection ////////////////////////////
void Camera::set_projection(float fov, float aspect, float znear, float zfar)
{
  m_fov = fov;
  m_aspect = aspect;
  m_znear = znear;
  m_zfar = zfar;
}


///////////////////////// Camera::proj //////////////////////////////////////
Matrix4f Camera::proj() const
{
  // y flipped, reverse z (camera.cpp:77-91)
  Matrix4f proj = {};

  float halftan = tan(m_fov/2);
  float depth = m_zfar - m_znear;

  proj(0, 0) = 1 / (m_aspect * halftan);
  proj(1, 1) = -1 / halftan;
  proj(2, 2) = m_zfar / depth - 1;
  proj(2, 3) = m_zfar * m_znear / depth;
  proj(3, 2) = -1;

  return proj;
}


///////////////////////// Camera::lookat ////////////////////////////////////
void Camera::lookat(Vec3 const &position, Vec3 const &target, Vec3 const &up)
{
  m_position = position;
  m_rotation = Transform::lookat(position, target, up).rotation();
}


//|---------------------- Ocean ---------------------------------------------
//|--------------------------------------------------------------------------

///////////////////////// OceanParams::Constructor //////////////////////////
OceanParams::OceanParams(int resolution)
  : resolution(resolution),
    seed((size_t)resolution * resolution * 2, 0.0f),
    height((size_t)resolution * resolution * 2, 0.0f),
    phase((size_t)resolution * resolution, 0.0f),
    hostphase(resolution <= OceanContext::WaveResolution),
    stateid(g_stateids++)
{
}


///////////////////////// OceanContext::Destructor //////////////////////////
OceanContext::~OceanContext()
{
  if (hip)
  {
    for(auto &p : parked)
      datum_ocean_device_free(hip, p.device);

    if (spare)
      datum_ocean_device_free(hip, spare);

    datum_ocean_destroy(hip);
  }
}


///////////////////////// release_parked_states /////////////////////////////
size_t release_parked_states(OceanContext &context, OceanParams const *keep)
{
  size_t freed = 0;

  for(size_t i = 0; i < context.parked.size(); )
  {
    if (keep && context.parked[i].stateid == keep->stateid)
    {
      ++i;
      continue;
    }

    if (context.hip && context.parked[i].device)
      datum_ocean_device_free(context.hip, context.parked[i].device);

    freed += datum_ocean_state_bytes(context.resolution);

    context.parked.erase(context.parked.begin() + i);
  }

  if (context.spare && context.hip)
  {
    datum_ocean_device_free(context.hip, context.spare);
    context.spare = nullptr;

    freed += datum_ocean_state_bytes(context.resolution);
  }

  return freed;
}


///////////////////////// to_pod / from_pod /////////////////////////////////
bool to_pod(OceanParams const &params, OceanParamsPod &pod)
{
  if (params.resolution != OceanContext::WaveResolution)
    throw runtime_error("ocean: OceanParamsPod is the reference's 64 x 64 OceanParams");

  // the host phase must contain every recorded step (fetch_ocean_state, or hostphase)
  if (params.phaseupdates != params.firstupdate + params.updates.size())
    return false;

  pod.plane = params.plane;
  pod.swelllength = params.swelllength;
  pod.swellamplitude = params.swellamplitude;
  pod.swellsteepness = params.swellsteepness;
  pod.swellspeed = params.swellspeed;
  pod.swelldirection = params.swelldirection;
  pod.wavescale = params.wavescale;
  pod.waveamplitude = params.waveamplitude;
  pod.windspeed = params.windspeed;
  pod.winddirection = params.winddirection;
  pod.choppiness = params.choppiness;
  pod.smoothing = params.smoothing;
  pod.swellphase = params.swellphase;
  pod.flow = params.flow;

  memcpy(pod.seed, params.seed.data(), sizeof(pod.seed));
  memcpy(pod.height, params.height.data(), sizeof(pod.height));
  memcpy(pod.phase, params.phase.data(), sizeof(pod.phase));

  return true;
}

OceanParams from_pod(OceanParamsPod const &pod)
{
  OceanParams params(OceanContext::WaveResolution);

  params.plane = pod.plane;
  params.swelllength = pod.swelllength;
  params.swellamplitude = pod.swellamplitude;
  params.swellsteepness = pod.swellsteepness;
  params.swellspeed = pod.swellspeed;
  params.swelldirection = pod.swelldirection;
  params.wavescale = pod.wavescale;
  params.waveamplitude = pod.waveamplitude;
  params.windspeed = pod.windspeed;
  params.winddirection = pod.winddirection;
  params.choppiness = pod.choppiness;
  params.smoothing = pod.smoothing;
  params.swellphase = pod.swellphase;
  params.flow = pod.flow;

  memcpy(params.seed.data(), pod.seed, sizeof(pod.seed));
  memcpy(params.height.data(), pod.height, sizeof(pod.height));
  memcpy(params.phase.data(), pod.phase, sizeof(pod.phase));

  params.heightid = g_stateids++;

  return params;
}


///////////////////////// seed_ocean ////////////////////////////////////////
void seed_ocean(OceanParams &params)
{
  mt19937 entropy(random_device{}());

  seed_with(params, entropy);
}

void seed_ocean(OceanParams &params, uint32_t rngseed)
{
  mt19937 entropy(rngseed);

  seed_with(params, entropy);
}


///////////////////////// lerp_ocean_swell /////////////////////////////////
void lerp_ocean_swell(OceanParams &params, float swelllength, float swellamplitude, float swellspeed, Vec2 swelldirection, float t)
{
  bool same = (params.swelllength == swelllength && params.swellamplitude == swellamplitude && params.swellspeed == swellspeed && params.swelldirection == swelldirection);

  if (same)
    return;

  params.swelllength = lerp(params.swelllength, swelllength, t);
  params.swellamplitude = lerp(params.swellamplitude, swellamplitude, t);
  params.swellspeed = lerp(params.swellspeed, swellspeed, t);
  params.swelldirection = normalise(lerp(params.swelldirection, swelldirection, t));
}


///////////////////////// lerp_ocean_waves //////////////////////////////////
void lerp_ocean_waves(OceanParams &params, float wavescale, float waveamplitude, float windspeed, Vec2 winddirection, float t)
{
  bool same = (params.wavescale == wavescale && params.waveamplitude == waveamplitude && params.windspeed == windspeed && params.winddirection == winddirection);

  if (same)
    return;

  params.wavescale = lerp(params.wavescale, wavescale, t);
  params.waveamplitude = lerp(params.waveamplitude, waveamplitude, t);
  params.windspeed = lerp(params.windspeed, windspeed, t);
  params.winddirection = normalise(lerp(params.winddirection, winddirection, t));

  // the spectrum envelope moved: new h0 from the stored seed (ocean.cpp:194-211), here or on the device
  if (params.deviceheight)
    params.heightid = g_stateids++;
  else
    rebuild_height(params);
}


///////////////////////// update_ocean ////////////////////////////////////
void update_ocean(OceanParams &params, float dt)
{
  // hostphase switched on late: the steps between the host phase and the recorded history are gone (trimmed while it was
  // off), so the host copy cannot be brought up to date from here -- fetch_ocean_state first.  Nothing is touched.
  if (params.hostphase && params.phaseupdates < params.firstupdate)
    throw runtime_error("ocean: OceanParams::hostphase was set after more than OceanParams::MaxRecordedUpdates update_ocean calls without it: the history behind OceanParams::phase is no longer recorded; fetch_ocean_state the params first, then set hostphase");

  params.swellphase = fmod(params.swellphase + (params.swellspeed * 2*pi<float>()/params.swelllength)*dt, 2*pi<float>());

  // phase[m][n] = fmod(phase[m][n] + dispersion(k)*dt, 2 pi) is done by the row-pass kernel, in history order, with the
  // dispersion of the wave scale in force now
  uint32_t dtbits, wsbits;
  memcpy(&dtbits, &dt, 4);
  memcpy(&wsbits, &params.wavescale, 4);

  uint64_t const prev = params.updates.empty() ? params.baselineage : params.updates.back().lineage;
  uint64_t const lineage = ((prev ^ dtbits) * 0x100000001b3ull ^ wsbits) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull;

  params.updates.push_back(OceanParams::Update{ dt, params.wavescale, lineage });

  // extension: keep the host copy of the phase current as the reference does (ocean.cpp:223-233), for a params that must be
  // renderable by a context that has never seen it at any time, however long ago it was last rendered or fetched
  if (params.hostphase)
  {
    while (params.phaseupdates < params.firstupdate + params.updates.size())
    {
      advance_host_phase(params, params.updates[params.phaseupdates - params.firstupdate]);

      params.phaseupdates += 1;
    }
  }

  if (params.updates.size() > OceanParams::MaxRecordedUpdates)
  {
    size_t const drop = params.updates.size() - OceanParams::MaxRecordedUpdates / 2;

    params.baselineage = params.updates[drop - 1].lineage;
    params.updates.erase(params.updates.begin(), params.updates.begin() + drop);
    params.firstupdate += drop;
  }

  params.flow += params.windspeed * params.winddirection * dt;
}


///////////////////////// ResourceManager::create ///////////////////////////
template<>
Ocean const *ResourceManager::create<Ocean>(int sizex, int sizey)
{
  OceanContext &context = *m_context;

  assert(context.hip);
  assert(sizex >= 2 && sizey >= 2);

  auto ocean = new Ocean;

  ocean->context = &context;
  ocean->sizex = sizex;
  ocean->sizey = sizey;
  ocean->state = Mesh::State::Empty;

  size_t cells = (size_t)(sizex-1) * (sizey-1);

  ocean->vertexbuffer.vertexcount = sizex * sizey;
  ocean->vertexbuffer.vertexsize = sizeof(Mesh::Vertex);
  ocean->vertexbuffer.indexcount = 6 * cells;
  ocean->vertexbuffer.indexsize = sizeof(uint32_t);

  if (datum_ocean_device_alloc(context.hip, (size_t)sizex * sizey * sizeof(Mesh::Vertex), &ocean->vertexbuffer.vertices) != DATUM_OCEAN_OK
   || datum_ocean_device_alloc(context.hip, 6 * cells * sizeof(uint32_t), &ocean->vertexbuffer.indices) != DATUM_OCEAN_OK)
  {
    string why = datum_ocean_last_error(context.hip);
    destroy<Ocean>(ocean);
    throw runtime_error("HIP Create VertexBuffer failed: " + why);
  }

  // two triangles per grid cell; with a = (x, y), b = (x+1, y), c = (x, y+1), d = (x+1, y+1):
  // (c, a, d) and (d, a, b) (ocean.cpp:275-286)
  vector<uint32_t> indices;
  indices.reserve(6 * cells);

  for(int y = 0; y + 1 < sizey; ++y)
  {
    for(int x = 0; x + 1 < sizex; ++x)
    {
      uint32_t a = y * sizex + x;
      uint32_t b = a + 1;
      uint32_t c = a + sizex;
      uint32_t d = c + 1;

      indices.insert(indices.end(), { c, a, d, d, a, b });
    }
  }

  check(context.hip, datum_ocean_device_write(context.hip, ocean->vertexbuffer.indices, indices.data(), indices.size() * sizeof(uint32_t)), "datum_ocean_device_write");

  ocean->state = Mesh::State::Ready;

  return ocean;
}


///////////////////////// ResourceManager::release //////////////////////////
template<>
void ResourceManager::release<Ocean>(Ocean const *ocean)
{
  // the reference defers the destroy until the frame that may still use it has retired
  // (ocean.cpp:304-308); here the stream is drained by datum_ocean_device_free
  destroy<Ocean>(ocean);
}


///////////////////////// ResourceManager::destroy //////////////////////////
template<>
void ResourceManager::destroy<Ocean>(Ocean const *ocean)
{
  if (ocean)
  {
    if (ocean->context && ocean->context->hip)
    {
      if (ocean->vertexbuffer.vertices)
        datum_ocean_device_free(ocean->context->hip, ocean->vertexbuffer.vertices);

      if (ocean->vertexbuffer.indices)
        datum_ocean_device_free(ocean->context->hip, ocean->vertexbuffer.indices);
    }

    delete ocean;
  }
}


///////////////////////// initialise_ocean_context //////////////////////////
void initialise_ocean_context(DatumPlatform::PlatformInterface &platform, OceanContext &context, uint32_t queueindex)
{
  // the reference picks queue `queueindex` of the platform's render device (ocean.cpp:331-333) and creates
  // its command buffer, fence and semaphore; here the HIP module owns one stream per handle, created in
  // prepare_ocean_context, so only the device is recorded
  (void)queueindex;

  // the shared object carries no soname: refuse a libdatum_ocean_hip.so built from another revision of the header (its error codes,
  // signatures or the layout of bound map buffers may differ) before anything is called through it
  if (datum_ocean_abi_version() != DATUM_OCEAN_ABI_VERSION)
    throw runtime_error("HIP ocean module: libdatum_ocean_hip.so reports ABI version " + to_string(datum_ocean_abi_version()) + ", this host library was built against " + to_string(DATUM_OCEAN_ABI_VERSION));

  context.device = platform.hipdevice;
}


///////////////////////// prepare_ocean_context /////////////////////////////
bool prepare_ocean_context(DatumPlatform::PlatformInterface &platform, OceanContext &context, AssetManager &assets)
{
  (void)platform;
  (void)assets;

  if (context.ready)
    return true;

  // pipelines, twiddles, work spectrum, displacement map (ocean.cpp:353-711) in one call; there are no
  // streamed shader assets to wait for, so this never returns false -- it throws if the device refuses
  int rc = datum_ocean_create(&context.hip, context.device, context.resolution, 1);

  if (rc != DATUM_OCEAN_OK)
    throw runtime_error(string("HIP ocean module create failed: ") + datum_ocean_last_error(nullptr));

  if (context.spectrumfp16)
    check(context.hip, datum_ocean_set_spectrum_format(context.hip, context.heightfp16 ? DATUM_OCEAN_SPECTRUM_FP16_H0 : DATUM_OCEAN_SPECTRUM_FP16), "datum_ocean_set_spectrum_format");

  if (context.literaltransform)
    check(context.hip, datum_ocean_set_literal_transform(context.hip, 1), "datum_ocean_set_literal_transform");

  context.ready = true;

  return true;
}


///////////////////////// make_oceanset /////////////////////////////////////
datum_ocean_set make_oceanset(Camera const &camera, OceanParams const &params)
{
  datum_ocean_set set = {};

  Matrix4f proj = camera.proj();
  Matrix4f invproj = inverse(proj);
  Transform transform = camera.transform();

  memcpy(set.proj, proj.m, sizeof(set.proj));
  memcpy(set.invproj, invproj.m, sizeof(set.invproj));

  float real[4] = { transform.real.w, transform.real.x, transform.real.y, transform.real.z };
  float dual[4] = { transform.dual.w, transform.dual.x, transform.dual.y, transform.dual.z };

  memcpy(set.camera_real, real, sizeof(real));
  memcpy(set.camera_dual, dual, sizeof(dual));

  set.plane[0] = params.plane.normal.x;
  set.plane[1] = params.plane.normal.y;
  set.plane[2] = params.plane.normal.z;
  set.plane[3] = params.plane.distance;

  set.swelllength = params.swelllength;
  set.swellamplitude = params.swellamplitude;
  set.swellsteepness = params.swellsteepness;
  set.swellphase = params.swellphase;
  set.swelldirection[0] = params.swelldirection.x;
  set.swelldirection[1] = params.swelldirection.y;

  set.scale = 1 / params.wavescale;
  set.choppiness = params.choppiness;
  set.smoothing = 1 / params.smoothing;

  set.size = params.resolution;

  return set;
}


///////////////////////// displace_ocean_surface ////////////////////////////
void displace_ocean_surface(OceanContext &context, OceanParams const &params)
{
  assert(context.ready);

  bind_state(context, params);

  check(context.hip, datum_ocean_displace(context.hip), "datum_ocean_displace");
}


///////////////////////// render ////////////////////////////////////////////
void render_ocean_surface(OceanContext &context, Ocean const *target, Camera const &camera, OceanParams const &params, void *const (&dependancies)[8])
{
  assert(context.ready);
  assert(target->ready());

  // no fence wait: work is ordered on the handle's stream and nothing on the host is overwritten
  // (the reference waits because it rewrites the mapped oceanset, ocean.cpp:725-749)

  for(void *dependancy : dependancies)
  {
    if (dependancy)
      check(context.hip, datum_ocean_wait_event(context.hip, dependancy), "datum_ocean_wait_event");
  }

  datum_ocean_set set = make_oceanset(camera, params);

  // sim -> fftx -> ffty -> map (ocean.cpp:769-789)
  displace_ocean_surface(context, params);

  // gen (ocean.cpp:791-793)
  check(context.hip, datum_ocean_gen(context.hip, 0, &set, target->sizex, target->sizey, target->vertexbuffer.vertices), "datum_ocean_gen");

  // submit signals rendercomplete (ocean.cpp:803)
  check(context.hip, datum_ocean_signal(context.hip, &context.rendercomplete), "datum_ocean_signal");
}


///////////////////////// fetch_ocean_state /////////////////////////////////
void fetch_ocean_state(OceanContext &context, OceanParams &params)
{
  assert(context.ready);

  bind_state(context, params);

  check(context.hip, datum_ocean_read_state(context.hip, 0, params.phase.data()), "datum_ocean_read_state");

  // the host phase now contains the whole history: none of it needs to be kept for a later upload of these params
  params.baselineage = params.lineage_at(params.firstupdate + params.updates.size());
  params.phaseupdates = params.firstupdate + params.updates.size();
  params.firstupdate = params.phaseupdates;
  params.updates.clear();

  if (params.deviceheight)
    check(context.hip, datum_ocean_read_height(context.hip, 0, params.height.data()), "datum_ocean_read_height");
}


///////////////////////// read_ocean_displacement ///////////////////////////
void read_ocean_displacement(OceanContext &context, float *maps)
{
  assert(context.ready);

  check(context.hip, datum_ocean_read_maps(context.hip, 0, maps), "datum_ocean_read_maps");
}


///////////////////////// read_ocean_vertices ///////////////////////////////
void read_ocean_vertices(OceanContext &context, Ocean const *ocean, Mesh::Vertex *vertices)
{
  assert(context.ready);

  check(context.hip, datum_ocean_device_read(context.hip, vertices, ocean->vertexbuffer.vertices, (size_t)ocean->sizex * ocean->sizey * sizeof(Mesh::Vertex)), "datum_ocean_device_read");
}


///////////////////////// ocean_twiddle_table ///////////////////////////////
vector<float> ocean_twiddle_table(int resolution)
{
  int stages = 0;
  while ((1 << stages) < resolution)
    ++stages;

  vector<float> weights((size_t)resolution * 2 * stages);

  if (datum_ocean_reference_weights(resolution, weights.data()) != DATUM_OCEAN_OK)
    throw runtime_error(string("ocean_twiddle_table: ") + datum_ocean_last_error(nullptr));

  return weights;
}
