//
// host_capi.cpp -- a flat C view of the C++ ocean API (ocean.h) so that the Python tests and bench.py
// can drive the host shim exactly as game code would (examples/ocean/ocean.cpp:31,46-52,59,135,165,179).
// Exceptions become a return code plus a message.
//

#include "ocean.h"

#include <cstdio>
#include <cstring>
#include <exception>

using namespace lml;

extern "C"
{
  struct datum_host_scalars
  {
    float plane[4];
    float swelllength, swellamplitude, swellsteepness, swellspeed;
    float swelldirection[2];
    float wavescale, waveamplitude, windspeed;
    float winddirection[2];
    float choppiness, smoothing;
    float swellphase;
    float flow[2];
    int resolution;
    int rejectedseeds;
    int pending;
  };

  struct datum_host_camera
  {
    float fov, aspect, znear, zfar;
    float position[3], target[3], up[3];
  };
}

namespace
{
  struct HostContext
  {
    DatumPlatform::PlatformInterface platform;
    AssetManager assets;
    OceanContext context;
    ResourceManager resources{context};
  };

  thread_local char g_error[512];

  int caught(std::exception const &e)
  {
    snprintf(g_error, sizeof(g_error), "%s", e.what());
    return -1;
  }

  Camera make_camera(datum_host_camera const &c)
  {
    Camera camera;
    camera.set_projection(c.fov, c.aspect, c.znear, c.zfar);
    camera.lookat(Vec3(c.position[0], c.position[1], c.position[2]), Vec3(c.target[0], c.target[1], c.target[2]), Vec3(c.up[0], c.up[1], c.up[2]));
    return camera;
  }
}

extern "C"
{
  char const *datum_host_last_error() { return g_error; }

  void *datum_host_params_create(int resolution) { return new OceanParams(resolution); }
  void datum_host_params_destroy(void *p) { delete static_cast<OceanParams*>(p); }
  void *datum_host_params_clone(void *p) { return new OceanParams(*static_cast<OceanParams*>(p)); }       // OceanParams is copyable, like the reference's POD

  void datum_host_params_get(void *p, datum_host_scalars *s)
  {
    OceanParams const &o = *static_cast<OceanParams*>(p);

    s->plane[0] = o.plane.normal.x; s->plane[1] = o.plane.normal.y; s->plane[2] = o.plane.normal.z; s->plane[3] = o.plane.distance;
    s->swelllength = o.swelllength; s->swellamplitude = o.swellamplitude; s->swellsteepness = o.swellsteepness; s->swellspeed = o.swellspeed;
    s->swelldirection[0] = o.swelldirection.x; s->swelldirection[1] = o.swelldirection.y;
    s->wavescale = o.wavescale; s->waveamplitude = o.waveamplitude; s->windspeed = o.windspeed;
    s->winddirection[0] = o.winddirection.x; s->winddirection[1] = o.winddirection.y;
    s->choppiness = o.choppiness; s->smoothing = o.smoothing;
    s->swellphase = o.swellphase;
    s->flow[0] = o.flow.x; s->flow[1] = o.flow.y;
    s->resolution = o.resolution;
    s->rejectedseeds = o.rejectedseeds;
    s->pending = (int)o.updates.size();      // recorded update_ocean calls (OceanParams::updates)
  }

  // tunables only; state (swellphase, flow, arrays) is left alone
  void datum_host_params_set(void *p, datum_host_scalars const *s)
  {
    OceanParams &o = *static_cast<OceanParams*>(p);

    o.plane = { { s->plane[0], s->plane[1], s->plane[2] }, s->plane[3] };
    o.swelllength = s->swelllength; o.swellamplitude = s->swellamplitude; o.swellsteepness = s->swellsteepness; o.swellspeed = s->swellspeed;
    o.swelldirection = Vec2(s->swelldirection[0], s->swelldirection[1]);
    o.wavescale = s->wavescale; o.waveamplitude = s->waveamplitude; o.windspeed = s->windspeed;
    o.winddirection = Vec2(s->winddirection[0], s->winddirection[1]);
    o.choppiness = s->choppiness; o.smoothing = s->smoothing;
  }

  void datum_host_params_set_deviceheight(void *p, int on) { static_cast<OceanParams*>(p)->deviceheight = (on != 0); }
  // 0, or -1 when the history behind the host phase is gone (hostphase stays off: fetch_ocean_state first)
  int datum_host_params_set_hostphase(void *p, int on)
  {
    OceanParams &o = *static_cast<OceanParams*>(p);

    if (on && o.phaseupdates < o.firstupdate)
    {
      snprintf(g_error, sizeof(g_error), "hostphase: the update history behind OceanParams::phase is no longer recorded; fetch_ocean_state first");
      return -1;
    }

    o.hostphase = (on != 0);

    return 0;
  }

  // the field itself, as C++ code would set it (no validation: update_ocean then throws)
  void datum_host_params_poke_hostphase(void *p, int on) { static_cast<OceanParams*>(p)->hostphase = (on != 0); }

  // the reference's POD (OceanParamsPod, 82000 bytes): 0 / 1 = not current (fetch_ocean_state first) / -1 = not a 64 x 64 params
  int datum_host_params_to_pod(void *p, void *pod)
  {
    try
    {
      return to_pod(*static_cast<OceanParams*>(p), *static_cast<OceanParamsPod*>(pod)) ? 0 : 1;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  void *datum_host_params_from_pod(void const *pod) { return new OceanParams(from_pod(*static_cast<OceanParamsPod const*>(pod))); }

  int datum_host_pod_bytes() { return (int)sizeof(OceanParamsPod); }

  float *datum_host_params_seed(void *p) { return static_cast<OceanParams*>(p)->seed.data(); }
  float *datum_host_params_height(void *p) { return static_cast<OceanParams*>(p)->height.data(); }
  float *datum_host_params_phase(void *p) { return static_cast<OceanParams*>(p)->phase.data(); }

  void datum_host_seed_ocean(void *p, uint32_t rngseed, int use_random_device)
  {
    if (use_random_device)
      seed_ocean(*static_cast<OceanParams*>(p));
    else
      seed_ocean(*static_cast<OceanParams*>(p), rngseed);
  }

  void datum_host_lerp_ocean_swell(void *p, float swelllength, float swellamplitude, float swellspeed, float dx, float dy, float t)
  {
    lerp_ocean_swell(*static_cast<OceanParams*>(p), swelllength, swellamplitude, swellspeed, Vec2(dx, dy), t);
  }

  void datum_host_lerp_ocean_waves(void *p, float wavescale, float waveamplitude, float windspeed, float dx, float dy, float t)
  {
    lerp_ocean_waves(*static_cast<OceanParams*>(p), wavescale, waveamplitude, windspeed, Vec2(dx, dy), t);
  }

  int datum_host_update_ocean(void *p, float dt)
  {
    try
    {
      update_ocean(*static_cast<OceanParams*>(p), dt);
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  void datum_host_make_oceanset(datum_host_camera const *camera, void *p, datum_ocean_set *out)
  {
    *out = make_oceanset(make_camera(*camera), *static_cast<OceanParams*>(p));
  }

  int datum_host_twiddle_table(int resolution, float *weights)
  {
    try
    {
      auto w = ocean_twiddle_table(resolution);
      memcpy(weights, w.data(), w.size() * sizeof(float));
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  // initialise_ocean_context + prepare_ocean_context (examples/ocean/ocean.cpp:31,165); flags: 1 = OceanContext::spectrumfp16, 2 = ::literaltransform, 4 = ::heightfp16
  void *datum_host_context_create_ex(int device, int resolution, int flags)
  {
    HostContext *hc = nullptr;

    try
    {
      hc = new HostContext;
      hc->platform.hipdevice = device;
      hc->context.resolution = resolution;
      hc->context.spectrumfp16 = (flags & 1) != 0;
      hc->context.literaltransform = (flags & 2) != 0;
      hc->context.heightfp16 = (flags & 4) != 0;

      initialise_ocean_context(hc->platform, hc->context, 0);

      while (!prepare_ocean_context(hc->platform, hc->context, hc->assets))
        ;

      return hc;
    }
    catch(std::exception const &e)
    {
      caught(e);
      delete hc;
      return nullptr;
    }
  }

  void *datum_host_context_create(int device, int resolution) { return datum_host_context_create_ex(device, resolution, 0); }

  void datum_host_context_destroy(void *c) { delete static_cast<HostContext*>(c); }

  void *datum_host_context_handle(void *c) { return static_cast<HostContext*>(c)->context.hip; }

  void *datum_host_ocean_create(void *c, int sizex, int sizey)
  {
    try
    {
      return const_cast<Ocean*>(static_cast<HostContext*>(c)->resources.create<Ocean>(sizex, sizey));
    }
    catch(std::exception const &e)
    {
      caught(e);
      return nullptr;
    }
  }

  void datum_host_ocean_release(void *c, void *ocean) { static_cast<HostContext*>(c)->resources.release<Ocean>(static_cast<Ocean const*>(ocean)); }

  void *datum_host_ocean_vertices(void *ocean) { return static_cast<Ocean const*>(ocean)->vertexbuffer.vertices; }

  int datum_host_ocean_indices(void *c, void *ocean, uint32_t *indices)
  {
    Ocean const *o = static_cast<Ocean const*>(ocean);

    return datum_ocean_device_read(static_cast<HostContext*>(c)->context.hip, indices, o->vertexbuffer.indices, (size_t)o->vertexbuffer.indexcount * sizeof(uint32_t));
  }

  int datum_host_render_ocean_surface(void *c, void *ocean, datum_host_camera const *camera, void *p)
  {
    try
    {
      render_ocean_surface(static_cast<HostContext*>(c)->context, static_cast<Ocean const*>(ocean), make_camera(*camera), *static_cast<OceanParams*>(p));
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  int datum_host_displace_ocean_surface(void *c, void *p)
  {
    try
    {
      displace_ocean_surface(static_cast<HostContext*>(c)->context, *static_cast<OceanParams*>(p));
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  size_t datum_host_release_parked_states(void *c, void *keep)
  {
    return release_parked_states(static_cast<HostContext*>(c)->context, static_cast<OceanParams const*>(keep));
  }

  int datum_host_parked_states(void *c) { return (int)static_cast<HostContext*>(c)->context.parked.size(); }

  int datum_host_fetch_ocean_state(void *c, void *p)
  {
    try
    {
      fetch_ocean_state(static_cast<HostContext*>(c)->context, *static_cast<OceanParams*>(p));
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  int datum_host_read_ocean_displacement(void *c, float *maps)
  {
    try
    {
      read_ocean_displacement(static_cast<HostContext*>(c)->context, maps);
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }

  int datum_host_read_ocean_vertices(void *c, void *ocean, float *vertices)
  {
    try
    {
      read_ocean_vertices(static_cast<HostContext*>(c)->context, static_cast<Ocean const*>(ocean), reinterpret_cast<Mesh::Vertex*>(vertices));
      return 0;
    }
    catch(std::exception const &e) { return caught(e); }
  }
}
