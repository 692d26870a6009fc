// The two storage blocks every ocean shader of datum declares (data/ocean.sim.comp:8-39 and its siblings), with
// WaveResolution a compile-time macro (-DWAVE_RESOLUTION=N; the reference hard-wires 64, src/renderer/ocean.h:16,
// data/ocean.*.comp "const uint WaveResolution = 64").  std430, row_major: byte offsets as src/renderer/ocean.cpp:33-68.
#ifndef WAVE_RESOLUTION
#define WAVE_RESOLUTION 64
#endif

const uint WaveResolution = WAVE_RESOLUTION;
const uint WavePoints = WaveResolution * WaveResolution;

struct DualQuat { vec4 real; vec4 dual; };          // data/transform.inc: (w, x, y, z) each

layout(set=0, binding=0, std430, row_major) readonly buffer OceanSet
{
  mat4 proj;
  mat4 invproj;
  DualQuat camera;
  vec4 plane;
  float swelllength, swellamplitude, swellsteepness, swellphase;
  vec2 swelldirection;
  float scale;              // 1 / wavescale
  float choppiness;
  float smoothing;          // 1 / params.smoothing
  uint size;
  vec2 h0[WavePoints];
  float phase[WavePoints];
} ocean;

layout(set=0, binding=1, std430, row_major) SPECTRUM_ACCESS buffer Spectrum
{
  vec2 h[WavePoints];
  vec2 hx[WavePoints];
  vec2 hy[WavePoints];
  float weights[WavePoints];      // twiddles: row i, columns 2 s and 2 s + 1 = cos, sin of -2 pi i / 2^(s+1) (ocean.cpp:686-700)
} spectrum;
