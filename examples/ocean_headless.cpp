//
// ocean_headless.cpp -- datum's example-ocean (examples/ocean/ocean.cpp:22-66,135,165-203) without a window:
// the same calls in the same order against datum_amd/host/ocean.h, then a few probe values on stdout so that a
// test can compare them with the CPU oracle.  Usage: ocean_headless [resolution=64] [frames=60] [seed=1000]
//

#include "../datum_amd/host/ocean.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace lml;

int main(int argc, char **argv)
{
  int resolution = (argc > 1) ? atoi(argv[1]) : 64;
  int frames = (argc > 2) ? atoi(argv[2]) : 60;
  unsigned seed = (argc > 3) ? (unsigned)atoi(argv[3]) : 1000u;

  try
  {
    DatumPlatform::PlatformInterface platform;
    AssetManager assets;

    OceanContext oceancontext;
    oceancontext.resolution = resolution;

    ResourceManager resources(oceancontext);

    initialise_ocean_context(platform, oceancontext, 0);                    // example_init, :31

    Camera camera;
    camera.set_projection(60.0f*pi<float>()/180.0f, 1920.0f/1080.0f);       // :33 with ocean.h:17-18

    OceanParams ocean(resolution);
    ocean.wavescale = 22.0f;                                                // :46-50
    ocean.waveamplitude = 0.0025f;
    ocean.swellamplitude = 0.8f;
    ocean.windspeed = 7.9f;
    ocean.smoothing = 320.0f;

    seed_ocean(ocean, seed);                                                // :52 (explicit entropy)

    while (!prepare_ocean_context(platform, oceancontext, assets))          // example_render Startup branch, :165
      ;

    Ocean const *oceanmesh = resources.create<Ocean>(64, 64);               // :59 (1024 x 1024 in the example)

    camera.lookat(Vec3(0, 0, 8), Vec3(1, 0, 8), Vec3(0, 0, 1));             // :63

    for(int frame = 0; frame < frames; ++frame)
    {
      update_ocean(ocean, 1.0f/60);                                         // example_update, :135
      render_ocean_surface(oceancontext, oceanmesh, camera, ocean);         // example_render, :179
    }

    fetch_ocean_state(oceancontext, ocean);

    std::vector<float> maps((size_t)2 * resolution * resolution * 4);
    read_ocean_displacement(oceancontext, maps.data());

    std::vector<Mesh::Vertex> vertices(64 * 64);
    read_ocean_vertices(oceancontext, oceanmesh, vertices.data());

    int const N = resolution;
    auto texel = [&](int layer, int y, int x, int c) { return maps[(((size_t)layer * N + y) * N + x) * 4 + c]; };

    double sumsq = 0;
    for(int y = 0; y < N; ++y)
      for(int x = 0; x < N; ++x)
        sumsq += (double)texel(0, y, x, 2) * texel(0, y, x, 2);

    printf("resolution %d frames %d seed %u rejectedseeds %d\n", resolution, frames, seed, ocean.rejectedseeds);
    printf("swellphase %.9g flow %.9g %.9g\n", ocean.swellphase, ocean.flow.x, ocean.flow.y);
    printf("phase[10][20] %.9g phase[%d][%d] %.9g\n", ocean.phase[(size_t)10 * N + 20], N - 1, N - 1, ocean.phase[(size_t)N * N - 1]);
    printf("dz_rms %.9g\n", sqrt(sumsq / ((double)N * N)));
    printf("map[0][5][7] %.9g %.9g %.9g map[1][5][7] %.9g %.9g %.9g\n", texel(0, 5, 7, 0), texel(0, 5, 7, 1), texel(0, 5, 7, 2), texel(1, 5, 7, 0), texel(1, 5, 7, 1), texel(1, 5, 7, 2));

    Mesh::Vertex const &v = vertices[40 * 64 + 33];
    printf("vertex[40][33] pos %.9g %.9g %.9g uv %.9g %.9g n %.9g %.9g %.9g t %.9g %.9g %.9g %.9g\n", v.position.x, v.position.y, v.position.z, v.texcoord.x, v.texcoord.y,
           v.normal.x, v.normal.y, v.normal.z, v.tangent.x, v.tangent.y, v.tangent.z, v.tangent.w);

    resources.release<Ocean>(oceanmesh);
  }
  catch(std::exception const &e)
  {
    fprintf(stderr, "ocean_headless: %s\n", e.what());
    return 1;
  }

  return 0;
}
