//
// ocean_farm.cpp -- the tile farm from C++ alone: N processes, one GPU each, independent 2048 x 2048 tiles (BASELINE.json
// configs[3]) seeded mt19937(1000 + tile), stepped with update_ocean + the displacement pass, and reassembled with ONE RCCL
// all-gather per batch through the C ABI (datum_ocean_farm_*).  No Python, no PyTorch, no MPI: the parent starts the ranks
// as child processes BEFORE anything touches a GPU and relays rank 0's 128-byte communicator id over pipes.
//
// Usage: ocean_farm [ranks=1] [resolution=2048] [batches=3] [steps_per_batch=20] [payload: 1 xyz32 | 2 xyz16 | 0 maps] [comm_cus=auto]
// comm_cus: compute units the collective gets to itself (datum_ocean_farm_partition; 0 = both streams on the whole device; default: the
// module's own share, an eighth of the device, skipped with a warning where it cannot be had)
// Every rank prints one line per batch: a checksum of every tile of the gathered field (all ranks must print the same), and
// whether its own tile in the gathered block equals what it packed.  Exit code 0 only if every rank succeeded.
//

#include "../datum_amd/host/ocean.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/wait.h>
#include <signal.h>
#include <unistd.h>

namespace
{
  bool read_all(int fd, void *dst, size_t bytes)
  {
    char *p = static_cast<char*>(dst);

    while (bytes)
    {
      ssize_t n = read(fd, p, bytes);
      if (n <= 0)
        return false;
      p += n;
      bytes -= (size_t)n;
    }

    return true;
  }

  #define CHECK(call) do { int rc_ = (call); if (rc_ != DATUM_OCEAN_OK) { fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, rc_, datum_ocean_last_error(hip)); return 1; } } while(0)

  int run_rank(int rank, int world, int idin, int idout, int N, int batches, int steps, int format, int commcus)
  {
    datum_ocean_t hip = nullptr;

    // the id: rank 0 makes it and sends it to the parent, which relays it to the other ranks
    unsigned char id[DATUM_OCEAN_FARM_ID_BYTES];

    if (rank == 0)
    {
      CHECK(datum_ocean_farm_unique_id(id, sizeof(id)));

      if (write(idout, id, sizeof(id)) != (ssize_t)sizeof(id))
        return 1;
    }
    else if (!read_all(idin, id, sizeof(id)))
      return 1;

    // device = rank: one GPU per process (DATUM_FARM_DEVICES = d spreads the ranks over d devices instead -- RCCL refuses two ranks
    // on one GPU, which is how the error path of datum_ocean_farm_init is exercised on a one-GPU box)
    int const devices = getenv("DATUM_FARM_DEVICES") ? atoi(getenv("DATUM_FARM_DEVICES")) : world;

    CHECK(datum_ocean_create(&hip, rank % (devices > 0 ? devices : 1), N, 1));

    // this rank's tile: example-ocean parameters, seed 1000 + global tile index
    {
      OceanParams tile(N);
      tile.wavescale = 22.0f;
      tile.waveamplitude = 0.0025f;
      tile.swellamplitude = 0.8f;
      tile.windspeed = 7.9f;
      tile.smoothing = 320.0f;

      seed_ocean(tile, 1000u + (unsigned)rank);

      CHECK(datum_ocean_set_cascade(hip, 0, tile.wavescale, tile.choppiness));
      CHECK(datum_ocean_upload_state(hip, 0, tile.height.data(), nullptr));
    }

    CHECK(datum_ocean_farm_init(hip, id, sizeof(id), rank, world, format, 2));

    // the collective's copying workgroups on compute units of their own, the step's kernels on the others
    // (the default asks for the module's own share and goes on unpartitioned where the device or the runtime has none to give -- a part with
    // fewer than 64 compute units, a runtime without CU masks --; a share the user asked for by number must be granted)
    if (commcus == DATUM_OCEAN_FARM_PARTITION_AUTO)
    {
      if (datum_ocean_farm_partition(hip, commcus) != DATUM_OCEAN_OK)
        fprintf(stderr, "rank %d: no compute-unit partition (%s): both streams on the whole device\n", rank, datum_ocean_last_error(hip));
    }
    else if (commcus > 0)
      CHECK(datum_ocean_farm_partition(hip, commcus));

    size_t bytes = 0;
    CHECK(datum_ocean_payload_bytes(hip, format, &bytes));

    std::vector<unsigned char> gathered(bytes * world), mine(bytes);
    void *payload = nullptr;
    CHECK(datum_ocean_device_alloc(hip, bytes, &payload));

    int previous = -1;

    for(int batch = 0; batch <= batches; ++batch)
    {
      int slot = -1;

      if (batch < batches)
      {
        for(int i = 0; i < steps; ++i)
        {
          CHECK(datum_ocean_update(hip, 1.0f/60));
          CHECK(datum_ocean_displace(hip));
        }

        // what this rank contributes, kept aside for the check below (the farm packs it again into its own slot)
        CHECK(datum_ocean_pack_displacement(hip, format, payload, bytes));
        CHECK(datum_ocean_device_read(hip, mine.data(), payload, bytes));

        CHECK(datum_ocean_farm_gather(hip, &slot));      // returns at once; the next batch's kernels overlap the collective
      }

      if (previous >= 0)
      {
        void *device = nullptr;
        size_t total = 0;
        float ms = 0;

        CHECK(datum_ocean_farm_result(hip, previous, nullptr, 1, &device, &total));
        CHECK(datum_ocean_device_read(hip, gathered.data(), device, total));
        CHECK(datum_ocean_farm_release(hip, previous, nullptr, 1));
        CHECK(datum_ocean_farm_wait(hip, previous, &ms));

        // (what the rank RECEIVED over the collective's time: the bus bandwidth DESIGN.md section 7's prediction assumes to be >= 270 GB/s at 8 ranks)
        printf("rank %d batch %d gather %.3f ms (%.0f GB/s received) tiles", rank, batch - 1, ms, ms > 0 ? (double)(world - 1) * bytes / (ms * 1e-3) / 1e9 : 0.0);

        for(int r = 0; r < world; ++r)
        {
          unsigned long long h = 1469598103934665603ull;      // FNV-1a over the tile's bytes
          for(size_t i = 0; i < bytes; ++i)
            h = (h ^ gathered[(size_t)r * bytes + i]) * 1099511628211ull;
          printf(" %016llx", h);
        }

        // (`mine` is already the NEXT batch's payload except after the last gather)
        if (batch == batches)
          printf(" own-tile %s", memcmp(gathered.data() + (size_t)rank * bytes, mine.data(), bytes) == 0 ? "ok" : "MISMATCH");

        printf("\n");
        fflush(stdout);

        if (batch == batches && memcmp(gathered.data() + (size_t)rank * bytes, mine.data(), bytes) != 0)
          return 1;
      }

      previous = slot;
    }

    CHECK(datum_ocean_device_free(hip, payload));
    CHECK(datum_ocean_farm_shutdown(hip));
    CHECK(datum_ocean_destroy(hip));

    return 0;
  }
}

int main(int argc, char **argv)
{
  int world = (argc > 1) ? atoi(argv[1]) : 1;
  int N = (argc > 2) ? atoi(argv[2]) : 2048;
  int batches = (argc > 3) ? atoi(argv[3]) : 3;
  int steps = (argc > 4) ? atoi(argv[4]) : 20;
  int format = (argc > 5) ? atoi(argv[5]) : DATUM_OCEAN_PAYLOAD_XYZ32;
  int commcus = (argc > 6) ? atoi(argv[6]) : DATUM_OCEAN_FARM_PARTITION_AUTO;

  if (world < 1 || world > 64)
    return 2;

  // the module carries no soname: refuse a libdatum_ocean_hip.so of another revision of the header (touches no GPU)
  if (datum_ocean_abi_version() != DATUM_OCEAN_ABI_VERSION)
  {
    fprintf(stderr, "libdatum_ocean_hip.so reports ABI version %d, built against %d\n", datum_ocean_abi_version(), DATUM_OCEAN_ABI_VERSION);
    return 2;
  }

  // nothing below touches a GPU in this process: the ranks are children, started before any HIP call
  int fromzero[2];
  std::vector<int> tochild(2 * world, -1);
  std::vector<pid_t> pids(world);

  if (pipe(fromzero) != 0)
    return 2;

  for(int r = 1; r < world; ++r)
    if (pipe(&tochild[2 * r]) != 0)
      return 2;

  for(int r = 0; r < world; ++r)
  {
    pids[r] = fork();

    if (pids[r] == 0)
    {
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);     // dmabuf IPC between the ranks' GPUs (RCCL)

      // keep only this rank's end of its pipe: a rank that never gets its id must see end-of-file, not wait for ever
      close(fromzero[0]);
      if (r != 0)
        close(fromzero[1]);
      for(int k = 1; k < world; ++k)
      {
        close(tochild[2 * k + 1]);
        if (k != r)
          close(tochild[2 * k]);
      }

      _exit(run_rank(r, world, r ? tochild[2 * r] : -1, r ? -1 : fromzero[1], N, batches, steps, format, commcus));
    }
  }

  close(fromzero[1]);

  unsigned char id[DATUM_OCEAN_FARM_ID_BYTES];
  bool relayed = read_all(fromzero[0], id, sizeof(id));

  for(int r = 1; r < world && relayed; ++r)
    relayed = write(tochild[2 * r + 1], id, sizeof(id)) == (ssize_t)sizeof(id);

  for(int r = 1; r < world; ++r)
  {
    close(tochild[2 * r]);
    close(tochild[2 * r + 1]);
  }

  int failed = relayed ? 0 : 1;

  // datum_ocean_farm_init is a collective with no timeout (ncclCommInitRank): a rank that died before it -- no device, no RCCL, a
  // failed upload -- leaves the others waiting for ever.  So the ranks are reaped in the order they END, and the first one that
  // ends badly (or an id that never made it round) takes the rest down instead of leaving the parent in waitpid.
  int left = world;
  bool culled = false;

  while (left > 0)
  {
    if (failed && !culled)
    {
      for(int r = 0; r < world; ++r)
        if (pids[r] > 0)
          kill(pids[r], SIGTERM);

      culled = true;
    }

    int status = 0;
    pid_t const who = waitpid(-1, &status, 0);

    if (who < 0)
      break;

    for(int r = 0; r < world; ++r)
    {
      if (pids[r] == who)
      {
        pids[r] = -1;
        left -= 1;

        // (a rank ended by the cull above is a casualty, not a second failure -- but the run has failed either way)
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
          failed += 1;
      }
    }
  }

  printf("ocean_farm: %d ranks, %d x %d tiles, %d batches of %d steps: %s\n", world, N, N, batches, steps, failed ? "FAILED" : "ok");

  return failed ? 1 : 0;
}
