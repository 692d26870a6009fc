// ocean_farm.hip -- the tile farm's one exchange step inside the C ABI: an RCCL all-gather of every rank's displacement
// field (north_star: "independent ocean cascades/tiles farm across the 8 GPUs of one node with a single RCCL
// all-gather over xGMI to reassemble the displacement field").  Nothing in the reference corresponds to it (one device).
//
// A C++ renderer that runs one process per GPU needs nothing but this module: rank 0 makes the 128-byte id
// (datum_ocean_farm_unique_id), hands it to the other ranks by whatever channel the host has, every rank calls
// datum_ocean_farm_init on its handle, and from then on
//
//     slot = farm_gather(ctx)                      pack (handle's stream, behind the last displace) + all-gather (the module's
//                                                  communication stream); returns at once: the next steps' kernels overlap it
//     farm_result(ctx, slot, stream, ...)          `stream` waits for that collective; pointer to the gathered block
//     farm_release(ctx, slot, stream)              `stream` is done reading: the slot's next collective waits for this point
//
// The payload and the gathered block are double-buffered (slots), every ordering is an event between streams, no call blocks
// the host except farm_wait.  RCCL is opened with dlopen at farm_init: a renderer that never farms has no dependency on it,
// and a process that already holds an RCCL (PyTorch's) shares that one instead of mapping a second copy.
//
// The choreography mirrors datum_amd/farm.py's TileGather, which stays as its CPU model (tests/test_farm_gloo.py: gloo,
// world sizes 2 and 8).

#pragma once

#include <dlfcn.h>

// The handful of RCCL declarations the module uses, spelled out (values and layouts as in rccl/rccl.h of ROCm 6 / 7, which are
// NCCL's): the module is built without RCCL's headers, so a renderer that never farms needs RCCL neither to run nor to BUILD.
extern "C"
{
  typedef struct ncclComm *ncclComm_t;
  typedef struct { char internal[128]; } ncclUniqueId;
  typedef enum { ncclSuccess = 0 } ncclResult_t;            // (the other values only ever travel as integers and through ncclGetErrorString)
  typedef enum { ncclInt8 = 0 } ncclDataType_t;             // the payload is gathered as bytes
}

namespace ocean
{
  struct RcclApi
  {
    void *lib = nullptr;
    std::string path;

    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;             // optional: the teardown of a communicator whose collective failed
    ncclResult_t (*AllGather)(void const*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    char const *(*GetErrorString)(ncclResult_t) = nullptr;
    char const *(*GetLastError)(ncclComm_t) = nullptr;           // optional: only adds RCCL's own text to an error
    ncclResult_t (*GetVersion)(int*) = nullptr;
  };

  // one per process; the first farm call opens it (a function-local static: initialised once, also with several handles on
  // several threads making their first farm call at the same time)
  struct RcclOpened
  {
    RcclApi api;
    std::string failure;

    RcclOpened()
    {
      char const *names[] = { getenv("DATUM_OCEAN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };

      for(char const *name : names)
      {
        if (!name || !*name)
          continue;

        // (RTLD_NOLOAD first: the copy the process already holds, e.g. PyTorch's, whatever directory it came from)
        void *lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);

        if (!lib)
          lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);

        if (lib)
        {
          api.lib = lib;
          api.path = name;
          break;
        }

        char const *err = dlerror();

        failure += std::string(failure.empty() ? "" : "; ") + (err ? err : name);
      }

      if (api.lib)
      {
        // required symbols: the first one missing ends the resolution (api.lib is what every later lookup would go through)
        struct { void **slot; char const *symbol; bool required; } const wanted[] = {
          { reinterpret_cast<void**>(&api.GetUniqueId), "ncclGetUniqueId", true },
          { reinterpret_cast<void**>(&api.CommInitRank), "ncclCommInitRank", true },
          { reinterpret_cast<void**>(&api.CommDestroy), "ncclCommDestroy", true },
          { reinterpret_cast<void**>(&api.AllGather), "ncclAllGather", true },
          { reinterpret_cast<void**>(&api.GetErrorString), "ncclGetErrorString", true },
          { reinterpret_cast<void**>(&api.GetVersion), "ncclGetVersion", true },
          { reinterpret_cast<void**>(&api.GetLastError), "ncclGetLastError", false },
          { reinterpret_cast<void**>(&api.CommAbort), "ncclCommAbort", false },
        };

        void *const lib = api.lib;

        for(auto const &w : wanted)
        {
          *w.slot = dlsym(lib, w.symbol);

          if (!*w.slot && w.required)
          {
            failure = std::string("RCCL library ") + api.path + " lacks " + w.symbol;
            api.lib = nullptr;
            break;
          }
        }
      }
    }
  };

  inline RcclApi *rccl_api(std::string *why)
  {
    static RcclOpened opened;

    if (!opened.api.lib)
    {
      if (why)
        *why = opened.failure;

      return nullptr;
    }

    return &opened.api;
  }

  struct FarmSlot
  {
    void *payload = nullptr;          // this rank's contribution (written by the pack on the handle's stream)
    void *gathered = nullptr;         // world x payload, ordered by rank (written by the collective on the communication stream)
    hipEvent_t packed = nullptr;      // behind the pack
    hipEvent_t start = nullptr;       // around the collective (timing)
    hipEvent_t done = nullptr;
    hipEvent_t consumed = nullptr;    // recorded by farm_release on the consumer's stream
    bool launched = false;            // a collective has been enqueued into this slot
    bool busy = false;                // a release is pending: the slot's next collective waits for `consumed`
    bool held = false;                // the result went to a stream other than the handle's and has not been released
  };

  struct Farm
  {
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;     // the communication stream
    int commcus = 0;                  // datum_ocean_farm_partition: the stream's compute units (0: all)
    int rank = 0, world = 1;
    int format = DATUM_OCEAN_PAYLOAD_XYZ32;
    size_t bytes = 0;                 // payload bytes per rank
    std::vector<FarmSlot> slots;
    int head = 0;                     // next slot
    unsigned long gathers = 0;
    bool failed = false;              // a collective on this communicator returned an error: torn down with ncclCommAbort (a destroy waits for its operations)
  };
}
