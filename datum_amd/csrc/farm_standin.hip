// farm_standin.hip -- MEASUREMENT AID for bench.py --standin-peers on ONE GPU; not part of the ocean path and not in
// include/datum_ocean_hip.h.  What an RCCL all-gather does to the chip that runs the displacement step while it is in
// flight, without a second GPU: a kernel of a few dozen workgroups (RCCL's channels are workgroups that copy) that stays
// resident for as long as the collective would, reads this rank's payload once per peer and writes the peers' payloads into
// the gathered buffer (the HBM writes of the tiles arriving over xGMI), paced to a stated bus bandwidth with the
// device-wide clock.  A device-to-device hipMemcpy (round 2's stand-in) occupies no compute unit and runs at HBM speed.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace
{
  constexpr int THREADS = 256;
  constexpr size_t CHUNK = (size_t)THREADS * 16 * 8;       // bytes a workgroup moves between two looks at the clock

  // s_memrealtime: 100 MHz, one clock for the whole device
  __device__ __forceinline__ unsigned long long realtime()
  {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
  }

  // dst[peer][...] = src[...] for `peers` peers; workgroup g takes chunks g, g + gridDim.x, ...; after each chunk it waits
  // until the clock has reached what `ticks_per_chunk` allows for the chunks it has moved
  // mode 0: copy; 1: resident only (no memory traffic); 2: reads only; 3: writes only (what arrives over xGMI costs no reads here)
  // mode 4 / 5 / 6: as 0 / 2 / 3 with NON-TEMPORAL loads and stores (the `nt` bit: streaming data that should not displace what
  // the caches hold) -- if the step's slowdown under the stand-in is Infinity-Cache eviction it shrinks with these, if it is
  // plain bandwidth contention it does not (tools/gather_overhead.sh)
  // mode 7 / 8 (round 5): as 0 / 3 with the destination WRAPPED into one peer's share of the gathered buffer -- the same bytes read and
  // written, a seventh of the footprint: what of the slowdown is the 352 MB gathered buffer pushing the step's working set out of the
  // 256 MiB Infinity Cache, and what is the bytes themselves
  typedef unsigned int u4_ __attribute__((ext_vector_type(4)));

  __global__ void __launch_bounds__(THREADS) standin_kernel(uint4 *dst, uint4 const *src, size_t bytes, int peers, float ticks_per_chunk, int mode)
  {
    // mode 9: as 0 with source AND destination wrapped into 2 MB -- the same bytes through the compute units' memory paths, nothing
    // displaced in the 256 MiB Infinity Cache: what of the slowdown is the cache, what the paths (profiles/r05_gather_overhead.txt)
    bool const tiny = mode == 9;
    bool const wrap = mode >= 7;

    if (wrap)
      mode = (mode == 8) ? 3 : 0;

    bool const nt = mode >= 4;

    if (nt)
      mode = (mode == 4) ? 0 : (mode == 5) ? 2 : 3;

    size_t const chunks = (bytes + CHUNK - 1) / CHUNK;
    size_t const total = chunks * peers;
    unsigned long long const t0 = realtime();
    size_t done = 0;

    for(size_t item = blockIdx.x; item < total; item += gridDim.x)
    {
      size_t const peer = wrap ? 0 : item / chunks, chunk = tiny ? (item % chunks) % 64 : item % chunks;
      size_t const first = chunk * (CHUNK / 16), last = min((chunk + 1) * (CHUNK / 16), bytes / 16);

      if (last - first == CHUNK / 16 && mode != 1)
      {
        uint4 v[8];                            // eight loads in flight per lane, then eight stores

        if (mode == 3)
        {
          #pragma unroll
          for(int k = 0; k < 8; ++k)
            v[k] = make_uint4(k, peer, chunk, 0);
        }
        else if (nt)
        {
          // (inline asm, one statement with its own wait: as a builtin the non-temporal load is merged with the plain one of
          // the other branch)
          u4_ t[8];
          uint4 const *at = src + first + threadIdx.x;

          asm volatile("global_load_dwordx4 %0, %8, off offset:-4096 nt\n\t"
                       "global_load_dwordx4 %1, %8, off nt\n\t"
                       "global_load_dwordx4 %2, %9, off offset:-4096 nt\n\t"
                       "global_load_dwordx4 %3, %9, off nt\n\t"
                       "global_load_dwordx4 %4, %10, off offset:-4096 nt\n\t"
                       "global_load_dwordx4 %5, %10, off nt\n\t"
                       "global_load_dwordx4 %6, %11, off offset:-4096 nt\n\t"
                       "global_load_dwordx4 %7, %11, off nt\n\t"
                       "s_waitcnt vmcnt(0)"
                       : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                       : "v"(at + THREADS), "v"(at + 3 * THREADS), "v"(at + 5 * THREADS), "v"(at + 7 * THREADS) : "memory");

          static_assert(THREADS * 16 == 4096, "offset:-4096 is one k step");

          #pragma unroll
          for(int k = 0; k < 8; ++k)
            v[k] = __builtin_bit_cast(uint4, t[k]);
        }
        else
        {
          #pragma unroll
          for(int k = 0; k < 8; ++k)
            v[k] = src[first + threadIdx.x + k * THREADS];
        }

        if (mode == 2)
        {
          unsigned acc = 0;

          #pragma unroll
          for(int k = 0; k < 8; ++k)
            acc |= v[k].x & v[k].y & v[k].z & v[k].w;

          if (acc == 0x12345678u)              // never: keeps the loads
            dst[threadIdx.x] = v[0];
        }
        else
        {
          if (nt)
          {
            #pragma unroll
            for(int k = 0; k < 8; ++k)
              __builtin_nontemporal_store(__builtin_bit_cast(u4_, v[k]), reinterpret_cast<u4_*>(dst) + peer * (bytes / 16) + first + threadIdx.x + k * THREADS);
          }
          else
          {
            #pragma unroll
            for(int k = 0; k < 8; ++k)
              dst[peer * (bytes / 16) + first + threadIdx.x + k * THREADS] = v[k];
          }
        }
      }
      else if (mode == 1)
      {
      }
      else
      {
        for(size_t i = first + threadIdx.x; i < last; i += THREADS)
          dst[peer * (bytes / 16) + i] = src[i];
      }

      done += 1;

      if (ticks_per_chunk > 0)
      {
        unsigned long long const until = t0 + (unsigned long long)(ticks_per_chunk * (float)done);

        while (realtime() < until)
          __builtin_amdgcn_s_sleep(8);
      }
    }
  }
}

// where the workgroups of a launch on `stream` ran: out[2 g] = XCC_ID, out[2 g + 1] = HW_ID of workgroup g (which compute units a
// CU-masked stream really covers: tools/cumask_probe.py)
namespace
{
  __global__ void __launch_bounds__(THREADS) where_kernel(unsigned *out)
  {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));

    // long enough that the launch spreads over every compute unit it may use
    unsigned long long const t0 = realtime();
    while (realtime() < t0 + 2000)
      __builtin_amdgcn_s_sleep(8);

    if (threadIdx.x == 0)
    {
      out[2 * blockIdx.x] = xcc;
      out[2 * blockIdx.x + 1] = hw;
    }
  }
}

extern "C" int datum_farm_standin_where(unsigned *out, int workgroups, void *stream)
{
  hipLaunchKernelGGL(where_kernel, dim3(workgroups), dim3(THREADS), 0, (hipStream_t)stream, out);

  return (int)hipGetLastError();
}

// bytes: a multiple of 16.  gbps: the bus bandwidth the copy is paced to (bytes * peers / duration), 0 = as fast as it goes.
extern "C" int datum_farm_standin_gather_mode(void *gathered, void const *payload, size_t bytes, int peers, int workgroups, double gbps, int mode, void *stream);

extern "C" int datum_farm_standin_gather(void *gathered, void const *payload, size_t bytes, int peers, int workgroups, double gbps, void *stream)
{
  return datum_farm_standin_gather_mode(gathered, payload, bytes, peers, workgroups, gbps, 0, stream);
}

extern "C" int datum_farm_standin_gather_mode(void *gathered, void const *payload, size_t bytes, int peers, int workgroups, double gbps, int mode, void *stream)
{
  if (!gathered || !payload || (bytes & 15) || peers < 1 || workgroups < 1)
    return -1;

  size_t const chunks = (bytes + CHUNK - 1) / CHUNK;

  // a workgroup moves total / workgroups chunks in the time the whole transfer may take
  double const seconds = gbps > 0 ? (double)bytes * peers / (gbps * 1e9) : 0.0;
  double const mine = (double)(chunks * peers) / workgroups;
  float const ticks = gbps > 0 ? (float)(seconds * 1e8 / mine) : 0.0f;

  hipLaunchKernelGGL(standin_kernel, dim3(workgroups), dim3(THREADS), 0, (hipStream_t)stream, (uint4*)gathered, (uint4 const*)payload, bytes, peers, ticks, mode);

  return (int)hipGetLastError();
}
