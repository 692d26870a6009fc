// Test helper (tests only, not part of the product): stands in for the Vulkan side of the external-memory handshake.
// Allocates device memory through HIP's virtual-memory API with an exportable POSIX file descriptor -- on amdgpu the
// same dma-buf descriptor VK_KHR_external_memory_fd hands out for OPAQUE_FD -- so that datum_ocean_import_memory_fd
// can be exercised without a Vulkan loader: what the module writes through the imported pointer must be visible
// through the exporter's own mapping.
#include <hip/hip_runtime.h>
#include <unistd.h>

extern "C"
{
  struct extmem
  {
    hipMemGenericAllocationHandle_t handle;
    void *va;
    size_t bytes;
  };

  // returns 0 on success; *fd is a fresh descriptor for the allocation (the importer takes it over)
  int extmem_create(size_t bytes, extmem *out, int *fd)
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;

    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0)
      return 1;

    bytes = (bytes + gran - 1) / gran * gran;

    if (hipMemCreate(&out->handle, bytes, &prop, 0) != hipSuccess)
      return 2;
    if (hipMemAddressReserve(&out->va, bytes, 0, nullptr, 0) != hipSuccess)
      return 3;
    if (hipMemMap(out->va, bytes, 0, out->handle, 0) != hipSuccess)
      return 4;

    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    if (hipMemSetAccess(out->va, bytes, &acc, 1) != hipSuccess)
      return 5;
    if (hipMemExportToShareableHandle(fd, out->handle, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess)
      return 6;

    out->bytes = bytes;

    return (hipMemset(out->va, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess) ? 0 : 7;
  }

  int extmem_read(extmem const *m, void *host, size_t bytes)
  {
    return (hipDeviceSynchronize() == hipSuccess && hipMemcpy(host, m->va, bytes, hipMemcpyDeviceToHost) == hipSuccess) ? 0 : 1;
  }

  int extmem_destroy(extmem *m)
  {
    (void)hipMemUnmap(m->va, m->bytes);
    (void)hipMemAddressFree(m->va, m->bytes);
    return hipMemRelease(m->handle) == hipSuccess ? 0 : 1;
  }
}
