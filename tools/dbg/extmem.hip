// Probe: can a POSIX-fd handle exported by HIP's virtual-memory API be imported back with hipImportExternalMemory (opaque fd),
// the call a Vulkan-exported VkDeviceMemory would go through?  And what do bad descriptors return?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#define SHOW(x) do { hipError_t e=(x); printf("%-70s -> %d %s\n", #x, (int)e, hipGetErrorString(e)); (void)hipGetLastError(); } while(0)
int main() {
  size_t bytes = 1 << 22;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  size_t gran = 0; SHOW(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)); printf("granularity %zu\n", gran);
  hipMemGenericAllocationHandle_t h; SHOW(hipMemCreate(&h, bytes, &prop, 0));
  void *va = nullptr; SHOW(hipMemAddressReserve(&va, bytes, 0, nullptr, 0)); SHOW(hipMemMap(va, bytes, 0, h, 0));
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite; SHOW(hipMemSetAccess(va, bytes, &acc, 1));
  int fd = -1; SHOW(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0)); printf("fd %d\n", fd);
  hipExternalMemoryHandleDesc d = {}; d.type = hipExternalMemoryHandleTypeOpaqueFd; d.handle.fd = fd; d.size = bytes;
  hipExternalMemory_t em = nullptr; SHOW(hipImportExternalMemory(&em, &d));
  if (em) {
    hipExternalMemoryBufferDesc b = {}; b.offset = 0; b.size = bytes; void *p = nullptr; SHOW(hipExternalMemoryGetMappedBuffer(&p, em, &b)); printf("mapped %p (original va %p)\n", p, va);
    if (p) { SHOW(hipMemset(p, 0x5a, 4096)); unsigned char host[16]; SHOW(hipMemcpy(host, va, 16, hipMemcpyDeviceToHost)); printf("read through the original mapping: %02x %02x\n", host[0], host[15]); }
    SHOW(hipDestroyExternalMemory(em));
  }
  // bad descriptors
  hipExternalMemoryHandleDesc bad = {}; bad.type = hipExternalMemoryHandleTypeOpaqueFd; bad.handle.fd = -1; bad.size = bytes; em = nullptr; SHOW(hipImportExternalMemory(&em, &bad));
  int pfd[2]; if (pipe(pfd) == 0) { bad.handle.fd = pfd[0]; em = nullptr; SHOW(hipImportExternalMemory(&em, &bad)); close(pfd[0]); close(pfd[1]); }
  hipExternalSemaphoreHandleDesc sd = {}; sd.type = hipExternalSemaphoreHandleTypeOpaqueFd; sd.handle.fd = -1; hipExternalSemaphore_t es = nullptr; SHOW(hipImportExternalSemaphore(&es, &sd));
  return 0;
}
