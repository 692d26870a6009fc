/* harness.c -- the reference's ocean displacement step (update_ocean on the host, then ocean.sim -> ocean.fftx ->
 * ocean.ffty -> ocean.map as Vulkan compute dispatches, src/renderer/ocean.cpp:217-236, :729-789) on whatever Vulkan
 * device the loader offers, CPU devices (lavapipe) preferred: north_star's CPU baseline.
 *
 * Builds only where <vulkan/vulkan.h>, libvulkan and glslangValidator exist (see Makefile); none of them exists in the
 * image this repository was written in, so THIS FILE HAS NEVER BEEN COMPILED OR RUN.  The shaders are this repository's
 * N-generalised restatements (ocean_*_n.comp), not the reference's files: the reference hard-wires N = 64.
 *
 *   make N=256 && python make_state.py 256 && VK_ICD_FILENAMES=/usr/share/vulkan/icd.d/lvp_icd.x86_64.json ./harness 256 200
 *
 * Input: state_<N>.bin = h0 (2 N^2 floats) followed by the twiddle table (N^2 floats, ocean.cpp:686-700), written by
 * make_state.py from the repository's host code (seed 1000, example-ocean parameters).  Output: grids/s and a checksum
 * of the height channel to hold against the oracle's.
 */
#include <vulkan/vulkan.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHECK(call) do { VkResult r_ = (call); if (r_ != VK_SUCCESS) { fprintf(stderr, "%s -> %d\n", #call, (int)r_); exit(1); } } while(0)

static VkPhysicalDevice physical;
static VkDevice device;

static uint32_t memory_type(uint32_t bits, VkMemoryPropertyFlags want)
{
  VkPhysicalDeviceMemoryProperties mp;
  vkGetPhysicalDeviceMemoryProperties(physical, &mp);
  for(uint32_t i = 0; i < mp.memoryTypeCount; ++i)
    if ((bits & (1u << i)) && (mp.memoryTypes[i].propertyFlags & want) == want)
      return i;
  fprintf(stderr, "no memory type\n");
  exit(1);
}

static VkBuffer make_buffer(VkDeviceSize size, VkBufferUsageFlags usage, VkMemoryPropertyFlags props, VkDeviceMemory *memory)
{
  VkBufferCreateInfo bi = { VK_STRUCTURE_TYPE_BUFFER_CREATE_INFO };
  bi.size = size;
  bi.usage = usage;
  VkBuffer buffer;
  CHECK(vkCreateBuffer(device, &bi, NULL, &buffer));
  VkMemoryRequirements req;
  vkGetBufferMemoryRequirements(device, buffer, &req);
  VkMemoryAllocateInfo ai = { VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO };
  ai.allocationSize = req.size;
  ai.memoryTypeIndex = memory_type(req.memoryTypeBits, props);
  CHECK(vkAllocateMemory(device, &ai, NULL, memory));
  CHECK(vkBindBufferMemory(device, buffer, *memory, 0));
  return buffer;
}

static VkPipeline make_pipeline(char const *path, VkPipelineLayout layout)
{
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "%s: not found (make N=...)\n", path); exit(1); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint32_t *code = malloc((size_t)n);
  if (fread(code, 1, (size_t)n, f) != (size_t)n) exit(1);
  fclose(f);
  VkShaderModuleCreateInfo mi = { VK_STRUCTURE_TYPE_SHADER_MODULE_CREATE_INFO };
  mi.codeSize = (size_t)n;
  mi.pCode = code;
  VkShaderModule module;
  CHECK(vkCreateShaderModule(device, &mi, NULL, &module));
  VkComputePipelineCreateInfo pi = { VK_STRUCTURE_TYPE_COMPUTE_PIPELINE_CREATE_INFO };
  pi.stage.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO;
  pi.stage.stage = VK_SHADER_STAGE_COMPUTE_BIT;
  pi.stage.module = module;
  pi.stage.pName = "main";
  pi.layout = layout;
  VkPipeline pipeline;
  CHECK(vkCreateComputePipelines(device, VK_NULL_HANDLE, 1, &pi, NULL, &pipeline));
  free(code);
  return pipeline;
}

static void barrier(VkCommandBuffer cb)
{
  /* the reference's barrier between dispatches (vulkan.cpp:1364-1377), with the compute -> compute masks it needs */
  VkMemoryBarrier mb = { VK_STRUCTURE_TYPE_MEMORY_BARRIER };
  mb.srcAccessMask = VK_ACCESS_SHADER_WRITE_BIT;
  mb.dstAccessMask = VK_ACCESS_SHADER_READ_BIT | VK_ACCESS_SHADER_WRITE_BIT;
  vkCmdPipelineBarrier(cb, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, 0, 1, &mb, 0, NULL, 0, NULL);
}

/* dispersion(k), src/renderer/ocean.cpp:82-87 */
static float dispersion(float kx, float ky)
{
  float k2 = kx * kx + ky * ky;
  return sqrtf(9.81f * sqrtf(k2) * (1 + k2 / (370 * 370)));
}

int main(int argc, char **argv)
{
  int const N = argc > 1 ? atoi(argv[1]) : 64;
  int const steps = argc > 2 ? atoi(argv[2]) : 200;
  float const wavescale = 22.0f, choppiness = 1.35f, dt = 1.0f / 60;   /* examples/ocean/ocean.cpp:46-50 */
  size_t const P = (size_t)N * N;

  /* -- instance, device (a CPU device if there is one), compute queue -- */
  VkApplicationInfo app = { VK_STRUCTURE_TYPE_APPLICATION_INFO };
  app.apiVersion = VK_API_VERSION_1_0;
  VkInstanceCreateInfo ii = { VK_STRUCTURE_TYPE_INSTANCE_CREATE_INFO };
  ii.pApplicationInfo = &app;
  VkInstance instance;
  CHECK(vkCreateInstance(&ii, NULL, &instance));

  uint32_t count = 0;
  CHECK(vkEnumeratePhysicalDevices(instance, &count, NULL));
  if (count == 0) { fprintf(stderr, "no Vulkan device\n"); return 1; }
  VkPhysicalDevice *all = malloc(count * sizeof(*all));
  CHECK(vkEnumeratePhysicalDevices(instance, &count, all));
  physical = all[0];
  for(uint32_t i = 0; i < count; ++i)
  {
    VkPhysicalDeviceProperties pp;
    vkGetPhysicalDeviceProperties(all[i], &pp);
    if (pp.deviceType == VK_PHYSICAL_DEVICE_TYPE_CPU) { physical = all[i]; break; }
  }
  VkPhysicalDeviceProperties props;
  vkGetPhysicalDeviceProperties(physical, &props);
  printf("device: %s (type %d), max workgroup invocations %u\n", props.deviceName, (int)props.deviceType, props.limits.maxComputeWorkGroupInvocations);
  if ((uint32_t)N > props.limits.maxComputeWorkGroupInvocations) { fprintf(stderr, "N exceeds the device's workgroup size (one invocation per point of a line)\n"); return 1; }

  uint32_t qcount = 0, family = 0;
  vkGetPhysicalDeviceQueueFamilyProperties(physical, &qcount, NULL);
  VkQueueFamilyProperties *qp = malloc(qcount * sizeof(*qp));
  vkGetPhysicalDeviceQueueFamilyProperties(physical, &qcount, qp);
  for(uint32_t i = 0; i < qcount; ++i)
    if (qp[i].queueFlags & VK_QUEUE_COMPUTE_BIT) { family = i; break; }

  float priority = 1.0f;
  VkDeviceQueueCreateInfo qi = { VK_STRUCTURE_TYPE_DEVICE_QUEUE_CREATE_INFO };
  qi.queueFamilyIndex = family;
  qi.queueCount = 1;
  qi.pQueuePriorities = &priority;
  VkDeviceCreateInfo di = { VK_STRUCTURE_TYPE_DEVICE_CREATE_INFO };
  di.queueCreateInfoCount = 1;
  di.pQueueCreateInfos = &qi;
  CHECK(vkCreateDevice(physical, &di, NULL, &device));
  VkQueue queue;
  vkGetDeviceQueue(device, family, 0, &queue);

  /* -- OceanSet (host visible, rewritten every frame as ocean.cpp:729-749 does), Spectrum, displacement image -- */
  VkDeviceSize const setsize = 216 + 12 * P, specsize = 28 * P;
  VkDeviceMemory setmem, specmem, imgmem, readmem;
  VkBuffer oceanset = make_buffer(setsize, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT, VK_MEMORY_PROPERTY_HOST_VISIBLE_BIT | VK_MEMORY_PROPERTY_HOST_COHERENT_BIT, &setmem);
  VkBuffer spectrum = make_buffer(specsize, VK_BUFFER_USAGE_STORAGE_BUFFER_BIT, VK_MEMORY_PROPERTY_HOST_VISIBLE_BIT | VK_MEMORY_PROPERTY_HOST_COHERENT_BIT, &specmem);
  VkBuffer readback = make_buffer(32 * P, VK_BUFFER_USAGE_TRANSFER_DST_BIT, VK_MEMORY_PROPERTY_HOST_VISIBLE_BIT | VK_MEMORY_PROPERTY_HOST_COHERENT_BIT, &readmem);

  VkImageCreateInfo imi = { VK_STRUCTURE_TYPE_IMAGE_CREATE_INFO };
  imi.imageType = VK_IMAGE_TYPE_2D;
  imi.format = VK_FORMAT_R32G32B32A32_SFLOAT;                /* ocean.cpp:706 */
  imi.extent.width = (uint32_t)N; imi.extent.height = (uint32_t)N; imi.extent.depth = 1;
  imi.mipLevels = 1;
  imi.arrayLayers = 2;
  imi.samples = VK_SAMPLE_COUNT_1_BIT;
  imi.tiling = VK_IMAGE_TILING_OPTIMAL;
  imi.usage = VK_IMAGE_USAGE_STORAGE_BIT | VK_IMAGE_USAGE_SAMPLED_BIT | VK_IMAGE_USAGE_TRANSFER_SRC_BIT;
  VkImage image;
  CHECK(vkCreateImage(device, &imi, NULL, &image));
  VkMemoryRequirements ireq;
  vkGetImageMemoryRequirements(device, image, &ireq);
  VkMemoryAllocateInfo iai = { VK_STRUCTURE_TYPE_MEMORY_ALLOCATE_INFO };
  iai.allocationSize = ireq.size;
  iai.memoryTypeIndex = memory_type(ireq.memoryTypeBits, 0);
  CHECK(vkAllocateMemory(device, &iai, NULL, &imgmem));
  CHECK(vkBindImageMemory(device, image, imgmem, 0));
  VkImageViewCreateInfo vi = { VK_STRUCTURE_TYPE_IMAGE_VIEW_CREATE_INFO };
  vi.image = image;
  vi.viewType = VK_IMAGE_VIEW_TYPE_2D_ARRAY;
  vi.format = imi.format;
  vi.subresourceRange.aspectMask = VK_IMAGE_ASPECT_COLOR_BIT;
  vi.subresourceRange.levelCount = 1;
  vi.subresourceRange.layerCount = 2;
  VkImageView view;
  CHECK(vkCreateImageView(device, &vi, NULL, &view));

  /* -- state: header, h0, twiddles -- */
  char name[64];
  snprintf(name, sizeof(name), "state_%d.bin", N);
  FILE *sf = fopen(name, "rb");
  if (!sf) { fprintf(stderr, "%s: not found (python make_state.py %d)\n", name, N); return 1; }
  unsigned char *setmap; float *specmap;
  CHECK(vkMapMemory(device, setmem, 0, setsize, 0, (void**)&setmap));
  CHECK(vkMapMemory(device, specmem, 0, specsize, 0, (void**)&specmap));
  memset(setmap, 0, 216);
  float scale = 1 / wavescale;
  uint32_t size = (uint32_t)N;
  memcpy(setmap + 200, &scale, 4);             /* OceanSet offsets: src/renderer/ocean.cpp:33-50 */
  memcpy(setmap + 204, &choppiness, 4);
  memcpy(setmap + 212, &size, 4);
  float *h0 = (float*)(setmap + 216), *phase = (float*)(setmap + 216 + 8 * P);
  if (fread(h0, 4, 2 * P, sf) != 2 * P) return 1;
  if (fread(specmap + 6 * P, 4, P, sf) != P) return 1;        /* Spectrum::weights behind h, hx, hy (ocean.cpp:61-68) */
  fclose(sf);
  memset(phase, 0, 4 * P);

  /* -- descriptors, pipelines -- */
  VkDescriptorSetLayoutBinding b[3] = { { 0, VK_DESCRIPTOR_TYPE_STORAGE_BUFFER, 1, VK_SHADER_STAGE_COMPUTE_BIT, NULL },
                                        { 1, VK_DESCRIPTOR_TYPE_STORAGE_BUFFER, 1, VK_SHADER_STAGE_COMPUTE_BIT, NULL },
                                        { 2, VK_DESCRIPTOR_TYPE_STORAGE_IMAGE, 1, VK_SHADER_STAGE_COMPUTE_BIT, NULL } };
  VkDescriptorSetLayoutCreateInfo li = { VK_STRUCTURE_TYPE_DESCRIPTOR_SET_LAYOUT_CREATE_INFO };
  li.bindingCount = 3;
  li.pBindings = b;
  VkDescriptorSetLayout setlayout;
  CHECK(vkCreateDescriptorSetLayout(device, &li, NULL, &setlayout));
  VkPipelineLayoutCreateInfo pli = { VK_STRUCTURE_TYPE_PIPELINE_LAYOUT_CREATE_INFO };
  pli.setLayoutCount = 1;
  pli.pSetLayouts = &setlayout;
  VkPipelineLayout layout;
  CHECK(vkCreatePipelineLayout(device, &pli, NULL, &layout));
  VkDescriptorPoolSize ps[2] = { { VK_DESCRIPTOR_TYPE_STORAGE_BUFFER, 2 }, { VK_DESCRIPTOR_TYPE_STORAGE_IMAGE, 1 } };
  VkDescriptorPoolCreateInfo dpi = { VK_STRUCTURE_TYPE_DESCRIPTOR_POOL_CREATE_INFO };
  dpi.maxSets = 1;
  dpi.poolSizeCount = 2;
  dpi.pPoolSizes = ps;
  VkDescriptorPool pool;
  CHECK(vkCreateDescriptorPool(device, &dpi, NULL, &pool));
  VkDescriptorSetAllocateInfo dai = { VK_STRUCTURE_TYPE_DESCRIPTOR_SET_ALLOCATE_INFO };
  dai.descriptorPool = pool;
  dai.descriptorSetCount = 1;
  dai.pSetLayouts = &setlayout;
  VkDescriptorSet set;
  CHECK(vkAllocateDescriptorSets(device, &dai, &set));
  VkDescriptorBufferInfo bi0 = { oceanset, 0, setsize }, bi1 = { spectrum, 0, specsize };
  VkDescriptorImageInfo ii2 = { VK_NULL_HANDLE, view, VK_IMAGE_LAYOUT_GENERAL };
  VkWriteDescriptorSet w[3] = { { VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET }, { VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET }, { VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET } };
  for(int i = 0; i < 3; ++i) { w[i].dstSet = set; w[i].dstBinding = (uint32_t)i; w[i].descriptorCount = 1; }
  w[0].descriptorType = VK_DESCRIPTOR_TYPE_STORAGE_BUFFER; w[0].pBufferInfo = &bi0;
  w[1].descriptorType = VK_DESCRIPTOR_TYPE_STORAGE_BUFFER; w[1].pBufferInfo = &bi1;
  w[2].descriptorType = VK_DESCRIPTOR_TYPE_STORAGE_IMAGE; w[2].pImageInfo = &ii2;
  vkUpdateDescriptorSets(device, 3, w, 0, NULL);

  char path[4][64];
  char const *stage[4] = { "sim", "fftx", "ffty", "map" };
  VkPipeline pipe[4];
  for(int i = 0; i < 4; ++i) { snprintf(path[i], sizeof(path[i]), "ocean_%s_%d.spv", stage[i], N); pipe[i] = make_pipeline(path[i], layout); }

  /* -- one command buffer: sim -> fftx -> ffty -> map (ocean.cpp:769-789) -- */
  VkCommandPoolCreateInfo cpi = { VK_STRUCTURE_TYPE_COMMAND_POOL_CREATE_INFO };
  cpi.queueFamilyIndex = family;
  VkCommandPool cpool;
  CHECK(vkCreateCommandPool(device, &cpi, NULL, &cpool));
  VkCommandBufferAllocateInfo cai = { VK_STRUCTURE_TYPE_COMMAND_BUFFER_ALLOCATE_INFO };
  cai.commandPool = cpool;
  cai.level = VK_COMMAND_BUFFER_LEVEL_PRIMARY;
  cai.commandBufferCount = 2;
  VkCommandBuffer cbs[2];
  CHECK(vkAllocateCommandBuffers(device, &cai, cbs));
  VkCommandBuffer cb = cbs[0], copycb = cbs[1];
  VkCommandBufferBeginInfo begin = { VK_STRUCTURE_TYPE_COMMAND_BUFFER_BEGIN_INFO };

  VkImageMemoryBarrier tolayout = { VK_STRUCTURE_TYPE_IMAGE_MEMORY_BARRIER };
  tolayout.oldLayout = VK_IMAGE_LAYOUT_UNDEFINED;
  tolayout.newLayout = VK_IMAGE_LAYOUT_GENERAL;
  tolayout.srcQueueFamilyIndex = tolayout.dstQueueFamilyIndex = VK_QUEUE_FAMILY_IGNORED;
  tolayout.image = image;
  tolayout.subresourceRange = vi.subresourceRange;
  tolayout.dstAccessMask = VK_ACCESS_SHADER_WRITE_BIT;

  CHECK(vkBeginCommandBuffer(cb, &begin));
  vkCmdPipelineBarrier(cb, VK_PIPELINE_STAGE_TOP_OF_PIPE_BIT, VK_PIPELINE_STAGE_COMPUTE_SHADER_BIT, 0, 0, NULL, 0, NULL, 1, &tolayout);
  vkCmdBindDescriptorSets(cb, VK_PIPELINE_BIND_POINT_COMPUTE, layout, 0, 1, &set, 0, NULL);
  vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_COMPUTE, pipe[0]);
  vkCmdDispatch(cb, (uint32_t)N / 16, (uint32_t)N / 16, 1);
  barrier(cb);
  vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_COMPUTE, pipe[1]);
  vkCmdDispatch(cb, 1, (uint32_t)N, 1);                       /* one workgroup per row */
  barrier(cb);
  vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_COMPUTE, pipe[2]);
  vkCmdDispatch(cb, (uint32_t)N, 1, 1);                       /* one workgroup per column */
  barrier(cb);
  vkCmdBindPipeline(cb, VK_PIPELINE_BIND_POINT_COMPUTE, pipe[3]);
  vkCmdDispatch(cb, (uint32_t)N / 16, (uint32_t)N / 16, 1);
  CHECK(vkEndCommandBuffer(cb));

  VkFenceCreateInfo fi = { VK_STRUCTURE_TYPE_FENCE_CREATE_INFO };
  VkFence fence;
  CHECK(vkCreateFence(device, &fi, NULL, &fence));
  VkSubmitInfo submit = { VK_STRUCTURE_TYPE_SUBMIT_INFO };
  submit.commandBufferCount = 1;
  submit.pCommandBuffers = &cb;

  /* -- steps: update_ocean's phase loop on the host (ocean.cpp:223-233), then the four dispatches -- */
  float const twopi = 2 * 3.14159265358979323846f, dk = twopi / wavescale;
  struct timespec t0, t1;
  for(int step = -10; step < steps; ++step)
  {
    if (step == 0)
      clock_gettime(CLOCK_MONOTONIC, &t0);
    for(int m = 0; m < N; ++m)
      for(int n = 0; n < N; ++n)
        phase[(size_t)m * N + n] = fmodf(phase[(size_t)m * N + n] + dispersion(dk * (n - 0.5f * N), dk * (m - 0.5f * N)) * dt, twopi);
    CHECK(vkQueueSubmit(queue, 1, &submit, fence));
    CHECK(vkWaitForFences(device, 1, &fence, VK_TRUE, UINT64_MAX));
    CHECK(vkResetFences(device, 1, &fence));
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

  /* -- read the maps back once: a checksum to hold against the oracle's -- */
  VkBufferImageCopy region = { 0 };
  region.imageSubresource.aspectMask = VK_IMAGE_ASPECT_COLOR_BIT;
  region.imageSubresource.layerCount = 2;
  region.imageExtent = imi.extent;
  CHECK(vkBeginCommandBuffer(copycb, &begin));
  vkCmdCopyImageToBuffer(copycb, image, VK_IMAGE_LAYOUT_GENERAL, readback, 1, &region);
  CHECK(vkEndCommandBuffer(copycb));
  submit.pCommandBuffers = &copycb;
  CHECK(vkQueueSubmit(queue, 1, &submit, fence));
  CHECK(vkWaitForFences(device, 1, &fence, VK_TRUE, UINT64_MAX));
  float *maps;
  CHECK(vkMapMemory(device, readmem, 0, 32 * P, 0, (void**)&maps));
  double sum = 0;
  for(size_t i = 0; i < P; ++i)
    sum += fabs((double)maps[4 * i + 2]);

  printf("%d x %d, %d steps after 10 warm-up: %.1f grids/s (%.3f ms per step); sum |dz| after %d steps of dt = 1/60: %.9g\n", N, N, steps, steps / seconds, 1e3 * seconds / steps, steps + 10, sum);

  return 0;
}
