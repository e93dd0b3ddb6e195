// Development aid: does the HIP virtual-memory API of this ROCm map ONE physical chunk at many addresses of a reserved range?  (The sharded arena
// of a multi-GPU rank backs only its own panels and the shared top; the ranges of the other ranks' panels alias one small chunk.)
//   hipcc --offload-arch=gfx950 -O3 scripts/vmm_probe.hip -o scripts/vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(double *p, size_t n, double v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (double)i; }
int main()
{
  int dev = 0; CK(hipSetDevice(dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  printf("granularity recommended %zu minimum %zu\n", gran, gmin);
  const size_t total = 64 * gran, owned = 2 * gran;
  void *va = nullptr; CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
  hipMemGenericAllocationHandle_t h_own, h_dummy;
  CK(hipMemCreate(&h_own, owned, &prop, 0)); CK(hipMemCreate(&h_dummy, gran, &prop, 0));
  CK(hipMemMap(va, owned, 0, h_own, 0));
  for (size_t o = owned; o < total; o += gran) CK(hipMemMap((char *)va + o, gran, 0, h_dummy, 0));
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (double *)va, total / 8, 1.0);
  CK(hipDeviceSynchronize());
  double a[2], b[2];
  CK(hipMemcpy(a, va, 16, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b, (char *)va + owned, 16, hipMemcpyDeviceToHost)); // the aliased chunk: whatever granule wrote last
  printf("owned[0..1] = %.1f %.1f (expect 1 2); aliased[0] = %.1f (some granule's first element: = 1 + k * %zu for a k >= 2)\n", a[0], a[1], b[0], gran / 8);
  CK(hipMemset((char *)va + gran, 0, 3 * gran)); // a memset across the boundary owned | aliased
  CK(hipMemUnmap(va, total)); CK(hipMemRelease(h_own)); CK(hipMemRelease(h_dummy)); CK(hipMemAddressFree(va, total));
  printf("vmm ok\n");
  return 0;
}
