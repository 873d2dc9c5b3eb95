// Does hipExtAnyOrderLaunch let kernel B start before kernel A (same stream) has finished on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
__global__ void kA(uint64_t *t, int spin_us) {
  uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  while (wall_clock64() - t0 < (uint64_t)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void kB(uint64_t *t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64();
}
int main() {
  uint64_t *d, h[3];
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int flags = 0; flags < 2; ++flags)
    for (int grid : {64, 256, 512}) {
      hipMemsetAsync(d, 0, 64, s);
      hipStreamSynchronize(s);
      int spin = 50;
      void *argsA[] = {&d, &spin};
      void *argsB[] = {&d};
      hipError_t e1 = hipExtLaunchKernel((const void *)kA, dim3(grid), dim3(1024), argsA, 0, s, nullptr, nullptr, 0);
      hipError_t e2 = hipExtLaunchKernel((const void *)kB, dim3(64), dim3(1024), argsB, 0, s, nullptr, nullptr, flags);
      hipError_t e3 = hipStreamSynchronize(s);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("flags=%d gridA=%d err=%d,%d,%d  A: %.1f us long; B started %.1f us after A's start (%s A's end)\n", flags, grid,
             (int)e1, (int)e2, (int)e3, (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0,
             h[2] < h[1] ? "BEFORE" : "after");
    }
  return 0;
}
