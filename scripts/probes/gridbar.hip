// How long does a grid-wide barrier take on gfx950 (256 workgroups x 1024 threads, one per CU)?
// Bounded spin: a workgroup that waits longer than ~2 ms gives up (no hang).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024) void kbar(unsigned *ctr, int nbar, uint64_t *t, int *fail) {
  const unsigned nwg = gridDim.x;
  uint64_t t0 = 0;
  for (int b = 0; b < nbar; ++b) {
    if (b == 1 && threadIdx.x == 0 && blockIdx.x == 0) t0 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
      // monotonically increasing counter: barrier b is passed when ctr >= (b+1)*nwg
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(b + 1) * nwg;
      uint64_t s = wall_clock64();
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - s > 200000) { *fail = 1; break; }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
int main() {
  unsigned *ctr; uint64_t *t, h[2]; int *fail, hf;
  hipMalloc(&ctr, 4); hipMalloc(&t, 16); hipMalloc(&fail, 4);
  for (int grid : {64, 128, 256}) for (int nbar : {101, 1001}) {
    hipMemset(ctr, 0, 4); hipMemset(fail, 0, 4);
    void *args[] = {&ctr, &nbar, &t, &fail};
    hipError_t e = hipLaunchCooperativeKernel((const void *)kbar, dim3(grid), dim3(1024), args, 0, 0);
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("grid=%d barriers=%d err=%d,%d timeout=%d: %.2f us per barrier\n", grid, nbar - 1, (int)e, (int)e2, hf,
           (h[1] - h[0]) / 100.0 / (nbar - 1));
  }
  return 0;
}
