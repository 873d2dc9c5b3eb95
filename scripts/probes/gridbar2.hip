// Hierarchical grid barrier: per-XCD arrival counters (workgroup id % 8 shares an XCD), the last
// arriver of an XCD bumps a global counter, everyone polls the global one.  Bounded spin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024) void kbar(unsigned *ctr, int nbar, uint64_t *t, int *fail) {
  const unsigned nwg = gridDim.x, x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;  // workgroups on this XCD slot
  unsigned *cx = ctr + 32 * (1 + x), *cg = ctr;
  uint64_t t0 = 0;
  for (int b = 0; b < nbar; ++b) {
    if (b == 1 && threadIdx.x == 0 && blockIdx.x == 0) t0 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned tk = __hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == (unsigned)(b + 1) * per - 1) __hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(b + 1) * (nwg < 8 ? nwg : 8);
      uint64_t s = wall_clock64();
      while (__hip_atomic_load(cg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
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
  hipMalloc(&ctr, 4096); hipMalloc(&t, 16); hipMalloc(&fail, 4);
  for (int grid : {64, 128, 256}) for (int nbar : {101, 1001}) {
    hipMemset(ctr, 0, 4096); hipMemset(fail, 0, 4);
    void *args[] = {&ctr, &nbar, &t, &fail};
    hipError_t e = hipLaunchCooperativeKernel((const void *)kbar, dim3(grid), dim3(1024), args, 0, 0);
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("hier grid=%d barriers=%d err=%d,%d timeout=%d: %.2f us per barrier\n", grid, nbar - 1, (int)e, (int)e2, hf,
           (h[1] - h[0]) / 100.0 / (nbar - 1));
  }
  return 0;
}
