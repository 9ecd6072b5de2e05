// What hipGetLastError() reports after [invalid launch, valid launch] on this runtime: the stage-level launch check of libairfe.so (cfg.check_launches,
// airfe_host.h ProfScope) asks once per stage, so it matters whether a later successful call overwrites the stored error.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/last_error_semantics.hip -o tools/microbench/last_error_semantics && ./tools/microbench/last_error_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p) { if (p) p[threadIdx.x] = threadIdx.x; }
int main() {
  int* d = nullptr;
  (void)hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(4096), 0, 0, d);            // invalid: 4096 threads per workgroup
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);              // valid
  const hipError_t peek = hipPeekAtLastError();
  const hipError_t ext = hipExtGetLastError();
  const hipError_t last = hipGetLastError();
  const hipError_t again = hipGetLastError();
  printf("after [invalid launch, valid launch]: hipPeekAtLastError = %s, hipExtGetLastError = %s, hipGetLastError = %s, again = %s\n", hipGetErrorName(peek),
         hipGetErrorName(ext), hipGetErrorName(last), hipGetErrorName(again));
  hipLaunchKernelGGL(k, dim3(1), dim3(4096), 0, 0, d);
  printf("right after an invalid launch: hipGetLastError = %s; sync = %s\n", hipGetErrorName(hipGetLastError()), hipGetErrorName(hipDeviceSynchronize()));
  return 0;
}
