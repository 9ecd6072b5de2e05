// How fast can a KERNEL write into pinned host memory, by allocation flag and grid size — against hipMemcpyAsync (SDMA) of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/pcie_kernel_write tools/microbench/pcie_kernel_write.hip && tools/microbench/pcie_kernel_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int main() {
  const size_t bytes = 64u << 20, n = bytes / 16;
  uint4* dsrc; CK(hipMalloc(&dsrc, bytes)); CK(hipMemset(dsrc, 1, bytes));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct F { const char* name; unsigned flags; } flags[] = {{"default", hipHostMallocDefault}, {"mapped|coherent", hipHostMallocMapped | hipHostMallocCoherent},
                                                            {"mapped|noncoherent", hipHostMallocMapped | hipHostMallocNonCoherent}};
  for (const F& f : flags) {
    uint4* h; CK(hipHostMalloc(reinterpret_cast<void**>(&h), bytes, f.flags)); memset(h, 0, bytes);
    void* dp; CK(hipHostGetDevicePointer(&dp, h, 0));
    for (int wgs : {16, 64, 256, 1024, 4096}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(copy16, dim3(wgs), dim3(256), 0, st, dsrc, reinterpret_cast<uint4*>(dp), n);
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      const unsigned char* hb = reinterpret_cast<const unsigned char*>(h);
      printf("%-20s kernel %5d workgroups: %7.3f ms = %6.1f GB/s  (last byte %d)\n", f.name, wgs, best, bytes / best / 1e6, hb[bytes - 1]);
    }
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(a, st)); CK(hipMemcpyAsync(h, dsrc, bytes, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-20s hipMemcpyAsync D2H        : %7.3f ms = %6.1f GB/s\n", f.name, best, bytes / best / 1e6);
    // small transfers (what a sequence time-step moves): 2 MB
    for (int wgs : {64, 256}) {
      float bs = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(copy16, dim3(wgs), dim3(256), 0, st, dsrc, reinterpret_cast<uint4*>(dp), (size_t)(2u << 20) / 16);
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < bs) bs = ms;
      }
      printf("%-20s kernel %5d workgroups, 2 MB: %7.3f ms = %6.1f GB/s\n", f.name, wgs, bs, (2u << 20) / bs / 1e6);
    }
    float bs = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a, st)); CK(hipMemcpyAsync(h, dsrc, 2u << 20, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < bs) bs = ms;
    }
    printf("%-20s hipMemcpyAsync D2H, 2 MB   : %7.3f ms = %6.1f GB/s\n", f.name, bs, (2u << 20) / bs / 1e6);
    CK(hipHostFree(h));
  }
  return 0;
}
