// How fast can EVERY CU pull the same 1.66 MB of weights out of L2 at the same time?  (The fused LightGlue block streams its weights once per workgroup;
// a design with two co-resident workgroups per CU would double that traffic: 256 CUs x 2 x 1.66 MB per ~30 us.)
// Every workgroup reads the whole buffer into registers `iters` times; reports us per sweep and the aggregate rate, for 1 / 2 workgroups per CU,
// 256 / 512 threads, all workgroups starting at the same offset or rotated (as lg_blockf rotates the feature blocks of its waves).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_weight_stream.hip -o tools/microbench/l2_weight_stream && tools/microbench/l2_weight_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void sweep(const uint4* w, size_t nvec, int iters, int rotate, unsigned long long* t, unsigned* sink) {
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned acc = 0;
  const size_t start = rotate ? ((size_t)(blockIdx.x >> 3) * 8192) % nvec : 0;     // 128 KiB steps among the workgroups of an XCD
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
    for (size_t i = tid; i < nvec; i += (size_t)nt * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { size_t j = start + i + (size_t)u * nt; if (j >= nvec) j -= nvec; v[u] = (i + (size_t)u * nt < nvec) ? w[j] : make_uint4(0, 0, 0, 0); }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  __syncthreads();
  const long long t1 = wall_clock64();
  if (tid == 0) { atomicAdd(t, (unsigned long long)(t1 - t0)); atomicAdd(t + 1, 1ull); }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t BYTES = 1703936, NVEC = BYTES / 16;
  const int ITERS = 50;
  uint4* w; unsigned long long* t; unsigned* sink;
  CHECK(hipMalloc(&w, BYTES)); CHECK(hipMemset(w, 1, BYTES)); CHECK(hipMalloc(&t, 64)); CHECK(hipMalloc(&sink, 64));
  printf("%5s %7s %6s | %12s %14s %12s\n", "WGs", "threads", "rotate", "us / sweep", "GB/s per WG", "TB/s chip");
  for (int grid : {28, 256, 512, 1024})
    for (int nt : {256, 512})
      for (int rot : {0, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
          CHECK(hipMemset(t, 0, 64));
          hipLaunchKernelGGL(sweep, dim3(grid), dim3(nt), nt == 256 ? 70000 : 0, 0, w, NVEC, ITERS, rot, t, sink);   // 70 KB of LDS: at most two 256-thread workgroups per CU
          CHECK(hipDeviceSynchronize());
        }
        unsigned long long ht[2];
        CHECK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost));
        const double us = ht[0] / (double)ht[1] / ITERS / 100.0;
        printf("%5d %7d %6d | %12.2f %14.1f %12.2f\n", grid, nt, rot, us, BYTES / us / 1e3, (grid > 512 ? 512 : grid) * BYTES / us / 1e6);
        fflush(stdout);
      }
  return 0;
}
