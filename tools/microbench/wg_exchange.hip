// What would a FEATURE-SPLIT LightGlue block cost at batch 1?  (lg_blockf_kernel streams ALL 1.7 MB of a block's weights through
// every workgroup: at 800 tokens that is 28 workgroups each bound by its own CU's ~100 GB/s vector-memory path, 24-27 us per block.)
// The alternative: G workgroups share a token tile, each owns 1/G of every GEMM's output features (1/G of the weights) and the activations go
// round through L2 between the GEMMs — four all-gathers per block, each = write my slice, group barrier, read the whole tile.
// This probe measures the two prices of that trade on the real part:
//   stream   us to pull this member's 1/G of the block's weights (L2-resident, the same 1.7 MB for every group) into registers, per workgroup
//   xchg     us per all-gather: T x K/G 2-byte slice written, agent-scope release, atomic counter + spin, acquire, whole T x K tile read back
// with agent-scope fences (L2 write-back + invalidate: correct wherever the members run) or the XCD-local protocol (members share one L2: nothing but
// an L1 invalidate; the errors column counts stale reads, 0x10000 each, and barrier time-outs),
// for G = 1 (today's shape: no exchange), 2, 4, 8, with the members of a group on ONE XCD (blockIdx = xcd + 8 slot: the dispatcher deals
// workgroups round-robin over the 8 XCDs) or dealt out in blockIdx order (eight different L2s).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/wg_exchange.hip -o tools/microbench/wg_exchange && tools/microbench/wg_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const uint4* wts; size_t wvec_per_wg;       // weight slice per workgroup, in 16-byte vectors
  uint4* tile; int T, K, G, ngroups, same_xcd, iters, local;   // local: XCD-local protocol (no L2 write-back / invalidate)
  unsigned* cnt; unsigned long long* t; unsigned* sink; unsigned* err;
};

// an atomic add that returns the old value, executed in the XCD's L2 whatever the compiler thinks of a "+ 0" (it folds an idempotent RMW into a load,
// which at workgroup scope may be served by the stale L1)
__device__ __forceinline__ unsigned l2_atomic_add(unsigned* p, unsigned v) {
  unsigned r;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
  return r;
}

__global__ __launch_bounds__(512, 1) void probe(Args a) {
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  const int tid = threadIdx.x;
  int group, member;
  if (a.same_xcd) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    group = xcd + 8 * (slot / a.G);
    member = slot % a.G;
  } else {
    group = blockIdx.x / a.G;
    member = blockIdx.x % a.G;
  }
  if (group >= a.ngroups) return;
  const size_t tile_vecs = (size_t)a.T * a.K * 2 / 16, slice_vecs = tile_vecs / a.G;
  uint4* tile = a.tile + (size_t)group * tile_vecs;
  unsigned acc = 0;
  unsigned long long t_stream = 0, t_xchg = 0;
  for (int it = 0; it < a.iters; ++it) {
    long long t0 = wall_clock64();
    // ---- weights: this workgroup's slice, 8 vectors per thread in flight
    const uint4* w = a.wts + (size_t)member * a.wvec_per_wg;              // (every group reads the same block of weights, like the real kernel)
    for (size_t i = tid; i < a.wvec_per_wg; i += 512 * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i + (size_t)u * 512 < a.wvec_per_wg) ? w[i + (size_t)u * 512] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    __syncthreads();
    long long t1 = wall_clock64();
    if (a.G > 1) {
      // ---- all-gather: my slice out, barrier, the whole tile in
      for (size_t i = tid; i < slice_vecs; i += 512) tile[(size_t)member * slice_vecs + i] = make_uint4(acc, it, member, (unsigned)i);
      if (a.local) {
        // members share ONE L2: the vector L1 is write-through, so a store is in L2 once vmcnt says it is done; the counter lives in L2 (atomics always
        // execute there); the reader only has to drop its own L1 (buffer_inv sc0) before it reads.  No L2 write-back, no L2 invalidate.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          l2_atomic_add(a.cnt + group, 1u);
          const unsigned target = (unsigned)a.G * (unsigned)(it + 1);
          int spins = 0;
          while (l2_atomic_add(a.cnt + group, 0u) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000) { atomicAdd(a.err, 1u); dead = 1; break; }
          }
        }
        __syncthreads();
        asm volatile("buffer_inv sc0" ::: "memory");
      } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(a.cnt + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)a.G * (unsigned)(it + 1);
        int spins = 0;
        while (__hip_atomic_load(a.cnt + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 20000) { atomicAdd(a.err, 1u); dead = 1; break; }
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      for (size_t i = tid; i < tile_vecs; i += 512) {
        const uint4 v = tile[i];
        acc += v.x + v.w;
        if (v.y != (unsigned)it) atomicAdd(a.err, 1u << 16);              // a stale line: the exchange is not coherent
      }
      __syncthreads();
    }
    long long t2 = wall_clock64();
    if (dead) break;                                                     // a barrier timed out (members that cannot see each other's counter): give up
    t_stream += (unsigned long long)(t1 - t0);
    t_xchg += (unsigned long long)(t2 - t1);
  }
  if (tid == 0) {
    atomicAdd(a.t, t_stream);
    atomicAdd(a.t + 1, t_xchg);
    atomicAdd(a.t + 2, 1ull);
  }
  if (acc == 0x12345678u) a.sink[0] = acc;
}

int main() {
  const size_t WBYTES = 1703936;                 // one fused block's weights: 256x256 + 512x512 + 512x256 + 256x768 two-byte elements
  const int TOK = 800, ITERS = 200;
  uint4 *wts, *tile;
  unsigned *cnt, *sink, *err;
  unsigned long long* t;
  CHECK(hipMalloc(&wts, WBYTES));
  CHECK(hipMemset(wts, 1, WBYTES));
  CHECK(hipMalloc(&tile, (size_t)64 * 128 * 512 * 2));
  CHECK(hipMalloc(&cnt, 4096)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&err, 64)); CHECK(hipMalloc(&t, 64));
  printf("%-10s %-6s %3s %5s %4s %7s | %10s %10s %8s\n", "placement", "fences", "G", "T", "WGs", "KB/WG", "stream us", "xchg us", "errors");
  for (int mode = 0; mode < 4; ++mode)                                     // {one XCD, blockIdx} x {agent-scope fences, XCD-local protocol}
    for (int G : {1, 2, 4, 8}) {
      const int same = mode == 0 || mode == 2, local = mode >= 2;
      if (G == 1 && mode) continue;
      const int T = 32 * G > 128 ? 128 : 32 * G;                         // tokens per group: 32 per member, at most 128
      const int ngroups = (TOK + T - 1) / T;
      for (int K : {256, 512}) {
        Args a;
        a.wts = wts; a.wvec_per_wg = WBYTES / G / 16; a.tile = tile; a.T = T; a.K = K; a.G = G; a.ngroups = ngroups; a.same_xcd = same; a.iters = ITERS; a.local = local;
        a.cnt = cnt; a.t = t; a.sink = sink; a.err = err;
        const int slots = (ngroups + 7) / 8 * G;                         // same-XCD placement: slots per XCD
        const int grid = same ? 8 * slots : ngroups * G;
        for (int rep = 0; rep < 2; ++rep) {                              // second launch is the warm one
          CHECK(hipMemset(cnt, 0, 4096)); CHECK(hipMemset(t, 0, 64)); CHECK(hipMemset(err, 0, 64));
          hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, a);
          CHECK(hipDeviceSynchronize());
        }
        unsigned long long ht[3]; unsigned herr;
        CHECK(hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        const double n = (double)ht[2] * ITERS;
        printf("%-10s %-6s %3d %5d %4d %7.0f | %10.2f %10.2f %8x   (tile %d x %d)\n", same ? "one XCD" : "blockIdx", local ? "local" : "agent", G, T, ngroups * G, WBYTES / G / 1024.0,
               ht[0] / n / 100.0, ht[1] / n / 100.0, herr, T, K);
        fflush(stdout);
      }
    }
  return 0;
}
