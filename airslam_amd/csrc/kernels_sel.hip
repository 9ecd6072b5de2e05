// airfe — keypoint selection: the compact candidate list of a heat map and the exact top-K (detect_point, src/plnet.cpp:309-355) on
// that list: a few thousand 8-byte candidates per image instead of six full-map select passes.  (simple_nms itself: kernels_nms512.hip
// for the reference's radius 4, kernels_img.hip for any other radius.)
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef unsigned long long u64;

// candidate key: valid bit | score bits (scores >= 0) | inverted raster index  => descending key order is
// "score descending, then raster index ascending" (the tie rule of SURVEY.md B.1)
__device__ __forceinline__ u64 make_key(float s, int idx) {
  return (1ull << 49) | ((u64)(__float_as_uint(s) & 0x7FFFFFFFu) << 18) | (u64)(0x3FFFF - idx);
}
__device__ __forceinline__ bool in_border_box(int x, int y, int W, int H, int border) {
  return !(x < border || x > W - border || y < border || y > H - border);      // upper bound INCLUSIVE (plnet.cpp:332)
}

// plain threshold + border compaction of a heat map into the candidate list (NMS off, or after the multi-pass NMS)
__global__ __launch_bounds__(256) void candidates_kernel(const float* __restrict__ heat, int H, int W, float thr, int border,
                                                         u64* __restrict__ cand, int* __restrict__ cand_cnt, int cand_cap) {
  // each workgroup owns one contiguous 1024-pixel span per iteration; LDS-aggregated append (one global atomic each)
  __shared__ u64 lkeys[1024];
  __shared__ int lcnt[2];
  const int b = blockIdx.y, N = H * W;
  const float* hm = heat + (size_t)b * N;
  for (int s0 = blockIdx.x * 1024; s0 < N; s0 += gridDim.x * 1024) {
    if (threadIdx.x == 0) lcnt[0] = 0;
    __syncthreads();
    const int i0 = s0 + threadIdx.x * 4;
    if (i0 < N) {
      const float4 v4 = *reinterpret_cast<const float4*>(hm + i0);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        if (i < N && !(v[k] < thr)) {
          const int y = i / W, x = i - y * W;
          if (in_border_box(x, y, W, H, border)) lkeys[atomicAdd(&lcnt[0], 1)] = make_key(v[k], i);
        }
      }
    }
    __syncthreads();
    const int n = lcnt[0];
    if (threadIdx.x == 0 && n > 0) lcnt[1] = atomicAdd(&cand_cnt[b], n);
    __syncthreads();
    const int base = lcnt[1];
    for (int i = threadIdx.x; i < n; i += 256)
      if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = lkeys[i];
    __syncthreads();
  }
}

// candidates_kernel over a * (a == max_pool2d(a, 3, stride 1, padding 1)) (HAWP's non_maximum_suppression, kernels_s0.hip s0_jnms_kernel) without the suppressed
// map in between: one launch and 2 x 64 KB per image less on the junction path
__global__ __launch_bounds__(256) void candidates_nms3_kernel(const float* __restrict__ heat, int H, int W, float thr, u64* __restrict__ cand,
                                                              int* __restrict__ cand_cnt, int cand_cap) {
  __shared__ u64 lkeys[1024];
  __shared__ int lcnt[2];
  const int b = blockIdx.y, N = H * W;
  const float* hm = heat + (size_t)b * N;
  for (int s0 = blockIdx.x * 1024; s0 < N; s0 += gridDim.x * 1024) {
    if (threadIdx.x == 0) lcnt[0] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = s0 + k * 256 + threadIdx.x;
      if (i < N) {
        const int y = i / W, x = i - y * W;
        const float a = hm[i];
        float m = a;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx, yy = y + dy;
            if (xx >= 0 && xx < W && yy >= 0 && yy < H) m = fmaxf(m, hm[yy * W + xx]);
          }
        const float v = (a == m) ? a : 0.f;
        if (!(v < thr)) lkeys[atomicAdd(&lcnt[0], 1)] = make_key(v, i);
      }
    }
    __syncthreads();
    const int n = lcnt[0];
    if (threadIdx.x == 0 && n > 0) lcnt[1] = atomicAdd(&cand_cnt[b], n);
    __syncthreads();
    const int base = lcnt[1];
    for (int i = threadIdx.x; i < n; i += 256)
      if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = lkeys[i];
    __syncthreads();
  }
}
void launch_candidates_nms3(const float* heat, int B, int H, int W, float thr, u64* cand, int* cand_cnt, int cand_cap, hipStream_t st) {
  (void)hipMemsetAsync(cand_cnt, 0, (size_t)B * sizeof(int), st);
  hipLaunchKernelGGL(candidates_nms3_kernel, dim3(64, B), dim3(256), 0, st, heat, H, W, thr, cand, cand_cnt, cand_cap);
}

void launch_candidates(const float* heat, int B, int H, int W, float thr, int border, u64* cand, int* cand_cnt, int cand_cap,
                       hipStream_t st) {
  (void)hipMemsetAsync(cand_cnt, 0, (size_t)B * sizeof(int), st);
  hipLaunchKernelGGL(candidates_kernel, dim3(64, B), dim3(256), 0, st, heat, H, W, thr, border, cand, cand_cnt, cand_cap);
}

// =============================================================================== exact top-K on the candidate list
// count <= K : all candidates in RASTER order (unsorted by score)       (plnet.cpp:348-353)
// count >  K : top K by key descending = score descending, ties by ascending raster index
// One 1024-thread workgroup per image: MSB radix select on the 49-bit key, then a bitonic sort of <= 1024 survivors.
__global__ __launch_bounds__(1024) void select_list_kernel(const u64* __restrict__ cand, const int* __restrict__ cand_cnt,
                                                           int cand_cap, int W, int topk, int cap, float* __restrict__ feat,
                                                           int* __restrict__ n_out) {
  __shared__ unsigned hist[2048];
  __shared__ u64 sk[1024];
  __shared__ unsigned wsum[16];
  __shared__ u64 s_prefix;
  __shared__ unsigned s_remaining, s_cnt, s_done;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = min(cand_cnt[b], cand_cap);
  const u64* keys = cand + (size_t)b * cand_cap;

  u64 prefix = 0;
  unsigned remaining = (unsigned)topk;
  const bool take_all = N <= topk;
  if (!take_all) {
    const int shifts[5] = {38, 27, 16, 5, 0};
    const int nbits[5] = {11, 11, 11, 11, 5};
    for (int pass = 0; pass < 5; ++pass) {
      for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
      __syncthreads();
      const int sh = shifts[pass];
      const u64 himask = ~((1ull << (sh + nbits[pass])) - 1ull);
      const unsigned dmask = (1u << nbits[pass]) - 1u;
      const u64 want = (prefix | (1ull << 49)) & himask;
      for (int i = tid; i < N; i += 1024) {
        const u64 k = keys[i];
        if ((k & himask) == want) atomicAdd(&hist[(unsigned)(k >> sh) & dmask], 1u);
      }
      __syncthreads();
      const unsigned h0 = hist[2047 - 2 * tid], h1 = hist[2046 - 2 * tid];
      const unsigned v = h0 + h1;
      unsigned incl = v;
      const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      unsigned woff = 0;
      for (int w2 = 0; w2 < wv; ++w2) woff += wsum[w2];
      incl += woff;
      const unsigned excl = incl - v;
      if (excl < remaining && remaining <= excl + h0) {
        s_prefix = prefix | ((u64)(2047 - 2 * tid) << sh);
        s_remaining = remaining - excl;
        s_done = (h0 == remaining - excl) ? 1u : 0u;
      } else if (excl + h0 < remaining && remaining <= incl) {
        s_prefix = prefix | ((u64)(2046 - 2 * tid) << sh);
        s_remaining = remaining - excl - h0;
        s_done = (h1 == remaining - excl - h0) ? 1u : 0u;
      }
      __syncthreads();
      prefix = s_prefix;
      remaining = s_remaining;
      const bool done = s_done != 0;
      __syncthreads();
      if (done) break;       // every key under this prefix is selected; the lower threshold bits stay 0
    }
  }
  const u64 T = prefix | (1ull << 49);
  if (tid == 0) s_cnt = 0;
  sk[tid] = ~0ull;
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const u64 k = keys[i];
    if (take_all || k >= T) {
      const unsigned slot = atomicAdd(&s_cnt, 1u);
      if (slot < 1024) {
        const u64 idx = 0x3FFFFull - (k & 0x3FFFFull);
        sk[slot] = take_all ? ((idx << 32) | ((k >> 18) & 0x7FFFFFFFull)) : ~k;      // raster asc / key desc
      }
    }
  }
  __syncthreads();
  const int n = min((int)s_cnt, min(topk, 1024));
  for (int k2 = 2; k2 <= 1024; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const int p = tid ^ j;
      if (p > tid) {
        const u64 a = sk[tid], c = sk[p];
        const bool up = (tid & k2) == 0;
        if ((a > c) == up) { sk[tid] = c; sk[p] = a; }
      }
      __syncthreads();
    }
  }
  if (tid < n) {
    const u64 e = sk[tid];
    int idx;
    unsigned sb;
    if (take_all) { idx = (int)(e >> 32); sb = (unsigned)(e & 0x7FFFFFFFull); }
    else { const u64 k = ~e; idx = 0x3FFFF - (int)(k & 0x3FFFFull); sb = (unsigned)((k >> 18) & 0x7FFFFFFFull); }
    const int y = idx / W, x = idx - y * W;
    float* f = feat + ((size_t)b * cap + tid) * 259;
    f[0] = __uint_as_float(sb);
    f[1] = (float)x;
    f[2] = (float)y;
  }
  if (tid == 0) n_out[b] = n;
}

void launch_select_list(const u64* cand, const int* cand_cnt, int cand_cap, int B, int W, int topk, int cap, float* feat,
                        int* n_out, hipStream_t st) {
  hipLaunchKernelGGL(select_list_kernel, dim3(B), dim3(1024), 0, st, cand, cand_cnt, cand_cap, W, topk, cap, feat, n_out);
}

}  // namespace airfe
