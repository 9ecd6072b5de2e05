// airfe — keypoint selection: simple_nms (radius 4) as five LDS-tiled max-pool launches, the last of which also emits the
// compact candidate list, and the exact top-K (detect_point, src/plnet.cpp:309-355) on that list: a few thousand 8-byte
// candidates per image instead of six full-map select passes.
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

// =============================================================================== simple_nms, radius 4
// eight sliding 9-maxima of 16 inputs with the three-input maximum: 14 + 8 = 22 v_max3_f32 (the two-input doubling scheme
// max(x, x+1) -> +2 -> +4 -> +x[8] took 45 v_max_f32)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void max9_of16(const float* x, float* o) {
  float t[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) t[i] = vmax3(x[i], x[i + 1], x[i + 2]);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = vmax3(t[i], t[i + 3], t[i + 6]);     // max over x[i .. i+8]
}

// Public SuperPoint simple_nms(scores, 4) = 5 dependent 9x9 max-pools, one LAUNCH per pool over 64x32 tiles with a 4-pixel
// halo (1.27x the work, 23 KB of LDS, 6 workgroups per CU); between launches only two byte planes (max_mask M, supp_mask D)
// travel through L2/HBM.  (Round 1a fused all five pools in one kernel: 20-pixel halo = a 48x48 tile computes 88x88, 78 KB
// of LDS, two workgroups per CU parked on barriers 60 % of the time — 0.69 ms per 64 images against 0.33 ms now.)
// Each pool is separable: 9-wide row maxima into T (8 outputs per thread from 16 inputs), then column maxima.
//   MODE 0: M  = S == pool(S)                                   (scores -> max_mask)
//   MODE 1: D  = pool(M) > 0                                    (max_mask -> supp_mask)
//   MODE 2: M |= !D && ss == pool(ss),  ss = D ? 0 : S          (supp_scores -> new max_mask)
//   MODE 3: MODE 2, then out = M ? S : 0 and the detect_point candidate list
constexpr int PT_W = 64, PT_H = 32, PR_W = PT_W + 8, PR_H = PT_H + 8, PP = PR_W + 1;

// Global traffic is 16-byte / 4-byte vectors only: a region row starts 4 pixels left of a 64-pixel tile, i.e. on a float4 (and
// uchar4) boundary, and W % 4 == 0 makes every vector wholly inside or wholly outside the image.  The byte planes are read and
// written through LDS byte tiles (one byte per lane and instruction was most of the load/store time of the first version).
template <int MODE>
__global__ __launch_bounds__(256) void nms_pool_kernel(const float* __restrict__ heat, unsigned char* __restrict__ Mg,
                                                       unsigned char* __restrict__ Dg, float* __restrict__ out, int H, int W,
                                                       int tiles_x, float thr, int border, u64* __restrict__ cand,
                                                       int* __restrict__ cand_cnt, int cand_cap) {
  __shared__ __attribute__((aligned(16))) float AT[2 * PR_H * PP];     // input region | row maxima; later the candidate keys
  __shared__ __attribute__((aligned(4))) unsigned char Dt[PR_H * PR_W];   // MODE 2/3: supp_mask of the region (0 outside the image)
  __shared__ __attribute__((aligned(4))) unsigned char Bt[PT_H * PT_W];   // byte plane of the tile: M in (MODE 2/3), result out
  __shared__ int lcnt[2];
  float* A = AT;
  float* T = AT + PR_H * PP;
  const int b = blockIdx.y, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int gy0 = ty * PT_H - 4, gx0 = tx * PT_W - 4;
  const size_t img = (size_t)b * H * W;
  const float* S = heat + img;
  unsigned char* M = Mg + img;
  unsigned char* D = Dg + img;
  constexpr int RV = PR_W / 4;                                          // 18 four-pixel vectors per region row
  for (int i = threadIdx.x; i < PR_H * RV; i += 256) {
    const int r = i / RV, c4 = i - r * RV, gy = gy0 + r, gx = gx0 + 4 * c4;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t gi = (size_t)gy * W + gx;
    float v[4];
    if (MODE == 0) {
      const float4 s4 = in ? *reinterpret_cast<const float4*>(S + gi) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      v[0] = s4.x; v[1] = s4.y; v[2] = s4.z; v[3] = s4.w;
    } else if (MODE == 1) {
      const uchar4 m4 = in ? *reinterpret_cast<const uchar4*>(M + gi) : make_uchar4(0, 0, 0, 0);
      v[0] = m4.x ? 1.f : 0.f; v[1] = m4.y ? 1.f : 0.f; v[2] = m4.z ? 1.f : 0.f; v[3] = m4.w ? 1.f : 0.f;
    } else {
      const float4 s4 = in ? *reinterpret_cast<const float4*>(S + gi) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      const uchar4 d4 = in ? *reinterpret_cast<const uchar4*>(D + gi) : make_uchar4(0, 0, 0, 0);
      v[0] = d4.x ? 0.f : s4.x; v[1] = d4.y ? 0.f : s4.y; v[2] = d4.z ? 0.f : s4.z; v[3] = d4.w ? 0.f : s4.w;
      *reinterpret_cast<uchar4*>(Dt + r * PR_W + 4 * c4) = d4;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) A[r * PP + 4 * c4 + k] = v[k];
  }
  if (MODE >= 2) {                                                      // max_mask of the tile itself, as 4-byte words
    for (int i = threadIdx.x; i < PT_H * PT_W / 4; i += 256) {
      const int r = i / (PT_W / 4), c4 = i - r * (PT_W / 4), gy = gy0 + 4 + r, gx = gx0 + 4 + 4 * c4;
      uchar4 m4 = make_uchar4(0, 0, 0, 0);
      if (gy < H && gx < W) m4 = *reinterpret_cast<const uchar4*>(M + (size_t)gy * W + gx);
      *reinterpret_cast<uchar4*>(Bt + r * PT_W + 4 * c4) = m4;
    }
  }
  if (MODE == 3 && threadIdx.x == 0) lcnt[0] = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < PR_H * (PT_W / 8); s += 256) {          // rows x 8 column strips
    const int k = s / PR_H, r = s - k * PR_H, c0 = 4 + k * 8;
    float x[16], o[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = A[r * PP + c0 - 4 + i];
    max9_of16(x, o);
#pragma unroll
    for (int i = 0; i < 8; ++i) T[r * PP + c0 + i] = o[i];
  }
  __syncthreads();
  float keep_v[8];                                                       // MODE 3: surviving scores of this thread's 8 pixels
  {
    const int s = threadIdx.x;                                           // 64 columns x 4 row strips = 256 tasks
    const int k = s >> 6, c = 4 + (s & 63), r0 = 4 + k * 8;
    float x[16], o[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = T[(r0 - 4 + i) * PP + c];
    max9_of16(x, o);
    const int gx = gx0 + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gy = gy0 + r0 + i;
      keep_v[i] = 0.f;
      const float a = A[(r0 + i) * PP + c];
      const int bi = (r0 - 4 + i) * PT_W + (c - 4);                      // this pixel in the byte tile
      if (MODE == 0) Bt[bi] = a == o[i];
      else if (MODE == 1) Bt[bi] = o[i] > 0.f;
      else {
        const bool d = Dt[(r0 + i) * PR_W + c];
        const bool m = Bt[bi] || (!d && a == o[i]);
        if (MODE == 2) Bt[bi] = m;
        else if (gy < H && gx < W) {
          const size_t gi = (size_t)gy * W + gx;
          // a kept pixel's score: `a` is the SUPPRESSED score (0 inside its own suppression zone), so the original is fetched —
          // only for the ~1 % of pixels that survive (a predicated load; the unconditional form re-read the whole map)
          float v = 0.f;
          if (m) v = S[gi];
          if (out) out[img + gi] = v;                    // the dense NMS map is optional: the batch path only needs the candidates
          keep_v[i] = v;
        }
      }
    }
  }
  if (MODE != 3) {                                                       // the byte plane of the tile leaves as 4-byte words
    __syncthreads();
    unsigned char* dst = MODE == 1 ? D : M;
    for (int i = threadIdx.x; i < PT_H * PT_W / 4; i += 256) {
      const int r = i / (PT_W / 4), c4 = i - r * (PT_W / 4), gy = gy0 + 4 + r, gx = gx0 + 4 + 4 * c4;
      if (gy < H && gx < W) *reinterpret_cast<uchar4*>(dst + (size_t)gy * W + gx) = *reinterpret_cast<const uchar4*>(Bt + r * PT_W + 4 * c4);
    }
  }
  if (MODE == 3) {
    // candidates are collected in LDS (both planes are free after the barrier: 2920 keys >= the 2048 pixels of a tile) and
    // appended with ONE global atomic per workgroup
    __syncthreads();
    u64* lkeys = reinterpret_cast<u64*>(AT);
    const int k = threadIdx.x >> 6, c = 4 + (threadIdx.x & 63), r0 = 4 + k * 8, gx = gx0 + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gy = gy0 + r0 + i;
      if (gy < H && gx < W && !(keep_v[i] < thr) && in_border_box(gx, gy, W, H, border))
        lkeys[atomicAdd(&lcnt[0], 1)] = make_key(keep_v[i], gy * W + gx);
    }
    __syncthreads();
    const int n = lcnt[0];
    if (threadIdx.x == 0 && n > 0) lcnt[1] = atomicAdd(&cand_cnt[b], n);
    __syncthreads();
    const int base = lcnt[1];
    for (int i = threadIdx.x; i < n; i += 256)
      if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = lkeys[i];
  }
}

// mask: 2 bytes per pixel of scratch ([B][H][W] max_mask, then [B][H][W] supp_mask)
void launch_nms4_candidates(const float* heat, float* out, unsigned char* mask, int B, int H, int W, float thr, int border,
                            u64* cand, int* cand_cnt, int cand_cap, hipStream_t st) {
  const int tiles_x = (W + PT_W - 1) / PT_W, tiles_y = (H + PT_H - 1) / PT_H;
  const dim3 grid(tiles_x * tiles_y, B);
  unsigned char* M = mask;
  unsigned char* D = mask + (size_t)B * H * W;
  (void)hipMemsetAsync(cand_cnt, 0, (size_t)B * sizeof(int), st);
#define NMS_POOL(MODE) hipLaunchKernelGGL(nms_pool_kernel<MODE>, grid, dim3(256), 0, st, heat, M, D, out, H, W, tiles_x, thr, border, cand, cand_cnt, cand_cap)
  NMS_POOL(0);
  NMS_POOL(1);
  NMS_POOL(2);
  NMS_POOL(1);
  NMS_POOL(3);
#undef NMS_POOL
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
