// airfe — PLNet line path (wireframe_matcher, stage-1 LOI head, line/junction filter) and the SuperGlue-specific
// pieces (keypoint encoder, log-domain Sinkhorn, decode).  Small, irregular, fp32: VALU + LDS, no MFMA.
#include <float.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace airfe {

// block-wide exclusive scan of one unsigned per thread (1024 threads); returns exclusive prefix, *total = sum
__device__ __forceinline__ unsigned block_excl_scan_1024(unsigned v, unsigned* wsum /*[16] LDS*/, unsigned* total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();            // protect wsum from the previous use
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  unsigned off = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    const unsigned t = wsum[w];
    if (w < wv) off += t;
    tot += t;
  }
  *total = tot;
  return off + incl - v;
}

// =============================================================================== wireframe_matcher
// src/plnet.cpp:272-307 on the device.  keep = raster-ordered indices with iskeep > 0; unique (min,max) junction pairs
// get ids in FIRST-SEEN order; rep[u] = position (in keep) of the first proposal of unique line u — which is also the
// `perm` the stage-1 graph rebuilds with its reversed ScatterElements (oracle/onnx_run.py).
// Three launches: the raster-ordered list of kept proposals is built by WF_WGS workgroups (count, then emit at the prefix of the counts
// — ONE workgroup walking the 49152-entry map was latency-bound at 47 us per frame), the unique pairs by one workgroup over that list.
// Every kernel of the line path takes one image per grid row (blockIdx.y): iskeep / imin / imax / juncs / lines_pred / thin / aux are
// image 0's pointers into its stage block (image b: + b * stage_stride floats), the work lists are dense per image.
constexpr int WF_WGS = 48;

__global__ __launch_bounds__(256) void wf_count_kernel(const float* __restrict__ iskeep, int n, int* __restrict__ counts, size_t stage_stride) {
  __shared__ int wsum[4];
  iskeep += (size_t)blockIdx.y * stage_stride;
  int* wg_counts = counts + (size_t)blockIdx.y * LINE_CNT_LD + 2;
  const int per_wg = (n + WF_WGS - 1) / WF_WGS, lo = blockIdx.x * per_wg, hi = min(lo + per_wg, n);
  int cnt = 0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) cnt += iskeep[i] > 0.f;
  cnt = (int)wave_sum((float)cnt);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) wg_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void wf_emit_kernel(const float* __restrict__ iskeep, int n, const int* __restrict__ counts,
                                                      int* __restrict__ keep, int cap, size_t stage_stride) {
  __shared__ int wcnt[4];
  iskeep += (size_t)blockIdx.y * stage_stride;
  keep += (size_t)blockIdx.y * cap;
  const int* wg_counts = counts + (size_t)blockIdx.y * LINE_CNT_LD + 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per_wg = (n + WF_WGS - 1) / WF_WGS, lo = blockIdx.x * per_wg, hi = min(lo + per_wg, n);
  int base = 0;
  for (int w = 0; w < (int)blockIdx.x; ++w) base += wg_counts[w];
  // every wave owns a contiguous quarter of the workgroup's run and walks it 64 at a time: positions by ballot + popcount
  const int seg = (hi - lo + 3) / 4, s_lo = lo + wv * seg, s_hi = min(s_lo + seg, hi);
  int wc = 0;
  for (int b0 = s_lo; b0 < s_hi; b0 += 64) {
    const int i = b0 + lane;
    wc += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < s_hi && iskeep[i] > 0.f));
  }
  if (lane == 0) wcnt[wv] = wc;
  __syncthreads();
  int off = base;
  for (int w = 0; w < wv; ++w) off += wcnt[w];
  for (int b0 = s_lo; b0 < s_hi; b0 += 64) {
    const int i = b0 + lane;
    const bool k = i < s_hi && iskeep[i] > 0.f;
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(k);
    if (k) {
      const int pos = off + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
      if (pos < cap) keep[pos] = i;
    }
    off += __builtin_popcountll(mk);
  }
}

__global__ __launch_bounds__(1024) void wireframe_kernel(const float* __restrict__ imin, const float* __restrict__ imax, int jn, int* table,
                                                         const int* __restrict__ keep, int* __restrict__ pairs, int* __restrict__ rep,
                                                         int cap, int line_cap, int* __restrict__ counts, size_t stage_stride) {
  __shared__ unsigned wcnt[16];
  {
    const size_t img = blockIdx.y;
    imin += img * stage_stride; imax += img * stage_stride;
    table += img * jn * jn; keep += img * cap; pairs += img * line_cap * 2; rep += img * line_cap; counts += img * LINE_CNT_LD;
  }
  const int* wg_counts = counts + 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned m1 = 0;
  for (int w = 0; w < WF_WGS; ++w) m1 += (unsigned)wg_counts[w];
  m1 = min(m1, (unsigned)cap);
  for (unsigned k = tid; k < m1; k += 1024) {
    const int i = keep[k];
    const int a = (int)imin[i], b = (int)imax[i];
    if (a >= 0 && a < jn && b >= 0 && b < jn) atomicMin(&table[a * jn + b], (int)k);
  }
  __syncthreads();
  // the first proposal of every unique pair, in keep order: the wave-segment walk over keep[0 .. m1)
  const int seg2 = ((int)m1 + 15) / 16, t_lo = wv * seg2, t_hi = min(t_lo + seg2, (int)m1);
  auto first_of_pair = [&](int k, int& a, int& b) {
    const int i = keep[k];
    a = (int)imin[i]; b = (int)imax[i];
    return (a >= 0 && a < jn && b >= 0 && b < jn) && __hip_atomic_load(&table[a * jn + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k;
  };
  unsigned wc2 = 0;
  for (int base = t_lo; base < t_hi; base += 64) {
    const int k = base + lane;
    int a, b;
    wc2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(k < t_hi && first_of_pair(k, a, b)));
  }
  if (lane == 0) wcnt[wv] = wc2;
  __syncthreads();
  unsigned off2 = 0, m2 = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wv) off2 += wcnt[w];
    m2 += wcnt[w];
  }
  for (int base = t_lo; base < t_hi; base += 64) {
    const int k = base + lane;
    int a = 0, b = 0;
    const bool f = k < t_hi && first_of_pair(k, a, b);
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(f);
    if (f) {
      const unsigned pos = off2 + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
      if (pos < (unsigned)line_cap) {
        rep[pos] = k;
        pairs[pos * 2] = b;           // (max, min): plnet.cpp:301
        pairs[pos * 2 + 1] = a;
      }
    }
    off2 += __builtin_popcountll(mk);
  }
  __syncthreads();
  for (unsigned k = tid; k < m1; k += 1024) {      // leave the table clean for the next call
    const int i = keep[k];
    const int a = (int)imin[i], b = (int)imax[i];
    if (a >= 0 && a < jn && b >= 0 && b < jn) table[a * jn + b] = 0x7FFFFFFF;
  }
  if (tid == 0) { counts[0] = (int)m1; counts[1] = (int)min(m2, (unsigned)line_cap); }
}

// counts: LINE_CNT_LD ints per image — [0] M1, [1] M2, [2 .. 2 + WF_WGS) scratch (per-workgroup counts);
// table [B][jn * jn] (0x7FFFFFFF everywhere), keep [B][cap], pairs [B][line_cap][2], rep [B][line_cap]
void launch_wireframe(const float* iskeep, const float* imin, const float* imax, int n, int jn, int* table, int* keep,
                      int* pairs, int* rep, int cap, int line_cap, int* counts, int B, size_t stage_stride, hipStream_t st) {
  hipLaunchKernelGGL(wf_count_kernel, dim3(WF_WGS, B), dim3(256), 0, st, iskeep, n, counts, stage_stride);
  hipLaunchKernelGGL(wf_emit_kernel, dim3(WF_WGS, B), dim3(256), 0, st, iskeep, n, counts, keep, cap, stage_stride);
  hipLaunchKernelGGL(wireframe_kernel, dim3(1, B), dim3(1024), 0, st, imin, imax, jn, table, keep, pairs, rep, cap, line_cap, counts,
                     stage_stride);
}

// =============================================================================== stage-1 LOI head
// plnet_s1.onnx restated (oracle/ref_nets.py::plnet_s1_forward), fp32 throughout.
// A feature plane is addressed as f[(y W + x) ps]: ps = 1 for the contract's CHW planes, the row pitch of the head GEMM's output when the
// LOI features are sampled where that GEMM left them (one pixel's 128 channels are then one 512-byte run: the 128 lanes' taps coalesce).
__device__ __forceinline__ float bil_plane(const float* __restrict__ f, int H, int W, int ps, float x, float y) {
  const float px = x - 0.5f, py = y - 0.5f;
  const float x0 = fminf(fmaxf(floorf(px), 0.f), (float)(W - 1)), y0 = fminf(fmaxf(floorf(py), 0.f), (float)(H - 1));
  const float x1 = fminf(fmaxf(x0 + 1.f, 0.f), (float)(W - 1)), y1 = fminf(fmaxf(y0 + 1.f, 0.f), (float)(H - 1));
  const int x0i = (int)x0, y0i = (int)y0, x1i = (int)x1, y1i = (int)y1;
  return f[(y0i * W + x0i) * ps] * (y1 - py) * (x1 - px) + f[(y1i * W + x0i) * ps] * (py - y0) * (x1 - px) +
         f[(y0i * W + x1i) * ps] * (y1 - py) * (px - x0) + f[(y1i * W + x1i) * ps] * (py - y0) * (px - x0);
}

constexpr int S1_LT = 8;   // lines per workgroup

template <int K>
__device__ __forceinline__ void s1_dense(const float* __restrict__ wt /*[K][128]*/, const float* __restrict__ bias,
                                         const float* xin /*LDS [LT][ldx]*/, int ldx, float* out /*[LT]*/, int n) {
  float acc[S1_LT];
#pragma unroll
  for (int l = 0; l < S1_LT; ++l) acc[l] = bias[n];
  for (int k = 0; k < K; ++k) {
    const float w = wt[k * 128 + n];
#pragma unroll
    for (int l = 0; l < S1_LT; ++l) acc[l] = fmaf(w, xin[l * ldx + k], acc[l]);
  }
#pragma unroll
  for (int l = 0; l < S1_LT; ++l) out[l] = acc[l];
}

struct S1Weights {
  const float *w0t, *b0, *w2t, *b2, *w4t, *b4, *wrt, *br, *wh, *bh, *tt;   // *t = transposed [K][128]; wh [2][128]
};
struct S1Loi {                       // where image b's LOI feature (channel ch, pixel p) lives: base[b * img + ch * cs + p * ps]
  const float* base;
  size_t img;
  int cs, ps;
};

__global__ __launch_bounds__(128) void plnet_s1_kernel(const float* __restrict__ juncs, const float* __restrict__ lines_pred,
                                                       const int* __restrict__ keep, const int* __restrict__ pairs,
                                                       const int* __restrict__ rep, const int* __restrict__ counts, S1Loi loi,
                                                       const float* __restrict__ thin, const float* __restrict__ aux, S1Weights w,
                                                       float* __restrict__ lines_adjusted, float* __restrict__ scores_line,
                                                       int keep_cap, int line_cap, size_t stage_stride) {
  __shared__ float xs[S1_LT][496];
  __shared__ float h0[S1_LT][128], h1[S1_LT][128];
  __shared__ float la[S1_LT][4], li[S1_LT][4];
  {
    const size_t img = blockIdx.y;
    juncs += img * stage_stride; lines_pred += img * stage_stride; thin += img * stage_stride; aux += img * stage_stride;
    keep += img * keep_cap; pairs += img * line_cap * 2; rep += img * line_cap; counts += img * LINE_CNT_LD;
    lines_adjusted += img * line_cap * 4; scores_line += img * line_cap;
    loi.base += img * loi.img;
  }
  const int m2 = counts[1];
  const int tid = threadIdx.x;
  // a workgroup walks the image's line tiles with the grid's stride (the count is on the device: no launch sized by it)
  for (int l0 = blockIdx.x * S1_LT; l0 < m2; l0 += gridDim.x * S1_LT) {
  __syncthreads();
  if (tid < S1_LT * 4) {
    const int l = tid >> 2, c = tid & 3, u = min(l0 + l, m2 - 1);
    const int j = pairs[u * 2 + (c >> 1)];
    const float v = juncs[j * 2 + (c & 1)];
    la[l][c] = v;
    li[l][c] = lines_pred[(size_t)keep[rep[u]] * 4 + c];
    if (l0 + l < m2) lines_adjusted[(size_t)(l0 + l) * 4 + c] = v;
  }
  __syncthreads();
  const float* lch = loi.base + (size_t)tid * loi.cs;
  for (int l = 0; l < S1_LT; ++l) {
    xs[l][tid] = bil_plane(lch, 128, 128, loi.ps, la[l][0], la[l][1]);
    xs[l][128 + tid] = bil_plane(lch, 128, 128, loi.ps, la[l][2], la[l][3]);
    if (tid < 120) {
      const int c = tid / 30, j = tid - c * 30;
      const float t = w.tt[j], t1 = 1.0f - t;
      xs[l][256 + tid] = bil_plane(thin + (size_t)c * 128 * 128, 128, 128, 1, la[l][0] * t + la[l][2] * t1, la[l][1] * t + la[l][3] * t1);
      xs[l][376 + tid] = bil_plane(aux + (size_t)c * 128 * 128, 128, 128, 1, li[l][0] * t + li[l][2] * t1, li[l][1] * t + li[l][3] * t1);
    }
  }
  __syncthreads();
  float o[S1_LT], r[S1_LT];
  s1_dense<496>(w.w0t, w.b0, &xs[0][0], 496, o, tid);
#pragma unroll
  for (int l = 0; l < S1_LT; ++l) h0[l][tid] = fmaxf(o[l], 0.f);
  s1_dense<240>(w.wrt, w.br, &xs[0][256], 496, r, tid);
  __syncthreads();
  s1_dense<128>(w.w2t, w.b2, &h0[0][0], 128, o, tid);
#pragma unroll
  for (int l = 0; l < S1_LT; ++l) h1[l][tid] = fmaxf(o[l], 0.f);
  __syncthreads();
  s1_dense<128>(w.w4t, w.b4, &h1[0][0], 128, o, tid);
  __syncthreads();
#pragma unroll
  for (int l = 0; l < S1_LT; ++l) h0[l][tid] = o[l] + fmaxf(r[l], 0.f);
  __syncthreads();
  if (tid < S1_LT * 2) {
    const int l = tid >> 1, c = tid & 1;
    float z = w.bh[c];
    for (int k = 0; k < 128; ++k) z = fmaf(w.wh[c * 128 + k], h0[l][k], z);
    h1[l][c] = z;
  }
  __syncthreads();
  if (tid < S1_LT && l0 + tid < m2) {
    const float z0 = h1[tid][0], z1 = h1[tid][1], m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    scores_line[l0 + tid] = e1 / (e0 + e1);
  }
  }
}

// loi: the contract's CHW block of the stage (loi_ps = 1, loi_cs = 128 * 128, loi_img = stage_stride) or the head GEMM's rows
// (loi_ps = row pitch, loi_cs = 1, loi_img = 128 * 128 * pitch).  keep [B][keep_cap], pairs / rep / lines_adjusted / scores_line [B][line_cap].
void launch_plnet_s1(const float* juncs, const float* lines_pred, const int* keep, const int* pairs, const int* rep,
                     const int* counts, const float* loi, size_t loi_img, int loi_cs, int loi_ps, const float* thin, const float* aux,
                     const float* const* w, float* lines_adjusted, float* scores_line, int keep_cap, int line_cap, int B,
                     size_t stage_stride, hipStream_t st) {
  S1Weights sw{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10]};
  S1Loi sl{loi, loi_img, loi_cs, loi_ps};
  const int gx = B == 1 ? 512 : (B <= 8 ? 128 : 48);
  hipLaunchKernelGGL(plnet_s1_kernel, dim3(gx, B), dim3(128), 0, st, juncs, lines_pred, keep, pairs, rep, counts, sl, thin, aux, sw,
                     lines_adjusted, scores_line, keep_cap, line_cap, stage_stride);
}

// =============================================================================== line filter + junction map
// src/plnet.cpp:519-558 (+ rescale :577-582).  junction_map[y][x] = p_valid(x,y) is a pure function of the pixel,
// so the sequential "assignment, not OR" of the reference is order-independent; lines are emitted in ascending i.
__global__ __launch_bounds__(1024) void line_filter_kernel(const float* __restrict__ la, const float* __restrict__ sc,
                                                           const int* __restrict__ counts, int border, float line_thr,
                                                           float len_thr, float w_scale, float h_scale, int R,
                                                           unsigned char* __restrict__ jmap, double* __restrict__ lines_out,
                                                           int capL, int* __restrict__ nlines, int* __restrict__ nfound, int line_cap) {
  __shared__ unsigned wsum[16];
  {
    const size_t img = blockIdx.x;
    la += img * line_cap * 4; sc += img * line_cap; counts += img * LINE_CNT_LD; jmap += img * R * R;
    lines_out += img * capL * 4; nlines += img;
    if (nfound) nfound += img;
  }
  const int tid = threadIdx.x, m2 = counts[1];
  const int per = (m2 + 1023) / 1024, lo = tid * per, hi = min(lo + per, m2);
  const float thr2 = __fmul_rn(len_thr, len_thr);
  border = max(border, 0);
  unsigned cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    unsigned off = 0, tot = 0;
    if (pass == 1) off = block_excl_scan_1024(cnt, wsum, &tot);
    for (int i = lo; i < hi; ++i) {
      const float s = sc[i];
      if (s < 0.5f) continue;
      const float x1 = __fmul_rn(la[i * 4], 4.f), y1 = __fmul_rn(la[i * 4 + 1], 4.f);
      const float x2 = __fmul_rn(la[i * 4 + 2], 4.f), y2 = __fmul_rn(la[i * 4 + 3], 4.f);
      if (pass == 0) {
        const int xi1 = (int)((double)x1 + 0.1), yi1 = (int)((double)y1 + 0.1);
        const int xi2 = (int)((double)x2 + 0.1), yi2 = (int)((double)y2 + 0.1);
        const bool p1 = xi1 > border && xi1 < R - border && yi1 > border && yi1 < R - border;
        const bool p2 = xi2 > border && xi2 < R - border && yi2 > border && yi2 < R - border;
        if (xi1 >= 0 && xi1 < R && yi1 >= 0 && yi1 < R) jmap[yi1 * R + xi1] = p1;
        if (xi2 >= 0 && xi2 < R && yi2 >= 0 && yi2 < R) jmap[yi2 * R + xi2] = p2;
      }
      if (s < line_thr) continue;
      const float dx = __fsub_rn(x2, x1), dy = __fsub_rn(y2, y1);
      const float l2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
      if (l2 < thr2) continue;
      if (pass == 0) ++cnt;
      else {
        if (off < (unsigned)capL) {
          lines_out[(size_t)off * 4 + 0] = (double)x1 * (double)w_scale;
          lines_out[(size_t)off * 4 + 1] = (double)y1 * (double)h_scale;
          lines_out[(size_t)off * 4 + 2] = (double)x2 * (double)w_scale;
          lines_out[(size_t)off * 4 + 3] = (double)y2 * (double)h_scale;
        }
        ++off;
      }
    }
    if (pass == 1 && tid == 0) {
      *nlines = min((int)tot, capL);
      if (nfound) *nfound = (int)tot;        // what passed the filter: more than capL is the caller's overflow to report
    }
  }
}

// one workgroup per image: la [B][line_cap][4], sc [B][line_cap], jmap [B][R * R] (zeroed by the caller), lines_out [B][capL][4], nlines /
// nfound [B] (nfound may be nullptr)
void launch_line_filter(const float* la, const float* sc, const int* counts, int border, float line_thr, float len_thr,
                        float w_scale, float h_scale, int R, unsigned char* jmap, double* lines_out, int capL, int* nlines, int* nfound,
                        int line_cap, int B, hipStream_t st) {
  hipLaunchKernelGGL(line_filter_kernel, dim3(B), dim3(1024), 0, st, la, sc, counts, border, line_thr, len_thr, w_scale,
                     h_scale, R, jmap, lines_out, capL, nlines, nfound, line_cap);
}

// junction_detector (src/plnet.cpp:425-448): raster scan of the junction map inside [border, R-border) (EXCLUSIVE upper)
// Two launches of 64 workgroups per image (one 1024-thread workgroup walking the whole 512 x 512 map took 192 us per frame — longer than
// the encoder at batch 1): every workgroup owns a contiguous run of pixels, 16 per thread; counts first, then the ordered emit with the
// sum of the preceding workgroups' counts as its base.  Raster order (the reference's scan order) is kept.
constexpr int JS_WGS = 64, JS_PT = 16;

__device__ __forceinline__ unsigned js_mask16(const unsigned char* __restrict__ jmap, int i0, int N, int R, int border) {
  unsigned m = 0;
  if (i0 + JS_PT <= N && (R % JS_PT) == 0 && (i0 % JS_PT) == 0) {   // the 16 pixels are in one row, aligned: one 16-byte load
    const uint4 q = *reinterpret_cast<const uint4*>(jmap + i0);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    const int y = i0 / R, x0 = i0 - y * R;
    const bool yin = y >= border && y < R - border;
#pragma unroll
    for (int k = 0; k < JS_PT; ++k) {
      const int x = x0 + k;
      if (((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) && yin && x >= border && x < R - border) m |= 1u << k;
    }
  } else {
    for (int k = 0; k < JS_PT; ++k) {
      const int i = i0 + k;
      if (i >= N) break;
      const int y = i / R, x = i - y * R;
      if (jmap[i] && x >= border && x < R - border && y >= border && y < R - border) m |= 1u << k;
    }
  }
  return m;
}

__global__ __launch_bounds__(256) void junction_count_kernel(const unsigned char* __restrict__ jmap, int R, int border, int* __restrict__ wg_counts) {
  __shared__ int wsum[4];
  const int N = R * R, per_wg = (N + JS_WGS - 1) / JS_WGS, per = (per_wg + 255) / 256;
  jmap += (size_t)blockIdx.y * N; wg_counts += (size_t)blockIdx.y * JS_WGS;
  border = max(border, 0);
  int cnt = 0;
  const int lo = blockIdx.x * per_wg + threadIdx.x * per, hi = min(min(lo + per, (blockIdx.x + 1) * per_wg), N);
  for (int i0 = lo; i0 < hi; i0 += JS_PT) cnt += __builtin_popcount(js_mask16(jmap, i0, min(hi, N), R, border));
  cnt = (int)wave_sum((float)cnt);                                   // < 2^24: exact in fp32
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) wg_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void junction_emit_kernel(const unsigned char* __restrict__ jmap, const float* __restrict__ heat, int R, int border,
                                                            float* __restrict__ feat, int cap, int* __restrict__ n_kept, int* __restrict__ n_found,
                                                            const int* __restrict__ wg_counts) {
  __shared__ int wsum[4];
  const int N = R * R, per_wg = (N + JS_WGS - 1) / JS_WGS, per = (per_wg + 255) / 256;
  {
    const size_t img = blockIdx.y;
    jmap += img * N; heat += img * N; feat += img * cap * 259; wg_counts += img * JS_WGS;
    n_kept += img; n_found += img;
  }
  border = max(border, 0);
  int base = 0;
  for (int w = 0; w < (int)blockIdx.x; ++w) base += wg_counts[w];
  const int lo = blockIdx.x * per_wg + threadIdx.x * per, hi = min(min(lo + per, (blockIdx.x + 1) * per_wg), N);
  int cnt = 0;
  for (int i0 = lo; i0 < hi; i0 += JS_PT) cnt += __builtin_popcount(js_mask16(jmap, i0, min(hi, N), R, border));
  // exclusive prefix over the 256 threads: within the wave by shuffles, then over the 4 waves
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += t;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int off = base + incl - cnt;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
  for (int i0 = lo; i0 < hi; i0 += JS_PT) {
    unsigned m = js_mask16(jmap, i0, min(hi, N), R, border);
    while (m) {
      const int k = __builtin_ctz(m);
      m &= m - 1;
      const int i = i0 + k, y = i / R, x = i - y * R;
      if (off < cap) {
        float* f = feat + (size_t)off * 259;
        f[0] = heat[i];
        f[1] = (float)x;
        f[2] = (float)y;
      }
      ++off;
    }
  }
  if (blockIdx.x == JS_WGS - 1 && threadIdx.x == 255) { *n_kept = min(off, cap); *n_found = off; }   // found > cap: the caller reports the overflow
}

// jmap / heat [B][R * R], feat [B][cap][259], n_kept / n_found [B], wg_counts [B][64] scratch
void launch_junction_scan(const unsigned char* jmap, const float* heat, int R, int border, float* feat, int cap, int* n_kept, int* n_found,
                          int* wg_counts, int B, hipStream_t st) {
  hipLaunchKernelGGL(junction_count_kernel, dim3(JS_WGS, B), dim3(256), 0, st, jmap, R, border, wg_counts);
  hipLaunchKernelGGL(junction_emit_kernel, dim3(JS_WGS, B), dim3(256), 0, st, jmap, heat, R, border, feat, cap, n_kept, n_found, wg_counts);
}

// =============================================================================== SuperGlue: keypoint encoder
// KeypointEncoder MLP [3,32,64,128,256,256] (Conv1d k=1, BN folded, ReLU) on (x, y, score), added to the descriptor.
// process_input layout: src/super_glue.cpp:199-246.  4 keypoints per 256-thread workgroup, weights transposed [K][N].
constexpr int KE_LT = 4;

template <int K, int N>
__device__ __forceinline__ void ke_layer(const float* __restrict__ wt, const float* __restrict__ b, const float* in, int ldi,
                                         float* out, int ldo, bool relu) {
  const int n = threadIdx.x;
  if (n < N) {
    float acc[KE_LT];
#pragma unroll
    for (int l = 0; l < KE_LT; ++l) acc[l] = b[n];
    for (int k = 0; k < K; ++k) {
      const float w = wt[k * N + n];
#pragma unroll
      for (int l = 0; l < KE_LT; ++l) acc[l] = fmaf(w, in[l * ldi + k], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < KE_LT; ++l) out[l * ldo + n] = relu ? fmaxf(acc[l], 0.f) : acc[l];
  }
}

struct SgPrepArgs {
  const float* f0; const float* f1; const int* n0; const int* n1;
  int ld, normalize; float cx, cy, linv;
  const float* w[10];   // w0t,b0,...,w4t,b4
  int B, cap, Np;
  float* x32; uint16_t* xb; int* lens;
  uint16_t* h128;       // SPLIT: [S * Np][128] 2-byte output of the third layer
};

// SPLIT: only the three small layers (3 -> 32 -> 64 -> 128: 10 of the 108 kFLOP per keypoint) run here; the kernel writes the 128
// hidden features (2-byte) and x = the descriptor, and the two large layers (128 -> 256 + ReLU, 256 -> 256 added to x) follow as MFMA
// GEMMs (airfe.hip).  As scalar FMA loops all five layers took 0.29 ms per 51200 keypoints, latency-bound on LDS broadcast reads.
template <class P, bool SPLIT>
__global__ __launch_bounds__(256) void sg_prepare_kernel(SgPrepArgs a) {
  __shared__ float bufa[KE_LT][256], bufb[KE_LT][256];
  const int s = blockIdx.y, n0r = blockIdx.x * KE_LT, tid = threadIdx.x;
  const int b = s >> 1, side = s & 1;
  const int len = side ? a.n1[b] : a.n0[b];
  if (blockIdx.x == 0 && tid == 0) a.lens[s] = len;
  const float* fbase = (side ? a.f1 : a.f0) + (size_t)b * a.cap * a.ld;
  if (tid < KE_LT * 3) {
    const int l = tid / 3, c = tid - l * 3, n = n0r + l;
    float v = 0.f;
    if (n < len) {
      const float* f = fbase + (size_t)n * a.ld;
      if (c == 2) v = f[0];                                     // score
      else {
        v = f[1 + c];
        if (a.normalize) v = __fmul_rn(__fsub_rn(v, c == 0 ? a.cx : a.cy), a.linv);
      }
    }
    bufa[l][c] = v;
  }
  __syncthreads();
  ke_layer<3, 32>(a.w[0], a.w[1], &bufa[0][0], 256, &bufb[0][0], 256, true);
  __syncthreads();
  ke_layer<32, 64>(a.w[2], a.w[3], &bufb[0][0], 256, &bufa[0][0], 256, true);
  __syncthreads();
  ke_layer<64, 128>(a.w[4], a.w[5], &bufa[0][0], 256, &bufb[0][0], 256, true);
  __syncthreads();
  if constexpr (SPLIT) {
    for (int l = 0; l < KE_LT; ++l) {
      const int n = n0r + l;
      if (n >= a.Np) break;
      const size_t row = (size_t)s * a.Np + n;
      const float v = (n < len) ? fbase[(size_t)n * a.ld + 3 + tid] : 0.f;
      a.x32[row * 256 + tid] = v;
      a.xb[row * 256 + tid] = P::from_f32(v);
      if (tid < 128) a.h128[row * 128 + tid] = P::from_f32(n < len ? bufb[l][tid] : 0.f);
    }
    return;
  }
  ke_layer<128, 256>(a.w[6], a.w[7], &bufb[0][0], 256, &bufa[0][0], 256, true);
  __syncthreads();
  ke_layer<256, 256>(a.w[8], a.w[9], &bufa[0][0], 256, &bufb[0][0], 256, false);
  __syncthreads();
  for (int l = 0; l < KE_LT; ++l) {
    const int n = n0r + l;
    if (n >= a.Np) break;
    const size_t row = (size_t)s * a.Np + n;
    float v = 0.f;
    if (n < len) v = fbase[(size_t)n * a.ld + 3 + tid] + bufb[l][tid];
    a.x32[row * 256 + tid] = v;
    a.xb[row * 256 + tid] = P::from_f32(v);
  }
}

void launch_sg_prepare(int prec, const float* f0, const float* f1, const int* n0, const int* n1, int ld, int normalize,
                       float cx, float cy, float linv, const float* const* w, int B, int cap, int Np, float* x32,
                       uint16_t* xb, int* lens, uint16_t* h128, hipStream_t st) {
  SgPrepArgs a;
  a.f0 = f0; a.f1 = f1; a.n0 = n0; a.n1 = n1; a.ld = ld; a.normalize = normalize; a.cx = cx; a.cy = cy; a.linv = linv;
  for (int i = 0; i < 10; ++i) a.w[i] = w[i];
  a.B = B; a.cap = cap; a.Np = Np; a.x32 = x32; a.xb = xb; a.lens = lens; a.h128 = h128;
  dim3 grid((Np + KE_LT - 1) / KE_LT, 2 * B);
  if (h128) {
    if (prec == 1) hipLaunchKernelGGL((sg_prepare_kernel<PF16, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sg_prepare_kernel<PBF16, true>), grid, dim3(256), 0, st, a);
  } else {
    if (prec == 1) hipLaunchKernelGGL((sg_prepare_kernel<PF16, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sg_prepare_kernel<PBF16, false>), grid, dim3(256), 0, st, a);
  }
}

// =============================================================================== SuperGlue: Sinkhorn
// log_optimal_transport (public SuperGlue; cf. the reference's CPU copy src/super_glue.cpp:369-435).
// Couplings are never materialised: C[i][j] = sim[i][j] inside, alpha on the dustbin row/column.
__device__ __forceinline__ float sg_coupling(const float* __restrict__ sim, int Np, int n0, int n1, float alpha, int i, int j) {
  return (i < n0 && j < n1) ? sim[(size_t)i * Np + j] : alpha;
}

// wave per row: u[i] = log_mu[i] - logsumexp_j(C[i][j] + v[j]), i in [0, n0]
__global__ void sg_sinkhorn_row_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int Lz, float alpha,
                                       float* __restrict__ u, const float* __restrict__ v) {
  const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (i > n0) return;
  const float* S = sim + (size_t)b * Np * Np;
  const float* vb = v + (size_t)b * Lz;
  float mx = -INFINITY;
  for (int j = lane; j <= n1; j += 64) mx = fmaxf(mx, sg_coupling(S, Np, n0, n1, alpha, i, j) + vb[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j <= n1; j += 64) s += expf(sg_coupling(S, Np, n0, n1, alpha, i, j) + vb[j] - mx);
  s = wave_sum(s);
  if (lane == 0) {
    const float norm = -logf((float)(n0 + n1));
    const float log_mu = (i < n0) ? norm : logf((float)n1) + norm;
    u[(size_t)b * Lz + i] = log_mu - (mx + logf(s));
  }
}

// thread per column: v[j] = log_nu[j] - logsumexp_i(C[i][j] + u[i]), j in [0, n1]
__global__ void sg_sinkhorn_col_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int Lz, float alpha,
                                       const float* __restrict__ u, float* __restrict__ v) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (j > n1) return;
  const float* S = sim + (size_t)b * Np * Np;
  const float* ub = u + (size_t)b * Lz;
  float mx = -INFINITY;
  for (int i = 0; i <= n0; ++i) mx = fmaxf(mx, sg_coupling(S, Np, n0, n1, alpha, i, j) + ub[i]);
  float s = 0.f;
  for (int i = 0; i <= n0; ++i) s += expf(sg_coupling(S, Np, n0, n1, alpha, i, j) + ub[i] - mx);
  const float norm = -logf((float)(n0 + n1));
  const float log_nu = (j < n1) ? norm : logf((float)n0) + norm;
  v[(size_t)b * Lz + j] = log_nu - (mx + logf(s));
}

// Z = C + u + v - norm  ->  scores [b][Lz][Lz] (row stride Lz), rows 0..n0, cols 0..n1
__global__ void sg_scores_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int Lz, float alpha,
                                 const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ Z) {
  const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (i > n0 || j > n1) return;
  const float norm = -logf((float)(n0 + n1));
  Z[((size_t)b * Lz + i) * Lz + j] =
      sg_coupling(sim + (size_t)b * Np * Np, Np, n0, n1, alpha, i, j) + u[(size_t)b * Lz + i] + v[(size_t)b * Lz + j] - norm;
}

// The whole of log_optimal_transport in ONE launch: G workgroups per pair (G * pairs ~ 256, so every CU works), each owning a slice of the
// rows in the u half-iteration and a slice of the columns in the v half-iteration; u and v are exchanged through global memory and the
// G workgroups of a pair meet at a monotonic per-pair counter after every half-iteration (agent-scope release / acquire as
// MI355X_MICROARCH.md 'Workgroup dispatch ... inter-workgroup visibility' prescribes: plain stores -> __syncthreads -> lane-0 release
// fence -> asm vmcnt(0) -> relaxed atomic arrive; one relaxed poll loop -> acquire fence -> __syncthreads -> plain loads).
// History: 2 x iters launches per step, each reading the couplings twice for all pairs: ~25 ms per 64 pairs at N = 400; ONE workgroup
// per pair (no inter-workgroup traffic at all): 21 ms — a single workgroup is latency-bound on its 643 KB matrix, and at N = 1024
// with 16 pairs it was 2x SLOWER than the launches.  All G x pairs workgroups are co-resident by construction (grid <= 128 on 256
// CUs); every spin is bounded all the same, a timeout poisons the pair's output with NaN instead of hanging.
template <int MAXC>
__global__ __launch_bounds__(512) void sg_sinkhorn_fused_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int Lz,
                                                               float alpha, int iters, int G, float* __restrict__ ug, float* __restrict__ vg,
                                                               float* __restrict__ Z, unsigned* __restrict__ counters) {
  constexpr int NW = 8, NT = 512;
  __shared__ float u[1040], v[1040];
  __shared__ float pm[NW][64], psum[NW][64];
  __shared__ int s_fail;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const float* S = sim + (size_t)b * Np * Np;
  float* ub = ug + (size_t)b * Lz;
  float* vb = vg + (size_t)b * Lz;
  unsigned* cnt = counters + b * 16;                      // one 64-byte line per pair
  const float norm = -logf((float)(n0 + n1));
  const float lmu_last = logf((float)n1) + norm, lnu_last = logf((float)n0) + norm;
  const int rper = (n0 + 1 + G - 1) / G, r_lo = g * rper, r_hi = min(r_lo + rper, n0 + 1);      // this workgroup's rows
  const int cper = (n1 + 1 + G - 1) / G, c_lo = g * cper, c_hi = min(c_lo + cper, n1 + 1);      // ... and columns
  for (int i = tid; i < 1040; i += NT) { u[i] = 0.f; v[i] = 0.f; }
  if (tid == 0) s_fail = 0;
  __syncthreads();
  unsigned target = 0;
  auto rendezvous = [&]() {                               // all G workgroups of this pair have published their slice
    target += (unsigned)G;
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18)) { s_fail = 1; break; }        // ~0.3 s: never reached unless a partner workgroup is not resident
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  };
  const int nchunk = (n1 + 1 + 63) >> 6;
  for (int it = 0; it < iters && !s_fail; ++it) {
    // ---- u[i] = log_mu[i] - logsumexp_j(C[i][j] + v[j]) for this workgroup's rows; one wave per row, the row in registers
    // (RU rows per wave and pass: the loads of all of them are in flight together — one row at a time left the wave waiting a full
    //  L2 / MALL round trip per row, the kernel was latency-bound at ~2 us per row)
    constexpr int RU = MAXC > 8 ? 2 : 4;
    for (int i0 = r_lo + wv * RU; i0 < r_hi; i0 += NW * RU) {
      float x[RU][MAXC];
      float mx[RU], sm[RU];
#pragma unroll
      for (int q = 0; q < RU; ++q) {
        const int i = i0 + q;
        mx[q] = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          const int j = lane + 64 * k;
          x[q][k] = -INFINITY;
          if (i < r_hi && k < nchunk && j <= n1) x[q][k] = ((i < n0 && j < n1) ? S[(size_t)i * Np + j] : alpha) + v[j];
          mx[q] = fmaxf(mx[q], x[q][k]);
        }
      }
#pragma unroll
      for (int q = 0; q < RU; ++q) mx[q] = wave_max(mx[q]);
#pragma unroll
      for (int q = 0; q < RU; ++q) {
        sm[q] = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
          if (k < nchunk) sm[q] += __expf(x[q][k] - mx[q]);          // exp(-inf) = 0 for the padding slots
      }
#pragma unroll
      for (int q = 0; q < RU; ++q) sm[q] = wave_sum(sm[q]);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < RU; ++q)
          if (i0 + q < r_hi) ub[i0 + q] = ((i0 + q < n0) ? norm : lmu_last) - (mx[q] + logf(sm[q]));
      }
    }
    if (G > 1) rendezvous(); else __syncthreads();
    for (int i = tid; i <= n0; i += NT) u[i] = ub[i];
    __syncthreads();
    // ---- v[j] = log_nu[j] - logsumexp_i(C[i][j] + u[i]) for this workgroup's columns: 64 columns x 8 row slices per pass
    for (int j0 = c_lo; j0 < c_hi; j0 += 64) {
      const int j = j0 + lane;
      float m = -INFINITY, sm = 0.f;
      if (j < c_hi) {
        for (int i0 = wv; i0 <= n0; i0 += NW * 16) {         // this slice's rows i0, i0 + NW, ... in chunks of 16 values
          float x[16];
          float cm = -INFINITY;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = i0 + NW * r;
            x[r] = -INFINITY;
            if (i <= n0) x[r] = ((i < n0 && j < n1) ? S[(size_t)i * Np + j] : alpha) + u[i];
            cm = fmaxf(cm, x[r]);
          }
          const float mn = fmaxf(m, cm);
          float cs = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) cs += __expf(x[r] - mn);
          sm = sm * __expf(m - mn) + cs;                     // m = -inf on the first chunk: exp(-inf) = 0, sm was 0
          m = mn;
        }
      }
      pm[wv][lane] = m;
      psum[wv][lane] = sm;
      __syncthreads();
      if (wv == 0 && j < c_hi) {
        float M = pm[0][lane];
#pragma unroll
        for (int q = 1; q < NW; ++q) M = fmaxf(M, pm[q][lane]);
        float T = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) T += psum[q][lane] * __expf(pm[q][lane] - M);     // empty slices: sum 0, exp(-inf - M) = 0
        vb[j] = ((j < n1) ? norm : lnu_last) - (M + logf(T));
      }
      __syncthreads();
    }
    if (G > 1) rendezvous(); else __syncthreads();
    for (int j = tid; j <= n1; j += NT) v[j] = vb[j];
    __syncthreads();
  }
  // ---- Z = C + u + v - norm for this workgroup's rows (cols 0..n1)
  const float poison = s_fail ? NAN : 0.f;
  for (int i = r_lo + wv; i < r_hi; i += NW)
    for (int j = lane; j <= n1; j += 64)
      Z[((size_t)b * Lz + i) * Lz + j] = ((i < n0 && j < n1) ? S[(size_t)i * Np + j] : alpha) + u[i] + v[j] - norm + poison;
}

// Register-resident form: the couplings of a pair never leave the register files of the G workgroups that share it.  Workgroup g owns
// a slice of the ROWS (wave w the rows r_lo + w, + 8, ...; lane l the columns l, l + 64, ...: RW x MAXC values per lane, 91 at N = 400,
// 153 at N = 1024), loaded once.  The u half-iteration is wave-local (a row is one wave: two wave reductions, u[i] stays in that wave);
// the v half-iteration reduces each column over the wave's own rows in registers, over the 8 waves through LDS, and over the G
// workgroups through ONE exchange of (max, sum) pairs per column and iteration — one rendezvous per iteration, ~6 KB per workgroup,
// instead of re-reading 643 KB of couplings twice.  The launch is COOPERATIVE (the runtime guarantees that all workgroups are
// resident, or refuses the launch and the streaming kernel above runs); spins are bounded all the same.
template <int CTRL>
__device__ __forceinline__ float sk_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sk_wave_max(float v) {
  v = fmaxf(v, sk_dpp<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, sk_dpp<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, sk_dpp<0x141>(v));     // row_half_mirror
  v = fmaxf(v, sk_dpp<0x140>(v));     // row_mirror
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float sk_wave_sum(float v) {
  v += sk_dpp<0xB1>(v);
  v += sk_dpp<0x4E>(v);
  v += sk_dpp<0x141>(v);
  v += sk_dpp<0x140>(v);
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

template <int RW, int MAXC>
__global__ __launch_bounds__(512) void sg_sinkhorn_reg_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int Lz,
                                                             float alpha, int iters, int G, float2* __restrict__ xch,
                                                             float* __restrict__ Z, unsigned* __restrict__ counters) {
  constexpr int NW = 8, NT = 512, NC = 64 * MAXC;
  __shared__ float pm[NW][NC], ps[NW][NC];
  __shared__ float vs[NC];
  __shared__ int s_fail;
  // The G workgroups of a pair should share an XCD (their exchange then stays in one L2): consecutive workgroup ids go round the 8
  // XCDs, so id -> (xcd = id % 8, slot = id / 8) and the pairs are dealt out per XCD.  Only a placement hint: any mapping is correct.
  int wid = blockIdx.x;
  {
    const int per_xcd = gridDim.x >> 3;
    if ((gridDim.x & 7) == 0 && per_xcd % G == 0) wid = (wid & 7) * per_xcd + (wid >> 3);
  }
  const int b = wid / G, g = wid - b * G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const float* S = sim + (size_t)b * Np * Np;
  unsigned* cnt = counters + b * 16;
  const float norm = -logf((float)(n0 + n1));
  const float lmu_last = logf((float)n1) + norm, lnu_last = logf((float)n0) + norm;
  const int rper = (n0 + 1 + G - 1) / G, r_lo = g * rper, r_hi = min(r_lo + rper, n0 + 1);      // rper <= NW * RW (host)
  float c[RW][MAXC], v[MAXC], u[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int i = r_lo + wv + NW * q;
    u[q] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int j = lane + 64 * k;
      c[q][k] = -INFINITY;
      if (i < r_hi && j <= n1) c[q][k] = (i < n0 && j < n1) ? S[(size_t)i * Np + j] : alpha;
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) v[k] = 0.f;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  unsigned target = 0;
  for (int it = 0; it < iters && !s_fail; ++it) {
    // ---- u[i] = log_mu[i] - logsumexp_j(C[i][j] + v[j]): a row is this wave's alone
    float mx[RW], sm[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      mx[q] = c[q][0] + v[0];
#pragma unroll
      for (int k = 1; k < MAXC; ++k) mx[q] = fmaxf(mx[q], c[q][k] + v[k]);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) mx[q] = sk_wave_max(mx[q]);
    // (the sums recompute c + v: without this fence hipcc keeps all RW x MAXC sums of the maximum pass alive and spills)
#pragma unroll
    for (int k = 0; k < MAXC; ++k) asm volatile("" : "+v"(v[k]));
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      const float off = (mx[q] == -INFINITY) ? 0.f : mx[q];           // a row slot beyond r_hi: all -inf, sum 0, u unused
      sm[q] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXC; ++k) sm[q] += __expf(c[q][k] + v[k] - off);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) sm[q] = sk_wave_sum(sm[q]);
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      const int i = r_lo + wv + NW * q;
      u[q] = (i < r_hi) ? ((i < n0) ? norm : lmu_last) - (mx[q] + __logf(sm[q])) : 0.f;
    }
    // ---- column partials over this wave's rows, then over the 8 waves
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      float cm = c[0][k] + u[0];
#pragma unroll
      for (int q = 1; q < RW; ++q) cm = fmaxf(cm, c[q][k] + u[q]);
      const float off = (cm == -INFINITY) ? 0.f : cm;
      float cs = 0.f;
#pragma unroll
      for (int q = 0; q < RW; ++q) cs += __expf(c[q][k] + u[q] - off);
      pm[wv][lane + 64 * k] = cm;
      ps[wv][lane + 64 * k] = cs;
    }
    __syncthreads();
    float2* mine = xch + (((size_t)b * 2 + (it & 1)) * G + g) * Lz;
    for (int j = tid; j <= n1; j += NT) {
      float M = pm[0][j];
#pragma unroll
      for (int w = 1; w < NW; ++w) M = fmaxf(M, pm[w][j]);
      const float off = (M == -INFINITY) ? 0.f : M;
      float T = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) T += ps[w][j] * __expf(pm[w][j] - off);
      if (G > 1) mine[j] = make_float2(M, T);
      else vs[j] = ((j < n1) ? norm : lnu_last) - (M + __logf(T));
    }
    if (G > 1) {
      // all G workgroups of the pair have published their partials (release / acquire at agent scope, see the kernel above)
      target += (unsigned)G;
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > (1 << 20)) { s_fail = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      const float2* all = xch + ((size_t)b * 2 + (it & 1)) * G * Lz;
      for (int j = tid; j <= n1; j += NT) {
        float M = -INFINITY, T = 0.f;
        for (int h = 0; h < G; ++h) {
          const float2 p = all[(size_t)h * Lz + j];
          const float Mn = fmaxf(M, p.x);
          const float off = (Mn == -INFINITY) ? 0.f : Mn;
          T = T * __expf(M - off) + p.y * __expf(p.x - off);       // first partial: T = 0, exp(-inf) = 0
          M = Mn;
        }
        vs[j] = ((j < n1) ? norm : lnu_last) - (M + __logf(T));
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int j = lane + 64 * k;
      v[k] = (j <= n1) ? vs[j] : 0.f;
    }
  }
  const float poison = s_fail ? NAN : 0.f;
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int i = r_lo + wv + NW * q;
    if (i < r_hi) {
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int j = lane + 64 * k;
        if (j <= n1) Z[((size_t)b * Lz + i) * Lz + j] = c[q][k] + u[q] + v[k] - norm + poison;
      }
    }
  }
}

template <int RW, int MAXC>
static bool try_sinkhorn_reg(const float* sim, const int* lens, int B, int Np, int Lz, float alpha, int iters, float2* xch, float* Z,
                             unsigned* counters, hipStream_t st) {
  int G = (Np + 1 + 8 * RW - 1) / (8 * RW);
  if (G > 16 || Np + 1 > 64 * MAXC || Lz < 64 * MAXC) return false;
  static int max_wgs = -1;                                   // co-resident workgroups of this instantiation on this device
  if (max_wgs < 0) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sg_sinkhorn_reg_kernel<RW, MAXC>, 512, 0) != hipSuccess)
      max_wgs = 0;
    else
      max_wgs = per_cu * prop.multiProcessorCount;
  }
  if (B * G > max_wgs) return false;
  (void)hipMemsetAsync(counters, 0, (size_t)B * 64, st);
  void* args[] = {(void*)&sim, (void*)&lens, (void*)&Np, (void*)&Lz, (void*)&alpha, (void*)&iters, (void*)&G, (void*)&xch, (void*)&Z, (void*)&counters};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&sg_sinkhorn_reg_kernel<RW, MAXC>), dim3(B * G), dim3(512), args, 0, st);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

void launch_sg_sinkhorn(const float* sim, const int* lens, int B, int Np, int Lz, float alpha, int iters, float* u, float* v,
                        float* Z, unsigned* counters, float* xch, hipStream_t st) {
  static const bool unfused = getenv("AIRFE_SINKHORN_UNFUSED") && atoi(getenv("AIRFE_SINKHORN_UNFUSED")) != 0;   // A/B runs
  static const bool noreg = getenv("AIRFE_SINKHORN_STREAM") && atoi(getenv("AIRFE_SINKHORN_STREAM")) != 0;
  if (!unfused && !noreg && counters && xch) {
    float2* x2 = reinterpret_cast<float2*>(xch);
    if (Np + 1 <= 448 && try_sinkhorn_reg<13, 7>(sim, lens, B, Np, Lz, alpha, iters, x2, Z, counters, st)) return;
    if (Np + 1 <= 1088 && try_sinkhorn_reg<9, 17>(sim, lens, B, Np, Lz, alpha, iters, x2, Z, counters, st)) return;
  }
  if (!unfused && Np <= 1024 && counters && B <= 128) {
    int G = 1;
    while (G < 16 && B * G * 2 <= 128) G *= 2;         // workgroups per pair: at ~140 registers ONE 8-wave workgroup fits a CU, so the grid stays
                                                       // at <= 128 workgroups — every one is resident even if half the CUs are busy elsewhere
    (void)hipMemsetAsync(counters, 0, (size_t)B * 64, st);
    if (Np + 1 <= 512) hipLaunchKernelGGL((sg_sinkhorn_fused_kernel<8>), dim3(B * G), dim3(512), 0, st, sim, lens, Np, Lz, alpha, iters, G, u, v, Z, counters);
    else hipLaunchKernelGGL((sg_sinkhorn_fused_kernel<17>), dim3(B * G), dim3(512), 0, st, sim, lens, Np, Lz, alpha, iters, G, u, v, Z, counters);
    return;
  }
  (void)hipMemsetAsync(u, 0, (size_t)B * Lz * 4, st);
  (void)hipMemsetAsync(v, 0, (size_t)B * Lz * 4, st);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(sg_sinkhorn_row_kernel, dim3((Np + 1 + 3) / 4, B), dim3(256), 0, st, sim, lens, Np, Lz, alpha, u, v);
    hipLaunchKernelGGL(sg_sinkhorn_col_kernel, dim3((Np + 1 + 63) / 64, B), dim3(64), 0, st, sim, lens, Np, Lz, alpha, u, v);
  }
  hipLaunchKernelGGL(sg_scores_kernel, dim3((Np + 1 + 63) / 64, Np + 1, B), dim3(64), 0, st, sim, lens, Np, Lz, alpha, u, v, Z);
}

// =============================================================================== SuperGlue: decode
// decode (src/super_glue.cpp:339-367) on Z: inner block rows < n0, cols < n1; strict '<' => first maximum wins.
__global__ void sg_rowmax_kernel(const float* __restrict__ Z, const int* __restrict__ lens, int Lz, int* __restrict__ idx0,
                                 float* __restrict__ max0) {
  const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (i >= n0) return;
  const float* r = Z + ((size_t)b * Lz + i) * Lz;
  float best = -FLT_MAX;                       // super_glue.cpp:261: strict '<' from -FLT_MAX, index 0 when nothing exceeds it
  int bj = 0x7FFFFFFF;
  for (int j = lane; j < n1; j += 64) { const float v = r[j]; if (v > best) { best = v; bj = j; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  if (lane == 0) { idx0[(size_t)b * Lz + i] = (bj == 0x7FFFFFFF) ? 0 : bj; max0[(size_t)b * Lz + i] = best; }
}

// 64 columns x 8 contiguous row slices per workgroup; the slices are combined in ascending row order with the same strict '>'
// (the first maximum wins, as in the reference's single ascending loop)
__global__ __launch_bounds__(512) void sg_colmax_kernel(const float* __restrict__ Z, const int* __restrict__ lens, int Lz, int* __restrict__ idx1) {
  __shared__ float sb[8][64];
  __shared__ int si[8][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6, j = blockIdx.x * 64 + lane;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const int per = (n0 + 7) / 8, lo = q * per, hi = min(lo + per, n0);
  float best = -FLT_MAX;
  int bi = 0;
  if (j < n1) {
    const float* c = Z + (size_t)b * Lz * Lz + j;
    for (int i = lo; i < hi; ++i) { const float v = c[(size_t)i * Lz]; if (v > best) { best = v; bi = i; } }
  }
  sb[q][lane] = best;
  si[q][lane] = bi;
  __syncthreads();
  if (q == 0 && j < n1) {
#pragma unroll
    for (int r = 1; r < 8; ++r)
      if (sb[r][lane] > best) { best = sb[r][lane]; bi = si[r][lane]; }
    idx1[(size_t)b * Lz + j] = bi;
  }
}

__global__ __launch_bounds__(1024) void sg_decode_kernel(const int* __restrict__ lens, int Lz, const int* __restrict__ idx0,
                                                         const float* __restrict__ max0, const int* __restrict__ idx1,
                                                         float thr, int32_t* __restrict__ out0, int32_t* __restrict__ out1,
                                                         float* __restrict__ ms0, float* __restrict__ ms1) {
  __shared__ float s_ms0[1024];
  __shared__ unsigned char s_valid0[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const int* i0 = idx0 + (size_t)b * Lz;
  const int* i1 = idx1 + (size_t)b * Lz;
  if (tid < n0) {
    const bool mutual = i1[i0[tid]] == tid;
    const float m = mutual ? expf(max0[(size_t)b * Lz + tid]) : 0.f;
    const bool valid = mutual && m > thr;
    s_ms0[tid] = m;
    s_valid0[tid] = valid;
    ms0[(size_t)b * Lz + tid] = m;
    out0[(size_t)b * Lz + tid] = valid ? i0[tid] : -1;
  }
  __syncthreads();
  if (tid < n1) {
    const int r = i1[tid];
    const bool mutual = i0[r] == tid;
    ms1[(size_t)b * Lz + tid] = mutual ? s_ms0[r] : 0.f;
    out1[(size_t)b * Lz + tid] = (mutual && s_valid0[r]) ? r : -1;
  }
}

void launch_sg_decode(const float* Z, const int* lens, int B, int Np, int Lz, float thr, int* idx0, float* max0, int* idx1,
                      int32_t* out0, int32_t* out1, float* ms0, float* ms1, hipStream_t st) {
  hipLaunchKernelGGL(sg_rowmax_kernel, dim3((Np + 3) / 4, B), dim3(256), 0, st, Z, lens, Lz, idx0, max0);
  hipLaunchKernelGGL(sg_colmax_kernel, dim3((Np + 63) / 64, B), dim3(512), 0, st, Z, lens, Lz, idx1);
  hipLaunchKernelGGL(sg_decode_kernel, dim3(B), dim3(1024), 0, st, lens, Lz, idx0, max0, idx1, thr, out0, out1, ms0, ms1);
}

// =============================================================================== BoW quantisation
// TemplatedVocabulary::transform(feature, word_id, weight) (3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1313-1352) for every
// feature of a frame, as Database::FrameToBow calls it (src/bow/database.cc:57-89): descend the vocabulary tree, at every node
// taking the child whose 256-d descriptor is nearest in squared L2 distance (FSuperpoint::distance, src/bow/FSuperpoint.cc:45-49),
// FIRST minimum on ties (strict '<'), until a leaf; emit the leaf's word id and weight.  One wave per feature: the feature sits in
// registers (4 floats per lane), every candidate child is one coalesced 1 KiB row read.
__global__ __launch_bounds__(256) void bow_transform_kernel(const float* __restrict__ feat, int ld, int off, int N,
                                                            const float* __restrict__ node_desc, const int* __restrict__ first_child,
                                                            const int* __restrict__ n_children, const int* __restrict__ word_id,
                                                            const float* __restrict__ weight, unsigned* __restrict__ out_word,
                                                            float* __restrict__ out_weight) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const float4 f = *reinterpret_cast<const float4*>(feat + (size_t)i * ld + off + lane * 4);
  int node = 0;
  for (int nc = n_children[0]; nc > 0; nc = n_children[node]) {
    const int c0 = first_child[node];
    float best = INFINITY;
    int bi = c0;
    for (int c = 0; c < nc; ++c) {
      const float4 d = *reinterpret_cast<const float4*>(node_desc + (size_t)(c0 + c) * 256 + lane * 4);
      const float dx = f.x - d.x, dy = f.y - d.y, dz = f.z - d.z, dw = f.w - d.w;
      const float dist = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw);
      if (dist < best) { best = dist; bi = c0 + c; }
    }
    node = bi;
  }
  if (lane == 0) {
    const float w = weight[node];
    out_word[i] = w > 0.f ? (unsigned)word_id[node] : 0xFFFFFFFFu;      // database.cc:77-83: stopped words -> UINT_MAX
    out_weight[i] = w;
  }
}

void launch_bow_transform(const float* feat, int ld, int off, int N, const float* node_desc, const int* first_child,
                          const int* n_children, const int* word_id, const float* weight, unsigned* out_word, float* out_weight,
                          hipStream_t st) {
  if (N < 1) return;
  hipLaunchKernelGGL(bow_transform_kernel, dim3((N + 3) / 4), dim3(256), 0, st, feat, ld, off, N, node_desc, first_child, n_children,
                     word_id, weight, out_word, out_weight);
}

// =============================================================================== point <-> line association
// AssignPointsToLines (src/line_processor.cc:68-120): for every line the points lying on it
// (bounding box +-3 px, point-line distance <= 3 px, endpoint / projection test), as a CSR list in ascending point
// index (= the iteration order of the reference's std::map<int, double>).  All arithmetic in double, every product
// rounded separately (no FMA contraction) like the oracle's numpy restatement; the distance is narrowed to float exactly
// where the reference narrows it.
// a product that must be rounded on its own: the empty asm makes it opaque to hipcc, which otherwise fuses a * b + c into
// an FMA even through __dmul_rn / __dadd_rn and `#pragma clang fp contract(off)` (seen: 1-ulp differences on points that
// lie exactly on a line, where the numerator cancels catastrophically)
__device__ __forceinline__ double rounded_mul(double a, double b) {
  double m = a * b;
  asm volatile("" : "+v"(m));
  return m;
}
__device__ __forceinline__ bool point_on_line(double lx1, double ly1, double lx2, double ly2, double px, double py, float& dist) {
  const double A = ly2 - ly1, B = lx1 - lx2;
  const double C = rounded_mul(lx2, ly1) - rounded_mul(lx1, ly2);
  const double D = __dsqrt_rn(rounded_mul(A, A) + rounded_mul(B, B));
  double min_lx = lx1, max_lx = lx2, min_ly = ly1, max_ly = ly2;
  if (lx1 > lx2) { min_lx = lx2; max_lx = lx1; }
  if (ly1 > ly2) { min_ly = ly2; max_ly = ly1; }
  if (px < min_lx - 3 || px > max_lx + 3 || py < min_ly - 3 || py > max_ly + 3) return false;
  const float pl = (float)__ddiv_rn(fabs((rounded_mul(A, px) + rounded_mul(B, py)) + C), D);
  if (pl > 3) return false;
  const double dx1 = lx1 - px, dy1 = ly1 - py, dx2 = lx2 - px, dy2 = ly2 - py;
  const double side1 = rounded_mul(dx1, dx1) + rounded_mul(dy1, dy1);
  const double side2 = rounded_mul(dx2, dx2) + rounded_mul(dy2, dy2);
  const double line_side = rounded_mul(D, D);
  dist = pl;
  return side1 <= 9 || side2 <= 9 || ((side1 < line_side + side2) && (side2 < line_side + side1));
}

// one wave per line; pass 0 counts, pass 1 writes at row_ptr[line] (ballot + prefix keeps ascending point order)
template <bool WRITE>
__global__ __launch_bounds__(256) void pl_assign_kernel(const double* __restrict__ lines, int L, const float* __restrict__ feat,
                                                        int N, int* __restrict__ counts, const int* __restrict__ row_ptr,
                                                        int* __restrict__ pt_idx, double* __restrict__ pt_dist, int cap) {
  const int line = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (line >= L) return;
  const double lx1 = lines[line * 4 + 0], ly1 = lines[line * 4 + 1], lx2 = lines[line * 4 + 2], ly2 = lines[line * 4 + 3];
  int cnt = 0;
  const int base = WRITE ? row_ptr[line] : 0;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    float d = 0.f;
    bool hit = false;
    if (j < N) hit = point_on_line(lx1, ly1, lx2, ly2, (double)feat[(size_t)j * 259 + 1], (double)feat[(size_t)j * 259 + 2], d);
    const unsigned long long m = __ballot(hit);
    if (WRITE && hit) {
      const int pos = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap) { pt_idx[pos] = j; pt_dist[pos] = (double)d; }
    }
    cnt += __popcll(m);
  }
  if (!WRITE && lane == 0) counts[line] = cnt;
}

// exclusive scan of counts[L] -> row_ptr[L+1] (L is a few hundred: one workgroup, serial per 256-chunk carry)
__global__ __launch_bounds__(256) void pl_scan_kernel(const int* __restrict__ counts, int L, int* __restrict__ row_ptr) {
  __shared__ int buf[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < L; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const int v = i < L ? counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < L) row_ptr[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += buf[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) row_ptr[L] = carry;
}

void launch_assign_points_to_lines(const double* lines, int L, const float* feat, int N, int* counts, int* row_ptr, int* pt_idx,
                                   double* pt_dist, int cap, hipStream_t st) {
  if (L <= 0) return;
  const dim3 grid((L + 3) / 4);
  hipLaunchKernelGGL(pl_assign_kernel<false>, grid, dim3(256), 0, st, lines, L, feat, N, counts, row_ptr, pt_idx, pt_dist, cap);
  hipLaunchKernelGGL(pl_scan_kernel, dim3(1), dim3(256), 0, st, counts, L, row_ptr);
  hipLaunchKernelGGL(pl_assign_kernel<true>, grid, dim3(256), 0, st, lines, L, feat, N, counts, row_ptr, pt_idx, pt_dist, cap);
}

// =============================================================================== MatchLines (src/line_processor.cc:122-172)
// The voting matrix M[l0][l1] = number of point matches (q, t) with q on line l0 of frame 0 and t on line l1 of frame 1 is an
// integer "GEMM" over the matches: one bit per (line, match) says whether the match's point lies on the line, and M is the
// popcount of the AND of two bit rows.  All index work: exact.
__global__ __launch_bounds__(256) void ml_bits_kernel(const int* __restrict__ row_ptr, const int* __restrict__ pt_idx, const int* __restrict__ matches,
                                                      int side, int nmatch, int W, unsigned* __restrict__ bits) {
  const int l = blockIdx.x, b = row_ptr[l], e = row_ptr[l + 1];
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    unsigned word = 0;
    for (int k = 0; k < 32; ++k) {
      const int m = w * 32 + k;
      if (m >= nmatch) break;
      const int p = matches[2 * m + side];
      for (int r = b; r < e; ++r)
        if (pt_idx[r] == p) { word |= 1u << k; break; }        // a std::map key occurs once per line
    }
    bits[(size_t)l * W + w] = word;
  }
}

// first maximum (value, index) over a block: larger value wins, ties go to the smaller index (Eigen maxCoeff visits in order and
// replaces on strict >)
__device__ __forceinline__ void ml_first_max(int& v, int& i, int* sv, int* si) {
  const int t = threadIdx.x;
  sv[t] = v; si[t] = i;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (t < o) {
      const int v2 = sv[t + o], i2 = si[t + o];
      if (v2 > sv[t] || (v2 == sv[t] && i2 < si[t])) { sv[t] = v2; si[t] = i2; }
    }
    __syncthreads();
  }
  v = sv[0]; i = si[0];
  __syncthreads();
}

__global__ __launch_bounds__(256) void ml_vote_rowmax_kernel(const unsigned* __restrict__ bits0, const unsigned* __restrict__ bits1, int L1, int W,
                                                             int* __restrict__ vote, int* __restrict__ row_loc, int* __restrict__ line_matches) {
  __shared__ int sv[256], si[256];
  const int l0 = blockIdx.x;
  const unsigned* r0 = bits0 + (size_t)l0 * W;
  int bv = -1, bi = 0x7fffffff;
  for (int l1 = threadIdx.x; l1 < L1; l1 += blockDim.x) {
    const unsigned* r1 = bits1 + (size_t)l1 * W;
    int v = 0;
    for (int w = 0; w < W; ++w) v += __popc(r0[w] & r1[w]);
    vote[(size_t)l0 * L1 + l1] = v;
    if (v > bv) { bv = v; bi = l1; }                             // l1 ascends within a thread: strict > keeps the first
  }
  ml_first_max(bv, bi, sv, si);
  if (threadIdx.x == 0) { row_loc[l0] = bi; line_matches[l0] = -1; }
}

__global__ __launch_bounds__(256) void ml_colmax_kernel(const int* __restrict__ vote, const int* __restrict__ row_loc, const int* __restrict__ row_ptr0,
                                                        const int* __restrict__ row_ptr1, int L0, int L1, int* __restrict__ line_matches) {
  __shared__ int sv[256], si[256];
  const int j = blockIdx.x;
  int bv = -1, bi = 0x7fffffff;
  for (int i = threadIdx.x; i < L0; i += blockDim.x) {
    const int v = vote[(size_t)i * L1 + j];
    if (v > bv) { bv = v; bi = i; }
  }
  ml_first_max(bv, bi, sv, si);
  if (threadIdx.x == 0) {
    if (bv < 2 || row_loc[bi] != j) return;                      // :171
    const int n0 = row_ptr0[bi + 1] - row_ptr0[bi], n1 = row_ptr1[j + 1] - row_ptr1[j];
    const float score = __fdiv_rn((float)(bv * bv), (float)min(n0, n1));        // :174 float / size_t -> float division
    if ((double)score < 0.8) return;                             // :175
    line_matches[bi] = j;                                        // distinct j cannot name the same row: row_loc[bi] == j
  }
}

void launch_match_lines(const int* row_ptr0, const int* pt_idx0, int L0, const int* row_ptr1, const int* pt_idx1, int L1, const int* matches,
                        int nmatch, unsigned* bits0, unsigned* bits1, int* vote, int* row_loc, int* line_matches, hipStream_t st) {
  const int W = (nmatch + 31) / 32 > 0 ? (nmatch + 31) / 32 : 1;
  hipLaunchKernelGGL(ml_bits_kernel, dim3(L0), dim3(256), 0, st, row_ptr0, pt_idx0, matches, 0, nmatch, W, bits0);
  hipLaunchKernelGGL(ml_bits_kernel, dim3(L1), dim3(256), 0, st, row_ptr1, pt_idx1, matches, 1, nmatch, W, bits1);
  hipLaunchKernelGGL(ml_vote_rowmax_kernel, dim3(L0), dim3(256), 0, st, bits0, bits1, L1, W, vote, row_loc, line_matches);
  hipLaunchKernelGGL(ml_colmax_kernel, dim3(L1), dim3(256), 0, st, vote, row_loc, row_ptr0, row_ptr1, L0, L1, line_matches);
}

}  // namespace airfe
