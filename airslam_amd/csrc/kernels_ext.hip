// airfe — PLNet line path (wireframe_matcher, stage-1 LOI head, line/junction filter) and the SuperGlue-specific
// pieces (keypoint encoder, log-domain Sinkhorn, decode).  Small, irregular, fp32: VALU + LDS, no MFMA.
#include <float.h>
#include <algorithm>
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

// head4 / prop4 [B][line_cap][4] (nullable together): per unique line the two junctions' coordinates (max, min: = stage 1's lines_adjusted) and the first proposal's four
// numbers — what stage 1 otherwise fetches through pairs -> juncs and rep -> keep -> lines_pred, three dependent loads deep at the head of every one of its tiles.
constexpr int WF_RC = 8;                 // rounds of 64 per wave kept in registers: <= 16 * 8 * 64 = 8192 kept proposals take the short path
__global__ __launch_bounds__(1024) void wireframe_kernel(const float* __restrict__ imin, const float* __restrict__ imax, int jn, int* table,
                                                         const int* __restrict__ keep, int* __restrict__ pairs, int* __restrict__ rep,
                                                         int cap, int line_cap, int* __restrict__ counts, const float* __restrict__ juncs,
                                                         const float* __restrict__ lines_pred, float* __restrict__ head4, float* __restrict__ prop4,
                                                         size_t stage_stride) {
  __shared__ unsigned wcnt[16];
  {
    const size_t img = blockIdx.y;
    imin += img * stage_stride; imax += img * stage_stride;
    table += img * jn * jn; keep += img * cap; pairs += img * line_cap * 2; rep += img * line_cap; counts += img * LINE_CNT_LD;
    if (head4) { juncs += img * stage_stride; lines_pred += img * stage_stride; head4 += img * line_cap * 4; prop4 += img * line_cap * 4; }
  }
  const int* wg_counts = counts + 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned m1 = 0;
  for (int w = 0; w < WF_WGS; ++w) m1 += (unsigned)wg_counts[w];
  m1 = min(m1, (unsigned)cap);
  // unique line `pos` = proposal keep[k] = i between junctions a < b
  auto emit = [&](unsigned pos, int k, int i, int a, int b) {
    if (pos >= (unsigned)line_cap) return;
    rep[pos] = k;
    pairs[pos * 2] = b;           // (max, min): plnet.cpp:301
    pairs[pos * 2 + 1] = a;
    if (head4) {
      const float2 jb = *reinterpret_cast<const float2*>(juncs + b * 2), ja = *reinterpret_cast<const float2*>(juncs + a * 2);
      *reinterpret_cast<float4*>(head4 + (size_t)pos * 4) = make_float4(jb.x, jb.y, ja.x, ja.y);
      *reinterpret_cast<float4*>(prop4 + (size_t)pos * 4) = *reinterpret_cast<const float4*>(lines_pred + (size_t)i * 4);
    }
  };
  // every wave owns a contiguous 16th of keep[0 .. m1) and walks it 64 at a time (first-seen order = keep order: positions by ballot + popcount)
  const int seg2 = ((int)m1 + 15) / 16, t_lo = wv * seg2, t_hi = min(t_lo + seg2, (int)m1);
  if (m1 <= 16u * WF_RC * 64u) {
    // the short path: a lane's proposals stay in registers through all four phases — keep -> (imin, imax) is fetched once instead of four times, and the rounds of
    // a phase issue together (the kernel is one workgroup per image: its time is its chain of dependent loads)
    int ci[WF_RC], ca[WF_RC], cb[WF_RC];
    unsigned ok = 0, first = 0;
#pragma unroll
    for (int r = 0; r < WF_RC; ++r) {
      const int k = t_lo + r * 64 + lane;
      ci[r] = k < t_hi ? keep[k] : -1;
    }
#pragma unroll
    for (int r = 0; r < WF_RC; ++r) {
      ca[r] = ci[r] >= 0 ? (int)imin[ci[r]] : -1;
      cb[r] = ci[r] >= 0 ? (int)imax[ci[r]] : -1;
    }
#pragma unroll
    for (int r = 0; r < WF_RC; ++r)
      if (ca[r] >= 0 && ca[r] < jn && cb[r] >= 0 && cb[r] < jn) {
        ok |= 1u << r;
        atomicMin(&table[ca[r] * jn + cb[r]], t_lo + r * 64 + lane);
      }
    __syncthreads();
    unsigned wc2 = 0;
#pragma unroll
    for (int r = 0; r < WF_RC; ++r) {
      const bool f = ((ok >> r) & 1u) && __hip_atomic_load(&table[ca[r] * jn + cb[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == t_lo + r * 64 + lane;
      first |= (unsigned)f << r;
      wc2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(f));
    }
    if (lane == 0) wcnt[wv] = wc2;
    __syncthreads();                    // (also: every read of the table is done)
    unsigned off2 = 0, m2 = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wv) off2 += wcnt[w];
      m2 += wcnt[w];
    }
#pragma unroll
    for (int r = 0; r < WF_RC; ++r) {
      const bool f = (first >> r) & 1u;
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(f);
      if (f) emit(off2 + __builtin_popcountll(mk & ((1ull << lane) - 1ull)), t_lo + r * 64 + lane, ci[r], ca[r], cb[r]);
      off2 += __builtin_popcountll(mk);
    }
#pragma unroll
    for (int r = 0; r < WF_RC; ++r)     // leave the table clean for the next call
      if ((ok >> r) & 1u) table[ca[r] * jn + cb[r]] = 0x7FFFFFFF;
    if (tid == 0) { counts[0] = (int)m1; counts[1] = (int)min(m2, (unsigned)line_cap); }
    return;
  }
  for (unsigned k = tid; k < m1; k += 1024) {
    const int i = keep[k];
    const int a = (int)imin[i], b = (int)imax[i];
    if (a >= 0 && a < jn && b >= 0 && b < jn) atomicMin(&table[a * jn + b], (int)k);
  }
  __syncthreads();
  // the first proposal of every unique pair, in keep order
  auto first_of_pair = [&](int k, int& i, int& a, int& b) {
    i = keep[k];
    a = (int)imin[i]; b = (int)imax[i];
    return (a >= 0 && a < jn && b >= 0 && b < jn) && __hip_atomic_load(&table[a * jn + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k;
  };
  unsigned wc2 = 0;
  for (int base = t_lo; base < t_hi; base += 64) {
    const int k = base + lane;
    int i, a, b;
    wc2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(k < t_hi && first_of_pair(k, i, a, b)));
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
    int i = 0, a = 0, b = 0;
    const bool f = k < t_hi && first_of_pair(k, i, a, b);
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(f);
    if (f) emit(off2 + __builtin_popcountll(mk & ((1ull << lane) - 1ull)), k, i, a, b);
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
                      int* pairs, int* rep, int cap, int line_cap, int* counts, bool counted, const float* juncs, const float* lines_pred, float* head4,
                      float* prop4, int B, size_t stage_stride, hipStream_t st) {
  if (!counted) hipLaunchKernelGGL(wf_count_kernel, dim3(WF_WGS, B), dim3(256), 0, st, iskeep, n, counts, stage_stride);
  hipLaunchKernelGGL(wf_emit_kernel, dim3(WF_WGS, B), dim3(256), 0, st, iskeep, n, counts, keep, cap, stage_stride);
  hipLaunchKernelGGL(wireframe_kernel, dim3(1, B), dim3(1024), 0, st, imin, imax, jn, table, keep, pairs, rep, cap, line_cap, counts, juncs, lines_pred,
                     head4 && prop4 ? head4 : nullptr, prop4, stage_stride);
}

// =============================================================================== stage-1 LOI head
// plnet_s1.onnx restated (oracle/ref_nets.py::plnet_s1_forward), fp32 throughout.
// Bilinear sampling as the stage-1 graph does it: the four tap positions + the point, then ONE expression for the value —
// every sampler of this file goes through bil_eval, so a value does not depend on which layout it was read from.
struct BilTap { int i00, i10, i01, i11; float px, py, x0, y0, x1, y1; };
__device__ __forceinline__ BilTap bil_setup(int H, int W, float x, float y) {
  BilTap t;
  t.px = x - 0.5f; t.py = y - 0.5f;
  t.x0 = fminf(fmaxf(floorf(t.px), 0.f), (float)(W - 1)); t.y0 = fminf(fmaxf(floorf(t.py), 0.f), (float)(H - 1));
  t.x1 = fminf(fmaxf(t.x0 + 1.f, 0.f), (float)(W - 1)); t.y1 = fminf(fmaxf(t.y0 + 1.f, 0.f), (float)(H - 1));
  const int x0i = (int)t.x0, y0i = (int)t.y0, x1i = (int)t.x1, y1i = (int)t.y1;
  t.i00 = y0i * W + x0i; t.i10 = y1i * W + x0i; t.i01 = y0i * W + x1i; t.i11 = y1i * W + x1i;
  return t;
}
__device__ __forceinline__ float bil_eval(float f00, float f10, float f01, float f11, const BilTap& t) {
  return f00 * (t.y1 - t.py) * (t.x1 - t.px) + f10 * (t.py - t.y0) * (t.x1 - t.px) + f01 * (t.y1 - t.py) * (t.px - t.x0) +
         f11 * (t.py - t.y0) * (t.px - t.x0);
}
// A feature plane is addressed as f[(y W + x) ps]: ps = 1 for the contract's CHW planes, the row pitch of the head GEMM's output when a
// feature is sampled where that GEMM left it.
__device__ __forceinline__ float bil_plane(const float* __restrict__ f, int H, int W, int ps, float x, float y) {
  const BilTap t = bil_setup(H, W, x, y);
  return bil_eval(f[t.i00 * ps], f[t.i10 * ps], f[t.i01 * ps], f[t.i11 * ps], t);
}

// One workgroup = 4 waves = one tile of 32 lines; wave w owns output features [32 w, 32 w + 32) of every layer, and the contractions run
// on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate — the f32 vector rate, MI355X_MICROARCH.md 'Matrix
// cores'): a lane supplies ONE weight (feature lane % 32, k = 2 step + lane / 32, straight from the transposed [K][128] table: 128-byte
// runs) and ONE activation (line lane % 32, same k, a conflict-free LDS read) per 2048 multiply-adds.  History, per 128 images of ~1070
// candidate lines: thread-per-feature scalar fma, 8 lines per workgroup 1.17 ms (9 GB of weight re-reads from L2); the same with 16 lines
// per thread on packed fma 1.03 ms — a group's activations were BROADCAST LDS reads (4 ds_read_b128 per k and wave: 32 LDS cycles per
// 32 VALU cycles, and the LDS pipe is shared by the CU's four SIMDs).
constexpr int S1_LT = 32;               // lines per workgroup
constexpr int S1_LP = 33;               // row pitch (floats) of the [k][line] tiles: the sampling threads write a column (stride 33: no conflict)

__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// acc[r] of lane l = output (feature f0 + 8 (r / 4) + 4 (l / 32) + r % 4, line l % 32)
// Weights: S1_NB buffers of S1_U registers in rotation, NO copies (a copy out of a buffer waits for the buffer's loads): trip i multiplies
// with buffer i % NB and then refills it for trip i + NB, so a fetch has NB - 1 trips = 24 MFMAs (x 2 waves per SIMD: ~3000 cycles) to
// arrive.  Per-phase timers (tools/s1_timing.py) showed ~1 us per dependent global access in this kernel and the MFMA loops running at a
// third of the matrix rate with one trip of cover.
#ifndef S1_SB
#define S1_SB 4                      // thin / aux (line, block, point) items of a thread in flight together
#endif
#ifndef S1_NBUF
#define S1_NBUF 4
#endif
constexpr int S1_U = 8, S1_NB = S1_NBUF;
static_assert(2 * S1_U * (2 * S1_NB - 1) <= S1_WPAD, "the weight tables' padding must cover the prefetch past the last row");
// Weights come through BUFFER loads: a resource descriptor in four scalar registers that never change, a 32-bit lane offset, the row
// as part of that offset.  Two other forms were tried first.  (1) Plain pointer loads: hipcc materialises a 64-bit vector address per
// (row, lane) and hoists all ~90 of them out of the tile loop: 430-640 bytes of scratch per lane.  (2) `global_load_dword v, v_off, s[base]`
// spelled as asm with a freshly computed scalar base per group of loads and hand-placed `s_waitcnt vmcnt(24)`: fast, parity-green — and
// WRONG under load: tools/experiments/plnet_determinism.py (the keyframe step at the bench size, outputs of 60 runs against the first)
// found a different line set in 1 % of the runs alone and in 30-45 % with the matcher running beside it; the same loads with
// `vmcnt(0)` before every trip: 0 of 60, compiler-managed loads: 0 of 380.  The scalar base registers were rewritten while earlier loads
// that name them were still queued — no hazard hipcc's own code would run into (it keeps such registers apart), and none it can see
// inside an asm block.  Lesson kept: no hand-scheduled memory operations whose operands live in registers that are recycled.
template <int K>
struct S1Wt {                              // the transposed weight table of one layer as a buffer resource
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit S1Wt(const float* wt)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, (K + S1_WPAD) * 128 * 4, 0x00020000)) {}
  // lane_off: this lane's byte offset inside a row pair (a vector register that never changes); row_off: the row's byte offset, uniform
  // (the scalar offset operand: kept out of the vector registers, where hipcc would hoist one copy per distinct row out of the tile loop)
  __device__ __forceinline__ float ld(unsigned lane_off, unsigned row_off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0));
  }
};

template <int K, bool PRESET = false>
__device__ __forceinline__ void s1_dense(const float* __restrict__ wt /*[K + S1_WPAD][128]*/, const float* bias /*LDS*/,
                                         const float* xin /*LDS [K][S1_LP]*/, f32x16& acc, int f0, int lane) {
  constexpr int U = S1_U, NB = S1_NB, TRIPS = K / (2 * U);
  static_assert(K % (2 * U) == 0 && U == 8, "K must be a multiple of 16");
  const int i = lane & 31, kk = lane >> 5;
  const S1Wt<K> W(wt);
  const unsigned lb = (unsigned)(kk * 128 + f0 + i) * 4u;
  const float* xp = xin + kk * S1_LP + i;
  if constexpr (!PRESET) {                 // (PRESET: the caller has put bias + the junction terms into acc)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[f0 + 8 * (r >> 2) + 4 * kk + (r & 3)];
  }
  float wb[NB][U];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int u = 0; u < U; ++u) wb[b][u] = W.ld(lb, (unsigned)(b * 2 * U + 2 * u) * 512u);
#pragma unroll 1
  for (int t0 = 0; t0 < TRIPS; t0 += NB) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k0 = (t0 + b) * 2 * U;
      if (k0 < K) {                         // (uniform; TRIPS need not be a multiple of NB)
        float xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xb[u] = xp[(k0 + 2 * u) * S1_LP];
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma_32x32x2(wb[b][u], xb[u], acc);
        __builtin_amdgcn_sched_barrier(0);  // the refill stays BEHIND the MFMAs that read the buffer (hoisted, it needs a second set of registers)
        const unsigned nrow = (unsigned)__builtin_amdgcn_readfirstlane((k0 + NB * 2 * U) * 512);
#pragma unroll
        for (int u = 0; u < U; ++u) wb[b][u] = W.ld(lb, nrow + (unsigned)u * 1024u);     // (past the end: the table's S1_WPAD zero rows, never used)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

#ifdef S1_TIMING
__device__ unsigned long long s1_dbg[16];      // summed over every wave 0 of a launch: wall_clock64 ticks (100 MHz) per phase, [15] = tiles
#define S1_T(i) if (tid == 0) { const long long now_ = wall_clock64(); tacc[i] += now_ - tp_; tp_ = now_; }
#else
#define S1_T(i)
#endif

struct S1Weights {
  const float *w0t, *b0, *w2t, *b2, *w4t, *b4, *wrt, *br, *wh, *bh, *tt;   // *t = transposed [K][128]; wh [2][128]
};
struct S1Loi {                       // where image b's LOI feature (channel ch, pixel p) lives: base[b * img + ch * cs + p * ps]
  const float* base;
  size_t img;
  int cs, ps;
  const float* jfeat;                // the device path: [B][jn][256] junction projections (s1_junc_proj_kernel);
  int jn;                            // ta8 [B][128*128][8]: thin0..3 | aux0..3 of a pixel (s0_decode_kernel): 512 KB per image, so the
  const float* ta8;                  // taps of an image's lines stay in L2
};

// The endpoints of every candidate line ARE junctions (lines_adjusted = juncs[pair]) and the first layer is linear in its input: the 256
// LOI columns of fc2.0 are applied ONCE PER JUNCTION (300 per image) instead of once per line (~1070, each junction in ~7 of them):
//     proj[b][j][0:128] = W0[:, 0:128] . loi(junction j),   proj[b][j][128:256] = W0[:, 128:256] . loi(junction j)
// and a line's first-layer sum opens with b0 + proj[j1][0:128] + proj[j2][128:256]; its own 240 thin / aux terms follow (a different order
// of the same 496 products + bias: ~1e-7 relative, the golden outputs of the real graph are pinned at 5e-5).  A third of the MFMAs of
// the whole head and half of its activation tile are gone.  8 junctions per workgroup; thread = (feature n, half h), fmaf chain in k order.
constexpr int S1_PJ = 8;
// LOI features of a junction: sampled from the fused head's rows (head, pitch ps: one image per call in practice), or combined from the four
// tap rows lrows [B][jn][4][128] that the gather GEMM made for exactly these taps (s1_junc_rows_kernel lists them; same bil_eval).
__global__ __launch_bounds__(256) void s1_junc_proj_kernel(const float* __restrict__ juncs, const float* __restrict__ head, size_t head_img,
                                                           int ps, const float* __restrict__ lrows, int jn,
                                                           const float* __restrict__ w0t /*[496][128]*/,
                                                           float* __restrict__ proj /*[B][jn][256]*/, size_t stage_stride) {
  __shared__ float fs[S1_PJ][128];
  const int t = threadIdx.x, n = t & 127, h = t >> 7, j0 = blockIdx.x * S1_PJ;
  const size_t img = blockIdx.y;
  const float* jp = juncs + img * stage_stride;
  for (int q = h; q < S1_PJ; q += 2) {
    const int j = min(j0 + q, jn - 1);
    if (lrows) {
      const BilTap bt = bil_setup(128, 128, jp[j * 2], jp[j * 2 + 1]);
      const float* r = lrows + ((img * jn + j) * 4) * 128 + n;
      fs[q][n] = bil_eval(r[0], r[128], r[256], r[384], bt);
    } else {
      fs[q][n] = bil_plane(head + img * head_img + n, 128, 128, ps, jp[j * 2], jp[j * 2 + 1]);
    }
  }
  __syncthreads();
  float acc[S1_PJ];
#pragma unroll
  for (int q = 0; q < S1_PJ; ++q) acc[q] = 0.f;
  const float* wp = w0t + (size_t)h * 128 * 128 + n;
#pragma unroll 8
  for (int k = 0; k < 128; ++k) {
    const float wv = wp[k * 128];
#pragma unroll
    for (int q = 0; q < S1_PJ; ++q) acc[q] = fmaf(wv, fs[q][k], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < S1_PJ; ++q)
    if (j0 + q < jn) proj[((img * jn + j0 + q) * 2 + h) * 128 + n] = acc[q];
}

// rows of the line-feature matrix [B][128*128][128] that the junctions' bilinear taps read, in bil_eval's order (i00, i10, i01, i11):
// the row list of the LOI head's gather GEMM — the head is computed at the <= 1200 pixels per image that are read, not at all 16384
__global__ void s1_junc_rows_kernel(const float* __restrict__ juncs, int jn, int* __restrict__ ridx, size_t stage_stride) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= jn) return;
  const size_t img = blockIdx.y;
  const float* jp = juncs + img * stage_stride + j * 2;
  const BilTap bt = bil_setup(128, 128, jp[0], jp[1]);
  int* o = ridx + (img * jn + j) * 4;
  const int base = (int)img * 128 * 128;
  o[0] = base + bt.i00; o[1] = base + bt.i10; o[2] = base + bt.i01; o[3] = base + bt.i11;
}
void launch_s1_junc_rows(const float* juncs, int jn, int* ridx, int B, size_t stage_stride, hipStream_t st) {
  hipLaunchKernelGGL(s1_junc_rows_kernel, dim3((jn + 63) / 64, B), dim3(64), 0, st, juncs, jn, ridx, stage_stride);
}
void launch_s1_junc_proj(const float* juncs, const float* head, size_t head_img, int ps, const float* lrows, int jn, const float* w0t,
                         float* proj, int B, size_t stage_stride, hipStream_t st) {
  hipLaunchKernelGGL(s1_junc_proj_kernel, dim3((jn + S1_PJ - 1) / S1_PJ, B), dim3(256), 0, st, juncs, head, head_img, ps, lrows, jn, w0t, proj,
                     stage_stride);
}

// PRE: the device path (junction projections + pixel-major thin / aux: 240-row activation tile, 4 workgroups per CU); !PRE: the contract's
// CHW tensors, all 496 input features per line (host-supplied stage-0 tensors: the golden / known-answer tests).
template <bool PRE>
__global__ __launch_bounds__(256, PRE ? 4 : 2) void plnet_s1_kernel(const float* __restrict__ juncs, const float* __restrict__ lines_pred,
                                                       const int* __restrict__ keep, const int* __restrict__ pairs,
                                                       const int* __restrict__ rep, const int* __restrict__ counts, S1Loi loi,
                                                       const float* __restrict__ thin, const float* __restrict__ aux, S1Weights w,
                                                       float* __restrict__ lines_adjusted, float* __restrict__ scores_line,
                                                       int keep_cap, int line_cap, size_t stage_stride) {
  constexpr int XR = PRE ? 256 : 496, XT = PRE ? 0 : 256;                    // rows of the tile; first thin / aux row
  __shared__ float xs[XR * S1_LP];                                         // [k][line]; h0 / h1 take its place once layer 0 has read it
  __shared__ __attribute__((aligned(16))) float la[S1_LT][4], li[S1_LT][4];
  __shared__ int lj[S1_LT][2];
  __shared__ float zh[S1_LT][2];
  __shared__ float tts[32];
  __shared__ float bs[4][128];                                             // biases of the four 128-feature layers
  __shared__ float whs[2][128];                                            // the 2-way head
  float* h0 = xs;                                                          // [128][S1_LP]
  float* h1 = xs + 128 * S1_LP;
  {
    const size_t img = blockIdx.y;
    juncs += img * stage_stride; lines_pred += img * stage_stride; thin += img * stage_stride; aux += img * stage_stride;
    keep += img * keep_cap; pairs += img * line_cap * 2; rep += img * line_cap; counts += img * LINE_CNT_LD;
    lines_adjusted += img * line_cap * 4; scores_line += img * line_cap;
    loi.base += img * loi.img;
    if constexpr (PRE) { loi.jfeat += img * loi.jn * 256; loi.ta8 += img * 128 * 128 * 8; }
  }
  const int m2 = counts[1];
  const int tid = threadIdx.x, n = tid & 127, g0 = (tid >> 7) * (S1_LT / 2);
  const int lane = tid & 63, f0 = (tid >> 6) * 32, col = lane & 31, rb = 4 * (lane >> 5);
  if (tid < 30) tts[tid] = w.tt[tid];
  if (tid < 128) { bs[0][tid] = w.b0[tid]; bs[1][tid] = w.br[tid]; bs[2][tid] = w.b2[tid]; bs[3][tid] = w.b4[tid]; whs[0][tid] = w.wh[tid]; whs[1][tid] = w.wh[128 + tid]; }
#ifdef S1_TIMING
  long long tacc[10] = {0}, tp_ = wall_clock64();
#endif
  // The header of a tile — which two junctions a candidate joins, which proposal it stands for — is three dependent global accesses
  // (pairs -> juncs; rep -> keep -> lines_pred): threads 0..127 fetch the NEXT tile's level by level between the phases of the current
  // one, so that no wave ever waits for it (at ~1 us per access it was 7.7 of a tile's 82 us).
  const int hl = tid >> 2, hc = tid & 3;
  int h_j = 0, h_k = 0;
  float h_v = 0.f, h_li = 0.f;
  auto header_l1 = [&](int l0n) {                       // level 1: indices
    if (tid < S1_LT * 4 && l0n < m2) {
      const int u = min(l0n + hl, m2 - 1);
      h_j = pairs[u * 2 + (hc >> 1)];
      h_k = rep[u];
    }
  };
  auto header_l2 = [&](int l0n) {                       // level 2: junction coordinate, kept-list entry
    if (tid < S1_LT * 4 && l0n < m2) {
      h_v = juncs[h_j * 2 + (hc & 1)];
      h_k = keep[h_k];
    }
  };
  auto header_l3 = [&](int l0n) {                       // level 3: the proposal's coordinate
    if (tid < S1_LT * 4 && l0n < m2) h_li = lines_pred[(size_t)h_k * 4 + hc];
  };
  auto header_put = [&](int l0n) {                      // -> LDS (after a barrier behind the last reader of the previous tile's)
    if (tid < S1_LT * 4 && l0n < m2) {
      la[hl][hc] = h_v;
      li[hl][hc] = h_li;
      if ((hc & 1) == 0) lj[hl][hc >> 1] = h_j;
      if (l0n + hl < m2) lines_adjusted[(size_t)(l0n + hl) * 4 + hc] = h_v;
    }
  };
  const int l_first = blockIdx.x * S1_LT, l_step = gridDim.x * S1_LT;
  header_l1(l_first); header_l2(l_first); header_l3(l_first);
  // a workgroup walks the image's line tiles with the grid's stride (the count is on the device: no launch sized by it)
  for (int l0 = l_first; l0 < m2; l0 += l_step) {
    __syncthreads();                                                       // the previous tile is done with xs, zh
    S1_T(0)
    header_put(l0);
    __syncthreads();
    header_l1(l0 + l_step);
    S1_T(1)
    // sampling: thread = (channel n, half of the tile's lines)
    if constexpr (PRE) {
      // thin / aux blocks: 30 points per line and block, a point's 4 channels are 16 bytes of the pixel-major copy: one (line, block, point)
      // per thread and trip = 4 taps of 16 bytes; rounds of S1_SB trips whose taps are in flight together (a dependent global access
      // costs 1-2 us in this kernel).  The LOI blocks do not exist here: their share of layer 0 is the junction projection.
      constexpr int ITEMS = (S1_LT / 2) * 60;
      constexpr int ROUNDS = (ITEMS + S1_SB * 128 - 1) / (S1_SB * 128);
#pragma unroll 1
      for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        BilTap bt[S1_SB];
        float4 a00[S1_SB], a10[S1_SB], a01[S1_SB], a11[S1_SB];
        int dst[S1_SB];
#pragma unroll
        for (int q = 0; q < S1_SB; ++q) {
          const int it = min(n + (rnd * S1_SB + q) * 128, ITEMS - 1);      // (a clamped trip repeats the last item: same value, same place)
          const int lq = it / 60, rr = it - lq * 60, kind = rr >= 30, j = rr - 30 * kind, l = g0 + lq;
          const float t = tts[j], t1 = 1.0f - t;
          const float4 e = *reinterpret_cast<const float4*>(kind ? li[l] : la[l]);
          bt[q] = bil_setup(128, 128, e.x * t + e.z * t1, e.y * t + e.w * t1);
          const float4* hp = reinterpret_cast<const float4*>(loi.ta8) + kind;
          a00[q] = hp[bt[q].i00 * 2]; a10[q] = hp[bt[q].i10 * 2]; a01[q] = hp[bt[q].i01 * 2]; a11[q] = hp[bt[q].i11 * 2];
          dst[q] = (120 * kind + j) * S1_LP + l;
        }
#pragma unroll
        for (int q = 0; q < S1_SB; ++q) {
          const float v00[4] = {a00[q].x, a00[q].y, a00[q].z, a00[q].w}, v10[4] = {a10[q].x, a10[q].y, a10[q].z, a10[q].w};
          const float v01[4] = {a01[q].x, a01[q].y, a01[q].z, a01[q].w}, v11[4] = {a11[q].x, a11[q].y, a11[q].z, a11[q].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) xs[dst[q] + 30 * c * S1_LP] = bil_eval(v00[c], v10[c], v01[c], v11[c], bt[q]);
        }
      }
    } else {
      const float* lch = loi.base + (size_t)n * loi.cs;
#pragma unroll 4
      for (int l = g0; l < g0 + S1_LT / 2; ++l) {
        xs[n * S1_LP + l] = bil_plane(lch, 128, 128, loi.ps, la[l][0], la[l][1]);
        xs[(128 + n) * S1_LP + l] = bil_plane(lch, 128, 128, loi.ps, la[l][2], la[l][3]);
        if (n < 120) {
          const int c = n / 30, j = n - c * 30;
          const float t = tts[j], t1 = 1.0f - t;
          xs[(256 + n) * S1_LP + l] = bil_plane(thin + (size_t)c * 128 * 128, 128, 128, 1, la[l][0] * t + la[l][2] * t1, la[l][1] * t + la[l][3] * t1);
          xs[(376 + n) * S1_LP + l] = bil_plane(aux + (size_t)c * 128 * 128, 128, 128, 1, li[l][0] * t + li[l][2] * t1, li[l][1] * t + li[l][3] * t1);
        }
      }
    }
    header_l2(l0 + l_step);
    S1_T(2)
    __syncthreads();
    S1_T(3)
    f32x16 o, r;
    if constexpr (PRE) {
      // b0 + W0[:, 0:128] . loi(j1) + W0[:, 128:256] . loi(j2): this lane's 16 features of its line, 4 consecutive per 16-byte load
      const float4* p1 = reinterpret_cast<const float4*>(loi.jfeat + (size_t)lj[col][0] * 256 + f0 + rb);
      const float4* p2 = reinterpret_cast<const float4*>(loi.jfeat + (size_t)lj[col][1] * 256 + 128 + f0 + rb);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 u = p1[2 * q4], v = p2[2 * q4];
        const float* bb = &bs[0][f0 + 8 * q4 + rb];
        o[4 * q4] = bb[0] + u.x + v.x; o[4 * q4 + 1] = bb[1] + u.y + v.y; o[4 * q4 + 2] = bb[2] + u.z + v.z; o[4 * q4 + 3] = bb[3] + u.w + v.w;
      }
      s1_dense<240, true>(w.w0t + 256 * 128, nullptr, xs, o, f0, lane);
    } else {
      s1_dense<496>(w.w0t, bs[0], xs, o, f0, lane);
    }
    header_l3(l0 + l_step);
    S1_T(4)
    s1_dense<240>(w.wrt, bs[1], xs + XT * S1_LP, r, f0, lane);
    S1_T(5)
    __syncthreads();                                                       // every wave is done with the x tile
    S1_T(6)
#pragma unroll
    for (int q = 0; q < 16; ++q) h0[(f0 + 8 * (q >> 2) + rb + (q & 3)) * S1_LP + col] = fmaxf(o[q], 0.f);
    __syncthreads();
    s1_dense<128>(w.w2t, bs[2], h0, o, f0, lane);
#pragma unroll
    for (int q = 0; q < 16; ++q) h1[(f0 + 8 * (q >> 2) + rb + (q & 3)) * S1_LP + col] = fmaxf(o[q], 0.f);
    __syncthreads();
    s1_dense<128>(w.w4t, bs[3], h1, o, f0, lane);
#pragma unroll
    for (int q = 0; q < 16; ++q) h0[(f0 + 8 * (q >> 2) + rb + (q & 3)) * S1_LP + col] = o[q] + fmaxf(r[q], 0.f);   // (h0 was last read before the previous barrier)
    __syncthreads();
    S1_T(7)
    if (tid < S1_LT * 2) {
      const int l = tid >> 1, c = tid & 1;
      float z = w.bh[c];
      for (int k = 0; k < 128; ++k) z = fmaf(whs[c][k], h0[k * S1_LP + l], z);
      zh[l][c] = z;
    }
    __syncthreads();
    if (tid < S1_LT && l0 + tid < m2) {
      const float z0 = zh[tid][0], z1 = zh[tid][1], m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m);
      scores_line[l0 + tid] = e1 / (e0 + e1);
    }
    S1_T(8)
#ifdef S1_TIMING
    if (tid == 0) tacc[9] += 1;
#endif
  }
#ifdef S1_TIMING
  if (tid == 0) for (int i = 0; i < 10; ++i) atomicAdd(&s1_dbg[i], (unsigned long long)tacc[i]);
#endif
}

#ifdef S1_TIMING
}  // namespace airfe
extern "C" int airfe_dbg_s1(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(airfe::s1_dbg), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(airfe::s1_dbg), sizeof(unsigned long long) * 16);
}
namespace airfe {
#endif

// =============================================================================== stage 1 on the 2-byte matrix pipe, fp32-accurate (round 5)
// The f32-input MFMA above runs at the f32 VECTOR rate (157 TFLOP/s: 368 MFMAs of 64 cycles per wave and 32-line tile, half of the kernel's time).  The reference
// runs this engine in FP16 (src/plnet.cpp:216); plain fp16 operands move 0.5-0.9 % of the kept lines across the 0.75 threshold with the real weights
// (profiles/r05_s1_fp16_emulation.txt).  Here every fp32 operand is a PAIR of fp16 values — hi = fp16(v), lo = fp16((v - hi) * 2^11): 22 bits of mantissa, lo
// scaled so that it is never a denormal — and every product three v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//     W x  =  hi_W hi_x  +  2^-11 (hi_W lo_x + lo_W hi_x)          (the lo_W lo_x term, 2^-22 relative, is dropped)
// 138 MFMAs of 32 cycles per wave and tile instead of 368 of 64 (5.3x less matrix time), scores within 2e-6 of the fp32 chain and NO candidate across the threshold
// in the emulation.  The exactness argument: hi_W hi_x is a product of two 11-bit significands — exact in the fp32 accumulator; so are the two cross terms; what
// differs from the f32 MFMA is the summation order (as between any two GEMM kernels) and the dropped 2^-22 term.
// (Compiles to 128 registers at four workgroups per CU with 64 bytes of scratch per lane: a dozen tile-loop-INVARIANT values — table pointers, lane offsets —
// stored once in the prologue and reloaded at four places per tile, outside the MFMA loops; uncapped it takes 141 registers = three workgroups per CU.)
// Layouts: weights in FRAGMENT order [4 feature blocks][K / 16][hi | lo][64 lanes][8] fp16 (a lane's A fragment = 8 consecutive k of one feature: one 16-byte buffer
// load, a wave's = 1 KB in a row; airfe_load.hip writes them); activations in LDS
// [line][K + pad] fp16, hi plane and lo plane (a lane's B fragment = 8 consecutive k of one line: one ds_read_b128).  Accumulator layout = the f32 kernel's.
constexpr int S1H_XP = 264;              // halves per line of the x tile (240 + pad)
constexpr int S1H_HP = 136;              // halves per line of a hidden tile (128 + pad; a multiple of 8: every line starts on a 16-byte boundary for ds_read_b128)
constexpr float S1H_LO = 2048.0f, S1H_LO_INV = 1.0f / 2048.0f;

typedef __attribute__((ext_vector_type(4))) _Float16 f16x4v;
__device__ __forceinline__ f32x16 mfma_32x32x16h(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void s1h_split(float v, _Float16& hi, _Float16& lo) {
  // range guard: a LOI / hidden value beyond fp16's 65504 would give hi = inf and NaN scores that the 0.75 threshold drops silently; clamped, the
  // value enters as +-65504 (hi exact, lo 0) and the head stays finite.  One v_med3_f32; values inside the range are untouched (the real head's are below 64).
  v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * S1H_LO);
}

template <int K>
struct S1Wh {                              // the (hi, lo) fragments of one layer as a buffer resource: [4 feature blocks][K / 16 steps][2 planes][64 lanes][8] fp16
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit S1Wh(const uint16_t* w) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(w), 0, 2 * 128 * K * 2, 0x00020000)) {}
  __device__ __forceinline__ f16x8 ld(unsigned lane_off, unsigned soff) const {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, soff, 0));
  }
};

// acc (+)= W[f0 + .., 0:K] x[0:K, line]; bias != nullptr: acc starts from the bias (else from what the caller put there)
template <int K>
__device__ __forceinline__ void s1h_dense(const uint16_t* __restrict__ w /*fragment order, see S1Wh*/, const float* bias /*LDS or nullptr*/, const uint16_t* xh /*LDS*/,
                                          const uint16_t* xl, int xp, f32x16& acc, int f0, int lane) {
  constexpr int STEPS = K / 16, D = 3;     // 16 k per step; D steps of weights in flight
  static_assert(K % 16 == 0, "K must be a multiple of 16");
  const int i = lane & 31, kk = lane >> 5;
  const S1Wh<K> W(w);
  // a wave's A fragments of one step are 2 KB in a row (hi | lo), every 128-byte line of them used in full by ONE instruction: with the row-major [feature][k] layout a
  // load touched 32 lines for 32 bytes each, and the other 96 bytes had to survive three steps in a vector cache that 16 waves stream through
  const unsigned lb = (unsigned)lane * 16u, wb = (unsigned)__builtin_amdgcn_readfirstlane(f0 >> 5) * (STEPS * 2048u);
  if (bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[f0 + 8 * (r >> 2) + 4 * kk + (r & 3)];
  }
  f32x16 lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) lo[r] = 0.f;
  const uint16_t* bh = xh + i * xp + 8 * kk;
  const uint16_t* bl = xl + i * xp + 8 * kk;
  f16x8 wh[D], wl[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < STEPS) { wh[d] = W.ld(lb, wb + (unsigned)d * 2048u); wl[d] = W.ld(lb, wb + (unsigned)d * 2048u + 1024u); }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const f16x8 ah = wh[s % D], al = wl[s % D];
    const f16x8 xhv = *reinterpret_cast<const f16x8*>(bh + s * 16), xlv = *reinterpret_cast<const f16x8*>(bl + s * 16);
    acc = mfma_32x32x16h(ah, xhv, acc);
    lo = mfma_32x32x16h(ah, xlv, lo);
    lo = mfma_32x32x16h(al, xhv, lo);
    __builtin_amdgcn_sched_barrier(0);     // the refill stays behind the MFMAs that read the slot, and no later step's loads are hoisted above this one (registers)
    if (s + D < STEPS) { wh[s % D] = W.ld(lb, wb + (unsigned)(s + D) * 2048u); wl[s % D] = W.ld(lb, wb + (unsigned)(s + D) * 2048u + 1024u); }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = fmaf(lo[r], S1H_LO_INV, acc[r]);
}

struct S1WeightsH {
  const uint16_t *w0, *wr, *w2, *w4;     // (hi, lo) planes: fc2.0's thin / aux columns [128][240], fc2_res.0 [128][240], fc2.2 and fc2.4 [128][128]
  const float *b0, *br, *b2, *b4, *wh, *bh, *tt;
};

// The device path of plnet_s1_kernel<true> (junction projections + pixel-major thin / aux) with the four dense layers on the 2-byte matrix pipe.  Same tile walk,
// same header prefetch, same sampler, same head; what changes is the activation tiles' layout ([line][k] fp16 pairs instead of [k][line] fp32) and s1h_dense.
__global__ __launch_bounds__(256, 4) void plnet_s1h_kernel(const int* __restrict__ pairs, const int* __restrict__ counts, S1Loi loi, S1WeightsH w,
                                                           const float* __restrict__ lines_adjusted, const float* __restrict__ prop4,
                                                           float* __restrict__ scores_line, int line_cap) {
  // one block, two uses in turn: x tile (hi | lo) [2][32][264] halves = 33792 B; hidden tiles h0 (hi | lo), h1 (hi | lo) [4][32][136] halves = 34816 B
  constexpr int XS_HALVES = 4 * S1_LT * S1H_HP > 2 * S1_LT * S1H_XP ? 4 * S1_LT * S1H_HP : 2 * S1_LT * S1H_XP;
  __shared__ __attribute__((aligned(16))) uint16_t xs[XS_HALVES];
  __shared__ __attribute__((aligned(16))) float la[S1_LT][4], li[S1_LT][4];
  __shared__ int lj[S1_LT][2];
  __shared__ float zp[4][2][S1_LT];                        // the 2-way head's partial sums per wave
  __shared__ float tts[32];
  __shared__ float bs[4][128];
  __shared__ float whs[2][128];
  uint16_t* xh = xs;
  uint16_t* xl = xs + S1_LT * S1H_XP;
  uint16_t* h0h = xs;
  uint16_t* h0l = xs + S1_LT * S1H_HP;
  uint16_t* h1h = xs + 2 * S1_LT * S1H_HP;
  uint16_t* h1l = xs + 3 * S1_LT * S1H_HP;
  static_assert(S1H_XP % 8 == 0 && S1H_HP % 8 == 0, "rows must start on 16-byte boundaries");
  {
    const size_t img = blockIdx.y;
    pairs += img * line_cap * 2; counts += img * LINE_CNT_LD;
    lines_adjusted += img * line_cap * 4; prop4 += img * line_cap * 4; scores_line += img * line_cap;
    loi.jfeat += img * loi.jn * 256; loi.ta8 += img * 128 * 128 * 8;
  }
  const int m2 = counts[1];
  const int tid = threadIdx.x, n = tid & 127, g0 = (tid >> 7) * (S1_LT / 2);
  const int lane = tid & 63, f0 = (tid >> 6) * 32, col = lane & 31, rb = 4 * (lane >> 5);
  // a tile's header — per line the two junctions (= lines_adjusted), the first proposal's four numbers, the junction indices — is three 16 / 16 / 8-byte records the
  // wireframe kernel wrote (the f32 kernel walks pairs -> juncs and rep -> keep -> lines_pred: three dependent loads deep, after the line count): fetched WITHOUT
  // waiting for the line count, rows past it zeroed when they go to LDS (junction 0 at the origin: in range for every gather, never stored)
  const int hl = tid & 31, hk = tid >> 5;
  float4 h_v = make_float4(0.f, 0.f, 0.f, 0.f);
  auto header_ld = [&](int l0n) {
    if (tid < 96) {
      const size_t u = (size_t)min(l0n + hl, line_cap - 1);
      if (hk == 0) h_v = *reinterpret_cast<const float4*>(lines_adjusted + u * 4);
      else if (hk == 1) h_v = *reinterpret_cast<const float4*>(prop4 + u * 4);
      else {
        const int2 pr = *reinterpret_cast<const int2*>(pairs + u * 2);
        h_v = make_float4(__int_as_float(pr.x), __int_as_float(pr.y), 0.f, 0.f);
      }
    }
  };
  auto header_put = [&](int l0n) {
    if (tid < 96) {
      const float4 v = l0n + hl < m2 ? h_v : make_float4(0.f, 0.f, 0.f, 0.f);
      if (hk == 0) *reinterpret_cast<float4*>(la[hl]) = v;
      else if (hk == 1) *reinterpret_cast<float4*>(li[hl]) = v;
      else { lj[hl][0] = __float_as_int(v.x); lj[hl][1] = __float_as_int(v.y); }
    }
  };
  // this lane's 4 consecutive features of a hidden layer (accumulator registers 4 q4 .. 4 q4 + 3) of line `col`, as hi | lo halves
  auto put_hidden = [&](uint16_t* ph, uint16_t* pl, const f32x16& v, bool relu) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      _Float16 hh[4], ll[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) s1h_split(relu ? fmaxf(v[4 * q4 + e], 0.f) : v[4 * q4 + e], hh[e], ll[e]);
      const int off = col * S1H_HP + f0 + 8 * q4 + rb;
      *reinterpret_cast<uint2*>(ph + off) = __builtin_bit_cast(uint2, f16x4v{hh[0], hh[1], hh[2], hh[3]});
      *reinterpret_cast<uint2*>(pl + off) = __builtin_bit_cast(uint2, f16x4v{ll[0], ll[1], ll[2], ll[3]});
    }
  };
#ifdef S1_TIMING
  long long tacc[11] = {0}, tp_ = wall_clock64();
#endif
  const int l_first = blockIdx.x * S1_LT, l_step = gridDim.x * S1_LT;
  header_ld(l_first);
  if (l_first >= m2) return;                               // the grid is sized for the densest image the arena admits: most of its workgroups have no tile
  if (tid < 30) tts[tid] = w.tt[tid];
  if (tid < 128) { bs[0][tid] = w.b0[tid]; bs[1][tid] = w.br[tid]; bs[2][tid] = w.b2[tid]; bs[3][tid] = w.b4[tid]; whs[0][tid] = w.wh[tid]; whs[1][tid] = w.wh[128 + tid]; }
  for (int l0 = l_first; l0 < m2; l0 += l_step) {
    __syncthreads();                                                       // the previous tile is done with xs, zp
#ifdef S1_TIMING
    if (l0 == l_first) { S1_T(10) } else { S1_T(0) }                       // [10]: the prologue (tables to LDS, the first tile's three dependent header loads)
#endif
    header_put(l0);
    __syncthreads();
    if (l0 + l_step < m2) header_ld(l0 + l_step);
    S1_T(1)
    {
      // thin / aux taps exactly as plnet_s1_kernel<true> takes them (same bil_setup / bil_eval: the same fp32 values), stored as fp16 pairs at [line][k]
      constexpr int ITEMS = (S1_LT / 2) * 60, SB = S1_SB;
      constexpr int ROUNDS = (ITEMS + SB * 128 - 1) / (SB * 128);
#pragma unroll 1
      for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        BilTap bt[SB];
        float4 a00[SB], a10[SB], a01[SB], a11[SB];
        int dst[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          const int it = min(n + (rnd * SB + q) * 128, ITEMS - 1);
          const int lq = it / 60, rr = it - lq * 60, kind = rr >= 30, j = rr - 30 * kind, l = g0 + lq;
          const float t = tts[j], t1 = 1.0f - t;
          const float4 e = *reinterpret_cast<const float4*>(kind ? li[l] : la[l]);
          bt[q] = bil_setup(128, 128, e.x * t + e.z * t1, e.y * t + e.w * t1);
          const float4* hp = reinterpret_cast<const float4*>(loi.ta8) + kind;
          a00[q] = hp[bt[q].i00 * 2]; a10[q] = hp[bt[q].i10 * 2]; a01[q] = hp[bt[q].i01 * 2]; a11[q] = hp[bt[q].i11 * 2];
          dst[q] = l * S1H_XP + 120 * kind + j;
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          const float v00[4] = {a00[q].x, a00[q].y, a00[q].z, a00[q].w}, v10[4] = {a10[q].x, a10[q].y, a10[q].z, a10[q].w};
          const float v01[4] = {a01[q].x, a01[q].y, a01[q].z, a01[q].w}, v11[4] = {a11[q].x, a11[q].y, a11[q].z, a11[q].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            _Float16 hh, ll;
            s1h_split(bil_eval(v00[c], v10[c], v01[c], v11[c], bt[q]), hh, ll);
            xh[dst[q] + 30 * c] = __builtin_bit_cast(uint16_t, hh);
            xl[dst[q] + 30 * c] = __builtin_bit_cast(uint16_t, ll);
          }
        }
      }
    }
    S1_T(2)
    __syncthreads();
    S1_T(3)
    f32x16 o, r;
    // the junction terms of layer 0 — b0 + W0[:, 0:128] . loi(j1) + W0[:, 128:256] . loi(j2), per junction by s1h_junc_proj_kernel — are two gathered rows per line:
    // asked for here, used after the residual layer (which needs nothing of them) has covered their latency
    float4 ju[4], jv[4];
    {
      const float4* p1 = reinterpret_cast<const float4*>(loi.jfeat + (size_t)lj[col][0] * 256 + f0 + rb);
      const float4* p2 = reinterpret_cast<const float4*>(loi.jfeat + (size_t)lj[col][1] * 256 + 128 + f0 + rb);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) { ju[q4] = p1[2 * q4]; jv[q4] = p2[2 * q4]; }
    }
    s1h_dense<240>(w.wr, bs[1], xh, xl, S1H_XP, r, f0, lane);
    S1_T(5)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float* bb = &bs[0][f0 + 8 * q4 + rb];
      o[4 * q4] = bb[0] + ju[q4].x + jv[q4].x; o[4 * q4 + 1] = bb[1] + ju[q4].y + jv[q4].y;
      o[4 * q4 + 2] = bb[2] + ju[q4].z + jv[q4].z; o[4 * q4 + 3] = bb[3] + ju[q4].w + jv[q4].w;
    }
    s1h_dense<240>(w.w0, nullptr, xh, xl, S1H_XP, o, f0, lane);
    S1_T(4)
    __syncthreads();                                                       // every wave is done with the x tile
    S1_T(6)
    put_hidden(h0h, h0l, o, true);
    __syncthreads();
    s1h_dense<128>(w.w2, bs[2], h0h, h0l, S1H_HP, o, f0, lane);
    put_hidden(h1h, h1l, o, true);                                         // (h1 does not overlap h0: no barrier between the last read of h0 and this)
    __syncthreads();
    s1h_dense<128>(w.w4, bs[3], h1h, h1l, S1H_HP, o, f0, lane);
    S1_T(7)
    // the 2-way head on the accumulators: this lane's 16 features of line `col`, then the other half of the wave (features + 4), then the four waves through LDS
    // (the f32 kernel writes the layer to LDS and lets 64 threads walk its 128 features: one more barrier and a 128-step chain per tile)
    {
      float z0 = 0.f, z1 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int f = f0 + 8 * (q >> 2) + rb + (q & 3);
        const float v = o[q] + fmaxf(r[q], 0.f);
        z0 = fmaf(whs[0][f], v, z0);
        z1 = fmaf(whs[1][f], v, z1);
      }
      z0 += __shfl_xor(z0, 32);
      z1 += __shfl_xor(z1, 32);
      if (lane < 32) { zp[tid >> 6][0][col] = z0; zp[tid >> 6][1][col] = z1; }
    }
    __syncthreads();
    if (tid < S1_LT && l0 + tid < m2) {
      const float z0 = w.bh[0] + ((zp[0][0][tid] + zp[1][0][tid]) + (zp[2][0][tid] + zp[3][0][tid]));
      const float z1 = w.bh[1] + ((zp[0][1][tid] + zp[1][1][tid]) + (zp[2][1][tid] + zp[3][1][tid]));
      const float m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m);
      scores_line[l0 + tid] = e1 / (e0 + e1);
    }
    S1_T(8)
#ifdef S1_TIMING
    if (tid == 0) tacc[9] += 1;
#endif
  }
#ifdef S1_TIMING
  if (tid == 0) for (int i = 0; i < 11; ++i) atomicAdd(&s1_dbg[i], (unsigned long long)tacc[i]);
#endif
}

// The junction projections proj[b][j][0:128] = W0[:, 0:128] . loi(j), [128:256] = W0[:, 128:256] . loi(j) (see s1_junc_proj_kernel) on the same pipe: tiles of 32
// junctions, the LOI features of a junction combined from its four tap rows exactly as s1_junc_proj_kernel does (bil_eval), split into fp16 pairs, two 128-deep
// products per tile.  103 us of scalar multiply-adds per 128 images -> a handful of MFMAs per junction tile.
__global__ __launch_bounds__(256, 4) void s1h_junc_proj_kernel(const float* __restrict__ juncs, const float* __restrict__ lrows, int jn, const uint16_t* __restrict__ wa,
                                                               const uint16_t* __restrict__ wb, float* __restrict__ proj, size_t stage_stride) {
  __shared__ __attribute__((aligned(16))) uint16_t xs[2 * S1_LT * S1H_HP];
  uint16_t* xh = xs;
  uint16_t* xl = xs + S1_LT * S1H_HP;
  const int tid = threadIdx.x, n = tid & 127, half = tid >> 7, lane = tid & 63, f0 = (tid >> 6) * 32, col = lane & 31, rb = 4 * (lane >> 5);
  const size_t img = blockIdx.y;
  const int j0 = blockIdx.x * S1_LT;
  const float* jp = juncs + img * stage_stride;
#pragma unroll 4
  for (int q = half; q < S1_LT; q += 2) {
    const int j = min(j0 + q, jn - 1);
    const BilTap bt = bil_setup(128, 128, jp[j * 2], jp[j * 2 + 1]);
    const float* rr = lrows + ((img * jn + j) * 4) * 128 + n;
    _Float16 hh, ll;
    s1h_split(bil_eval(rr[0], rr[128], rr[256], rr[384], bt), hh, ll);
    xh[q * S1H_HP + n] = __builtin_bit_cast(uint16_t, hh);
    xl[q * S1H_HP + n] = __builtin_bit_cast(uint16_t, ll);
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    s1h_dense<128>(h ? wb : wa, nullptr, xh, xl, S1H_HP, acc, f0, lane);
    if (j0 + col < jn) {
      float* o = proj + (img * jn + j0 + col) * 256 + h * 128 + f0 + rb;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<float4*>(o + 8 * q4) = make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
    }
  }
}
void launch_s1h_junc_proj(const float* juncs, const float* lrows, int jn, const uint16_t* wa, const uint16_t* wb, float* proj, int B, size_t stage_stride,
                          hipStream_t st) {
  hipLaunchKernelGGL(s1h_junc_proj_kernel, dim3((jn + S1_LT - 1) / S1_LT, B), dim3(256), 0, st, juncs, lrows, jn, wa, wb, proj, stage_stride);
}

void launch_plnet_s1h(const int* pairs, const int* counts, const float* proj, const float* ta8, const uint16_t* const* wsplit /*w0 wr w2 w4*/, const float* const* w,
                      const float* lines_adjusted, const float* prop4, float* scores_line, int line_cap, int B, hipStream_t st) {
  S1WeightsH sw{wsplit[0], wsplit[1], wsplit[2], wsplit[3], w[1], w[7], w[3], w[5], w[8], w[9], w[10]};
  S1Loi sl{nullptr, 0, 128 * 128, 1, proj, 300, ta8};
  const int gx = B == 1 ? 512 : (B <= 8 ? 128 : 64);
  hipLaunchKernelGGL(plnet_s1h_kernel, dim3(gx, B), dim3(256), 0, st, pairs, counts, sl, sw, lines_adjusted, prop4, scores_line, line_cap);
}

// proj != nullptr (the device path): [B][300][256] junction projections (launch_s1_junc_proj) + ta8 [B][128*128][8] thin | aux pixel-major
// (launch_s0_decode); loi / thin / aux are not used.  proj == nullptr: the contract's CHW tensors — loi [128][128*128] (one image: loi_img
// = 0), thin / aux the stage's CHW planes — all 496 features per line.
// keep [B][keep_cap], pairs / rep / lines_adjusted / scores_line [B][line_cap].
void launch_plnet_s1(const float* juncs, const float* lines_pred, const int* keep, const int* pairs, const int* rep,
                     const int* counts, const float* loi, size_t loi_img, const float* proj, const float* ta8, const float* thin,
                     const float* aux, const float* const* w, float* lines_adjusted, float* scores_line, int keep_cap, int line_cap, int B,
                     size_t stage_stride, hipStream_t st) {
  S1Weights sw{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10]};
  S1Loi sl{loi, loi_img, 128 * 128, 1, proj, 300, ta8};
  const int gx = B == 1 ? 512 : (B <= 8 ? 128 : 64);
  if (proj && ta8)
    hipLaunchKernelGGL(plnet_s1_kernel<true>, dim3(gx, B), dim3(256), 0, st, juncs, lines_pred, keep, pairs, rep, counts, sl, thin, aux, sw,
                       lines_adjusted, scores_line, keep_cap, line_cap, stage_stride);
  else
    hipLaunchKernelGGL(plnet_s1_kernel<false>, dim3(gx, B), dim3(256), 0, st, juncs, lines_pred, keep, pairs, rep, counts, sl, thin, aux, sw,
                       lines_adjusted, scores_line, keep_cap, line_cap, stage_stride);
}

// =============================================================================== line filter + junction map
// src/plnet.cpp:519-558 (+ rescale :577-582).  junction_map[y][x] = p_valid(x,y) is a pure function of the pixel,
// so the sequential "assignment, not OR" of the reference is order-independent; lines are emitted in ascending i.
__global__ __launch_bounds__(1024) void line_filter_kernel(const float* __restrict__ la, const float* __restrict__ sc,
                                                           const int* __restrict__ counts, int border, float line_thr,
                                                           float len_thr, float w_scale, float h_scale, int R,
                                                           unsigned char* __restrict__ jmap, double* __restrict__ lines_out,
                                                           int capL, int* __restrict__ nlines, int* __restrict__ nfound, int line_cap, int nj) {
  __shared__ unsigned wsum[16];
  const bool mark = (int)blockIdx.x < nj;          // only the images whose junctions are wanted keep a junction map
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
      if (pass == 0 && mark) {
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

// one workgroup per image: la [B][line_cap][4], sc [B][line_cap], lines_out [B][capL][4], nlines / nfound [B] (nfound may be nullptr);
// jmap [nj][R * R], zeroed by the caller: the junction maps of the first nj images (the others' junctions are not wanted)
void launch_line_filter(const float* la, const float* sc, const int* counts, int border, float line_thr, float len_thr,
                        float w_scale, float h_scale, int R, unsigned char* jmap, int nj, double* lines_out, int capL, int* nlines, int* nfound,
                        int line_cap, int B, hipStream_t st) {
  hipLaunchKernelGGL(line_filter_kernel, dim3(B), dim3(1024), 0, st, la, sc, counts, border, line_thr, len_thr, w_scale,
                     h_scale, R, jmap, lines_out, capL, nlines, nfound, line_cap, nj);
}

// hipMemsetAsync's byte fill ran at ~110 GB/s on 33 MB of junction maps (0.3 ms per step): 16-byte stores instead
__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
void launch_zero16(void* p, size_t bytes, hipStream_t st) {        // bytes % 16 == 0, p 16-byte aligned
  const size_t n16 = bytes / 16;
  if (n16 == 0) return;
  hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 4096)), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n16);
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
// GEMMs (airfe_match.hip).  As scalar FMA loops all five layers took 0.29 ms per 51200 keypoints, latency-bound on LDS broadcast reads.
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

// The whole of log_optimal_transport in ONE launch.  History: 2 x iters launches per step, each reading the couplings twice for all pairs
// (the sg_sinkhorn_row / _col kernels above, still the fallback): ~25 ms per 64 pairs at N = 400; ONE workgroup per pair (no inter-workgroup
// traffic at all): 21 ms — a single workgroup is latency-bound on its 643 KB matrix; G workgroups per pair streaming the matrix from L2 and
// meeting at a counter: 13.6 ms (retired: it was launched non-cooperatively, i.e. it ASSUMED co-residency of its workgroups); below: 1.46 ms.
// Register-resident form: the couplings of a pair never leave the register files of the G workgroups that share it.  Workgroup g owns
// a slice of the ROWS (wave w the rows r_lo + w, + 8, ...; lane l the columns l, l + 64, ...: RW x MAXC values per lane, 91 at N = 400,
// 153 at N = 1024), loaded once.  The u half-iteration is wave-local (a row is one wave: two wave reductions, u[i] stays in that wave);
// the v half-iteration reduces each column over the wave's own rows in registers, over the 8 waves through LDS, and over the G
// workgroups through ONE exchange of (max, sum) pairs per column and iteration — one rendezvous per iteration, ~6 KB per workgroup,
// instead of re-reading 643 KB of couplings twice.  The launch is COOPERATIVE (the runtime guarantees that all workgroups are
// resident, or refuses the launch and the per-iteration launches run); spins are bounded all the same: a timeout poisons that pair's
// output with NaN AND raises *fail_flag, which the host entry points check after their synchronisation (airfe.hip: sg_host).
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
                                                             float* __restrict__ Z, unsigned* __restrict__ counters, unsigned* __restrict__ fail_flag) {
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
                             unsigned* counters, unsigned* fail_flag, hipStream_t st) {
  int G = (Np + 1 + 8 * RW - 1) / (8 * RW);
  if (G > 16 || Np + 1 > 64 * MAXC || Lz < 64 * MAXC) return false;
  static int max_wgs[64];                                    // co-resident workgroups of this instantiation, per device ordinal (0 = not asked yet)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (max_wgs[dev] == 0) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sg_sinkhorn_reg_kernel<RW, MAXC>, 512, 0) != hipSuccess)
      max_wgs[dev] = -1;
    else
      max_wgs[dev] = std::max(per_cu * prop.multiProcessorCount, 1);
  }
  if (B * G > max_wgs[dev]) return false;
  (void)hipMemsetAsync(counters, 0, (size_t)B * 64, st);
  void* args[] = {(void*)&sim, (void*)&lens, (void*)&Np, (void*)&Lz, (void*)&alpha, (void*)&iters, (void*)&G, (void*)&xch, (void*)&Z, (void*)&counters,
                  (void*)&fail_flag};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&sg_sinkhorn_reg_kernel<RW, MAXC>), dim3(B * G), dim3(512), args, 0, st);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

// counters: B x 64 bytes of rendezvous counters; fail_flag: one word, raised (never cleared here) when a rendezvous timed out
void launch_sg_sinkhorn(const float* sim, const int* lens, int B, int Np, int Lz, float alpha, int iters, float* u, float* v,
                        float* Z, unsigned* counters, unsigned* fail_flag, float* xch, hipStream_t st) {
  if (counters && xch && fail_flag) {
    float2* x2 = reinterpret_cast<float2*>(xch);
    if (Np + 1 <= 448 && try_sinkhorn_reg<13, 7>(sim, lens, B, Np, Lz, alpha, iters, x2, Z, counters, fail_flag, st)) return;
    if (Np + 1 <= 1088 && try_sinkhorn_reg<9, 17>(sim, lens, B, Np, Lz, alpha, iters, x2, Z, counters, fail_flag, st)) return;
  }
  // what does not fit the register files, or a device that cannot hold the cooperative grid (a partition, a busy GPU): one launch per half-iteration
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
    const float m = mutual ? expf_like_glibc(max0[(size_t)b * Lz + tid]) : 0.f;
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
// FIRST minimum on ties (strict '<'), until a leaf; emit the leaf's word id and weight.
//
// The distance is summed in the ORDER the reference's build sums it, so that near-ties between children fall the way they fall there
// (oracle/_ref, tests/test_gpu_ref_pin.py): Eigen reduces a 256-float `(a - b).squaredNorm()` on SSE2 as two 4-lane packet accumulators —
// chain j (0..7) adds the squares of elements j, j + 8, j + 16, ... one after the other — then r0 += r1 and (a0 + a2) + (a1 + a3)
// (oracle/ref_post.py::_eigen_sse2_sum).  One wave per feature: lane = 8 * child + chain, eight children per pass; a lane keeps its 32
// feature elements in registers and walks its chain sequentially (every product and sum rounded on its own: no contraction).  One load
// instruction of the wave reads 8 rows x 32 B; the upper levels of the tree stay in L2.
__global__ __launch_bounds__(256) void bow_transform_kernel(const float* __restrict__ feat, int ld, int off, int N,
                                                            const float* __restrict__ node_desc, const int* __restrict__ first_child,
                                                            const int* __restrict__ n_children, const int* __restrict__ word_id,
                                                            const float* __restrict__ weight, unsigned* __restrict__ out_word,
                                                            float* __restrict__ out_weight, int* __restrict__ out_node) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const int chain = lane & 7, slot = lane >> 3;
  const float* fp = feat + (size_t)i * ld + off + chain;
  float f[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) f[m] = fp[8 * m];
  int node = 0;
  for (int nc = n_children[0]; nc > 0; nc = n_children[node]) {
    const int c0 = first_child[node];
    float best = INFINITY;
    int bi = c0;
    for (int cb = 0; cb < nc; cb += 8) {
      const int c = cb + slot;
      float acc = INFINITY;
      if (c < nc) {
        const float* dp = node_desc + (size_t)(c0 + c) * 256 + chain;
        float d[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) d[m] = dp[8 * m];
        const float t0 = __fsub_rn(f[0], d[0]);
        acc = __fmul_rn(t0, t0);
        asm volatile("" : "+v"(acc));
#pragma unroll
        for (int m = 1; m < 32; ++m) {
          const float t = __fsub_rn(f[m], d[m]);
          float sq = __fmul_rn(t, t);
          asm volatile("" : "+v"(sq));                       // opaque to hipcc, which fuses sq + acc into an FMA even through __fmul_rn (see rounded_mul below)
          acc = __fadd_rn(acc, sq);
        }
      }
      acc = __fadd_rn(acc, __shfl_down(acc, 4));           // chains 0..3: r0[j] + r1[j]
      const float a02 = __fadd_rn(acc, __shfl_down(acc, 2));  // chain 0: a0 + a2; chain 1: a1 + a3
      const float dist = __fadd_rn(a02, __shfl_down(a02, 1)); // chain 0: (a0 + a2) + (a1 + a3)
#pragma unroll
      for (int k = 0; k < 8; ++k) {                         // children of this pass in order, strict '<' (TemplatedVocabulary.h:1336)
        const float dk = __shfl(dist, 8 * k);
        if (cb + k < nc && dk < best) { best = dk; bi = c0 + cb + k; }
      }
    }
    node = bi;
  }
  if (lane == 0) {
    const float w = weight[node];
    out_word[i] = w > 0.f ? (unsigned)word_id[node] : 0xFFFFFFFFu;      // database.cc:77-83: stopped words -> UINT_MAX
    out_weight[i] = w;
    if (out_node) out_node[i] = node;
  }
}

void launch_bow_transform(const float* feat, int ld, int off, int N, const float* node_desc, const int* first_child,
                          const int* n_children, const int* word_id, const float* weight, unsigned* out_word, float* out_weight,
                          int* out_node, hipStream_t st) {
  if (N < 1) return;
  hipLaunchKernelGGL(bow_transform_kernel, dim3((N + 3) / 4), dim3(256), 0, st, feat, ld, off, N, node_desc, first_child, n_children,
                     word_id, weight, out_word, out_weight, out_node);
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

// one wave per line; pass 0 counts, pass 1 writes at row_ptr[line] (ballot + prefix keeps ascending point order).  blockIdx.y = frame: every
// array is `frame stride` apart, line / point counts are read from the device (the PLNet batch entries leave them there)
template <bool WRITE>
__global__ __launch_bounds__(256) void pl_assign_kernel(PlAssignArgs a) {
  const int b = blockIdx.y;
  const int L = min(a.nlines[b], a.capL), N = min(a.npts[b], a.cap);
  const int lane = threadIdx.x & 63;
  const double* lines = a.lines + (size_t)b * a.capL * 4;
  const float* feat = a.feat + (size_t)b * a.cap * 259;
  int* pt_idx = a.pt_idx + (size_t)b * a.capE;
  double* pt_dist = a.pt_dist + (size_t)b * a.capE;
  // a frame's lines are dealt out over the grid's x dimension (the line COUNT lives on the device: the grid is sized for ~256 lines per frame and
  // walks further with its stride — a capL-sized grid of mostly empty workgroups cost more than the work)
  for (int line = blockIdx.x * 4 + (threadIdx.x >> 6); line < L; line += gridDim.x * 4) {
    const double lx1 = lines[line * 4 + 0], ly1 = lines[line * 4 + 1], lx2 = lines[line * 4 + 2], ly2 = lines[line * 4 + 3];
    int cnt = 0;
    const int base = WRITE ? a.row_ptr[(size_t)b * (a.capL + 1) + line] : 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
      const int j = j0 + lane;
      float d = 0.f;
      bool hit = false;
      if (j < N) hit = point_on_line(lx1, ly1, lx2, ly2, (double)feat[(size_t)j * 259 + 1], (double)feat[(size_t)j * 259 + 2], d);
      const unsigned long long m = __ballot(hit);
      if (WRITE && hit) {
        const int pos = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < a.capE) { pt_idx[pos] = j; pt_dist[pos] = (double)d; }
      }
      cnt += __popcll(m);
    }
    if (!WRITE && lane == 0) a.counts[(size_t)b * a.capL + line] = cnt;
  }
}

// exclusive scan of counts[L] -> row_ptr[L+1] per frame (L is a few hundred: one workgroup per frame, serial per 256-chunk carry)
__global__ __launch_bounds__(256) void pl_scan_kernel(PlAssignArgs a) {
  __shared__ int buf[256];
  __shared__ int carry;
  const int b = blockIdx.x;
  const int L = min(a.nlines[b], a.capL);
  const int* counts = a.counts + (size_t)b * a.capL;
  int* row_ptr = a.row_ptr + (size_t)b * (a.capL + 1);
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
  if (threadIdx.x == 0) {
    row_ptr[L] = carry;
    if (a.total) a.total[b] = carry;
  }
}

void launch_assign_points_to_lines(const PlAssignArgs& a, int B, hipStream_t st) {
  if (B <= 0 || a.capL <= 0) return;
  const dim3 grid(std::min((a.capL + 3) / 4, 64), B);
  hipLaunchKernelGGL(pl_assign_kernel<false>, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(pl_scan_kernel, dim3(B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(pl_assign_kernel<true>, grid, dim3(256), 0, st, a);
}

// =============================================================================== MatchLines (src/line_processor.cc:122-172)
// The voting matrix M[l0][l1] = number of point matches (q, t) with q on line l0 of frame 0 and t on line l1 of frame 1 is an
// integer "GEMM" over the matches: one bit per (line, match) says whether the match's point lies on the line, and M is the
// popcount of the AND of two bit rows.  All index work: exact.
__global__ __launch_bounds__(256) void ml_bits_kernel(MlArgs a, int side) {
  const int b = blockIdx.y;
  const int L = min((side ? a.nlines1 : a.nlines0)[b], a.capL);
  const int* row_ptr = (side ? a.row_ptr1 : a.row_ptr0) + (size_t)b * (a.capL + 1);
  const int* pt_idx = (side ? a.pt_idx1 : a.pt_idx0) + (size_t)b * a.capE;
  const int* matches = a.matches + (size_t)b * a.mcap * 2;
  const int nmatch = min(a.nmatch[b], a.mcap);
  // one WAVE per line (4 lines per workgroup), a lane per 32-match word; lines dealt out with the grid's stride
  for (int l = blockIdx.x * 4 + (threadIdx.x >> 6); l < L; l += gridDim.x * 4) {
  unsigned* bits = (side ? a.bits1 : a.bits0) + ((size_t)b * a.capL + l) * a.W;
  // (clamped to the relation's capacity: after an overflowed AssignPointsToLines — d_total > capE — row_ptr runs past the entries that were written)
  const int rb = min(row_ptr[l], a.capE), re = min(row_ptr[l + 1], a.capE);
  for (int w = threadIdx.x & 63; w < a.W; w += 64) {
    unsigned word = 0;
    for (int k = 0; k < 32; ++k) {
      const int m = w * 32 + k;
      if (m >= nmatch) break;
      if (a.filter_on) {           // Frame::AddRightFeatures, src/frame.cc:147-160: stereo matches outside the camera's disparity band are dropped first
        const float* f0 = a.feat0 + ((size_t)b * a.cap + matches[2 * m]) * 259;
        const float* f1 = a.feat1 + ((size_t)b * a.cap + matches[2 * m + 1]) * 259;
        const double dx = (double)fabsf(f0[1] - f1[1]), dy = (double)fabsf(f0[2] - f1[2]);
        if (!(dx > a.min_x_diff && dx < a.max_x_diff && dy <= a.max_y_diff)) continue;
      }
      const int p = matches[2 * m + side];
      for (int r = rb; r < re; ++r)
        if (pt_idx[r] == p) { word |= 1u << k; break; }        // a std::map key occurs once per line
    }
    bits[w] = word;
  }
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

__global__ __launch_bounds__(256) void ml_vote_rowmax_kernel(MlArgs a) {
  __shared__ int sv[256], si[256];
  const int b = blockIdx.y;
  const int L0 = min(a.nlines0[b], a.capL), L1 = min(a.nlines1[b], a.capL);
  for (int l0 = blockIdx.x; l0 < L0; l0 += gridDim.x) {          // (L0 is workgroup-uniform: every thread takes the same trips, barriers are safe)
    const unsigned* r0 = a.bits0 + ((size_t)b * a.capL + l0) * a.W;
    int bv = -1, bi = 0x7fffffff;
    for (int l1 = threadIdx.x; l1 < L1; l1 += blockDim.x) {
      const unsigned* r1 = a.bits1 + ((size_t)b * a.capL + l1) * a.W;
      int v = 0;
      for (int w = 0; w < a.W; ++w) v += __popc(r0[w] & r1[w]);
      if (v > bv) { bv = v; bi = l1; }                           // l1 ascends within a thread: strict > keeps the first
    }
    ml_first_max(bv, bi, sv, si);
    if (threadIdx.x == 0) { a.row_loc[(size_t)b * a.capL + l0] = bi; a.line_matches[(size_t)b * a.capL + l0] = -1; }
  }
}

__global__ __launch_bounds__(256) void ml_colmax_kernel(MlArgs a) {
  __shared__ int sv[256], si[256];
  const int b = blockIdx.y;
  const int L0 = min(a.nlines0[b], a.capL), L1 = min(a.nlines1[b], a.capL);
  if (L0 == 0 || a.npts0[b] == 0 || a.npts1[b] == 0) return;                      // src/line_processor.cc:132
  const int* row_ptr0 = a.row_ptr0 + (size_t)b * (a.capL + 1);
  const int* row_ptr1 = a.row_ptr1 + (size_t)b * (a.capL + 1);
  for (int j = blockIdx.x; j < L1; j += gridDim.x) {
    int bv = -1, bi = 0x7fffffff;
    // (the votes of column j again from the bit rows — W words per entry — instead of a [capL][capL] matrix per frame: that matrix was 256 MB of scratch
    // at 64 frames x 1024 line slots and grew with capL^2)
    const unsigned* r1 = a.bits1 + ((size_t)b * a.capL + j) * a.W;
    for (int i = threadIdx.x; i < L0; i += blockDim.x) {
      const unsigned* r0 = a.bits0 + ((size_t)b * a.capL + i) * a.W;
      int v = 0;
      for (int w = 0; w < a.W; ++w) v += __popc(r0[w] & r1[w]);
      if (v > bv) { bv = v; bi = i; }
    }
    ml_first_max(bv, bi, sv, si);
    if (threadIdx.x == 0 && bv >= 2 && a.row_loc[(size_t)b * a.capL + bi] == j) {          // :171
      const int n0 = row_ptr0[bi + 1] - row_ptr0[bi], n1 = row_ptr1[j + 1] - row_ptr1[j];
      const float score = __fdiv_rn((float)(bv * bv), (float)min(n0, n1));        // :174 float / size_t -> float division
      if (!((double)score < 0.8)) a.line_matches[(size_t)b * a.capL + bi] = j;    // :175; distinct j cannot name the same row: row_loc[bi] == j
    }
  }
}

void launch_match_lines(const MlArgs& a, int B, hipStream_t st) {
  if (B <= 0 || a.capL <= 0) return;
  const dim3 gbits(std::min((a.capL + 3) / 4, 64), B), grid(std::min(a.capL, 256), B);      // sized for ~256 lines per frame; the kernels walk further with the stride
  hipLaunchKernelGGL(ml_bits_kernel, gbits, dim3(256), 0, st, a, 0);
  hipLaunchKernelGGL(ml_bits_kernel, gbits, dim3(256), 0, st, a, 1);
  hipLaunchKernelGGL(ml_vote_rowmax_kernel, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(ml_colmax_kernel, grid, dim3(256), 0, st, a);
}

}  // namespace airfe
