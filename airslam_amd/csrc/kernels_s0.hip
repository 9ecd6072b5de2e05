// airfe — PLNet stage-0 LINE BRANCH on the device (the tensors of SURVEY.md Appendix A.1 that src/plnet.cpp:453-462 fetches from
// the stage-0 engine and :468-509 feeds to stage 1).  plnet_s0.onnx is absent from the reference checkout, so the head below is
// the PUBLISHED architecture those tensors come from — HAWPv3 (Xue et al., "Holistically-Attracted Wireframe Parsing", hawp/fsl/
// model/detector.py: hafm_decoding, non_maximum_suppression + get_junctions, wireframe_matcher) — on the shared VGG trunk:
//     conv3a features [128][128][128]  -> 3x3 conv 128 -> 128 + ReLU  (MFMA, kernels_conv128r.hip)
//                                      -> 1x1 conv 128 -> 145         (MFMA GEMM): loi_features (128) | md (3) dis res | jloc (2) | joff (2) | thin (4) | aux (4)
// and the decode kernels of this file.  Contract and layouts are the reference's (names, shapes, the [3][128][128] proposal
// order, float-typed junction indices); the WEIGHTS are seeded synthetic ones, parity = against the oracle restatement
// (oracle/ref_nets.py::plnet_s0_lines), UNPINNED against the missing model.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace airfe {

constexpr int S0_F = 128;                 // feature-map side (512 / 4)
constexpr int S0_NPX = S0_F * S0_F;
constexpr int S0_LD = 160;                // row pitch of the fused head GEMM output (145 valid)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The per-pixel decode, shared by the two kernels that run it (one expression each: the same bits from either).
// hafm_decoding: md_un = (md0 - 0.5) 2 pi; st_un = md1 pi/2; ed_un = -md2 pi/2; scale = 5; 3 proposals = residual sign -1, 0, +1
__device__ __forceinline__ void s0_hafm(float o_md0, float o_md1, float o_md2, float o_dis, float o_res, int p, float4 (&l)[3]) {
  const float md0 = sigmoidf_(o_md0), md1 = sigmoidf_(o_md1), md2 = sigmoidf_(o_md2), dis = sigmoidf_(o_dis), res = sigmoidf_(o_res);
  const float PI = 3.14159265358979323846f;
  const float md_un = (md0 - 0.5f) * PI * 2.0f, st_un = md1 * PI / 2.0f, ed_un = -md2 * PI / 2.0f;
  const float cs = cosf(md_un), ss = sinf(md_un), yst = tanf(st_un), yed = tanf(ed_un);
  const float x0 = (float)(p & (S0_F - 1)), y0 = (float)(p >> 7);
  const float lim = (float)(S0_F - 1);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float d = fminf(fmaxf(dis + res * (float)(s - 1), 0.f), 1.f);
    const float xs = (cs - ss * yst) * d * 5.0f, ys = (ss + cs * yst) * d * 5.0f;
    const float xe = (cs - ss * yed) * d * 5.0f, ye = (ss + cs * yed) * d * 5.0f;
    l[s].x = fminf(fmaxf(xs + x0, 0.f), lim); l[s].y = fminf(fmaxf(ys + y0, 0.f), lim);
    l[s].z = fminf(fmaxf(xe + x0, 0.f), lim); l[s].w = fminf(fmaxf(ye + y0, 0.f), lim);
  }
}
// jloc = softmax(o5, o6)[1]
__device__ __forceinline__ float s0_jloc(float o5, float o6) {
  const float mx = fmaxf(o5, o6), e0 = expf(o5 - mx), e1 = expf(o6 - mx);
  return e1 / (e0 + e1);
}

// per feature-map pixel: HAFM decoding, junction probability / offset maps, thin / aux CHW (+ pixel-major)
__global__ __launch_bounds__(256) void s0_decode_kernel(const float* __restrict__ head /*[NPX][ld]*/, float* __restrict__ lines_pred /*[3*NPX][4]*/,
                                                        float* __restrict__ jloc /*[NPX]*/, float* __restrict__ joff /*[2][NPX]*/,
                                                        float* __restrict__ thin /*[4][NPX]*/, float* __restrict__ aux /*[4][NPX]*/,
                                                        float* __restrict__ ta8 /*[B][NPX][8] or nullptr*/, size_t stage_stride, int ld,
                                                        int off) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S0_NPX) return;
  const size_t img = blockIdx.y;                 // one image per grid row: head / jloc / joff are dense per image, the rest sits in its stage block
  head += img * S0_NPX * ld; jloc += img * S0_NPX; joff += img * 2 * S0_NPX;
  lines_pred += img * stage_stride; thin += img * stage_stride; aux += img * stage_stride;
  const float* o = head + (size_t)p * ld + off;             // md0..2 dis res | jloc0 jloc1 | joffx joffy | thin0..3 | aux0..3
  const float4 a = *reinterpret_cast<const float4*>(o), b = *reinterpret_cast<const float4*>(o + 4), c4 = *reinterpret_cast<const float4*>(o + 8),
               d4 = *reinterpret_cast<const float4*>(o + 12);
  const float o16 = o[16];
  jloc[p] = s0_jloc(b.y, b.z);
  joff[p] = sigmoidf_(b.w) - 0.5f;
  joff[S0_NPX + p] = sigmoidf_(c4.x) - 0.5f;
  thin[p] = c4.y; thin[S0_NPX + p] = c4.z; thin[2 * S0_NPX + p] = c4.w; thin[3 * S0_NPX + p] = d4.x;
  aux[p] = d4.y; aux[S0_NPX + p] = d4.z; aux[2 * S0_NPX + p] = d4.w; aux[3 * S0_NPX + p] = o16;
  if (ta8) {                                     // the same eight values pixel-major: stage 1 samples a point's 4 + 4 channels as two 16-byte taps
    float4* t8 = reinterpret_cast<float4*>(ta8 + (img * S0_NPX + p) * 8);
    t8[0] = make_float4(c4.y, c4.z, c4.w, d4.x);
    t8[1] = make_float4(d4.y, d4.z, d4.w, o16);
  }
  float4 l[3];
  s0_hafm(a.x, a.y, a.z, a.w, b.x, p, l);
#pragma unroll
  for (int s = 0; s < 3; ++s) *reinterpret_cast<float4*>(lines_pred + ((size_t)s * S0_NPX + p) * 4) = l[s];
}

// The 17-channel 1x1 head AND its decode in one pass over the line features (the batched path): a wave takes 16 pixels at a time, multiplies
// their 128 features by the head's first 64-feature block exactly as gemm_small_kernel does (same packed slabs, same fragments, K in
// ascending 32-wide steps from zero, bias added after the sum: the same bits as every GEMM kernel of this library), and decodes straight
// from the accumulators: lane group g = lane / 16 of pixel l15 holds channels 8g .. 8g+7, so group 0 has md0..2 dis res jloc0 jloc1 (+ joffx)
// and does the HAFM decode + jloc, group 1 has joffy thin0..3 aux0..2 (+ joffx from group 0, aux3 from group 2 by lane shuffles) and writes
// joff + the pixel-major thin | aux.  The separate form moved 537 MB in, 268 MB out (no-LDS GEMM, 2.4 TB/s) and 284 MB in, 260 MB out
// (decode) per 128 images; this one 537 MB in, 193 MB out — the CHW thin / aux planes of the contract are not written (only the
// host-tensor path and the inspection hook read them, and those run the fused head + s0_decode_kernel).
template <class P>
__global__ __launch_bounds__(256) void s0_head_decode_kernel(const uint16_t* __restrict__ X /*[ntiles*16][128]*/, const uint16_t* __restrict__ Wp,
                                                             const float* __restrict__ bias, int ntiles, float* __restrict__ lines_pred,
                                                             float* __restrict__ jloc, float* __restrict__ joff, float* __restrict__ ta8,
                                                             size_t stage_stride) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const char* wbase = reinterpret_cast<const char*>(Wp);
  typename P::vec8 wf[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rr = u * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wf[u][ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(
          wbase + (ks >> 1) * SLAB_BYTES + rr * 128 + ((((ks & 1) * 4 + g) ^ swz128(rr)) << 4)));
  }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = bias[g * 8 + e];
  const int nw = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  typename P::vec8 xn[4];
  auto fetch = [&](int t) {
    const uint16_t* xr = X + ((size_t)t * 16 + l15) * 128 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xn[ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(xr + ks * 32));
  };
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += nw) {
    typename P::vec8 xf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xf[ks] = xn[ks];
    if (tile + nw < ntiles) fetch(tile + nw);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = P::mfma(wf[u][ks], xf[ks], acc[u]);
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = acc[0][e];
      f[4 + e] = acc[1][e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += bv[e];
    const float joffx = __shfl(f[7], l15), aux3 = __shfl(f[0], l15 + 32);
    const int row = tile * 16 + l15, img = row >> 14, p = row & (S0_NPX - 1);
    if (g == 0) {
      float4 l[3];
      s0_hafm(f[0], f[1], f[2], f[3], f[4], p, l);
      float* lp = lines_pred + (size_t)img * stage_stride;
#pragma unroll
      for (int s = 0; s < 3; ++s) *reinterpret_cast<float4*>(lp + ((size_t)s * S0_NPX + p) * 4) = l[s];
      jloc[(size_t)img * S0_NPX + p] = s0_jloc(f[5], f[6]);
    } else if (g == 1) {
      float* jo = joff + (size_t)img * 2 * S0_NPX;
      jo[p] = sigmoidf_(joffx) - 0.5f;
      jo[S0_NPX + p] = sigmoidf_(f[0]) - 0.5f;
      float4* t8 = reinterpret_cast<float4*>(ta8 + ((size_t)img * S0_NPX + p) * 8);
      t8[0] = make_float4(f[1], f[2], f[3], f[4]);
      t8[1] = make_float4(f[5], f[6], f[7], aux3);
    }
  }
}

// non_maximum_suppression: a * (a == max_pool2d(a, 3, stride 1, padding 1))
__global__ __launch_bounds__(256) void s0_jnms_kernel(const float* __restrict__ jloc, float* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S0_NPX) return;
  jloc += (size_t)blockIdx.y * S0_NPX; out += (size_t)blockIdx.y * S0_NPX;
  const int x = p & (S0_F - 1), y = p >> 7;
  const float a = jloc[p];
  float m = a;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx, yy = y + dy;
      if (xx >= 0 && xx < S0_F && yy >= 0 && yy < S0_F) m = fmaxf(m, jloc[yy * S0_F + xx]);
    }
  out[p] = (a == m) ? a : 0.f;
}

// loi_features: NHWC rows of the head GEMM -> the contract's CHW [128][128][128]; 32 pixels x 128 channels per workgroup via LDS
__global__ __launch_bounds__(256) void s0_loi_chw_kernel(const float* __restrict__ head, float* __restrict__ loi) {
  __shared__ float t[32][129];
  const int p0 = blockIdx.x * 32, tid = threadIdx.x;
  for (int i = tid; i < 32 * 128; i += 256) {
    const int px = i >> 7, ch = i & 127;
    t[px][ch] = head[(size_t)(p0 + px) * S0_LD + ch];
  }
  __syncthreads();
  for (int i = tid; i < 32 * 128; i += 256) {
    const int ch = i >> 5, px = i & 31;
    loi[(size_t)ch * S0_NPX + p0 + px] = t[px][ch];
  }
}

// get_junctions: rows (score, x, y) of the top-K selection -> juncs_pred [jn][2] = (x + joff_x + 0.5, y + joff_y + 0.5)
__global__ void s0_juncs_kernel(const float* __restrict__ sel /*[B][sel_cap][259]*/, const int* __restrict__ n_sel /*[B]*/,
                                const float* __restrict__ joff /*[B][2][NPX]*/, float* __restrict__ juncs, int jn, int sel_cap,
                                size_t stage_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= jn) return;
  const size_t img = blockIdx.y;
  sel += img * sel_cap * 259; joff += img * 2 * S0_NPX; juncs += img * stage_stride;
  float x = 0.f, y = 0.f;
  if (i < n_sel[img]) {
    const float fx = sel[(size_t)i * 259 + 1], fy = sel[(size_t)i * 259 + 2];
    const int idx = (int)fy * S0_F + (int)fx;
    x = __fadd_rn(__fadd_rn(fx, joff[idx]), 0.5f);
    y = __fadd_rn(__fadd_rn(fy, joff[S0_NPX + idx]), 0.5f);
  }
  juncs[i * 2] = x;
  juncs[i * 2 + 1] = y;
}

// wireframe_matcher of HAWP (NOT the C++ routine of that name): nearest junction of both endpoints of every proposal (squared
// distance, first minimum), idx_junc_to_end_min / _max, iskeep = (min < max) and both squared distances < j2l threshold (10)
__global__ __launch_bounds__(256) void s0_j2l_kernel(const float* __restrict__ lines_pred, const float* __restrict__ juncs, int jn, int n,
                                                     float thr, float* __restrict__ iskeep, float* __restrict__ imin, float* __restrict__ imax,
                                                     size_t stage_stride) {
  __shared__ float jx[320], jy[320];
  {
    const size_t o = (size_t)blockIdx.y * stage_stride;
    lines_pred += o; juncs += o; iskeep += o; imin += o; imax += o;
  }
  for (int i = threadIdx.x; i < jn; i += 256) { jx[i] = juncs[i * 2]; jy[i] = juncs[i * 2 + 1]; }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float4 l = *reinterpret_cast<const float4*>(lines_pred + (size_t)p * 4);
  float c1 = INFINITY, c2 = INFINITY;
  int i1 = 0, i2 = 0;
  for (int j = 0; j < jn; ++j) {
    const float ax = __fsub_rn(l.x, jx[j]), ay = __fsub_rn(l.y, jy[j]);
    const float bx = __fsub_rn(l.z, jx[j]), by = __fsub_rn(l.w, jy[j]);
    const float d1 = __fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), d2 = __fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by));
    if (d1 < c1) { c1 = d1; i1 = j; }
    if (d2 < c2) { c2 = d2; i2 = j; }
  }
  const int lo = min(i1, i2), hi = max(i1, i2);
  iskeep[p] = (lo < hi && c1 < thr && c2 < thr) ? 1.0f : 0.0f;
  imin[p] = (float)lo;
  imax[p] = (float)hi;
}

// The same match for what the line path CONSUMES, 8x cheaper (the brute-force kernel above — 49152 proposals x 300 junctions —
// was 0.63 ms per 128 images): wireframe_matcher (plnet.cpp:272-307) reads idx_junc_to_end_min / _max only where iskeep > 0, and iskeep
// needs both nearest squared distances below thr (10).  Junctions are binned into J2L_CELL x J2L_CELL-pixel cells (counting sort in LDS,
// per workgroup) and an endpoint looks at the 3 x 3 cells around its own: a junction outside that block is more than J2L_CELL pixels away
// in x or y, so for thr < J2L_CELL^2 iskeep is exact for every proposal and min / max are exact for every KEPT one; entries of proposals
// that are not kept hold the block's nearest (or 0) instead of the global nearest.  Ties go to the lower junction index explicitly (the
// cell order is not the index order), which is the brute-force loop's "first minimum".  The inspection hook keeps the brute-force kernel.
// (8-pixel cells: ~10 candidates per endpoint, 157 us per 128 images; 4-pixel cells: ~2.6.)
constexpr int J2L_PT = 4;                 // proposals per thread (1 .. 16 measured the same: the searches dominate, not the binning)
constexpr int J2L_CELL = 4, J2L_NC = S0_F / J2L_CELL, J2L_CELLS = J2L_NC * J2L_NC, J2L_CPT = J2L_CELLS / 256;
static_assert(J2L_CELLS % 256 == 0, "cells are dealt out evenly over the 256 threads");

__global__ __launch_bounds__(256) void s0_j2l_grid_kernel(const float* __restrict__ lines_pred, const float* __restrict__ juncs, int jn, int n,
                                                          float thr, float* __restrict__ iskeep, float* __restrict__ imin,
                                                          float* __restrict__ imax, int* __restrict__ wg_counts, size_t stage_stride) {
  __shared__ float2 sxy[320];
  __shared__ int sj[320];
  __shared__ int cs[J2L_CELLS + 4], fill[J2L_CELLS];
  __shared__ int wtot[4];
  {
    const size_t o = (size_t)blockIdx.y * stage_stride;
    lines_pred += o; juncs += o; iskeep += o; imin += o; imax += o;
  }
  const int t = threadIdx.x;
  constexpr float INV = 1.0f / J2L_CELL;
  auto cell1 = [](float v) { return min(J2L_NC - 1, max(0, (int)(v * INV))); };
#pragma unroll
  for (int q = 0; q < J2L_CPT; ++q) fill[t * J2L_CPT + q] = 0;
  __syncthreads();
  float2 mine[2];
  int mc[2] = {-1, -1};
  for (int r = 0; r < 2; ++r) {
    const int i = t + r * 256;
    if (i < jn) {
      mine[r] = make_float2(juncs[i * 2], juncs[i * 2 + 1]);
      mc[r] = cell1(mine[r].y) * J2L_NC + cell1(mine[r].x);
      atomicAdd(&fill[mc[r]], 1);
    }
  }
  __syncthreads();
  {                                                          // exclusive prefix of the cell counts: J2L_CPT consecutive cells per thread
    int v[J2L_CPT], tot = 0;
#pragma unroll
    for (int q = 0; q < J2L_CPT; ++q) { v[q] = fill[t * J2L_CPT + q]; tot += v[q]; }
    const int lane = t & 63, wv = t >> 6;
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int off = incl - tot;
    for (int w = 0; w < wv; ++w) off += wtot[w];
#pragma unroll
    for (int q = 0; q < J2L_CPT; ++q) { cs[t * J2L_CPT + q] = off; off += v[q]; fill[t * J2L_CPT + q] = 0; }
    if (t == 255) cs[J2L_CELLS] = off;
  }
  __syncthreads();
  for (int r = 0; r < 2; ++r)
    if (mc[r] >= 0) {
      const int pos = cs[mc[r]] + atomicAdd(&fill[mc[r]], 1);
      sxy[pos] = mine[r];
      sj[pos] = t + r * 256;
    }
  __syncthreads();
  auto nearest = [&](float px, float py, float& best, int& bi) {
    best = INFINITY; bi = 0;
    const int cx = cell1(px), cy = cell1(py);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, J2L_NC - 1);
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, J2L_NC - 1); ++yy)
      for (int i = cs[yy * J2L_NC + x0], e = cs[yy * J2L_NC + x1 + 1]; i < e; ++i) {
        const float2 q = sxy[i];
        const int j = sj[i];
        const float ax = __fsub_rn(px, q.x), ay = __fsub_rn(py, q.y);
        const float d = __fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay));
        if (d < best || (d == best && j < bi)) { best = d; bi = j; }
      }
  };
  int kept = 0;
  for (int r = 0; r < J2L_PT; ++r) {
    const int p = (blockIdx.x * J2L_PT + r) * 256 + t;
    if (p >= n) break;
    const float4 l = *reinterpret_cast<const float4*>(lines_pred + (size_t)p * 4);
    float c1, c2;
    int i1, i2;
    nearest(l.x, l.y, c1, i1);
    nearest(l.z, l.w, c2, i2);
    const int lo = min(i1, i2), hi = max(i1, i2);
    const bool k = lo < hi && c1 < thr && c2 < thr;
    kept += k;
    iskeep[p] = k ? 1.0f : 0.0f;
    imin[p] = (float)lo;
    imax[p] = (float)hi;
  }
  // the kept proposals of this workgroup's run of 256 * J2L_PT: the count wf_count_kernel would take over the same run (launch_wireframe's first launch)
  if (wg_counts) {
    kept = (int)wave_sum((float)kept);
    __syncthreads();                    // (wtot's earlier readers)
    if ((t & 63) == 0) wtot[t >> 6] = kept;
    __syncthreads();
    if (t == 0) wg_counts[(size_t)blockIdx.y * LINE_CNT_LD + 2 + blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  }
}

// B images per launch (grid.y): head [B][NPX][ld] with the 17 decoded channels from column off (the fused 145-channel head: ld 160, off 128;
// the 17-channel head of the batched path: ld 32, off 0), jloc / jnms [B][NPX], joff [B][2][NPX] dense; lines_pred, thin, aux, loi, juncs, iskeep,
// imin, imax are image 0's pointers into its stage block, image b's are stage_stride floats further.  loi != nullptr: the contract's CHW
// copy of IMAGE 0's LOI features out of the fused head's rows (pitch 160).
void launch_s0_decode(const float* head, int ld, int off, float* lines_pred, float* jloc, float* jnms, float* joff, float* thin, float* aux,
                      float* loi, float* ta8, int B, size_t stage_stride, hipStream_t st) {
  hipLaunchKernelGGL(s0_decode_kernel, dim3(S0_NPX / 256, B), dim3(256), 0, st, head, lines_pred, jloc, joff, thin, aux, ta8, stage_stride, ld, off);
  if (jnms) hipLaunchKernelGGL(s0_jnms_kernel, dim3(S0_NPX / 256, B), dim3(256), 0, st, jloc, jnms);      // (nullptr: the caller takes the candidates with launch_candidates_nms3)
  if (loi) hipLaunchKernelGGL(s0_loi_chw_kernel, dim3(S0_NPX / 32), dim3(256), 0, st, head, loi);
}
// X [B * 128*128][128] line features (2-byte), Wp / bias: the packed 17-channel head (its first 64-feature block); jnms = 3x3 NMS of jloc
void launch_s0_head_decode(int prec, const uint16_t* X, const uint16_t* Wp, const float* bias, float* lines_pred, float* jloc, float* jnms,
                           float* joff, float* ta8, int B, size_t stage_stride, hipStream_t st) {
  const int ntiles = B * S0_NPX / 16;
  const int wgs = std::min(ntiles / 4, 256 * 16);
  if (prec == 1)
    hipLaunchKernelGGL(s0_head_decode_kernel<PF16>, dim3(wgs), dim3(256), 0, st, X, Wp, bias, ntiles, lines_pred, jloc, joff, ta8, stage_stride);
  else
    hipLaunchKernelGGL(s0_head_decode_kernel<PBF16>, dim3(wgs), dim3(256), 0, st, X, Wp, bias, ntiles, lines_pred, jloc, joff, ta8, stage_stride);
  if (jnms) hipLaunchKernelGGL(s0_jnms_kernel, dim3(S0_NPX / 256, B), dim3(256), 0, st, jloc, jnms);      // (nullptr: the caller takes the candidates with launch_candidates_nms3)
}
void launch_s0_juncs(const float* sel, const int* n_sel, const float* joff, float* juncs, int jn, int sel_cap, int B, size_t stage_stride,
                     hipStream_t st) {
  hipLaunchKernelGGL(s0_juncs_kernel, dim3((jn + 63) / 64, B), dim3(64), 0, st, sel, n_sel, joff, juncs, jn, sel_cap, stage_stride);
}
bool launch_s0_j2l(const float* lines_pred, const float* juncs, int jn, int n, float thr, float* iskeep, float* imin, float* imax, int* counts,
                   int B, size_t stage_stride, int exact_all, hipStream_t st) {
  if (exact_all || thr >= (float)(J2L_CELL * J2L_CELL - 1) || jn > 320) {
    hipLaunchKernelGGL(s0_j2l_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, lines_pred, juncs, jn, n, thr, iskeep, imin, imax,
                       stage_stride);
    return false;
  }
  const int wgs = (n + 256 * J2L_PT - 1) / (256 * J2L_PT);
  const bool counted = counts && wgs == WF_WGS && n % WF_WGS == 0 && n / WF_WGS == 256 * J2L_PT;      // the wireframe list's runs are this kernel's
  hipLaunchKernelGGL(s0_j2l_grid_kernel, dim3(wgs, B), dim3(256), 0, st, lines_pred, juncs, jn, n, thr, iskeep, imin, imax, counted ? counts : nullptr,
                     stage_stride);
  return counted;
}

}  // namespace airfe
