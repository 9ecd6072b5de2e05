// airfe — matcher kernels shared by LightGlue / SuperGlue: token preparation (+ Fourier positional
// encoding), flash attention on MFMA (d_head = 64), LayerNorm+GELU, similarity GEMM, and the
// LightGlue log-assignment + on-device filter_matches (src/light_glue.cpp:214-266).
#include <float.h>

#include "common.h"
#include "kernels.h"

namespace airfe {

// =============================================================================== prepare
// Builds the residual stream from the 259/258-float feature rows (the reference's process_input,
// src/light_glue.cpp:172-212, plus PointMatcher::NormalizeKeypoints, src/point_matcher.cc:39-48).
template <class P>
__global__ __launch_bounds__(256) void lg_prepare_kernel(LgPrepArgs a) {
  const int s = blockIdx.y, n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= a.Np) return;
  const int Bt = a.f0x ? 2 : a.B;                        // pairs of this call
  if (s >= 2 * Bt) {                                     // the arena's slack rows behind the last sequence: back to zero (see reset_slack_rows)
    const int r = (s - 2 * Bt) * a.Np + n;
    if (r < a.slack_rows) {
      const size_t row = (size_t)2 * Bt * a.Np + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a.x32[row * 256 + lane + 64 * j] = 0.f;
        a.xb[row * 256 + lane + 64 * j] = 0;
      }
    }
    return;
  }
  const int b = s >> 1, side = s & 1;
  const bool second = a.f0x && b == 1;
  const int len = second ? (side ? *a.n1x : *a.n0x) : (side ? a.n1[b] : a.n0[b]);
  if (n == 0 && lane == 0) a.lens[s] = len;
  const size_t row = (size_t)s * a.Np + n;
  float* x32 = a.x32 + row * 256;
  uint16_t* xb = a.xb + row * 256;
  if (n < len) {
    const float* f = second ? (side ? a.f1x : a.f0x) + (size_t)n * a.ld : (side ? a.f1 : a.f0) + ((size_t)b * a.cap + n) * a.ld;
    float kx = f[a.kp_off], ky = f[a.kp_off + 1];
    if (a.normalize) {
      kx = __fmul_rn(__fsub_rn(kx, a.cx), a.linv);
      ky = __fmul_rn(__fsub_rn(ky, a.cy), a.linv);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      const float v = f[a.kp_off + 2 + c];
      x32[c] = v;
      xb[c] = P::from_f32(v);
    }
    if (lane < 32) {
      const float pr = a.wr[lane * 2] * kx + a.wr[lane * 2 + 1] * ky;
      a.rot_cos[row * 32 + lane] = cosf(pr);
      a.rot_sin[row * 32 + lane] = sinf(pr);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      x32[c] = 0.f;
      xb[c] = 0;
    }
    if (lane < 32) {
      a.rot_cos[row * 32 + lane] = 1.f;
      a.rot_sin[row * 32 + lane] = 0.f;
    }
  }
}

void launch_lg_prepare(int prec, const LgPrepArgs& a, hipStream_t st) {
  dim3 grid((a.Np + 3) / 4, 2 * (a.f0x ? 2 : a.B) + (a.slack_rows + a.Np - 1) / a.Np);
  if (prec == 1) hipLaunchKernelGGL(lg_prepare_kernel<PF16>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(lg_prepare_kernel<PBF16>, grid, dim3(256), 0, st, a);
}

// =============================================================================== LayerNorm + GELU
// exact-erf GELU 0.5 y (1 + erf(y / sqrt 2)) with a branch-free erf: Abramowitz & Stegun 7.1.26, |err| <= 1.5e-7
// (far below the 2-byte storage the result is rounded to), instead of libm's branchy erff (9 divergent branches/row).
__device__ __forceinline__ float gelu_exact(float y) {
  const float x = fabsf(y) * 0.70710678118654752f;
  const float t = 1.0f / fmaf(0.3275911f, x, 1.0f);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfa = 1.0f - poly * t * __expf(-x * x);
  return 0.5f * y * (1.0f + copysignf(erfa, y));
}

template <class P>
__global__ __launch_bounds__(256) void ln_gelu_kernel(uint16_t* __restrict__ h, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int M) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  uint4* p = reinterpret_cast<uint4*>(h + (size_t)row * 512) + lane;
  float v[8];
  unpack8<P>(*p, v);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) sum += v[e];
  const float mean = wave_sum(sum) * (1.0f / 512.0f);
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; var += d * d; }
  var = wave_sum(var) * (1.0f / 512.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + lane * 8), g1 = *reinterpret_cast<const float4*>(gamma + lane * 8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + lane * 8), b1 = *reinterpret_cast<const float4*>(beta + lane * 8 + 4);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float y = (v[e] - mean) * rstd * gg[e] + bb[e];
    v[e] = gelu_exact(y);
  }
  *p = pack8<P>(v);
}

void launch_ln_gelu(int prec, uint16_t* h, const float* gamma, const float* beta, int M, hipStream_t st) {
  dim3 grid((M + 3) / 4);
  if (prec == 1) hipLaunchKernelGGL(ln_gelu_kernel<PF16>, grid, dim3(256), 0, st, h, gamma, beta, M);
  else hipLaunchKernelGGL(ln_gelu_kernel<PBF16>, grid, dim3(256), 0, st, h, gamma, beta, M);
}

// =============================================================================== matchability
__device__ __forceinline__ float logsigmoidf(float z) { return fminf(z, 0.f) - log1pf(expf(-fabsf(z))); }

// z[row] = logsigmoid(w . x + b): the assignment only ever uses the matchability through its log-sigmoid, so it is taken
// once per token here instead of once per (row, column) of the score matrix
__global__ void rowdot256_kernel(const float* __restrict__ x32, const float* __restrict__ w, float b,
                                 float* __restrict__ z, int M) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4 x = *(reinterpret_cast<const float4*>(x32 + (size_t)row * 256) + lane);
  const float4 ww = *(reinterpret_cast<const float4*>(w) + lane);
  const float d = wave_sum(x.x * ww.x + x.y * ww.y + x.z * ww.z + x.w * ww.w);
  if (lane == 0) z[row] = logsigmoidf(d + b);
}

void launch_rowdot256(const float* x32, const float* w, float b, float* z, int M, hipStream_t st) {
  hipLaunchKernelGGL(rowdot256_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x32, w, b, z, M);
}

// =============================================================================== similarity
// sim[b][i][j] = md[2b][i] . md[2b+1][j], K = 256; each wave a 64 x 64 tile, fragments straight from L2: 64 KB of operands per 128
// MFMAs (the first version's 16 x 64 tiles read 40 KB per 32 MFMAs — 7 MB per pair for 400 KB of data, 60 us per 64 pairs).
// Same K order per output as before: same bits.
template <class P>
__global__ __launch_bounds__(256) void sim_kernel(const uint16_t* __restrict__ md, float* __restrict__ sim, int Np) {
  const int b = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int i0 = (blockIdx.x * 4 + wave) * 64, j0 = blockIdx.y * 64;
  if (i0 >= Np) return;
  const uint16_t* A = md + ((size_t)(2 * b) * Np + i0 + l15) * 256;
  const uint16_t* Bm = md + ((size_t)(2 * b + 1) * Np + j0 + l15) * 256;
  f32x4 acc[4][4];
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    typename P::vec8 af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {                        // Np is a multiple of 16, not of 64: whole 16-row / 16-column sub-tiles drop out
      af[t] = (i0 + t * 16 < Np) ? __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(A + (size_t)t * 16 * 256 + ks * 32 + g * 8)) : typename P::vec8{};
      bf[t] = (j0 + t * 16 < Np) ? __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(Bm + (size_t)t * 16 * 256 + ks * 32 + g * 8)) : typename P::vec8{};
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[it][jt] = P::mfma(af[it], bf[jt], acc[it][jt]);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (i0 + it * 16 >= Np) continue;
    float* out = sim + ((size_t)b * Np + i0 + it * 16) * Np + j0;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if (j0 + jt * 16 >= Np) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(g * 4 + r) * Np + jt * 16 + l15] = acc[it][jt][r];
    }
  }
}

void launch_sim(int prec, const uint16_t* md, float* sim, int B, int Np, hipStream_t st) {
  dim3 grid((Np + 255) / 256, (Np + 63) / 64, B);
  if (prec == 1) hipLaunchKernelGGL(sim_kernel<PF16>, grid, dim3(256), 0, st, md, sim, Np);
  else hipLaunchKernelGGL(sim_kernel<PBF16>, grid, dim3(256), 0, st, md, sim, Np);
}

// =============================================================================== assignment + filter

// wave per row: log-sum-exp over j < n1
__device__ __forceinline__ void lg_rowlse_body(int bx, const float* __restrict__ sim, const int* __restrict__ lens, int Np,
                                               float* __restrict__ rowlse) {
  const int b = blockIdx.y, i = bx * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (i >= n0) return;
  const float* r = sim + ((size_t)b * Np + i) * Np;
  float mx = -INFINITY;
  for (int j = lane; j < n1; j += 64) mx = fmaxf(mx, r[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < n1; j += 64) s += expf(r[j] - mx);
  s = wave_sum(s);
  if (lane == 0) rowlse[(size_t)b * Np + i] = mx + logf(s);
}

// 64 columns x 16 row slices per workgroup: log-sum-exp over i < n0 (coalesced across the block's columns; a single thread per
// column walked its 400 rows as one dependent chain of L2 loads and took longer than the similarity GEMM; four slices: 54 us)
constexpr int LG_CS = 16;                     // row slices of the column kernels
constexpr int LG_MERGE_MAX_B = 8;             // up to this many pairs the row and the column kernel of a stage share a launch
__device__ __forceinline__ void lg_collse_body(int bx, const float* __restrict__ sim, const int* __restrict__ lens, int Np,
                                               float* __restrict__ collse) {
  __shared__ float part[LG_CS][64];
  const int b = blockIdx.y, jj = threadIdx.x & 63, q = threadIdx.x >> 6, j = bx * 64 + jj;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const bool live = j < n1;
  const float* c = sim + (size_t)b * Np * Np + j;
  float mx = -INFINITY;
  if (live)
    for (int i = q; i < n0; i += LG_CS) mx = fmaxf(mx, c[(size_t)i * Np]);
  part[q][jj] = mx;
  __syncthreads();
  mx = part[0][jj];
#pragma unroll
  for (int k = 1; k < LG_CS; ++k) mx = fmaxf(mx, part[k][jj]);
  __syncthreads();
  float sm = 0.f;
  if (live)
    for (int i = q; i < n0; i += LG_CS) sm += expf(c[(size_t)i * Np] - mx);
  part[q][jj] = sm;
  __syncthreads();
  if (q == 0 && live) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < LG_CS; ++k) tot += part[k][jj];
    collse[(size_t)b * Np + j] = mx + logf(tot);
  }
}
__global__ void lg_rowlse_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, float* __restrict__ rowlse) {
  lg_rowlse_body(blockIdx.x, sim, lens, Np, rowlse);
}
__global__ __launch_bounds__(64 * LG_CS) void lg_collse_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, float* __restrict__ collse) {
  lg_collse_body(blockIdx.x, sim, lens, Np, collse);
}
// row and column log-sum-exp as ONE launch for small batches (round 4: a dependent launch costs ~4.7 us whatever it does, and the batch-1 matcher has 48 of them):
// workgroups [0, nrb) take 16 rows each (a wave per row), the rest 64 columns each.  Same bodies as the two kernels above, which large batches keep
// (there the four-row workgroups fill the chip better: 0.13 against 0.17 ms per 64 pairs).
__global__ __launch_bounds__(64 * LG_CS) void lg_lse_kernel(const float* __restrict__ sim, const int* __restrict__ lens, int Np, int nrb,
                                                            float* __restrict__ rowlse, float* __restrict__ collse) {
  if ((int)blockIdx.x < nrb) lg_rowlse_body(blockIdx.x, sim, lens, Np, rowlse);
  else lg_collse_body(blockIdx.x - nrb, sim, lens, Np, collse);
}

__device__ __forceinline__ float lg_score(float sv, float rl, float cl, float c0, float c1) {
  return ((sv - rl) + (sv - cl)) + (c0 + c1);
}

// wave per row: scores (optionally materialised) + row arg-max, first maximum wins
__device__ __forceinline__ void lg_rowarg_body(int bx, const float* __restrict__ sim, const float* __restrict__ z, const int* __restrict__ lens,
                                               int Np, const float* __restrict__ rowlse, const float* __restrict__ collse,
                                               float* __restrict__ scores_out, int* __restrict__ rowarg, float* __restrict__ rowval) {
  const int b = blockIdx.y, i = bx * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  if (i >= n0) return;
  const float* r = sim + ((size_t)b * Np + i) * Np;
  const float rl = rowlse ? rowlse[(size_t)b * Np + i] : 0.f;
  const float c0 = rowlse ? z[(size_t)(2 * b) * Np + i] : 0.f;
  const float* z1 = z + (size_t)(2 * b + 1) * Np;
  const float* cl = collse + (size_t)b * Np;
  float best = -FLT_MAX;                       // light_glue.cpp:219: strict '>' from -FLT_MAX
  int bj = 0x7FFFFFFF;
  for (int j = lane; j < n1; j += 64) {
    const float sc = rowlse ? lg_score(r[j], rl, cl[j], c0, z1[j]) : r[j];      // rowlse == nullptr: `sim` already holds scores
    if (scores_out) scores_out[((size_t)b * Np + i) * Np + j] = sc;
    if (sc > best) { best = sc; bj = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oj = __shfl_xor(bj, o);
    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  if (lane == 0) {
    // nothing above -FLT_MAX: row_max[row] keeps its value-initialised pair (0, 0.0f) (std::vector::resize, light_glue.cpp:217)
    rowarg[(size_t)b * Np + i] = (bj == 0x7FFFFFFF) ? 0 : bj;
    rowval[(size_t)b * Np + i] = (bj == 0x7FFFFFFF) ? 0.f : best;
  }
}

// 64 columns x 4 row slices per workgroup: column arg-max over rows, first maximum wins (strict '>' inside a slice, lowest row
// index between slices); z holds log-sigmoid matchabilities.
__device__ __forceinline__ void lg_colarg_body(int bx, const float* __restrict__ sim, const float* __restrict__ z,
                                               const int* __restrict__ lens, int Np, const float* __restrict__ rowlse,
                                               const float* __restrict__ collse, int* __restrict__ colarg) {
  __shared__ float pbest[64 * LG_CS];
  __shared__ int pidx[64 * LG_CS];
  const int b = blockIdx.y, jj = threadIdx.x & 63, q = threadIdx.x >> 6, j = bx * 64 + jj;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const float* z0 = z + (size_t)(2 * b) * Np;
  const bool live = j < n1;
  float best = -FLT_MAX;
  int bi = 0x7FFFFFFF;
  if (live) {
    const float* c = sim + (size_t)b * Np * Np + j;
    const float cl = rowlse ? collse[(size_t)b * Np + j] : 0.f;
    const float c1 = rowlse ? z[(size_t)(2 * b + 1) * Np + j] : 0.f;
    const float* rl = rowlse + (size_t)b * Np;
    for (int i = q; i < n0; i += LG_CS) {
      const float sc = rowlse ? lg_score(c[(size_t)i * Np], rl[i], cl, z0[i], c1) : c[(size_t)i * Np];
      if (sc > best) { best = sc; bi = i; }
    }
  }
  pbest[q * 64 + jj] = best;
  pidx[q * 64 + jj] = bi;
  __syncthreads();
  if (q == 0 && live) {
#pragma unroll
    for (int k = 1; k < LG_CS; ++k) {
      const float ob = pbest[k * 64 + jj];
      const int oi = pidx[k * 64 + jj];
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    colarg[(size_t)b * Np + j] = (bi == 0x7FFFFFFF) ? 0 : bi;
  }
}
__global__ void lg_rowarg_kernel(const float* __restrict__ sim, const float* __restrict__ z, const int* __restrict__ lens, int Np,
                                 const float* __restrict__ rowlse, const float* __restrict__ collse, float* __restrict__ scores_out,
                                 int* __restrict__ rowarg, float* __restrict__ rowval) {
  lg_rowarg_body(blockIdx.x, sim, z, lens, Np, rowlse, collse, scores_out, rowarg, rowval);
}
__global__ __launch_bounds__(64 * LG_CS) void lg_colarg_kernel(const float* __restrict__ sim, const float* __restrict__ z, const int* __restrict__ lens,
                                                               int Np, const float* __restrict__ rowlse, const float* __restrict__ collse,
                                                               int* __restrict__ colarg) {
  lg_colarg_body(blockIdx.x, sim, z, lens, Np, rowlse, collse, colarg);
}
// row and column arg-max as one launch (see lg_lse_kernel)
__global__ __launch_bounds__(64 * LG_CS) void lg_arg_kernel(const float* __restrict__ sim, const float* __restrict__ z, const int* __restrict__ lens,
                                                            int Np, int nrb, const float* __restrict__ rowlse, const float* __restrict__ collse,
                                                            float* __restrict__ scores_out, int* __restrict__ rowarg, float* __restrict__ rowval,
                                                            int* __restrict__ colarg) {
  if ((int)blockIdx.x < nrb) lg_rowarg_body(blockIdx.x, sim, z, lens, Np, rowlse, collse, scores_out, rowarg, rowval);
  else lg_colarg_body(blockIdx.x - nrb, sim, z, lens, Np, rowlse, collse, colarg);
}

// one 1024-thread workgroup per pair: mutual check + exp(score) > thr, ordered compaction (ascending row)
__global__ __launch_bounds__(1024) void lg_filter_kernel(const int* __restrict__ lens, int Np, int cap, float thr,
                                                         const int* __restrict__ rowarg, const float* __restrict__ rowval,
                                                         const int* __restrict__ colarg, int32_t* __restrict__ idx,
                                                         float* __restrict__ score, int* __restrict__ nmatch) {
  __shared__ unsigned wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = lens[2 * b];
  bool ok = false;
  int col = 0;
  float e = 0.f;
  if (tid < n0) {
    col = rowarg[(size_t)b * Np + tid];
    e = expf_like_glibc(rowval[(size_t)b * Np + tid]);
    ok = (colarg[(size_t)b * Np + col] == tid) && (e > thr);
  }
  const unsigned long long bal = __ballot(ok);
  const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  unsigned off = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wv) off += wsum[w];
    tot += wsum[w];
  }
  const unsigned slot = off + before;
  if (ok && slot < (unsigned)cap) {
    idx[((size_t)b * cap + slot) * 2] = tid;
    idx[((size_t)b * cap + slot) * 2 + 1] = col;
    score[(size_t)b * cap + slot] = e;
  }
  if (tid == 0) nmatch[b] = min((int)tot, cap);
}

void launch_lg_assign(const float* sim, const float* z, const int* lens, int B, int Np, int cap, float thr, float* rowlse,
                      float* collse, float* scores_out, int* rowarg, float* rowval, int* colarg, int32_t* idx,
                      float* score, int* nmatch, hipStream_t st) {
  if (B <= LG_MERGE_MAX_B) {           // the batch-1 .. batch-8 calls of the SLAM loop: two launches instead of four
    const int nrb = (Np + LG_CS - 1) / LG_CS, ncb = (Np + 63) / 64;     // LG_CS waves per workgroup: a row each / 64 columns per workgroup
    hipLaunchKernelGGL(lg_lse_kernel, dim3(nrb + ncb, B), dim3(64 * LG_CS), 0, st, sim, lens, Np, nrb, rowlse, collse);
    hipLaunchKernelGGL(lg_arg_kernel, dim3(nrb + ncb, B), dim3(64 * LG_CS), 0, st, sim, z, lens, Np, nrb, rowlse, collse, scores_out, rowarg, rowval,
                       colarg);
  } else {
    hipLaunchKernelGGL(lg_rowlse_kernel, dim3((Np + 3) / 4, B), dim3(256), 0, st, sim, lens, Np, rowlse);
    hipLaunchKernelGGL(lg_collse_kernel, dim3((Np + 63) / 64, B), dim3(64 * LG_CS), 0, st, sim, lens, Np, collse);
    hipLaunchKernelGGL(lg_rowarg_kernel, dim3((Np + 3) / 4, B), dim3(256), 0, st, sim, z, lens, Np, rowlse, collse, scores_out, rowarg, rowval);
    hipLaunchKernelGGL(lg_colarg_kernel, dim3((Np + 63) / 64, B), dim3(64 * LG_CS), 0, st, sim, z, lens, Np, rowlse, collse, colarg);
  }
  hipLaunchKernelGGL(lg_filter_kernel, dim3(B), dim3(1024), 0, st, lens, Np, cap, thr, rowarg, rowval, colarg, idx, score,
                     nmatch);
}

// filter_matches alone on finished score matrices [B][Np][Np] (test hook: hand-built ties, -inf, threshold-exact values)
void launch_lg_filter_scores(const float* scores, const int* lens, int B, int Np, int cap, float thr, int* rowarg, float* rowval,
                             int* colarg, int32_t* idx, float* score, int* nmatch, hipStream_t st) {
  const int nrb = (Np + LG_CS - 1) / LG_CS, ncb = (Np + 63) / 64;
  hipLaunchKernelGGL(lg_arg_kernel, dim3(nrb + ncb, B), dim3(64 * LG_CS), 0, st, scores, (const float*)nullptr, lens, Np, nrb, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr, rowarg, rowval, colarg);
  hipLaunchKernelGGL(lg_filter_kernel, dim3(B), dim3(1024), 0, st, lens, Np, cap, thr, rowarg, rowval, colarg, idx, score, nmatch);
}

// =============================================================================== assignment without the similarity matrix (round 5)
// The round-2 tail wrote sim [B][Np][Np] (41 MB per 64 pairs) and read it back FOUR times (row / column log-sum-exp, row / column arg-max: 243 MB per step,
// profiles/r04_hbm_traffic.txt).  Here the 64 x 64 similarity tile a wave holds in its MFMA accumulators is reduced where it is:
//   launch 1 (lg_sim_lse_kernel):  per tile, per row the (max, sum exp) over the tile's valid columns and per column over its valid rows -> partials
//                                  [B][2][nT][Np] float2 (nT = tiles per side: 7 at Np = 400) — 2.9 MB instead of 41;
//   launch 2 (lg_sim_arg_kernel):  the tile AGAIN (same fragments, same K order: the same bits — 256-deep products from L2-resident descriptors are cheaper than
//                                  a trip through HBM), the tiles' partials folded into rowlse / collse by each wave for its own 64 rows and columns, the
//                                  log-assignment score of light_glue.cpp's engine tail, and per tile the first maximum of every row and column -> [B][2][nT][Np];
//   launch 3 (lg_filter_fused_kernel): per pair the nT partial maxima folded (ties -> lowest index, like the reference's strict '>' scan), then filter_matches
//                                  (src/light_glue.cpp:214-266) exactly as lg_filter_kernel does it.
// Three dependent launches where large batches had six and small ones four.  The log-sum-exp is now a sum of per-tile sums instead of one 64-strided sum per
// wave, and its exponentials are the hardware's (v_exp_f32 through __expf: 128 per lane and tile): scores move in the last bits against the round-2 form (bounded by the tests' 0.05 gate against the fp32 oracle; match sets identical), and they are
// reproducible run to run (fixed order everywhere).  `scores_out` / `sim_out` (inspection hooks, the trace) are written only when asked for.
struct LgTile {                                  // what both launches need of a tile: its accumulators and where it sits
  f32x4 acc[4][4];
  int i0, j0, b, lane, l15, g;
};

template <class P>
__device__ __forceinline__ bool lg_tile_sim(const uint16_t* __restrict__ md, int Np, LgTile& t) {
  t.b = blockIdx.z; t.lane = threadIdx.x & 63; t.l15 = t.lane & 15; t.g = t.lane >> 4;
  const int wave = threadIdx.x >> 6;
  t.i0 = (blockIdx.x * 4 + wave) * 64; t.j0 = blockIdx.y * 64;
  if (t.i0 >= Np) return false;
  const uint16_t* A = md + ((size_t)(2 * t.b) * Np + t.i0 + t.l15) * 256;
  const uint16_t* Bm = md + ((size_t)(2 * t.b + 1) * Np + t.j0 + t.l15) * 256;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) t.acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {               // (the loop of sim_kernel, operand for operand: the two launches and the round-2 kernel agree bit for bit)
    typename P::vec8 af[4], bf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      af[q] = (t.i0 + q * 16 < Np) ? __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(A + (size_t)q * 16 * 256 + ks * 32 + t.g * 8)) : typename P::vec8{};
      bf[q] = (t.j0 + q * 16 < Np) ? __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(Bm + (size_t)q * 16 * 256 + ks * 32 + t.g * 8)) : typename P::vec8{};
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) t.acc[it][jt] = P::mfma(af[it], bf[jt], t.acc[it][jt]);
  }
  return true;
}
// lane (l15, g) holds rows it * 16 + g * 4 + r and columns jt * 16 + l15 of the tile: a row lives in the 16 lanes of one g, a column in the 4 lanes l15 + 16 g'.
// Reductions are TRANSPOSING butterflies: with 16 per-row partials in each of a group's 16 lanes, step 1 exchanges 8 of them with lane ^ 8, step 2 four with lane ^ 4,
// ... — 15 exchanges instead of the 64 of a value-by-value butterfly (the first build of these kernels: 0.17 ms per 64 pairs against the matrix form's 0.14,
// profiles/r05_assign_ab.txt), and the group ends with lane l15 holding row (l15 >> 2, l15 & 3) complete: all 64 lanes store one row each.  Columns: 4 per lane, two
// steps across the four g's, lane g ends with column jt = g.
struct MS { float m, s; };                                   // a partial log-sum-exp: max and sum of exp(v - max)
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
  const float m = fmaxf(a.m, b.m);
  // (an empty side has m = -inf, s = 0: exp(-inf - m) = 0 unless both are empty, where 0 * exp(nan) must stay 0)
  const float ea = a.s > 0.f ? a.s * __expf(a.m - m) : 0.f, eb = b.s > 0.f ? b.s * __expf(b.m - m) : 0.f;
  return MS{m, ea + eb};
}
struct BA { float v; int i; };                               // a partial first maximum: value and index (ties -> lower index)
__device__ __forceinline__ BA ba_merge(BA a, BA b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ MS xchg(MS x, int mask) { return MS{__shfl_xor(x.m, mask), __shfl_xor(x.s, mask)}; }
__device__ __forceinline__ BA xchg(BA x, int mask) { return BA{__shfl_xor(x.v, mask), __shfl_xor(x.i, mask)}; }
__device__ __forceinline__ MS merge(MS a, MS b) { return ms_merge(a, b); }
__device__ __forceinline__ BA merge(BA a, BA b) { return ba_merge(a, b); }
// v[16] (index rho = it * 4 + r) in each of a group's 16 lanes -> the complete reduction of rho = l15 in lane l15
// (one template instance per step: every register index is a compile-time constant — a runtime `half` put the arrays into scratch)
template <int HALF, class T>
__device__ __forceinline__ void reduce_step(T (&v)[16], int l15) {
  const bool up = (l15 & HALF) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const T send = up ? v[k] : v[k + HALF], keep = up ? v[k + HALF] : v[k];
    v[k] = merge(keep, xchg(send, HALF));
  }
}
template <class T>
__device__ __forceinline__ T reduce_rows16(T (&v)[16], int l15) {
  reduce_step<8>(v, l15);
  reduce_step<4>(v, l15);
  reduce_step<2>(v, l15);
  reduce_step<1>(v, l15);
  return v[0];
}
// v[4] (index jt) in each of the 4 lanes l15 + 16 g -> the complete reduction of jt = g in lane g
template <class T>
__device__ __forceinline__ T reduce_cols4(T (&v)[4], int g) {
  {
    const bool up = (g & 2) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const T send = up ? v[k] : v[k + 2], keep = up ? v[k + 2] : v[k];
      v[k] = merge(keep, xchg(send, 32));
    }
  }
  const bool up = (g & 1) != 0;
  const T send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
  return merge(keep, xchg(send, 16));
}

template <class P>
__global__ __launch_bounds__(256, 2) void lg_sim_lse_kernel(const uint16_t* __restrict__ md, const int* __restrict__ lens, int Np, int nT,
                                                         float2* __restrict__ part, float* __restrict__ sim_out) {
  LgTile t;
  if (!lg_tile_sim<P>(md, Np, t)) return;
  const int n0 = lens[2 * t.b], n1 = lens[2 * t.b + 1];
  float2* rpart = part + (((size_t)t.b * 2 + 0) * nT + blockIdx.y) * Np;              // row partials of tile column blockIdx.y
  float2* cpart = part + (((size_t)t.b * 2 + 1) * nT + (t.i0 >> 6)) * Np;             // column partials of tile row i0 / 64
  bool cv[4], rv[16];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) cv[jt] = t.j0 + jt * 16 + t.l15 < n1;
#pragma unroll
  for (int q = 0; q < 16; ++q) rv[q] = t.i0 + (q >> 2) * 16 + t.g * 4 + (q & 3) < n0;
  MS rows[16], cols[4];
#pragma unroll
  for (int q = 0; q < 16; ++q) {                             // this lane's 4 columns of row q
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) mx = cv[jt] ? fmaxf(mx, t.acc[q >> 2][jt][q & 3]) : mx;
    float sm = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) sm += cv[jt] ? __expf(t.acc[q >> 2][jt][q & 3] - mx) : 0.f;
    rows[q] = MS{mx, sm};
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {                           // this lane's 16 rows of column jt
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) mx = rv[q] ? fmaxf(mx, t.acc[q >> 2][jt][q & 3]) : mx;
    float sm = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sm += rv[q] ? __expf(t.acc[q >> 2][jt][q & 3] - mx) : 0.f;
    cols[jt] = MS{mx, sm};
  }
  const MS rr = reduce_rows16(rows, t.l15);
  const MS cc = reduce_cols4(cols, t.g);
  const int i = t.i0 + (t.l15 >> 2) * 16 + t.g * 4 + (t.l15 & 3), j = t.j0 + t.g * 16 + t.l15;
  if (i < n0) rpart[i] = make_float2(rr.m, rr.s);
  if (j < n1) cpart[j] = make_float2(cc.m, cc.s);
  if (sim_out) {                                   // (the trace and the round-2 consumers: the matrix as sim_kernel writes it)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (t.i0 + it * 16 >= Np) continue;
      float* out = sim_out + ((size_t)t.b * Np + t.i0 + it * 16) * Np + t.j0;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        if (t.j0 + jt * 16 >= Np) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(t.g * 4 + r) * Np + jt * 16 + t.l15] = t.acc[it][jt][r];
      }
    }
  }
}

// the partial (max, sum exp) of one row / column over its `ntile` valid tiles -> its log-sum-exp (tiles in ascending order; an empty set gives -inf like the round-2 form)
__device__ __forceinline__ float lg_fold_lse(const float2* __restrict__ p, int Np, int k, int ntile) {
  MS a{-INFINITY, 0.f};
  for (int q = 0; q < ntile; ++q) {
    const float2 v = p[(size_t)q * Np + k];
    a = ms_merge(a, MS{v.x, v.y});
  }
  return a.m + logf(a.s);
}

template <class P>
__global__ __launch_bounds__(256, 2) void lg_sim_arg_kernel(const uint16_t* __restrict__ md, const float* __restrict__ z, const int* __restrict__ lens, int Np, int nT,
                                                         const float2* __restrict__ part, float* __restrict__ rowlse, float* __restrict__ collse,
                                                         float2* __restrict__ argpart, float* __restrict__ scores_out) {
  __shared__ __attribute__((aligned(16))) float sv[4][4][64];         // per wave: rowlse, z0 of its 64 rows; collse, z1 of its 64 columns
  LgTile t;
  if (!lg_tile_sim<P>(md, Np, t)) return;
  const int n0 = lens[2 * t.b], n1 = lens[2 * t.b + 1], wave = threadIdx.x >> 6;
  // log-sum-exp of this wave's 64 rows (over ALL columns < n1) and 64 columns (over all rows < n0): lane l folds row i0 + l and column j0 + l, the wave's LDS block
  // hands them to the lanes that need them (wave-private: no workgroup barrier, a wave whose rows lie beyond Np has left already)
  const int tr = (n0 + 63) >> 6, tc = (n1 + 63) >> 6;                // tiles that hold a valid row / column
  float rl_own = 0.f, cl_own = 0.f;
  if (t.i0 + t.lane < n0) rl_own = lg_fold_lse(part + ((size_t)t.b * 2 + 0) * nT * Np, Np, t.i0 + t.lane, tc);
  if (t.j0 + t.lane < n1) cl_own = lg_fold_lse(part + ((size_t)t.b * 2 + 1) * nT * Np, Np, t.j0 + t.lane, tr);
  if (blockIdx.y == 0 && t.i0 + t.lane < n0) rowlse[(size_t)t.b * Np + t.i0 + t.lane] = rl_own;         // (kept for the trace / inspection: one writer per value)
  if (t.i0 == 0 && t.j0 + t.lane < n1) collse[(size_t)t.b * Np + t.j0 + t.lane] = cl_own;
  sv[wave][0][t.lane] = rl_own;
  sv[wave][1][t.lane] = (t.i0 + t.lane < n0) ? z[(size_t)(2 * t.b) * Np + t.i0 + t.lane] : 0.f;
  sv[wave][2][t.lane] = cl_own;
  sv[wave][3][t.lane] = (t.j0 + t.lane < n1) ? z[(size_t)(2 * t.b + 1) * Np + t.j0 + t.lane] : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float rl[16], c0[16], cl[4], c1[4];
  bool cv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const float4 a = *reinterpret_cast<const float4*>(&sv[wave][0][it * 16 + t.g * 4]), b = *reinterpret_cast<const float4*>(&sv[wave][1][it * 16 + t.g * 4]);
    rl[4 * it] = a.x; rl[4 * it + 1] = a.y; rl[4 * it + 2] = a.z; rl[4 * it + 3] = a.w;
    c0[4 * it] = b.x; c0[4 * it + 1] = b.y; c0[4 * it + 2] = b.z; c0[4 * it + 3] = b.w;
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    cl[jt] = sv[wave][2][jt * 16 + t.l15];
    c1[jt] = sv[wave][3][jt * 16 + t.l15];
    cv[jt] = t.j0 + jt * 16 + t.l15 < n1;
  }
  BA rows[16], cols[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) cols[jt] = BA{-FLT_MAX, 0x7FFFFFFF};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int it = q >> 2, r = q & 3, i = t.i0 + it * 16 + t.g * 4 + r;
    const bool rv = i < n0;
    BA best{-FLT_MAX, 0x7FFFFFFF};                 // light_glue.cpp:219: strict '>' from -FLT_MAX, columns ascending
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const float sc = lg_score(t.acc[it][jt][r], rl[q], cl[jt], c0[q], c1[jt]);
      const int j = t.j0 + jt * 16 + t.l15;
      if (rv && cv[jt]) {
        if (scores_out) scores_out[((size_t)t.b * Np + i) * Np + j] = sc;
        if (sc > best.v) best = BA{sc, j};
        if (sc > cols[jt].v) cols[jt] = BA{sc, i};                   // rows ascend with q for a fixed g
      }
    }
    rows[q] = best;
  }
  const BA rr = reduce_rows16(rows, t.l15);        // first maximum of a row over the tile's 64 columns: lane l15 ends with row (l15 >> 2, l15 & 3)
  const BA cc = reduce_cols4(cols, t.g);           // ... of a column over the tile's 64 rows: lane g ends with column jt = g
  float2* rarg = argpart + (((size_t)t.b * 2 + 0) * nT + blockIdx.y) * Np;
  float2* carg = argpart + (((size_t)t.b * 2 + 1) * nT + (t.i0 >> 6)) * Np;
  const int i = t.i0 + (t.l15 >> 2) * 16 + t.g * 4 + (t.l15 & 3), j = t.j0 + t.g * 16 + t.l15;
  if (i < n0) rarg[i] = make_float2(rr.v, __int_as_float(rr.i));
  if (j < n1) carg[j] = make_float2(cc.v, __int_as_float(cc.i));
}

// one 1024-thread workgroup per pair: the tiles' partial maxima folded (first maximum: larger value, then lower index), then lg_filter_kernel's body
__global__ __launch_bounds__(1024) void lg_filter_fused_kernel(const int* __restrict__ lens, int Np, int nT, int cap, float thr, const float2* __restrict__ argpart,
                                                               int* __restrict__ rowarg, float* __restrict__ rowval, int* __restrict__ colarg,
                                                               int32_t* __restrict__ idx, float* __restrict__ score, int* __restrict__ nmatch) {
  __shared__ unsigned wsum[16];
  __shared__ int scol[1024];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = lens[2 * b], n1 = lens[2 * b + 1];
  const int tr = (n0 + 63) >> 6, tc = (n1 + 63) >> 6;
  auto fold = [&](int side, int k, int ntile, float& best, int& bi) {
    best = -FLT_MAX; bi = 0x7FFFFFFF;
    const float2* p = argpart + ((size_t)b * 2 + side) * nT * Np;
    for (int q = 0; q < ntile; ++q) {
      const float2 v = p[(size_t)q * Np + k];
      const int oi = __float_as_int(v.y);
      if (v.x > best || (v.x == best && oi < bi)) { best = v.x; bi = oi; }
    }
  };
  if (tid < n1) {
    float cb; int ci;
    fold(1, tid, tr, cb, ci);
    scol[tid] = (ci == 0x7FFFFFFF) ? 0 : ci;
    colarg[(size_t)b * Np + tid] = scol[tid];
  }
  bool ok = false;
  int col = 0;
  float e = 0.f;
  float rb = 0.f; int rj = 0x7FFFFFFF;
  if (tid < n0) {
    fold(0, tid, tc, rb, rj);
    // nothing above -FLT_MAX: row_max[row] keeps its value-initialised pair (0, 0.0f) (std::vector::resize, light_glue.cpp:217)
    col = (rj == 0x7FFFFFFF) ? 0 : rj;
    const float val = (rj == 0x7FFFFFFF) ? 0.f : rb;
    rowarg[(size_t)b * Np + tid] = col;
    rowval[(size_t)b * Np + tid] = val;
    e = expf_like_glibc(val);
  }
  __syncthreads();
  if (tid < n0 && n1 > 0) ok = (scol[col] == tid) && (e > thr);      // (col < n1 whenever n1 > 0; an empty second image has no matches: point_matcher.cc:53-55)
  const unsigned long long bal = __ballot(ok);
  const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  unsigned off = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wv) off += wsum[w];
    tot += wsum[w];
  }
  const unsigned slot = off + before;
  if (ok && slot < (unsigned)cap) {
    idx[((size_t)b * cap + slot) * 2] = tid;
    idx[((size_t)b * cap + slot) * 2 + 1] = col;
    score[(size_t)b * cap + slot] = e;
  }
  if (tid == 0) nmatch[b] = min((int)tot, cap);
}

size_t lg_assign_part_floats(int B, int Np) { return (size_t)B * 2 * ((Np + 63) / 64) * Np * 2; }      // floats of ONE partial array (float2 per entry)

void launch_lg_assign_fused(int prec, const uint16_t* md, const float* z, const int* lens, int B, int Np, int cap, float thr, float* part, float* argpart,
                            float* rowlse, float* collse, float* sim_out, float* scores_out, int* rowarg, float* rowval, int* colarg, int32_t* idx,
                            float* score, int* nmatch, hipStream_t st) {
  const int nT = (Np + 63) / 64;
  dim3 grid((Np + 255) / 256, nT, B);
  float2* p2 = reinterpret_cast<float2*>(part);
  float2* a2 = reinterpret_cast<float2*>(argpart);
  if (prec == 1) {
    hipLaunchKernelGGL(lg_sim_lse_kernel<PF16>, grid, dim3(256), 0, st, md, lens, Np, nT, p2, sim_out);
    hipLaunchKernelGGL(lg_sim_arg_kernel<PF16>, grid, dim3(256), 0, st, md, z, lens, Np, nT, p2, rowlse, collse, a2, scores_out);
  } else {
    hipLaunchKernelGGL(lg_sim_lse_kernel<PBF16>, grid, dim3(256), 0, st, md, lens, Np, nT, p2, sim_out);
    hipLaunchKernelGGL(lg_sim_arg_kernel<PBF16>, grid, dim3(256), 0, st, md, z, lens, Np, nT, p2, rowlse, collse, a2, scores_out);
  }
  hipLaunchKernelGGL(lg_filter_fused_kernel, dim3(B), dim3(1024), 0, st, lens, Np, nT, cap, thr, a2, rowarg, rowval, colarg, idx, score, nmatch);
}

// =============================================================================== fault hunting: state checksums
// Position-dependent 64-bit checksum of every `unit_words`-word unit of a buffer (airfe_debug_trace): sums are commutative, so the result
// does not depend on how the threads are scheduled.  One workgroup per unit.
__global__ __launch_bounds__(256) void trace_hash_kernel(const uint32_t* __restrict__ p, unsigned unit_words, unsigned long long* __restrict__ out) {
  const uint32_t* base = p + (size_t)blockIdx.x * unit_words;
  unsigned long long h = 0;
  for (unsigned i = threadIdx.x; i < unit_words; i += 256) {
    unsigned long long x = ((unsigned long long)base[i] + 1ull) * 0x9E3779B97F4A7C15ull + (unsigned long long)i * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    h += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// per-slot digest of the unit checksums: slot i covers units [off[i], off[i+1])
__global__ __launch_bounds__(256) void trace_digest_kernel(const unsigned long long* __restrict__ tab, const unsigned* __restrict__ off, unsigned long long* __restrict__ dig) {
  const unsigned a = off[blockIdx.x], b = off[blockIdx.x + 1];
  unsigned long long h = 0;
  for (unsigned i = a + threadIdx.x; i < b; i += 256) {
    unsigned long long x = (tab[i] ^ ((unsigned long long)(i - a) * 0xC2B2AE3D27D4EB4Full)) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 31;
    h += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) dig[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
void launch_trace_hash(const void* p, unsigned unit_words, unsigned units, unsigned long long* out, hipStream_t st) {
  hipLaunchKernelGGL(trace_hash_kernel, dim3(units), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(p), unit_words, out);
}
void launch_trace_digest(const unsigned long long* tab, const unsigned* off, int slots, unsigned long long* dig, hipStream_t st) {
  hipLaunchKernelGGL(trace_digest_kernel, dim3(slots), dim3(256), 0, st, tab, off, dig);
}

}  // namespace airfe
