// airfe — fp32 CORRECTNESS path (cfg.precision = 2; BASELINE.json configs[1]: "fp32 on 1 x MI355X, outputs diffed").  Storage AND
// arithmetic in fp32: the contractions run on the f32-input MFMA v_mfma_f32_16x16x4_f32 (exact f32, = an fmaf chain in k order, the
// f32 vector rate = 1/16 of the 2-byte MFMA — MI355X_MICROARCH.md 'Matrix cores'), everything else on the VALU.  These kernels are
// deliberately plain (no LDS tiling, fragments straight from L1 / L2): the mode exists to pin the 2-byte kernels and the oracle
// against each other at 1e-5, not to be fast (it measures ~25x slower than the fp16 path).
#include "common.h"
#include "kernels.h"

namespace airfe {

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- conv1a: fp32 image [B][H+2][W+2] (zero border) -> fp32 NHWC [B][H+2][W+2][64] (zero border), 3x3, ReLU
__global__ __launch_bounds__(256) void conv1a_f32_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int H, int W) {
  const long total = (long)B * H * W * 16;              // 4 channels per thread
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c4 = (int)(t & 15);
    const long pix = t >> 4;
    const int b = (int)(pix / ((long)H * W)), rem = (int)(pix - (long)b * H * W), y = rem / W, x = rem - y * W;
    const float* ip = img + ((size_t)b * (H + 2) + y) * (W + 2) + x;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float s = bias[c4 * 4 + c];
#pragma unroll
      for (int k = 0; k < 9; ++k) s = fmaf(w[(c4 * 4 + c) * 9 + k], ip[(size_t)(k / 3) * (W + 2) + (k % 3)], s);
      v[c] = fmaxf(s, 0.f);
    }
    *reinterpret_cast<float4*>(out + (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * 64 + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- 3x3 conv, pad 1, ReLU: X [B][H+2][W+2][CIN] (zero border) -> Y [B][H+2p][W+2p][COUT]; Wt [9][CIN][COUT]
// one wave = 16 pixels of a row x 64 output channels: D[cout][pixel] += Wt[tap][k][cout] . X[pixel + tap][k] in 4-wide k steps
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                          float* __restrict__ Y, int B, int H, int W, int CIN, int COUT, int opad) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int xb = W / 16, cb = COUT / 64;
  const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6), ntask = (long)B * H * xb * cb;
  if (task >= ntask) return;
  const int c0 = (int)(task % cb) * 64;
  long r = task / cb;
  const int x0 = (int)(r % xb) * 16; r /= xb;
  const int y = (int)(r % H), b = (int)(r / H);
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + c0 + 16 * t + 4 * g);     // D row = cout 4 g + r of tile t
    acc[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
  }
  const float* xr = X + (((size_t)b * (H + 2) + y) * (W + 2) + x0 + l15) * CIN + g;
  for (int tap = 0; tap < 9; ++tap) {
    const float* xp = xr + ((size_t)(tap / 3) * (W + 2) + (tap % 3)) * CIN;
    const float* wp = Wt + ((size_t)tap * CIN + g) * COUT + c0 + l15;
    for (int k0 = 0; k0 < CIN; k0 += 4) {
      const float bfr = xp[k0];                                        // B[k = g][j = l15] = X[pixel l15][k0 + g]
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma_f32(wp[(size_t)k0 * COUT + 16 * t], bfr, acc[t]);   // A[i = l15][k = g] = Wt[k0 + g][c0 + 16 t + l15]
    }
  }
  float* yo = Y + (((size_t)b * (H + 2 * opad) + y + opad) * (W + 2 * opad) + x0 + l15 + opad) * COUT + c0 + 4 * g;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    *reinterpret_cast<float4*>(yo + 16 * t) = make_float4(fmaxf(acc[t][0], 0.f), fmaxf(acc[t][1], 0.f), fmaxf(acc[t][2], 0.f), fmaxf(acc[t][3], 0.f));
}

// ---- 2x2 max-pool of a bordered NHWC map into a bordered NHWC map
__global__ __launch_bounds__(256) void maxpool2_f32_kernel(const float* __restrict__ X, float* __restrict__ Y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, c4n = C / 4;
  const long total = (long)B * Ho * Wo * c4n;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c4 = (int)(t % c4n);
    long r = t / c4n;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho), b = (int)(r / Ho);
    const float* p = X + (((size_t)b * (H + 2) + 2 * y + 1) * (W + 2) + 2 * x + 1) * C + c4 * 4;
    const float4 a = *reinterpret_cast<const float4*>(p), bq = *reinterpret_cast<const float4*>(p + C);
    const float4 c = *reinterpret_cast<const float4*>(p + (size_t)(W + 2) * C), d = *reinterpret_cast<const float4*>(p + (size_t)(W + 2) * C + C);
    *reinterpret_cast<float4*>(Y + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * C + c4 * 4) =
        make_float4(fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y)),
                    fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w)));
  }
}

// ---- Y[m][n] (+)= act(sum_k X[m][k] W[n][k] + bias[n]) ; X = [X1 (K1 columns, ld1) | X2 (K - K1 columns, ld2)], W row-major [N][K]
// one wave = 16 rows x 64 columns: D[n][m] with A = W rows, B = X rows
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args a) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int nb = (a.N + 63) / 64, mb = (a.M + 15) / 16;
  const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= (long)nb * mb) return;
  const int n0 = (int)(task % nb) * 64, m0 = (int)(task / nb) * 16;
  const int m = min(m0 + l15, a.M - 1);
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* x1 = a.X1 + (size_t)m * a.ld1 + g;
  const float* x2 = a.X2 ? a.X2 + (size_t)m * a.ld2 + g : nullptr;
  const float* wr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wr[t] = a.W + (size_t)min(n0 + 16 * t + l15, a.N - 1) * a.K + g;
  for (int k0 = 0; k0 < a.K; k0 += 4) {
    const float bfr = k0 < a.K1 ? x1[k0] : x2[k0 - a.K1];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma_f32(wr[t][k0], bfr, acc[t]);
  }
  if (m0 + l15 >= a.M) return;
  float* yr = a.Y + (size_t)(m0 + l15) * a.ldy;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * t + 4 * g + r;
      if (n < a.N) {
        float v = (acc[t][r] + (a.bias ? a.bias[n] : 0.f)) * a.scale;
        if (a.relu) v = fmaxf(v, 0.f);
        yr[n] = a.accumulate ? yr[n] + v : v;
      }
    }
}

// ---- LightGlue rotary on the q and k halves of [M][ld] (columns [0, 512): q heads then k heads), pairs (2i, 2i+1), cos / sin [M][32]
__global__ void rotary_f32_kernel(float* __restrict__ qk, int ld, const float* __restrict__ rc, const float* __restrict__ rs, int M) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M * 256) return;
  const int m = (int)(t >> 8), p = (int)(t & 255);           // pair p covers columns 2p, 2p + 1 (p < 128: q, else k)
  const int i = p & 31;                                      // pair index inside its head
  float* v = qk + (size_t)m * ld + 2 * p;
  const float c = rc[(size_t)m * 32 + i], s = rs[(size_t)m * 32 + i], x0 = v[0], x1 = v[1];
  v[0] = x0 * c - x1 * s;
  v[1] = x1 * c + x0 * s;
}

// ---- attention, one wave per (sequence, head, query): scores over the keys with one key per lane and pass, exact soft-max, then
// lanes over the 64 output dims.  Q rows at Q[(s Np + q) ldq + h 64], K / V rows of sequence skv likewise; O [S][Np][256].
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                            const float* __restrict__ V, int ldv, float* __restrict__ O,
                                                            const int* __restrict__ lens, int H, int Np, int cross, float scale) {
  __shared__ float pbuf[4][1024];
  __shared__ float qs[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long task = (long)blockIdx.x * 4 + wv;
  const int q = (int)(task % Np), h = (int)((task / Np) % H), s = (int)(task / ((long)Np * H));
  const int skv = cross ? (s ^ 1) : s, n = lens[skv];
  const bool live = q < lens[s];
  qs[wv][lane] = Q[((size_t)s * Np + q) * ldq + h * 64 + lane];
  __syncthreads();
  float sc[16];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int j = lane + 64 * c;
    sc[c] = -INFINITY;
    if (live && j < n) {
      const float4* kr = reinterpret_cast<const float4*>(K + ((size_t)skv * Np + j) * ldk + h * 64);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float4 kv = kr[e];
        d = fmaf(qs[wv][4 * e], kv.x, d); d = fmaf(qs[wv][4 * e + 1], kv.y, d);
        d = fmaf(qs[wv][4 * e + 2], kv.z, d); d = fmaf(qs[wv][4 * e + 3], kv.w, d);
      }
      sc[c] = d * scale;
    }
    mx = fmaxf(mx, sc[c]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const float p = (sc[c] == -INFINITY) ? 0.f : expf(sc[c] - mx);
    sum += p;
    pbuf[wv][lane + 64 * c] = p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float acc = 0.f;
  if (live)
    for (int j = 0; j < n; ++j) acc = fmaf(pbuf[wv][j], V[((size_t)skv * Np + j) * ldv + h * 64 + lane], acc);
  O[((size_t)s * Np + q) * 256 + h * 64 + lane] = (live && sum > 0.f) ? acc / sum : 0.f;
}

// ---- in-place LayerNorm(512, eps 1e-5) + exact GELU (erf) on fp32 rows
__global__ __launch_bounds__(256) void ln_gelu_f32_kernel(float* __restrict__ h, const float* __restrict__ gamma, const float* __restrict__ beta, int M) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float* p = h + (size_t)row * 512 + lane * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = p[e];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) sum += v[e];
  const float mean = wave_sum(sum) * (1.0f / 512.0f);
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; var += d * d; }
  var = wave_sum(var) * (1.0f / 512.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float y = (v[e] - mean) * rstd * gamma[lane * 8 + e] + beta[lane * 8 + e];
    p[e] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752f));
  }
}

// ---- launches
void launch_conv1a_f32(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(conv1a_f32_kernel, dim3(4096), dim3(256), 0, st, img, w, bias, out, B, H, W);
}
void launch_conv3x3_f32(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int CIN, int COUT, int opad,
                        hipStream_t st) {
  const long ntask = (long)B * H * (W / 16) * (COUT / 64);
  hipLaunchKernelGGL(conv3x3_f32_kernel, dim3((unsigned)((ntask + 3) / 4)), dim3(256), 0, st, X, Wt, bias, Y, B, H, W, CIN, COUT, opad);
}
void launch_maxpool2_f32(const float* X, float* Y, int B, int H, int W, int C, hipStream_t st) {
  hipLaunchKernelGGL(maxpool2_f32_kernel, dim3(4096), dim3(256), 0, st, X, Y, B, H, W, C);
}
void launch_gemm_f32(const GemmF32Args& a, hipStream_t st) {
  const long ntask = (long)((a.N + 63) / 64) * ((a.M + 15) / 16);
  if (ntask < 1) return;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((ntask + 3) / 4)), dim3(256), 0, st, a);
}
void launch_rotary_f32(float* qk, int ld, const float* rc, const float* rs, int M, hipStream_t st) {
  hipLaunchKernelGGL(rotary_f32_kernel, dim3((unsigned)(((long)M * 256 + 255) / 256)), dim3(256), 0, st, qk, ld, rc, rs, M);
}
void launch_attention_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, const int* lens, int S, int H,
                          int Np, int cross, float scale, hipStream_t st) {
  const long ntask = (long)S * H * Np;                      // Np <= 1024 keys fit the 16 per-lane score slots
  hipLaunchKernelGGL(attention_f32_kernel, dim3((unsigned)((ntask + 3) / 4)), dim3(256), 0, st, Q, ldq, K, ldk, V, ldv, O, lens, H, Np, cross, scale);
}
void launch_ln_gelu_f32(float* h, const float* gamma, const float* beta, int M, hipStream_t st) {
  hipLaunchKernelGGL(ln_gelu_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, st, h, gamma, beta, M);
}

}  // namespace airfe
