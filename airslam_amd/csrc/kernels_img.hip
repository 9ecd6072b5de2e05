// airfe — image-side kernels: cv::resize-compatible pre-process, conv1a (Cin = 1), detector heads
// (softmax-65 + depth-to-space, descriptor L2 norm, simple_nms), exact top-K keypoint selection and
// bilinear descriptor sampling.  All HBM-bound; written for coalesced 16-byte lanes.
#include <float.h>

#include "common.h"
#include "kernels.h"

namespace airfe {

// =============================================================================== pre-process
// cv::resize(INTER_LINEAR) 8-bit fixed-point path + `float(px)/255.0`
// (reference: src/plnet.cpp:246-270, src/super_point.cpp:111-116,146-165).
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, int stride, size_t img_stride,
                                  const int4* __restrict__ xtab, const int4* __restrict__ ytab,
                                  const float* __restrict__ lut, float* __restrict__ out, int RH, int RW) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= RW) return;
  const int4 xt = xtab[x], yt = ytab[y];
  const uint8_t* s = src + (size_t)b * img_stride;
  const uint8_t* r0 = s + (size_t)yt.x * stride;
  const uint8_t* r1 = s + (size_t)yt.y * stride;
  const int h0 = (int)r0[xt.x] * xt.z + (int)r0[xt.y] * xt.w;
  const int h1 = (int)r1[xt.x] * xt.z + (int)r1[xt.y] * xt.w;
  int v = (((yt.z * (h0 >> 4)) >> 16) + ((yt.w * (h1 >> 4)) >> 16) + 2) >> 2;
  v = min(max(v, 0), 255);
  out[((size_t)b * (RH + 2) + y + 1) * (RW + 2) + x + 1] = lut[v];
}

// Same arithmetic, one workgroup per output row: the two source rows it blends are staged into LDS with 4-byte loads (the
// per-pixel form issues four 1-byte global loads per output pixel: 64 bytes per wave instruction), then every thread picks its
// taps from LDS.  Rows that are not 4-byte aligned, or wider than the staging buffer, take the per-pixel kernel.
constexpr int PRE_MAX_W = 4096;
__global__ __launch_bounds__(256) void preprocess_rows_kernel(const uint8_t* __restrict__ src, int stride, size_t img_stride, int w,
                                                              const int4* __restrict__ xtab, const int4* __restrict__ ytab,
                                                              const float* __restrict__ lut, float* __restrict__ out, int RH, int RW) {
  __shared__ uint32_t sm[2][PRE_MAX_W / 4];
  const int y = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int4 yt = ytab[y];
  const uint8_t* s = src + (size_t)b * img_stride;
  const uint8_t* r0 = s + (size_t)yt.x * stride;
  const uint8_t* r1 = s + (size_t)yt.y * stride;
  const int nfull = w >> 2;
  for (int i = t; i < nfull; i += 256) {
    sm[0][i] = reinterpret_cast<const uint32_t*>(r0)[i];
    sm[1][i] = reinterpret_cast<const uint32_t*>(r1)[i];
  }
  if (t < (w & 3)) {                                    // tail bytes one by one: never read past the row
    reinterpret_cast<uint8_t*>(sm[0])[nfull * 4 + t] = r0[nfull * 4 + t];
    reinterpret_cast<uint8_t*>(sm[1])[nfull * 4 + t] = r1[nfull * 4 + t];
  }
  __syncthreads();
  const uint8_t* a0 = reinterpret_cast<const uint8_t*>(sm[0]);
  const uint8_t* a1 = reinterpret_cast<const uint8_t*>(sm[1]);
  float* orow = out + ((size_t)b * (RH + 2) + y + 1) * (RW + 2) + 1;
  for (int x = t; x < RW; x += 256) {
    const int4 xt = xtab[x];
    const int h0 = (int)a0[xt.x] * xt.z + (int)a0[xt.y] * xt.w;
    const int h1 = (int)a1[xt.x] * xt.z + (int)a1[xt.y] * xt.w;
    int v = (((yt.z * (h0 >> 4)) >> 16) + ((yt.w * (h1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    orow[x] = lut[v];
  }
}

// Round 5: FOUR output rows per workgroup.  The one-row form is a chain of dependent latencies per 2 KB of output (row table -> two source rows -> barrier ->
// tap table + LUT -> store) run by 512 x B tiny workgroups: 48 us for 64 images = 1.9 TB/s with its waves parked 81 % of the time.  Here the eight source
// rows of four output rows are requested together, the tap table entry of a column is read once for the four rows and the 1 KB LUT sits in LDS.  Same
// integer arithmetic, same bits.
#ifndef AIRFE_PRE_R
#define AIRFE_PRE_R 4
#endif
constexpr int PRE_R = AIRFE_PRE_R, PRE_R_MAX_W = 2048;
__global__ __launch_bounds__(256) void preprocess_rows4_kernel(const uint8_t* __restrict__ src, int stride, size_t img_stride, int w,
                                                               const int4* __restrict__ xtab, const int4* __restrict__ ytab,
                                                               const float* __restrict__ lut, float* __restrict__ out, int RH, int RW) {
  __shared__ uint32_t sm[PRE_R][2][PRE_R_MAX_W / 4];
  __shared__ float slut[256];
  const int y0 = blockIdx.x * PRE_R, b = blockIdx.y, t = threadIdx.x;
  const uint8_t* s = src + (size_t)b * img_stride;
  int4 yt[PRE_R];
#pragma unroll
  for (int r = 0; r < PRE_R; ++r) yt[r] = ytab[y0 + r];
  const int nfull = w >> 2;
#pragma unroll
  for (int r = 0; r < PRE_R; ++r) {
    const uint8_t* r0 = s + (size_t)yt[r].x * stride;
    const uint8_t* r1 = s + (size_t)yt[r].y * stride;
    for (int i = t; i < nfull; i += 256) {
      sm[r][0][i] = reinterpret_cast<const uint32_t*>(r0)[i];
      sm[r][1][i] = reinterpret_cast<const uint32_t*>(r1)[i];
    }
    if (t < (w & 3)) {                                  // tail bytes one by one: never read past the row
      reinterpret_cast<uint8_t*>(sm[r][0])[nfull * 4 + t] = r0[nfull * 4 + t];
      reinterpret_cast<uint8_t*>(sm[r][1])[nfull * 4 + t] = r1[nfull * 4 + t];
    }
  }
  slut[t] = lut[t];
  __syncthreads();
  float* obase = out + ((size_t)b * (RH + 2) + y0 + 1) * (RW + 2) + 1;
  for (int x = t; x < RW; x += 256) {
    const int4 xt = xtab[x];
#pragma unroll
    for (int r = 0; r < PRE_R; ++r) {
      const uint8_t* a0 = reinterpret_cast<const uint8_t*>(sm[r][0]);
      const uint8_t* a1 = reinterpret_cast<const uint8_t*>(sm[r][1]);
      const int h0 = (int)a0[xt.x] * xt.z + (int)a0[xt.y] * xt.w;
      const int h1 = (int)a1[xt.x] * xt.z + (int)a1[xt.y] * xt.w;
      int v = (((yt[r].z * (h0 >> 4)) >> 16) + ((yt[r].w * (h1 >> 4)) >> 16) + 2) >> 2;
      v = min(max(v, 0), 255);
      obase[(size_t)r * (RW + 2) + x] = slut[v];
    }
  }
}

// =============================================================================== rectification
// cv::remap(src, dst, map1, map2, INTER_LINEAR) with CV_32FC1 maps and the default BORDER_CONSTANT(0) as Camera::UndistortImage
// calls it (src/camera.cc:161-182), 8-bit single channel, restated from OpenCV 4.x imgproc/src/imgwarp.cpp (RemapInvoker +
// remapBilinear<FixedPtCast<int, uchar, 15>>):  sx = cvRound(mapx * 32), sy = cvRound(mapy * 32) (round half to even);
// integer part = s >> 5, fraction index = s & 31; weights from the 32 x 32 bilinear table of 15-bit integers
// w = (32 - fx)(32 - fy) * 32 ... — exact products, except the (0, 0) entry where saturate_cast<short>(32768) = 32767 and the
// table's sum correction puts the missing 1 on the diagonal tap; result = (sum w_i p_i + 2^14) >> 15, taps outside the image = 0.
__global__ __launch_bounds__(256) void remap_linear_kernel(const uint8_t* __restrict__ src, int B, int h, int w, int stride, size_t img_stride,
                                                          const float* __restrict__ mapx, const float* __restrict__ mapy,
                                                          uint8_t* __restrict__ dst, int dstride, size_t dimg_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
  if (x >= w || y >= h) return;
  const uint8_t* S = src + (size_t)b * img_stride;
  const int sx = __float2int_rn(__fmul_rn(mapx[(size_t)y * w + x], 32.f)), sy = __float2int_rn(__fmul_rn(mapy[(size_t)y * w + x], 32.f));
  // XY = saturate_cast<short>(s >> 5): far-away coordinates clamp to +-32767 / -32768 and land outside any image either way
  const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));
  const int fx = sx & 31, fy = sy & 31;
  int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
  auto tap = [&](int xx, int yy) -> int { return ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) ? (int)S[(size_t)yy * stride + xx] : 0; };
  int v = 0;
  if (!(ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0))
    v = (tap(ix, iy) * w00 + tap(ix + 1, iy) * w01 + tap(ix, iy + 1) * w10 + tap(ix + 1, iy + 1) * w11 + (1 << 14)) >> 15;
  dst[(size_t)b * dimg_stride + (size_t)y * dstride + x] = (uint8_t)min(max(v, 0), 255);
}

void launch_remap_linear(const uint8_t* src, int B, int h, int w, int stride, size_t img_stride, const float* mapx, const float* mapy,
                         uint8_t* dst, int dstride, size_t dimg_stride, hipStream_t st) {
  hipLaunchKernelGGL(remap_linear_kernel, dim3((w + 63) / 64, (h + 3) / 4, B), dim3(256), 0, st, src, B, h, w, stride, img_stride, mapx, mapy,
                     dst, dstride, dimg_stride);
}

void launch_preprocess(const uint8_t* src, int B, int h, int w, int stride, size_t img_stride, const int* xtab,
                       const int* ytab, const float* lut, float* out, int RH, int RW, hipStream_t st) {
  (void)h;
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | (uintptr_t)img_stride) & 3) == 0;
  if (aligned && w <= PRE_R_MAX_W && RH % PRE_R == 0) {
    hipLaunchKernelGGL(preprocess_rows4_kernel, dim3(RH / PRE_R, B), dim3(256), 0, st, src, stride, img_stride, w,
                       reinterpret_cast<const int4*>(xtab), reinterpret_cast<const int4*>(ytab), lut, out, RH, RW);
    return;
  }
  if (aligned && w <= PRE_MAX_W) {
    hipLaunchKernelGGL(preprocess_rows_kernel, dim3(RH, B), dim3(256), 0, st, src, stride, img_stride, w,
                       reinterpret_cast<const int4*>(xtab), reinterpret_cast<const int4*>(ytab), lut, out, RH, RW);
    return;
  }
  dim3 grid((RW + 255) / 256, RH, B);
  hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, st, src, stride, img_stride,
                     reinterpret_cast<const int4*>(xtab), reinterpret_cast<const int4*>(ytab), lut, out, RH, RW);
}

// =============================================================================== detector head
// softmax over 65 logits, drop the dustbin, 8x8 depth-to-space (SuperPoint head)
__global__ void softmax_d2s_kernel(const float* __restrict__ logits, int ldl, float* __restrict__ heat, int ncell,
                                   int HC, int WC) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  const float* l = logits + (size_t)cell * ldl;
  float v[65];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 64; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(l + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
  v[64] = l[64];
#pragma unroll
  for (int i = 0; i < 65; ++i) mx = fmaxf(mx, v[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 65; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
  const float inv = 1.0f / sum;
  const int b = cell / (HC * WC), rem = cell - b * HC * WC;
  const int cy = rem / WC, cx = rem - cy * WC;
  float* o = heat + ((size_t)b * HC * 8 + (size_t)cy * 8) * (WC * 8) + cx * 8;
#pragma unroll
  for (int dy = 0; dy < 8; ++dy) {
    float* r = o + (size_t)dy * (WC * 8);
    *reinterpret_cast<float4*>(r) = make_float4(v[dy * 8] * inv, v[dy * 8 + 1] * inv, v[dy * 8 + 2] * inv, v[dy * 8 + 3] * inv);
    *reinterpret_cast<float4*>(r + 4) = make_float4(v[dy * 8 + 4] * inv, v[dy * 8 + 5] * inv, v[dy * 8 + 6] * inv, v[dy * 8 + 7] * inv);
  }
}

void launch_softmax_d2s(const float* logits, int ldl, float* heat, int B, int HC, int WC, hipStream_t st) {
  const int ncell = B * HC * WC;
  hipLaunchKernelGGL(softmax_d2s_kernel, dim3((ncell + 127) / 128), dim3(128), 0, st, logits, ldl, heat, ncell, HC, WC);
}

// F.normalize(dim=channel) of one 256-channel row spread over a wave, 4 channels per lane
__device__ __forceinline__ void l2norm_lane4(float4& v) {
  const float ss = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
}

// dense form, one wave per row (inspection hook only; the product normalises lazily inside sample_desc_kernel)
__global__ void l2norm256_kernel(float* __restrict__ d, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4* p = reinterpret_cast<float4*>(d + (size_t)row * 256) + lane;
  float4 v = *p;
  l2norm_lane4(v);
  *p = v;
}

void launch_l2norm256(float* d, int rows, hipStream_t st) {
  hipLaunchKernelGGL(l2norm256_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, d, rows);
}

// =============================================================================== simple_nms
// Upstream SuperPoint simple_nms: 5 max-pools of (2r+1)^2 with mask logic, done as separable passes.
__global__ void nms_poolh_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int r) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t base = ((size_t)blockIdx.z * H + y) * W;
  float m = -INFINITY;
  for (int dx = -r; dx <= r; ++dx) {
    const int xx = x + dx;
    if (xx >= 0 && xx < W) m = fmaxf(m, in[base + xx]);
  }
  out[base + x] = m;
}

// mode 0: M = (S == pv)
// mode 1: SUPP = pv > 0 ; SS = SUPP ? 0 : S
// mode 2: M |= (SS == pv) & !SUPP
// mode 3: like 2, and OUT = M ? S : 0
__global__ void nms_poolv_kernel(const float* __restrict__ A, const float* __restrict__ S, float* __restrict__ M,
                                 float* __restrict__ SS, float* __restrict__ SUPP, float* __restrict__ OUT, int H, int W,
                                 int r, int mode) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t img = (size_t)blockIdx.z * H * W;
  float pv = -INFINITY;
  for (int dy = -r; dy <= r; ++dy) {
    const int yy = y + dy;
    if (yy >= 0 && yy < H) pv = fmaxf(pv, A[img + (size_t)yy * W + x]);
  }
  const size_t i = img + (size_t)y * W + x;
  if (mode == 0) {
    M[i] = (S[i] == pv) ? 1.f : 0.f;
  } else if (mode == 1) {
    const bool sp = pv > 0.f;
    SUPP[i] = sp ? 1.f : 0.f;
    SS[i] = sp ? 0.f : S[i];
  } else {
    const bool nm = (SS[i] == pv);
    const bool m = (M[i] > 0.f) || (nm && !(SUPP[i] > 0.f));
    M[i] = m ? 1.f : 0.f;
    if (mode == 3) OUT[i] = m ? S[i] : 0.f;
  }
}

void launch_simple_nms(const float* heat, float* out, float* tmp, int B, int H, int W, int radius, hipStream_t st) {
  const size_t n = (size_t)B * H * W;
  float *A = tmp, *M = tmp + n, *SS = tmp + 2 * n, *SUPP = tmp + 3 * n;
  dim3 grid((W + 255) / 256, H, B), blk(256);
  hipLaunchKernelGGL(nms_poolh_kernel, grid, blk, 0, st, heat, A, H, W, radius);
  hipLaunchKernelGGL(nms_poolv_kernel, grid, blk, 0, st, A, heat, M, SS, SUPP, out, H, W, radius, 0);
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL(nms_poolh_kernel, grid, blk, 0, st, M, A, H, W, radius);
    hipLaunchKernelGGL(nms_poolv_kernel, grid, blk, 0, st, A, heat, M, SS, SUPP, out, H, W, radius, 1);
    hipLaunchKernelGGL(nms_poolh_kernel, grid, blk, 0, st, SS, A, H, W, radius);
    hipLaunchKernelGGL(nms_poolv_kernel, grid, blk, 0, st, A, heat, M, SS, SUPP, out, H, W, radius, it == 1 ? 3 : 2);
  }
}

// =============================================================================== descriptor sampling
// extract_descriptors (src/plnet.cpp:369-417 == src/super_point.cpp:224-272) on a dense NHWC fp32 map,
// one wave per keypoint (lane = 4 channels, 1 KiB coalesced row reads), then the final rescale
// (plnet.cpp:574-575).  _rn intrinsics keep hipcc from contracting the reference's mul/add pairs.
__device__ __forceinline__ int clipi(int v, int mx) { return v < 0 ? 0 : min(v, mx - 1); }

// the four taps of a keypoint (grid_sample bilinear, align_corners semantics of extract_descriptors) and their weights
struct DescTaps { int c[4]; float w[4]; };       // cell = iy * WC + ix of nw, ne, sw, se
__device__ __forceinline__ DescTaps desc_taps(float x, float y, float sx, float bx, float sy, float by, int HC, int WC) {
  float kx = __fadd_rn(__fmul_rn(x, sx), bx), ky = __fadd_rn(__fmul_rn(y, sy), by);
  kx = __fmul_rn(__fadd_rn(kx, 1.0f), 0.5f);
  ky = __fmul_rn(__fadd_rn(ky, 1.0f), 0.5f);
  const float ix = __fmul_rn(kx, (float)(WC - 1)), iy = __fmul_rn(ky, (float)(HC - 1));
  const int ix_nw = clipi((int)floorf(ix), WC), iy_nw = clipi((int)floorf(iy), HC);
  const int ix_ne = clipi(ix_nw + 1, WC), iy_ne = clipi(iy_nw, HC);
  const int ix_sw = clipi(ix_nw, WC), iy_sw = clipi(iy_nw + 1, HC);
  const int ix_se = clipi(ix_nw + 1, WC), iy_se = clipi(iy_nw + 1, HC);
  DescTaps t;
  t.w[0] = __fmul_rn(__fsub_rn((float)ix_se, ix), __fsub_rn((float)iy_se, iy));
  t.w[1] = __fmul_rn(__fsub_rn(ix, (float)ix_sw), __fsub_rn((float)iy_sw, iy));
  t.w[2] = __fmul_rn(__fsub_rn((float)ix_ne, ix), __fsub_rn(iy, (float)iy_ne));
  t.w[3] = __fmul_rn(__fsub_rn(ix, (float)ix_nw), __fsub_rn(iy, (float)iy_nw));
  t.c[0] = iy_nw * WC + ix_nw; t.c[1] = iy_ne * WC + ix_ne; t.c[2] = iy_sw * WC + ix_sw; t.c[3] = iy_se * WC + ix_se;
  return t;
}

// rows of the dense head input that the keypoints of a batch will sample: idx[(b * cap + k) * 4 + tap] = (b0 + b) * HC * WC + cell
// (slots k >= n[b]: row 0, computed and never read).  The descriptor head then runs as a GATHER GEMM over these rows only.
__global__ void desc_cells_kernel(const float* __restrict__ feat, const int* __restrict__ n, int cap, int B, int b0, int HC, int WC,
                                  float sx, float bx, float sy, float by, int* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * cap) return;
  const int b = i / cap, k = i - b * cap;
  int4 o = make_int4(0, 0, 0, 0);
  if (k < n[b]) {
    const float* f = feat + (size_t)i * 259;
    const DescTaps t = desc_taps(f[1], f[2], sx, bx, sy, by, HC, WC);
    const int base = (b0 + b) * HC * WC;
    o = make_int4(base + t.c[0], base + t.c[1], base + t.c[2], base + t.c[3]);
  }
  reinterpret_cast<int4*>(idx)[i] = o;
}

// compact != 0: `desc` holds the four head rows of keypoint (b, k) at rows (b * cap + k) * 4 + tap (the gather GEMM's output)
// instead of the dense [B][HC][WC][256] map — same values, same operations from there on
__global__ __launch_bounds__(256) void sample_desc_kernel(const float* __restrict__ desc, int HC, int WC,
                                                          float* __restrict__ feat, const int* __restrict__ n, int cap,
                                                          float sx, float bx, float sy, float by, float w_scale,
                                                          float h_scale, int normalise, int compact, int* __restrict__ flag) {
  const int b = blockIdx.y, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= n[b]) return;
  float* f = feat + ((size_t)b * cap + k) * 259;
  const float x = f[1], y = f[2];
  const DescTaps tp = desc_taps(x, y, sx, bx, sy, by, HC, WC);
  const float nw = tp.w[0], ne = tp.w[1], sw = tp.w[2], se = tp.w[3];
  float4 a0, a1, a2, a3;
  if (compact) {
    const float* d = desc + ((size_t)b * cap + k) * 4 * 256 + lane * 4;
    a0 = *reinterpret_cast<const float4*>(d);
    a1 = *reinterpret_cast<const float4*>(d + 256);
    a2 = *reinterpret_cast<const float4*>(d + 512);
    a3 = *reinterpret_cast<const float4*>(d + 768);
  } else {
    const float* d = desc + (size_t)b * HC * WC * 256 + lane * 4;
    a0 = *reinterpret_cast<const float4*>(d + (size_t)tp.c[0] * 256);
    a1 = *reinterpret_cast<const float4*>(d + (size_t)tp.c[1] * 256);
    a2 = *reinterpret_cast<const float4*>(d + (size_t)tp.c[2] * 256);
    a3 = *reinterpret_cast<const float4*>(d + (size_t)tp.c[3] * 256);
  }
  if (normalise) {       // F.normalize(dim=channel) of the four cells, operation for operation as l2norm256_kernel does it
    l2norm_lane4(a0);
    l2norm_lane4(a1);
    l2norm_lane4(a2);
    l2norm_lane4(a3);
  }
  float v[4];
  const float n0[4] = {a0.x, a0.y, a0.z, a0.w}, n1[4] = {a1.x, a1.y, a1.z, a1.w};
  const float n2[4] = {a2.x, a2.y, a2.z, a2.w}, n3[4] = {a3.x, a3.y, a3.z, a3.w};
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = __fmul_rn(n0[j], nw);
    t = __fadd_rn(t, __fmul_rn(n1[j], ne));
    t = __fadd_rn(t, __fmul_rn(n2[j], sw));
    t = __fadd_rn(t, __fmul_rn(n3[j], se));
    v[j] = t;
    ss = __fadd_rn(ss, __fmul_rn(t, t));
  }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  if (flag && lane == 0 && !(ss <= FLT_MAX)) *reinterpret_cast<volatile int*>(flag + 1) = 1;      // inf / NaN descriptor: the detector's 2-byte activations overflowed (airfe.h, "activation range")
  // Eigen (>= 3.3) colwise().normalize(): `if (squaredNorm() > 0) v /= sqrt(squaredNorm())` — a zero column stays zero, no NaN
#pragma unroll
  for (int j = 0; j < 4; ++j) f[3 + lane * 4 + j] = (ss > 0.f) ? v[j] / nrm : v[j];
  if (lane == 0) {
    f[1] = __fmul_rn(x, w_scale);
    f[2] = __fmul_rn(y, h_scale);
  }
}

static void desc_grid_coeffs(int HC, int WC, float& sx, float& bx, float& sy, float& by) {
  const int s = 8;
  sx = (float)(2.0 / (WC * s - s / 2 - 0.5)); bx = (float)((1 - s) / (WC * s - s / 2 - 0.5) - 1);
  sy = (float)(2.0 / (HC * s - s / 2 - 0.5)); by = (float)((1 - s) / (HC * s - s / 2 - 0.5) - 1);
}

void launch_sample_desc(const float* desc, int B, int HC, int WC, float* feat, const int* n, int cap, float w_scale,
                        float h_scale, int normalise, hipStream_t st, int compact, int* flag) {
  float sx, bx, sy, by;
  desc_grid_coeffs(HC, WC, sx, bx, sy, by);
  hipLaunchKernelGGL(sample_desc_kernel, dim3((cap + 3) / 4, B), dim3(256), 0, st, desc, HC, WC, feat, n, cap, sx, bx,
                     sy, by, w_scale, h_scale, normalise, compact, flag);
}

void launch_desc_cells(const float* feat, const int* n, int cap, int B, int b0, int HC, int WC, int* idx, hipStream_t st) {
  float sx, bx, sy, by;
  desc_grid_coeffs(HC, WC, sx, bx, sy, by);
  hipLaunchKernelGGL(desc_cells_kernel, dim3((B * cap + 255) / 256), dim3(256), 0, st, feat, n, cap, B, b0, HC, WC, sx, bx, sy, by, idx);
}

}  // namespace airfe
