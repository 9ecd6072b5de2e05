// airfe — simple_nms (radius 4) on the 512 x 512 score map as THREE register-resident launches, no LDS, no barriers.
//
// The public SuperPoint simple_nms is five dependent 9x9 max-pools: max_mask M0 = (S == pool S); twice: D = pool(M) > 0,
// ss = D ? 0 : S, M |= !D && (ss == pool ss).  The two mask pools are OR-dilations and cost nothing as bit operations, so the
// work is three float pools — one launch each, with only a BIT plane (32 KB per image) travelling between them; each consumer
// dilates the producer's bits itself.  (The five-launch LDS-tiled version this replaces moved two BYTE planes through L2 between
// launches and spent 75 % of its time waiting at barriers: 0.44 ms per 128 images.)
//
// One WAVE owns one 8-row band of one image at its full width: lane L holds columns 8L .. 8L+7, so a row is two 16-byte loads per
// lane and 2 KB per wave.  The 9-row maxima of 16 loaded rows are taken per column in registers (max9_of16: 22 v_max3_f32 for
// 8 outputs); the 9-column maxima from prefix / suffix maxima of the lane's own 8 values and of its two neighbour lanes, fetched
// with the wave-shift DPP modifiers (wave_shr:1 / wave_shl:1 — lane 0 / 63 keep the -inf fill: the image border).  A bit plane is
// [image][band][lane] 64-bit words: byte r = row 8*band + r, bit j = column 8*lane + j; a consumer reads the words of its own
// band and of the bands above and below and dilates them with byte-parallel shifts.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef unsigned long long u64;

namespace {

constexpr int NR = 512, NBAND = NR / 8;

__device__ __forceinline__ u64 nms_key(float s, int idx) {        // = make_key of kernels_sel.hip
  return (1ull << 49) | ((u64)(__float_as_uint(s) & 0x7FFFFFFFu) << 18) | (u64)(0x3FFFF - idx);
}
__device__ __forceinline__ float vmax3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// eight sliding 9-maxima of 16 inputs: 14 + 8 three-input maxima
__device__ __forceinline__ void max9of16(const float* x, float* o) {
  float t[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) t[i] = vmax3f(x[i], x[i + 1], x[i + 2]);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = vmax3f(t[i], t[i + 3], t[i + 6]);
}
// value of lane - 1 (lane 0: fill) / lane + 1 (lane 63: fill)
__device__ __forceinline__ float lane_left(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ float lane_right(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
__device__ __forceinline__ unsigned lane_left_u(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned lane_right_u(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false); }

// 9-column maxima of one row: v = the lane's 8 values; the window of column 8L+j is [8L+j-4, 8L+j+4]
__device__ __forceinline__ void hmax9(const float* v, float* o) {
  const float NI = -INFINITY;
  float p[8], s[8];
  p[0] = v[0];
  s[7] = v[7];
#pragma unroll
  for (int k = 1; k < 8; ++k) p[k] = fmaxf(p[k - 1], v[k]);          // p[k] = max v[0..k]
#pragma unroll
  for (int k = 6; k >= 0; --k) s[k] = fmaxf(s[k + 1], v[k]);         // s[k] = max v[k..7]
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = fmaxf(lane_left(s[4 + j], NI), p[j + 4]);        // left lane's columns 4+j..7, own 0..j+4
#pragma unroll
  for (int j = 4; j < 8; ++j) o[j] = fmaxf(s[j - 4], lane_right(p[j - 4], NI));       // own j-4..7, right lane's 0..j-4
}

// OR-dilation by +-4 columns of four byte rows (one byte = the lane's 8 columns), neighbours' words lw / rw
__device__ __forceinline__ unsigned hdilate4(unsigned w, unsigned lw, unsigned rw) {
  unsigned d = w;
  d |= ((w << 1) & 0xFEFEFEFEu) | ((w >> 1) & 0x7F7F7F7Fu);
  d |= ((w << 2) & 0xFCFCFCFCu) | ((w >> 2) & 0x3F3F3F3Fu);
  d |= ((w << 3) & 0xF8F8F8F8u) | ((w >> 3) & 0x1F1F1F1Fu);
  d |= ((w << 4) & 0xF0F0F0F0u) | ((w >> 4) & 0x0F0F0F0Fu);
  const unsigned t = (lw >> 4) & 0x0F0F0F0Fu;                          // left lane's columns 4..7 at bits 0..3
  d |= (t | (t >> 1) | (t >> 2) | (t >> 3)) & 0x0F0F0F0Fu;            // own column j <- left columns 4+j..7
  const unsigned v = rw & 0x0F0F0F0Fu;                                 // right lane's columns 0..3
  d |= ((v | (v << 1) | (v << 2) | (v << 3)) & 0x0F0F0F0Fu) << 4;     // own column 4+i <- right columns 0..i
  return d;
}

struct U192 { u64 a, b, c; };                                          // 24 byte rows, row 0 in the low byte of a
__device__ __forceinline__ U192 shr_rows(U192 z, int rows) {          // rows = 1, 2, 4 (bits = 8 * rows < 64)
  const int s = 8 * rows;
  return U192{(z.a >> s) | (z.b << (64 - s)), (z.b >> s) | (z.c << (64 - s)), z.c >> s};
}
__device__ __forceinline__ U192 or3(U192 x, U192 y) { return U192{x.a | y.a, x.b | y.b, x.c | y.c}; }

}  // namespace

//   MODE 0: Mout = (S == pool S)
//   MODE 1: D = dilate(Min); ss = D ? 0 : S; Mout = Min | (!D && ss == pool ss)
//   MODE 2: MODE 1, then the detect_point candidate list of M ? S : 0 (and, if out != nullptr, that dense map)
template <int MODE>
__global__ __launch_bounds__(64) void nms512_kernel(const float* __restrict__ heat, const u64* __restrict__ Min, u64* __restrict__ Mout,
                                                    float* __restrict__ out, int B, float thr, int border, u64* __restrict__ cand,
                                                    int* __restrict__ cand_cnt, int cand_cap) {
  // XCD-aware order: consecutive workgroup ids go round the 8 XCDs, so XCD x gets the contiguous run of bands
  // [x * n/8, (x+1) * n/8) — neighbouring bands (which share 8 of their 16 rows) meet in one L2
  const int n = B * NBAND, per = n >> 3;
  int wid = blockIdx.x;
  if ((n & 7) == 0) wid = (wid & 7) * per + (wid >> 3);
  const int b = wid / NBAND, band = wid - b * NBAND, lane = threadIdx.x;
  const float* S = heat + (size_t)b * NR * NR;
  const int y0 = band * 8;

  unsigned D[4] = {0, 0, 0, 0};               // supp_mask of rows y0-4 .. y0+11 (byte = row, bit = column), MODE >= 1
  u64 Mc = 0;
  if (MODE >= 1) {
    const u64* Mp = Min + (size_t)b * NBAND * 64 + lane;
    U192 z;
    z.a = band > 0 ? Mp[(band - 1) * 64] : 0ull;
    z.b = Mp[band * 64];
    z.c = band < NBAND - 1 ? Mp[(band + 1) * 64] : 0ull;
    Mc = z.b;
    // rows i .. i+8 of the 24 -> centre row i+4: the low 16 rows of the result are the rows y0-4 .. y0+11
    const U192 a1 = or3(z, shr_rows(z, 1));
    const U192 a2 = or3(a1, shr_rows(a1, 2));
    const U192 a4 = or3(a2, shr_rows(a2, 4));
    const u64 lo = a4.a | z.b, hi = a4.b | z.c;                        // | (z >> 8 rows)
    unsigned v[4] = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
#pragma unroll
    for (int k = 0; k < 4; ++k) D[k] = hdilate4(v[k], lane_left_u(v[k]), lane_right_u(v[k]));
    if (band == 0) D[0] = 0;                                           // rows outside the image stay -inf padding
    if (band == NBAND - 1) D[3] = 0;
  }

  float x[16][8];
  const float NI = -INFINITY;
  const float* src = S + (size_t)(y0 - 4) * NR + lane * 8;
  if (band > 0 && band < NBAND - 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 q0 = *reinterpret_cast<const float4*>(src + (size_t)i * NR);
      const float4 q1 = *reinterpret_cast<const float4*>(src + (size_t)i * NR + 4);
      x[i][0] = q0.x; x[i][1] = q0.y; x[i][2] = q0.z; x[i][3] = q0.w; x[i][4] = q1.x; x[i][5] = q1.y; x[i][6] = q1.z; x[i][7] = q1.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int y = y0 - 4 + i;
      float4 q0 = make_float4(NI, NI, NI, NI), q1 = q0;
      if (y >= 0 && y < NR) {
        q0 = *reinterpret_cast<const float4*>(S + (size_t)y * NR + lane * 8);
        q1 = *reinterpret_cast<const float4*>(S + (size_t)y * NR + lane * 8 + 4);
      }
      x[i][0] = q0.x; x[i][1] = q0.y; x[i][2] = q0.z; x[i][3] = q0.w; x[i][4] = q1.x; x[i][5] = q1.y; x[i][6] = q1.z; x[i][7] = q1.w;
    }
  }
  if (MODE >= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = __builtin_amdgcn_sbfe((int)D[i >> 2], 8 * (i & 3) + j, 1);      // 0 or -1
        x[i][j] = __int_as_float(__float_as_int(x[i][j]) & ~m);
      }
  }

  float V[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float col[16], o[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) col[i] = x[i][j];
    max9of16(col, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) V[r][j] = o[r];
  }
  unsigned eq[2] = {0, 0};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float o[8];
    hmax9(V[r], o);
    unsigned byte = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) byte |= (x[r + 4][j] == o[j]) ? (1u << j) : 0u;
    eq[r >> 2] |= byte << (8 * (r & 3));
  }
  u64 M = ((u64)eq[1] << 32) | eq[0];
  if (MODE >= 1) M = Mc | (M & ~(((u64)D[2] << 32) | D[1]));           // rows y0 .. y0+7 of D are its bytes 4 .. 11
  if (MODE <= 1) {
    Mout[((size_t)b * NBAND + band) * 64 + lane] = M;
    return;
  }

  // ---- MODE 2: M ? S : 0 -> candidates (threshold + border box), one global atomic per wave and round
  const size_t img = (size_t)b * NR * NR;
  if (out) {
    const float4 zf = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float* d = out + img + (size_t)(y0 + r) * NR + lane * 8;
      *reinterpret_cast<float4*>(d) = zf;
      *reinterpret_cast<float4*>(d + 4) = zf;
    }
  }
  // Rounds of up to four surviving pixels per lane (a lane's 8 x 8 block holds at most four maxima that are more than 4 pixels apart;
  // only exact ties give more): the four score loads of a round are in flight together and a round costs ONE global atomic per wave.
  u64 kept = M;
  while (__builtin_amdgcn_ballot_w64(kept != 0) != 0) {
    float v[4];
    int gi[4];
    bool pass[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool has = kept != 0;
      const int k = has ? __builtin_ctzll(kept) : 0;
      kept &= kept - 1;
      const int gy = y0 + (k >> 3), gx = lane * 8 + (k & 7);
      gi[t] = gy * NR + gx;
      // the ORIGINAL score: the registers hold the suppressed one (0 inside a maximum's own suppression zone)
      v[t] = has ? S[gi[t]] : 0.f;
      pass[t] = has && !(gx < border || gx > NR - border || gy < border || gy > NR - border);
      if (out && has) out[img + gi[t]] = v[t];
    }
    u64 pm[4];
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pass[t] = pass[t] && !(v[t] < thr);
      pm[t] = __builtin_amdgcn_ballot_w64(pass[t]);
      cnt += __builtin_popcountll(pm[t]);
    }
    if (cnt != 0) {                                                    // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(&cand_cnt[b], cnt);
      base = __builtin_amdgcn_readfirstlane(base);
      const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int pos = base + __builtin_popcountll(pm[t] & lt);
        if (pass[t] && pos < cand_cap) cand[(size_t)b * cand_cap + pos] = nms_key(v[t], gi[t]);
        base += __builtin_popcountll(pm[t]);
      }
    }
  }
}

// planes: 2 x [B][64][64] 64-bit words of scratch
void launch_nms512_candidates(const float* heat, float* out, void* planes, int B, float thr, int border, u64* cand, int* cand_cnt,
                              int cand_cap, hipStream_t st) {
  u64* P0 = reinterpret_cast<u64*>(planes);
  u64* P1 = P0 + (size_t)B * NBAND * 64;
  (void)hipMemsetAsync(cand_cnt, 0, (size_t)B * sizeof(int), st);
  const dim3 grid(B * NBAND), block(64);
  hipLaunchKernelGGL(nms512_kernel<0>, grid, block, 0, st, heat, (const u64*)nullptr, P0, (float*)nullptr, B, thr, border, cand, cand_cnt, cand_cap);
  hipLaunchKernelGGL(nms512_kernel<1>, grid, block, 0, st, heat, (const u64*)P0, P1, (float*)nullptr, B, thr, border, cand, cand_cnt, cand_cap);
  hipLaunchKernelGGL(nms512_kernel<2>, grid, block, 0, st, heat, (const u64*)P1, (u64*)nullptr, out, B, thr, border, cand, cand_cnt, cand_cap);
}

}  // namespace airfe
