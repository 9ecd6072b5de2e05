// airfe — LightGlue "post-attention block" in ONE kernel:
//     msg = Wo . attn + bo ;  h = W1 . cat(x, msg) + b1 ;  h = GELU(LayerNorm(h)) ;  x += W2 . h + b2
// (reference model: LightGlue SelfBlock / CrossBlock tail — out_proj / to_out, ffn.0, ffn.1 LayerNorm, ffn.2 GELU, ffn.3,
//  residual; executed by the reference through TensorRT, src/light_glue.cpp:76-113 infer()).
//
// Why: as four launches (3 GEMMs + LayerNorm/GELU) these are 63 % of LightGlue's linear FLOPs but moved 8.7 KB of
// activations per token through HBM and ran at ~350 TFLOP/s.  Here a wave owns 32 tokens for the whole chain:
//   * swapped-operand MFMA (A = weight rows, B = tokens): the accumulator layout of one GEMM, rounded to 2 bytes, IS the
//     B-fragment layout of the next (lane (token, g) holds 8 consecutive features per tile pair), so msg and h never leave
//     registers — no LDS round trip, no cross-wave LayerNorm reduction (a token's 512 features live in 4 lanes of one wave);
//   * HBM traffic per token: read attn 512 B + x 512 B + fp32 residual 1 KB, write x 512 B + residual 1 KB (2.5x less);
//   * the 896 KB of weights per block stream through a 4 x 32 KiB LDS ring by LDS-DMA (prefetch distance 3 stages) in the
//     exact order the MFMAs consume them ("fragment-linear" packing done once on the host), one s_barrier per stage.
// The kernel is LDS-read bound by construction (each 1 KB weight fragment feeds only 2 MFMAs per wave): ~50 % of the
// MFMA peak is the ceiling of this decomposition; it is still 3x the separate GEMMs.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr_lb;

// LDS-DMA, scalar-base form: 16 bytes per lane from sbase + voff to LDS address M0 + lane*16.  The base is a RUNNING scalar
// pointer: with per-stage constant offsets hipcc hoisted all 224 64-bit piece addresses out of the tile loop and spilled them.
__device__ __forceinline__ void lb_glds16(unsigned voff, const void* sbase, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_off)
               : "memory");
}

// one ring slot = KS k-steps x 8 feature tiles of 1 KiB fragments: KS = 8 -> 64 KiB stages (2 + 8 + 4 per block),
// KS = 4 -> 32 KiB stages (4 + 16 + 8); always two slots
#ifndef LB_SLOTS_DEF
#define LB_SLOTS_DEF 4
#endif
constexpr int LB_SLOTS = LB_SLOTS_DEF;             // ring depth: the stage fetched at a boundary is consumed LB_SLOTS - 1 boundaries later
#ifndef LB_NM_DEF
#define LB_NM_DEF 2
#endif
constexpr int LB_NM = LB_NM_DEF;                   // 16-token tiles per wave: 1 -> 8 waves (two per SIMD), 2 -> 4 waves
// geometry: LB_NM 16-token tiles per wave, LB_NW waves, LB_KS k-steps per stage.
//   (2, 4, 8): 128 tokens per workgroup, one workgroup per CU, 512 registers per wave
//   (1, 4, 4):  64 tokens per workgroup, TWO workgroups per CU (72 KiB LDS, 256 registers): two waves per SIMD without a shared
//               barrier, at the price of streaming the weights once per 64 tokens
#ifndef LB_NW_DEF
#define LB_NW_DEF 4
#endif
#ifndef LB_KS_DEF
#define LB_KS_DEF 4
#endif
constexpr int LB_NW = LB_NW_DEF, LB_KS = LB_KS_DEF;
constexpr int LB_STAGE = LB_KS * 8192;
constexpr int LB_NSTAGE = LGB_STREAM_STAGES * 4 / LB_KS;
constexpr int LB_PARAM_OFF = LB_SLOTS * LB_STAGE;
constexpr int LB_LDS = LB_PARAM_OFF + LGB_PARAM_FLOATS * 4;
// parameter block (floats): bo[256] | b1[512] | gamma[512] | beta[512] | b2[256]
constexpr int LB_BO = 0, LB_B1 = 256, LB_GAMMA = 768, LB_BETA = 1280, LB_B2 = 1792;

__device__ __forceinline__ float lb_gelu(float y) {          // exact-erf GELU, Abramowitz & Stegun 7.1.26 (as ln_gelu_kernel)
  const float x = fabsf(y) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfa = 1.0f - poly * t * __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
  return 0.5f * y * (1.0f + copysignf(erfa, y));
}

// ds_read_b128 issued from inline asm: the read stays where it is written (hipcc sinks a plain LDS load down to its first
// use, and with 2 MFMAs per fragment every pair then waits out a full LDS latency; sched_group_barrier pins the order too
// but its solver does not terminate in reasonable time on a 1800-MFMA region).  The matching lgkmcnt wait is placed by hand.
template <class V>
__device__ __forceinline__ void lb_ds_read(V& dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off) : "memory");
}
// 128 MFMAs of one stage: 8 k-steps x 8 feature tiles x 2 token tiles; B fragments b[OFF + ks][m] (OFF is a compile-time
// constant: every register array here must be indexed by constants only, or hipcc demotes it to scratch).  The 8 weight
// fragments of k-step ks+1 are requested before the 16 MFMAs of k-step ks issue and awaited after them.
template <class P, int OFF, int N, int NM, int KS>
__device__ __forceinline__ void lb_stage_mfma(unsigned sb, const typename P::vec8 (&b)[N][NM], f32x4 (&acc)[8][NM]) {
  if constexpr (NM == 1) {
    // two waves per SIMD: the sibling wave's MFMAs cover this wave's LDS latency; one fragment buffer (32 registers)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename P::vec8 wf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) lb_ds_read(wf[t], sb, (ks * 8 + t) * 1024);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t][0] = P::mfma(wf[t], b[OFF + ks][0], acc[t][0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    typename P::vec8 wf[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) lb_ds_read(wf[0][t], sb, t * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < 8; ++t) lb_ds_read(wf[(ks + 1) & 1][t], sb, ((ks + 1) * 8 + t) * 1024);
        __builtin_amdgcn_sched_barrier(0);   // or the MFMAs below are hoisted above the reads and the two buffers collapse into one
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[t][m] = P::mfma(wf[ks & 1][t], b[OFF + ks][m], acc[t][m]);
      if (ks + 1 < KS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}
// accumulators start at the bias of their 4 features
template <int NM>
__device__ __forceinline__ void lb_init_acc(const float* bias, int g, f32x4 (&acc)[8][NM]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (t >> 1) * 32 + g * 8 + (t & 1) * 4);
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[t][m] = bv;
  }
}
// features tp*32 + g*8 .. +7 of both token tiles, rounded to 2 bytes = the next GEMM's B fragments dst[DST + tp][m]
template <class P, int DST, int N, int NM>
__device__ __forceinline__ void lb_pack_chunk(const f32x4 (&acc)[8][NM], typename P::vec8 (&dst)[N][NM]) {
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      uint4 u;
      u.x = P::pack2(acc[2 * tp][m][0], acc[2 * tp][m][1]);
      u.y = P::pack2(acc[2 * tp][m][2], acc[2 * tp][m][3]);
      u.z = P::pack2(acc[2 * tp + 1][m][0], acc[2 * tp + 1][m][1]);
      u.w = P::pack2(acc[2 * tp + 1][m][2], acc[2 * tp + 1][m][3]);
      dst[DST + tp][m] = __builtin_bit_cast(typename P::vec8, u);
    }
}

// Code size matters here: fully unrolled, the 14 stages are ~100 KB of instructions, more than the 64 KB instruction cache
// two CUs share, and every tile re-fetched all of it from L2 (measured: removing the MFMAs AND the LDS reads from that
// version barely changed its run time).  So each GEMM is a ROLLED loop over 128-feature chunks whose body always uses the
// same registers; the packed results enter msgf / hf through a 4-entry shift (register moves, ~300 per tile) instead of
// through chunk-dependent register indices.
// NM = 16-token tiles per wave, NW waves, KS k-steps per weight stage
template <class P, int NM, int NW, int KS>
__global__ __launch_bounds__(NW * 64, (NM == 1 && NW == 4) ? 2 : 1) void lg_block_kernel(LgBlockArgs a) {
  constexpr int NT = NW * 64;                       // threads
  constexpr int TW = NW * NM * 16;                  // tokens per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr_lb)smem);
  float* prm = reinterpret_cast<float*>(smem + LB_PARAM_OFF);
  for (int i = tid; i < LGB_PARAM_FLOATS; i += NT) prm[i] = a.params[i];
  __syncthreads();

  const unsigned wvoff = tid * 16;
  const char* wq = reinterpret_cast<const char*>(a.wstream);        // wave-uniform cursor into the cyclic weight stream
  const unsigned sbase = lds_base + lane * 16;      // LDS byte address of this lane's 16 bytes inside a fragment
  int sg = 0;                                       // stage (0..13) the NEXT LB_DMA fetches; ring slot = sg & 1
#define LB_DMA()                                                                                           \
  {                                                                                                        \
    const unsigned dst_ = lds_base + (sg % LB_SLOTS) * LB_STAGE + wave * 1024;                             \
    _Pragma("unroll") for (int j_ = 0; j_ < LB_STAGE / (NT * 16); ++j_) lb_glds16(wvoff, wq + j_ * (NT * 16), dst_ + j_ * (NT * 16)); \
    wq += LB_STAGE;                                                                                        \
    if (++sg == LB_NSTAGE) { sg = 0; wq = reinterpret_cast<const char*>(a.wstream); }                      \
  }
  // stage boundary: my pieces of the current stage have landed (issued one stage = ~2000 MFMA cycles ago), everyone's have
  // (barrier), the other slot — read during the previous stage — is refilled with the next stage; then 128 MFMAs.
  // (LB_NSTAGE is even: the stage being computed sits in the slot the next LB_DMA does NOT target.)
#define LB_RUN(ARR, OFF, N)                                                                                \
  {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LB_SLOTS - 2) * (LB_STAGE / (NT * 16))) : "memory");         \
    __builtin_amdgcn_s_barrier();                                                                          \
    const unsigned cur_ = sbase + ((sg + 1) % LB_SLOTS) * LB_STAGE;                                        \
    LB_DMA()                                                                                               \
    lb_stage_mfma<P, OFF, N, NM, KS>(cur_, ARR, acc);                                                      \
  }
  /* 8 k-steps of B fragments starting at OFF: one 64 KiB stage or two 32 KiB stages */
#define LB_RUN8(ARR, OFF, N)                                                                               \
  {                                                                                                        \
    LB_RUN(ARR, OFF, N)                                                                                    \
    if constexpr (KS <= 4) LB_RUN(ARR, (OFF) + KS, N)                                                       \
    if constexpr (KS == 2) { LB_RUN(ARR, (OFF) + 4, N) LB_RUN(ARR, (OFF) + 6, N) }                          \
  }

  const int ntiles = a.M / TW;
  int tile = blockIdx.x;
  static_assert(LB_NSTAGE % LB_SLOTS == 0, "ring slots must stay aligned across tiles");
  if (tile < ntiles) {
#pragma unroll
    for (int d = 0; d < LB_SLOTS - 1; ++d) LB_DMA()
  }

  // ---- token fragments straight from HBM: lane (token l15, g) holds k = ks*32 + g*8 .. +7.  With one wave per SIMD nothing
  // hides a load but the wave's own MFMAs, so every HBM read is issued a stage or more before its first use: the attention
  // rows of tile i+1 during the last two stages of tile i, x at the top of the tile (first used in stage 2), the fp32
  // residual at the top of each ffn.3 chunk.  (The memory-clobbering asm of the stage boundaries keeps the loads in place.)
  typename P::vec8 obf[8][NM];
#define LB_LOAD_ATTN(TILE)                                                                                 \
  _Pragma("unroll") for (int m = 0; m < NM; ++m) {                                                         \
    const size_t row_ = (size_t)((TILE) * TW + wave * (16 * NM) + m * 16 + l15) * 256 + g * 8;             \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                       \
      obf[ks][m] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(a.attn + row_ + ks * 32)); \
  }
  constexpr bool PREF = NM == 2;                    // one wave per SIMD: nothing but explicit prefetch hides HBM latency
  if (PREF && tile < ntiles) { LB_LOAD_ATTN(tile) }

  for (; tile < ntiles; tile += gridDim.x) {
    const int tok0 = tile * TW + wave * (16 * NM);
    typename P::vec8 xf[8][NM], msgf[8][NM], hf[16][NM];
    if (!PREF) { LB_LOAD_ATTN(tile) }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const size_t row = (size_t)(tok0 + m * 16 + l15) * 256 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xf[ks][m] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(a.xb + row + ks * 32));
    }
    f32x4 acc[8][NM];

    // ---- out-projection: msg[256] = Wo . attn + bo           (stages 0..1)
#pragma unroll 1
    for (int fc = 0; fc < 2; ++fc) {
      lb_init_acc<NM>(prm + LB_BO + fc * 128, g, acc);
      LB_RUN8(obf, 0, 8)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) msgf[i][m] = msgf[i + 4][m];
      lb_pack_chunk<P, 4, 8, NM>(acc, msgf);
    }
    // ---- ffn.0: h[512] = W1 . cat(x, msg) + b1 ; LayerNorm statistics from the fp32 accumulators   (stages 2..9)
    float s1[NM], s2[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
#pragma unroll 1
    for (int fc = 0; fc < 4; ++fc) {
      lb_init_acc<NM>(prm + LB_B1 + fc * 128, g, acc);
      LB_RUN8(xf, 0, 8)
      LB_RUN8(msgf, 0, 8)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            s1[m] += acc[t][m][r];
            s2[m] = fmaf(acc[t][m][r], acc[t][m][r], s2[m]);
          }
#pragma unroll
      for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) hf[i][m] = hf[i + 4][m];
      lb_pack_chunk<P, 12, 16, NM>(acc, hf);
    }
    // ---- LayerNorm(512) + GELU, in registers: a token's features sit in the 4 lanes {l15, l15+16, l15+32, l15+48}
    float mean[NM], rstd[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const float a1 = rows_sum(s1[m]), a2 = rows_sum(s2[m]);
      mean[m] = a1 * (1.0f / 512.0f);
      const float var = fmaxf(a2 * (1.0f / 512.0f) - mean[m] * mean[m], 0.f);
      rstd[m] = 1.0f / sqrtf(var + 1e-5f);
    }
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {                    // 4 k-steps per trip: take the front of the shift register, append behind
      typename P::vec8 res[4][NM];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* gp = prm + LB_GAMMA + (q * 4 + i) * 32 + g * 8;
        const float* bp = prm + LB_BETA + (q * 4 + i) * 32 + g * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          float v[8];
          unpack8<P>(__builtin_bit_cast(uint4, hf[i][m]), v);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = lb_gelu(fmaf((v[e] - mean[m]) * rstd[m], g0[e], b0[e]));
            v[4 + e] = lb_gelu(fmaf((v[4 + e] - mean[m]) * rstd[m], g1[e], b1[e]));
          }
          res[i][m] = __builtin_bit_cast(typename P::vec8, pack8<P>(v));
        }
      }
#pragma unroll
      for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) hf[i][m] = hf[i + 4][m];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NM; ++m) hf[12 + i][m] = res[i][m];
    }
    // ---- ffn.3 + residual: x += W2 . h + b2     (stages 10..13)
#pragma unroll 1
    for (int fc = 0; fc < 2; ++fc) {
      lb_init_acc<NM>(prm + LB_B2 + fc * 128, g, acc);
      [[maybe_unused]] f32x4 rsd[NM][4][2];
      if constexpr (PREF) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const float* xr = a.x32 + (size_t)(tok0 + m * 16 + l15) * 256 + fc * 128 + g * 8;
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            rsd[m][tp][0] = *reinterpret_cast<const f32x4*>(xr + tp * 32);
            rsd[m][tp][1] = *reinterpret_cast<const f32x4*>(xr + tp * 32 + 4);
          }
        }
      }
      LB_RUN8(hf, 0, 16)
      if (PREF && fc == 1 && tile + (int)gridDim.x < ntiles) { LB_LOAD_ATTN(tile + gridDim.x) }
      LB_RUN8(hf, 8, 16)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const size_t row = (size_t)(tok0 + m * 16 + l15) * 256 + fc * 128 + g * 8;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          float* xr = a.x32 + row + tp * 32;
          f32x4 r0, r1;
          if constexpr (PREF) { r0 = rsd[m][tp][0]; r1 = rsd[m][tp][1]; }
          else { r0 = *reinterpret_cast<const f32x4*>(xr); r1 = *reinterpret_cast<const f32x4*>(xr + 4); }
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[2 * tp][m][e] + r0[e];
            v[4 + e] = acc[2 * tp + 1][m][e] + r1[e];
          }
          *reinterpret_cast<f32x4*>(xr) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(xr + 4) = f32x4{v[4], v[5], v[6], v[7]};
          *reinterpret_cast<uint4*>(a.xb + row + tp * 32) = pack8<P>(v);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetched first stage of a tile that does not exist
#undef LB_DMA
#undef LB_LOAD_ATTN
#undef LB_RUN
#undef LB_RUN8
}

// requires M % (LB_NW * LB_NM * 16) == 0 (128 in the default geometry).  attn / xb: [M][256] 2-byte rows, x32: [M][256] fp32 residual stream (xb and x32 updated in place)
void launch_lg_block(int prec, const LgBlockArgs& a, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lg_block_kernel<PBF16, LB_NM, LB_NW, LB_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lg_block_kernel<PF16, LB_NM, LB_NW, LB_KS>), hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS);
    attr_done = true;
  }
  constexpr int TW = LB_NW * LB_NM * 16, WGS = (LB_NM == 1 && LB_NW == 4) ? 512 : 256;
  const int ntiles = a.M / TW;
  const int grid = ntiles < WGS ? ntiles : WGS;
  if (prec == 1) hipLaunchKernelGGL((lg_block_kernel<PF16, LB_NM, LB_NW, LB_KS>), dim3(grid), dim3(LB_NW * 64), LB_LDS, st, a);
  else hipLaunchKernelGGL((lg_block_kernel<PBF16, LB_NM, LB_NW, LB_KS>), dim3(grid), dim3(LB_NW * 64), LB_LDS, st, a);
}

}  // namespace airfe
