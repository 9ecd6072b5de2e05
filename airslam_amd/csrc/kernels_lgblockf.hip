// airfe — LightGlue post-attention block, FEATURE-SPLIT form:  msg = Wo·attn + bo;  h = GELU(LN(W1·cat(x, msg) + b1));
// x += W2·h + b2  for 128 tokens per workgroup, as ONE kernel.
//
// The first fused form (lg_block_kernel, retired) kept a token's whole chain inside one wave (msg and h never left registers),
// which forced every wave to read every weight through LDS, one wave per SIMD, one workgroup-wide barrier per 32 KiB of weights:
// 116 us per call at 51200 tokens against 80 for this form.  This is the other cut: the 8 waves (two per SIMD) split the OUTPUT FEATURES of each GEMM, every wave fetches only its own rows of the
// packed weight slabs straight from global memory into MFMA A-fragments (each weight byte is read once per workgroup,
// no LDS, no barrier), and the activations are the B operand shared through LDS: attn / x / msg tiles of [128][256] 2-byte
// (64 KiB) and the h tile of [128][512] (128 KiB, over the dead attn/x and msg tiles).  Five workgroup barriers.
//   LDS:  R0 [0, 64K) attn -> x -> h(lo)   R1 [64K, 128K) msg -> h(hi)   ST [128K, 136K) LayerNorm partial sums
//   rows are swizzled by XOR of the 16-byte piece index with (token & 15): conflict-free ds_read_b128 / ds_write_b128.
// Weights are the SAME packed slabs the separate launches use (LinW::w, [N/64][K/64][8 KiB]); the feature order inside a slab
// (slab_row_to_feature) is what makes a lane's accumulators 8 contiguous features of one token.
//
// One workgroup = one pass over 128 tokens.  (Dealing the token tiles out evenly over 256 persistent workgroups in passes of
// 6-7 tiles — 51200 tokens are 400 fixed tiles on 256 CUs, a second round for 144 of them — measured the same: with every CU
// in the same phase at the same time the passes do not get shorter in proportion to their tokens.  Round 2 tried the cheap form of the
// same idea — workgroups 0..255 with 112 tokens, the second round's 235 with 96 — and measured 1.91 -> 1.90 / 1.94 -> 1.88 ms of linears
// per step on one box: noise; not kept.)
//
// What the per-phase timers said (round 1, 128-token pass, us): attn 3.0 | out-proj 3.9 | ffn.0 5.1 + 1.6 + 4.0 | LN sums 2.9
// | GELU 10.3 | ffn.3 4.7 | epilogue 2.6.  Removing the LDS reads changes nothing, removing the weight loads 10 %: the GEMMs
// run at 75-80 % of the MFMA rate (a pure MFMA stream: 85 % with two waves per SIMD); the GELU phase is VALU-bound (see lf_gelu2).
#include "common.h"
#include "kernels.h"
#ifdef LF_TIMING   // per-phase wall-clock timers of wave 0 (tools/lf_timing.py; a measurement build, never the shipped library)
// -DLF_TWICE on top: every workgroup runs its pass a second time (results are garbage: x += ... twice) with its timers in slots 16..31 — the same
// instructions on the same data, but fetched from a warm instruction cache: what the straight-line, executed-once code of a pass costs in fetches.
__device__ unsigned long long lf_dbg[32];
#define LF_T(i) { const long long now_ = wall_clock64(); if (L.wave == 0 && L.lane == 0) atomicAdd(&lf_dbg[L.tb + i], (unsigned long long)(now_ - tp_)); tp_ = now_; }
extern "C" void airfe_dbg_lf(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[32] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lf_dbg), z, sizeof(z)); }
  else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lf_dbg), 32 * sizeof(unsigned long long));
}
#else
#define LF_T(i)
#endif
#ifdef LF_TIMING2   // one level below LF_TIMING (tools/lf_timing2.py; a measurement build): inside the K loop of ONE GEMM of a pass — ffn.0's x half, 4 slabs of 64-feature
// tiles — the shader-clock time of every wave between: trip start | weight prefetch issued | this trip's weights landed (vmcnt) | per 4-tile group: B fragments requested
// (ds_read_b128) | landed (lgkmcnt(0)) | MFMAs issued | register moves.  The two forced waits replace the compiler's counted ones (a measurement build).
__device__ unsigned long long lf2_dbg[16];
extern "C" void airfe_dbg_lf2(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lf2_dbg), z, sizeof(z)); }
  else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lf2_dbg), 16 * sizeof(unsigned long long));
}
#define LF2_NOW(v) { __builtin_amdgcn_sched_barrier(0); v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#endif

namespace airfe {

constexpr int LF_R0 = 0, LF_R1 = 65536, LF_ST = 131072;
constexpr int LF_LDS = LF_ST + 8 * 128 * 8;

// WEIGHT LAYOUT (round 6).  The A fragments come straight from global memory: one 16-byte load per lane, tile and 32-wide K half.  In the packed SLAB image
// (LinW::w: [64 rows][64 k], 128-byte rows, swizzled — the image the LDS-staged kernels want) lane (l15, g) of such a load reads row l15, 16-byte piece g: the 16
// lanes of a quarter-wave touch 16 DIFFERENT 128-byte lines, 16 bytes of each, and the four quarter-waves touch the same 16 lines again — 64 tag look-ups for 1 KB.
// The timers inside the K loop (LF_TIMING2, profiles/r06_att_blockf.txt) put 72 % of a trip into ISSUING its eight loads: ~67 cycles per load, the vector
// memory path of the CU saturated by look-ups, not by bytes.  LinW::wf is the same bytes in FRAGMENT order — [16-row tile T][slab s][half h][lane][16 B]:
// a wave's load is 1 KB in a row, eight whole lines.  LF_FRAG = 0 keeps the slab image (A/B builds).
#ifndef LF_FRAG
#define LF_FRAG 1
#endif
constexpr int LF_SS = LF_FRAG ? 2048 : SLAB_BYTES;                               // bytes from a tile's slab s to its slab s + 1
constexpr int lf_ts(int K) { return LF_FRAG ? (K / 64) * 2048 : 2048; }         // bytes from tile T to tile T + 1 of a linear with K inputs
constexpr int LF_TS256 = lf_ts(256), LF_TS512 = lf_ts(512);
// byte offset of (64-feature block cb, tile t0 of it, slab 0) in the weights of a linear with K inputs
__device__ __forceinline__ size_t lf_woff(int K, int cb, int t0) {
  return LF_FRAG ? (size_t)(cb * 4 + t0) * (K / 64) * 2048 : (size_t)cb * (K / 64) * SLAB_BYTES + (size_t)t0 * 2048;
}

__device__ __forceinline__ void lf_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

// GELU(y) = 0.5 y (1 + erf(y / sqrt 2)) = max(y, 0) - 0.5 |y| erfc(|y| / sqrt 2), with erfc(x) = 2^(-x^2 log2 e + R(x)),
// R = log2(erfcx) a degree-8 polynomial in u = x/3 - 1 on x in [0, 6] (Chebyshev fit, |dR| < 1e-5; beyond 6 erfc < 3e-17).
// One transcendental instead of the two (rcp, exp) of Abramowitz & Stegun 7.1.26 in ln_gelu_kernel, no cancellation on the
// negative side (max abs error 1.1e-6, relative 2e-4 in the far negative tail where 7.1.26 returns 0), and written on float
// PAIRS so that the multiply-adds become v_pk_fma_f32 / v_pk_mul_f32: this epilogue is 128 elements per lane per workgroup.
#ifndef LF_GELU_ERFC
// Cheaper form (default): GELU(y) = y (0.5 + E(t)),  E(t) = 0.5 erf(t / sqrt 2) = t P(t^2),  t = clamp(y, +-4.5): an odd polynomial of degree
// 17 (least-squares Chebyshev fit in x = 2 t^2 / 4.5^2 - 1, weighted by t^2 = the weight of E in the result), no transcendental: 13
// packed multiply-adds + 2 v_med3 per PAIR against ~20 + 2 v_exp_f32 (quarter rate) for the erfc form below.  Max abs error of the
// fp32 evaluation 4.2e-5 (at y = 0.6; beyond |y| = 4.5 the result is y resp. y * 2e-6): a tenth of the 2-byte rounding the h tile
// gets next.  -DLF_GELU_ERFC selects the erfc form (max abs error 1.1e-6).
__device__ __forceinline__ f32x2 lf_gelu2(f32x2 y) {
  const f32x2 t = {__builtin_amdgcn_fmed3f(y.x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(y.y, -4.5f, 4.5f)};
  const f32x2 x = (t * t) * (2.0f / 20.25f) - 1.0f;
  f32x2 r = x * 0.0031705170404165983f - 0.009152771905064583f;
  r = r * x + 0.012448843568563461f;
  r = r * x - 0.016971856355667114f;
  r = r * x + 0.027542514726519585f;
  r = r * x - 0.040475402027368546f;
  r = r * x + 0.05482625961303711f;
  r = r * x - 0.07717858254909515f;
  r = r * x + 0.15690208971500397f;
  return y * (t * r + 0.5f);
}
// Four pairs at once, one Horner level at a time (round 5).  Evaluated pair by pair the polynomial is ONE dependent chain of 13 packed operations: hipcc, short of
// registers in this phase, emits exactly that — every v_pk_fma_f32 waits for the one before it and is followed by the wait state the hazard asks for (533 s_nop in
// the LayerNorm + GELU phase of a 112-token pass).  Level by level over four independent pairs the dependent operations are four instructions apart.  The same
// operations per element: the same bits.
__device__ __forceinline__ void lf_gelu2x4(f32x2 (&y)[4]) {
  f32x2 t[4], x[4], r[4];
#define LF_G4(expr) _Pragma("unroll") for (int k = 0; k < 4; ++k) { expr; } __builtin_amdgcn_sched_barrier(0);
  LF_G4((t[k] = f32x2{__builtin_amdgcn_fmed3f(y[k].x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(y[k].y, -4.5f, 4.5f)}))
  LF_G4(x[k] = t[k] * t[k])
  LF_G4(x[k] = x[k] * (2.0f / 20.25f) - 1.0f)
  LF_G4(r[k] = x[k] * 0.0031705170404165983f - 0.009152771905064583f)
  LF_G4(r[k] = r[k] * x[k] + 0.012448843568563461f)
  LF_G4(r[k] = r[k] * x[k] - 0.016971856355667114f)
  LF_G4(r[k] = r[k] * x[k] + 0.027542514726519585f)
  LF_G4(r[k] = r[k] * x[k] - 0.040475402027368546f)
  LF_G4(r[k] = r[k] * x[k] + 0.05482625961303711f)
  LF_G4(r[k] = r[k] * x[k] - 0.07717858254909515f)
  LF_G4(r[k] = r[k] * x[k] + 0.15690208971500397f)
  LF_G4(r[k] = t[k] * r[k] + 0.5f)
  LF_G4(y[k] = y[k] * r[k])
#undef LF_G4
}
#else
__device__ __forceinline__ f32x2 lf_gelu2(f32x2 y) {
  const f32x2 a = __builtin_elementwise_abs(y);
  f32x2 x = a * 0.70710678118654752f;
  x = __builtin_elementwise_min(x, f32x2{6.0f, 6.0f});
  const f32x2 u = x * (1.0f / 3.0f) - 1.0f;
  f32x2 r = u * 0.008399954997003078f - 0.02763195149600506f;
  r = r * u + 0.04838801920413971f;
  r = r * u - 0.08357185125350952f;
  r = r * u + 0.1567264348268509f;
  r = r * u - 0.28920042514801025f;
  r = r * u + 0.5534077286720276f;
  r = r * u - 1.3146467208862305f;
  r = r * u - 2.481963634490967f;
  const f32x2 p = (x * x) * -1.4426950408889634f + r;
  const f32x2 q = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  return __builtin_elementwise_max(y, f32x2{0.f, 0.f}) - (x * 0.70710678118654752f) * q;
}
__device__ __forceinline__ void lf_gelu2x4(f32x2 (&y)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) y[k] = lf_gelu2(y[k]);
}
#endif

// [16 nmt tokens][256] 2-byte rows of `src` -> LDS region by LDS-DMA: 8 nmt wave-instructions of 1 KiB (two rows each), nmt per
// wave; lane i of an instruction lands at +16 i, so the swizzle is applied on the global side.
__device__ __forceinline__ void lf_stage_rows(const uint16_t* src, int row0, int nmt, int region, int wave, int lane) {
  for (int i = 0; i < nmt; ++i) {
    const int inst = wave * nmt + i;
    const int r = inst * 2 + (lane >> 5), pp = lane & 31;
    lf_glds16(src + (size_t)(row0 + r) * 256 + ((pp ^ (r & 15)) << 3), (unsigned)(region + inst * 1024));
  }
}

template <class P>
__device__ __forceinline__ typename P::vec8 lf_ldg(const char* p) {
  return __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(p));
}

// The first slab's A fragments of a GEMM: issued ahead of the phase that precedes it (barrier, pack, GELU), so that a GEMM
// never starts by waiting one L2 round trip.
template <class P, int NT, int TS>
__device__ __forceinline__ void lf_first(typename P::vec8 (&cur)[NT][2], const char* w0, const char* w1) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    cur[t][0] = lf_ldg<P>(w0 + t * TS);
    cur[t][1] = lf_ldg<P>(w1 + t * TS);
  }
}

struct LfNoHook { __device__ __forceinline__ void operator()(int) const {} };

#ifndef LF_GELU_X4
#define LF_GELU_X4 1
#endif


// acc[t][m] += W(tiles t of this wave, slabs 0..nslab-1) x B(token tiles m < NMT): w0 / w1 = this lane's fragment address in
// slab 0, tile 0 for the two 32-wide halves of the slab's 64-wide K chunk; tile t at +2048 t, slab s at +8192 s.  `cur` holds
// slab 0 on entry; the next slab's fragments are fetched while the current one feeds 2 NT NMT MFMAs, and during the last slab
// the fetch goes to n0 / n1 (the first slab of whatever this wave multiplies next), which `cur` holds on exit.  (Fetching two
// or three slabs ahead for the 32-feature GEMMs measured no faster on the whole block — round 1, and again in round 5 at the kernel level under rocprofv3: a rolled
// two-slab form of this loop for the 112-token passes, 114.9 / 116.6 us against 114.5 / 114.3, profiles/r05_probe_blockf_depth2.txt.  Nor do the `cur = nxt`
// register moves that end a trip cost anything measurable: two steps per trip with the fragment sets changing roles — VALU : MFMA inside the loops 0.25 instead
// of 0.6-0.9 — ran 121.6 / 121.3 us against 118.3 / 123.0 on one box, same file.)
// TS / TSN: tile stride of this linear's weights / of the linear whose first slab is prefetched during the last trip (n0 / n1).
template <class P, int NT, int NMT, int TS, int TSN, class Hook = LfNoHook, bool SWAP = false, bool TIMED = false>
__device__ __forceinline__ void lf_mma(f32x4 (&acc)[NT][NMT], typename P::vec8 (&cur)[NT][2], const char* w0, const char* w1, int nslab,
                                       const char* n0, const char* n1, const char* breg, int pitch, int l15, int g, Hook hook = Hook()) {
  const char* brow = breg + l15 * pitch;
  if constexpr (NMT <= 4) {
    // Small token tiles (32 / 64 tokens per workgroup: 1-16 pairs per call).  A slab feeds only 4 NT NMT MFMAs (~0.1-0.25 us of matrix pipe) but
    // the phase timers say a slab STEP takes 0.6-0.85 us (profiles/r04_lf_blockf_phase_timers.txt): one slab ahead, the GEMMs wait an L2 round trip
    // per slab.  Here ALL of a GEMM's remaining slabs are requested up front (nslab is 4 or 8: the trip is unrolled completely so that every
    // fragment has its own registers and the compiler's wait counts stay exact — a runtime loop over a register ring made it wait for everything
    // at the back edge: r04 probe 1 measured no gain from that form), then consumed in order.  Same MFMAs in the same order: bit-identical.
    typename P::vec8 ring[8][NT][2];
    auto fetch = [&](int k, typename P::vec8 (&dst)[NT][2]) {
      const char* p0 = k == nslab ? n0 : w0 + k * LF_SS;
      const char* p1 = k == nslab ? n1 : w1 + k * LF_SS;
      const int ts = k == nslab ? TSN : TS;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        dst[t][0] = lf_ldg<P>(p0 + t * ts);
        dst[t][1] = lf_ldg<P>(p1 + t * ts);
      }
    };
    constexpr int AHEAD = (NT == 4 ? 4 : 8) / (NMT > 2 ? 2 : 1);   // 32-token tiles: 4 slabs of the 64-feature GEMM (128 registers) / all 8 of a 32-feature one in flight; 64-token tiles: half (registers)
#pragma unroll
    for (int d = 1; d <= AHEAD; ++d)
      if (d <= nslab) fetch(d, ring[d - 1]);
    __builtin_amdgcn_sched_barrier(0);      // keep the requests up here (hipcc sinks loads towards their first use)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nslab) {
        hook(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int boff = (((s * 8 + h * 4 + g) ^ l15) << 4);
          typename P::vec8 bf[NMT];
#pragma unroll
          for (int j = 0; j < NMT; ++j) bf[j] = lds_frag<P>(brow, boff + j * 16 * pitch);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NMT; ++j) {
              if constexpr (SWAP) acc[t][j] = P::mfma(bf[j], cur[t][h], acc[t][j]);
              else acc[t][j] = P::mfma(cur[t][h], bf[j], acc[t][j]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          cur[t][0] = ring[s % AHEAD][t][0];
          cur[t][1] = ring[s % AHEAD][t][1];
        }
        if (s + 1 + AHEAD <= nslab) fetch(s + 1 + AHEAD, ring[s % AHEAD]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
  typename P::vec8 nxt[NT][2];
#ifdef LF_NOLDS          // diagnostic build (tools/build_timing_variants.sh): the K loop WITHOUT its B-fragment reads from LDS — four fragments read once, reused by every group
  typename P::vec8 bf0[4];          // (garbage results; kernel times only: the ceiling of any fix of the LDS side of the loop)
#pragma unroll
  for (int j = 0; j < 4; ++j) bf0[j] = lds_frag<P>(brow, ((g ^ l15) << 4) + j * 16 * pitch);
#endif
#ifdef LF_TIMING2
  [[maybe_unused]] unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0;
  [[maybe_unused]] unsigned long long a_pref = 0, a_vm = 0, a_rd = 0, a_lgkm = 0, a_mma = 0, a_mov = 0, a_trips = 0;
#endif
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
#ifdef LF_TIMING2
    if constexpr (TIMED) LF2_NOW(q0)
#endif
    const bool last = s + 1 == nslab;
    const char* p0 = last ? n0 : w0 + (s + 1) * LF_SS;
    const char* p1 = last ? n1 : w1 + (s + 1) * LF_SS;
#ifdef LF_NOWEIGHTS      // diagnostic build (tools/build_timing_variants.sh): the K loop WITHOUT its weight stream (results are garbage) — the ceiling of any fix of that stream
    if (true) {
      (void)p0; (void)p1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        nxt[t][0] = cur[t][0];
        nxt[t][1] = cur[t][1];
      }
    } else
#endif
    if constexpr (TS == TSN) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        nxt[t][0] = lf_ldg<P>(p0 + t * TS);
        nxt[t][1] = lf_ldg<P>(p1 + t * TS);
      }
    } else {
      const int ts = last ? TSN : TS;                      // (wave-uniform)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        nxt[t][0] = lf_ldg<P>(p0 + t * ts);
        nxt[t][1] = lf_ldg<P>(p1 + t * ts);
      }
    }
    hook(s);                                // vector-memory work that must queue BEHIND this trip's prefetch (vmcnt retires in order)
    __builtin_amdgcn_sched_barrier(0);      // keep the whole prefetch at the top of the trip (hipcc sinks loads towards their use)
#ifdef LF_TIMING2
    if constexpr (TIMED) {
      LF2_NOW(q1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NT) : "memory");      // this trip's A fragments (requested one trip ago) have landed
      LF2_NOW(q2)
      a_pref += q1 - q0; a_vm += q2 - q1;
    }
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int boff = (((s * 8 + h * 4 + g) ^ l15) << 4);
#pragma unroll
      for (int mb = 0; mb < NMT; mb += 4) {
#ifdef LF_TIMING2
        if constexpr (TIMED) LF2_NOW(q3)
#endif
        typename P::vec8 bf[4];
#ifdef LF_NOLDS
        (void)boff;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = bf0[j];
#else
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (mb + j < NMT) bf[j] = lds_frag<P>(brow, boff + (mb + j) * 16 * pitch);
#endif
#ifdef LF_TIMING2
        if constexpr (TIMED) {
          LF2_NOW(q4)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          LF2_NOW(q5)
        }
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (mb + j < NMT) {
              if constexpr (SWAP) acc[t][mb + j] = P::mfma(bf[j], cur[t][h], acc[t][mb + j]);     // tokens x features: the transposed-V form
              else acc[t][mb + j] = P::mfma(cur[t][h], bf[j], acc[t][mb + j]);
            }
#ifdef LF_TIMING2
        if constexpr (TIMED) {
          LF2_NOW(q6)
          a_rd += q4 - q3; a_lgkm += q5 - q4; a_mma += q6 - q5;
        }
#endif
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      cur[t][0] = nxt[t][0];
      cur[t][1] = nxt[t][1];
    }
#ifdef LF_TIMING2
    if constexpr (TIMED) {
      LF2_NOW(q1)
      a_mov += q1 - q6; a_trips += 1;
    }
#endif
  }
#ifdef LF_TIMING2
  if constexpr (TIMED) {
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&lf2_dbg[0], a_pref); atomicAdd(&lf2_dbg[1], a_vm); atomicAdd(&lf2_dbg[2], a_rd); atomicAdd(&lf2_dbg[3], a_lgkm); atomicAdd(&lf2_dbg[4], a_mma);
      atomicAdd(&lf2_dbg[5], a_mov); atomicAdd(&lf2_dbg[6], a_trips); atomicAdd(&lf2_dbg[7], (unsigned long long)(2 * ((NMT + 3) / 4)) * a_trips);
      atomicAdd(&lf2_dbg[8], (unsigned long long)(2 * NT * NMT) * a_trips);      // MFMAs issued
    }
  }
#endif
}

struct LfLane {                       // per-lane constants of the whole kernel
  int lane, wave, l15, g, wf, cb, tp, fo0, fo1;
#ifdef LF_TIMING
  int tb;
#endif
  const char *wob, *w1b, *w2b;
};

// One pass over NMT token tiles (16 tokens each) starting at token `row0`; its attn tile is already in flight into R0.
// RELU = the SuperGlue propagation block (merge, mlp.0 + folded BatchNorm + ReLU, mlp.3: super_glue GNN layer) — the same three GEMMs
// with ReLU in place of LayerNorm + GELU.
// FOLD = the NEXT attention layer's projections computed here from the block's result while it is still in the workgroup (no re-read of x,
// one launch less per layer): 1 = the cross block's shared q/k projection (256 features, head-major) + V (transposed); 2 = the self block's
// q | k (512 features, rotary) + V.  Same fragments, same K order, bias after the sum and the same rotary code as gemmr_body
// (kernels_gemmr.hip) / the tiled kernels: the three forms give the same bits.
// FOLDO = the out-projection is already inside W1's message half (airfe_tuning::fold_out_proj, airfe_load.hip make_ffn0_folded): ffn.0 reads cat(x, attn) — the
// attn tile sits in R1 where msg would be, the x tile in R0, both in flight when the pass starts — and the 256 x 256 GEMM, the msg pack, one barrier and the
// mid-GEMM wait for the x tile are gone.  (Letting the x tile land UNDER the attention half — its DMA issued last, two slabs of weights ahead so that no
// counted wait catches it — measured the same, 1.84 / 1.87 against 1.87 / 1.87 ms of lg_gemm: the first barrier comes ~5 us after the launch whether it waits
// for two tiles or for one tile and 128 KB of weights, profiles/r05_fold_out_ab.txt; not kept.)
template <class P, int NMT, bool RELU, int FOLD, bool FOLDO>
__device__ __forceinline__ void lf_pass(const LgBlockFArgs& a, char* smem, const LfLane& L, int row0) {
  const int lane = L.lane, wave = L.wave, l15 = L.l15, g = L.g, wf = L.wf, cb = L.cb, tp = L.tp, fo0 = L.fo0, fo1 = L.fo1;
#ifdef LF_TIMING
  long long tp_ = wall_clock64();
  if (wave == 0 && lane == 0) atomicAdd(&lf_dbg[L.tb + 15], 1ull);
#endif
  typename P::vec8 c2[2][2], c4[4][2];                            // A fragments in flight: 32-feature GEMMs / the 64-feature one
  [[maybe_unused]] f32x4 bo2[2];
  f32x4 b14[4];                                                   // biases of the first two GEMMs: fetched with the attn tile
  if constexpr (FOLDO) {
    lf_first<P, 4, LF_TS512>(c4, L.w1b + 4 * LF_SS + fo0, L.w1b + 4 * LF_SS + fo1);
  } else {
    lf_first<P, 2, LF_TS256>(c2, L.wob + fo0, L.wob + fo1);
#pragma unroll
    for (int u = 0; u < 2; ++u) bo2[u] = *reinterpret_cast<const f32x4*>(a.bo + cb * 64 + tp * 32 + g * 8 + u * 4);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) b14[t] = *reinterpret_cast<const f32x4*>(a.b1 + wf * 64 + (t >> 1) * 32 + g * 8 + (t & 1) * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LF_T(0)

  // ---- msg = Wo attn + bo -> R1
  if constexpr (!FOLDO) {
    {
    f32x4 acc[2][NMT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < NMT; ++m) acc[u][m] = bo2[u];
    lf_mma<P, 2, NMT, LF_TS256, LF_TS256>(acc, c2, L.wob + fo0, L.wob + fo1, 4, L.wob + fo0, L.wob + fo1, smem + LF_R0, 512, l15, g);
    lf_first<P, 4, LF_TS512>(c4, L.w1b + 4 * LF_SS + fo0, L.w1b + 4 * LF_SS + fo1);  // lands while msg is packed and the barrier drains
    const int piece = cb * 8 + tp * 4 + g;
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[0][m][e];
        v[4 + e] = acc[1][m][e];
      }
      *reinterpret_cast<uint4*>(smem + LF_R1 + (m * 16 + l15) * 512 + ((piece ^ l15) << 4)) = pack8<P>(v);
    }
    }
    __syncthreads();                                              // msg complete, attn dead
  }
  LF_T(1)

  // ---- h = W1 cat(x, msg) + b1: the msg half first; the x tile's DMA into R0 is issued behind the msg half's LAST weight
  // prefetch — vmcnt retires in order, so issued any earlier every wait for weights would also wait for the HBM-latency DMA
  f32x4 h[4][NMT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < NMT; ++m) h[t][m] = b14[t];
  if constexpr (FOLDO) {                                          // the attention half (R1 holds the attn tile itself); the x tile landed with it
    lf_mma<P, 4, NMT, LF_TS512, LF_TS512>(h, c4, L.w1b + 4 * LF_SS + fo0, L.w1b + 4 * LF_SS + fo1, 4, L.w1b + fo0, L.w1b + fo1, smem + LF_R1, 512, l15, g);
    LF_T(2)
    LF_T(3)
  } else {
    auto xhook = [&](int s) { if (s == 3) lf_stage_rows(a.xb, row0, NMT, LF_R0, wave, lane); };
    lf_mma<P, 4, NMT, LF_TS512, LF_TS512, decltype(xhook)>(h, c4, L.w1b + 4 * LF_SS + fo0, L.w1b + 4 * LF_SS + fo1, 4, L.w1b + fo0, L.w1b + fo1, smem + LF_R1, 512, l15, g,
                                                             xhook);
    LF_T(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LF_T(3)
  }
#ifdef LF_TIMING2
  lf_mma<P, 4, NMT, LF_TS512, LF_TS512, LfNoHook, false, true>(h, c4, L.w1b + fo0, L.w1b + fo1, 4, L.w1b + fo0, L.w1b + fo1, smem + LF_R0, 512, l15, g);
#else
  lf_mma<P, 4, NMT, LF_TS512, LF_TS512>(h, c4, L.w1b + fo0, L.w1b + fo1, 4, L.w1b + fo0, L.w1b + fo1, smem + LF_R0, 512, l15, g);
#endif

  // LayerNorm scale / shift of this wave's 64 features: fetched now, used after the next barrier
  [[maybe_unused]] f32x4 gam[2][2], bet[2][2];
  if constexpr (!RELU) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        gam[q][u] = *reinterpret_cast<const f32x4*>(a.gamma + wf * 64 + q * 32 + g * 8 + u * 4);
        bet[q][u] = *reinterpret_cast<const f32x4*>(a.beta + wf * 64 + q * 32 + g * 8 + u * 4);
      }
  }
  // ---- LayerNorm(512): per-wave partial sums over its 64 features, exchanged through ST (indexed by feature block, so the
  // summation order does not depend on which wave owned it)
  if constexpr (!RELU) {
    float2* st = reinterpret_cast<float2*>(smem + LF_ST);
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += h[t][m][e];
          s2 = fmaf(h[t][m][e], h[t][m][e], s2);
        }
      s1 = rows_sum(s1);
      s2 = rows_sum(s2);
      if (g == 0) st[wf * 128 + m * 16 + l15] = make_float2(s1, s2);
    }
  }
  LF_T(4)
  __syncthreads();                                                // sums visible; x and msg tiles dead
  LF_T(5)
  // ffn.3's first weight fragments are fetched now, the fp32 residual rows half way: their latency hides under the GELU arithmetic
  const int co = cb * 64 + tp * 32 + g * 8;
  float* xr0 = a.x32 + (size_t)(row0 + l15) * 256 + co;
  float4 r0[NMT], r1[NMT];
  constexpr bool EARLY_W2 = NMT < 8;                              // (128-token passes are a few registers short: they fetch them after the GELU)
  if constexpr (EARLY_W2) lf_first<P, 2, LF_TS512>(c2, L.w2b + fo0, L.w2b + fo1);
  f32x4 b22[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) b22[u] = *reinterpret_cast<const f32x4*>(a.b2 + co + u * 4);
  __builtin_amdgcn_sched_barrier(0);
  {
    const float2* st = reinterpret_cast<const float2*>(smem + LF_ST);
    [[maybe_unused]] float nmr[NMT], rstd[NMT];                   // (h - mean) rstd = h rstd + nmr
#pragma unroll
    for (int m = 0; m < (RELU ? 0 : NMT); ++m) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float2 p = st[w * 128 + m * 16 + l15];
        s1 += p.x;
        s2 += p.y;
      }
      const float mean = s1 * (1.0f / 512.0f);
      const float var = fmaxf(s2 * (1.0f / 512.0f) - mean * mean, 0.f);
      rstd[m] = 1.0f / sqrtf(var + 1e-5f);
      nmr[m] = -mean * rstd[m];
    }
    // ---- GELU(LN(h)) -> h tile [16 NMT][512] 2-byte over R0 + R1
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int piece = wf * 8 + q * 4 + g;
      if constexpr (RELU) {
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(h[2 * q][m][e], 0.f);
            v[4 + e] = fmaxf(h[2 * q + 1][m][e], 0.f);
          }
          *reinterpret_cast<uint4*>(smem + (m * 16 + l15) * 1024 + ((piece ^ l15) << 4)) = pack8<P>(v);
        }
      } else {
      const f32x4 g0 = gam[q][0], g1 = gam[q][1], be0 = bet[q][0], be1 = bet[q][1];
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const f32x2 rs = {rstd[m], rstd[m]}, nm = {nmr[m], nmr[m]};
        float v[8];
#if LF_GELU_X4
        f32x2 y[4];                                                // the four feature pairs of this token tile: (tile 2q, e = 0), (2q, 2), (2q + 1, 0), (2q + 1, 2)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          y[e >> 1] = (f32x2{h[2 * q][m][e], h[2 * q][m][e + 1]} * rs + nm) * f32x2{g0[e], g0[e + 1]} + f32x2{be0[e], be0[e + 1]};
          y[2 + (e >> 1)] = (f32x2{h[2 * q + 1][m][e], h[2 * q + 1][m][e + 1]} * rs + nm) * f32x2{g1[e], g1[e + 1]} + f32x2{be1[e], be1[e + 1]};
        }
        lf_gelu2x4(y);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          v[e] = y[e >> 1].x; v[e + 1] = y[e >> 1].y;
          v[4 + e] = y[2 + (e >> 1)].x; v[5 + e] = y[2 + (e >> 1)].y;
        }
#else
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 y0 = (f32x2{h[2 * q][m][e], h[2 * q][m][e + 1]} * rs + nm) * f32x2{g0[e], g0[e + 1]} + f32x2{be0[e], be0[e + 1]};
          const f32x2 y1 = (f32x2{h[2 * q + 1][m][e], h[2 * q + 1][m][e + 1]} * rs + nm) * f32x2{g1[e], g1[e + 1]} + f32x2{be1[e], be1[e + 1]};
          const f32x2 o0 = lf_gelu2(y0), o1 = lf_gelu2(y1);
          v[e] = o0.x; v[e + 1] = o0.y;
          v[4 + e] = o1.x; v[5 + e] = o1.y;
        }
#endif
        *reinterpret_cast<uint4*>(smem + (m * 16 + l15) * 1024 + ((piece ^ l15) << 4)) = pack8<P>(v);
      }
      }
      if (q == 0) {           // half of h's registers are free now: the residual rows take them
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          r0[m] = *reinterpret_cast<const float4*>(xr0 + (size_t)m * 16 * 256);
          r1[m] = *reinterpret_cast<const float4*>(xr0 + (size_t)m * 16 * 256 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if constexpr (!EARLY_W2) lf_first<P, 2, LF_TS512>(c2, L.w2b + fo0, L.w2b + fo1);
  LF_T(6)
  __syncthreads();
  LF_T(7)

  // ---- x += W2 h + b2
  constexpr bool KEEP_X = FOLD != 0 && NMT < 8;                   // 128-token passes have no registers to spare: they re-read their x rows
  [[maybe_unused]] uint4 xpk[KEEP_X ? NMT : 1];
  {
    f32x4 acc[2][NMT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < NMT; ++m) acc[u][m] = b22[u];
    const char* nq = FOLD ? reinterpret_cast<const char*>(a.nqk_w) + lf_woff(256, cb, 2 * tp) : L.w2b;   // FOLD: first unit = q/k unit wf
    lf_mma<P, 2, NMT, LF_TS512, (FOLD ? LF_TS256 : LF_TS512)>(acc, c2, L.w2b + fo0, L.w2b + fo1, 8, nq + fo0, nq + fo1, smem, 1024, l15, g);
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      const size_t row = (size_t)(row0 + m * 16 + l15);
      float* xr = xr0 + (size_t)m * 16 * 256;
      float v[8] = {acc[0][m][0] + r0[m].x, acc[0][m][1] + r0[m].y, acc[0][m][2] + r0[m].z, acc[0][m][3] + r0[m].w,
                    acc[1][m][0] + r1[m].x, acc[1][m][1] + r1[m].y, acc[1][m][2] + r1[m].z, acc[1][m][3] + r1[m].w};
      *reinterpret_cast<float4*>(xr) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(xr + 4) = make_float4(v[4], v[5], v[6], v[7]);
      const uint4 xp = pack8<P>(v);
      if constexpr (KEEP_X) xpk[m] = xp;
      *reinterpret_cast<uint4*>(a.xb + row * 256 + co) = xp;
    }
  }
  LF_T(8)
  if constexpr (FOLD != 0) {
    __syncthreads();                                              // every wave is done with the h tile
    {
      const int piece = cb * 8 + tp * 4 + g;
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        uint4 xp;
        if constexpr (KEEP_X) xp = xpk[m];
        else xp = *reinterpret_cast<const uint4*>(a.xb + (size_t)(row0 + m * 16 + l15) * 256 + co);      // this lane's own store of a moment ago
        *reinterpret_cast<uint4*>(smem + LF_R0 + (m * 16 + l15) * 512 + ((piece ^ l15) << 4)) = xp;
      }
    }
    __syncthreads();                                              // the new x tile [16 NMT][256] is complete in R0
    LF_T(9)
    const int H = a.H, Np = a.Np;
    // q / k units: 32 features each, (cb', tp') = (id >> 1, id & 1); this wave takes id = wf (and wf + 8 of the 512-feature q | k)
    constexpr int NQ = FOLD == 2 ? 2 : 1;
    const char* vbase = reinterpret_cast<const char*>(a.nv_w) + lf_woff(256, cb, 2 * tp);
#pragma unroll
    for (int qu = 0; qu < NQ; ++qu) {
      const int cbq = cb + 4 * qu;
      const char* wq = reinterpret_cast<const char*>(a.nqk_w) + lf_woff(256, cbq, 2 * tp);
      const char* nx = (qu + 1 < NQ) ? reinterpret_cast<const char*>(a.nqk_w) + lf_woff(256, cbq + 4, 2 * tp) : vbase;
      const int cq = cbq * 64 + tp * 32 + g * 8;                  // this lane's 8 features of the projection
      f32x4 bq[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) bq[u] = *reinterpret_cast<const f32x4*>(a.nqk_b + cq + u * 4);
      f32x4 acc[2][NMT];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      lf_mma<P, 2, NMT, LF_TS256, LF_TS256>(acc, c2, wq + fo0, wq + fo1, 4, nx + fo0, nx + fo1, smem + LF_R0, 512, l15, g);
      const int sel = cq >> 8, hh = (cq & 255) >> 6, d = cq & 63;
      uint16_t* ob = sel ? a.k_out : a.q_out;
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const int row = row0 + m * 16 + l15;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[0][m][e] + bq[0][e];                          // bias after the K sum, as gemmr_body
          v[4 + e] = acc[1][m][e] + bq[1][e];
        }
        if constexpr (FOLD == 2) {
          const f32x4 cs = *reinterpret_cast<const f32x4*>(a.rot_cos + (size_t)row * 32 + tp * 16 + g * 4);
          const f32x4 sn = *reinterpret_cast<const f32x4*>(a.rot_sin + (size_t)row * 32 + tp * 16 + g * 4);
          rotate_pairs(v, cs, sn);                               // (single instructions, not packed math: see common.h)
        }
        // no row guard: the rows of a ragged last pass are garbage tokens of "sequences" S, S + 1 whose outputs land in the arena's slack
        // (alloc_matcher_arena), exactly like the surplus rows of the separate launches
        const int sq = row / Np, nn = row - sq * Np;
        *reinterpret_cast<uint4*>(ob + (((size_t)sq * H + hh) * Np + nn) * 64 + d) = pack8<P>(v);
      }
    }
    LF_T(10)
    // V unit id = wf, transposed: lane (l15 = feature column, g) holds 4 consecutive tokens per accumulator
    {
      float bt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) bt[u] = a.nv_b[cb * 64 + slab_row_to_feature((2 * tp + u) * 16 + l15)];
      f32x4 acc[2][NMT];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      lf_mma<P, 2, NMT, LF_TS256, LF_TS256, LfNoHook, true>(acc, c2, vbase + fo0, vbase + fo1, 4, vbase + fo0, vbase + fo1, smem + LF_R0, 512, l15, g);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int cv = cb * 64 + slab_row_to_feature((2 * tp + u) * 16 + l15);
        const int hh = cv >> 6, d = cv & 63;
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          const int rw = row0 + m * 16 + g * 4;
          const int sq = rw / Np, nn = rw - sq * Np;
          *reinterpret_cast<uint2*>(a.vt_out + (((size_t)sq * H + hh) * 64 + d) * Np + nn) =
                pack4<P>(acc[u][m][0] + bt[u], acc[u][m][1] + bt[u], acc[u][m][2] + bt[u], acc[u][m][3] + bt[u]);
        }
      }
    }
    LF_T(11)
  }
}

__device__ __forceinline__ void lf_lane_init(const LgBlockFArgs& a, LfLane& L) {
  const int tid = threadIdx.x;
  L.lane = tid & 63;
  L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  L.l15 = L.lane & 15;
  L.g = L.lane >> 4;
  if (LF_FRAG) {                                                  // fragment order: a lane's 16 bytes at lane * 16, the second 32-wide half 1 KB on
    L.fo0 = L.lane * 16;
    L.fo1 = L.lane * 16 + 1024;
  } else {
    const int sw = (L.l15 >> 1) & 7;                              // swz128 of this lane's slab row (same for every tile)
    L.fo0 = L.l15 * 128 + ((L.g ^ sw) << 4);
    L.fo1 = L.l15 * 128 + (((4 + L.g) ^ sw) << 4);
  }
  // Which feature block a wave owns rotates with the workgroup (among the workgroups of one XCD: blockIdx / 8), so that the
  // 32 CUs of an XCD do not all ask its L2 for the same weight lines at the same moment.  Numerics do not depend on it.
  L.wf = (L.wave + (blockIdx.x >> 3)) & 7;
  L.cb = L.wf >> 1;                                               // 256-feature GEMMs: wave = 32 features (one tile pair)
  L.tp = L.wf & 1;
  L.wob = reinterpret_cast<const char*>(a.wo) + lf_woff(256, L.cb, 2 * L.tp);
  L.w1b = reinterpret_cast<const char*>(a.w1) + lf_woff(512, L.wf, 0);
  L.w2b = reinterpret_cast<const char*>(a.w2) + lf_woff(512, L.cb, 2 * L.tp);
}

template <class P, int NMT, bool RELU, int FOLD, bool FOLDO>
__global__ __launch_bounds__(512, 1) void lg_blockf_kernel(LgBlockFArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LfLane L;
  lf_lane_init(a, L);
  const int row0 = blockIdx.x * (16 * NMT);
  lf_stage_rows(a.attn, row0, NMT, FOLDO ? LF_R1 : LF_R0, L.wave, L.lane);
  if constexpr (FOLDO) lf_stage_rows(a.xb, row0, NMT, LF_R0, L.wave, L.lane);
#ifdef LF_TIMING
  L.tb = 0;
#endif
  lf_pass<P, NMT, RELU, FOLD, FOLDO>(a, smem, L, row0);
#ifdef LF_TWICE
  __syncthreads();
  L.tb = 16;
  lf_stage_rows(a.attn, row0, NMT, FOLDO ? LF_R1 : LF_R0, L.wave, L.lane);
  if constexpr (FOLDO) lf_stage_rows(a.xb, row0, NMT, LF_R0, L.wave, L.lane);
  lf_pass<P, NMT, RELU, FOLD, FOLDO>(a, smem, L, row0);
#endif
}

// Two rounds of unequal passes (round 5).  A launch over 51200 tokens in 112-token passes is 458 workgroups on 256 CUs: 202 CUs run two passes = 14 token tiles,
// 54 run one, and the launch lasts as long as the 14.  At full load a pass costs in proportion to its tiles (128- against 112-token passes: 127 against 113 us per
// launch), so the first n7 workgroups — one per CU — take 7 tiles and the rest 6: every CU runs 13.  Which pass a token rides in changes nothing about its result
// (tests/test_gpu_lightglue.py: the tile sizes give the same bits).  The 6-tile passes may run up to 95 rows past M (arena slack, like the 111 of the uniform form).
template <class P, bool RELU, int FOLD>
__global__ __launch_bounds__(512, 1) void lg_blockf_mixed_kernel(LgBlockFArgs a, int n7) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LfLane L;
  lf_lane_init(a, L);
#ifdef LF_TIMING
  L.tb = 0;
#endif
  if ((int)blockIdx.x < n7) {
    const int row0 = blockIdx.x * 112;
    lf_stage_rows(a.attn, row0, 7, LF_R1, L.wave, L.lane);
    lf_stage_rows(a.xb, row0, 7, LF_R0, L.wave, L.lane);
    lf_pass<P, 7, RELU, FOLD, true>(a, smem, L, row0);
  } else {
    const int row0 = n7 * 112 + ((int)blockIdx.x - n7) * 96;
    lf_stage_rows(a.attn, row0, 6, LF_R1, L.wave, L.lane);
    lf_stage_rows(a.xb, row0, 6, LF_R0, L.wave, L.lane);
    lf_pass<P, 6, RELU, FOLD, true>(a, smem, L, row0);
  }
}

template <class P, bool RELU, int FOLD>
static bool launch_mixed(const LgBlockFArgs& a, hipStream_t st) {
  const int tiles = (a.M + 15) / 16, W = a.n_cu;
  if (!a.mixed || a.wo || W <= 0 || tiles <= 7 * W || tiles > 13 * W) return false;      // (folded out-projection only; one full round of 7-tile passes + at most one of 6)
  const int n7 = W, n6 = (tiles - 7 * W + 5) / 6;
  static PerDeviceOnce attr_once;
  auto kfn = lg_blockf_mixed_kernel<P, RELU, FOLD>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS);
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)(n7 + n6)), dim3(512), LF_LDS, st, a, n7);
  return true;
}

template <class P, int NMT, bool RELU, int FOLD, bool FOLDO>
static void launch_fo(const LgBlockFArgs& a, hipStream_t st) {
  static PerDeviceOnce attr_once;
  auto kfn = lg_blockf_kernel<P, NMT, RELU, FOLD, FOLDO>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS);
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)((a.M + 16 * NMT - 1) / (16 * NMT))), dim3(512), LF_LDS, st, a);
}

template <class P, int NMT, bool RELU, int FOLD = 0>
static void launch_f(const LgBlockFArgs& a, hipStream_t st) {
  if (a.wo) launch_fo<P, NMT, RELU, FOLD, false>(a, st);
  else launch_fo<P, NMT, RELU, FOLD, true>(a, st);          // the out-projection is inside w1 (airfe_tuning::fold_out_proj)
}

// Tokens per workgroup: one workgroup fits a CU (136 KiB of LDS), so a launch is ceil(M / tokens) workgroups in rounds of 256.
// 51200 tokens (64 pairs) in 128-token passes are 400 workgroups = a full round + a 56 % round; in 112-token passes 458 = two rounds
// that are each 1/8 shorter.  a.tokens_per_wg = 112 needs M rows + 111 of slack behind the last pass (the matcher arena has it).
template <class P, bool RELU, int FOLD>
static void launch_nmt(const LgBlockFArgs& a, hipStream_t st) {
  switch (a.tokens_per_wg) {
    case 32: launch_f<P, 2, RELU, FOLD>(a, st); break;      // small token counts (batch 1: 800 tokens = 25 workgroups instead of 7)
    case 64: launch_f<P, 4, RELU, FOLD>(a, st); break;
    case 112: if (!launch_mixed<P, RELU, FOLD>(a, st)) launch_f<P, 7, RELU, FOLD>(a, st); break;
    default: launch_f<P, 8, RELU, FOLD>(a, st); break;
  }
}

bool lg_blockf_frag_weights() { return LF_FRAG != 0; }

void launch_lg_blockf(int prec, const LgBlockFArgs& a, hipStream_t st) {
  if (a.relu) {
    if (prec == 1) launch_nmt<PF16, true, 0>(a, st); else launch_nmt<PBF16, true, 0>(a, st);
    return;
  }
  if (a.nqk_w) {                                       // with the next layer's projections folded in
    if (a.nqk_n == 512) {
      if (prec == 1) launch_nmt<PF16, false, 2>(a, st); else launch_nmt<PBF16, false, 2>(a, st);
    } else {
      if (prec == 1) launch_nmt<PF16, false, 1>(a, st); else launch_nmt<PBF16, false, 1>(a, st);
    }
    return;
  }
  if (prec == 1) launch_nmt<PF16, false, 0>(a, st); else launch_nmt<PBF16, false, 0>(a, st);
}

}  // namespace airfe
