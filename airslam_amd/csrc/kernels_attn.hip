// airfe — flash attention on the 32x32x16 MFMA (d_head = 64), the matcher's self / cross attention (LightGlue, SuperGlue).
//
// One (sequence, head, 128-query block) per workgroup, four waves of 32 queries, 64-key tiles double-buffered in LDS.
// Both products are computed transposed so that ONE LANE OWNS ONE QUERY:
//   S^T = K . Q^T   A = 32 key rows from LDS, B = the wave's 32 queries (registers)  -> lane (q = lane & 31, h = lane >> 5)
//                   holds 16 of a 32-key sub-tile's scores of query q, its partner lane q + 32 the other 16
//   O^T = V^T . P^T A = 32 rows of V^T from LDS, B = P packed to 2 bytes straight from the S^T accumulators
// so the online soft-max is per-lane arithmetic plus ONE cross-lane exchange per tile (v_permlane32_swap of the row maximum),
// and P never touches LDS.  Why 32x32x16 and not the 16x16x32 of round 1 (kernels_lg.hip, 15.6 % of the MFMA peak, an
// instruction issued in 94 % of all cycles): per 64-key tile and wave it needs 16 MFMAs instead of 32, 16 ds_read_b128
// instead of 8 + 16 ds_read_b64 (each K / V fragment now feeds 32 queries), 2 cross-lane operations instead of 8, and no
// per-16-lane-group bookkeeping.  Further: the soft-max scale is folded into the q / k projection weights and the running shift
// into the MFMA's C operand, so a probability costs ONE v_exp_f32 (see the comment at the kernel), the key tail runs one 32-key sub-tile when that is all that is left (400 keys = 6 tiles + ONE
// sub-tile: 416 key slots instead of 448), and waves / workgroups whose queries all lie beyond the sequence length do no
// arithmetic (400 queries in 128-query blocks left 112 of 512 slots computing on clamped rows).
//
// The MFMA rows of S^T take the keys of every 16-key group permuted (8b + 4h + i <-> 8h + 4b + i, just a different LDS row per
// lane): the contraction index of the P operand, as it falls out of the S^T accumulators, is then in natural key order and reads
// ONE 16-byte fragment of V^T instead of two 8-byte halves; K and V^T tiles are plain (swizzled) copies, staged by LDS-DMA.
#include "common.h"
#include "kernels.h"

#ifdef ATT_TIMING   // per-phase shader-clock timers of every ACTIVE wave (tools/att_timing.py; a measurement build, never the shipped library).  s_memtime ticks =
// shader cycles; each reading is consumed only after the tile's barrier, so that no s_waitcnt of its own sits between the phases it separates.
__device__ unsigned long long att_dbg[32];
extern "C" void airfe_dbg_att(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[32] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(att_dbg), z, sizeof(z)); }
  else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(att_dbg), 32 * sizeof(unsigned long long));
}
#define ATT_NOW(v) { __builtin_amdgcn_sched_barrier(0); v = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#define ATT_ADD(i, d) { if (lane == 0) atomicAdd(&att_dbg[i], (unsigned long long)(d)); }      // (only behind the tile loop: an atomic per tile is vector-memory work the loop's own vmcnt waits would see)
#else
#define ATT_NOW(v)
#define ATT_ADD(i, d)
#endif

namespace airfe {


template <class P> struct Mfma32;
template <> struct Mfma32<PBF16> {
  static __device__ __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32<PF16> {
  static __device__ __forceinline__ f32x16 run(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// value of the partner lane (lane ^ 32) combined with the own one
__device__ __forceinline__ float pair_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
}

// 16 bytes per lane from sbase + byte offset voff straight into LDS at lds_off + lane * 16 (M0 carries the LDS address)
__device__ __forceinline__ void att_glds16(const void* sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_off)
               : "memory");
}

#ifndef ATT_RETRY_LOOP
#define ATT_RETRY_LOOP 0
#endif
#ifndef ATT_SUBTILE        // 1: a tile runs as a pipeline of its two 32-key sub-tiles inside the wave (round 6: the same bits, measured no faster — 41.6 / 41.3 against 41.2 / 40.6 us,
#define ATT_SUBTILE 0      // profiles/r06_attention_deletions.txt; kept as the tested alternative); 0 (default): exponentials of the whole tile, then its PV (rounds 3-6)
#endif
#ifdef ATT_NOEXP           // diagnostic build (garbage results; kernel times only): the soft-max without its exponentials (and without the overflow re-run they would trigger)
#define ATT_EXP2(x) (x)
#else
#define ATT_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
constexpr float ATT_PSUM_MAX = 16384.0f;   // a lane's partial row sum above this sends the tile through the re-centring path

// The soft-max scale is NOT applied here: sqrt(scale * log2 e) is folded into the packed q and k projection weights
// (airfe_host.h: ATT_QK_FOLD), so the accumulators already hold s = log2(e) * q.k / sqrt(d).  The running shift m of the online
// soft-max enters through the C operand of the first MFMA of every score chain (a 16-register broadcast of -m that changes only
// when the shift does), so a tile's probabilities are ONE v_exp_f32 per score: p = 2^(s - m), no fma, no per-tile row maximum.
// m is exact after the first tile (explicit maximum there); afterwards it is stale by design — soft-max is shift-invariant, the
// shift only has to keep p inside the 2-byte range.  Every p is bounded by its lane's partial row sum, which is needed anyway:
// if any lane's sum exceeds 2^14 the tile is recomputed and re-centred (maximum -> 0, accumulators rescaled).  The true row
// maximum is never below m, so the dominant probabilities are always >= 1: nothing that matters underflows.
template <class P, int OCC>
__global__ __launch_bounds__(256, OCC) void attention32_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K,
                                                          const uint16_t* __restrict__ Vt, uint16_t* __restrict__ O,
                                                          const int* __restrict__ lens, int H, int Np, int cross, int nqb) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
#ifdef ATT_TIMING
  unsigned long long t_start, t_a, t_b, t_c, t_d, t_e, t_f, t_g, t_pro;
  unsigned long long s_b = 0, s_c = 0, s_d = 0, s_e = 0, s_f = 0, s_g = 0, s_tiles = 0, s_retry = 0;
  ATT_NOW(t_start)
#endif
#ifdef ATT_EMPTY           // diagnostic build (garbage results; kernel times only): what 2048 workgroups cost that do nothing
  if (Np > 0) return;
#endif
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: it forms the LDS address M0 carries for the DMA
  // XCD-aware workgroup -> (sequence, head, query block) map: the nqb query blocks of one (sequence, head) re-read its K and V,
  // so they take consecutive slots of ONE XCD (workgroup L runs on XCD L % 8) and share that XCD's L2.
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
  const int grp = (li / nqb) * 8 + xcd, qb = li - (li / nqb) * nqb;
  const int s = grp / H, h = grp - s * H;
  const int len_q = lens[s];
  if (qb * 128 >= len_q) return;                       // the whole block lies in the padding rows (wave-uniform, before any barrier)
  const int q0 = qb * 128 + wave * 32;
  const bool active = q0 < len_q;                      // wave-uniform: inactive waves only help staging
  const int skv = cross ? (s ^ 1) : s;
  const int len_kv = lens[skv];
  const uint16_t* Qh = Q + ((size_t)s * H + h) * Np * 64;
  const uint16_t* Kh = K + ((size_t)skv * H + h) * Np * 64;
  const uint16_t* Vh = Vt + ((size_t)skv * H + h) * 64 * Np;

  const int nkv = (len_kv + 63) >> 6;
  float m_i = 0.f, l_i = 0.f;
  f32x16 cinit = f32x16{};                               // broadcast of -m_i: the C operand that opens every score chain
  f32x16 o[2] = {f32x16{}, f32x16{}};

  // staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows of a tile per instruction, no staging registers, no
  // ds_write): a K / V^T tile is 8 + 8 instructions, two + two per wave.  The LDS image is chunk-swizzled (swz128), so lane
  // (row 8 i + lane / 8, physical chunk pc = lane & 7) fetches logical chunk pc ^ swz128(row).  Tile t + 1 is issued at the top of
  // iteration t into the buffer whose readers passed the barrier that ended iteration t - 1.  (A three-buffer ring with tile t + 2 in flight
  // and `vmcnt(4)` waits measured 0.77 vs 0.75 ms of attention per step: the tile DMA is not what the waves wait for; not kept.)
  unsigned koff[2], voff[2];                            // element offsets of this lane's four 16-byte pieces inside tile 0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3), pc = lane & 7;
    koff[i] = (unsigned)(row * 64 + ((pc ^ swz128(row)) << 3));
    voff[i] = (unsigned)(row * Np + ((pc ^ swz128(row)) << 3));
  }
  auto dma_tile = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      att_glds16(Kh, (koff[i] + (unsigned)kt * 4096u) * 2u, (unsigned)(buf * 16384 + (wave * 2 + i) * 1024));
      att_glds16(Vh, (voff[i] + (unsigned)kt * 64u) * 2u, (unsigned)(buf * 16384 + 8192 + (wave * 2 + i) * 1024));
    }
  };
#ifndef ATT_Q_FIRST        // (-DATT_Q_FIRST: rounds 3-5's order — Q fragments, wait, first tile, wait: two dependent memory round trips per workgroup)
  if (nkv > 0) dma_tile(0, 0);                           // round 6: the first K / V tile is requested BEFORE the Q fragments, one wait covers both
#endif
  typename P::vec8 qf[4];
  {
    const int row = min(q0 + l31, Np - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 u = *reinterpret_cast<const uint4*>(Qh + (size_t)row * 64 + ks * 16 + hh * 8);
      qf[ks] = __builtin_bit_cast(typename P::vec8, u);
    }
  }
  // The Q fragments must have landed before the tile loop's FIRST prefetch is issued: vmcnt retires in order, so the compiler's own wait for qf
  // (a counted vmcnt at their first use, inside the tile loop) would otherwise also wait for the NEWEST tile prefetch in every
  // iteration — measured: 123 us per launch instead of 47.  The builtin (not an asm string) lets hipcc's wait-count pass see it.
  __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), expcnt / lgkmcnt untouched: Q fragments AND (the DMA being older) tile 0
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));      // (pins the loads above this point: hipcc otherwise sinks them to their first use in the tile loop)
#ifdef ATT_Q_FIRST
  if (nkv > 0) dma_tile(0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef ATT_PROLOGUE_ONLY   // diagnostic build (garbage results; kernel times only): the workgroups leave behind their prologue — Q fragments, first K / V tile, barrier
  if (Np > 0) { if (lane == 0 && __builtin_bit_cast(uint4, qf[0]).x == 0x12345678u) O[0] = 1; return; }      // (keeps the Q loads alive)
#endif

  // fragment addresses inside a tile (a row of a 32-row sub-tile, 16-byte chunk 2 ks + hh, swizzled; rows r and r + 32 swizzle alike).
  // K rows are read PERMUTED — MFMA row i takes key (i with bits 2 and 3 swapped) — so that accumulator r of lane (q, hh) is key
  // 16 (r >> 3) + 8 hh + (r & 7): eight CONSECUTIVE keys per 16-key contraction step, i.e. P packs straight into the B operand
  // of V^T . P^T against V^T fragments in natural key order (one 16-byte read each, V^T staged as a plain copy).
  const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int ksw = swz128(krow), fsw = swz128(l31);

#ifdef ATT_TIMING
  ATT_NOW(t_g)
  t_pro = t_g;
#endif
  for (int kt = 0; kt < nkv; ++kt) {
    ATT_NOW(t_a)
#ifdef ATT_NODMA       // diagnostic build (tools/build_timing_variants.sh): no K / V staging behind tile 0 (garbage results) — the ceiling of any fix of the staging traffic
    if (false)
#endif
    if (kt + 1 < nkv) dma_tile(kt + 1, (kt + 1) & 1);
#ifdef ATT_NODMA
    const char* kb = smem;                               // (tile 0's data every time: valid scores, no staging)
#else
    const char* kb = smem + (kt & 1) * 16384;
#endif
    const char* vb = kb + 8192;
    if (active) {
      const int left = len_kv - kt * 64;                 // keys left from this tile on (wave-uniform)
      const bool two = left > 32;                        // the second 32-key sub-tile holds keys
      f32x16 st0, st1;                                   // st1 is touched only under `two`
      auto scores = [&]() {                              // S^T - m for this tile: the shift rides in as the chains' C operand
        // all fragment reads of the tile first (hipcc otherwise sinks every ds_read to its MFMA: read - wait - MFMA, four times over),
        // then the two independent chains interleaved so that the matrix pipe always has an MFMA that does not wait for its C
        typename P::vec8 kf0[4], kf1[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf0[ks] = lds_frag<P>(kb, krow * 128 + (((ks * 2 + hh) ^ ksw) << 4));
        if (two) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) kf1[ks] = lds_frag<P>(kb, (32 + krow) * 128 + (((ks * 2 + hh) ^ ksw) << 4));
          __builtin_amdgcn_sched_barrier(0);
          st0 = Mfma32<P>::run(kf0[0], qf[0], cinit);
          st1 = Mfma32<P>::run(kf1[0], qf[0], cinit);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks) {
            st0 = Mfma32<P>::run(kf0[ks], qf[ks], st0);
            st1 = Mfma32<P>::run(kf1[ks], qf[ks], st1);
          }
        } else {
          __builtin_amdgcn_sched_barrier(0);
          st0 = Mfma32<P>::run(kf0[0], qf[0], cinit);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks) st0 = Mfma32<P>::run(kf0[ks], qf[ks], st0);
        }
        if (left < 64) {                                 // only the last tile can hold keys beyond the sequence ...
          // ... and only ONE of its sub-tiles straddles the end (the other is whole, or not run at all).  Accumulator r holds key
          // 16 (r >> 3) + 8 hh + (r & 7) of its sub-tile: compare the compile-time part with ONE per-lane bound.
          int lo = two ? left - 32 : left;               // opaque to the optimiser: otherwise the compares are hoisted out of this
          asm volatile("" : "+s"(lo));                  // branch and issued for every tile
          const int la = lo - 8 * hh;
          // (value selects on BOTH accumulators: an if / else that writes st1 or st0 makes hipcc select between the two register
          //  arrays by pointer and move them to scratch memory — 192 bytes per lane, 8x slower kernel)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool hit = 16 * (r >> 3) + (r & 7) >= la;
            st0[r] = (hit && !two) ? -INFINITY : st0[r];
            st1[r] = (hit && two) ? -INFINITY : st1[r];
          }
        }
      };
#if !ATT_SUBTILE
      float ps0, ps1;
      auto exps = [&]() {                                // p = 2^(s - m) in place + the lane's partial row sum
        // (the two partial sums as ONE float pair added in the accumulators' own order: written as two scalars, hipcc's SLP pass paired them the other way
        //  round, put every second exponential into the "wrong" register of its pair and moved sixteen values back per tile)
        f32x2 ps = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          st0[r] = ATT_EXP2(st0[r]);
          st0[r + 1] = ATT_EXP2(st0[r + 1]);
          ps += f32x2{st0[r], st0[r + 1]};
        }
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            st1[r] = ATT_EXP2(st1[r]);
            st1[r + 1] = ATT_EXP2(st1[r + 1]);
            ps += f32x2{st1[r], st1[r + 1]};
          }
        }
        ps0 = ps.x; ps1 = ps.y;
      };
#endif
      auto recentre_tile = [&]() {                         // the shift from the tile's explicit row maximum: scores -> scores - max, accumulators rescaled
        float mx = max3f(st0[0], st0[1], st0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3f(mx, st0[r], st0[r + 1]);
        mx = fmaxf(mx, st0[15]);
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) mx = max3f(mx, st1[r], st1[r + 1]);
        }
        mx = pair_max(mx);                               // finite: every tile that is run holds at least one real key
        if (kt > 0) {                                    // (nothing accumulated yet on the first tile)
          const float alpha = __builtin_amdgcn_exp2f(-mx);
          l_i *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        m_i += mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cinit[r] = -m_i;
          st0[r] -= mx;
        }
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st1[r] -= mx;
        }
      };
#if !ATT_SUBTILE
#if ATT_RETRY_LOOP
      bool recentre = (kt == 0);                         // the first tile fixes the shift from its explicit row maximum
      for (;;) {                                         // (one copy of the tile's code: a tile that overflows simply goes round again)
        scores();
        if (recentre) recentre_tile();
        exps();
        if (recentre || !__any(!(ps0 + ps1 <= ATT_PSUM_MAX))) break;      // (the negated compare also catches inf / NaN sums)
        recentre = true;
      }
#else
      // Straight line (round 5): as a loop ("a tile that overflows simply goes round again") the score registers were loop-carried, and hipcc moved the second
      // sub-tile's sixteen probabilities from where v_exp_f32 put them back into the loop's registers — 14 v_mov_b32 per tile on a kernel that is bound by its
      // issue slots.  The rare second round is now a second copy of the code.
      scores();
#ifdef ATT_TIMING
      ATT_NOW(t_b)                                       // K fragments read, QK^T chains issued
      { float w_; asm volatile("v_mov_b32 %0, %1" : "=v"(w_) : "v"(two ? st1[15] : st0[15])); asm volatile("" :: "v"(w_)); }
      ATT_NOW(t_c)                                       // ... and complete (the move waits for the last accumulator)
#endif
      if (kt == 0) recentre_tile();                      // the first tile fixes the shift from its explicit row maximum
      exps();
#ifdef ATT_NOEXP
      if (false) {
#else
      if (kt != 0 && __any(!(ps0 + ps1 <= ATT_PSUM_MAX))) {               // (the negated compare also catches inf / NaN sums)
#endif
        scores();
        recentre_tile();
        exps();
#ifdef ATT_TIMING
        s_retry += 1;
#endif
      }
#endif
      ATT_NOW(t_d)                                       // probabilities + partial row sums
      l_i += ps0 + ps1;
      typename P::vec8 pf[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = st0[8 * j + e];
        pf[j] = __builtin_bit_cast(typename P::vec8, pack8<P>(pv));
      }
      if (two) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float pv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = st1[8 * j + e];
          pf[2 + j] = __builtin_bit_cast(typename P::vec8, pack8<P>(pv));
        }
      }
      // ---- O^T += V^T . P^T : 16-key contraction steps j4, 32-row d tiles dt (two independent chains, alternated)
      {
        typename P::vec8 vf[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int j4 = 0; j4 < 2; ++j4) vf[dt][j4] = lds_frag<P>(vb, (dt * 32 + l31) * 128 + (((j4 * 2 + hh) ^ fsw) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) {
          o[0] = Mfma32<P>::run(vf[0][j4], pf[j4], o[0]);
          o[1] = Mfma32<P>::run(vf[1][j4], pf[j4], o[1]);
        }
        if (two) {                                       // (same four registers again: eight fragments in flight spilled)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) vf[dt][j4] = lds_frag<P>(vb, (dt * 32 + l31) * 128 + ((((j4 + 2) * 2 + hh) ^ fsw) << 4));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j4 = 0; j4 < 2; ++j4) {
            o[0] = Mfma32<P>::run(vf[0][j4], pf[2 + j4], o[0]);
            o[1] = Mfma32<P>::run(vf[1][j4], pf[2 + j4], o[1]);
          }
        }
      }
#else
      // ---- round 6: the tile as a pipeline of its two 32-key SUB-TILES inside the wave (tools/microbench/valu_rates.hip: this kernel is bound by the latency of each
      // wave's serial chain scores -> exponentials -> PV, not by issue slots).  Order: all V^T fragments requested right behind the score chains (they take the K
      // fragments' registers; their LDS latency passes under the exponentials), exponentials + pack of sub-tile 0, PV of sub-tile 0 ISSUED, then the exponentials of
      // sub-tile 1 while those four MFMAs run, pack, PV of sub-tile 1.  Same operations on the same values in the same order per accumulator as the tile-at-once form
      // (-DATT_SUBTILE=0): the same bits, except in a tile that re-centres — the unit of the online soft-max is now the sub-tile, and a re-centring never LOWERS the shift
      // (rows of the wave that did not overflow keep theirs: alpha = 1).
      f32x2 ps = {0.f, 0.f};                             // the lane's partial row sums of this tile (one float pair, added in the accumulators' own order)
      auto exps_sub = [&](f32x16& st) {                  // p = 2^(s - m) in place, sums into ps
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          st[r] = __builtin_amdgcn_exp2f(st[r]);
          st[r + 1] = __builtin_amdgcn_exp2f(st[r + 1]);
          ps += f32x2{st[r], st[r + 1]};
        }
      };
      auto rowmax = [&](const f32x16& st) {              // over the lane's 16 keys of the sub-tile and its partner lane's 16
        float mx = max3f(st[0], st[1], st[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3f(mx, st[r], st[r + 1]);
        return pair_max(fmaxf(mx, st[15]));
      };
      auto rescale = [&](float mx) {                     // the shift moves up by mx >= 0: everything accumulated so far shrinks by 2^-mx
        const float alpha = __builtin_amdgcn_exp2f(-mx);
        l_i *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        m_i += mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = -m_i;
        return alpha;
      };
      auto pack_sub = [&](const f32x16& st, typename P::vec8 (&pf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float pv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = st[8 * j + e];
          pf[j] = __builtin_bit_cast(typename P::vec8, pack8<P>(pv));
        }
      };
      scores();
#ifdef ATT_TIMING
      ATT_NOW(t_b)                                       // K fragments read, QK^T chains issued
      ATT_NOW(t_c)
#endif
      typename P::vec8 vf[2][2];                         // V^T fragments of a sub-tile: [16-key step j4][32-row d tile dt]
#pragma unroll
      for (int j4 = 0; j4 < 2; ++j4)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[j4][dt] = lds_frag<P>(vb, (dt * 32 + l31) * 128 + (((j4 * 2 + hh) ^ fsw) << 4));
      __builtin_amdgcn_sched_barrier(0);
      if (kt == 0) recentre_tile();                      // the first tile fixes the shift from its explicit row maximum (both sub-tiles; nothing accumulated yet)
      typename P::vec8 pf0[2], pf1[2];
      // ---- sub-tile 0
      exps_sub(st0);
      if (kt != 0 && __any(!(ps.x + ps.y <= ATT_PSUM_MAX))) {     // (the negated compare also catches inf / NaN sums)
        scores();                                        // st0 was exponentiated in place: both chains again (a rare path)
        const float mx = fmaxf(rowmax(st0), 0.f);
        rescale(mx);
#pragma unroll
        for (int r = 0; r < 16; ++r) st0[r] -= mx;
        if (two) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st1[r] -= mx;
        }
        ps = f32x2{0.f, 0.f};
        exps_sub(st0);
#ifdef ATT_TIMING
        s_retry += 1;
#endif
      }
      pack_sub(st0, pf0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j4 = 0; j4 < 2; ++j4) {                   // O^T += V^T . P^T over the sub-tile's two 16-key steps, the two d tiles alternated
        o[0] = Mfma32<P>::run(vf[j4][0], pf0[j4], o[0]);
        o[1] = Mfma32<P>::run(vf[j4][1], pf0[j4], o[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- sub-tile 1: its V^T fragments are requested and its exponentials run while the four MFMAs above do
      if (two) {
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vf[j4][dt] = lds_frag<P>(vb, (dt * 32 + l31) * 128 + ((((j4 + 2) * 2 + hh) ^ fsw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 ps_sub0 = ps;
        exps_sub(st1);
        if (__any(!(ps.x + ps.y <= ATT_PSUM_MAX))) {     // (the first tile cannot get here: its shift is the exact maximum, every p <= 1)
          scores();                                      // (st0 comes back too; it is dead: sub-tile 0 is inside o and ps_sub0 already)
          const float mx = fmaxf(rowmax(st1), 0.f);
          const float alpha = rescale(mx);               // o holds sub-tile 0's PV under the old shift: it shrinks with the rest
#pragma unroll
          for (int r = 0; r < 16; ++r) st1[r] -= mx;
          ps = ps_sub0 * alpha;
          exps_sub(st1);
#ifdef ATT_TIMING
          s_retry += 1;
#endif
        }
        pack_sub(st1, pf1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) {
          o[0] = Mfma32<P>::run(vf[j4][0], pf1[j4], o[0]);
          o[1] = Mfma32<P>::run(vf[j4][1], pf1[j4], o[1]);
        }
      }
      ATT_NOW(t_d)
      l_i += ps.x + ps.y;
#endif
    }
    ATT_NOW(t_e)                                         // P packed, V fragments read, PV chains issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of tile kt + 1 have landed ...
    ATT_NOW(t_f)
#ifndef ATT_NOBAR      // diagnostic build (with ATT_NODMA; garbage results, kernel times only): the tile loop without its barrier — every wave at its own pace
    __syncthreads();                                     // ... and so have everyone else's; buffer kt & 1 is free again
#endif
#ifdef ATT_TIMING
    ATT_NOW(t_g)
    if (active) { s_b += t_b - t_a; s_c += t_c - t_b; s_d += t_d - t_c; s_e += t_e - t_d; s_f += t_f - t_e; s_g += t_g - t_f; s_tiles += 1; }
#endif
  }

  if (!active) return;
  const float l = pair_sum(l_i);
  const float inv = (l > 0.f) ? 1.0f / l : 0.f;
  const int q = q0 + l31;
  // lane (q, hh) holds d = 32 dt + 8 g + 4 hh + 0..3 in o[dt][4g .. 4g+3]; one permlane32 swap per word pair hands lane hh = 0 the
  // whole 8-feature run of the even g and lane hh = 1 that of the odd g: 16-byte stores instead of 8-byte ones at a 16-byte stride
  uint16_t* orow = O + ((size_t)s * Np + min(q, Np - 1)) * (H * 64) + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int ge = 2 * gp, go = 2 * gp + 1;
      const uint32_t x0 = P::pack2(o[dt][4 * ge] * inv, o[dt][4 * ge + 1] * inv), x1 = P::pack2(o[dt][4 * ge + 2] * inv, o[dt][4 * ge + 3] * inv);
      const uint32_t y0 = P::pack2(o[dt][4 * go] * inv, o[dt][4 * go + 1] * inv), y1 = P::pack2(o[dt][4 * go + 2] * inv, o[dt][4 * go + 3] * inv);
      const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
      // hh = 0: {own x0, own x1, partner x0, partner x1} = d 8 ge + 0..7;  hh = 1: {partner y0, partner y1, own y0, own y1} = d 8 go + 0..7
      const uint4 v = make_uint4((uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]);
      if (q < Np) *reinterpret_cast<uint4*>(orow + dt * 32 + (hh ? go : ge) * 8) = v;
    }
  }
#ifdef ATT_TIMING
  ATT_NOW(t_a)
  ATT_ADD(0, t_pro - t_start) ATT_ADD(15, 1)
  ATT_ADD(1, s_b) ATT_ADD(2, s_c) ATT_ADD(3, s_d) ATT_ADD(4, s_e) ATT_ADD(5, s_f) ATT_ADD(6, s_g) ATT_ADD(13, s_tiles) ATT_ADD(14, s_retry)
  ATT_ADD(7, t_a - t_g)
  ATT_ADD(8, t_a - t_start)
#endif
}

void launch_attention32(int prec, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, const int* lens,
                        int S, int H, int Np, int cross, hipStream_t st) {
  const int nqb = (Np + 127) / 128;
  dim3 grid((unsigned)(nqb * H * S));                 // 1-D, decoded XCD-aware by the kernel; S * H % 8 == 0
  if (prec == 1) hipLaunchKernelGGL((attention32_kernel<PF16, 3>), grid, dim3(256), 0, st, Q, K, Vt, O, lens, H, Np, cross, nqb);
  else hipLaunchKernelGGL((attention32_kernel<PBF16, 3>), grid, dim3(256), 0, st, Q, K, Vt, O, lens, H, Np, cross, nqb);
}

}  // namespace airfe
