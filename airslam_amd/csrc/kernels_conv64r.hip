// airfe — persistent 3x3 convolution for the 64-input-channel layers (conv1b, conv2a, conv2b, conv3a) with the filter
// bank RESIDENT IN REGISTERS, 64 output channels per pass.
//
// Why: in the LDS-resident form (weights + pixels both fetched from LDS, wave tile 64 pixels x 64 couts) a wave issues
// 8 ds_read_b128 per 16 MFMAs; four waves then need 576 KiB of LDS reads per 16x16 tile = 4608 cycles at 128 B/clk, exactly
// the 4608 MFMA cycles of the tile: the LDS pipe and the matrix pipe were co-limiting and the kernel sat at ~45 % of the
// MFMA peak.  Here the four waves of a workgroup are a 2 x 2 grid (pixel rows 0-7 / 8-15  x  couts 0-31 / 32-63):
//   * a wave's share of the filters is 9 taps x 64 cin x 32 cout = 36 A-fragments = 144 VGPRs, loaded ONCE per workgroup;
//   * pixel fragments are fetched once per (column shift, channel half) and reused by all three filter rows:
//     10 ds_read_b128 per 48 MFMAs -> 240 KiB of LDS reads per tile (2.4x less), the kernel becomes MFMA-bound;
//   * LDS now only holds the double-buffered halo tile (2 x 40.5 KiB), filled one tile ahead by LDS-DMA.
#include "common.h"
#ifdef C64R_TIMING
__device__ long long c64_dbg[256 * 8 * 6];
#define C64R_T(x) { const long long now_ = wall_clock64(); x += now_ - tp_; tp_ = now_; }
#else
#define C64R_T(x)
#endif
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr64r;

__device__ __forceinline__ void c64r_glds4(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

__device__ __forceinline__ void c64r_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int C64R_TILE_STRIDE = 21 * 16 * 128;    // 43008: buffer pitch, padded to 21 whole groups of 16 pixels (fused conv1a)
constexpr int C64R_PIECES = 18 * 18 * 8;           // 2592 sixteen-byte pieces
#ifndef C64R_SCHED
#define C64R_SCHED 1
#endif
constexpr bool SCHED = C64R_SCHED;
#ifndef C64R_PIN
#define C64R_PIN 1
#endif
#ifndef C64R_RW
#define C64R_RW 4
#endif
#ifndef C64R_SPLIT_PROD
#define C64R_SPLIT_PROD 1
#endif
#ifndef C64R_EARLY_EPI
#define C64R_EARLY_EPI 1
#endif

constexpr int C64R_PATCH = 20 * 20;                // fused conv1a: fp32 image patch per tile (halo 2)
constexpr int C64R_CONST_OFF = 2 * C64R_TILE_STRIDE + 3 * C64R_PATCH * 4;   // conv1a A fragments (4 KiB) + bias (256 B)

// RW = pixel rows per wave: 8 -> 4 waves (one per SIMD, 512 registers each), 4 -> 8 waves (two per SIMD)
template <class P, bool POOL, bool FUSE1A, int RW>
__global__ __launch_bounds__(2048 / RW, 1) void conv64r_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool CONST_IN_LDS = RW == 4;               // two waves per SIMD: 256 registers each, small constants live in LDS
  constexpr int NT = 2048 / RW;                        // threads per workgroup
  constexpr int NG = (21 * RW + 31) / 32;              // fused conv1a: 16-pixel groups per wave (21 groups over 32/RW waves)
  constexpr int NPC = (C64R_PIECES + NT - 1) / NT;     // LDS-DMA pieces per thread
  const int ph = wave >> 1, ch = wave & 1;             // pixel-row block, cout half of this wave
  const int H = a.H, W = a.W, COUT = a.COUT;
  const size_t in_row = (size_t)(W + 2) * 128;
  const size_t in_img = (size_t)(H + 2) * in_row;
  const int per_img = tiles_x * tiles_y;
  const bool p2 = (tiles_x & (tiles_x - 1)) == 0 && (per_img & (per_img - 1)) == 0;
  const int sh_x = p2 ? __builtin_ctz(tiles_x) : -1, sh_img = p2 ? __builtin_ctz(per_img) : -1;

  // ---- filters: slab rows (2*ch + tt)*16 + l15 of every tap, both channel halves, as MFMA A fragments
  typename P::vec8 wreg[9][2][2];
  {
    const char* wp = reinterpret_cast<const char*>(a.Wp) + (size_t)cb0 * 9 * SLAB_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int rr = (2 * ch + tt) * 16 + l15;
          const uint4 u = *reinterpret_cast<const uint4*>(wp + tap * SLAB_BYTES + rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
          wreg[tap][ks][tt] = __builtin_bit_cast(typename P::vec8, u);
        }
  }
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = a.bias[cb0 * 64 + ch * 32 + g * 8 + e];

  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * COUT;

  // ---- loop-invariant offsets: LDS-DMA piece q = j*256 + tid lands at LDS byte q*16; it is channel chunk
  // (q&7) ^ swz(pixel) of pixel q>>3 (the swizzle is applied on the SOURCE address)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr64r)smem);
  int goff[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int q = min(j * NT + tid, C64R_PIECES - 1);
    const int p = q >> 3;
    const int pr = p / 18, pc = p - pr * 18;
    const int c = (q & 7) ^ ((pc >> 1) & 7);           // chunk swizzle by pixel COLUMN (see cbase below)
    goff[j] = pr * (int)in_row + pc * 128 + c * 16;
  }
  const bool last_piece = (NPC - 1) * NT + tid < C64R_PIECES;
  auto stage = [&](int tile, int buf) {
    int b, ty, tx;
    tile_decode(tile, per_img, tiles_x, sh_img, sh_x, b, ty, tx);
    const char* xin = reinterpret_cast<const char*>(a.X) + (size_t)b * in_img + (size_t)ty * 16 * in_row + (size_t)tx * 16 * 128;
    const unsigned dst = lds_base + buf * C64R_TILE_STRIDE + wave * 1024;
#pragma unroll
    for (int j = 0; j < NPC - 1; ++j) c64r_glds16(xin + goff[j], dst + j * (NT * 16));
    if (last_piece) c64r_glds16(xin + goff[NPC - 1], dst + (NPC - 1) * (NT * 16));
  };
  // pixel fragments: tile rows ph*RW + {0..RW+1}, column shifts {0,1,2}; channel half ks = 1 is XOR 64.  The 16-byte chunks
  // are swizzled with the pixel COLUMN only (a fragment's 16 consecutive pixels of a row still hit 16 different bank groups;
  // rows are 2304 bytes = whole bank rows apart), so the address is (per-lane column term) + (compile-time row offset):
  // 3 address registers instead of 3 * (RW + 2).
  int cbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) cbase[dx] = (ph * RW * 18 + l15 + dx) * 128 + ((g ^ (((l15 + dx) >> 1) & 7)) << 4);

  // ---- fused conv1a (Cin = 1, FUSE1A): the 18x18x64 input tile of conv1b is COMPUTED here from a 20x20 patch of the
  // fp32 image instead of being read back from HBM: conv1a's 32 MiB/image of output never exists (conv1b alone is at
  // the HBM ridge: 458 flop/byte).  conv1a runs on the matrix pipe too, as one 32-wide fp16 k-step: lane group g = 0..2
  // carries filter row dy = g (3 taps in k slots g*8 + {0,1,2}), g = 3 carries a constant 1 against the bias, so that
  // the epilogue is just ReLU + pack; halo pixels outside the image get an all-zero B column -> exact 0.
  [[maybe_unused]] f16x8 w1[4];
  [[maybe_unused]] const unsigned pbase = lds_base + 2 * C64R_TILE_STRIDE;
  [[maybe_unused]] const float* pbuf = reinterpret_cast<const float*>(smem + 2 * C64R_TILE_STRIDE);
  if constexpr (FUSE1A) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int co = slab_row_to_feature(t * 16 + l15);
      f16x8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (_Float16)0.f;
      if (g < 3) {
#pragma unroll
        for (int e = 0; e < 3; ++e) w[e] = (_Float16)a.w1a[co * 9 + g * 3 + e];
      } else {
        w[0] = (_Float16)a.b1a[co];
      }
      w1[t] = w;
      if constexpr (CONST_IN_LDS) { if (wave == 0) *reinterpret_cast<f16x8*>(smem + C64R_CONST_OFF + (t * 64 + lane) * 16) = w; }
    }
  }
  if constexpr (CONST_IN_LDS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) reinterpret_cast<float*>(smem + C64R_CONST_OFF + 4096)[(ch * 4 + g) * 8 + e] = bias[e];
  }
  // 20x20 fp32 patch of tile t -> pbuf[buf] by 4-byte LDS-DMA.  The image buffer has a 1-pixel zero border; rows/cols
  // beyond even that are clamped: they only feed halo pixels outside the image, which are forced to 0 anyway.
  auto stage_patch = [&](int t, int buf) {
    int b, ty, tx;
    tile_decode(t, per_img, tiles_x, sh_img, sh_x, b, ty, tx);
    const float* img = a.img + (size_t)b * (H + 2) * (W + 2);
#pragma unroll
    for (int j = 0; j < (C64R_PATCH + NT - 1) / NT; ++j) {
      const int q = j * NT + tid;
      if (q < C64R_PATCH) {
        const int r = q / 20, cc = q - r * 20;
        const int gy = min(max(ty * 16 - 1 + r, 0), H + 1), gx = min(max(tx * 16 - 1 + cc, 0), W + 1);
        c64r_glds4(img + (size_t)gy * (W + 2) + gx, pbase + buf * (C64R_PATCH * 4) + (j * NT + wave * 64) * 4);
      }
    }
  };
  // conv1a + ReLU of one tile = 21 groups of 16 halo pixels; wave w owns groups w, w+4, ..  (the sixth slot of every wave
  // is group 20: four identical copies, cheaper than a wave-divergent tail).  Everything that does not depend on the tile
  // is hoisted; the per-group work is branch-free so that it can be scheduled INTO the main MFMA loop (one group per combo).
  [[maybe_unused]] int prd[NG], pwr[NG], pyx[NG];
  if constexpr (FUSE1A) {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int k = min(wave + (32 / RW) * j, 20);
      const int p = k * 16 + l15, pc = min(p, 323);              // p >= 324 (group 20, lanes 4..15): lands in the pad rows
      const int py = pc / 18, px = pc - py * 18;
      prd[j] = ((py + min(g, 2)) * 20 + px) * 4;
      pwr[j] = p * 128 + ((g ^ (((p % 18) >> 1) & 7)) << 4);
      pyx[j] = (py << 8) | px;
    }
  }
  struct Taps { float q0, q1, q2; };
  auto prod_load_at = [&](int prd_j, int buf) {
    const float* q = reinterpret_cast<const float*>(smem + 2 * C64R_TILE_STRIDE + buf * (C64R_PATCH * 4) + prd_j);
    return Taps{q[0], q[1], q[2]};
  };
  auto prod_finish_at = [&](int pwr_j, int pyx_j, const Taps& tp3, int ty, int tx, int xbuf) {
    const int gy = ty * 16 - 1 + (pyx_j >> 8), gx = tx * 16 - 1 + (pyx_j & 255);
    const float m = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? 1.f : 0.f;
    const bool taps = g < 3;
    f16x8 bfr;
    bfr[0] = (_Float16)((taps ? tp3.q0 : 1.f) * m);
    bfr[1] = (_Float16)((taps ? tp3.q1 : 0.f) * m);
    bfr[2] = (_Float16)((taps ? tp3.q2 : 0.f) * m);
#pragma unroll
    for (int e = 3; e < 8; ++e) bfr[e] = (_Float16)0.f;
    f32x4 c1[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      f16x8 wf = w1[tt];
      if constexpr (CONST_IN_LDS) wf = *reinterpret_cast<const f16x8*>(smem + C64R_CONST_OFF + (tt * 64 + lane) * 16);
      c1[tt] = PF16::mfma(wf, bfr, f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = c1[2 * tp][e];
        v[4 + e] = c1[2 * tp + 1][e];
      }
      *reinterpret_cast<uint4*>(smem + xbuf * C64R_TILE_STRIDE + (pwr_j ^ (tp << 6))) = relu_packed(pack8<P>(v));
    }
  };
  auto prod_load = [&](int j, int buf) { return prod_load_at(prd[j], buf); };
  auto prod_finish = [&](int j, const Taps& tp3, int ty, int tx, int xbuf) { prod_finish_at(pwr[j], pyx[j], tp3, ty, tx, xbuf); };
  auto tile_xy = [&](int t, int& ty, int& tx) {
    int b_;
    tile_decode(t, per_img, tiles_x, sh_img, sh_x, b_, ty, tx);
  };

  int tile = blockIdx.x;
  if constexpr (FUSE1A) {
    if (tile < ntiles) stage_patch(tile, 0);
  } else {
    if (tile < ntiles) stage(tile, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (FUSE1A) {
    // pipeline prologue: patch(1) in flight, conv1a of tile(0) produced, both visible before the loop
    if (tile < ntiles) {
      if (tile + (int)gridDim.x < ntiles) stage_patch(tile + gridDim.x, 1);
      if (tile + 2 * (int)gridDim.x < ntiles) stage_patch(tile + 2 * gridDim.x, 2);
      int ty0, tx0;
      tile_xy(tile, ty0, tx0);
#pragma unroll
      for (int j = 0; j < NG; ++j) prod_finish(j, prod_load(j, 0), ty0, tx0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // (s_setprio for one wave half, or alternating per combo, does not help: younger high -8 % on the stage, alternating -37 %, older high
  //  within noise — profiles/r03_probe3_priority_ab_and_conv64r_timers.txt)
  int pb3 = 0, pb1 = 1;                                // i % 3 and (i + 1) % 3
#ifdef C64R_TIMING
  long long t_top = 0, t_mma = 0, t_wait = 0, t_bar = 0, t_epi = 0, tp_ = wall_clock64();
#endif
  for (int i = 0; tile < ntiles; ++i, tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if constexpr (FUSE1A) {
      // patch(i+3) -> pbuf[i % 3] (consumed by the conv1a of one iteration ago).  Three patch buffers: the wait at the end
      // of this iteration only covers patch(i+2), issued a whole iteration earlier — with two buffers it also covered
      // the patch issued a few hundred cycles ago, i.e. an HBM round trip per tile (25 % of the kernel, measured).
      if (tile + 3 * (int)gridDim.x < ntiles) stage_patch(tile + 3 * gridDim.x, pb3);
    } else {
      if (next < ntiles) stage(next, (i + 1) & 1);
    }

    f32x4 acc[RW][2];
#pragma unroll
    for (int m = 0; m < RW; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (CONST_IN_LDS) acc[m][t] = *reinterpret_cast<const f32x4*>(smem + C64R_CONST_OFF + 4096 + (ch * 4 + g) * 32 + t * 16);
        else acc[m][t] = f32x4{bias[t * 4], bias[t * 4 + 1], bias[t * 4 + 2], bias[t * 4 + 3]};
      }
    // 6 combos (column shift dx, channel half ks); the 10 pixel-row fragments of combo c+1 are requested before the
    // 48 MFMAs (3 filter rows x 8 pixel rows x 2 cout tiles) of combo c
    // PIN (RW = 4): fragment reads are inline-asm ds_read_b128 pinned one combo ahead of their MFMAs with sched_barrier(0)
    // and an explicit lgkmcnt wait — left to itself hipcc sinks every read down to its first use, and sched_group_barrier
    // pipelines made this variant slower.
    constexpr bool PIN = RW == 4 && C64R_PIN;
    constexpr int NB = (RW == 8 || PIN) ? 2 : 1;
    typename P::vec8 bf[NB][RW + 2];
    const int xoff = (i & 1) * C64R_TILE_STRIDE;
    [[maybe_unused]] int nty = 0, ntx = 0;
    if constexpr (FUSE1A) tile_xy(next < ntiles ? next : tile, nty, ntx);   // past the end: harmless rewrite of a dead buffer
    auto load_combo = [&](int c, int set) {
      const int dx = c >> 1, ks = c & 1;
      const char* fb = smem + ((cbase[dx] ^ (ks << 6)) + xoff);
#pragma unroll
      for (int r = 0; r < RW + 2; ++r) bf[set][r] = lds_frag<P>(fb, r * (18 * 128));
    };
    auto pin_combo = [&](int c, int set) {
      const int dx = c >> 1, ks = c & 1;
      const unsigned fa = lds_base + ((cbase[dx] ^ (ks << 6)) + xoff);
#pragma unroll
      for (int r = 0; r < RW + 2; ++r)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(bf[set][r]) : "v"(fa), "n"(r * (18 * 128)) : "memory");
    };
    if constexpr (PIN) {
      pin_combo(0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load_combo(0, 0);
      if (SCHED) __builtin_amdgcn_sched_group_barrier(0x100, RW + 2, 0);
    }
    C64R_T(t_top)
    // conv1a of the NEXT tile (its patch landed an iteration ago) is vector work + 4 small MFMAs per group.  The two waves of a SIMD are
    // served oldest first by the matrix pipe: the YOUNGER half of the waves produces its groups HERE, while it would wait for the pipe
    // anyway, the OLDER half after its MFMAs, while the younger half still runs its own (before: spread over the combos of every wave;
    // per-wave timers then: combos 669 vs 927 us per launch, barrier wait 324 vs 24 us — the younger waves' production sat in the
    // tail where they run alone).  (The older half producing ALL groups after its MFMAs measured 11 % slower: it becomes the critical path.)
    constexpr bool SPLIT_PROD = FUSE1A && C64R_SPLIT_PROD;
    if constexpr (SPLIT_PROD) {
      if (wave >= 16 / RW) {
#pragma unroll
        for (int j = 0; j < NG; ++j) prod_finish(j, prod_load(j, pb1), nty, ntx, (i + 1) & 1);
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if constexpr (PIN) {
        if (c + 1 < 6) { pin_combo(c + 1, (c + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
      } else {
        if (NB == 2 && c + 1 < 6) load_combo(c + 1, (c + 1) & 1);
        if (NB == 1 && c > 0) load_combo(c, 0);
      }
      [[maybe_unused]] Taps taps{};
      constexpr int PG = 6 / NG;                               // one conv1a group every PG combos
      const bool prod_here = FUSE1A && !SPLIT_PROD && c % PG == 0;
      if constexpr (FUSE1A) { if (prod_here) taps = prod_load(c / PG, pb1); }   // conv1a of the NEXT tile, group c (its patch landed an iteration ago)
      const int dx = c >> 1, ks = c & 1;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[m][t] = P::mfma(wreg[dy * 3 + dx][ks][t], bf[c & (NB - 1)][m + dy], acc[m][t]);
      if constexpr (FUSE1A) { if (prod_here) prod_finish(c / PG, taps, nty, ntx, (i + 1) & 1); }
      // issue-order pipeline for the scheduler: one fragment read of the NEXT combo per 4 MFMAs of this one (hipcc otherwise
      // sinks the reads down to their first use and every 6 MFMAs wait out a full LDS latency)
      if constexpr (PIN) {
        if (c + 1 < 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      }
      if (SCHED && NB == 2 && !PIN) {
#pragma unroll
        for (int sidx = 0; sidx < RW + 2; ++sidx) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (prod_here) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        if (prod_here) __builtin_amdgcn_sched_group_barrier(0x008, 6 * RW - 4 * (RW + 2) + 4, 0);
        else __builtin_amdgcn_sched_group_barrier(0x008, 6 * RW - 4 * (RW + 2), 0);
      }
    }
    if constexpr (SPLIT_PROD) {
      if (wave < 16 / RW) {
#pragma unroll
        for (int j = 0; j < NG; ++j) prod_finish(j, prod_load(j, pb1), nty, ntx, (i + 1) & 1);
      }
    }

    auto epilogue = [&]() {
    // ---- epilogue (the bias is already in the accumulators): round to the 2-byte storage type FIRST, then ReLU and the
    // 2x2 max-pool on packed pairs (v_pk_max_i16; rounding is monotonic, so pooling after it gives the same bits)
    int b, ty, tx;
    tile_decode(tile, per_img, tiles_x, sh_img, sh_x, b, ty, tx);
    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow + cb0 * 64 + ch * 32 + g * 8;
    uint4 pk[RW];
#pragma unroll
    for (int m = 0; m < RW; ++m) {
      pk[m].x = P::pack2(acc[m][0][0], acc[m][0][1]);
      pk[m].y = P::pack2(acc[m][0][2], acc[m][0][3]);
      pk[m].z = P::pack2(acc[m][1][0], acc[m][1][1]);
      pk[m].w = P::pack2(acc[m][1][2], acc[m][1][3]);
    }
    if constexpr (!POOL) {
#pragma unroll
      for (int m = 0; m < RW; ++m) {
        const int y = ty * 16 + ph * RW + m, x = tx * 16 + l15;
        *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT) = relu_packed(pk[m]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < RW / 4; ++q) {
        uint4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint4 vert = max_packed_pre_relu(pk[4 * q + 2 * h], pk[4 * q + 2 * h + 1]);
          v[h] = max_packed_pre_relu(vert, dpp_xor1(vert));
        }
        // both lanes of a column pair hold both pooled rows: the even lane stores row 2q, the odd lane row 2q + 1
        const uint4 r = relu_packed((l15 & 1) ? v[1] : v[0]);
        const int y = (ty * 16 + ph * RW) / 2 + 2 * q + (l15 & 1), x = tx * 8 + (l15 >> 1);
        *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT) = r;
      }
    }
    };
    // The OLDER half of the waves (first served by the matrix pipe) is done a quarter of a tile before the younger half: it packs, pools
    // and stores its outputs NOW, inside that slack, and starts the next tile's MFMAs right after the barrier — while the younger half
    // does its epilogue.  (Both halves doing it after the barrier left the matrix pipe idle for the length of an epilogue, ~10 % of a tile.)
    const bool early_epi = C64R_EARLY_EPI && wave < 16 / RW;
    if (early_epi) epilogue();
    // wait for the next tile's LDS-DMA BEFORE this tile's stores are issued (vmcnt counts stores too), then one barrier:
    // "next tile complete" and "the buffer just read is free"
    C64R_T(t_mma)
    if constexpr (FUSE1A) {
      // only patch(i+2) has to be down: the youngest vector-memory operation of every wave, patch(i+3), may stay in flight
      // (raw s_barrier: __syncthreads() would add its own vmcnt(0) for the output stores)
      constexpr int NPP = (C64R_PATCH + NT - 1) / NT;
      // (an early epilogue has just issued NST stores: they are younger than patch(i+2) and may stay in flight too)
      constexpr int NST = POOL ? RW / 4 : RW;
      if (tile + 3 * (int)gridDim.x < ntiles) {
        if (early_epi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPP + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPP) : "memory");
      } else {                                                              // tail: no patch(i+3) was issued
        if (early_epi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      C64R_T(t_wait)
      __builtin_amdgcn_s_barrier();
      C64R_T(t_bar)
      pb3 = pb3 == 2 ? 0 : pb3 + 1;
      pb1 = pb1 == 2 ? 0 : pb1 + 1;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (!early_epi) epilogue();
    C64R_T(t_epi)
  }
#ifdef C64R_TIMING
  if (lane == 0 && FUSE1A) {
    long long* o = c64_dbg + (blockIdx.x * 8 + wave) * 6;
    o[0] = t_top; o[1] = t_mma; o[2] = t_wait; o[3] = t_bar; o[4] = t_epi;
  }
#endif
}

#ifdef C64R_TIMING
}  // namespace airfe
extern "C" int airfe_dbg_c64(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c64_dbg), sizeof(long long) * 256 * 8 * 6); }
namespace airfe {
#endif

template <class P, bool POOL, bool FUSE1A>
static void conv64r_launch_t(const ConvArgs& a, hipStream_t st) {
  constexpr int RW = C64R_RW;
  constexpr int LDS = C64R_CONST_OFF + 4096 + 256;
  static PerDeviceOnce attr_once;
  auto kfn = conv64r_kernel<P, POOL, FUSE1A, RW>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 16;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int cb0 = 0; cb0 < a.COUT / 64; ++cb0)          // 64 output channels per pass
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(2048 / RW), LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

// requires CIN == 64, COUT % 64 == 0, relu, H % 16 == 0, W % 16 == 0.  a.img != nullptr selects the fused conv1a + conv1b
// form (input = fp32 image [B][H+2][W+2] with zero border, a.w1a [64][9], a.b1a [64]; always followed by the 2x2 max-pool).
void launch_conv64r(int prec, const ConvArgs& a, hipStream_t st) {
  if (a.img) {
    if (prec == 1) conv64r_launch_t<PF16, true, true>(a, st); else conv64r_launch_t<PBF16, true, true>(a, st);
    return;
  }
  if (prec == 1) {
    if (a.pool) conv64r_launch_t<PF16, true, false>(a, st); else conv64r_launch_t<PF16, false, false>(a, st);
  } else {
    if (a.pool) conv64r_launch_t<PBF16, true, false>(a, st); else conv64r_launch_t<PBF16, false, false>(a, st);
  }
}

}  // namespace airfe
