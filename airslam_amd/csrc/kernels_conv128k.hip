// airfe — persistent 3x3 convolution for the 128-input-channel layers, K-SPLIT form (round 4): same tile (16 x 8 pixels, 128 output
// channels per pass), same register-resident filters, but a wave owns 32 output channels x HALF the input channels instead of 16 x all.
//
// Why: in conv128r_kernel every one of the 8 waves reads every pixel fragment of the tile (12 combos x 10 rows = 120 ds_read_b128 per wave and
// tile = 983 KB of LDS reads per tile and workgroup: ~7700 LDS cycles against ~9200 MFMA cycles — the waves are bound by fragment reads, DESIGN.md
// §8.3).  Here a wave multiplies its 32 output channels (two 16-row MFMA tiles that share each pixel fragment) by 64 of the 128 input channels:
// 2 phases x 6 combos x 6 rows = 72 fragment reads for the same 288 MFMAs, and the two waves of a pair (input-channel halves 0 / 1 of the same
// outputs) exchange partial sums through LDS: the tile's rows 0-3 are finished by the half-0 wave, rows 4-7 by the half-1 wave (8 ds_write_b128 +
// 8 ds_read_b128 per wave and tile).  LDS traffic per tile 983 -> 590 + 131 KB.  Filters: 9 taps x 64 cin x 32 cout = 144 VGPRs, as before.
// A lane ends up with 8 CONTIGUOUS output channels of a pixel (tiles 2c, 2c + 1 of a slab: slab_row_to_feature), so results leave as 16-byte
// stores straight from the registers: no output staging tile, no transposition barrier.
// Sum order differs from conv128r_kernel (two partial sums per output instead of one chain): results agree to fp32 rounding, not bit for bit.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr128k;

__device__ __forceinline__ void c128k_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int C128K_TILE_BYTES = 10 * 18 * 256;    // 46080: 16x8 pixels + 1-pixel halo, 128 channels of 2 bytes
constexpr int C128K_PIECES = 10 * 18 * 16;         // 2880 sixteen-byte pieces
constexpr int C128K_XB = 2 * C128K_TILE_BYTES;     // partial-sum exchange: 2 phases x 4 pairs x 8 KiB
constexpr int C128K_LDS = C128K_XB + 2 * 4 * 8192;

template <class P, bool POOL>
__global__ __launch_bounds__(512, 1) void conv128k_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w sits on SIMD w & 3 (slot w >> 2): every SIMD gets ONE half-0 and ONE half-1 wave, so that while one of them is in its epilogue the other
  // keeps the SIMD's matrix pipe fed (with both waves of a SIMD in the same half the pipe idled through both epilogues: measured 8 % slower than conv128r)
  const int simd = wave & 3, slot = wave >> 2;
  const int cg = (simd >> 1) + 2 * slot, kh = (simd ^ slot) & 1;      // 32-cout group of the 128-cout pass, input-channel half
  const int cb = cb0 + (cg >> 1), c2 = cg & 1;         // 64-cout block and which pair of 16-row tiles inside its packed slabs
  const int H = a.H, W = a.W, COUT = a.COUT;
  const size_t in_row = (size_t)(W + 2) * 256;
  const size_t in_img = (size_t)(H + 2) * in_row;
  const int per_img = tiles_x * tiles_y;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr128k)smem);

  // ---- filters: slab (cb, tap, cin half kh), rows (2 c2 + tt) * 16 + l15, k-step ks -> MFMA A fragments
  typename P::vec8 wreg[9][2][2];
  {
    const char* wp = reinterpret_cast<const char*>(a.Wp) + (size_t)cb * 18 * SLAB_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int rr = (2 * c2 + tt) * 16 + l15;
          const uint4 u = *reinterpret_cast<const uint4*>(wp + (tap * 2 + kh) * SLAB_BYTES + rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
          wreg[tap][ks][tt] = __builtin_bit_cast(typename P::vec8, u);
        }
  }
  // the lane's 8 couts: features c2 * 32 + g * 8 + {0..7} of block cb (tile tt holds 4 of them)
  const int f0 = c2 * 32 + g * 8;
  const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + f0);
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + f0 + 4);

  int goff[6];                                         // input DMA: piece q = j*512 + tid -> pixel q>>4, LDS slot q&15 (as conv128r_kernel)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = min(j * 512 + tid, C128K_PIECES - 1);
    const int p = q >> 4;
    const int pr = p / 18, pc = p - pr * 18;
    const int c = (q & 15) ^ (pc & 15);
    goff[j] = pr * (int)in_row + pc * 256 + c * 16;
  }
  const bool last_piece = 5 * 512 + tid < C128K_PIECES;
  auto stage_tile = [&](int tile, int buf) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const char* xin = reinterpret_cast<const char*>(a.X) + (size_t)b * in_img + (size_t)ty * 8 * in_row + (size_t)tx * 16 * 256;
    const unsigned dst = lds_base + buf * C128K_TILE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < 5; ++j) c128k_glds16(xin + goff[j], dst + j * 8192);
    if (last_piece) c128k_glds16(xin + goff[5], dst + 5 * 8192);
  };
  int cbase[3];                                        // pixel fragments: column l15 + dx, chunk g of this wave's channel half
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) cbase[dx] = ((l15 + dx) * 256 + ((g ^ ((l15 + dx) & 15)) << 4)) ^ (kh << 7);

  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * COUT;
  char* xb_mine[2];                                    // [phase]: where this pair's exchange block of the phase lives
  xb_mine[0] = smem + C128K_XB + (0 * 4 + cg) * 8192 + lane * 16;
  xb_mine[1] = smem + C128K_XB + (1 * 4 + cg) * 8192 + lane * 16;

  // 4 output rows [r0, r0 + 4) of the tile in `xoff`: partial sums over this wave's 64 input channels
  auto phase = [&](int r0, int xoff, f32x4 (&acc)[4][2]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int dx = c >> 1, ks = c & 1;
      typename P::vec8 bf[6];
      const char* fb = smem + ((cbase[dx] ^ (ks << 6)) + xoff) + r0 * (18 * 256);
#pragma unroll
      for (int r = 0; r < 6; ++r) bf[r] = lds_frag<P>(fb, r * (18 * 256));
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[m][tt] = P::mfma(wreg[dy * 3 + dx][ks][tt], bf[m + dy], acc[m][tt]);
    }
  };
  // the finishing wave's epilogue for rows [r0, r0 + 4): + the partner's partial sums, round, ReLU / 2x2 max-pool on packed pairs, 16-byte stores
  auto finish = [&](int r0, int tile, const char* xb, f32x4 (&acc)[4][2]) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow + cb * 64 + f0;
    uint4 pk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(xb + (m * 2 + 0) * 1024);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(xb + (m * 2 + 1) * 1024);
      const f32x4 s0 = acc[m][0] + p0, s1 = acc[m][1] + p1;
      pk[m] = uint4{P::pack2(s0[0], s0[1]), P::pack2(s0[2], s0[3]), P::pack2(s1[0], s1[1]), P::pack2(s1[2], s1[3])};
    }
    if constexpr (!POOL) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        uint4 v = pk[m];
        if (a.relu) v = relu_packed(v);
        const int y = ty * 8 + r0 + m, x = tx * 16 + l15;
        *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT) = v;
      }
    } else {
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        uint4 v = max_packed_pre_relu(pk[2 * mp], pk[2 * mp + 1]);
        v = max_packed_pre_relu(v, dpp_xor1(v));
        if (a.relu) v = relu_packed(v);
        const int y = ty * 4 + (r0 >> 1) + mp, x = tx * 8 + (l15 >> 1);
        if (!(l15 & 1)) *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT) = v;
      }
    }
  };
  auto put_partial = [&](char* xb, const f32x4 (&acc)[4][2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) *reinterpret_cast<f32x4*>(xb + (m * 2 + tt) * 1024) = acc[m][tt];
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  if (tile < ntiles) stage_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[4][2];
  for (int i = 0; tile < ntiles; ++i, tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) stage_tile(next, (i + 1) & 1);
    const int xoff = (i & 1) * C128K_TILE_BYTES;
    // ---- phase 0: rows 0-3.  The half-0 wave carries the bias and will finish them; the half-1 wave hands its partial sums over.
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m][0] = kh == 0 ? bias0 : zero4; acc[m][1] = kh == 0 ? bias1 : zero4; }
    phase(0, xoff, acc);
    if (kh == 1) put_partial(xb_mine[0], acc);
    __syncthreads();                                   // B1: the phase-0 partials are in LDS
    if (kh == 0) finish(0, tile, xb_mine[0], acc);     // (its stores are YOUNGER than this trip's tile prefetch: see the wait below)
    // ---- phase 1: rows 4-7.  The half-1 wave carries the bias and finishes them (after the tile barrier); the half-0 wave hands over.
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m][0] = kh == 1 ? bias0 : zero4; acc[m][1] = kh == 1 ? bias1 : zero4; }
    phase(4, xoff, acc);
    if (kh == 0) put_partial(xb_mine[1], acc);
    // next tile landed: vmcnt retires in order and the half-0 waves' 4 (2 with pooling) result stores were issued AFTER this trip's prefetch,
    // so allowing that many to stay outstanding still proves the prefetch complete (waiting for the stores too would cost a memory round trip)
    if (kh == 0) {
      if constexpr (POOL) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                   // B2: phase-1 partials in LDS, everyone done reading this tile buffer, next tile visible
    if (kh == 1) finish(4, tile, xb_mine[1], acc);
  }
}

template <class P, bool POOL>
static void conv128k_launch_t(const ConvArgs& a, hipStream_t st) {
  static PerDeviceOnce attr_once;
  auto kfn = conv128k_kernel<P, POOL>;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, C128K_LDS);
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 8;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int cb0 = 0; cb0 < a.COUT / 64; cb0 += 2)          // 128 output channels per pass
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), C128K_LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

// requires CIN == 128, COUT % 128 == 0, H % 8 == 0, W % 16 == 0
void launch_conv128k(int prec, const ConvArgs& a, hipStream_t st) {
  if (prec == 1) {
    if (a.pool) conv128k_launch_t<PF16, true>(a, st); else conv128k_launch_t<PF16, false>(a, st);
  } else {
    if (a.pool) conv128k_launch_t<PBF16, true>(a, st); else conv128k_launch_t<PBF16, false>(a, st);
  }
}

}  // namespace airfe
