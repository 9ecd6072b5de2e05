// airfe — K = 256 linears of the matcher at large token counts (q/k/v projections, final projection): a STREAMING GEMM with the
// weights resident in registers.
//
// These GEMMs are HBM-bound (K = 256: 512 B in, 512-1024 B out per token for 131-262 kFLOP), yet the tiled kernels ran them at
// 1.8 TB/s: a 256 x 256 output tile has only four K chunks, so every workgroup is mostly pipeline fill and drain, one workgroup
// per CU.  Here a persistent workgroup owns 256 output features for the whole launch: wave w holds its 32 features x 256 K as
// 16 MFMA A-fragments (64 registers, loaded once), token tiles of 32 rows stream through an 8-slot LDS ring by LDS-DMA (seven
// tiles = 112 KiB in flight per CU), and each tile costs one barrier, 16 ds_read_b128, 32 MFMAs and 2-4 stores per wave.
// N = 512 (self q|k) runs as two feature groups, and q|k + v of a layer as ONE launch of three (gemmr_pair_kernel); the
// workgroups that read the same token tile sit on the same XCD (blockIdx, blockIdx + 8, ...) so the later reads are L2 hits.
//
// Nothing the compiler can see loads from global memory inside the tile loop (biases are preloaded into registers): a compiler-placed s_waitcnt would count only its own loads and drain the whole DMA ring.  The hand-placed
// waits rely on loads retiring in order: "at most 2*(tiles still in flight behind the wanted one)" outstanding.
// Rotary (self q|k): the cos / sin rows of a tile's tokens (2 x 4 KiB of fp32) ride the ring with it — one more DMA instruction per
// wave and tile, six slots of 24 KiB instead of eight of 16 — and the epilogue rotates the accumulator pairs from LDS.
// GATHER (round 5): the descriptor head over the SAMPLED cells of a large batch (airfe_detect.hip: four cells per keypoint, 204800 rows of convDa's map at 64 pairs)
// is the same shape — K = 256, N = 256, HBM-bound (512 B in, 1024 B of fp32 out per row) — and ran in the tiled 8-wave kernel (gemm8, rowidx form) at 2.3 TB/s
// with its waves parked 67 % of the time.  Here row r of a streamed tile comes from X1 row rowidx[r]: the indices of ALL of a workgroup's tiles go into LDS
// before the loop (a compiler-visible global load inside the loop would drain the DMA ring, see below), a lane reads its row's index from there (lgkmcnt) and
// the tile streams through the same ring; the epilogue stores fp32 rows (EPI_STORE_F32).  Same fragments, same K order, bias after the sum: the bits of gemm8.
#include "common.h"
#include "kernels.h"

namespace airfe {

constexpr int GR_MT = 2;                          // 16-token MFMA tiles per streamed tile
constexpr int GR_TT = 16 * GR_MT;                 // 32 tokens
constexpr int GR_XBYTES = GR_TT * 512;            // [32][256] 2-byte = 16 KiB
constexpr int GR_RBYTES = 2 * GR_TT * 128;        // cos | sin rows of the tile's tokens, [32][32] fp32 each = 8 KiB

template <int N>
__device__ __forceinline__ void gr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void gr_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

// One workgroup's whole launch: 256-feature slice `group` of linear `a`, token tiles first, first + per_group, ...
template <class P, bool TRANS, bool ROT, bool GATHER = false>
__device__ __forceinline__ void gemmr_body(const GemmArgs& a, char* smem, int group, int first, int per_group, int ntiles) {
  static_assert(!GATHER || (!TRANS && !ROT), "the gather form is the plain K = 256 linear with fp32 rows out");
  constexpr int GR_SLOT = GR_XBYTES + (ROT ? GR_RBYTES : 0);
  constexpr int GR_SLOTS = ROT ? 6 : 8;
  constexpr int PER = ROT ? 3 : 2;                                    // DMA instructions per wave and tile
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = first < ntiles ? (ntiles - first + per_group - 1) / per_group : 0;
  if (n == 0) return;

  // ---- this wave's weights: feature block cb, tile pair tp -> 16 fragments
  const int cb = group * 4 + (wave >> 1), tp = wave & 1;
  typename P::vec8 wreg[2][8];
  {
    const int sw = (l15 >> 1) & 7;
    const char* wb = reinterpret_cast<const char*>(a.Wp) + (size_t)cb * 4 * SLAB_BYTES + 2 * tp * 2048 + l15 * 128;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        wreg[u][ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(
            wb + u * 2048 + (ks >> 1) * SLAB_BYTES + ((((ks & 1) * 4 + g) ^ sw) << 4)));
  }
  f32x4 binit[2];                                                     // !TRANS: bias of the lane's 8 features
  float bt[2];                                                        // TRANS: bias of the lane's feature column per tile
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    binit[u] = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + tp * 32 + g * 8 + u * 4);
    bt[u] = a.bias[cb * 64 + slab_row_to_feature((2 * tp + u) * 16 + l15)];
  }
  // GATHER: the source rows of every tile this workgroup will stream, [n][32] behind the ring
  [[maybe_unused]] int* idx_lds = reinterpret_cast<int*>(smem + GR_SLOTS * GR_SLOT);
  if constexpr (GATHER) {       // (a.rowidx == nullptr: the identity — every row in order, fp32 rows out: the DENSE descriptor head of the junction images)
    for (int e = tid; e < n * GR_TT; e += 512) {
      const int r = (first + (e >> 5) * per_group) * GR_TT + (e & 31);
      idx_lds[e] = a.rowidx ? a.rowidx[r] : r;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (GATHER) __syncthreads();

  // ---- DMA of the workgroup's j-th tile (t = first + j per_group; 32 rows x 512 B) into ring slot `slot`: 16 wave-instructions of 1 KiB, two per wave, two rows each
  const int ld = a.ld1;
  auto dma = [&](int j, int slot) {
    [[maybe_unused]] const int t = first + j * per_group;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int inst = wave * 2 + i;
      const int r = inst * 2 + (lane >> 5), pp = lane & 31;
      size_t srow;
      if constexpr (GATHER) srow = (size_t)idx_lds[j * GR_TT + r];
      else srow = (size_t)(t * GR_TT + r);
      gr_glds16(a.X1 + srow * ld + ((pp ^ (r & 15)) << 3), (unsigned)(slot * GR_SLOT + inst * 1024));
    }
    if constexpr (ROT) {                                              // waves 0-3: cos rows, 4-7: sin rows; 8 rows of 128 B per instruction
      const float* src = (wave < 4 ? a.rot_cos : a.rot_sin) + ((size_t)t * GR_TT + (wave & 3) * 8) * 32 + lane * 4;
      gr_glds16(src, (unsigned)(slot * GR_SLOT + GR_XBYTES + wave * 1024));
    }
  };
#pragma unroll
  for (int j = 0; j < GR_SLOTS; ++j)
    if (j < n) dma(j, j);
  // tile 0 landed: at most PER * min(n - 1, GR_SLOTS - 1) younger DMA instructions may still be in flight
  if (n >= GR_SLOTS) gr_wait_vm<PER * (GR_SLOTS - 1)>();
  else gr_wait_vm<0>();
  __syncthreads();

  const int boff0 = l15 * 512;
  for (int i = 0; i < n; ++i) {
    const int t = first + i * per_group, slot = i % GR_SLOTS;          // (six slots with rotary: not a power of two)
    const char* xs = smem + slot * GR_SLOT + boff0;
    f32x4 acc[2][GR_MT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < GR_MT; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      typename P::vec8 bf[GR_MT];
#pragma unroll
      for (int m = 0; m < GR_MT; ++m) bf[m] = lds_frag<P>(xs, m * 16 * 512 + (((ks * 4 + g) ^ l15) << 4));
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < GR_MT; ++m) {
          if constexpr (TRANS) acc[u][m] = P::mfma(bf[m], wreg[u][ks], acc[u][m]);
          else acc[u][m] = P::mfma(wreg[u][ks], bf[m], acc[u][m]);
        }
    }
    // tile i+1 landed (this wave's part): behind it at most the tiles i+2 .. i+GR_SLOTS-1 are in flight, PER instructions each
    const int behind = n - 2 - i;
    if (behind >= GR_SLOTS - 2) gr_wait_vm<PER * (GR_SLOTS - 2)>();
    else if (behind == 5) gr_wait_vm<PER * 5>();
    else if (behind == 4) gr_wait_vm<PER * 4>();
    else if (behind == 3) gr_wait_vm<PER * 3>();
    else if (behind == 2) gr_wait_vm<PER * 2>();
    else if (behind == 1) gr_wait_vm<PER>();
    else gr_wait_vm<0>();

    // ---- epilogue of tile i (stores only)
    if constexpr (!TRANS) {
      const int co = cb * 64 + tp * 32 + g * 8;
#pragma unroll
      for (int m = 0; m < GR_MT; ++m) {
        const int row = t * GR_TT + m * 16 + l15;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[0][m][e] + binit[0][e];          // bias after the K sum, like the tiled kernels: bit-identical results
          v[4 + e] = acc[1][m][e] + binit[1][e];
        }
        if constexpr (ROT) {                                          // rotary on the pairs (2i, 2i+1) of the head dimension: light_glue rotary, as gemm_store_run
          const char* rt = smem + slot * GR_SLOT + GR_XBYTES + (m * 16 + l15) * 128 + (tp * 16 + g * 4) * 4;
          const f32x4 cs = *reinterpret_cast<const f32x4*>(rt), sn = *reinterpret_cast<const f32x4*>(rt + GR_RBYTES / 2);
          rotate_pairs(v, cs, sn);
        }
        if constexpr (GATHER) {                                       // EPI_STORE_F32: the lane's 8 features of output row `row` (dense row order, not the source's)
          float* o = reinterpret_cast<float*>(a.out) + (size_t)row * a.ldo + co;
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (a.epi == EPI_HEADS) {
          const int s = row / a.Np, nn = row - s * a.Np;
          const int sel = co >> 8, cw = co & 255, h = cw >> 6, d = cw & 63;
          uint16_t* o = reinterpret_cast<uint16_t*>(sel ? a.out2 : a.out) + (((size_t)s * a.H + h) * a.Np + nn) * 64 + d;
          *reinterpret_cast<uint4*>(o) = pack8<P>(v);
        } else {                                                      // EPI_STORE
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (size_t)row * a.ldo + co) = pack8<P>(v);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int co = cb * 64 + slab_row_to_feature((2 * tp + u) * 16 + l15);
        const int h = co >> 6, d = co & 63;
#pragma unroll
        for (int m = 0; m < GR_MT; ++m) {
          const int row0 = t * GR_TT + m * 16 + g * 4;
          const int sq = row0 / a.Np, nn = row0 - sq * a.Np;
          uint16_t* o = reinterpret_cast<uint16_t*>(a.out) + (((size_t)sq * a.H + h) * 64 + d) * a.Np + nn;
          *reinterpret_cast<uint2*>(o) = pack4<P>(acc[u][m][0] + bt[u], acc[u][m][1] + bt[u], acc[u][m][2] + bt[u], acc[u][m][3] + bt[u]);
        }
      }
    }
    __syncthreads();                                                  // every wave is done with slot i and has its part of tile i+1
    if (i + GR_SLOTS < n) dma(i + GR_SLOTS, slot);
  }
}

// Workgroup w -> (feature group, first tile): the groups that read the same token tile are w, w + 8, ... = the same XCD, so the
// tile comes from HBM once and from that XCD's L2 afterwards.
template <class P, bool TRANS, bool ROT>
__global__ __launch_bounds__(512, 1) void gemmr_kernel(GemmArgs a, int ntiles, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = blockIdx.x;
  gemmr_body<P, TRANS, ROT>(a, smem, (w >> 3) % ngroups, (w & 7) + 8 * (w / (8 * ngroups)), gridDim.x / ngroups, ntiles);
}

template <class P>
__global__ __launch_bounds__(512, 1) void gemmr_gather_kernel(GemmArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemmr_body<P, false, false, true>(a, smem, 0, blockIdx.x, gridDim.x, ntiles);
}

// q|k (or the cross block's shared qk) and v of one attention layer in ONE launch: the first ng_a feature groups belong to linear
// `a` (head-major rows, rotary when ROT), the last one to linear `b` (V, stored transposed) — the token tiles are read from HBM once
// for all of them and a launch is saved per layer.
template <class P, bool ROT>
__global__ __launch_bounds__(512, 1) void gemmr_pair_kernel(GemmArgs a, GemmArgs b, int ntiles, int ng_a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ngroups = ng_a + 1, w = blockIdx.x;
  const int group = (w >> 3) % ngroups, first = (w & 7) + 8 * (w / (8 * ngroups)), per_group = gridDim.x / ngroups;
  if (group < ng_a) gemmr_body<P, false, ROT>(a, smem, group, first, per_group, ntiles);
  else gemmr_body<P, true, false>(b, smem, 0, first, per_group, ntiles);
}

template <class P, bool ROT>
static void gemmr_pair_launch_t(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
  constexpr int LDS = ROT ? 6 * (GR_XBYTES + GR_RBYTES) : 8 * GR_XBYTES;
  static PerDeviceOnce attr_once;
  auto kfn = gemmr_pair_kernel<P, ROT>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int ng_a = a.cb_total / 4, ngroups = ng_a + 1;
  const int nwg = std::max(a.gr_wgs / (8 * ngroups), 1) * (8 * ngroups);   // 256 for two groups, 240 for three
  hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(512), LDS, st, a, b, a.M / GR_TT, ng_a);
}

template <class P, bool TRANS, bool ROT>
static void gemmr_launch_t(const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = ROT ? 6 * (GR_XBYTES + GR_RBYTES) : 8 * GR_XBYTES;
  static PerDeviceOnce attr_once;
  auto kfn = gemmr_kernel<P, TRANS, ROT>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int ngroups = a.cb_total / 4;
  const int nwg = std::max(a.gr_wgs / (8 * ngroups), 1) * (8 * ngroups);
  hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(512), LDS, st, a, a.M / GR_TT, ngroups);
}

// The gather form: y[r] = W x[rowidx[r]] + b as fp32 rows, K = N = 256 (the descriptor head over sampled cells).
bool gemmr_gather_applicable(int K, const GemmArgs& a) {
  return K == 256 && !a.X2 && !a.rot_cos && a.act == ACT_NONE && a.cb_total == 4 && a.N == 256 && a.M % GR_TT == 0 && a.epi == EPI_STORE_F32 && a.ldo >= 256 &&
         a.ld1 % 8 == 0 && (a.M / GR_TT + std::max(a.gr_wgs, 1) - 1) / std::max(a.gr_wgs, 1) <= 240;      // (the busiest workgroup's index list fits behind the ring)
}

void launch_gemmr_gather(int prec, const GemmArgs& a, hipStream_t st) {
  const int ntiles = a.M / GR_TT;
  const int nwg = std::max(std::min(a.gr_wgs, ntiles), 1);
  const int nmax = (ntiles + nwg - 1) / nwg;                          // tiles of the busiest workgroup: its index list sits behind the ring
  const int lds = 8 * GR_XBYTES + nmax * GR_TT * 4;
  if (prec == 1) {
    static PerDeviceOnce once;
    if (auto once_token = once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmr_gather_kernel<PF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(gemmr_gather_kernel<PF16>, dim3((unsigned)nwg), dim3(512), lds, st, a, ntiles);
  } else {
    static PerDeviceOnce once;
    if (auto once_token = once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmr_gather_kernel<PBF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(gemmr_gather_kernel<PBF16>, dim3((unsigned)nwg), dim3(512), lds, st, a, ntiles);
  }
}

// GATHER, K = N = 128 (round 6): the LOI head of the PLNet line branch at the junctions' tap rows (airfe_detect.hip line_tail_dev: 1200 rows per image, 153600 at
// 64 pairs; 256 B in, 512 B of fp32 out per row: HBM work) ran in the tiled 8-wave kernel (gemm8, rowidx form) at 0.8 TB/s — two K chunks per 256-row tile are
// all fill and drain.  The same streaming shape as above with the geometry of this head: a tile = 64 gathered rows of 256 B (16 KiB: the ring's slot), 16
// wave-instructions of four rows each; wave w owns the 32 features (w & 3) of rows 32 (w >> 2) .. + 32 — 8 weight fragments in registers, 8 ds_read_b128 and
// 16 MFMAs per tile.  Same fragments, same K order (four 32-wide steps, ascending), bias after the sum: the bits of gemm8.
constexpr int G128_TT = 64;                        // rows per streamed tile
template <class P>
__global__ __launch_bounds__(512, 1) void gemmr_gather128_kernel(GemmArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOTS = 8, SLOT = G128_TT * 256, PER = 2;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = blockIdx.x, per_group = gridDim.x;
  const int n = first < ntiles ? (ntiles - first + per_group - 1) / per_group : 0;
  if (n == 0) return;
  const int cb = (wave & 3) >> 1, tp = wave & 1, rh = wave >> 2;
  typename P::vec8 wreg[2][4];
  {
    const int sw = (l15 >> 1) & 7;
    const char* wb = reinterpret_cast<const char*>(a.Wp) + (size_t)cb * 2 * SLAB_BYTES + 2 * tp * 2048 + l15 * 128;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        wreg[u][ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(
            wb + u * 2048 + (ks >> 1) * SLAB_BYTES + ((((ks & 1) * 4 + g) ^ sw) << 4)));
  }
  f32x4 binit[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) binit[u] = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + tp * 32 + g * 8 + u * 4);
  int* idx_lds = reinterpret_cast<int*>(smem + SLOTS * SLOT);
  for (int e = tid; e < n * G128_TT; e += 512) idx_lds[e] = a.rowidx[(size_t)(first + (e >> 6) * per_group) * G128_TT + (e & 63)];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int ld = a.ld1;
  auto dma = [&](int j, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int inst = wave * 2 + i;
      const int r = inst * 4 + (lane >> 4), pp = lane & 15;
      const size_t srow = (size_t)idx_lds[j * G128_TT + r];
      gr_glds16(a.X1 + srow * ld + ((pp ^ (r & 15)) << 3), (unsigned)(slot * SLOT + inst * 1024));
    }
  };
#pragma unroll
  for (int j = 0; j < SLOTS; ++j)
    if (j < n) dma(j, j);
  if (n >= SLOTS) gr_wait_vm<PER * (SLOTS - 1)>();
  else gr_wait_vm<0>();
  __syncthreads();

  for (int i = 0; i < n; ++i) {
    const int t = first + i * per_group, slot = i & (SLOTS - 1);
    const char* xs = smem + slot * SLOT + (rh * 32 + l15) * 256;
    f32x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typename P::vec8 bf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) bf[m] = lds_frag<P>(xs, m * 16 * 256 + (((ks * 4 + g) ^ l15) << 4));
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[u][m] = P::mfma(wreg[u][ks], bf[m], acc[u][m]);
    }
    const int behind = n - 2 - i;
    if (behind >= SLOTS - 2) gr_wait_vm<PER * (SLOTS - 2)>();
    else if (behind == 5) gr_wait_vm<PER * 5>();
    else if (behind == 4) gr_wait_vm<PER * 4>();
    else if (behind == 3) gr_wait_vm<PER * 3>();
    else if (behind == 2) gr_wait_vm<PER * 2>();
    else if (behind == 1) gr_wait_vm<PER>();
    else gr_wait_vm<0>();
    const int co = cb * 64 + tp * 32 + g * 8;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int row = t * G128_TT + rh * 32 + m * 16 + l15;
      float* o = reinterpret_cast<float*>(a.out) + (size_t)row * a.ldo + co;
      *reinterpret_cast<float4*>(o) = make_float4(acc[0][m][0] + binit[0][0], acc[0][m][1] + binit[0][1], acc[0][m][2] + binit[0][2], acc[0][m][3] + binit[0][3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(acc[1][m][0] + binit[1][0], acc[1][m][1] + binit[1][1], acc[1][m][2] + binit[1][2], acc[1][m][3] + binit[1][3]);
    }
    __syncthreads();
    if (i + SLOTS < n) dma(i + SLOTS, slot);
  }
}

bool gemmr_gather128_applicable(const GemmArgs& a) {
  return a.rowidx && !a.X2 && !a.rot_cos && a.act == ACT_NONE && a.cb_total == 2 && a.N == 128 && a.K1 == 128 && a.M % G128_TT == 0 && a.epi == EPI_STORE_F32 &&
         a.ldo >= 128 && a.ld1 % 8 == 0 && (a.M / G128_TT + std::max(a.gr_wgs, 1) - 1) / std::max(a.gr_wgs, 1) <= 120;      // (the busiest workgroup's index list fits behind the ring)
}

void launch_gemmr_gather128(int prec, const GemmArgs& a, hipStream_t st) {
  const int ntiles = a.M / G128_TT;
  const int nwg = std::max(std::min(a.gr_wgs, ntiles), 1);
  const int nmax = (ntiles + nwg - 1) / nwg;
  const int lds = 8 * G128_TT * 256 + nmax * G128_TT * 4;
  if (prec == 1) {
    static PerDeviceOnce once;
    if (auto once_token = once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmr_gather128_kernel<PF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(gemmr_gather128_kernel<PF16>, dim3((unsigned)nwg), dim3(512), lds, st, a, ntiles);
  } else {
    static PerDeviceOnce once;
    if (auto once_token = once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemmr_gather128_kernel<PBF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(gemmr_gather128_kernel<PBF16>, dim3((unsigned)nwg), dim3(512), lds, st, a, ntiles);
  }
}

bool gemmr_applicable(int K, bool trans, const GemmArgs& a) {
  const int ng = a.cb_total / 4;
  return K == 256 && !a.X2 && (!a.rot_cos || (!trans && a.epi == EPI_HEADS)) && a.act == ACT_NONE && a.cb_total % 4 == 0 && (ng == 1 || ng == 2) && a.N == a.cb_total * 64 &&
         a.M % GR_TT == 0 && (trans ? a.epi == EPI_HEADS_T : (a.epi == EPI_HEADS || (a.epi == EPI_STORE && a.ldo >= a.N)));
}

bool gemmr_pair_applicable(const GemmArgs& a, const GemmArgs& b) {
  return gemmr_applicable(256, false, a) && gemmr_applicable(256, true, b) && a.epi == EPI_HEADS && b.cb_total == 4 && a.X1 == b.X1 &&
         a.ld1 == b.ld1 && a.M == b.M;
}

void launch_gemmr_pair(int prec, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
  if (prec == 1) {
    if (a.rot_cos) gemmr_pair_launch_t<PF16, true>(a, b, st); else gemmr_pair_launch_t<PF16, false>(a, b, st);
  } else {
    if (a.rot_cos) gemmr_pair_launch_t<PBF16, true>(a, b, st); else gemmr_pair_launch_t<PBF16, false>(a, b, st);
  }
}

void launch_gemmr(int prec, bool trans, const GemmArgs& a, hipStream_t st) {
  if (prec == 1) {
    if (trans) gemmr_launch_t<PF16, true, false>(a, st);
    else if (a.rot_cos) gemmr_launch_t<PF16, false, true>(a, st);
    else gemmr_launch_t<PF16, false, false>(a, st);
  } else {
    if (trans) gemmr_launch_t<PBF16, true, false>(a, st);
    else if (a.rot_cos) gemmr_launch_t<PBF16, false, true>(a, st);
    else gemmr_launch_t<PBF16, false, false>(a, st);
  }
}

}  // namespace airfe
