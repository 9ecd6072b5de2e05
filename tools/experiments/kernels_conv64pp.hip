// EXPERIMENT, NOT BUILT (round 1): the fused conv1a + conv1b + 2x2 max-pool layer as TWO independent 4-wave
// workgroups per CU with ONE tile buffer each ("ping-pong").  250 VGPRs, no spills, exact parity on the detector and stereo GPU
// suites; measured 3.069 ms against 3.082 ms for kernels_conv64r.hip's 8-wave workgroup, i.e. no gain: without the conv1a
// production interleaved into the MFMA combos the workgroup loses what the desynchronisation wins.  Kept as the starting point for
// a two-workgroup form WITH interleaved production.  To try it: add the file to airslam_amd/build.py, declare
// `void launch_conv64pp(int prec, const ConvArgs& a, hipStream_t st);` in kernels.h and call it instead of launch_conv64r for a.img != nullptr.
//
//   wave w of 4: pixel rows 8 (w>>1) .. +8 in two passes of 4 rows (the pass shares the filter registers: 144 VGPRs), couts
//   32 (w&1) .. +32.  Filter packing, column swizzle, conv1a on the matrix pipe and the packed epilogue are the 8-wave kernel's.
#include "../../airslam_amd/csrc/common.h"
#include "../../airslam_amd/csrc/kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr64pp;

__device__ __forceinline__ void cpp_glds4(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int CPP_TILE = 21 * 16 * 128;            // 43008: the 18x18x64 halo tile padded to 21 whole groups of 16 pixels
constexpr int CPP_PATCH = 20 * 20;                 // fp32 image patch per tile (halo 2)
constexpr int CPP_PATCH_OFF = CPP_TILE;            // three patch buffers
constexpr int CPP_CONST_OFF = CPP_TILE + 3 * CPP_PATCH * 4;   // conv1a A fragments (4 KiB) + bias (256 B)
constexpr int CPP_LDS = CPP_CONST_OFF + 4096 + 256;
constexpr int CPP_NT = 256;
constexpr int CPP_NPP = (CPP_PATCH + CPP_NT - 1) / CPP_NT;    // patch DMA instructions per thread (2)

template <class P>
__global__ __launch_bounds__(CPP_NT, 2) void conv64pp_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int phb = wave >> 1, ch = wave & 1;            // 8-row block, cout half of this wave
  const int H = a.H, W = a.W, COUT = a.COUT;
  const int per_img = tiles_x * tiles_y;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr64pp)smem);

  typename P::vec8 wreg[9][2][2];                      // filters: 36 A fragments = 144 VGPRs
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
  if (wave == 0) {                                     // conv1a's A fragments (g = filter row, g = 3: bias against a constant 1)
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
      *reinterpret_cast<f16x8*>(smem + CPP_CONST_OFF + (t * 64 + lane) * 16) = w;
    }
  }
  if (wave < 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) reinterpret_cast<float*>(smem + CPP_CONST_OFF + 4096)[(wave * 4 + g) * 8 + e] = a.bias[cb0 * 64 + wave * 32 + g * 8 + e];
  }

  const int Ho = H / 2, Wo = W / 2;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * COUT;

  int cbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) cbase[dx] = (phb * 8 * 18 + l15 + dx) * 128 + ((g ^ (((l15 + dx) >> 1) & 7)) << 4);

  auto stage_patch = [&](int t, int buf) {
    const int b = t / per_img, rem = t - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const float* img = a.img + (size_t)b * (H + 2) * (W + 2);
#pragma unroll
    for (int j = 0; j < CPP_NPP; ++j) {
      const int q = j * CPP_NT + tid;
      if (q < CPP_PATCH) {
        const int r = q / 20, cc = q - r * 20;
        const int gy = min(max(ty * 16 - 1 + r, 0), H + 1), gx = min(max(tx * 16 - 1 + cc, 0), W + 1);
        cpp_glds4(img + (size_t)gy * (W + 2) + gx, lds_base + CPP_PATCH_OFF + buf * (CPP_PATCH * 4) + (j * CPP_NT + wave * 64) * 4);
      }
    }
  };
  auto produce = [&](int buf, int ty, int tx) {
#pragma unroll 1                                          // rolled: unrolled, hipcc hoists every group's address arithmetic out of the tile loop and spills
    for (int j = 0; j < 6; ++j) {
      const int k = min(wave + 4 * j, 20);
      const int p = k * 16 + l15, pc = min(p, 323);
      const int py = pc / 18, px = pc - py * 18;
      const float* q = reinterpret_cast<const float*>(smem + CPP_PATCH_OFF + buf * (CPP_PATCH * 4) + ((py + min(g, 2)) * 20 + px) * 4);
      const float q0 = q[0], q1 = q[1], q2 = q[2];
      const int gy = ty * 16 - 1 + py, gx = tx * 16 - 1 + px;
      const float m = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? 1.f : 0.f;
      const bool taps = g < 3;
      f16x8 bfr;
      bfr[0] = (_Float16)((taps ? q0 : 1.f) * m);
      bfr[1] = (_Float16)((taps ? q1 : 0.f) * m);
      bfr[2] = (_Float16)((taps ? q2 : 0.f) * m);
#pragma unroll
      for (int e = 3; e < 8; ++e) bfr[e] = (_Float16)0.f;
      f32x4 c1[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const f16x8 wf = *reinterpret_cast<const f16x8*>(smem + CPP_CONST_OFF + (tt * 64 + lane) * 16);
        c1[tt] = PF16::mfma(wf, bfr, f32x4{0.f, 0.f, 0.f, 0.f});
      }
      const int pwr = p * 128 + ((g ^ (((p % 18) >> 1) & 7)) << 4);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = c1[2 * tp][e];
          v[4 + e] = c1[2 * tp + 1][e];
        }
        *reinterpret_cast<uint4*>(smem + (pwr ^ (tp << 6))) = relu_packed(pack8<P>(v));
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  stage_patch(tile, 0);
  if (tile + (int)gridDim.x < ntiles) stage_patch(tile + gridDim.x, 1);
  if (tile + 2 * (int)gridDim.x < ntiles) stage_patch(tile + 2 * gridDim.x, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int pb = 0;                                           // patch buffer of the current tile: i % 3
  for (; tile < ntiles; tile += gridDim.x) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    produce(pb, ty, tx);                                // conv1a of this tile into the single tile buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // tile complete; patch buffer pb is free
    if (tile + 3 * (int)gridDim.x < ntiles) stage_patch(tile + 3 * gridDim.x, pb);

    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow + cb0 * 64 + ch * 32 + g * 8;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 acc[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = *reinterpret_cast<const f32x4*>(smem + CPP_CONST_OFF + 4096 + (ch * 4 + g) * 32 + t * 16);
      typename P::vec8 bf[2][6];
      auto pin_combo = [&](int c, int set) {
        const int dx = c >> 1, ks = c & 1;
        const unsigned fa = lds_base + ((cbase[dx] ^ (ks << 6)) + pass * (4 * 18 * 128));
#pragma unroll
        for (int r = 0; r < 6; ++r)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(bf[set][r]) : "v"(fa), "n"(r * (18 * 128)) : "memory");
      };
      pin_combo(0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c + 1 < 6) { pin_combo(c + 1, (c + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
        const int dx = c >> 1, ks = c & 1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[m][t] = P::mfma(wreg[dy * 3 + dx][ks][t], bf[c & 1][m + dy], acc[m][t]);
        if (c + 1 < 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      }
      if (pass == 1) {
        // the next tile's patch must be down before the barrier below; waited for BEFORE this pass's stores are issued (vmcnt counts
        // stores too).  Wave 3 issues one DMA instruction per patch (the 400 floats end in wave 2), the others two.
        if (tile + 3 * (int)gridDim.x >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wave == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      uint4 pk[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        pk[m].x = P::pack2(acc[m][0][0], acc[m][0][1]);
        pk[m].y = P::pack2(acc[m][0][2], acc[m][0][3]);
        pk[m].z = P::pack2(acc[m][1][0], acc[m][1][1]);
        pk[m].w = P::pack2(acc[m][1][2], acc[m][1][3]);
      }
      uint4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint4 vert = max_packed_pre_relu(pk[2 * h], pk[2 * h + 1]);
        v[h] = max_packed_pre_relu(vert, dpp_xor1(vert));
      }
      const uint4 r = relu_packed((l15 & 1) ? v[1] : v[0]);
      const int y = (ty * 16 + phb * 8 + pass * 4) / 2 + (l15 & 1), x = tx * 8 + (l15 >> 1);
      *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT) = r;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // tile buffer free for the next produce; next patch visible
    pb = pb == 2 ? 0 : pb + 1;
  }
}

template <class P>
static void conv64pp_launch_t(const ConvArgs& a, hipStream_t st) {
  static bool attr_done = false;
  auto kfn = conv64pp_kernel<P>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, CPP_LDS);
    attr_done = true;
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 16;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 512 ? ntiles : 512;          // two workgroups per CU
  for (int cb0 = 0; cb0 < a.COUT / 64; ++cb0)
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(CPP_NT), CPP_LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

void launch_conv64pp(int prec, const ConvArgs& a, hipStream_t st) {
  if (prec == 1) conv64pp_launch_t<PF16>(a, st); else conv64pp_launch_t<PBF16>(a, st);
}

}  // namespace airfe
