// airfe — the detector pipeline (encoder, heads, NMS, top-K, descriptors) and the PLNet line path on the device (see airfe_host.h)
#include "airfe_host.h"

namespace airfe_host {

int ensure_tables(airfe_ctx* c, int h, int w) {
  if (c->tab_w == w && c->tab_h == h) return 0;
  const auto xt = resize_table(AIRFE_INTERNAL_SIZE, w), yt = resize_table(AIRFE_INTERNAL_SIZE, h);
  // the image size changed: a pre-process of the previous size may still be reading the tables on a CALLER's stream (the *_dev entry
  // points), so the whole device is drained before they are rewritten — once per size change, not per call
  if (c->tab_w != -1) HIPCHK(c, hipDeviceSynchronize());   // (-1: no table yet, nothing can be reading it)
  HIPCHK(c, hipMemcpyAsync(c->xtab, xt.data(), xt.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->ytab, yt.data(), yt.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // host vectors go out of scope
  c->tab_w = w;
  c->tab_h = h;
  return 0;
}

int run_conv(airfe_ctx* c, const ConvW& w, const uint16_t* x, uint16_t* y, int B, int H, int W, int pool, int out_pad,
             hipStream_t st) {
  ConvArgs a;
  a.X = x; a.Wp = w.w; a.bias = w.b; a.Y = y;
  a.B = B; a.H = H; a.W = W; a.CIN = w.cin; a.COUT = w.cout;
  a.pool = pool; a.out_pad = out_pad; a.relu = 1;
  const double px = (double)B * H * W;
  const double ob = px / (pool ? 4 : 1) * w.cout * 2;
  {
    ProfScope ps(c, w.cin == 64 ? ST_CONV3X3_C64 : ST_CONV3X3_C128, st, 2.0 * px * w.cin * w.cout * 9,
                 px * w.cin * 2 + ob + 9.0 * w.cin * w.cout * 2);
    launch_conv3x3(c->prec, a, st);
  }
  return c->cfg.check_launches ? launch_status(c) : 0;       // (without the flag: once per pipeline, at its end)
}

// ---- fp32 correctness path: the SuperPoint-VGG encoder + heads up to the dense logits / descriptor maps (what follows — soft-max,
// NMS, top-K, descriptor sampling — is fp32 in every mode and shared)
void f32_conv(const airfe_ctx::F32Conv& w, const float* x, float* y, int B, int H, int W, int opad, hipStream_t st) {
  launch_conv3x3_f32(x, w.w, w.b, y, B, H, W, w.cin, w.cout, opad, st);
}
int encode_f32(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, hipStream_t st) {
  const int R = AIRFE_INTERNAL_SIZE;
  for (int c0 = 0; c0 < B; c0 += c->f_B) {
    const int cb = std::min(c->f_B, B - c0);
    launch_preprocess(d_gray + (size_t)c0 * img_stride, cb, h, w, stride, img_stride, c->xtab, c->ytab, c->lut, c->img32, R, R, st);
    launch_conv1a_f32(c->img32, c->c1a_w, c->c1a_b, c->f1a, cb, R, R, st);
    f32_conv(c->f_c1b, c->f1a, c->f1b, cb, R, R, 1, st);
    launch_maxpool2_f32(c->f1b, c->fp1, cb, R, R, 64, st);
    f32_conv(c->f_c2a, c->fp1, c->f2a, cb, R / 2, R / 2, 1, st);
    f32_conv(c->f_c2b, c->f2a, c->f2b, cb, R / 2, R / 2, 1, st);
    launch_maxpool2_f32(c->f2b, c->fp2, cb, R / 2, R / 2, 64, st);
    f32_conv(c->f_c3a, c->fp2, c->f3a, cb, R / 4, R / 4, 1, st);
    f32_conv(c->f_c3b, c->f3a, c->f3b, cb, R / 4, R / 4, 1, st);
    launch_maxpool2_f32(c->f3b, c->fp3, cb, R / 4, R / 4, 128, st);
    f32_conv(c->f_c4a, c->fp3, c->f4a, cb, R / 8, R / 8, 1, st);
    f32_conv(c->f_c4b, c->f4a, c->f4b, cb, R / 8, R / 8, 1, st);
    f32_conv(c->f_cPa, c->f4b, c->fPa, cb, R / 8, R / 8, 0, st);
    f32_conv(c->f_cDa, c->f4b, c->fDa, cb, R / 8, R / 8, 0, st);
    const int cells = cb * (R / 8) * (R / 8);
    const size_t cell0 = (size_t)c0 * (R / 8) * (R / 8);
    GemmF32Args g;
    g.X1 = c->fPa; g.ld1 = 256; g.K1 = 256; g.K = 256; g.W = c->f_cPb.w; g.bias = c->f_cPb.b; g.M = cells; g.N = 65;
    g.Y = c->logits + cell0 * 72; g.ldy = 72;
    launch_gemm_f32(g, st);
    g.X1 = c->fDa; g.W = c->f_cDb.w; g.bias = c->f_cDb.b; g.N = 256; g.Y = c->desc + cell0 * 256; g.ldy = 256;
    launch_gemm_f32(g, st);
  }
  launch_softmax_d2s(c->logits, 72, c->heat, B, R / 8, R / 8, st);
  c->desc_normalised = false;
  return launch_status(c);
}

// convDb over every cell of the batch -> c->desc [B][64][64][256] fp32, un-normalised
int dense_desc_head(airfe_ctx* c, int B, hipStream_t st) {
  const int R = AIRFE_INTERNAL_SIZE, cells = B * (R / 8) * (R / 8);
  GemmArgs g;
  g.X1 = c->aDa; g.ld1 = 256; g.K1 = 256; g.Wp = c->cDb.w; g.bias = c->cDb.b;
  g.M = cells; g.N = 256; g.cb_total = c->cDb.cbt; g.epi = EPI_STORE_F32; g.out = c->desc; g.ldo = 256;
  g.small_max = c->gemm_small_max; g.g8_min = c->gemm8_min; g.gr_min = c->gemmr_min; g.gr_wgs = c->gemmr_wgs;
  {
    ProfScope ps(c, ST_HEAD_GEMM, st, 2.0 * cells * 256 * 256, (double)cells * (512 + 1024));
    // large batches (round 6): 512 B in and 1 KB of fp32 out per cell is HBM work — the tiled kernel ran it at 1.8 TB/s (225 us per 64 images); the streaming kernel's
    // identity-index form (kernels_gemmr.hip) has the same fragments, K order and bias placement: the same bits
    if (c->desc_gather_stream && cells >= c->gemmr_min && gemmr_gather_applicable(256, g)) launch_gemmr_gather(c->prec, g, st);
    else launch_gemm(c->prec, 256, false, g, st);
  }
  // F.normalize of the dense map is applied lazily: sample_desc_kernel normalises just the 4 taps each keypoint reads
  // (same operations, same bits) — a dense pass moved 8 MB/image to serve 400 x 4 cell reads.
  c->desc_normalised = false;
  return c->cfg.check_launches ? launch_status(c) : 0;
}

// Detector over ONE batch of B images, or — d_gray1 != nullptr — over the 2 B images of B stereo pairs in one pass (images 0 .. B-1 from
// d_gray, B .. 2B-1 from d_gray1; features to d_feat / d_feat1): every whole-batch kernel then runs once over twice the tiles instead
// of twice (half the launches, prologues and tails of the second half of the network; per-image results do not depend on the batch).
int detect_dev2(airfe_ctx* c, const uint8_t* d_gray, const uint8_t* d_gray1, int Bs, int h, int w, int stride, size_t img_stride,
                float* d_feat, float* d_feat1, int cap, int* d_n, int* d_n1, hipStream_t st) {
  if (!c->has_sp) return fail(c, "detector weights were not loaded (cfg.superpoint_pack)");
  const int B = d_gray1 ? 2 * Bs : Bs;
  if (Bs < 1 || Bs > c->Bmax || B > c->Dmax) return fail(c, "batch exceeds cfg.max_batch");
  // Two sources / two destinations that are in fact ONE array (the batch-1 keyframe entry lays left and right out back to back): the per-side
  // launches below become one launch over the 2 Bs images — per image the same work, so the same bits.
  const bool src_contig = d_gray1 && d_gray1 == d_gray + (size_t)Bs * img_stride;
  const bool dst_contig = d_gray1 && d_feat1 == d_feat + (size_t)Bs * cap * AIRFE_FEAT_DIM && d_n1 == d_n + Bs;
  if (h < 1 || w < 1) return fail(c, "empty image");
  if (cap < c->cfg.max_keypoints) return fail(c, "feature capacity < max_keypoints");
  if (ensure_tables(c, h, w)) return 1;
  const int R = AIRFE_INTERNAL_SIZE;
  bool sparse_desc = false;
  if (c->prec == 2) {
    if (d_gray1) return fail(c, "detect_dev2: the fp32 path takes one source");
    if (encode_f32(c, d_gray, B, h, w, stride, img_stride, st)) return 1;
  } else {
    for (int c0 = 0, cb = 0; c0 < B; c0 += cb) {
      cb = std::min(c->chunk, B - c0);
      {                                                                  // a chunk may straddle the two sources: one pre-process launch per source
        ProfScope ps(c, ST_PREPROCESS, st, 0, (double)cb * ((double)h * w + (double)R * R * 4));
        int n0 = std::min(std::max(Bs - c0, 0), cb);                    // images of this chunk that come from d_gray
        if (src_contig) n0 = cb;                                        // (the second source lies right behind the first: one launch)
        if (n0 > 0) launch_preprocess(d_gray + (size_t)c0 * img_stride, n0, h, w, stride, img_stride, c->xtab, c->ytab, c->lut, c->img32, R, R, st);
        if (cb > n0)
          launch_preprocess(d_gray1 + (size_t)(c0 + n0 - Bs) * img_stride, cb - n0, h, w, stride, img_stride, c->xtab, c->ytab, c->lut,
                            c->img32 + (size_t)n0 * (R + 2) * (R + 2), R, R, st);
      }
      {
        // conv1a (Cin = 1) fused into the persistent conv1b kernel: its 64-channel full-resolution output never
        // reaches HBM (kernels_conv64r.hip).  FLOPs/bytes below are the algorithmic ones of conv1a + conv1b.
        ConvArgs a;
        a.Wp = c->c1b.w; a.bias = c->c1b.b; a.Y = c->a1b; a.B = cb; a.H = R; a.W = R; a.CIN = 64; a.COUT = 64;
        a.pool = 1; a.out_pad = 1; a.relu = 1;
        a.img = c->img32; a.w1a = c->c1a_w; a.b1a = c->c1a_b;
        const double px = (double)cb * R * R;
        ProfScope ps(c, ST_CONV1_FUSED, st, 2.0 * px * 9 * 64 + 2.0 * px * 64 * 64 * 9, px * 4 + px / 4 * 128 + 9.0 * 64 * 64 * 2);
        launch_conv64r(c->prec, a, st);
      }
      if (run_conv(c, c->c2a, c->a1b, c->a2a, cb, R / 2, R / 2, 0, 1, st)) return 1;
      if (run_conv(c, c->c2b, c->a2a, c->a2b + (size_t)c0 * (R / 4 + 2) * (R / 4 + 2) * 64, cb, R / 2, R / 2, 1, 1, st)) return 1;
    }
    if (run_conv(c, c->c3a, c->a2b, c->a3a, B, R / 4, R / 4, 0, 1, st)) return 1;
    if (run_conv(c, c->c3b, c->a3a, c->a3b, B, R / 4, R / 4, 1, 1, st)) return 1;
    if (run_conv(c, c->c4a, c->a3b, c->a4a, B, R / 8, R / 8, 0, 1, st)) return 1;
    if (run_conv(c, c->c4b, c->a4a, c->a4b, B, R / 8, R / 8, 0, 1, st)) return 1;
    if (run_conv(c, c->cPa, c->a4b, c->aPa, B, R / 8, R / 8, 0, 0, st)) return 1;
    if (run_conv(c, c->cDa, c->a4b, c->aDa, B, R / 8, R / 8, 0, 0, st)) return 1;
    const int cells = B * (R / 8) * (R / 8);
    {
      GemmArgs g;
      g.X1 = c->aPa; g.ld1 = 256; g.K1 = 256; g.Wp = c->cPb.w; g.bias = c->cPb.b;
      g.M = cells; g.N = 65; g.cb_total = c->cPb.cbt; g.epi = EPI_STORE_F32; g.out = c->logits; g.ldo = 72;
      g.small_max = c->gemm_small_max; g.g8_min = c->gemm8_min; g.gr_min = c->gemmr_min; g.gr_wgs = c->gemmr_wgs;
      // soft-max + depth-to-space in the GEMM's epilogue, at EVERY batch size (one summation order): no logits in memory
      g.epi = EPI_SOFTMAX_D2S; g.out = c->heat; g.d2s_hc = R / 8; g.d2s_wc = R / 8; g.flag = c->sat_flag;
      ProfScope ps(c, ST_HEAD_GEMM, st, 2.0 * cells * 256 * 65, (double)cells * (512 + 256));
      launch_gemm8(c->prec, 256, false, g, st);
    }
    // The descriptor head convDb (1x1, 256 -> 256) is only ever READ at the <= 4 cells each keypoint samples: large batches run it as a
    // gather GEMM over those rows after the top-K (below) — 1600 of 4096 cells per image at 400 keypoints, and 1.6 instead of 4 MB
    // of fp32 written.  The dense map stays for small batches (the batch-1 line path samples junction descriptors from it) and
    // for the inspection hook, which rebuilds it on demand.  Same kernel, same K order: the rows are bit-identical either way.
    sparse_desc = B > 2 && cap * 4 <= (R / 8) * (R / 8) && cap <= 1024;
    if (!sparse_desc && dense_desc_head(c, B, st)) return 1;
    c->desc_dense_valid = !sparse_desc;
    c->last_B = B;
  }
  const int ccap = R * R;
  {
    ProfScope ps(c, ST_NMS, st, 0, (double)B * R * R * 8);
    if (c->cfg.nms_radius == 4) {                        // the reference's radius: simple_nms in registers (kernels_nms512.hip; R = 512)
      // the dense NMS'd map is consumed only by the batch-1 line path (junction scores) and the inspection hook: large batches skip
      // its 1 MB / image write (airfe_debug_detector_maps rebuilds it on demand)
      c->nms_map_valid = B <= 2 || c->force_nms_map;
      launch_nms512_candidates(c->heat, c->nms_map_valid ? c->heat_nms : nullptr, c->nms_mask, B, c->cfg.keypoint_threshold,
                               c->cfg.remove_borders, c->cand, c->cand_cnt, ccap, st);
    } else if (c->cfg.nms_radius > 0) {
      c->nms_map_valid = true;
      launch_simple_nms(c->heat, c->heat_nms, c->nms_tmp, B, R, R, c->cfg.nms_radius, st);
      launch_candidates(c->heat_nms, B, R, R, c->cfg.keypoint_threshold, c->cfg.remove_borders, c->cand, c->cand_cnt, ccap, st);
    } else {
      launch_candidates(c->heat, B, R, R, c->cfg.keypoint_threshold, c->cfg.remove_borders, c->cand, c->cand_cnt, ccap, st);
    }
  }
  const int nhalf = (d_gray1 && !dst_contig) ? 2 : 1;
  const int Bh = nhalf == 2 ? Bs : B;                                    // images per destination
  for (int half = 0; half < nhalf; ++half) {                             // the two feature destinations: one launch each
    const int b0 = half * Bs;
    ProfScope ps(c, ST_SELECT, st, 0, (double)Bh * 8192 * 8);
    launch_select_list(c->cand + (size_t)b0 * ccap, c->cand_cnt + b0, ccap, Bh, R, c->cfg.max_keypoints, cap, half ? d_feat1 : d_feat,
                       half ? d_n1 : d_n, st);
  }
  if (sparse_desc) {
    const int M = B * cap * 4, Mp = (M + 255) / 256 * 256;
    for (int half = 0; half < nhalf; ++half)
      launch_desc_cells(half ? d_feat1 : d_feat, half ? d_n1 : d_n, cap, Bh, half * Bs, R / 8, R / 8, c->desc_idx + (size_t)half * Bs * cap * 4, st);
    if (Mp > M) HIPCHK(c, hipMemsetAsync(c->desc_idx + M, 0, (size_t)(Mp - M) * 4, st));
    GemmArgs g;
    g.X1 = c->aDa; g.ld1 = 256; g.K1 = 256; g.Wp = c->cDb.w; g.bias = c->cDb.b; g.rowidx = c->desc_idx;
    g.M = Mp; g.N = 256; g.cb_total = c->cDb.cbt; g.epi = EPI_STORE_F32; g.out = c->desc; g.ldo = 256;
    ProfScope ps(c, ST_HEAD_GEMM, st, 2.0 * Mp * 256 * 256, (double)Mp * (512 + 1024));
    // large batches: the streaming kernel with gathered rows (kernels_gemmr.hip, GATHER) — the tiled 8-wave kernel ran this HBM-bound shape at 2.3 TB/s; the same bits
    g.gr_wgs = c->gemmr_wgs;
    if (c->desc_gather_stream && Mp >= c->gemmr_min && gemmr_gather_applicable(256, g)) launch_gemmr_gather(c->prec, g, st);
    else launch_gemm8(c->prec, 256, false, g, st);
  }
  for (int half = 0; half < nhalf; ++half) {
    const int b0 = half * Bs;
    ProfScope ps(c, ST_SAMPLE, st, 0, (double)Bh * c->cfg.max_keypoints * (4096 + 1036));
    if (sparse_desc)
      launch_sample_desc(c->desc + (size_t)b0 * cap * 4 * 256, Bh, R / 8, R / 8, half ? d_feat1 : d_feat, half ? d_n1 : d_n, cap, (float)w / (float)R,
                         (float)h / (float)R, 1, st, 1, c->sat_flag);
    else
      launch_sample_desc(c->desc + (size_t)b0 * (R / 8) * (R / 8) * 256, Bh, R / 8, R / 8, half ? d_feat1 : d_feat, half ? d_n1 : d_n, cap,
                         (float)w / (float)R, (float)h / (float)R, c->desc_normalised ? 0 : 1, st, 0, c->sat_flag);
  }
  return launch_status(c);
}

int detect_dev(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, float* d_feat,
               int cap, int* d_n, hipStream_t st) {
  return detect_dev2(c, d_gray, nullptr, B, h, w, stride, img_stride, d_feat, nullptr, cap, d_n, nullptr, st);
}

// PLNet stage-0 LINE branch of images [i0, i0 + nb) of the batch the detector just ran on: fills stage slots 0 .. nb-1 with the Appendix
// A.1 tensors in the contract's own layouts, so that everything downstream (wireframe dedup, stage 1, filters) is the code the golden
// tests pin.  chw: also the contract's CHW loi_features of slot 0 (the inspection hook; the line path samples the head rows directly).
int line_branch_dev(airfe_ctx* c, hipStream_t st, int i0, int nb, bool chw) {
  if (!c->has_s0) return fail(c, "the detector pack carries no line branch (line.* tensors)");
  if (!c->s0_stage) return fail(c, "the line path arena is not allocated (cfg.plnet_s1_pack)");
  if (nb < 1 || nb > c->Lmax || i0 < 0 || i0 + nb > c->Dmax) return fail(c, "line branch: image range outside the arena");
  const int NP = KEEP_CAP, F = 128;
  float* d = c->s0_stage;
  // Two forms of the 1x1 head.  FUSED (fp32 mode, inspection hook; one image): all 145 channels at every pixel -> l_head [128*128][160].
  // SPLIT (everything else): the 17 decoded channels at every pixel, decoded in the same pass (or, airfe_tuning::fuse_dec = 0, -> l_dec [nb][128*128][32]
  // and a decode pass of its own); the 128 LOI channels — read only at the four
  // bilinear taps of the <= 300 junctions — by a gather GEMM over those <= 1200 rows per image once the junctions are known (line_tail_dev):
  // the fused head wrote 1.07 GB of LOI features per 128 images to read 7 % of them.  Same kernel, same K order: the same bits.
  const bool fused = c->prec == 2 || chw;
  if (fused && (nb != 1 || i0 != 0)) return fail(c, "the fused line head (fp32 mode, inspection hook) runs one image at a time");
  c->line_sparse = !fused;
  bool head_done = false;
  if (c->prec == 2) {
    launch_conv3x3_f32(c->f3a, c->f_cL1.w, c->f_cL1.b, c->fL1, 1, F, F, 128, 128, 0, st);
    GemmF32Args g;
    g.X1 = c->fL1; g.ld1 = 128; g.K1 = 128; g.K = 128; g.W = c->f_cLh.w; g.bias = c->f_cLh.b; g.M = F * F; g.N = 145; g.Y = c->l_head; g.ldy = 160;
    launch_gemm_f32(g, st);
  } else {
    // conv3a features (zero-bordered NHWC, still in the arena for the whole batch) -> [nb * 128*128][128]
    if (run_conv(c, c->cL1, c->a3a + (size_t)i0 * (F + 2) * (F + 2) * 128, c->l_feat, nb, F, F, 0, 0, st)) return 1;
    if (!fused && c->fuse_dec) {        // the 17-channel head and its decode in one pass over the line features (kernels_s0.hip)
      ProfScope ps(c, ST_PL_DECODE, st, 2.0 * nb * F * F * 128 * 17, (double)nb * F * F * (256 + 92));
      launch_s0_head_decode(c->prec, c->l_feat, c->cLh_dec.w, c->cLh_dec.b, d + SG_LP, c->l_jloc, nullptr, c->l_joff, c->l_ta8, nb, SG_STRIDE,
                            st);
      head_done = true;
    } else {
      const LinW& hw = fused ? c->cLh : c->cLh_dec;
      GemmArgs g;
      g.X1 = c->l_feat; g.ld1 = 128; g.K1 = 128; g.Wp = hw.w; g.bias = hw.b;
      g.M = nb * F * F; g.N = hw.N; g.cb_total = hw.cbt; g.epi = EPI_STORE_F32; g.out = fused ? c->l_head : c->l_dec; g.ldo = fused ? 160 : 32;
      g.small_max = c->gemm_small_max; g.g8_min = c->gemm8_min; g.gr_min = c->gemmr_min; g.gr_wgs = c->gemmr_wgs;
      if (!fused) g.small_max = 1 << 30;      // one 64-feature block: the no-LDS kernel computes 64 columns per row instead of the tiled kernels' 256
      ProfScope ps(c, ST_HEAD_GEMM, st, 2.0 * nb * F * F * 128 * hw.N, (double)nb * F * F * (256 + 4.0 * g.ldo));
      launch_gemm(c->prec, 128, false, g, st);
    }
  }
  {
  // head rows read once, 49152 proposals + maps written; the j2l match reads them again
  ProfScope ps(c, ST_PL_DECODE, st, 0, (double)nb * (128.0 * 128 * (17 * 4 + 3 * 16 + 11 * 4) + 3.0 * 49152 * (16 + 12)));
  if (head_done) {
    // (lines_pred, jloc, jnms, joff and the pixel-major thin | aux are already there)
  } else if (fused)
    launch_s0_decode(c->l_head, 160, 128, d + SG_LP, c->l_jloc, nullptr, c->l_joff, d + SG_THIN, d + SG_AUX, chw ? c->s0_loi : nullptr, c->l_ta8, nb,
                     SG_STRIDE, st);
  else
    launch_s0_decode(c->l_dec, 32, 0, d + SG_LP, c->l_jloc, nullptr, c->l_joff, d + SG_THIN, d + SG_AUX, nullptr, c->l_ta8, nb, SG_STRIDE, st);
  // get_junctions: top-300 of the 3x3-suppressed junction map (score descending, raster ascending on ties)
  const int ccap = F * F;
  hipStream_t s3 = st;
  launch_candidates_nms3(c->l_jloc, nb, F, F, 1e-30f, c->l_cand, c->l_cand_cnt, ccap, s3);      // non_maximum_suppression (3x3) inside the compaction
  launch_select_list(c->l_cand, c->l_cand_cnt, ccap, nb, F, 300, 320, c->l_sel, c->l_nsel, s3);
  launch_s0_juncs(c->l_sel, c->l_nsel, c->l_joff, d + SG_JUNCS, 300, 320, nb, SG_STRIDE, s3);
  c->wf_counted = launch_s0_j2l(d + SG_LP, d + SG_JUNCS, 300, NP, 10.0f, d + SG_KEEP, d + SG_MIN, d + SG_MAX, c->wf_counts, nb, SG_STRIDE, chw ? 1 : 0, s3);
  }
  return launch_status(c);
}

// Everything behind the stage-0 tensors for stage slots 0 .. nb-1 (= images i0 .. i0+nb-1 of the detector batch): wireframe_matcher,
// stage 1, the line / junction filter (plnet.cpp:272-307, 468-558) and, for the first nj of them, junction_detector + descriptors
// (plnet.cpp:425-448).  LOI features: the head GEMM's rows (loi == nullptr) or a CHW block.  Results go to DEVICE buffers:
// d_lines [nb][capL][4], d_nlines / d_lfound [nb], d_junc [nj][capJ][259], d_njunc / d_jfound [nj] (found > cap = the caller's overflow).
// phase: 1 = the lines (needs nothing of the point branch), 2 = the junctions (score maps, descriptor maps of the point branch), 3 = both.
int line_tail_dev(airfe_ctx* c, int i0, int nb, const float* loi_chw, int h, int w, double* d_lines, int capL, int* d_nlines, int* d_lfound,
                  float* d_junc, int capJ, int* d_njunc, int* d_jfound, int nj, hipStream_t st, int phase) {
  if (!c->has_s1) return fail(c, "PLNet stage-1 weights were not loaded (cfg.plnet_s1_pack)");
  if (nb < 1 || nb > c->Lmax || nj < 0 || nj > nb) return fail(c, "line path: image range outside the arena");
  const int R = AIRFE_INTERNAL_SIZE, NP = KEEP_CAP;
  float* d = c->s0_stage;
  const float ws = (float)w / (float)R, hs = (float)h / (float)R;
  if (phase & 1) {
  hipStream_t s4 = st;
  if (nj > 0) launch_zero16(c->jmap, (size_t)nj * R * R, s4);
  {
  ProfScope ps(c, ST_PL_STAGE1, st, 0, (double)nb * 49152 * 12);
  launch_wireframe(d + SG_KEEP, d + SG_MIN, d + SG_MAX, NP, 300, c->wf_table, c->wf_keep, c->wf_pairs, c->wf_rep, KEEP_CAP, LINE_CAP,
                   c->wf_counts, c->wf_counted, d + SG_JUNCS, d + SG_LP, c->s1_la, c->wf_prop, nb, SG_STRIDE, s4);
  c->wf_counted = false;
  if (loi_chw) {                    // host-supplied contract tensors: all 496 features per line from the CHW blocks
    launch_plnet_s1(d + SG_JUNCS, d + SG_LP, c->wf_keep, c->wf_pairs, c->wf_rep, c->wf_counts, loi_chw, 0, nullptr, nullptr, d + SG_THIN, d + SG_AUX,
                    c->s1_w, c->s1_la, c->s1_sc, KEEP_CAP, LINE_CAP, nb, SG_STRIDE, st);
  } else {
    if (c->line_sparse) {           // the LOI head at the junctions' tap rows only
      const int M = nb * 1200, Mp = (M + 255) / 256 * 256;
      hipStream_t s5 = st;
      launch_s1_junc_rows(d + SG_JUNCS, 300, c->l_ridx, nb, SG_STRIDE, s5);
      if (Mp > M) HIPCHK(c, hipMemsetAsync(c->l_ridx + M, 0, (size_t)(Mp - M) * 4, s5));
      GemmArgs g;
      g.X1 = c->l_feat; g.ld1 = 128; g.K1 = 128; g.Wp = c->cLh_loi.w; g.bias = c->cLh_loi.b; g.rowidx = c->l_ridx;
      g.M = Mp; g.N = 128; g.cb_total = c->cLh_loi.cbt; g.epi = EPI_STORE_F32; g.out = c->l_lrows; g.ldo = 128;
      g.gr_wgs = c->gemmr_wgs;
      // large batches: the streaming kernel with gathered rows (kernels_gemmr.hip, K = N = 128 form) — the tiled kernel ran this HBM-bound shape at 0.8 TB/s; the same bits
      if (c->desc_gather_stream && Mp >= c->gemmr_min && c->prec != 2 && gemmr_gather128_applicable(g)) launch_gemmr_gather128(c->prec, g, s5);
      else launch_gemm8(c->prec, 128, false, g, s5);
      if (c->cfg.line_precision == 3) launch_s1h_junc_proj(d + SG_JUNCS, c->l_lrows, 300, c->s1_wsplit[4], c->s1_wsplit[5], c->s1_jfeat, nb, SG_STRIDE, s5);
      else launch_s1_junc_proj(d + SG_JUNCS, nullptr, 0, 0, c->l_lrows, 300, c->s1_w[0], c->s1_jfeat, nb, SG_STRIDE, s5);
    } else {
      launch_s1_junc_proj(d + SG_JUNCS, c->l_head, (size_t)128 * 128 * 160, 160, nullptr, 300, c->s1_w[0], c->s1_jfeat, nb, SG_STRIDE, st);
    }
    if (c->cfg.line_precision == 3)      // the four dense layers on the 2-byte matrix pipe, operands as fp16 (hi, lo) pairs: fp32-accurate (kernels_ext.hip)
      launch_plnet_s1h(c->wf_pairs, c->wf_counts, c->s1_jfeat, c->l_ta8, c->s1_wsplit, c->s1_w, c->s1_la, c->wf_prop, c->s1_sc, LINE_CAP, nb, st);
    else
      launch_plnet_s1(d + SG_JUNCS, d + SG_LP, c->wf_keep, c->wf_pairs, c->wf_rep, c->wf_counts, nullptr, 0, c->s1_jfeat, c->l_ta8, d + SG_THIN,
                      d + SG_AUX, c->s1_w, c->s1_la, c->s1_sc, KEEP_CAP, LINE_CAP, nb, SG_STRIDE, st);
  }
  }
  ProfScope ps(c, ST_PL_FILTER, st, 0, (double)nb * R * R);
  launch_line_filter(c->s1_la, c->s1_sc, c->wf_counts, c->cfg.remove_borders, c->cfg.line_threshold, c->cfg.line_length_threshold, ws, hs, R,
                     c->jmap, nj, d_lines, capL, d_nlines, d_lfound, LINE_CAP, nb, st);
  }
  if ((phase & 2) && nj > 0) {
    ProfScope ps(c, ST_PL_FILTER, st, 0, (double)nj * R * R * 2);
    if (c->cfg.nms_radius > 0 && !c->nms_map_valid) return fail(c, "line path: the NMS'd score maps of this batch were not kept");
    const float* hsel = (c->cfg.nms_radius > 0 ? c->heat_nms : c->heat) + (size_t)i0 * R * R;
    launch_junction_scan(c->jmap, hsel, R, c->cfg.remove_borders, d_junc, capJ, d_njunc, d_jfound, c->d_njunc + 2 * c->Lmax, nj, st);
    if (!c->desc_dense_valid) return fail(c, "line path: the dense descriptor map of this batch was not made");
    launch_sample_desc(c->desc + (size_t)i0 * (R / 8) * (R / 8) * 256, nj, R / 8, R / 8, d_junc, d_njunc, capJ, ws, hs,
                       c->desc_normalised ? 0 : 1, st);
  }
  return launch_status(c);
}

}  // namespace airfe_host
