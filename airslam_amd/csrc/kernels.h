// airfe — host-callable launch wrappers for every gfx950 kernel of the front end.
// All pointers are device pointers; `prec` selects the 2-byte storage type (0 = bf16, 1 = fp16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace airfe {

enum Epi {
  EPI_STORE = 0,      // bias (+act) -> 2-byte [M][ldo]
  EPI_STORE_F32 = 1,  // bias        -> fp32  [M][ldo], only features < N
  EPI_RESID = 2,      // x32[M][ldo] += bias + acc ; out = 2-byte copy of x32
  EPI_HEADS = 3,      // bias (+rotary) -> 2-byte head-major [S][H][Np][64]; feature/256 selects out/out2
  EPI_HEADS_T = 4,    // (TRANS kernels) bias -> 2-byte [S][H][64][Np]
  EPI_SOFTMAX_D2S = 5 // launch_gemm8 only (its streaming head kernel), K = 256, N = 65: bias, soft-max over the 65 logits of a row (= cell), drop the dustbin, 8x8 depth-to-space ->
                      // fp32 score map [B][8 d2s_hc][8 d2s_wc] (the SuperPoint detector head in the GEMM's epilogue)
};
enum Act { ACT_NONE = 0, ACT_RELU = 1 };

struct GemmArgs {
  const uint16_t* X1 = nullptr;  // rows [M][ld1], first K1 input features
  const uint16_t* X2 = nullptr;  // rows [M][ld2], remaining K-K1 features (cat(x, msg) inputs)
  int ld1 = 0, ld2 = 0, K1 = 0;
  const uint16_t* Wp = nullptr;  // slabs [N/64][K/64][8 KiB]
  const float* bias = nullptr;   // [>= cb_total*64], natural feature order
  int M = 0;                     // rows, multiple of the block's row tile
  int N = 0;                     // valid output features
  int cb_total = 0;              // ceil(N/64)
  int epi = EPI_STORE;
  int act = ACT_NONE;
  void* out = nullptr;
  void* out2 = nullptr;
  int ldo = 0;
  float* x32 = nullptr;
  const float* rot_cos = nullptr;  // [M][32] (rotary on when non-null)
  const float* rot_sin = nullptr;
  int Np = 0, H = 4;
  // kernel choice by row count (measured on MI355X, LightGlue at 400 keypoints per image): M <= small_max -> gemm_small_kernel
  // (no LDS, one round trip), M >= g8_min and M % 256 == 0 -> gemm8_kernel (256x256 tiles want >= 60 row tiles), else gemm_kernel
  int small_max = 4096, g8_min = 16000;
  int gr_min = 8192;             // rows from which the K = 256 linears go to the streaming kernel (where it applies)
  int gr_wgs = 256;              // its persistent workgroups (tests lower it so that a small batch wraps the DMA ring)
  const int* rowidx = nullptr;   // gemm8 only: row r of the GEMM reads X1 row rowidx[r] (a gather; [M] entries)
  int d2s_hc = 0, d2s_wc = 0;    // EPI_SOFTMAX_D2S: cells per image column / row
  int* flag = nullptr;           // EPI_SOFTMAX_D2S: flag[0] = 1 when a cell's logits are not finite — the 2-byte activations upstream left their range
};

// out[M][N] = X[M][K] * W^T ; K in {128,256,512}; trans => EPI_HEADS_T (operand roles swapped)
void launch_gemm(int prec, int K, bool trans, const GemmArgs& a, hipStream_t st);
// M % 256 == 0: 256x128 tile, 8 waves, three-stage LDS-DMA ring (launch_gemm dispatches to it for large M)
void launch_gemm8(int prec, int K, bool trans, const GemmArgs& a, hipStream_t st);
// K = 256, N in {256, 512}, no rotary: persistent streaming kernel with register-resident weights (kernels_gemmr.hip)
bool gemmr_applicable(int K, bool trans, const GemmArgs& a);
void launch_gemmr(int prec, bool trans, const GemmArgs& a, hipStream_t st);
bool gemmr_gather_applicable(int K, const GemmArgs& a);          // y[r] = W x[rowidx[r]] + b, fp32 rows out, K = N = 256: the streaming kernel with gathered source rows
void launch_gemmr_gather(int prec, const GemmArgs& a, hipStream_t st);
bool gemmr_gather128_applicable(const GemmArgs& a);                    // ... and K = N = 128 (the LOI head at the junctions' tap rows)
void launch_gemmr_gather128(int prec, const GemmArgs& a, hipStream_t st);
// head-major q|k linear `a` + transposed-V linear `b` over the same rows in one launch
bool gemmr_pair_applicable(const GemmArgs& a, const GemmArgs& b);
void launch_gemmr_pair(int prec, const GemmArgs& a, const GemmArgs& b, hipStream_t st);

// ---- fused LightGlue post-attention block (kernels_lgblockf.hip): out-proj -> ffn.0 -> LayerNorm -> GELU -> ffn.3 -> residual;
// weights straight from the packed slabs of the separate linears
struct LgBlockFArgs {
  const uint16_t* attn;      // [M][256]
  uint16_t* xb;              // [M][256], updated in place
  float* x32;                // [M][256], updated in place
  const uint16_t *wo, *w1, *w2;                    // LinW::wf (fragment order; LinW::w when !lg_blockf_frag_weights()) of out-proj (256->256), ffn.0 (512->512), ffn.3 (512->256); wo == nullptr: the out-projection is inside w1's message half (fold_out_proj) and `attn` is ffn.0's second operand
  const float *bo, *b1, *gamma, *beta, *b2;
  int M;                     // tokens, multiple of 128
  int tokens_per_wg = 128;   // 128 or 112 (see launch_lg_blockf)
  int mixed = 0, n_cu = 0;   // mixed = 1 (with tokens_per_wg = 112 and the folded out-projection): one round of 7-tile passes on n_cu CUs, then 6-tile passes (kernels_lgblockf.hip)
  int relu = 0;              // 1: ReLU instead of LayerNorm + GELU (the SuperGlue block; gamma / beta unused)
  // the NEXT attention layer's projections, computed from the block's result (nqk_w == nullptr: none).  nqk_n = 512: q | k with rotary
  // (rows 0..255 -> q_out, 256..511 -> k_out), 256: the cross block's shared projection (-> q_out); nv: V, stored transposed.
  // Layouts as EPI_HEADS / EPI_HEADS_T; a ragged last pass stores its surplus rows too (arena slack).
  const uint16_t *nqk_w = nullptr, *nv_w = nullptr;
  const float *nqk_b = nullptr, *nv_b = nullptr, *rot_cos = nullptr, *rot_sin = nullptr;
  int nqk_n = 0, Np = 0, H = 4;
  uint16_t *q_out = nullptr, *k_out = nullptr, *vt_out = nullptr;
};
void launch_lg_blockf(int prec, const LgBlockFArgs& a, hipStream_t st);
bool lg_blockf_frag_weights();     // true: wo / w1 / w2 / nqk_w / nv_w are LinW::wf (fragment order), false: LinW::w (slab image)

struct ConvArgs {
  const uint16_t* X = nullptr;  // [B][H+2][W+2][CIN], zero border
  const uint16_t* Wp = nullptr; // slabs [COUT/64][9][CIN/64][8 KiB]
  const float* bias = nullptr;
  uint16_t* Y = nullptr;        // [B][Ho+2][Wo+2][COUT] (out_pad=1) or [B][Ho][Wo][COUT] (out_pad=0)
  int B = 0, H = 0, W = 0, CIN = 0, COUT = 0;
  int pool = 0;                 // fused 2x2 max-pool
  int out_pad = 1;
  int relu = 1;
  // fused conv1a -> conv1b (launch_conv64r only): fp32 image [B][H+2][W+2], conv1a weights [64][9] and bias [64]
  const float* img = nullptr;
  const float* w1a = nullptr;
  const float* b1a = nullptr;
};
void launch_conv3x3(int prec, const ConvArgs& a, hipStream_t st);
// persistent weight-stationary variant for CIN == COUT == 64 (launch_conv3x3 dispatches to it)
void launch_conv64r(int prec, const ConvArgs& a, hipStream_t st);
void launch_conv128r(int prec, const ConvArgs& a, hipStream_t st);
// persistent tap-streamed variant for CIN == 128, COUT % 128 == 0 (launch_conv3x3 dispatches to it)

// cv::resize(INTER_LINEAR, 8-bit fixed point) + /255 -> fp32 [B][RH+2][RW+2] interior
void launch_preprocess(const uint8_t* src, int B, int h, int w, int stride, size_t img_stride,
                       const int* xtab /*[RW][4]: x0,x1,a0,a1*/, const int* ytab /*[RH][4]*/, const float* lut /*[256]*/,
                       float* out, int RH, int RW, hipStream_t st);

// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) of B 8-bit images through one pair of CV_32FC1 maps [h][w] (Camera::UndistortImage, camera.cc:161-182)
void launch_remap_linear(const uint8_t* src, int B, int h, int w, int stride, size_t img_stride, const float* mapx, const float* mapy,
                         uint8_t* dst, int dstride, size_t dimg_stride, hipStream_t st);

// ---- detector heads -----------------------------------------------------------------------
// logits fp32 [B*HC*WC][ldl] (65 valid) -> heat fp32 [B][HC*8][WC*8]
void launch_softmax_d2s(const float* logits, int ldl, float* heat, int B, int HC, int WC, hipStream_t st);
// in-place channel L2 normalisation of fp32 [rows][256]
void launch_l2norm256(float* d, int rows, hipStream_t st);
// SuperPoint simple_nms(radius) on fp32 [B][H][W]; tmp: 3 maps of the same size
void launch_simple_nms(const float* heat, float* out, float* tmp, int B, int H, int W, int radius, hipStream_t st);
// simple_nms(4) on the 512 x 512 map as three register-resident launches (kernels_nms512.hip): out = NMS'd map (may be nullptr: no dense
// map), and the detect_point candidates (score >= thr inside the border box) are appended to cand [B][cand_cap] (49-bit keys) /
// cand_cnt [B]; planes = 2 x [B][64][64] 64-bit words
void launch_nms512_candidates(const float* heat, float* out, void* planes, int B, float thr, int border, unsigned long long* cand,
                              int* cand_cnt, int cand_cap, hipStream_t st);
// candidates from a finished map (NMS off, or radius != 4 through the multi-pass launch_simple_nms)
void launch_candidates(const float* heat, int B, int H, int W, float thr, int border, unsigned long long* cand,
                       int* cand_cnt, int cand_cap, hipStream_t st);
// the same (border 0) over the 3x3-suppressed map a * (a == max3x3(a)) of `heat`, taken on the fly (PLNet's junction map)
void launch_candidates_nms3(const float* heat, int B, int H, int W, float thr, unsigned long long* cand, int* cand_cnt, int cand_cap, hipStream_t st);
// exact top-K (score desc, raster asc) / raster order when count <= K, on the candidate list
//   feat [B][cap][259] rows: score,x,y written (x,y in 512-space, unscaled); n_out [B]
void launch_select_list(const unsigned long long* cand, const int* cand_cnt, int cand_cap, int B, int W, int topk, int cap,
                        float* feat, int* n_out, hipStream_t st);
// bilinear descriptor sampling + L2 norm (plnet.cpp:369-417), then x,y *= (w_scale,h_scale)
//   desc fp32 [B][HC][WC][256]
// flag (may be NULL): flag[1] = 1 when a sampled descriptor is not finite (2-byte activations upstream left their range)
void launch_sample_desc(const float* desc, int B, int HC, int WC, float* feat, const int* n, int cap,
                        float w_scale, float h_scale, int normalise, hipStream_t st, int compact = 0, int* flag = nullptr);
// idx[(b * cap + k) * 4 + tap] = row (b0 + b) * HC * WC + cell of the dense head input that keypoint k of image b samples (k >= n[b]: 0):
// the row list of the descriptor head's gather GEMM; compact = 1 in launch_sample_desc reads that GEMM's output
void launch_desc_cells(const float* feat, const int* n, int cap, int B, int b0, int HC, int WC, int* idx, hipStream_t st);

// ---- LightGlue ------------------------------------------------------------------------------
struct LgPrepArgs {
  const float* f0; const float* f1;   // [B][cap][ld] feature rows (ld = 259: score,x,y,desc ; ld = 258: x,y,desc)
  const int* n0; const int* n1;       // [B]
  int ld, kp_off;                     // kp_off: column of x
  int normalize;                      // apply PointMatcher::NormalizeKeypoints (point_matcher.cc:39-48)
  float cx, cy, linv;                 // width/2, height/2 (integer division), scale/max(w,h)
  const float* wr;                    // posenc.Wr.weight [32][2]
  int B, cap, Np;
  float* x32; uint16_t* xb;           // [2B][Np][256]
  float* rot_cos; float* rot_sin;     // [2B][Np][32]
  int* lens;                          // [2B]
  // a SECOND pair whose features live elsewhere (B must be 1; it becomes pair 1 of a batch of two: the stereo and the temporal match of one keyframe)
  const float *f0x = nullptr, *f1x = nullptr; const int *n0x = nullptr, *n1x = nullptr;
  int slack_rows = 0;                 // token rows behind the last sequence that are reset to zero as well (the arena's slack, airfe_load.hip: alloc_matcher_arena)
};
void launch_lg_prepare(int prec, const LgPrepArgs& a, hipStream_t st);
// flash attention over head-major Q,K [S][H][Np][64] and Vt [S][H][64][Np] -> O [S][Np][256]; cross => kv sequence s^1; on the 32x32x16
// MFMA, one query per lane (kernels_attn.hip).  q and k arrive PRE-SCALED by sqrt(scale * log2 e) each (folded into their projection
// weights: ATT_QK_FOLD in airfe_host.h)
void launch_attention32(int prec, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, const int* lens,
                        int S, int H, int Np, int cross, hipStream_t st);
// in-place LayerNorm(512, eps) + exact GELU on 2-byte [M][512]
void launch_ln_gelu(int prec, uint16_t* h, const float* gamma, const float* beta, int M, hipStream_t st);
// z[M] = logsigmoid-ready matchability: dot(x32[m], w) + b
void launch_rowdot256(const float* x32, const float* w, float b, float* z, int M, hipStream_t st);
// sim[b][i][j] = md[2b][i] . md[2b+1][j]  (fp32 [B][Np][Np])
void launch_sim(int prec, const uint16_t* md, float* sim, int B, int Np, hipStream_t st);
// LightGlue assignment + filter_matches (light_glue.cpp:214-266) fully on device
//   scores_out (optional) [B][Np][Np] log-assignment; idx [B][cap][2], score [B][cap], nmatch [B]
void launch_lg_assign(const float* sim, const float* z, const int* lens, int B, int Np, int cap, float thr,
                      float* rowlse, float* collse, float* scores_out, int* rowarg, float* rowval, int* colarg,
                      int32_t* idx, float* score, int* nmatch, hipStream_t st);

// filter_matches alone on finished log-assignment matrices [B][Np][Np] (test hook)
// fault hunting (airfe_debug_trace): checksums of `units` units of `unit_words` 32-bit words each; per-slot digests
void launch_trace_hash(const void* p, unsigned unit_words, unsigned units, unsigned long long* out, hipStream_t st);
void launch_trace_digest(const unsigned long long* tab, const unsigned* off, int slots, unsigned long long* dig, hipStream_t st);
// the same assignment WITHOUT the similarity matrix in HBM (kernels_lg.hip "assignment without the similarity matrix"): part / argpart = lg_assign_part_floats()
// floats each; sim_out / scores_out only for the trace and the inspection hooks
size_t lg_assign_part_floats(int B, int Np);
void launch_lg_assign_fused(int prec, const uint16_t* md, const float* z, const int* lens, int B, int Np, int cap, float thr, float* part, float* argpart,
                            float* rowlse, float* collse, float* sim_out, float* scores_out, int* rowarg, float* rowval, int* colarg, int32_t* idx,
                            float* score, int* nmatch, hipStream_t st);
void launch_lg_filter_scores(const float* scores, const int* lens, int B, int Np, int cap, float thr, int* rowarg, float* rowval,
                             int* colarg, int32_t* idx, float* score, int* nmatch, hipStream_t st);

// ---- PLNet line path (src/plnet.cpp:272-307, 468-558) -------------------------------------------------------
// Every launcher takes B images per launch (one per grid row).  iskeep / imin / imax / juncs / lines_pred / thin / aux are IMAGE 0's
// pointers into its stage block, image b's are b * stage_stride floats further; work lists and outputs are dense per image.
constexpr int LINE_CNT_LD = 64;       // ints of `counts` per image: [0] M1 kept proposals, [1] M2 unique lines, [2 .. 50) scratch
constexpr int WF_WGS = 48;            // workgroups that build the raster-ordered list of kept proposals (one contiguous run each)
// table: [B][jn*jn] ints pre-filled with INT_MAX (left clean by the kernel); keep [B][cap], pairs [B][line_cap][2], rep [B][line_cap]
// counted: the per-workgroup counts are already in `counts` (launch_s0_j2l); head4 / prop4 [B][line_cap][4] or both nullptr: per unique line (juncs[max], juncs[min]) = stage 1's lines_adjusted, and lines_pred of its first proposal
void launch_wireframe(const float* iskeep, const float* imin, const float* imax, int n, int jn, int* table, int* keep,
                      int* pairs, int* rep, int cap, int line_cap, int* counts, bool counted, const float* juncs, const float* lines_pred, float* head4,
                      float* prop4, int B, size_t stage_stride, hipStream_t st);
// w: 11 device pointers {W0t[496][128], b0, W2t, b2, W4t, b4, Wrt[240][128], br, Wh[2][128], bh, t[30]}; every transposed table is followed by
// S1_WPAD readable rows (the kernel's weight prefetch runs past the last row)
constexpr int S1_WPAD = 128;
// the device path: proj [B][300][256] = the LOI half of fc2.0 applied per junction (launch_s1_junc_proj: LOI features either sampled from the
// fused head's rows `head` [B][128*128][ps], or combined from the four tap rows lrows [B][300][4][128] of the LOI head's gather GEMM whose row
// list launch_s1_junc_rows writes: ridx [B][300][4], rows of the [B][128*128] line-feature matrix) + ta8 [B][128*128][8] (launch_s0_decode).
// proj == nullptr: the contract's CHW tensors, loi [128][128*128] (+ b * loi_img), thin / aux the stage's planes.
void launch_s1_junc_rows(const float* juncs, int jn, int* ridx, int B, size_t stage_stride, hipStream_t st);
void launch_s1_junc_proj(const float* juncs, const float* head, size_t head_img, int ps, const float* lrows, int jn, const float* w0t,
                         float* proj, int B, size_t stage_stride, hipStream_t st);
void launch_plnet_s1(const float* juncs, const float* lines_pred, const int* keep, const int* pairs, const int* rep,
                     const int* counts, const float* loi, size_t loi_img, const float* proj, const float* ta8, const float* thin,
                     const float* aux, const float* const* w, float* lines_adjusted, float* scores_line, int keep_cap, int line_cap, int B,
                     size_t stage_stride, hipStream_t st);
// the device path of launch_plnet_s1 (proj + ta8) with the dense layers on the 2-byte matrix pipe, operands as fp16 (hi, lo) pairs (cfg.line_precision = 3):
// wsplit[4] = the fp16 (hi, lo) fragments of fc2.0's thin / aux columns, fc2_res.0, fc2.2, fc2.4 (airfe_load.hip); w = the fp32 set (biases, head, sample_t);
// lines_adjusted / prop4 = launch_wireframe's head4 / prop4
void launch_plnet_s1h(const int* pairs, const int* counts, const float* proj, const float* ta8, const uint16_t* const* wsplit, const float* const* w,
                      const float* lines_adjusted, const float* prop4, float* scores_line, int line_cap, int B, hipStream_t st);
// launch_s1_junc_proj's gather form (lrows) on the same pipe: wa / wb = the (hi, lo) planes [2][128][128] of fc2.0's columns 0..127 / 128..255
void launch_s1h_junc_proj(const float* juncs, const float* lrows, int jn, const uint16_t* wa, const uint16_t* wb, float* proj, int B, size_t stage_stride,
                          hipStream_t st);
// la [B][line_cap][4], sc [B][line_cap], lines_out [B][capL][4], nlines [B] (<= capL), nfound [B] or nullptr; jmap [nj][R*R] (zeroed by the
// caller): the first nj images write their junction maps
void launch_line_filter(const float* la, const float* sc, const int* counts, int border, float line_thr, float len_thr,
                        float w_scale, float h_scale, int R, unsigned char* jmap, int nj, double* lines_out, int capL, int* nlines, int* nfound,
                        int line_cap, int B, hipStream_t st);
void launch_zero16(void* p, size_t bytes, hipStream_t st);        // bytes % 16 == 0, 16-byte aligned
// jmap / heat [B][R*R], feat [B][cap][259], n_kept / n_found [B], wg_counts [B][64] scratch
void launch_junction_scan(const unsigned char* jmap, const float* heat, int R, int border, float* feat, int cap, int* n_kept, int* n_found,
                          int* wg_counts, int B, hipStream_t st);

// ---- fp32 correctness path (kernels_f32.hip; cfg.precision = 2): fp32 storage, f32-input MFMA, plain kernels
struct GemmF32Args {
  const float* X1 = nullptr; int ld1 = 0, K1 = 0;     // first K1 input columns
  const float* X2 = nullptr; int ld2 = 0;             // remaining K - K1 columns (cat(x, msg) inputs)
  const float* W = nullptr;                           // [N][K] row-major (PyTorch Linear layout)
  const float* bias = nullptr;                        // [N] or nullptr
  float* Y = nullptr; int ldy = 0;
  int M = 0, N = 0, K = 0;
  float scale = 1.f;                                  // applied to (acc + bias)
  int relu = 0, accumulate = 0;                       // accumulate: Y += result (residual)
};
void launch_conv1a_f32(const float* img, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t st);
// X [B][H+2][W+2][CIN] zero-bordered -> Y [B][H+2 opad][W+2 opad][COUT]; Wt [9][CIN][COUT]; ReLU; W % 16 == 0, COUT % 64 == 0, CIN % 4 == 0
void launch_conv3x3_f32(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int CIN, int COUT, int opad,
                        hipStream_t st);
void launch_maxpool2_f32(const float* X, float* Y, int B, int H, int W, int C, hipStream_t st);
void launch_gemm_f32(const GemmF32Args& a, hipStream_t st);
void launch_rotary_f32(float* qk, int ld, const float* rc, const float* rs, int M, hipStream_t st);
void launch_attention_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, const int* lens, int S, int H,
                          int Np, int cross, float scale, hipStream_t st);
void launch_ln_gelu_f32(float* h, const float* gamma, const float* beta, int M, hipStream_t st);

// ---- PLNet stage-0 line branch (kernels_s0.hip): decode of the fused head GEMM output [128*128][160] fp32
//      (128 LOI channels | md0..2 dis res | jloc0 jloc1 | joffx joffy | thin0..3 | aux0..3) into the Appendix A.1 tensors
// B images per launch: head [B][128*128][160], jloc / jnms [B][128*128], joff [B][2][128*128] dense; lines_pred [3*128*128][4], thin / aux
// CHW [4][128*128] in the image's stage block (+ b * stage_stride floats).  loi != nullptr: CHW [128][128*128] copy of IMAGE 0's LOI channels.
// ta8 != nullptr: [B][128*128][8], thin0..3 | aux0..3 pixel-major (what stage 1 samples on the device path).  head [B][128*128][ld], the 17
// decoded channels from column off (fused 145-channel head: ld 160, off 128; the 17-channel head: ld 32, off 0).
void launch_s0_decode(const float* head, int ld, int off, float* lines_pred, float* jloc, float* jnms, float* joff, float* thin, float* aux,
                      float* loi, float* ta8, int B, size_t stage_stride, hipStream_t st);
// the 17-channel head and its decode in one pass (the batched path): X [B*128*128][128] 2-byte line features, Wp / bias the packed head;
// writes lines_pred (stage), jloc / jnms / joff, ta8 — not the contract's CHW thin / aux planes
void launch_s0_head_decode(int prec, const uint16_t* X, const uint16_t* Wp, const float* bias, float* lines_pred, float* jloc, float* jnms,
                           float* joff, float* ta8, int B, size_t stage_stride, hipStream_t st);
// rows (score, x, y) of the junction top-K, sel [B][sel_cap][259], n_sel [B] -> juncs_pred [jn][2] in the stage block
void launch_s0_juncs(const float* sel, const int* n_sel, const float* joff, float* juncs, int jn, int sel_cap, int B, size_t stage_stride,
                     hipStream_t st);
// HAWP wireframe_matcher: nearest junctions of both endpoints of n proposals -> iskeep, idx_junc_to_end_min / _max (floats).
// exact_all = 0: iskeep exact everywhere, min / max exact where iskeep > 0 (all that plnet.cpp:272-307 reads), by a cell search;
// exact_all = 1: the contract's tensors in full (every proposal against every junction)
// counts != nullptr: [B][LINE_CNT_LD], and the return value says whether the kernel left launch_wireframe's per-workgroup counts there (counted = true)
bool launch_s0_j2l(const float* lines_pred, const float* juncs, int jn, int n, float thr, float* iskeep, float* imin, float* imax, int* counts,
                   int B, size_t stage_stride, int exact_all, hipStream_t st);

// ---- SuperGlue ----------------------------------------------------------------------------------------------
// h128 != nullptr: only the first three layers run (-> h128 [S*Np][128] 2-byte, x = descriptor); the caller adds the last two as GEMMs
// w: 10 device pointers {W0t[3][32], b0, W1t[32][64], b1, W2t[64][128], b2, W3t[128][256], b3, W4t[256][256], b4}
void launch_sg_prepare(int prec, const float* f0, const float* f1, const int* n0, const int* n1, int ld, int normalize,
                       float cx, float cy, float linv, const float* const* w, int B, int cap, int Np, float* x32,
                       uint16_t* xb, int* lens, uint16_t* h128, hipStream_t st);
// u, v: [B][Lz]; Z: [B][Lz][Lz] log-assignment incl. dustbins (rows 0..n0, cols 0..n1)
// counters: B * 16 unsigned of scratch (one 64-byte line per pair) for the fused kernel's per-pair rendezvous; nullptr = the launch-per-half-iteration form
// xch: B * 64 * Lz floats of scratch (the register-resident kernel's per-iteration exchange of column partials); nullptr = streaming kernels only
void launch_sg_sinkhorn(const float* sim, const int* lens, int B, int Np, int Lz, float alpha, int iters, float* u, float* v,
                        float* Z, unsigned* counters /*B x 16 words*/, unsigned* fail_flag /*raised on a rendezvous time-out*/, float* xch,
                        hipStream_t st);
void launch_sg_decode(const float* Z, const int* lens, int B, int Np, int Lz, float thr, int* idx0, float* max0, int* idx1,
                      int32_t* out0, int32_t* out1, float* ms0, float* ms1, hipStream_t st);

// ---- BoW quantisation: TemplatedVocabulary::transform per feature (tree descent by nearest child descriptor); feature i's descriptor
//      starts at feat + i * ld + off; out_word = word id or UINT_MAX when the leaf's weight is <= 0
void launch_bow_transform(const float* feat, int ld, int off, int N, const float* node_desc, const int* first_child,
                          const int* n_children, const int* word_id, const float* weight, unsigned* out_word, float* out_weight, int* out_node /*leaf node per feature, may be nullptr*/,
                          hipStream_t st);

// ---- point <-> line association (AssignPointsToLines, src/line_processor.cc:68-120) as CSR: row_ptr [L+1], entries
//      (point index ascending, distance) per line; counts [L] is scratch
// AssignPointsToLines / MatchLines over B frames (frame pairs) whose lines, features and counts live on the device (kernels_ext.hip)
struct PlAssignArgs {
  const double* lines = nullptr;      // [B][capL][4]
  const int* nlines = nullptr;        // [B]
  const float* feat = nullptr;        // [B][cap][259]
  const int* npts = nullptr;          // [B]
  int capL = 0, cap = 0, capE = 0;
  int* counts = nullptr;              // scratch [B][capL]
  int* row_ptr = nullptr;             // [B][capL + 1]
  int* pt_idx = nullptr;              // [B][capE]
  double* pt_dist = nullptr;          // [B][capE]
  int* total = nullptr;               // [B] entries found (may exceed capE: overflow), or nullptr
};
void launch_assign_points_to_lines(const PlAssignArgs& a, int B, hipStream_t st);
struct MlArgs {
  const int *row_ptr0 = nullptr, *pt_idx0 = nullptr, *nlines0 = nullptr, *npts0 = nullptr;
  const int *row_ptr1 = nullptr, *pt_idx1 = nullptr, *nlines1 = nullptr, *npts1 = nullptr;
  const int* matches = nullptr;       // [B][mcap][2]
  const int* nmatch = nullptr;        // [B]
  int capL = 0, capE = 0, mcap = 0, W = 0, cap = 0;
  int filter_on = 0;                  // Frame::AddRightFeatures' disparity band (src/frame.cc:147-160) on feat0 / feat1 [B][cap][259]
  double min_x_diff = 0, max_x_diff = 0, max_y_diff = 0;
  const float *feat0 = nullptr, *feat1 = nullptr;
  unsigned *bits0 = nullptr, *bits1 = nullptr;      // scratch [B][capL][W]
  int* row_loc = nullptr;                           // scratch [B][capL]
  int* line_matches = nullptr;        // [B][capL]
};
void launch_match_lines(const MlArgs& a, int B, hipStream_t st);

}  // namespace airfe
