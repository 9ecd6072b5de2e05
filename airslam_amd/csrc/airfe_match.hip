// airfe — LightGlue / SuperGlue forwards on device-resident features and the fault-hunting trace (see airfe_host.h)
#include "airfe_host.h"

namespace airfe_host {

// LightGlue forward in fp32 (same call contract as lightglue_dev): q|k|v from ONE [768][256] projection with the rows regrouped
// head-major, rotary, exact soft-max attention, out-projection, FFN (LayerNorm, erf GELU), residual; the assignment tail is the
// shared fp32 code
int lightglue_dev_f32(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int ld, int kp_off,
                      int normalize, int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, float* scores_out, hipStream_t st) {
  const int S = 2 * B, Np = c->Np, M = S * Np;
  LgPrepArgs pa;
  pa.f0 = f0; pa.f1 = f1; pa.n0 = n0; pa.n1 = n1; pa.ld = ld; pa.kp_off = kp_off; pa.normalize = normalize;
  pa.cx = (float)(c->cfg.image_width / 2);
  pa.cy = (float)(c->cfg.image_height / 2);
  pa.linv = (float)(1.0 / std::max(c->cfg.image_width, c->cfg.image_height) * (double)0.5f);
  pa.wr = c->lg_wr; pa.B = B; pa.cap = cap; pa.Np = Np;
  pa.x32 = c->x32; pa.xb = c->xb; pa.rot_cos = c->rot_cos; pa.rot_sin = c->rot_sin; pa.lens = c->lens;
  pa.slack_rows = (int)(c->arena_rows - (size_t)M);      // the slack rows go back to zero with the same launch (see reset_slack_rows)
  launch_lg_prepare(1, pa, st);
  auto lin = [&](const airfe_ctx::F32Lin& w, const float* x1, int ld1, int K1, const float* x2, int ld2, float* y, int ldy, int acc, float scale = 1.f) {
    GemmF32Args g;
    g.X1 = x1; g.ld1 = ld1; g.K1 = K1; g.X2 = x2; g.ld2 = ld2; g.W = w.w; g.bias = w.b; g.Y = y; g.ldy = ldy;
    g.M = M; g.N = w.N; g.K = w.K; g.accumulate = acc; g.scale = scale;
    launch_gemm_f32(g, st);
  };
  auto ffn = [&](const airfe_ctx::F32Lin& f0w, const float* g, const float* b, const airfe_ctx::F32Lin& f3w) {
    lin(f0w, c->x32, 256, 256, c->m_msg, 256, c->m_h, 512, 0);
    launch_ln_gelu_f32(c->m_h, g, b, M, st);
    lin(f3w, c->m_h, 512, 512, nullptr, 0, c->x32, 256, 1);
  };
  for (const auto& l : c->f_lg) {
    lin(l.qkv, c->x32, 256, 256, nullptr, 0, c->m_qkv, 768, 0);
    launch_rotary_f32(c->m_qkv, 768, c->rot_cos, c->rot_sin, M, st);
    launch_attention_f32(c->m_qkv, 768, c->m_qkv + 256, 768, c->m_qkv + 512, 768, c->m_ctx, c->lens, S, 4, Np, 0, 0.125f, st);
    lin(l.out, c->m_ctx, 256, 256, nullptr, 0, c->m_msg, 256, 0);
    ffn(l.ffn0, l.ln_g, l.ln_b, l.ffn3);
    lin(l.cqk, c->x32, 256, 256, nullptr, 0, c->m_qkv, 768, 0);
    lin(l.cv, c->x32, 256, 256, nullptr, 0, c->m_qkv + 512, 768, 0);
    launch_attention_f32(c->m_qkv, 768, c->m_qkv, 768, c->m_qkv + 512, 768, c->m_ctx, c->lens, S, 4, Np, 1, 0.125f, st);
    lin(l.cout, c->m_ctx, 256, 256, nullptr, 0, c->m_msg, 256, 0);
    ffn(l.cffn0, l.cln_g, l.cln_b, l.cffn3);
  }
  lin(c->f_lgfinal, c->x32, 256, 256, nullptr, 0, c->m_md, 256, 0, 0.25f);      // d^-1/4 on both sides, d = 256
  launch_rowdot256(c->x32, c->lg_mw, c->lg_mb, c->zbuf, M, st);
  for (int b = 0; b < B; ++b) {                                                   // sim[b] = md[2b] . md[2b+1]^T
    GemmF32Args g;
    g.X1 = c->m_md + (size_t)(2 * b) * Np * 256; g.ld1 = 256; g.K1 = 256; g.K = 256; g.W = c->m_md + (size_t)(2 * b + 1) * Np * 256;
    g.Y = c->simbuf + (size_t)b * Np * Np; g.ldy = Np; g.M = Np; g.N = Np;
    launch_gemm_f32(g, st);
  }
  launch_lg_assign(c->simbuf, c->zbuf, c->lens, B, Np, mcap, 0.1f, c->rowlse, c->collse, scores_out, c->rowarg, c->rowval, c->colarg, d_idx,
                   d_score, d_nmatch, st);
  return launch_status(c);
}

// SuperGlue forward in fp32 (same call contract as superglue_dev; cfg.matcher_precision = 2): keypoint encoder as fp32 FMA loops (the prepare kernel's unsplit
// form), per GNN layer q | k | v from ONE [768][256] projection with head-major rows, exact soft-max attention (self / cross alternate: names = ['self', 'cross'] * 9),
// merge, mlp.0 on cat(x, message) + ReLU (BatchNorm is folded into mlp.0 in the pack), mlp.3 added to x; final_proj with d^-1/4 on both sides; the optimal-transport
// tail (Sinkhorn in log space, decode) is the shared fp32 code.
static int superglue_dev_f32(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int normalize, hipStream_t st) {
  const int S = 2 * B, Np = c->Np, M = S * Np;
  const float cx = (float)(c->cfg.image_width / 2), cy = (float)(c->cfg.image_height / 2);
  const float linv = (float)(1.0 / std::max(c->cfg.image_width, c->cfg.image_height) * (double)0.7f);   // point_matcher.cc:58
  reset_slack_rows(c, M, st);
  launch_sg_prepare(1, f0, f1, n0, n1, AIRFE_FEAT_DIM, normalize, cx, cy, linv, c->sg_kenc, B, cap, Np, c->x32, c->xb, c->lens, nullptr, st);
  auto lin = [&](const airfe_ctx::F32Lin& w, const float* x1, int ld1, int K1, const float* x2, int ld2, float* y, int ldy, int acc, int relu, float scale = 1.f) {
    GemmF32Args g;
    g.X1 = x1; g.ld1 = ld1; g.K1 = K1; g.X2 = x2; g.ld2 = ld2; g.W = w.w; g.bias = w.b; g.Y = y; g.ldy = ldy;
    g.M = M; g.N = w.N; g.K = w.K; g.accumulate = acc; g.relu = relu; g.scale = scale;
    launch_gemm_f32(g, st);
  };
  int li = 0;
  for (const auto& l : c->f_sg) {
    const int cross = li & 1;
    ++li;
    lin(l.qkv, c->x32, 256, 256, nullptr, 0, c->m_qkv, 768, 0, 0);
    launch_attention_f32(c->m_qkv, 768, c->m_qkv + 256, 768, c->m_qkv + 512, 768, c->m_ctx, c->lens, S, 4, Np, cross, 0.125f, st);
    lin(l.merge, c->m_ctx, 256, 256, nullptr, 0, c->m_msg, 256, 0, 0);
    lin(l.mlp0, c->x32, 256, 256, c->m_msg, 256, c->m_h, 512, 0, 1);
    lin(l.mlp3, c->m_h, 512, 512, nullptr, 0, c->x32, 256, 1, 0);
  }
  lin(c->f_sgfinal, c->x32, 256, 256, nullptr, 0, c->m_md, 256, 0, 0, 0.25f);      // scores / 256^.5 split over both sides
  for (int b = 0; b < B; ++b) {                                                      // sim[b] = md[2b] . md[2b+1]^T
    GemmF32Args g;
    g.X1 = c->m_md + (size_t)(2 * b) * Np * 256; g.ld1 = 256; g.K1 = 256; g.K = 256; g.W = c->m_md + (size_t)(2 * b + 1) * Np * 256;
    g.Y = c->simbuf + (size_t)b * Np * Np; g.ldy = Np; g.M = Np; g.N = Np;
    launch_gemm_f32(g, st);
  }
  launch_sg_sinkhorn(c->simbuf, c->lens, B, Np, c->Lz, c->sg_alpha, c->cfg.sinkhorn_iters, c->sg_u, c->sg_v, c->sg_Z, c->sg_cnt, c->sg_cnt + (size_t)c->Pmax * 16, c->sg_xch, st);
  launch_sg_decode(c->sg_Z, c->lens, B, Np, c->Lz, 0.2f, c->sg_idx0, c->sg_max0, c->sg_idx1, c->sg_out0, c->sg_out1, c->sg_ms0, c->sg_ms1, st);
  return launch_status(c);
}

// airfe_debug_trace: checksum `words` 32-bit words of p in units of unit_words (slot = one call; no-op unless tracing)
void trace(airfe_ctx* c, hipStream_t st, const char* what, size_t li, const char* blk, const void* p, size_t words, unsigned unit_words) {
  if (!c->trace_on) return;
  const unsigned off = c->trace_slots.empty() ? 0u : c->trace_slots.back().off + c->trace_slots.back().units;
  const unsigned units = (unsigned)(words / unit_words);
  if (units == 0) return;                                          // nothing to hash (and a 0-sized grid is a launch error)
  if (c->trace_slots.size() >= 1024 || (size_t)off + units > c->trace_cap) { c->trace_overflow = true; return; }   // reported by trace_finish
  launch_trace_hash(p, unit_words, units, c->trace_tab + off, st);
  c->trace_slots.push_back({std::string("L") + std::to_string(li) + "." + blk + "." + what, off, units, unit_words, p, words});
  if ((int)c->trace_slots.size() - 1 == c->trace_stop) c->trace_halt = true;
}
int trace_finish(airfe_ctx* c, hipStream_t st) {
  if (c->trace_on && c->trace_overflow) {
    c->trace_overflow = false;
    return fail(c, "airfe_debug_trace: slot / unit table overflow — slots were dropped, slot indices do not name the launches of a full run");
  }
  if (c->trace_on && !c->trace_slots.empty()) {
    c->trace_off_h.clear();
    for (const auto& t : c->trace_slots) c->trace_off_h.push_back(t.off);
    c->trace_off_h.push_back(c->trace_slots.back().off + c->trace_slots.back().units);
    HIPCHK(c, hipMemcpyAsync(c->trace_off, c->trace_off_h.data(), c->trace_off_h.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    launch_trace_digest(c->trace_tab, c->trace_off, (int)c->trace_slots.size(), c->trace_dig, st);
  }
  return launch_status(c);
}
#define TRACE_HALT do { if (c->trace_halt) return trace_finish(c, st); } while (0)

// (the matrix launches below return the launch status of their stage under cfg.check_launches, 0 otherwise: lightglue_dev / superglue_dev look once more at their end)
#define STAGE_RC(c) ((c)->cfg.check_launches ? launch_status(c) : 0)
#define RUN(x) do { if (x) return 1; } while (0)

int run_linear(airfe_ctx* c, const LinW& w, const uint16_t* x1, int ld1, int K1, const uint16_t* x2, int ld2, int M,
               int epi, int act, void* out, int ldo, hipStream_t st, bool trans, void* out2,
               float* x32, const float* rc, const float* rs) {
  GemmArgs g;
  g.X1 = x1; g.ld1 = ld1; g.K1 = K1; g.X2 = x2; g.ld2 = ld2;
  g.Wp = w.w; g.bias = w.b; g.M = M; g.N = w.N; g.cb_total = w.cbt;
  g.epi = epi; g.act = act; g.out = out; g.out2 = out2; g.ldo = ldo; g.x32 = x32;
  g.rot_cos = rc; g.rot_sin = rs; g.Np = c->Np; g.H = 4;
  g.small_max = c->gemm_small_max; g.g8_min = c->gemm8_min; g.gr_min = c->gemmr_min; g.gr_wgs = c->gemmr_wgs;
  { ProfScope ps(c, ST_LG_GEMM, st, 2.0 * M * w.K * w.N, (double)M * (w.K + w.N) * 2 + (double)w.K * w.N * 2); launch_gemm(c->mprec, w.K, trans, g, st); }
  return STAGE_RC(c);
}

void run_attention(airfe_ctx* c, int prec, const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, const int* lens, int S,
                   int H, int Np, int cross, float scale, hipStream_t st) {
  // `scale` (1/sqrt(d_head)) and log2 e are already inside q and k (ATT_QK_FOLD), so the kernels exponentiate the raw products:
  // the round-1 kernel is told scale * log2 e = 1
  (void)scale;
  launch_attention32(prec, Q, K, Vt, O, lens, S, H, Np, cross, st);
}

// The attention inputs of one layer: head-major q|k (`qk`, rotary when rc != nullptr; q -> qout, k -> kout, or both roles in qout
// for the cross block's shared projection) and transposed V (`v`).  One streaming launch where kernels_gemmr.hip applies (large
// token counts), else the two linears separately — same arithmetic either way.
int run_qkv(airfe_ctx* c, const LinW& qk, const LinW& v, int M, void* qout, void* kout, const float* rc, const float* rs, hipStream_t st) {
  GemmArgs a, b;
  a.X1 = c->xb; a.ld1 = 256; a.K1 = 256; a.Wp = qk.w; a.bias = qk.b; a.M = M; a.N = qk.N; a.cb_total = qk.cbt;
  a.epi = EPI_HEADS; a.out = qout; a.out2 = kout; a.rot_cos = rc; a.rot_sin = rs; a.Np = c->Np; a.H = 4;
  b.X1 = c->xb; b.ld1 = 256; b.K1 = 256; b.Wp = v.w; b.bias = v.b; b.M = M; b.N = v.N; b.cb_total = v.cbt;
  b.epi = EPI_HEADS_T; b.out = c->vtb; b.Np = c->Np; b.H = 4;
  a.gr_wgs = b.gr_wgs = c->gemmr_wgs;
  if (c->qkv_pair && M >= c->gemmr_min && qk.K == 256 && v.K == 256 && gemmr_pair_applicable(a, b)) {
    { ProfScope ps(c, ST_LG_GEMM, st, 2.0 * M * 256.0 * (qk.N + v.N), (double)M * (256 + qk.N + v.N) * 2 + 256.0 * (qk.N + v.N) * 2); launch_gemmr_pair(c->mprec, a, b, st); }
    return STAGE_RC(c);
  }
  RUN(run_linear(c, qk, c->xb, 256, 256, nullptr, 0, M, EPI_HEADS, ACT_NONE, qout, 0, st, false, kout, nullptr, rc, rs));
  return run_linear(c, v, c->xb, 256, 256, nullptr, 0, M, EPI_HEADS_T, ACT_NONE, c->vtb, 0, st, true);
}

// out-proj + FFN + residual of one block as ONE kernel (kernels_lgblockf.hip); flops/bytes are the algorithmic ones
int lg_blockf(airfe_ctx* c, const LinW& out, const LinW& f0, const float* g, const float* b, const LinW& f3, int M, hipStream_t st, int relu = 0,
               const LinW* nqk = nullptr, const LinW* nv = nullptr, bool rotary = false) {
  LgBlockFArgs a;
  a.relu = relu;
  const bool fr = lg_blockf_frag_weights();                   // the block kernel reads its A fragments from global memory: weights in fragment order (LinW::wf)
  a.attn = c->ob; a.xb = c->xb; a.x32 = c->x32; a.wo = fr ? out.wf : out.w; a.w1 = fr ? f0.wf : f0.w; a.w2 = fr ? f3.wf : f3.w;
  a.bo = out.b; a.b1 = f0.b; a.gamma = g; a.beta = b; a.b2 = f3.b; a.M = M;
  if (c->fold_out) { a.wo = nullptr; a.bo = nullptr; }       // the out-projection lives inside f0 (airfe_load.hip, make_ffn0_folded): the kernel's ffn.0 reads cat(x, attn)
  // one workgroup per CU and pass: ceil(M / T) workgroups run in rounds of 256, a round lasts ~T — take the T with the smaller product
  a.tokens_per_wg = ((M + 111) / 112 + 255) / 256 * 112 < ((M + 127) / 128 + 255) / 256 * 128 ? 112 : 128;
  // small token counts (the batch-1 calls of the SLAM loop: 800 tokens): 112-token passes would occupy 8 of the 256 CUs — 32- / 64-token passes
  // spread the same rows over 4x / 2x as many workgroups as long as that is still ONE round
  if (M <= 256 * 32) a.tokens_per_wg = 32;
  else if (M <= 256 * 64) a.tokens_per_wg = 64;
  if (c->lgb_tokens > 0) a.tokens_per_wg = c->lgb_tokens;          // airfe_tuning::lgb_tokens (measurement switch)
  a.mixed = c->lgb_tokens <= 0;                                     // the library's own choice: a second round of 6-tile passes where that balances the CUs (lgb_tokens = 112 forces uniform passes)
  a.n_cu = c->n_cu;
  // (algorithmic = what this context's packed network asks for: with the out-projection folded into ffn.0 its 2 * 256 * 256 FLOPs per token do not exist)
  double fl = 2.0 * M * ((c->fold_out ? 0.0 : 256.0 * 256) + 512.0 * 512 + 512.0 * 256), by = (double)M * (512 + 512 + 1024 + 512 + 1024) + (c->fold_out ? 786432.0 : 917504.0);
  if (nqk && nv) {            // the next attention layer's projections ride along (kernels_lgblockf.hip, FOLD)
    a.nqk_w = fr ? nqk->wf : nqk->w; a.nqk_b = nqk->b; a.nqk_n = nqk->N; a.nv_w = fr ? nv->wf : nv->w; a.nv_b = nv->b;
    a.rot_cos = rotary ? c->rot_cos : nullptr; a.rot_sin = rotary ? c->rot_sin : nullptr;
    a.q_out = c->qb; a.k_out = c->kb; a.vt_out = c->vtb; a.Np = c->Np; a.H = 4;
    fl += 2.0 * M * 256.0 * (nqk->N + nv->N);
    by += (double)M * (nqk->N + nv->N) * 2 + 256.0 * (nqk->N + nv->N) * 2;
  }
  { ProfScope ps(c, ST_LG_GEMM, st, fl, by); launch_lg_blockf(c->mprec, a, st); }
  return STAGE_RC(c);
}

int lg_ffn(airfe_ctx* c, const LinW& f0, const float* g, const float* b, const LinW& f3, int M, hipStream_t st) {
  RUN(run_linear(c, f0, c->xb, 256, 256, c->fold_out ? c->ob : c->msg, 256, M, EPI_STORE, ACT_NONE, c->hb, 512, st));      // fold_out: cat(x, attention output)
  { ProfScope ps(c, ST_LG_LNGELU, st, 0, (double)M * 2048); launch_ln_gelu(c->mprec, c->hb, g, b, M, st); }
  return run_linear(c, f3, c->hb, 512, 512, nullptr, 0, M, EPI_RESID, ACT_NONE, c->xb, 256, st, false, nullptr, c->x32);
}

// The surplus rows behind the last real token (alloc_matcher_arena's slack) go through every block like real ones: their residual
// stream would keep growing from step to step (x += f(x), never re-initialised) until the 2-byte shadow overflows — and the last
// sequence's final key tile multiplies those rows' V by probability 0, which is NaN once they are not finite.  Back to zero per call.
void reset_slack_rows(airfe_ctx* c, int M, hipStream_t st) {
  if ((size_t)M >= c->arena_rows) return;
  launch_zero16(c->x32 + (size_t)M * 256, (c->arena_rows - (size_t)M) * 256 * sizeof(float), st);
  launch_zero16(c->xb + (size_t)M * 256, (c->arena_rows - (size_t)M) * 256 * sizeof(uint16_t), st);
}

// LightGlue forward on B pairs whose feature rows live on the device
int lightglue_dev(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int ld,
                  int kp_off, int normalize, int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, float* scores_out,
                  hipStream_t st, const LgSecondPair* x2) {
  if (!c->has_lg) return fail(c, "LightGlue weights were not loaded (cfg.lightglue_pack)");
  if (B < 1 || B > c->Pmax) return fail(c, "pair batch exceeds cfg.max_batch / 2");
  if (cap > c->Np) return fail(c, "feature capacity exceeds the matcher arena (max_keypoints)");
  // (the second pair was looked at only behind the fp32 dispatch, which dropped it silently)
  if (x2 && (B != 1 || c->Pmax < 2 || c->mprec == 2)) return fail(c, "lightglue: a second pair needs B = 1, max_batch >= 2 and fp16 / bf16 (matcher_precision)");
  if (c->mprec == 2) return lightglue_dev_f32(c, f0, n0, f1, n1, B, cap, ld, kp_off, normalize, d_idx, d_score, mcap, d_nmatch, scores_out, st);
  LgPrepArgs pa;
  if (x2) {                                      // the stereo and the temporal pair of one keyframe as a batch of two
    pa.f0x = x2->f0; pa.f1x = x2->f1; pa.n0x = x2->n0; pa.n1x = x2->n1;
    B = 2;
  }
  const int S = 2 * B, Np = c->Np, M = S * Np;
  const int Mg = (M + 127) / 128 * 128;          // rows the matrix kernels run over (surplus rows: arena slack, see alloc_matcher_arena)
  pa.f0 = f0; pa.f1 = f1; pa.n0 = n0; pa.n1 = n1; pa.ld = ld; pa.kp_off = kp_off; pa.normalize = normalize;
  // PointMatcher::NormalizeKeypoints (src/point_matcher.cc:39-48): integer width/2, L_inv = 1.0/max(w,h)*scale
  pa.cx = (float)(c->cfg.image_width / 2);
  pa.cy = (float)(c->cfg.image_height / 2);
  pa.linv = (float)(1.0 / std::max(c->cfg.image_width, c->cfg.image_height) * (double)0.5f);
  pa.wr = c->lg_wr; pa.B = x2 ? 1 : B; pa.cap = cap; pa.Np = Np;
  pa.x32 = c->x32; pa.xb = c->xb; pa.rot_cos = c->rot_cos; pa.rot_sin = c->rot_sin; pa.lens = c->lens;
  // the arena's slack rows go back to zero with the same launch (see reset_slack_rows; ADVICE r03: this line had moved to the fp32
  // path only, so that the 2-byte path's slack rows kept their running residual from call to call)
  // (only the rows a kernel of THIS call can touch: the 112- / 128-row rounding of the matrix kernels + one key tile)
  pa.slack_rows = (int)std::min(c->arena_rows - (size_t)M, (size_t)512);
  if (c->trace_on) { c->trace_slots.clear(); c->trace_overflow = false; }
  c->trace_halt = false;
  const size_t Mw = (size_t)M * 128;                     // 32-bit words of a [M][256] 2-byte buffer
  auto tr_x = [&](size_t li, const char* blk) {
    trace(c, st, "x32", li, blk, c->x32, (size_t)M * 256, 4096);
    trace(c, st, "xb", li, blk, c->xb, Mw, 2048);
  };
  auto tr_qkv = [&](size_t li, const char* blk, bool k) {
    trace(c, st, "q", li, blk, c->qb, Mw, 512);
    if (k) trace(c, st, "k", li, blk, c->kb, Mw, 512);
    trace(c, st, "vt", li, blk, c->vtb, Mw, (unsigned)Np / 2);
  };
  { ProfScope ps(c, ST_LG_PREPARE, st, 0, (double)M * (1036 + 1536 + 256)); launch_lg_prepare(c->mprec, pa, st); }
  tr_x(0, "prep");
  // the arena's slack rows as this call starts (must be zero: ADVICE r03 / test_slack_rows_are_reset_on_every_call) and, at the end, as it leaves them
  const size_t slack_words = (size_t)(pa.slack_rows / 16 * 16) * 256;
  if (slack_words) trace(c, st, "x32slack", 0, "prep", c->x32 + (size_t)M * 256, slack_words, 4096);
  trace(c, st, "rc", 0, "prep", c->rot_cos, (size_t)M * 32, 512);
  trace(c, st, "rs", 0, "prep", c->rot_sin, (size_t)M * 32, 512);
  TRACE_HALT;
  // The fused block (kernels_lgblockf.hip) streams 0.9 MB of weights per workgroup whatever the batch: with 112- / 128-token passes only, the four
  // separate launches were quicker below 3200 tokens (profiles/r01d_small_batch_sweeps.txt); with 32- / 64-token passes for small token counts
  // (lg_blockf() picks them) the fused form wins everywhere (profiles/r04_lg_small_batch_sweep.txt) and block_min is 0.
  const bool fused_block = c->fuse_lg_block == 1 || (c->fuse_lg_block < 0 && Mg >= c->block_min);
  // With the fused block the projections of the NEXT attention layer are computed inside it (FOLD): only the very first q | k | v
  // projection is a launch of its own.
  const bool fold = fused_block && c->fold_qkv;
  const bool fold_c = fold, fold_s = fold;
  for (size_t li = 0; li < c->lg.size(); ++li) {
    const LgLayer& l = c->lg[li];
    const LgLayer* nl = li + 1 < c->lg.size() ? &c->lg[li + 1] : nullptr;
    // ---- self block
    if (!fold_s || li == 0) { RUN(run_qkv(c, l.qk, l.v, Mg, c->qb, c->kb, c->rot_cos, c->rot_sin, st)); tr_qkv(li, "self.qkv", true); }
    TRACE_HALT;
    { ProfScope ps(c, ST_LG_ATTENTION, st, 4.0 * S * Np * (double)Np * 256, (double)M * 2048); run_attention(c, c->mprec, c->qb, c->kb, c->vtb, c->ob, c->lens, S, 4, Np, 0, 0.125f, st); }
    RUN(STAGE_RC(c));
    trace(c, st, "o", li, "self.attn", c->ob, Mw, 2048);
    TRACE_HALT;
    if (fused_block) {
      RUN(lg_blockf(c, l.out, l.ffn0, l.ln_g, l.ln_b, l.ffn3, Mg, st, 0, fold_c ? &l.cqk : nullptr, fold_c ? &l.cv : nullptr, false));
      tr_x(li, "self.block");
      TRACE_HALT;
      if (fold_c) tr_qkv(li, "self.block", false);
      TRACE_HALT;
    } else {
      if (!c->fold_out) {
        RUN(run_linear(c, l.out, c->ob, 256, 256, nullptr, 0, Mg, EPI_STORE, ACT_NONE, c->msg, 256, st));
        trace(c, st, "msg", li, "self.out", c->msg, Mw, 2048);
        TRACE_HALT;
      }
      RUN(lg_ffn(c, l.ffn0, l.ln_g, l.ln_b, l.ffn3, Mg, st));
      tr_x(li, "self.ffn");
      TRACE_HALT;
    }
    // ---- cross block (one shared projection for q and k; the two sides swap roles)
    if (!fold_c) { RUN(run_qkv(c, l.cqk, l.cv, Mg, c->qb, nullptr, nullptr, nullptr, st)); tr_qkv(li, "cross.qkv", false); }
    TRACE_HALT;
    { ProfScope ps(c, ST_LG_ATTENTION, st, 4.0 * S * Np * (double)Np * 256, (double)M * 2048); run_attention(c, c->mprec, c->qb, c->qb, c->vtb, c->ob, c->lens, S, 4, Np, 1, 0.125f, st); }
    RUN(STAGE_RC(c));
    trace(c, st, "o", li, "cross.attn", c->ob, Mw, 2048);
    TRACE_HALT;
    if (fused_block) {
      const bool fn = fold_s && nl;
      RUN(lg_blockf(c, l.cout, l.cffn0, l.cln_g, l.cln_b, l.cffn3, Mg, st, 0, fn ? &nl->qk : nullptr, fn ? &nl->v : nullptr, true));
      tr_x(li, "cross.block");
      TRACE_HALT;
      if (fn) tr_qkv(li, "cross.block", true);
      TRACE_HALT;
    } else {
      if (!c->fold_out) {
        RUN(run_linear(c, l.cout, c->ob, 256, 256, nullptr, 0, Mg, EPI_STORE, ACT_NONE, c->msg, 256, st));
        trace(c, st, "msg", li, "cross.out", c->msg, Mw, 2048);
        TRACE_HALT;
      }
      RUN(lg_ffn(c, l.cffn0, l.cln_g, l.cln_b, l.cffn3, Mg, st));
      tr_x(li, "cross.ffn");
      TRACE_HALT;
    }
  }
  const size_t LF = c->lg.size();
  RUN(run_linear(c, c->lg_final, c->xb, 256, 256, nullptr, 0, Mg, EPI_STORE, ACT_NONE, c->mdb, 256, st));
  trace(c, st, "md", LF, "final", c->mdb, Mw, 2048);
  if (slack_words) trace(c, st, "x32slack", LF, "final", c->x32 + (size_t)M * 256, slack_words, 4096);
  TRACE_HALT;
  // the assignment tail in a scope of its own: its stage's launch status is noted before trace_finish looks at it
  const bool fused_assign = c->assign_fused == 1;      // (measured both ways on one box: profiles/r05_assign_ab.txt; ONE form for every batch size: a pair's bits do not depend on the batch)
  const int tail = [&]() -> int {
    // algorithmic: the similarity product (fused form: twice) and, round-2 form, the matrix written once and read four times / fused form, the descriptors + partials
    ProfScope ps(c, ST_LG_ASSIGN, st, (fused_assign ? 4.0 : 2.0) * B * Np * (double)Np * 256,
                 fused_assign ? (double)B * Np * (2.0 * 512 + 4.0 * 8 * ((Np + 63) / 64)) : (double)B * Np * Np * 4 * 6);
#define TAIL_HALT do { if (c->trace_halt) return 2; } while (0)
    launch_rowdot256(c->x32, c->lg_mw, c->lg_mb, c->zbuf, M, st);
    trace(c, st, "z", LF, "final", c->zbuf, (size_t)M, 16);
    TAIL_HALT;
    if (fused_assign) {           // no similarity matrix in HBM: log-sum-exp and arg-max partials are taken in the similarity tiles (kernels_lg.hip)
      launch_lg_assign_fused(c->mprec, c->mdb, c->zbuf, c->lens, B, Np, mcap, 0.1f, c->lg_part, c->lg_argpart, c->rowlse, c->collse,
                             c->trace_on ? c->simbuf : nullptr, scores_out, c->rowarg, c->rowval, c->colarg, d_idx, d_score, d_nmatch, st);
      trace(c, st, "sim", LF, "final", c->simbuf, (size_t)B * Np * Np, 16u * (unsigned)Np);
      TAIL_HALT;
    } else {
      launch_sim(c->mprec, c->mdb, c->simbuf, B, Np, st);
      trace(c, st, "sim", LF, "final", c->simbuf, (size_t)B * Np * Np, 16u * (unsigned)Np);
      TAIL_HALT;
      launch_lg_assign(c->simbuf, c->zbuf, c->lens, B, Np, mcap, 0.1f, c->rowlse, c->collse, scores_out, c->rowarg, c->rowval,
                       c->colarg, d_idx, d_score, d_nmatch, st);
    }
    trace(c, st, "rowlse", LF, "assign", c->rowlse, (size_t)B * Np, (unsigned)Np);
    TAIL_HALT;
    trace(c, st, "collse", LF, "assign", c->collse, (size_t)B * Np, (unsigned)Np);
    TAIL_HALT;
    trace(c, st, "rowval", LF, "assign", c->rowval, (size_t)B * Np, (unsigned)Np);
    TAIL_HALT;
    trace(c, st, "rowarg", LF, "assign", c->rowarg, (size_t)B * Np, (unsigned)Np);
    TAIL_HALT;
    trace(c, st, "colarg", LF, "assign", c->colarg, (size_t)B * Np, (unsigned)Np);
    return 0;
#undef TAIL_HALT
  }();
  (void)tail;
  return trace_finish(c, st);
}

// SuperGlue forward on B pairs of device feature matrices (259-float rows) -> decode outputs [B][Lz]
int superglue_dev(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int normalize,
                  hipStream_t st) {
  if (!c->has_sg) return fail(c, "SuperGlue weights were not loaded (cfg.superglue_pack)");
  if (B < 1 || B > c->Pmax) return fail(c, "pair batch exceeds cfg.max_batch");
  if (cap > c->Np) return fail(c, "feature capacity exceeds the matcher arena (max_keypoints)");
  if (c->mprec == 2) return superglue_dev_f32(c, f0, n0, f1, n1, B, cap, normalize, st);
  const int S = 2 * B, Np = c->Np, M = S * Np;
  const int Mg = (M + 127) / 128 * 128;
  const float cx = (float)(c->cfg.image_width / 2), cy = (float)(c->cfg.image_height / 2);
  const float linv = (float)(1.0 / std::max(c->cfg.image_width, c->cfg.image_height) * (double)0.7f);   // point_matcher.cc:58
  // keypoint encoder: from block_min tokens on, its two large layers (98 of 108 kFLOP per keypoint) run as MFMA GEMMs
  const bool kenc_gemm = c->sg_kenc_gemm == 1 || (c->sg_kenc_gemm < 0 && Mg >= c->block_min);
  reset_slack_rows(c, M, st);
  launch_sg_prepare(c->mprec, f0, f1, n0, n1, AIRFE_FEAT_DIM, normalize, cx, cy, linv, c->sg_kenc, B, cap, Np, c->x32, c->xb,
                    c->lens, kenc_gemm ? c->msg : nullptr, st);
  if (kenc_gemm) {
    if (Mg > M) {          // the surplus rows of the 128-row rounding: zero inputs, so that the residual add leaves x = b4-ish garbage, not a running sum
      HIPCHK(c, hipMemsetAsync(c->msg + (size_t)M * 128, 0, (size_t)(Mg - M) * 128 * 2, st));
      HIPCHK(c, hipMemsetAsync(c->x32 + (size_t)M * 256, 0, (size_t)(Mg - M) * 256 * 4, st));
    }
    RUN(run_linear(c, c->sg_k3, c->msg, 128, 128, nullptr, 0, Mg, EPI_STORE, ACT_RELU, c->hb, 256, st));
    RUN(run_linear(c, c->sg_k4, c->hb, 256, 256, nullptr, 0, Mg, EPI_RESID, ACT_NONE, c->xb, 256, st, false, nullptr, c->x32));
  }
  const bool fused_block = c->fuse_lg_block == 1 || (c->fuse_lg_block < 0 && Mg >= c->block_min);
  int li = 0;
  for (const SgLayer& l : c->sg) {
    const int cross = li & 1;      // names = ['self','cross'] * 9
    ++li;
    RUN(run_qkv(c, l.qk, l.v, Mg, c->qb, c->kb, nullptr, nullptr, st));
    {
      ProfScope ps(c, ST_LG_ATTENTION, st, 4.0 * S * Np * (double)Np * 256, (double)M * 2048);
      run_attention(c, c->mprec, c->qb, c->kb, c->vtb, c->ob, c->lens, S, 4, Np, cross, 0.125f, st);
    }
    if (fused_block) {          // merge + mlp.0 + ReLU + mlp.3 + residual as ONE kernel (the LightGlue block kernel with ReLU for LN + GELU)
      RUN(lg_blockf(c, l.merge, l.mlp0, nullptr, nullptr, l.mlp3, Mg, st, 1));
      continue;
    }
    if (!c->fold_out) RUN(run_linear(c, l.merge, c->ob, 256, 256, nullptr, 0, Mg, EPI_STORE, ACT_NONE, c->msg, 256, st));
    RUN(run_linear(c, l.mlp0, c->xb, 256, 256, c->fold_out ? c->ob : c->msg, 256, Mg, EPI_STORE, ACT_RELU, c->hb, 512, st));
    RUN(run_linear(c, l.mlp3, c->hb, 512, 512, nullptr, 0, Mg, EPI_RESID, ACT_NONE, c->xb, 256, st, false, nullptr, c->x32));
  }
  RUN(run_linear(c, c->sg_final, c->xb, 256, 256, nullptr, 0, Mg, EPI_STORE, ACT_NONE, c->mdb, 256, st));
  launch_sim(c->mprec, c->mdb, c->simbuf, B, Np, st);
  launch_sg_sinkhorn(c->simbuf, c->lens, B, Np, c->Lz, c->sg_alpha, c->cfg.sinkhorn_iters, c->sg_u, c->sg_v, c->sg_Z, c->sg_cnt, c->sg_cnt + (size_t)c->Pmax * 16, c->sg_xch, st);
  launch_sg_decode(c->sg_Z, c->lens, B, Np, c->Lz, 0.2f, c->sg_idx0, c->sg_max0, c->sg_idx1, c->sg_out0, c->sg_out1, c->sg_ms0,
                   c->sg_ms1, st);
  return launch_status(c);
}


}  // namespace airfe_host
