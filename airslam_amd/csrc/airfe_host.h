// airfe — host side of libairfe.so, shared by its translation units: the context (persistent device arena, weights, staging blocks, stage timers),
// error / allocation helpers and the internal entry points of the pipelines.
//   airfe_load.hip    weight-pack loading and slab packing (≙ TensorRT engine build, src/plnet.cpp:24-196), the matcher arena
//   airfe_detect.hip  the detector pipeline (encoder, heads, NMS, top-K, descriptors) and the PLNet line path
//   airfe_match.hip   LightGlue / SuperGlue forwards and the fault-hunting trace
//   airfe.hip         context life cycle and the C ABI (include/airfe.h, include/airfe_debug.h)
#pragma once
#include "../../include/airfe.h"
#include "../../include/airfe_debug.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "common.h"

using namespace airfe;


namespace airfe_host {

extern thread_local std::string g_err;

struct Tensor {
  std::vector<int> dims;
  std::vector<float> data;
};
typedef std::map<std::string, Tensor> Pack;

// sqrt(scale * log2 e), scale = 1/sqrt(d_head) = 0.125: folded into BOTH the q and the k projection (weights and biases; rotary is
// linear, so it commutes), so that q.k comes out of the attention MFMA as log2(e) * (q.k) / 8, ready for v_exp_f32 — the product of two
// packed operands, each rounded once, exactly like the unscaled q and k were
constexpr float ATT_QK_FOLD = 0.42466090014400953f;

struct ConvW { uint16_t* w = nullptr; float* b = nullptr; int cin = 0, cout = 0; };
struct LinW { uint16_t* w = nullptr; float* b = nullptr; int K = 0, N = 0, cbt = 0; uint16_t* wf = nullptr; /* the same weights in MFMA fragment order (make_linear) */ };
struct LgLayer {
  LinW qk, v, out, ffn0, ffn3, cqk, cv, cout, cffn0, cffn3;
  float *ln_g = nullptr, *ln_b = nullptr, *cln_g = nullptr, *cln_b = nullptr;
};
struct SgLayer { LinW qk, v, merge, mlp0, mlp3; };
constexpr int LINE_CAP = 45056;     // unique candidate lines per image: 300 junctions give at most 300 * 299 / 2 = 44850 (min, max) pairs
constexpr int KEEP_CAP = 3 * 128 * 128;
constexpr int JUNC_CAP = 2048;
// One image's stage-0 line tensors (SURVEY.md Appendix A.1 layouts) inside its stage block, in floats; the CHW loi_features of the
// batch-1 / host-supplied path live in their own block (s0_loi): the batched path samples them from the head GEMM's rows instead.
constexpr size_t SG_JUNCS = 0, SG_LP = 600, SG_KEEP = SG_LP + (size_t)KEEP_CAP * 4, SG_MIN = SG_KEEP + KEEP_CAP, SG_MAX = SG_MIN + KEEP_CAP,
                 SG_THIN = SG_MAX + KEEP_CAP, SG_AUX = SG_THIN + 4 * 128 * 128, SG_STRIDE = (SG_AUX + 4 * 128 * 128 + 63) / 64 * 64;


}  // namespace airfe_host
using namespace airfe_host;

// airfe_stereo_keyframe's captured queue (one configuration at a time) and the host-side flags that describe what the queue leaves on the device
struct airfe_ctx;
struct KfState {
  bool nms_map_valid, desc_normalised, desc_dense_valid, line_sparse; int last_B;
  void save(const airfe_ctx* c);
  void restore(airfe_ctx* c) const;
};
struct KfGraph {
  struct Key {
    int h, w, stride, capL, capJ; bool want_j, match; const void *pin, *blk, *img;
    bool operator==(const Key& o) const {
      return h == o.h && w == o.w && stride == o.stride && capL == o.capL && capJ == o.capJ && want_j == o.want_j && match == o.match && pin == o.pin &&
             blk == o.blk && img == o.img;
    }
  } key{};
  hipGraphExec_t exec = nullptr;
  int seen = 0;
  KfState state{};
  void reset() { if (exec) (void)hipGraphExecDestroy(exec); exec = nullptr; seen = 0; }
};
struct airfe_ctx {
  airfe_cfg cfg;
  std::string err;
  std::string launch_err;        // cfg.check_launches: the first failed launch since the last report, with its stage's name (launch_status())
  int fail_stage = -1;           // airfe_debug_fail_next_launch: the next ProfScope of this stage makes a deliberately invalid launch first (tests)
  void* copy_ring = nullptr;     // airfe_copy_rows_dev: host-mapped ring of job lists (csrc/airfe_seq.hip)
  int copy_ring_cap = 0, copy_ring_slot = 0, copy_wgs = 64;   // airfe_tuning::copy_wgs
  int *sat_host = nullptr, *sat_flag = nullptr;   // two words of host-mapped, coherent memory (and their device address): [0] = non-finite detector logits, [1] = non-finite
                                 // sampled descriptor — written by the head / sampling kernels only when the 2-byte activations overflowed, read by the host after its synchronisation: no copy
  bool fuse_dec = true;          // airfe_tuning::fuse_dec
  int assign_fused = 0;          // airfe_tuning::assign_fused (1: partials in the similarity tiles; measured -0.010 ms per 64-pair step, +0.012 ms per batch-1 keyframe: not the default, profiles/r05_assign_ab.txt)
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;         // airfe_stereo_plnet_batch_dev: the line branch runs here while the matcher runs on the caller's stream
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_feat = nullptr;
  bool overlap_lines = true;             // line path on stream2 beside the matcher (airfe_stereo_plnet_batch_dev); airfe_tuning::overlap_lines = 0: one stream
  std::vector<void*> allocs;
  int prec = 0;                  // detector storage type
  int mprec = 1;                 // matcher storage type (cfg.matcher_precision)
  int pack_prec = 0;             // storage type make_linear packs for (set by each load_* before it packs)
  int Bmax = 1, chunk = 1, Np = 64, Pmax = 1;
  int lgb_tokens = 0;            // airfe_tuning::lgb_tokens: forces the fused block's tokens per workgroup (32 / 64 / 112 / 128)
  // batch-1 host entries: ONE pinned host block and contiguous device blocks, so that a call is one H2D and one D2H (the reference's
  // BufferManager does a cudaMalloc + one synchronous memcpy per binding and call: 3rdparty/tensorrtbuffer/include/buffers.h:237-417)
  uint8_t* pin = nullptr;        // hipHostMalloc'ed
  size_t pin_bytes = 0;
  uint8_t *io_in = nullptr, *io_out = nullptr;   // device: [n0 n1 .. | feat0 | feat1] and [nmatch .. | idx | score]
  bool trace_overflow = false;   // a trace slot was dropped (table full): trace_finish fails instead of mis-numbering launches
  size_t arena_rows = 0;         // token rows of the matcher arena, slack included (alloc_matcher_arena)
  int Dmax = 1;                  // images the detector arena holds: 2 x Bmax when a stereo step detects left and right as one batch
  bool has_sp = false, has_lg = false;
  uint8_t* pl_stage = nullptr;   // staging of airfe_assign_points_to_lines / airfe_match_lines
  size_t pl_bytes = 0;
  uint8_t *pl_scratch = nullptr, *ml_scratch = nullptr;   // scratch of airfe_assign_points_to_lines_batch_dev (counts) / airfe_match_lines_batch_dev (bit rows, row maxima): one each
  size_t pl_scratch_bytes = 0, ml_scratch_bytes = 0;
  hipStream_t pl_scratch_stream = nullptr, ml_scratch_stream = nullptr;   // the stream each was last used on (synchronised before the block is replaced)
  bool nms_map_valid = true;     // heat_nms holds the last batch's NMS'd maps (large batches skip writing them)
  bool force_nms_map = false;    // the batched PLNet path reads junction scores from them: written at every batch size while set
  int Lmax = 1;                  // images the line-path arena holds (= Dmax)
  bool desc_normalised = false;  // dense descriptor map currently holds F.normalize'd rows (only after the inspection hook)
  int gemm_small_max = 4096, gemm8_min = 16000, gemmr_min = 8192, gemmr_wgs = 256;   // GemmArgs::small_max / g8_min / gr_min / gr_wgs (airfe_tuning::gemm_small_max_m, gemm8_min_m, gemmr_min_m, gemmr_wgs)
  int block_min = 0;             // tokens from which the fused LightGlue block is used (airfe_tuning::block_min_m).  Round 4: with 32- / 64-token passes for small
                                 // token counts the fused kernel wins at EVERY size (profiles/r04_lg_small_batch_sweep.txt: 1 pair 0.70 vs 0.78 ms, 4 pairs 0.75 vs 1.20);
                                 // with 112- / 128-token passes only (rounds 1-3) the four separate launches were quicker below 3200 tokens
  bool qkv_pair = true;          // q|k and v of a layer in one streaming launch (airfe_tuning::qkv_pair = 0: two launches)
  int fuse_lg_block = -1;        // LightGlue out-proj + FFN + residual as one kernel: -1 by token count, airfe_tuning::fuse_lg_block = 0 / 1 forces
  int sg_kenc_gemm = -1;         // airfe_tuning::sg_kenc_gemm = 0 / 1: SuperGlue keypoint encoder's large layers as scalar loops / GEMMs (default: by token count)
  bool desc_dense_valid = true;  // c->desc holds the dense map of the last batch (else: the gather GEMM's rows)
  int last_B = 0;
  int* desc_idx = nullptr;       // row list of the descriptor head's gather GEMM
  int n_cu = 0;                  // compute units of cfg.device (hipDeviceAttributeMultiprocessorCount): the fused block's two-round split
  bool fold_qkv = true;          // airfe_tuning::fold_qkv = 0: q | k | v projections as launches of their own (A/B runs)
  bool desc_gather_stream = true;   // airfe_tuning::desc_gather_stream = 0: the descriptor head over sampled cells in the tiled kernel (gemm8) at every size (A/B, bit-identity test)
  bool fold_out = true;          // airfe_tuning::fold_out_proj: out_proj / to_out / merge multiplied into the message half of ffn.0 / mlp.0 at pack time (2-byte matcher only);
                                 // the packed ffn.0 / mlp.0 of the context then expect cat(x, attention output), and no entry runs the out-projection

  // detector weights
  float *c1a_w = nullptr, *c1a_b = nullptr;
  ConvW c1b, c2a, c2b, c3a, c3b, c4a, c4b, cPa, cDa;
  LinW cPb, cDb;
  // detector arena
  float* img32 = nullptr;
  uint16_t *a1b = nullptr, *a2a = nullptr, *a2b = nullptr, *a3a = nullptr, *a3b = nullptr, *a4a = nullptr,
           *a4b = nullptr, *aPa = nullptr, *aDa = nullptr;
  float *logits = nullptr, *heat = nullptr, *heat_nms = nullptr, *nms_tmp = nullptr, *desc = nullptr;
  unsigned char* nms_mask = nullptr;   // max_mask + supp_mask planes of the per-pool NMS launches
  int *xtab = nullptr, *ytab = nullptr;
  float* lut = nullptr;
  unsigned long long* cand = nullptr;   // [Bmax][512*512] detect_point candidate keys
  int* cand_cnt = nullptr;
  int tab_w = -1, tab_h = -1;
  // BoW vocabulary tree
  float *bow_desc = nullptr, *bow_weight = nullptr, *bow_outw = nullptr;
  std::vector<double> bow_weight_h;   // the vocabulary's WordValue weights as the reference holds them (double): the host entry returns these
  int* bow_outn = nullptr;            // leaf node per feature of the last host call
  int *bow_first = nullptr, *bow_nch = nullptr, *bow_word = nullptr;
  unsigned* bow_out = nullptr;
  int bow_nodes = 0;
  // rectification maps of Camera (camera.cc:60-75), one pair per side, and the rectified-image staging
  float* rmap[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  int rmap_h[2] = {0, 0}, rmap_w[2] = {0, 0};
  uint8_t* st_rect = nullptr; size_t st_rect_bytes = 0;
  // host-API staging
  uint8_t* st_img = nullptr; size_t st_img_bytes = 0;
  uint8_t* kf_blk = nullptr; size_t kf_bytes = 0;   // airfe_stereo_keyframe's device block (grows on demand)
  uint8_t *tk_blk = nullptr, *ref_blk = nullptr; size_t tk_bytes = 0, ref_bytes = 0;   // airfe_track_frame: outputs; the last keyframe's features
  int ref_n = -1;
  int tk_n = -1;                                    // keypoints of the last airfe_track_frame's frame (its rows are still in tk_blk); -1: none
  uint8_t* pr_blk = nullptr; size_t pr_bytes = 0;   // airfe_promote_frame: [counts | right rows | idx | score]
  int kf_spec_lines = 1024, kf_spec_juncs = 512;   // line / junction rows airfe_stereo_keyframe copies back before it knows the counts (airfe_tuning::kf_spec_rows)
  bool kf_graph_on = false;                         // airfe_tuning::kf_graph
  KfGraph kf_graph;
  float *st_feat0 = nullptr, *st_feat1 = nullptr, *st_score = nullptr;
  int *st_n0 = nullptr, *st_n1 = nullptr, *st_nm = nullptr;
  int32_t* st_idx = nullptr;
  float* st_scores_full = nullptr;

  // LightGlue
  std::vector<LgLayer> lg;
  LinW lg_final;
  float *lg_wr = nullptr, *lg_mw = nullptr;
  float lg_mb = 0.f;
  float *x32 = nullptr, *rot_cos = nullptr, *rot_sin = nullptr, *zbuf = nullptr, *simbuf = nullptr, *rowlse = nullptr,
        *collse = nullptr, *rowval = nullptr;
  uint16_t *xb = nullptr, *qb = nullptr, *kb = nullptr, *vtb = nullptr, *ob = nullptr, *msg = nullptr, *hb = nullptr,
           *mdb = nullptr;
  int *lens = nullptr, *rowarg = nullptr, *colarg = nullptr;
  float *lg_part = nullptr, *lg_argpart = nullptr;   // [Pmax][2][tiles][Np] float2 each: launch_lg_assign_fused
  bool has_arena = false;

  // SuperGlue
  bool has_sg = false;
  std::vector<SgLayer> sg;
  LinW sg_final;
  float sg_alpha = 1.f;
  const float* sg_kenc[10] = {nullptr};
  LinW sg_k3, sg_k4;             // keypoint-encoder layers 3 (128 -> 256) and 4 (256 -> 256) for the GEMM path
  int Lz = 0;
  float *sg_u = nullptr, *sg_v = nullptr, *sg_Z = nullptr, *sg_max0 = nullptr, *sg_ms0 = nullptr, *sg_ms1 = nullptr;
  int *sg_idx0 = nullptr, *sg_idx1 = nullptr;
  float* sg_xch = nullptr;       // [P][2][16][Lz] (max, sum) column partials of the register-resident Sinkhorn kernel
  unsigned* sg_cnt = nullptr;    // per-pair rendezvous counters of the fused Sinkhorn kernel
  int32_t *sg_out0 = nullptr, *sg_out1 = nullptr;

  // fp32 correctness path (cfg.precision = 2 / matcher_precision = 2): fp32 weights and activations, kernels_f32.hip
  struct F32Conv { float* w = nullptr; float* b = nullptr; int cin = 0, cout = 0; };
  struct F32Lin { float* w = nullptr; float* b = nullptr; int K = 0, N = 0; };
  struct F32LgLayer { F32Lin qkv, out, ffn0, ffn3, cqk, cv, cout, cffn0, cffn3; float *ln_g, *ln_b, *cln_g, *cln_b; };
  F32Conv f_c1b, f_c2a, f_c2b, f_c3a, f_c3b, f_c4a, f_c4b, f_cPa, f_cDa, f_cL1;
  struct F32SgLayer { F32Lin qkv, merge, mlp0, mlp3; };      // SuperGlue GNN layer: q | k | v rows head-major, merge's input columns head-major
  F32Lin f_cPb, f_cDb, f_cLh, f_lgfinal, f_sgfinal;
  std::vector<F32LgLayer> f_lg;
  std::vector<F32SgLayer> f_sg;
  int f_B = 0;                   // images per pass of the fp32 encoder (its activations are 4 bytes: 2 images at a time)
  float *f1a = nullptr, *f1b = nullptr, *fp1 = nullptr, *f2a = nullptr, *f2b = nullptr, *fp2 = nullptr, *f3a = nullptr, *f3b = nullptr,
        *fp3 = nullptr, *f4a = nullptr, *f4b = nullptr, *fPa = nullptr, *fDa = nullptr, *fL1 = nullptr;
  float *m_qkv = nullptr, *m_ctx = nullptr, *m_msg = nullptr, *m_h = nullptr, *m_md = nullptr;
  // PLNet stage-0 line branch (HAWP-style head on the shared trunk; weights ride in the detector pack as line.*)
  bool has_s0 = false;
  ConvW cL1;                     // line.conv1: 3x3 128 -> 128 on the conv3a features
  LinW cLh;                      // line.head : 1x1 128 -> 145 = loi (128) | md0-2 dis res | jloc0-1 | joffx joffy | thin0-3 | aux0-3
  LinW cLh_loi, cLh_dec;         // the same rows as two heads: the 128 LOI channels (run on the junctions' tap rows only) and the 17 decoded ones
  bool line_sparse = false;      // the last line_branch_dev ran the split heads (else: the fused head over one image, l_head)
  float* l_dec = nullptr;        // [Lmax][128*128][32]: the 17-channel head
  int* l_ridx = nullptr;         // [Lmax * 1200 (+ pad)]: tap rows of the junctions
  float* l_lrows = nullptr;      // [Lmax * 1200 (+ pad)][128]: LOI features of those rows
  uint16_t* l_feat = nullptr;    // [Lmax][128*128][128] 2-byte
  float *l_ta8 = nullptr /*[Lmax][128*128][8] thin | aux pixel-major*/, *l_head = nullptr, *l_jloc = nullptr, *l_joff = nullptr, *l_sel = nullptr;
  int* l_nsel = nullptr;
  unsigned long long* l_cand = nullptr;   // [Lmax][128*128] junction candidates (its own list: the line branch may run beside the point branch's tail)
  int* l_cand_cnt = nullptr;
  // PLNet stage 1 + line path
  bool has_s1 = false;
  const float* s1_w[11] = {nullptr};
  const uint16_t* s1_wsplit[6] = {nullptr};   // fc2.0 (thin / aux columns), fc2_res.0, fc2.2, fc2.4, fc2.0's LOI columns of end point 1 / 2 as fp16 (hi, lo) planes
  int *wf_table = nullptr, *wf_keep = nullptr, *wf_pairs = nullptr, *wf_rep = nullptr, *wf_counts = nullptr;
  bool wf_counted = false;                     // the stage block's per-workgroup keep counts are in wf_counts (launch_s0_j2l wrote them): launch_wireframe skips its count pass
  float* wf_prop = nullptr;                    // [L][LINE_CAP][4]: lines_pred of every unique line's first proposal (wireframe_kernel -> plnet_s1h_kernel)
  float *s1_la = nullptr, *s1_sc = nullptr, *s1_jfeat = nullptr /*[Lmax][300][256]*/, *s0_stage = nullptr /*[Lmax][SG_STRIDE]*/, *s0_loi = nullptr /*CHW [128][128][128], one image*/,
        *junc_feat = nullptr;
  unsigned char* jmap = nullptr;
  double* d_lines = nullptr;
  int *d_nlines = nullptr /*[Lmax] kept | [Lmax] found*/, *d_njunc = nullptr /*[Lmax] kept | [Lmax] found | [Lmax][64] scan scratch*/;

  // fault hunting (airfe_debug_trace*): checksums of the matcher's state behind every launch of lightglue_dev
  struct TraceSlot { std::string name; unsigned off, units, unit_words; const void* p; size_t words; };
  bool trace_on = false, trace_halt = false;
  int trace_stop = -1;           // >= 0: the forward pass returns right behind this slot (its buffer stays as that launch left it)
  unsigned long long *trace_tab = nullptr, *trace_dig = nullptr;
  unsigned* trace_off = nullptr;
  size_t trace_cap = 0;
  std::vector<TraceSlot> trace_slots;
  std::vector<unsigned> trace_off_h;

  // per-stage hipEvent timers (airfe_profile_*): events are recorded on the launch stream only
  struct Mark { int stage; hipEvent_t a, b; double flops, bytes; };
  uint32_t prof_mask = 0;        // bit i = stage i is bracketed by events
  std::vector<Mark> marks;
  std::vector<hipEvent_t> ev_pool;
};

enum Stage {
  ST_PREPROCESS = 0, ST_CONV1_FUSED /* conv1a + conv1b + pool: the dominant kernel, its own stage */, ST_CONV3X3_C64, ST_CONV3X3_C128, ST_HEAD_GEMM, ST_HEAD_ELTWISE, ST_NMS, ST_SELECT,
  ST_SAMPLE, ST_LG_PREPARE, ST_LG_GEMM, ST_LG_ATTENTION, ST_LG_LNGELU, ST_LG_ASSIGN, ST_PL_DECODE, ST_PL_STAGE1, ST_PL_FILTER, ST_LINE_ASSOC, ST_RECTIFY, ST_BOW, ST_COUNT
};
static const char* kStageNames[ST_COUNT] = {
  "preprocess", "conv1_fused", "conv3x3_cin64", "conv3x3_cin128", "head_gemm", "head_eltwise", "simple_nms", "select_topk",
  "sample_desc", "lg_prepare", "lg_gemm", "lg_attention", "lg_ln_gelu", "lg_assign", "plnet_s0_decode", "plnet_stage1", "plnet_filter", "line_assoc", "rectify", "bow"};

void note_launch(airfe_ctx* c, int stage);      // airfe.hip
void fail_launch_now(hipStream_t st);            // airfe.hip: a deliberately invalid launch (airfe_debug_fail_next_launch)
struct ProfScope {
  airfe_ctx* c; hipStream_t st; bool on; int stage; airfe_ctx::Mark m;
  ProfScope(airfe_ctx* c_, int stage_, hipStream_t st_, double flops, double bytes) : c(c_), st(st_), on((c_->prof_mask >> stage_) & 1u), stage(stage_) {
    if (c->fail_stage == stage) { c->fail_stage = -1; fail_launch_now(st); if (c->cfg.check_launches) note_launch(c, stage); }
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
      else (void)hipEventCreate(&e);
      return e;
    };
    m.stage = stage; m.flops = flops; m.bytes = bytes; m.a = get(); m.b = get();
    (void)hipEventRecord(m.a, st);
  }
  ~ProfScope() {
    if (c->cfg.check_launches) note_launch(c, stage);      // the launches of this stage: reported by launch_status() with the stage's name
    if (!on) return;
    (void)hipEventRecord(m.b, st);
    try { c->marks.push_back(m); } catch (...) { c->ev_pool.clear(); }      // (a destructor must not throw; the events leak rather than the process die)
  }
};

namespace airfe_host {

#define HIPCHK(ctx, expr)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
      g_err = (ctx)->err;                                                                         \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)

int fail(airfe_ctx* c, const std::string& m);
int fail_noexcept(airfe_ctx* c, const char* what, const char* detail) noexcept;   // for the catch blocks at the C boundary
// 0 when no launch failed since the last report; else the context's error = "<stage>: kernel launch failed: <hip error>" and 1.  With
// cfg.check_launches every stage's launches are looked at as the stage ends (ProfScope); without it this is one hipGetLastError().
int launch_status(airfe_ctx* c);
// Reads and clears the detector's saturation words (after the caller's synchronisation): non-zero -> the context's error says that the 2-byte activations left the
// fp16 / bf16 range and 1 is returned — never keypoints of a poisoned score map.
int saturation_status(airfe_ctx* c);

template <class T>
T* dalloc(airfe_ctx* c, size_t n, bool zero = true) {
  void* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
  if (zero) (void)hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T));
  c->allocs.push_back(p);
  return reinterpret_cast<T*>(p);
}
template <class T>
T* dupload(airfe_ctx* c, const std::vector<T>& v) {
  T* p = dalloc<T>(c, v.size(), false);
  if (p && !v.empty()) (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

// ---- airfe_load.hip
int load_superpoint(airfe_ctx* c, const char* path);
int load_lightglue(airfe_ctx* c, const char* path);
int load_superglue(airfe_ctx* c, const char* path);
int load_plnet_s1(airfe_ctx* c, const char* path);
std::vector<int> resize_table(int dsize, int ssize);
std::vector<uint16_t> pack_slabs(int cbt, int nslab, int prec, const std::function<float(int, int, int)>& get);
uint16_t f2bf(float f);
uint16_t f2h(float f);
float h2f(uint16_t h);
inline uint16_t cvt2(float f, int prec) { return prec == 1 ? f2h(f) : f2bf(f); }
inline float back2(uint16_t v, int prec) {
  if (prec == 1) return h2f(v);
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
bool make_linear(airfe_ctx* c, const float* W, const float* bias, int K, int N, LinW& out, float scale = 1.f,
                 const std::function<int(int)>* src_row = nullptr, const std::function<int(int)>* src_col = nullptr);
// ---- airfe_detect.hip
int ensure_tables(airfe_ctx* c, int h, int w);
int run_conv(airfe_ctx* c, const ConvW& w, const uint16_t* x, uint16_t* y, int B, int H, int W, int pool, int out_pad, hipStream_t st);
int dense_desc_head(airfe_ctx* c, int B, hipStream_t st);
int detect_dev2(airfe_ctx* c, const uint8_t* d_gray, const uint8_t* d_gray1, int Bs, int h, int w, int stride, size_t img_stride,
                float* d_feat, float* d_feat1, int cap, int* d_n, int* d_n1, hipStream_t st);
int detect_dev(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, float* d_feat, int cap, int* d_n, hipStream_t st);
int line_branch_dev(airfe_ctx* c, hipStream_t st, int i0, int nb, bool chw);
int line_tail_dev(airfe_ctx* c, int i0, int nb, const float* loi_chw, int h, int w, double* d_lines, int capL, int* d_nlines, int* d_lfound,
                  float* d_junc, int capJ, int* d_njunc, int* d_jfound, int nj, hipStream_t st, int phase = 3);
// ---- airfe_match.hip
void trace(airfe_ctx* c, hipStream_t st, const char* what, size_t li, const char* blk, const void* p, size_t words, unsigned unit_words);
int trace_finish(airfe_ctx* c, hipStream_t st);
int run_linear(airfe_ctx* c, const LinW& w, const uint16_t* x1, int ld1, int K1, const uint16_t* x2, int ld2, int M, int epi, int act, void* out,
                int ldo, hipStream_t st, bool trans = false, void* out2 = nullptr, float* x32 = nullptr, const float* rc = nullptr, const float* rs = nullptr);
void reset_slack_rows(airfe_ctx* c, int M, hipStream_t st);
struct LgSecondPair { const float *f0, *f1; const int *n0, *n1; };      // a second pair for a B = 1 call (same ld / kp_off / normalize): outputs of pair 1 follow pair 0's
int lightglue_dev(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int ld, int kp_off, int normalize,
                  int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, float* scores_out, hipStream_t st, const LgSecondPair* x2 = nullptr);
int superglue_dev(airfe_ctx* c, const float* f0, const int* n0, const float* f1, const int* n1, int B, int cap, int normalize, hipStream_t st);

}  // namespace airfe_host
