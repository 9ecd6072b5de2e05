/* airfe — C ABI of the MI355X-native per-frame front end (feature detect + match) for AirSLAM.
 *
 * This header is the drop-in boundary: every entry point names the reference interface it
 * replaces (paths under the AirSLAM checkout).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - return value: 0 = ok, non-zero = failure (airfe_last_error() gives the text); never throws: every entry point catches C++ exceptions
 *     (allocation failures of its host-side bookkeeping included) at the boundary and reports them as a failure.
 *   - one ctx = one HIP stream = one calling thread (the reference wrappers are not re-entrant either:
 *     include/plnet.h:38-63, include/light_glue.h:39-47).
 *   - feature rows are 259 contiguous floats [score, x, y, d0..d255]: byte-identical to one COLUMN of the
 *     reference's column-major Eigen::Matrix<float,259,Dynamic> (include/feature_detector.h:8-31), so the
 *     C++ shim does features.resize(259,n) + one memcpy.
 *   - activation range: with 2-byte detector storage (cfg.precision 0 / 1) an activation above 65504 (fp16) overflows.  Weight packs made by tools/onnx_to_pack.py
 *     are rescaled between layers by exact powers of two (airslam_amd.weights.fold_activation_scales) so that calibration maxima sit at <= 2048; if a frame still
 *     drives the score logits or a sampled descriptor to inf / NaN, the host entries FAIL for that call (airfe_last_error says so) and the asynchronous *_dev
 *     entries report it through airfe_sync / airfe_superglue_status — keypoints of a poisoned score map are never handed out.  cfg.precision = 2 has fp32 range.
 *   - *_dev entry points take DEVICE pointers and an optional hipStream_t (NULL = the ctx stream); they are
 *     asynchronous.  They have no reference counterpart (the reference is batch-1, host buffers only).
 */
#ifndef AIRFE_H_
#define AIRFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIRFE_FEAT_DIM 259
#define AIRFE_INTERNAL_SIZE 512 /* reference resizes every image to 512x512: src/plnet.cpp:17-18,258 */

typedef struct airfe_ctx airfe_ctx;

/* Every field: -1 = the library's default / automatic choice.  None of these changes a result beyond what the tests state (most forms are bit-identical). */
typedef struct airfe_tuning {
  int fuse_lg_block;     /* LightGlue / SuperGlue out-projection + FFN + residual as one kernel: 0 / 1 force, -1 by token count */
  int gemm_small_max_m;  /* rows up to which the no-LDS GEMM is used */
  int gemm8_min_m;       /* rows from which the 8-wave tiled GEMM is used */
  int gemmr_min_m;       /* rows from which the streaming (DMA ring) q|k|v projection is used */
  int gemmr_wgs;         /* its persistent workgroups (tests lower it so that a small batch wraps the ring) */
  int qkv_pair;          /* q|k and v of a layer in one streaming launch (1) or two (0) */
  int block_min_m;       /* tokens from which the fused block is used when fuse_lg_block = -1 */
  int lgb_tokens;        /* tokens per workgroup of the fused block: 32 / 64 / 112 / 128 */
  int sg_kenc_gemm;      /* SuperGlue keypoint encoder's large layers as GEMMs (1) or scalar loops (0) */
  int fold_qkv;          /* the next attention layer's projections inside the fused block (1) or as launches of their own (0) */
  int overlap_lines;     /* PLNet line path on a second stream beside the matcher (1) or behind it (0) */
  int kf_graph;          /* airfe_stereo_keyframe replays a captured hipGraph (1); default 0 (measured: <= 1 %) */
  int kf_spec_rows;      /* line / junction rows airfe_stereo_keyframe copies back before it knows the counts */
  int fuse_dec;          /* PLNet stage-0: 17-channel head + decode in one pass (1) or two (0) */
  int assign_fused;      /* LightGlue assignment: log-sum-exp / arg-max partials taken in the similarity tiles (1; no similarity matrix in HBM)
                            or the round-2 form: similarity matrix + four passes over it (0, the default: the A/B is in profiles/r05_assign_ab.txt) */
  int fold_out_proj;     /* LightGlue / SuperGlue: the attention out-projection (out_proj / to_out / merge) multiplied into the message half of ffn.0 / mlp.0 when
                            the weights are packed — two linear maps with nothing between them are one: W1m (Wo a + bo) = (W1m Wo) a + W1m bo — so a block runs
                            ffn.0 on cat(x, attention output) and the 256x256 GEMM, its barrier and its message tile are gone (1, the default for fp16 / bf16);
                            0: the out-projection as a GEMM of its own (the round 1-4 form; profiles/r05_fold_out_ab.txt) */
  int desc_gather_stream;/* the descriptor head over the sampled cells of a large batch (>= gemmr_min_m rows) in the streaming kernel with gathered rows (1, default)
                            or the tiled 8-wave kernel (0); the same bits (tests/test_gpu_detector.py).  Round 6: also the LOI head at the junctions' tap rows
                            (K = N = 128) and the dense descriptor head of the junction images */
  int copy_wgs;          /* airfe_copy_rows_dev: workgroups of the copy kernel (default 64: PCIe-bound copies into pinned memory need stores in flight, not CUs) */
  int reserved[5];       /* must be -1 */
} airfe_tuning;

/* Mirrors the knobs of PLNetConfig / SuperPointConfig / PointMatcherConfig (include/read_configs.h:9-103). */
typedef struct airfe_cfg {
  int device;                  /* HIP device ordinal */
  int precision;               /* detector storage type: 2 = fp32 storage AND arithmetic (correctness mode, f32-input MFMA; BASELINE configs[1]),
                                  1 = fp16 (default; the reference's engines are built with kFP16,
                                  super_point.cpp:97, plnet.cpp:216 — and the only 2-byte type that meets the 1e-3 descriptor-cosine
                                  tolerance once descriptors are decorrelated: bf16 measures 2e-2), 0 = bf16; accumulation is always fp32 */
  int max_batch;               /* images per detect batch / 2x pairs per match batch the arena is sized for */
  int enc_chunk;               /* images per pass through the full-resolution conv layers (cache blocking) */
  int max_keypoints;           /* plnet.max_keypoints        (<= 1024, light_glue.cpp:52) */
  float keypoint_threshold;    /* plnet.keypoint_threshold */
  int remove_borders;          /* plnet.remove_borders */
  int nms_radius;              /* SuperPoint simple_nms radius inside the model graph (4 upstream; 0 = off) */
  float line_threshold;        /* plnet.line_threshold */
  float line_length_threshold; /* plnet.line_length_threshold */
  int matcher;                 /* point_matcher.matcher: 0 = LightGlue, 1 = SuperGlue */
  int image_width;             /* point_matcher.image_width / image_height (NormalizeKeypoints) */
  int image_height;
  int sinkhorn_iters;          /* SuperGlue: iterations baked into the exported graph (100 upstream) */
  const char* superpoint_pack; /* weight packs (airslam_amd/weights.py format); NULL = that model is unavailable */
  const char* plnet_s1_pack;
  const char* lightglue_pack;
  const char* superglue_pack;
  int matcher_precision;       /* storage type of the LightGlue / SuperGlue tokens and weights: 1 = fp16 (default: the reference builds
                                  both matcher engines with BuilderFlag::kFP16, light_glue.cpp:115, super_glue.cpp:132; measured 8x
                                  closer to the fp32 oracle than bf16), 0 = bf16, 2 = fp32 (correctness mode, both matchers: f32-input MFMA GEMMs, exact soft-max), -1 = same as `precision` */
  int line_precision;          /* how the PLNet stage-1 LOI head's matrix products (src/plnet.cpp:468-514) are computed on the device path:
                                  3 = fp32 operands as PAIRS of fp16 values on the 2-byte MFMA (hi.hi + hi.lo + lo.hi, fp32 accumulation): the lines of the
                                      fp32 chain (scores within 2e-6, no candidate across the 0.75 threshold), on the pipe the reference runs this engine on
                                      (BuilderFlag::kFP16, src/plnet.cpp:216) without its rounding.  Range: the operands (LOI / thin / aux samples, hidden
                                      activations) are clamped to +-65504 before the split — the real head's stay below 64; a value beyond fp16's range enters
                                      as +-65504 instead of turning a line's score into NaN;
                                  2 = fp32 operands on the f32-input MFMA (157 TFLOP/s): the same lines, 1.8x the stage's time;
                                  1 = plain fp16 operands, REFUSED: emulated with the real weights it moves 0.5-0.9 % of the kept lines across the 0.75
                                      threshold (profiles/r05_s1_fp16_emulation.txt);
                                  0 = default: 3, and 2 in fp32 mode (precision = 2) */
  int check_launches;          /* 1 = hipGetLastError() behind every stage's launches: a failed launch is reported by the call that made it, with
                                  the stage's name (tests run with it); 0 = once per pipeline (default) */
  const airfe_tuning* tuning;       /* kernel-selection overrides (NULL = the library's own choices): A/B measurements and tests that must reach every
                                  kernel form.  Read once by airfe_create; the library never reads the environment. */
} airfe_cfg;

void airfe_default_cfg(airfe_cfg* cfg);
void airfe_default_tuning(airfe_tuning* t);      /* every field -1 */

/* ≙ the build() calls made by FeatureDetector / PointMatcher constructors
 *   (src/feature_detector.cc:7-34, src/point_matcher.cc:6-37): loads + packs weights, allocates the
 *   persistent device arena (replaces the per-infer cudaMalloc of 3rdparty/tensorrtbuffer, buffers.h:253-271). */
int airfe_create(const airfe_cfg* cfg, airfe_ctx** out);
void airfe_destroy(airfe_ctx* ctx);
const char* airfe_last_error(const airfe_ctx* ctx); /* ctx may be NULL (creation errors) */

/* ≙ SuperPoint::infer (src/super_point.cpp:103-144).  gray: h x w uint8, `stride` bytes per row
 *   (cv::Mat::step).  feat: caller buffer [cap][259]; *n receives the keypoint count (<= max_keypoints).
 *   x,y are in ORIGINAL image pixels.  Fails (non-zero) on an empty image, like the reference returns false. */
int airfe_detect_points(airfe_ctx* ctx, const uint8_t* gray, int h, int w, int stride, float* feat, int cap, int* n);

/* ≙ PLNet::infer (src/plnet.cpp:221-244).  Point branch as above.  Line branch: stage0 == NULL (what the shim passes) runs
 *   the stage-0 line head ON THE DEVICE when the detector pack carries it (tensors line.conv1.*, line.head.*: a HAWPv3-style
 *   head producing the Appendix A.1 tensors juncs_pred, lines_pred, iskeep, idx_junc_to_end_min/max, loi_features[_thin|_aux] —
 *   plnet_s0.onnx itself is absent from the reference checkout, so its weights here are synthetic); a non-NULL stage0 supplies
 *   those tensors from the HOST instead (known-answer tests of everything downstream).  Downstream of them wireframe_matcher
 *   :272-307, the stage-1 LOI head :468-514, the line/junction filter :519-558, junction_detector :425-448 and the rescale
 *   :569-582 run on the device.  lines: [capL][4] doubles (x1,y1,x2,y2) original pixels (std::vector<Eigen::Vector4d> layout);
 *   junc: [capJ][259].  No line branch in the pack and stage0 == NULL -> points only (counts 0).  The reference has no limit on lines
 *   or junctions: results that do not fit capL / capJ (45056 lines, 2048 junctions always do) are an ERROR, never a shorter list. */
typedef struct airfe_plnet_stage0 {
  const float* juncs_pred;          /* [300][2]        */
  const float* lines_pred;          /* [3*128*128][4]  */
  const float* iskeep;              /* [3*128*128]     */
  const float* idx_junc_to_end_min; /* [3*128*128]     */
  const float* idx_junc_to_end_max; /* [3*128*128]     */
  const float* loi_features;        /* [128][128][128] CHW */
  const float* loi_features_thin;   /* [4][128][128]   */
  const float* loi_features_aux;    /* [4][128][128]   */
} airfe_plnet_stage0;
int airfe_has_line_branch(const airfe_ctx* ctx); /* 1 when the detector pack carried line.* tensors AND stage 1 is loaded: infer() yields lines */
int airfe_detect_plnet(airfe_ctx* ctx, const uint8_t* gray, int h, int w, int stride, const airfe_plnet_stage0* stage0,
                       float* feat, int cap, int* n, double* lines, int capL, int* nlines, float* junc, int capJ,
                       int* njunc, int want_junctions);
/* ONE stereo keyframe through host buffers — the batch-1 entry for what src/map_builder.cc:85-86 does in two facade calls:
 *   _feature_detector->Detect(left, right, left_features, right_features, left_lines, right_lines, junctions)   (feature_detector.cc:97-108:
 *   PLNet::infer on the left image with junctions, on the right one without) and _point_matcher->MatchingPoints(left_features, right_features,
 *   stereo_matches, false) (point_matcher.cc:50-107 with LightGlue).
 * Both images go up in one copy, the detector runs over them as one batch of two, the line path runs beside LightGlue, the results come back in
 * two copies.  Per image / per pair the outputs are the bits airfe_detect_plnet x2 + airfe_match_lightglue (on NormalizeKeypoints'ed rows) return:
 *   featL / featR [cap >= max_keypoints][259] rows {score, x, y, desc[256]} + *nL / *nR;  linesL / linesR [capL][4] doubles + counts;
 *   juncL [capJ][259] + *njuncL (NULL: no junction detection);  match_idx [mcap >= max_keypoints][2] (left, right), match_score (the reference's
 *   DMatch::distance is 1 - score), *nmatch — match_idx == NULL: detection only (the 7-argument Detect overload alone).
 * Needs a detector arena of two images (cfg.max_batch >= 2, or the detector and LightGlue packs both loaded) and fp16 / bf16 arithmetic. */
int airfe_stereo_keyframe(airfe_ctx* ctx, const uint8_t* left, const uint8_t* right, int h, int w, int stride, float* featL, float* featR, int cap,
                          int* nL, int* nR, double* linesL, double* linesR, int capL, int* nlinesL, int* nlinesR, float* juncL, int capJ,
                          int* njuncL, int32_t* match_idx, float* match_score, int mcap, int* nmatch);
/* airfe_stereo_keyframe + the temporal match of the same frame: src/map_builder.cc:85-86 AND :96 — MatchingPoints(features_last_keyframe, left_features,
 * matches, true) — which every keyframe candidate runs as well.  The two LightGlue calls are independent, so they ride in ONE forward as a batch of two pairs
 * (at this size a forward costs the same for one pair as for two).  ref_feat [n_ref][259] = the last keyframe's features (NULL: the ones already on the device,
 * shared with airfe_track_frame); track_idx [mcap][2] = (reference index, left index), track_score, *ntrack.  Everything else as airfe_stereo_keyframe; per
 * pair the bits of two separate airfe_match_lightglue calls.  Needs cfg.max_batch >= 2. */
int airfe_stereo_keyframe_tracked(airfe_ctx* ctx, const uint8_t* left, const uint8_t* right, int h, int w, int stride, float* featL, float* featR, int cap,
                                  int* nL, int* nR, double* linesL, double* linesR, int capL, int* nlinesL, int* nlinesR, float* juncL, int capJ,
                                  int* njuncL, int32_t* match_idx, float* match_score, int mcap, int* nmatch, const float* ref_feat, int n_ref,
                                  int32_t* track_idx, float* track_score, int* ntrack);
/* ONE tracked (non-keyframe) frame through host buffers — src/map_builder.cc:94-101: Detect(image_left_rect, left_features) followed by
 * MatchingPoints(features_last_keyframe, left_features, matches, true) (the F-matrix RANSAC behind the matcher, point_matcher.cc:95-104, stays the
 * reference's).  ref_feat [n_ref][259] = the last keyframe's features: uploaded when given, KEPT on the device when NULL (pass them once per keyframe);
 * feat / *n = the new frame's features; match_idx [mcap][2] = (reference index, new index), match_score, *nmatch.  Same bits as airfe_detect_points +
 * airfe_match_lightglue on NormalizeKeypoints'ed rows.  Needs the detector and LightGlue packs in one context, fp16 / bf16. */
int airfe_track_frame(airfe_ctx* ctx, const uint8_t* gray, int h, int w, int stride, const float* ref_feat, int n_ref, float* feat, int cap, int* n,
                      int32_t* match_idx, float* match_score, int mcap, int* nmatch);

/* PROMOTION of the frame of the last airfe_track_frame — src/map_builder.cc:104-108: AddKeyframeCheck wants the normal frame as a keyframe, so the
 * feature thread runs Detect(image_right_rect, right_features) + MatchingPoints(left_features, right_features, stereo_matches, false) on it — as one queue:
 * the left features are the rows airfe_track_frame left on the device.  featR / *nR = the right image's features, match_idx [mcap][2] = (left, right).
 * Same bits as airfe_detect_points + airfe_match_lightglue on NormalizeKeypoints'ed rows.  Fails when no airfe_track_frame preceded it in this context. */
int airfe_promote_frame(airfe_ctx* ctx, const uint8_t* right, int h, int w, int stride, float* featR, int cap, int* nR, int32_t* match_idx,
                        float* match_score, int mcap, int* nmatch);
/* ≙ `_last_keyframe_feature = frame` (src/map_builder.cc:139-141) for a promoted frame: the rows of the last airfe_track_frame become the reference of the
 * following airfe_track_frame / airfe_stereo_keyframe_tracked calls (ref_feat == NULL) by a device-side copy — nothing crosses PCIe. */
int airfe_adopt_reference(airfe_ctx* ctx);

/* ≙ SuperPointLightGlue::infer (src/light_glue.cpp:120-170).  f0/f1: [n][258] rows = (x,y already normalised by
 *   PointMatcher::NormalizeKeypoints, d0..d255) — the contiguous temporary Eigen makes for bottomRows(258)
 *   at src/point_matcher.cc:67.  idx: [cap][2] (row-major, ascending in idx0), score = exp(log score). */
int airfe_match_lightglue(airfe_ctx* ctx, const float* f0, int n0, const float* f1, int n1, int32_t* idx, float* score,
                          int cap, int* nmatch);

/* ≙ SuperGlue::infer (src/super_glue.cpp:136-197).  f0/f1: [n][259] rows with normalised x,y.
 *   idx0 [n0], idx1 [n1] (-1 = unmatched), ms0 [n0], ms1 [n1] doubles (decode, src/super_glue.cpp:339-367). */
int airfe_match_superglue(airfe_ctx* ctx, const float* f0, int n0, const float* f1, int n1, int32_t* idx0, int32_t* idx1,
                          double* ms0, double* ms1);

/* ---- next row after the path ---------------------------------------------------- */
/* ≙ AssignPointsToLines (src/line_processor.cc:68-120), called on the path's own outputs (frame.cc:125,177,184).
 *   lines [L][4] doubles (x1,y1,x2,y2) = the std::vector<Eigen::Vector4d> storage; feat [N][259] rows (x,y = floats 1,2).
 *   Result in CSR form: row_ptr [L+1]; for line i the entries row_ptr[i] .. row_ptr[i+1]-1 of pt_idx / pt_dist are the
 *   (point index, distance) pairs of relation[i] in ascending point index = the iteration order of its std::map<int,double>.
 *   cap = capacity of pt_idx / pt_dist; *total = row_ptr[L]; fails (and writes nothing past cap) if total > cap. */
int airfe_assign_points_to_lines(airfe_ctx* ctx, const double* lines, int L, const float* feat, int N, int32_t* row_ptr,
                                 int32_t* pt_idx, double* pt_dist, int cap, int* total);

/* ≙ MatchLines (src/line_processor.cc:122-172), called right after the point matcher on two frames' relations (frame.cc:177-190).
 *   (row_ptr0, pt_idx0) / (row_ptr1, pt_idx1): the CSR relations of airfe_assign_points_to_lines for frame 0 / frame 1 (L0 / L1 lines,
 *   point_num0 / point_num1 keypoints); matches [M][2] = (queryIdx, trainIdx) of the cv::DMatch list.
 *   line_matches [L0]: index of the matched line of frame 1 or -1 (all -1 when any of the four counts is 0, :132). */
int airfe_match_lines(airfe_ctx* ctx, const int32_t* row_ptr0, const int32_t* pt_idx0, int L0, int point_num0, const int32_t* row_ptr1,
                      const int32_t* pt_idx1, int L1, int point_num1, const int32_t* matches, int M, int32_t* line_matches);

/* ---- the step AFTER the path: BoW quantisation of the descriptors ------------------------------------------ */
/* ≙ the vocabulary Database's constructor loads (src/bow/database.cc; voc/point_voc_L4.bin is absent from the reference checkout):
 *   the tree of TemplatedVocabulary (3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:298-323) flattened so that the children of
 *   node i are the n_children[i] consecutive nodes from first_child[i] on (0 children = leaf = word); node 0 is the root;
 *   node_desc [n_nodes][256] floats, word_id / weight per node (meaningful at leaves). */
int airfe_bow_load(airfe_ctx* ctx, const float* node_desc, const int32_t* first_child, const int32_t* n_children, const int32_t* word_id,
                   const double* weight, int n_nodes);
/* ≙ the loop of Database::FrameToBow (src/bow/database.cc:66-84): TemplatedVocabulary::transform(feature, id, w) (…Vocabulary.h:1313-1352)
 *   for each of the N feature rows [N][259]: word_of_features[i] = leaf word id, or UINT_MAX where the leaf weight is <= 0;
 *   weight_of_features[i] = w (may be NULL).  bow_vector.addWeight / normalize and word_features (std::map work) stay reference code. */
int airfe_bow_transform(airfe_ctx* ctx, const float* feat, int N, uint32_t* word_of_features, double* weight_of_features);
int airfe_bow_transform_dev(airfe_ctx* ctx, const float* d_feat, int N, uint32_t* d_word, float* d_weight, void* stream);

/* ---- the step BEFORE the path: rectification ------------------------------------------------------- */
/* ≙ the maps Camera's constructor builds with cv::initUndistortRectifyMap (src/camera.cc:60-75; _mapl1/_mapl2 = side 0, _mapr1/_mapr2 =
 *   side 1): CV_32FC1 x / y maps [h][w], uploaded once.  The map CONSTRUCTION (stereoRectify etc.) stays reference code. */
int airfe_set_rectify_maps(airfe_ctx* ctx, int side, const float* mapx, const float* mapy, int h, int w);
/* ≙ Camera::UndistortImage (src/camera.cc:161-182: cv::remap(..., INTER_LINEAR), BORDER_CONSTANT 0) + FeatureDetector::Detect on its
 *   result, in one call: raw HOST image in; rect_out (HOST, h x w tight rows, may be NULL) receives the rectified image the tracker and
 *   the visualisation still need (map_builder.cc:43,71-72,177); feat / n as airfe_detect_points (feat may be NULL: rectify only).
 *   The rectified image goes from the remap kernel straight into the detector's pre-process without leaving the device. */
int airfe_rectify_detect_points(airfe_ctx* ctx, int side, const uint8_t* raw, int h, int w, int stride, uint8_t* rect_out, float* feat,
                                int cap, int* n);
/* device-resident batch form: d_raw / d_rect [B] images (image b at + b * img_stride, rows stride bytes apart) */
int airfe_rectify_batch_dev(airfe_ctx* ctx, int side, const uint8_t* d_raw, int B, int h, int w, int stride, size_t img_stride,
                            uint8_t* d_rect, int rstride, size_t rimg_stride, void* stream);

/* ---- device-resident batch pipeline (NEW: no reference counterpart) ------------------------------------ */
/* d_gray: [B] images, image b at d_gray + b*img_stride, rows `stride` bytes apart.  d_feat [B][cap][259], d_n [B]. */
int airfe_detect_points_batch_dev(airfe_ctx* ctx, const uint8_t* d_gray, int B, int h, int w, int stride,
                                  size_t img_stride, float* d_feat, int cap, int* d_n, void* stream);
/* LightGlue on B pairs of device feature matrices (259-float rows, ORIGINAL pixel coords; NormalizeKeypoints with
 *   cfg.image_width/height is applied on the device exactly as src/point_matcher.cc:39-48 does on the host).
 *   d_idx [B][mcap][2], d_score [B][mcap], d_nmatch [B]. */
int airfe_match_lightglue_batch_dev(airfe_ctx* ctx, const float* d_f0, const int* d_n0, const float* d_f1, const int* d_n1,
                                    int B, int cap, int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, void* stream);
/* SuperGlue on B pairs (as above; NormalizeKeypoints scale 0.7, src/point_matcher.cc:58): d_idx0 / d_idx1 [B][cap] (-1 = unmatched,
 *   decode semantics of src/super_glue.cpp:339-367), d_ms0 / d_ms1 [B][cap] floats. */
int airfe_match_superglue_batch_dev(airfe_ctx* ctx, const float* d_f0, const int* d_n0, const float* d_f1, const int* d_n1, int B, int cap,
                                    int32_t* d_idx0, int32_t* d_idx1, float* d_ms0, float* d_ms1, void* stream);
/* One "stereo detect+match pair" x B (≙ map_builder.cc:85-86: Detect(L,R) + MatchingPoints(L,R)). */
int airfe_stereo_batch_dev(airfe_ctx* ctx, const uint8_t* d_left, const uint8_t* d_right, int B, int h, int w, int stride,
                           size_t img_stride, float* d_featL, float* d_featR, int cap, int* d_nL, int* d_nR,
                           int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, void* stream);
/* NEW (no reference counterpart: PLNet::infer, src/plnet.cpp:221-244, is one image per call): PLNet over B device-resident images in one
 * pass — points exactly as airfe_detect_points_batch_dev; the line branch (stage-0 line head, wireframe_matcher, stage 1, line filter)
 * for every image; junction_detector (+ descriptors) for the FIRST `junction_images` images.  Per image the results are the bits
 * airfe_detect_plnet returns for it.
 *   d_lines [B][capL][4] doubles (x1,y1,x2,y2 in original-image pixels), d_nlines [B] (<= capL)
 *   d_junc [junction_images][capJ][259], d_njunc [junction_images] (<= capJ)
 *   d_found (may be NULL) [B + junction_images]: lines that passed the filter per image, then junctions found per image — a value
 *   above capL / capJ is an overflow the caller must treat as an error (the reference has no limits; the batch-1 entry reports it itself) */
int airfe_detect_plnet_batch_dev(airfe_ctx* ctx, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, float* d_feat,
                                 int cap, int* d_n, double* d_lines, int capL, int* d_nlines, float* d_junc, int capJ, int* d_njunc,
                                 int junction_images, int* d_found, void* stream);
/* One "stereo detect+match pair" x B with the PLNet detector (≙ map_builder.cc:85-86 with use_superpoint = 0: Detect(L, R, lines, junctions)
 * + MatchingPoints): one detector pass over the 2 B images, lines of all of them (d_lines [2B][capL][4], d_nlines [2B]: left images first),
 * junctions of the left ones only (feature_detector.cc:100-101), LightGlue on the points.  d_found (may be NULL) [3B]. */
int airfe_stereo_plnet_batch_dev(airfe_ctx* ctx, const uint8_t* d_left, const uint8_t* d_right, int B, int h, int w, int stride,
                                 size_t img_stride, float* d_featL, float* d_featR, int cap, int* d_nL, int* d_nR, double* d_lines,
                                 int capL, int* d_nlines, float* d_juncL, int capJ, int* d_njuncL, int* d_found, int32_t* d_idx,
                                 float* d_score, int mcap, int* d_nmatch, void* stream);
/* NEW (no reference counterpart as a batch; the per-frame functions are AssignPointsToLines / MatchLines, src/line_processor.cc:68-180, called
 * right behind Detect / MatchingPoints at src/frame.cc:125,177,184): the same two functions over B device-resident frames, reading the outputs of
 * airfe_detect_plnet_batch_dev / airfe_stereo_plnet_batch_dev / airfe_match_lightglue_batch_dev IN PLACE.  Per frame the results are the bits
 * airfe_assign_points_to_lines / airfe_match_lines return for it.
 *   relation as CSR per frame: d_row_ptr [B][capL + 1], d_pt_idx [B][capE], d_pt_dist [B][capE] doubles (ascending point index per line, like the
 *   reference's std::map<int, double>); d_total (may be NULL) [B] = entries found: a value above capE is an overflow (entries beyond capE are dropped) */
int airfe_assign_points_to_lines_batch_dev(airfe_ctx* ctx, const double* d_lines, const int* d_nlines, int capL, const float* d_feat, const int* d_n,
                                           int cap, int B, int32_t* d_row_ptr, int32_t* d_pt_idx, double* d_pt_dist, int capE, int* d_total,
                                           void* stream);
/*   d_matches [B][mcap][2] + d_nmatch [B]: the matcher's (idx0, idx1) lists; filter3 (HOST, may be NULL) = {min_x_diff, max_x_diff, max_y_diff}: the
 *   disparity band Frame::AddRightFeatures applies to the stereo matches before MatchLines (src/frame.cc:147-160), evaluated on d_feat0 / d_feat1
 *   [B][cap][259]; d_line_matches [B][capL]: index of the matched line of frame 1 or -1, for the first d_nlines0[b] lines of every frame */
int airfe_match_lines_batch_dev(airfe_ctx* ctx, const int32_t* d_row_ptr0, const int32_t* d_pt_idx0, const int* d_nlines0, const int* d_n0,
                                const int32_t* d_row_ptr1, const int32_t* d_pt_idx1, const int* d_nlines1, const int* d_n1, int capL, int capE,
                                const int32_t* d_matches, const int* d_nmatch, int mcap, int B, const double* filter3, const float* d_feat0,
                                const float* d_feat1, int cap, int32_t* d_line_matches, void* stream);
/* NEW (no reference counterpart: the reference copies whole bindings back inside every infer(), src/plnet.cpp:237, buffers.h:237-417): the VALID rows of device
 * result buffers to wherever a kernel can write — device memory or host-mapped pinned memory (hipHostMalloc) — in ONE launch on `stream`.  Job j copies
 * min(*cnt[j], cap[j]) rows (cnt[j] == NULL: cap[j] rows) of row_bytes[j] bytes (a multiple of 4) from src[j] to dst[j]; the counts are read on the device, so the
 * caller needs no synchronisation to learn them first, and only the rows that exist cross PCIe (a batch entry's junction buffer is 1 MB per image at its capacity for
 * ~150 rows of 1 KB).  The five arrays are HOST arrays of njobs entries, read before the call returns.  Asynchronous: the copies are complete when `stream` is. */
int airfe_copy_rows_dev(airfe_ctx* ctx, int njobs, const void* const* src, void* const* dst, const int* const* cnt, const uint32_t* row_bytes,
                        const uint32_t* cap, void* stream);
/* The same, packed: job j's valid rows go to d_packed + d_offsets[j] (16-byte aligned, jobs back to back in order), d_offsets[njobs] = the bytes used; the
 * offsets are an exclusive scan of the counts taken ON THE DEVICE (two launches on `stream`).  d_packed (device) must hold the sum of the jobs' capacities rounded
 * up to 16 bytes each; d_offsets (device) njobs + 1 entries.  For a consumer that wants the results of a large batch on the host through the copy ENGINES: read
 * d_offsets (a few KB), then copy d_packed[0 .. d_offsets[njobs]) with hipMemcpyAsync — no kernel sits on the stream's hardware queue while the bytes cross PCIe
 * (bench.py --io host: one kernel writing the rows into pinned memory itself reaches the same 55 GB/s but runs in line with the next step's kernels wherever the
 * two streams share a hardware queue). */
int airfe_pack_rows_dev(airfe_ctx* ctx, int njobs, const void* const* src, const int* const* cnt, const uint32_t* row_bytes, const uint32_t* cap, void* d_packed,
                        unsigned long long* d_offsets, void* stream);
/* synchronises the context's own stream; also reports (once) a Sinkhorn rendezvous time-out of an earlier SuperGlue call */
int airfe_sync(airfe_ctx* ctx);
/* SuperGlue's error channel for callers of the asynchronous *_dev entries who synchronise their OWN stream: synchronises `stream` (the one the
 * call was issued on; NULL = the context's), then reads and clears the Sinkhorn time-out flag.  0 = the scores of every call so far are valid;
 * 1 = a cooperative rendezvous timed out in one of them (its outputs are NaN / -1), airfe_last_error says so.  The reference has no counterpart:
 * its engine call is synchronous and cannot fail this way (src/super_glue.cpp:185-189). */
int airfe_superglue_status(airfe_ctx* ctx, void* stream);

/* ---- per-stage hipEvent timers (measurement; SURVEY.md §5 "tracing") ----------------------------------- */
/* select + reset: on = 0 off, on < 0 every stage, on > 0 bit mask (bit i = stage i).  A selected stage has each of its
   kernel groups bracketed by two events on the launch stream; an event pair costs ~4 us of stream time (measured: all
   stages on = +8 % per step), so timed runs select only the stage they need. */
int airfe_profile_enable(airfe_ctx* ctx, int on);
int airfe_profile_stages(void);
const char* airfe_profile_stage_name(int i);
/* synchronises, then sums per stage: elapsed ms, algorithmic FLOPs, algorithmic bytes, launch groups; resets */
int airfe_profile_read(airfe_ctx* ctx, double* ms, double* flops, double* bytes, int* launches);

/* The inspection hooks the parity tests use (airfe_debug_*) are NOT part of the product boundary: include/airfe_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* AIRFE_H_ */
