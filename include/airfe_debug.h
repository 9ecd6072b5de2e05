/* airfe_debug.h — inspection and fault-hunting hooks of libairfe.so.  NOT the drop-in boundary (that is include/airfe.h):
 * nothing on the reference's side binds these; they exist so that tests/ can read internal maps back, feed single kernels
 * with hand-built host tensors, and trace the matcher launch by launch.  Exported by the same library; an integrator may
 * ignore this header entirely. */
#ifndef AIRFE_DEBUG_H_
#define AIRFE_DEBUG_H_
#include "airfe.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- inspection hooks used by the parity tests ------------------------------------------------------ */
/* after a detect call with B images: copy internal maps to HOST buffers (NULL = skip).
 *   heat_raw/heat_nms [B][512][512]; desc [B][64][64][256] (NHWC, channel-normalised). */
int airfe_debug_detector_maps(airfe_ctx* ctx, int B, float* heat_raw, float* heat_nms, float* desc);
/* run LightGlue on one HOST pair (258-float rows) and return the full log-assignment scores [n0][n1] */
int airfe_debug_lightglue_scores(airfe_ctx* ctx, const float* f0, int n0, const float* f1, int n1, float* scores);
/* the post-processing kernels alone on HOST score matrices (hand-built ties, -inf rows, threshold-exact values):
 *   filter_matches (src/light_glue.cpp:214-266) on scores [n0][n1]; decode (src/super_glue.cpp:339-367) on Z [n0+1][n1+1] */
int airfe_debug_lg_filter(airfe_ctx* ctx, const float* scores, int n0, int n1, int32_t* idx, float* score, int cap, int* nmatch);
int airfe_debug_sg_decode(airfe_ctx* ctx, const float* Z, int n0, int n1, int32_t* idx0, int32_t* idx1, double* ms0, double* ms1);
/* kernel-level checks on HOST fp32 tensors (test only): NCHW conv3x3(+ReLU, optional 2x2 max-pool) and
 *   y[M][N] = x[M][K] w[N][K]^T + b through the same MFMA kernels the pipelines use. */
/* SuperGlue on one HOST pair ([n][259] rows, normalised x,y): the engine's `scores` output [n0+1][n1+1] */
int airfe_debug_superglue_scores(airfe_ctx* ctx, const float* f0, int n0, const float* f1, int n1, float* scores);
/* the on-device stage-0 line branch of the last detected image in the Appendix A.1 layouts + the junction probability / offset
 *   maps behind juncs_pred (any pointer may be NULL): juncs_pred [300][2], lines_pred [49152][4], iskeep / idx_min / idx_max [49152],
 *   loi [128][128][128], thin / aux [4][128][128], jloc [128][128], joff [2][128][128] */
int airfe_debug_plnet_stage0(airfe_ctx* ctx, float* juncs_pred, float* lines_pred, float* iskeep, float* idx_min, float* idx_max,
                             float* loi, float* thin, float* aux, float* jloc, float* joff);
/* the junction-to-line match (HAWP wireframe_matcher) of the LAST detected image: fast = 1 as the line path runs it (cell search: iskeep
 * exact everywhere, idx_junc_to_end_min / _max exact where iskeep > 0 — all that src/plnet.cpp:272-307 reads), fast = 0 the contract's
 * tensors in full.  [3*128*128] floats each, NULL = skip. */
int airfe_debug_plnet_j2l(airfe_ctx* ctx, int fast, float* iskeep, float* idx_min, float* idx_max);
/* wireframe_matcher + stage-1 LOI head alone: lines_adjusted [cap][4], scores_line [cap], *m2 = unique lines */
int airfe_debug_plnet_s1(airfe_ctx* ctx, const airfe_plnet_stage0* stage0, float* lines_adjusted, float* scores_line,
                         int cap, int* m2);
/* the stage-1 outputs of image 0 of the LAST PLNet call as the line path left them on the device (the device path's own kernel: cfg.line_precision decides which):
 * lines_adjusted [cap][4], scores_line [cap], *m2 = unique candidate lines */
int airfe_debug_plnet_s1_last(airfe_ctx* ctx, float* lines_adjusted, float* scores_line, int cap, int* m2);
/* the pre-process alone (cv::resize + /255, src/plnet.cpp:246-270): HOST gray image -> HOST fp32 [512][512] */
int airfe_debug_preprocess(airfe_ctx* ctx, const uint8_t* gray, int h, int w, int stride, float* out);
int airfe_debug_conv3x3(airfe_ctx* ctx, const float* x, int B, int cin, int H, int W, const float* w, const float* b,
                        int cout, int pool, float* y);
int airfe_debug_gemm(airfe_ctx* ctx, const float* x, int M, int K, const float* w, const float* b, int N, int relu, float* y);
/* the matcher's flash attention alone (kernels_attn.hip) on HOST fp32 tensors, rounded to the matcher's 2-byte type on the way in: q, k [S][H][n][64] with the
 * soft-max scale and log2 e ALREADY inside (the kernel computes p = 2^(q.k - shift)), v [S][H][n][64], lens [S] (keys / queries beyond lens[s] are padding), cross:
 * sequence s attends to sequence s ^ 1.  out [S][n][H*64] fp32.  n <= max_keypoints (rounded up to 16 inside); S * H a multiple of 8.  The way to drive the kernel's
 * re-centring path (a tile whose partial row sums leave the 2-byte range) with hand-built logits. */
int airfe_debug_attention(airfe_ctx* ctx, const float* q, const float* k, const float* v, const int* lens, int S, int H, int n, int cross, float* out);
/* error-path test of cfg.check_launches: the NEXT group of launches of profiling stage `stage` (airfe_profile_stage_name's index) is preceded by one
 * deliberately invalid launch (4096 threads per workgroup), so that the entry that makes it must fail with "<stage name>: kernel launch failed: ..." —
 * with check_launches = 0 the same failure surfaces at the pipeline's end without the stage.  -1 disarms. */
int airfe_debug_fail_next_launch(airfe_ctx* ctx, int stage);
/* fault hunting (tools/experiments/matcher_trace.py): position-dependent 64-bit checksums of the LightGlue forward's state behind EVERY launch
 * (src/light_glue.cpp:120-170 is one opaque engine call; here it is ~60 launches) — residual stream, token shadow, q / k / v^T, attention
 * output, descriptors, similarity, assignment vectors — in units of 16 token rows (v^T: one feature row).  airfe_debug_trace(ctx, 1) switches it on
 * for the following matcher calls (+~2 ms per 64-pair step); _slots / _slot describe the slots of the last call (name, first unit, units,
 * 32-bit words per unit); _read synchronises `stream` (NULL: the context's) and copies one digest per slot and / or the whole unit table. */
int airfe_debug_trace(airfe_ctx* ctx, int on);
/* slot >= 0: the forward pass returns right behind that slot's launch, so that its buffer can be read as the launch left it
 * (airfe_debug_trace_buffer: the first `bytes` bytes of the slot's buffer to the host); -1: run to the end */
int airfe_debug_trace_stop(airfe_ctx* ctx, int slot);
int airfe_debug_trace_buffer(airfe_ctx* ctx, int slot, void* host, size_t bytes);
int airfe_debug_trace_slots(airfe_ctx* ctx);
int airfe_debug_trace_slot(airfe_ctx* ctx, int i, char* name, int name_cap, unsigned* off, unsigned* units, unsigned* unit_words);
int airfe_debug_trace_read(airfe_ctx* ctx, void* stream, unsigned long long* digests, unsigned long long* table);

#ifdef __cplusplus
}
#endif
#endif /* AIRFE_DEBUG_H_ */
