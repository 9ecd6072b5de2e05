/* airfe_seq — the feature thread's per-frame loop over S stereo sequences in lock-step, as a native driver of the airfe entry points.
 *
 * Replaces (as its caller): MapBuilder::ExtractFeatureThread, /root/reference/src/map_builder.cc:55-147 — per frame
 *     keyframe candidate (`!_init || _insert_next_keyframe`, :83-92):  Detect(left, right, features, lines, junctions) + MatchingPoints(left, right)
 *     normal frame (:93-97):                                           Detect(left, features)
 *     every frame once initialised (:99-121):                          MatchingPoints(last keyframe, left) + AddKeyframeCheck (:429-466); a normal frame with
 *                                                                      result 0 is promoted: Detect(right) + MatchingPoints(left, right) (:104-108)
 *     `_last_keyframe_feature = frame` for every frame that is not a normal one (:139-141)
 * with the shipped `use_superpoint: 1` (line 2 of every yaml under configs/visual_odometry): two detector objects, PLNet for keyframes and SuperPoint for normal frames
 * (src/feature_detector.cc:7-34) = the two contexts `kf` and `nf` here.  The reference's loop is C++ and handles one sequence, one frame per iteration; this driver
 * runs S independent sequences per time-step: the sequences are grouped by branch (candidates -> one PLNet stereo batch on a second stream beside the normal
 * frames' SuperPoint batch; all temporal matches -> one LightGlue batch; promotions -> one more detector + matcher batch) through the device-resident
 * airfe_*_batch_dev entries, with ONE host synchronisation per decision point, and everything the host side reads comes back in one packing launch that
 * writes only the valid rows into pinned memory.
 *
 * The keyframe POLICY (AddKeyframeCheck, the stereo count of Frame::AddRightFeatures, src/frame.cc:141-172) is the caller's in the reference; it is restated
 * in the driver (file:line at each function) because the loop cannot run without it.  Not restated: the F-matrix RANSAC behind MatchingPoints(..., true)
 * (src/point_matcher.cc:95-104) and the IMU branches (UseIMU() is false in the VO configurations).
 *
 * Per (sequence, frame) the results are the bytes airslam_amd.seq.SequenceFrontEnd returns through the one-call host entries (tests/test_gpu_seq.py).
 * Same conventions as include/airfe.h: 0 = ok, non-zero = failure with airfe_seq_last_error(); never throws; reads no environment.
 */
#ifndef AIRFE_SEQ_H_
#define AIRFE_SEQ_H_

#include "airfe.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct airfe_seq airfe_seq;

/* configs/visual_odometry/vo_euroc.yaml:16-22 (include/read_configs.h:128-147) + the camera's stereo band (src/camera.cc:50-51, :25) */
typedef struct airfe_seq_policy {
  int min_init_stereo_feature; /* 90 */
  int min_num_match;           /* 30 */
  int max_num_match;           /* 80 */
  float tracking_point_rate;   /* 0.65 */
  float tracking_parallax_rate;/* 0.1 */
  double min_x_diff, max_x_diff, max_y_diff; /* 1, 200, 5 */
  int image_width, image_height;
} airfe_seq_policy;
void airfe_seq_default_policy(airfe_seq_policy* p);
/* The two policy functions the driver applies, callable on their own (pure host functions of HOST arrays; feature rows are [n][259] = (score, x, y, descriptor)):
 *   MapBuilder::AddKeyframeCheck (src/map_builder.cc:429-466, UseIMU() == false): 0 = make this frame a keyframe, 1 = the next one, 2 = neither;
 *     idx [m][2] = (reference index, current index) of the temporal matches;
 *   the count Frame::AddRightFeatures returns (src/frame.cc:141-172): stereo matches inside the camera's band whose signed parallax is inside it too. */
int airfe_seq_add_keyframe_check(const airfe_seq_policy* p, const float* ref_feat, int ref_n, const float* cur_feat, int cur_n, const int32_t* idx, int m);
int airfe_seq_good_stereo_points(const airfe_seq_policy* p, const float* feat_left, const float* feat_right, const int32_t* idx, int m);

/* What one iteration of the loop hands to the tracking thread (TrackingData, map_builder.cc:132-137) + what it decided on the way.  The pointers are
 * into the driver's pinned staging memory: valid until the end of the NEXT airfe_seq_end of the same driver (two staging sets alternate).
 * A count of -1 = "this branch did not run for this frame" (pointer NULL). */
typedef struct airfe_seq_frame {
  int frame_type;         /* 0 normal, 1 keyframe, 2 initial keyframe: FrameType, include/map_builder.h:40-44 */
  int candidate;          /* took the keyframe branch (:83) */
  int promoted;           /* took the promotion branch (:104-108) */
  int dropped;            /* "Not enough stereo points to initialize!" (:122-125): the frame is not handed on */
  int enough_match;       /* AddKeyframeCheck's result; -1: not initialised yet */
  int good_stereo_point;
  int n_left, n_right, n_lines_left, n_lines_right, n_junctions, n_stereo, n_matches;
  const float* features_left;   /* [n_left][259] rows (score, x, y, descriptor) */
  const float* features_right;  /* [n_right][259] */
  const double* lines_left;     /* [n_lines_left][4] */
  const double* lines_right;
  const float* junctions;       /* [n_junctions][259] */
  const int32_t* stereo_idx;    /* [n_stereo][2] (left index, right index) */
  const float* stereo_score;
  const int32_t* matches_idx;   /* temporal: [n_matches][2] (last keyframe's index, this frame's index) */
  const float* matches_score;
} airfe_seq_frame;

/* kf: a context with the PLNet detector (+ stage 1) and LightGlue, max_batch >= S; nf: SuperPoint + LightGlue, max_batch >= S; the same max_keypoints and device.
 * The contexts stay the caller's (destroy them after the driver) and must not be used by anyone else while the driver lives (one stream, one calling thread).
 * cap_lines / cap_junc: rows kept per image; more lines / junctions than that is an error of the step (the reference has no limits).
 * d_tidx [S][max_keypoints][2], d_tscore [S][max_keypoints], d_tn [S] (device, may all be NULL): where the temporal match lists of a time-step are left on the
 * device — row j = the j-th initialised sequence in ascending order — for a caller that forwards them (the K-frame gather of BASELINE configs[3]). */
int airfe_seq_create(airfe_ctx* kf, airfe_ctx* nf, int S, const airfe_seq_policy* policy, int cap_lines, int cap_junc, int32_t* d_tidx, float* d_tscore,
                     int* d_tn, airfe_seq** out);
void airfe_seq_destroy(airfe_seq* s);
const char* airfe_seq_last_error(const airfe_seq* s); /* s may be NULL (creation errors) */

/* One time-step: d_left / d_right = the S left / right images of this time-step on the device (image i at + i * img_stride, rows `stride` bytes apart,
 * h * stride a multiple of 4), out [S].  airfe_seq_step = airfe_seq_begin + airfe_seq_end:
 *   begin  queues the detector batches, the temporal match and the packing of the results (asynchronous);
 *   end    waits for them, takes the decisions, runs the promotions (a second, short device pass + wait), updates the reference rows on the device.
 * Two drivers (each with its own pair of contexts) driven as begin(A) begin(B) end(A) begin(A) end(B) begin(B) ... keep the device busy with one group while
 * the host decides for the other. */
int airfe_seq_begin(airfe_seq* s, const uint8_t* d_left, const uint8_t* d_right, int h, int w, int stride, size_t img_stride);
int airfe_seq_end(airfe_seq* s, airfe_seq_frame* out);
int airfe_seq_step(airfe_seq* s, const uint8_t* d_left, const uint8_t* d_right, int h, int w, int stride, size_t img_stride, airfe_seq_frame* out);
/* the stream the temporal match lists are complete on (for a caller that forwards d_tidx / d_tscore / d_tn: order behind it) */
void* airfe_seq_stream(airfe_seq* s);
/* where a time-step's wall time went since the last call (seconds, summed; host_syncs counted): queueing device work, waiting for the device, the host side of
 * the loop.  Resets the sums. */
int airfe_seq_wall_split(airfe_seq* s, double* queue_s, double* wait_s, double* host_s, int* host_syncs, int* steps);

#ifdef __cplusplus
}
#endif
#endif /* AIRFE_SEQ_H_ */
