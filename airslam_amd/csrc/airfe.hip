// airfe — context life cycle, host staging and the C ABI of libairfe.so (include/airfe.h, include/airfe_debug.h); the pipelines behind it live in
// airfe_load.hip / airfe_detect.hip / airfe_match.hip (see airfe_host.h).
#include "airfe_host.h"

namespace airfe_host {
thread_local std::string g_err;

int fail(airfe_ctx* c, const std::string& m) {
  if (c) c->err = m;
  g_err = m;
  return 1;
}

// the catch blocks of the C boundary end here: nothing in it may throw again
int fail_noexcept(airfe_ctx* c, const char* what, const char* detail) noexcept {
  try {
    std::string m = std::string("airfe: C++ exception at the C boundary (") + what + "): " + (detail ? detail : "unknown");
    if (c) c->err = m;
    g_err = m;
  } catch (...) {
  }
  return 1;
}

int launch_status(airfe_ctx* c) {
  std::string m;
  if (c->launch_err.empty()) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    m = std::string("kernel launch failed: ") + hipGetErrorString(e);
  } else {
    m.swap(c->launch_err);
  }
  // an error path: whatever the call had queued before the failed launch (uploads from the pinned block, kernels on either stream) is drained, so that the
  // next call starts from an idle context
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  return fail(c, m);
}

int saturation_status(airfe_ctx* c) {
  if (!c->sat_host) return 0;
  const int v0 = reinterpret_cast<volatile int*>(c->sat_host)[0], v1 = reinterpret_cast<volatile int*>(c->sat_host)[1];
  if (!v0 && !v1) return 0;
  c->sat_host[0] = c->sat_host[1] = 0;
  return fail(c, std::string("the detector's 2-byte activations left the ") + (c->prec == 1 ? "fp16" : "bf16") + " range (" + (v0 ? "non-finite score logits" : "") +
                     (v0 && v1 ? ", " : "") + (v1 ? "non-finite descriptors" : "") +
                     "): no keypoints are returned for this call.  Re-pack the detector weights with airslam_amd.weights.fold_activation_scales "
                     "(tools/onnx_to_pack.py does it: exact power-of-two rescaling between layers) or run cfg.precision = 2");
}

}  // namespace airfe_host

void note_launch(airfe_ctx* c, int stage) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess || !c->launch_err.empty()) return;
  try { c->launch_err = std::string(kStageNames[stage]) + ": kernel launch failed: " + hipGetErrorString(e); } catch (...) {}
}
namespace { __global__ void never_launched_kernel() {} }
void fail_launch_now(hipStream_t st) { hipLaunchKernelGGL(never_launched_kernel, dim3(1), dim3(4096), 0, st); }      // 4096 threads per workgroup: hipErrorInvalidConfiguration

// Every extern "C" entry is a function-try-block ending in AIRFE_CATCH: std::bad_alloc from a std::vector / std::string of the host-side
// bookkeeping (or anything else thrown below) becomes a non-zero return + airfe_last_error(), never an exception crossing the C ABI.
#define AIRFE_CATCH(c)                                                                                                   \
  catch (const std::exception& e_) { return airfe_host::fail_noexcept((c), __func__, e_.what()); }                       \
  catch (...) { return airfe_host::fail_noexcept((c), __func__, nullptr); }

void KfState::save(const airfe_ctx* c) {
  nms_map_valid = c->nms_map_valid; desc_normalised = c->desc_normalised; desc_dense_valid = c->desc_dense_valid; line_sparse = c->line_sparse; last_B = c->last_B;
}
void KfState::restore(airfe_ctx* c) const {
  c->nms_map_valid = nms_map_valid; c->desc_normalised = desc_normalised; c->desc_dense_valid = desc_dense_valid; c->line_sparse = line_sparse; c->last_B = last_B;
}

namespace {

// grow-on-demand staging block: the previous block is freed (it used to stay in `allocs` until destroy)
// (user: a caller's stream the block's previous contents may still be in use on — the *_batch_dev entries run on the stream they are given)
int ensure_block(airfe_ctx* c, uint8_t*& blk, size_t& have, size_t bytes, hipStream_t user = nullptr) {
  if (bytes <= have) return 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (user && user != c->stream) HIPCHK(c, hipStreamSynchronize(user));
  void* p = nullptr;
  HIPCHK(c, hipMalloc(&p, bytes));
  if (blk) {
    auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)blk);
    if (it != c->allocs.end()) c->allocs.erase(it);
    (void)hipFree(blk);
  }
  c->allocs.push_back(p);
  blk = reinterpret_cast<uint8_t*>(p);
  have = bytes;
  return 0;
}
int ensure_stage_img(airfe_ctx* c, size_t bytes) { return ensure_block(c, c->st_img, c->st_img_bytes, bytes); }

// host image -> c->st_img with the SAME row pitch: exactly (h - 1) * stride + w bytes are read (a cv::Mat ROI / numpy view has no
// bytes behind its last row's w-th pixel that are ours to read)
// the pinned host block (grows by replacement; the stream is idle whenever a host entry starts: they all end with a synchronisation)
int ensure_pin(airfe_ctx* c, size_t bytes) {
  if (bytes <= c->pin_bytes) return 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  uint8_t* p = nullptr;
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault));
  if (c->pin) (void)hipHostFree(c->pin);
  c->pin = p;
  c->pin_bytes = bytes;
  return 0;
}

// A host entry that returns with an error AFTER it queued work must not leave that work in flight: the next call memcpy's into the pinned block a
// pending D2H may still write, and reference rows that were only partly uploaded must not be matched against.  Armed once queueing starts,
// disarmed on success.
// A host entry reports the overflow of ITS OWN call: a saturation word an earlier asynchronous *_batch_dev call left behind (whose caller never asked: airfe_sync /
// airfe_superglue_status) must not fail a later healthy call with a stale "left the fp16 range" (ADVICE r05).  Cleared where a host entry starts queueing.
inline void clear_saturation(airfe_ctx* c) {
  if (c->sat_host) c->sat_host[0] = c->sat_host[1] = 0;
}
struct DrainOnError {
  airfe_ctx* c; bool armed = false, ref_uploaded = false;
  explicit DrainOnError(airfe_ctx* c_, bool armed_ = false) : c(c_), armed(armed_) { clear_saturation(c_); }
  ~DrainOnError() {
    if (!armed) return;
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream2);
    if (ref_uploaded) c->ref_n = -1;
  }
};

int upload_image(airfe_ctx* c, const uint8_t* gray, int h, int w, int stride) {
  if (!gray || h < 1 || w < 1) return fail(c, "empty image");     // plnet.cpp:247
  if (stride < w) return fail(c, "image stride smaller than its width");
  const size_t bytes = (size_t)(h - 1) * stride + w;
  if (ensure_stage_img(c, (size_t)h * stride)) return 1;
  // through the pinned block: a pageable hipMemcpyAsync is staged by the runtime in chunks, synchronously (measured: 2.4x the time)
  if (ensure_pin(c, bytes)) return 1;
  clear_saturation(c);
  memcpy(c->pin, gray, bytes);
  HIPCHK(c, hipMemcpyAsync(c->st_img, c->pin, bytes, hipMemcpyHostToDevice, c->stream));
  return 0;
}

}  // namespace

// ================================================================================== C ABI
extern "C" {

void airfe_default_cfg(airfe_cfg* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->device = 0;
  cfg->precision = 1;              // fp16 storage: what the reference builds its engines with (super_point.cpp:97, plnet.cpp:216)
  cfg->max_batch = 2;
  cfg->enc_chunk = 64;   // measured: per-launch fixed costs dominate below ~16 images (16: -3 %, 32: -1.7 % against 64, 128: +1.7 % on the conv64 stage); no Infinity-Cache benefit from small chunks
  cfg->max_keypoints = 400;        // configs/visual_odometry/vo_euroc.yaml:3-5
  cfg->keypoint_threshold = 0.004f;
  cfg->remove_borders = 4;
  cfg->nms_radius = 4;
  cfg->line_threshold = 0.75f;
  cfg->line_length_threshold = 50.f;
  cfg->matcher = 0;
  cfg->image_width = 752;
  cfg->image_height = 480;
  cfg->sinkhorn_iters = 100;
  cfg->matcher_precision = 1;      // fp16, what the reference builds its matcher engines with (light_glue.cpp:115, super_glue.cpp:132)
  cfg->line_precision = 0;         // stage 1: fp32 operands as fp16 (hi, lo) pairs on the 2-byte matrix pipe; f32-input MFMA in fp32 mode (include/airfe.h)
  cfg->check_launches = 0;
  cfg->tuning = nullptr;
}

void airfe_default_tuning(airfe_tuning* t) {
  if (!t) return;
  int* p = reinterpret_cast<int*>(t);
  for (size_t i = 0; i < sizeof(*t) / sizeof(int); ++i) p[i] = -1;
}

const char* airfe_last_error(const airfe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// Every entry that takes a context makes the context's device current first: a process may hold contexts on several devices (cfg.device)
// and the calling thread's current device is whatever the application left it at.
static inline int enter_device(airfe_ctx* c) {
  int d = -1;
  if (hipGetDevice(&d) == hipSuccess && d == c->cfg.device) return 0;
  if (hipSetDevice(c->cfg.device) != hipSuccess) return fail(c, "hipSetDevice(cfg.device) failed");
  return 0;
}
#define AIRFE_ENTER(c) do { if (!(c)) return 1; if (enter_device(c)) return 1; } while (0)

int airfe_create(const airfe_cfg* cfg, airfe_ctx** out) try {
  if (!cfg || !out) return fail(nullptr, "airfe_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, "airfe_create: no HIP device visible (the product path has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "airfe_create: bad device ordinal");
  if (cfg->max_keypoints < 1 || cfg->max_keypoints > 1024) return fail(nullptr, "airfe_create: max_keypoints must be 1..1024");
  if (cfg->precision < 0 || cfg->precision > 2) return fail(nullptr, "airfe_create: precision must be 0 (bf16), 1 (fp16) or 2 (fp32)");
  if (cfg->matcher_precision < -1 || cfg->matcher_precision > 2)
    return fail(nullptr, "airfe_create: matcher_precision must be -1 (= precision), 0 (bf16), 1 (fp16) or 2 (fp32)");
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, "airfe_create: hipSetDevice failed");
  // (owned by a guard until the very end: an exception below — a weight pack that does not fit the host's memory, say — must not leak the arena)
  struct Guard { airfe_ctx* p; ~Guard() { if (p) airfe_destroy(p); } } guard{new airfe_ctx()};
  airfe_ctx* c = guard.p;
  c->cfg = *cfg;
  c->prec = cfg->precision;
  c->mprec = cfg->matcher_precision < 0 ? cfg->precision : cfg->matcher_precision;
  c->pack_prec = c->prec;
  if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess) c->n_cu = 0;
  c->Bmax = std::max(cfg->max_batch, 1);
  c->chunk = std::min(std::max(cfg->enc_chunk, 1), c->Bmax);
  c->Pmax = c->Bmax;
  c->Np = (cfg->max_keypoints + 15) / 16 * 16;      // matcher rows per sequence: whole 16-token MFMA tiles, no further padding
  if (cfg->line_precision < 0 || cfg->line_precision > 3) { return fail(nullptr, "airfe_create: line_precision must be 2 (fp32 operands, f32-input MFMA) or 3 (fp32 operands as fp16 pairs on the 2-byte MFMA)"); }
  if (c->cfg.line_precision == 0) c->cfg.line_precision = c->prec == 2 ? 2 : 3;      // fp32 mode: every product of the path on f32 operands
  if (c->cfg.line_precision == 1)
    return fail(nullptr, "airfe_create: line_precision = 1 (plain fp16 operands in PLNet stage 1) is refused: measured with the real weights it moves 0.5-0.9 % of the kept "
                         "lines across the 0.75 threshold (profiles/r05_s1_fp16_emulation.txt); line_precision = 3 runs the same products on the 2-byte matrix pipe "
                         "with fp16 (hi, lo) operand pairs and keeps the lines");
  c->cfg.tuning = nullptr;                       // (the caller's struct need not outlive this call)
  if (const airfe_tuning* t = cfg->tuning) {     // kernel-selection overrides: -1 = keep the default
    for (int r : t->reserved)
      if (r != -1) { return fail(nullptr, "airfe_create: airfe_tuning.reserved must be -1 (use airfe_default_tuning)"); }
    if (t->fuse_lg_block >= 0) c->fuse_lg_block = t->fuse_lg_block != 0;
    if (t->gemm_small_max_m >= 0) c->gemm_small_max = t->gemm_small_max_m;
    if (t->gemm8_min_m >= 0) c->gemm8_min = t->gemm8_min_m;
    if (t->gemmr_min_m >= 0) c->gemmr_min = t->gemmr_min_m;
    if (t->gemmr_wgs >= 0) c->gemmr_wgs = t->gemmr_wgs;
    if (t->qkv_pair >= 0) c->qkv_pair = t->qkv_pair != 0;
    if (t->block_min_m >= 0) c->block_min = t->block_min_m;
    if (t->lgb_tokens >= 0) {
      if (t->lgb_tokens != 32 && t->lgb_tokens != 64 && t->lgb_tokens != 112 && t->lgb_tokens != 128) { return fail(nullptr, "airfe_create: tuning.lgb_tokens must be 32, 64, 112 or 128"); }
      c->lgb_tokens = t->lgb_tokens;
    }
    if (t->sg_kenc_gemm >= 0) c->sg_kenc_gemm = t->sg_kenc_gemm != 0;
    if (t->fold_qkv >= 0) c->fold_qkv = t->fold_qkv != 0;
    if (t->overlap_lines >= 0) c->overlap_lines = t->overlap_lines != 0;
    if (t->kf_graph >= 0) c->kf_graph_on = t->kf_graph != 0;
    if (t->kf_spec_rows >= 0) c->kf_spec_lines = c->kf_spec_juncs = std::max(t->kf_spec_rows, 1);   // (tests: force the second round trip)
    if (t->fuse_dec >= 0) c->fuse_dec = t->fuse_dec != 0;
    if (t->assign_fused >= 0) c->assign_fused = t->assign_fused != 0 ? 1 : 0;
    if (t->fold_out_proj >= 0) c->fold_out = t->fold_out_proj != 0;
    if (t->desc_gather_stream >= 0) c->desc_gather_stream = t->desc_gather_stream != 0;
    if (t->copy_wgs > 0) c->copy_wgs = t->copy_wgs;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_feat, hipEventDisableTiming) != hipSuccess)
    return fail(nullptr, "airfe_create: stream creation failed");      // (the guard's airfe_destroy releases whatever was created)
  // a context with a detector AND the stereo matcher runs airfe_stereo_batch_dev over left + right images as one detector batch
  c->Dmax = (c->prec != 2 && cfg->superpoint_pack && cfg->lightglue_pack) ? 2 * c->Bmax : c->Bmax;
  c->chunk = std::min(std::max(cfg->enc_chunk, 1), c->Dmax);   // (a stereo step's 2 B images may go through the first layers as ONE chunk)
  c->Lmax = c->Dmax;
  int rc = 0;
  if (cfg->superpoint_pack) rc = load_superpoint(c, cfg->superpoint_pack);
  if (!rc && cfg->lightglue_pack) rc = load_lightglue(c, cfg->lightglue_pack);
  if (!rc && cfg->superglue_pack) rc = load_superglue(c, cfg->superglue_pack);
  if (!rc && cfg->plnet_s1_pack) rc = load_plnet_s1(c, cfg->plnet_s1_pack);
  if (!rc) {
    const size_t capf = (size_t)c->Np * AIRFE_FEAT_DIM;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->sat_host), 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&c->sat_flag), c->sat_host, 0) != hipSuccess)
      c->sat_flag = nullptr;
    else
      memset(c->sat_host, 0, 64);
    c->io_in = dalloc<uint8_t>(c, 64 + 2 * capf * 4);
    c->io_out = dalloc<uint8_t>(c, 64 + (size_t)c->Np * 12);
    if (!c->io_in || !c->io_out || !c->sat_flag) rc = fail(c, "device allocation failed (staging)");
    else {
      c->st_n0 = reinterpret_cast<int*>(c->io_in);
      c->st_n1 = c->st_n0 + 1;
      c->st_feat0 = reinterpret_cast<float*>(c->io_in + 64);
      c->st_feat1 = c->st_feat0 + capf;
      c->st_nm = reinterpret_cast<int*>(c->io_out);
      c->st_idx = reinterpret_cast<int32_t*>(c->io_out + 64);
      c->st_score = reinterpret_cast<float*>(c->io_out + 64 + (size_t)c->Np * 8);
      if (hipHostMalloc(reinterpret_cast<void**>(&c->pin), 64 + 2 * capf * 4, hipHostMallocDefault) != hipSuccess) rc = fail(c, "hipHostMalloc failed (staging)");
      else c->pin_bytes = 64 + 2 * capf * 4;
    }
  }
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, "device error during weight upload");
  if (rc) {
    g_err = c->err;
    return rc;
  }
  guard.p = nullptr;
  *out = c;
  return 0;
} AIRFE_CATCH(nullptr)

void airfe_destroy(airfe_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  (void)hipDeviceSynchronize();
  c->kf_graph.reset();
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->sat_host) (void)hipHostFree(c->sat_host);
  if (c->copy_ring) (void)hipHostFree(c->copy_ring);
  for (auto& m : c->marks) { (void)hipEventDestroy(m.a); (void)hipEventDestroy(m.b); }
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_feat) (void)hipEventDestroy(c->ev_feat);
  delete c;
}

int airfe_profile_enable(airfe_ctx* c, int on) try {
  AIRFE_ENTER(c);
  HIPCHK(c, hipDeviceSynchronize());             // the events may have been recorded on a caller's stream (the *_dev entry points)
  for (auto& m : c->marks) { c->ev_pool.push_back(m.a); c->ev_pool.push_back(m.b); }
  c->marks.clear();
  c->prof_mask = on < 0 ? 0xFFFFFFFFu : (uint32_t)on;
  return 0;
} AIRFE_CATCH(c)

int airfe_profile_stages(void) { return ST_COUNT; }
const char* airfe_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

int airfe_profile_read(airfe_ctx* c, double* ms, double* flops, double* bytes, int* launches) try {
  AIRFE_ENTER(c);
  HIPCHK(c, hipDeviceSynchronize());
  for (int i = 0; i < ST_COUNT; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
  for (auto& m : c->marks) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, m.a, m.b) == hipSuccess) {
      ms[m.stage] += t; flops[m.stage] += m.flops; bytes[m.stage] += m.bytes; launches[m.stage] += 1;
    }
    c->ev_pool.push_back(m.a); c->ev_pool.push_back(m.b);
  }
  c->marks.clear();
  return 0;
} AIRFE_CATCH(c)

/* fault hunting: checksums of the matcher's state behind every launch of the LightGlue forward (x32, xb, q, k, v^T, attention output, ...)
   in units of 16 token rows; off by default.  airfe_debug_trace_read synchronises the context's stream. */
int airfe_debug_trace(airfe_ctx* c, int on) try {
  AIRFE_ENTER(c);
  if (on && !c->trace_tab) {
    c->trace_cap = (size_t)3 << 20;
    c->trace_tab = dalloc<unsigned long long>(c, c->trace_cap);
    c->trace_dig = dalloc<unsigned long long>(c, 1024);
    c->trace_off = dalloc<unsigned>(c, 1025);
    if (!c->trace_tab || !c->trace_dig || !c->trace_off) return fail(c, "device allocation failed (trace)");
  }
  c->trace_on = on != 0;
  c->trace_slots.clear();
  return 0;
} AIRFE_CATCH(c)
int airfe_debug_trace_stop(airfe_ctx* c, int slot) try {
  AIRFE_ENTER(c);
  c->trace_stop = slot;
  return 0;
} AIRFE_CATCH(c)
int airfe_debug_trace_buffer(airfe_ctx* c, int slot, void* host, size_t bytes) try {
  if (c && enter_device(c)) return 1;
  if (!c || slot < 0 || slot >= (int)c->trace_slots.size()) return fail(c, "trace_buffer: no such slot");
  const auto& t = c->trace_slots[(size_t)slot];
  if (bytes > t.words * 4) return fail(c, "trace_buffer: more bytes than the slot covers");
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(host, t.p, bytes, hipMemcpyDeviceToHost));
  return 0;
} AIRFE_CATCH(c)
int airfe_debug_trace_slots(airfe_ctx* c) { return c ? (int)c->trace_slots.size() : 0; }
int airfe_debug_trace_slot(airfe_ctx* c, int i, char* name, int name_cap, unsigned* off, unsigned* units, unsigned* unit_words) try {
  if (!c || i < 0 || i >= (int)c->trace_slots.size()) return 1;
  const auto& t = c->trace_slots[(size_t)i];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
  if (off) *off = t.off;
  if (units) *units = t.units;
  if (unit_words) *unit_words = t.unit_words;
  return 0;
} AIRFE_CATCH(c)
int airfe_debug_trace_read(airfe_ctx* c, void* stream, unsigned long long* digests, unsigned long long* table) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->trace_tab) return fail(c, "trace is off");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIPCHK(c, hipStreamSynchronize(st));
  const size_t n = c->trace_slots.size();
  if (n == 0) return 0;
  if (digests) HIPCHK(c, hipMemcpy(digests, c->trace_dig, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (table) HIPCHK(c, hipMemcpy(table, c->trace_tab, ((size_t)c->trace_slots.back().off + c->trace_slots.back().units) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
} AIRFE_CATCH(c)

static int sinkhorn_failed(airfe_ctx* c);
int airfe_sync(airfe_ctx* c) try {
  AIRFE_ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (saturation_status(c)) return 1;                     // (the batch entry points are asynchronous: an activation overflow of an earlier call surfaces here)
  return sinkhorn_failed(c);         // (... and a Sinkhorn time-out)
} AIRFE_CATCH(c)

int airfe_superglue_status(airfe_ctx* c, void* stream) try {
  AIRFE_ENTER(c);
  HIPCHK(c, hipStreamSynchronize(stream ? (hipStream_t)stream : c->stream));
  if (saturation_status(c)) return 1;
  return sinkhorn_failed(c);
} AIRFE_CATCH(c)

int airfe_detect_points_batch_dev(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride,
                                  float* d_feat, int cap, int* d_n, void* stream) try {
  AIRFE_ENTER(c);
  return detect_dev(c, d_gray, B, h, w, stride, img_stride, d_feat, cap, d_n, stream ? (hipStream_t)stream : c->stream);
} AIRFE_CATCH(c)

int airfe_detect_points(airfe_ctx* c, const uint8_t* gray, int h, int w, int stride, float* feat, int cap, int* n) try {
  AIRFE_ENTER(c);
  if (cap < c->cfg.max_keypoints) return fail(c, "feature capacity < max_keypoints");
  if (upload_image(c, gray, h, w, stride)) return 1;
  if (detect_dev(c, c->st_img, 1, h, w, stride, (size_t)h * stride, c->st_feat0, c->Np, c->st_n0, c->stream)) return 1;
  // one D2H of [count | max_keypoints feature rows] into the pinned block (st_n0 and st_feat0 are one device block), rows copied out after the sync
  const size_t out_bytes = 64 + (size_t)c->cfg.max_keypoints * AIRFE_FEAT_DIM * 4;
  if (ensure_pin(c, out_bytes)) return 1;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->io_in, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (saturation_status(c)) return 1;
  const int nn = std::min(*reinterpret_cast<const int*>(c->pin), c->cfg.max_keypoints);
  if (nn > 0) memcpy(feat, c->pin + 64, (size_t)nn * AIRFE_FEAT_DIM * 4);
  *n = nn;
  return 0;
} AIRFE_CATCH(c)

/* ---- BoW quantisation behind the path: Database::FrameToBow's per-feature tree descent --------------- */
int airfe_bow_load(airfe_ctx* c, const float* node_desc, const int32_t* first_child, const int32_t* n_children, const int32_t* word_id,
                   const double* weight, int n_nodes) try {
  AIRFE_ENTER(c);
  if (!node_desc || !first_child || !n_children || !word_id || !weight || n_nodes < 1) return fail(c, "bow_load: bad argument");
  for (int i = 0; i < n_nodes; ++i) {                    // the device follows these indices: validate them here, once
    if (n_children[i] < 0 || (n_children[i] > 0 && (first_child[i] <= i || first_child[i] + n_children[i] > n_nodes)))
      return fail(c, "bow_load: children must lie after their parent and inside the node table");
  }
  if (c->bow_nodes) return fail(c, "bow_load: a vocabulary is already loaded in this context");
  std::vector<float> d(node_desc, node_desc + (size_t)n_nodes * 256), w(n_nodes);
  for (int i = 0; i < n_nodes; ++i) w[i] = (float)weight[i];
  std::vector<int> fc(first_child, first_child + n_nodes), nc(n_children, n_children + n_nodes), wi(word_id, word_id + n_nodes);
  c->bow_desc = dupload(c, d); c->bow_weight = dupload(c, w);
  c->bow_first = dupload(c, fc); c->bow_nch = dupload(c, nc); c->bow_word = dupload(c, wi);
  c->bow_out = dalloc<unsigned>(c, 1024); c->bow_outw = dalloc<float>(c, 1024); c->bow_outn = dalloc<int>(c, 1024);
  c->bow_weight_h.assign(weight, weight + n_nodes);
  if (!c->bow_desc || !c->bow_weight || !c->bow_first || !c->bow_nch || !c->bow_word || !c->bow_out || !c->bow_outw || !c->bow_outn)
    return fail(c, "device allocation failed (vocabulary)");
  c->bow_nodes = n_nodes;
  return 0;
} AIRFE_CATCH(c)

int airfe_bow_transform_dev(airfe_ctx* c, const float* d_feat, int N, uint32_t* d_word, float* d_weight, void* stream) try {
  AIRFE_ENTER(c);
  if (!c->bow_nodes) return fail(c, "bow_transform: no vocabulary loaded (airfe_bow_load)");
  if (N < 0 || (N > 0 && (!d_feat || !d_word || !d_weight))) return fail(c, "bow_transform: bad argument");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  ProfScope ps(c, ST_BOW, st, 0, (double)N * 259 * 4);
  launch_bow_transform(d_feat, AIRFE_FEAT_DIM, 3, N, c->bow_desc, c->bow_first, c->bow_nch, c->bow_word, c->bow_weight, d_word, d_weight,
                       d_weight == c->bow_outw ? c->bow_outn : nullptr, st);
  HIPCHK(c, hipGetLastError());
  return 0;
} AIRFE_CATCH(c)

int airfe_bow_transform(airfe_ctx* c, const float* feat, int N, uint32_t* word_of_features, double* weight_of_features) try {
  AIRFE_ENTER(c);
  if (N == 0) return 0;                                  // database.cc:60
  if (N < 0 || N > c->Np || N > 1024 || !feat || !word_of_features) return fail(c, "bow_transform: bad argument / more features than max_keypoints");
  HIPCHK(c, hipMemcpyAsync(c->st_feat0, feat, (size_t)N * AIRFE_FEAT_DIM * 4, hipMemcpyHostToDevice, c->stream));
  if (airfe_bow_transform_dev(c, c->st_feat0, N, c->bow_out, c->bow_outw, c->stream)) return 1;
  std::vector<int> node(N);
  HIPCHK(c, hipMemcpyAsync(word_of_features, c->bow_out, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(node.data(), c->bow_outn, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (weight_of_features)        // WordValue is a double in the reference (3rdparty/DBoW2 BowVector.h): the leaf's own weight, not a float round trip
    for (int i = 0; i < N; ++i) weight_of_features[i] = c->bow_weight_h[(size_t)node[i]];
  return 0;
} AIRFE_CATCH(c)

/* ---- rectification in front of the path: Camera::UndistortImage, src/camera.cc:161-182 ------------- */
int airfe_set_rectify_maps(airfe_ctx* c, int side, const float* mapx, const float* mapy, int h, int w) try {
  AIRFE_ENTER(c);
  if (side < 0 || side > 1 || !mapx || !mapy || h < 1 || w < 1) return fail(c, "set_rectify_maps: bad argument");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)h * w;
  for (int k = 0; k < 2; ++k) {
    if (c->rmap[side][k]) {
      auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)c->rmap[side][k]);
      if (it != c->allocs.end()) c->allocs.erase(it);
      (void)hipFree(c->rmap[side][k]);
    }
    c->rmap[side][k] = dalloc<float>(c, n, false);
    if (!c->rmap[side][k]) return fail(c, "device allocation failed (rectification maps)");
    HIPCHK(c, hipMemcpy(c->rmap[side][k], k ? mapy : mapx, n * 4, hipMemcpyHostToDevice));
  }
  c->rmap_h[side] = h;
  c->rmap_w[side] = w;
  return 0;
} AIRFE_CATCH(c)

int airfe_rectify_batch_dev(airfe_ctx* c, int side, const uint8_t* d_raw, int B, int h, int w, int stride, size_t img_stride,
                            uint8_t* d_rect, int rstride, size_t rimg_stride, void* stream) try {
  AIRFE_ENTER(c);
  if (side < 0 || side > 1 || !c->rmap[side][0]) return fail(c, "rectify: no maps set for this side (airfe_set_rectify_maps)");
  if (h != c->rmap_h[side] || w != c->rmap_w[side]) return fail(c, "rectify: image size differs from the maps'");
  if (stride < w || rstride < w) return fail(c, "rectify: stride smaller than the width");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  ProfScope ps(c, ST_RECTIFY, st, 0, (double)B * h * w * (1 + 8 + 1));          // source pixels + two float maps + rectified pixels
  launch_remap_linear(d_raw, B, h, w, stride, img_stride, c->rmap[side][0], c->rmap[side][1], d_rect, rstride, rimg_stride, st);
  HIPCHK(c, hipGetLastError());
  return 0;
} AIRFE_CATCH(c)

/* raw HOST image -> rectified image (HOST, tight rows, may be NULL) + point features of the RECTIFIED image: the rectified
   image never leaves the device on its way into the detector */
int airfe_rectify_detect_points(airfe_ctx* c, int side, const uint8_t* raw, int h, int w, int stride, uint8_t* rect_out, float* feat,
                                int cap, int* n) try {
  AIRFE_ENTER(c);
  if (feat && cap < c->cfg.max_keypoints) return fail(c, "feature capacity < max_keypoints");
  if (upload_image(c, raw, h, w, stride)) return 1;
  if (ensure_block(c, c->st_rect, c->st_rect_bytes, (size_t)h * w)) return 1;
  if (airfe_rectify_batch_dev(c, side, c->st_img, 1, h, w, stride, (size_t)h * stride, c->st_rect, w, (size_t)h * w, c->stream)) return 1;
  int nn = 0;
  if (feat) {
    if (detect_dev(c, c->st_rect, 1, h, w, w, (size_t)h * w, c->st_feat0, c->Np, c->st_n0, c->stream)) return 1;
    HIPCHK(c, hipMemcpyAsync(&nn, c->st_n0, 4, hipMemcpyDeviceToHost, c->stream));
  }
  if (rect_out) HIPCHK(c, hipMemcpyAsync(rect_out, c->st_rect, (size_t)h * w, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (feat && saturation_status(c)) return 1;
  if (feat && nn > 0) HIPCHK(c, hipMemcpy(feat, c->st_feat0, (size_t)nn * AIRFE_FEAT_DIM * 4, hipMemcpyDeviceToHost));
  if (n) *n = nn;
  return 0;
} AIRFE_CATCH(c)

int airfe_debug_detector_maps(airfe_ctx* c, int B, float* heat_raw, float* heat_nms, float* desc) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_sp || B > c->Dmax) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t R = AIRFE_INTERNAL_SIZE;
  if (heat_raw) HIPCHK(c, hipMemcpy(heat_raw, c->heat, (size_t)B * R * R * 4, hipMemcpyDeviceToHost));
  if (heat_nms && c->cfg.nms_radius > 0 && !c->nms_map_valid) {      // the batch path skipped the dense map: rebuild it from the heat maps
    launch_nms512_candidates(c->heat, c->heat_nms, c->nms_mask, (int)B, c->cfg.keypoint_threshold, c->cfg.remove_borders, c->cand, c->cand_cnt,
                             R * R, c->stream);          // (the map is only ever skipped on the radius-4 path)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nms_map_valid = true;
  }
  if (heat_nms) HIPCHK(c, hipMemcpy(heat_nms, c->cfg.nms_radius > 0 ? c->heat_nms : c->heat, (size_t)B * R * R * 4, hipMemcpyDeviceToHost));
  if (desc) {
    if (!c->desc_dense_valid) {     // the batch path ran the head on the sampled cells only: build the dense map from the kept activations
      dense_desc_head(c, c->last_B, c->stream);
      c->desc_dense_valid = true;
    }
    if (!c->desc_normalised) {      // the inspection hook returns the map the reference would hold: normalised
      launch_l2norm256(c->desc, c->Dmax * 64 * 64, c->stream);
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->desc_normalised = true;
    }
    HIPCHK(c, hipMemcpy(desc, c->desc, (size_t)B * 64 * 64 * 256 * 4, hipMemcpyDeviceToHost));
  }
  return 0;
} AIRFE_CATCH(c)

static int lg_host(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, int32_t* idx, float* score, int cap,
                   int* nmatch, float* scores_full) {
  AIRFE_ENTER(c);
  if (n0 < 1 || n1 < 1) { if (nmatch) *nmatch = 0; return 0; }   // point_matcher.cc:53-55
  if (n0 > c->cfg.max_keypoints || n1 > c->cfg.max_keypoints) return fail(c, "keypoint count exceeds max_keypoints");
  // ONE H2D: [n0 n1 | n0 rows of f0 | n1 rows of f1] assembled in the pinned block (f1 sits right behind f0's rows on the device too)
  const size_t b0 = (size_t)n0 * 258 * 4, b1 = (size_t)n1 * 258 * 4;
  if (ensure_pin(c, 64 + b0 + b1)) return 1;
  reinterpret_cast<int*>(c->pin)[0] = n0;
  reinterpret_cast<int*>(c->pin)[1] = n1;
  memcpy(c->pin + 64, f0, b0);
  memcpy(c->pin + 64 + b0, f1, b1);
  HIPCHK(c, hipMemcpyAsync(c->io_in, c->pin, 64 + b0 + b1, hipMemcpyHostToDevice, c->stream));
  const float* d_f1 = c->st_feat0 + (size_t)n0 * 258;
  if (lightglue_dev(c, c->st_feat0, c->st_n0, d_f1, c->st_n1, 1, c->Np, 258, 0, 0, c->st_idx, c->st_score, c->Np,
                    c->st_nm, scores_full ? c->st_scores_full : nullptr, c->stream))
    return 1;
  // ONE D2H: [nmatch | idx | score] (a few KB whatever the count), the rows copied out after the synchronisation
  const size_t ob = 64 + (size_t)c->Np * 12;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->io_out, ob, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int nm = *reinterpret_cast<const int*>(c->pin);
  if (idx && score) {
    nm = std::min(nm, cap);
    if (nm > 0) {
      memcpy(idx, c->pin + 64, (size_t)nm * 8);
      memcpy(score, c->pin + 64 + (size_t)c->Np * 8, (size_t)nm * 4);
    }
  }
  if (nmatch) *nmatch = nm;
  if (scores_full)
    for (int i = 0; i < n0; ++i)
      HIPCHK(c, hipMemcpy(scores_full + (size_t)i * n1, c->st_scores_full + (size_t)i * c->Np, (size_t)n1 * 4, hipMemcpyDeviceToHost));
  return 0;
}

int airfe_match_lightglue(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, int32_t* idx, float* score,
                          int cap, int* nmatch) try {
  if (c && enter_device(c)) return 1;
  return lg_host(c, f0, n0, f1, n1, idx, score, cap, nmatch, nullptr);
} AIRFE_CATCH(c)

int airfe_debug_lightglue_scores(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, float* scores) try {
  if (c && enter_device(c)) return 1;
  return lg_host(c, f0, n0, f1, n1, nullptr, nullptr, 0, nullptr, scores);
} AIRFE_CATCH(c)

/* filter_matches (src/light_glue.cpp:214-266) alone on one HOST score matrix [n0][n1] */
int airfe_debug_lg_filter(airfe_ctx* c, const float* scores, int n0, int n1, int32_t* idx, float* score, int cap, int* nmatch) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_arena) return fail(c, "debug_lg_filter: no matcher loaded");
  if (n0 < 1 || n1 < 1 || n0 > c->cfg.max_keypoints || n1 > c->cfg.max_keypoints || !scores || !idx || !score || !nmatch)
    return fail(c, "debug_lg_filter: bad argument");
  const int lens[2] = {n0, n1};
  HIPCHK(c, hipMemcpyAsync(c->lens, lens, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->simbuf, (size_t)c->Np * 4, scores, (size_t)n1 * 4, (size_t)n1 * 4, n0, hipMemcpyHostToDevice, c->stream));
  launch_lg_filter_scores(c->simbuf, c->lens, 1, c->Np, c->Np, 0.1f, c->rowarg, c->rowval, c->colarg, c->st_idx, c->st_score,
                          c->st_nm, c->stream);
  int nm = 0;
  HIPCHK(c, hipMemcpyAsync(&nm, c->st_nm, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  nm = std::min(nm, cap);
  if (nm > 0) {
    HIPCHK(c, hipMemcpy(idx, c->st_idx, (size_t)nm * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(score, c->st_score, (size_t)nm * 4, hipMemcpyDeviceToHost));
  }
  *nmatch = nm;
  return 0;
} AIRFE_CATCH(c)

int airfe_match_lightglue_batch_dev(airfe_ctx* c, const float* d_f0, const int* d_n0, const float* d_f1, const int* d_n1,
                                    int B, int cap, int32_t* d_idx, float* d_score, int mcap, int* d_nmatch, void* stream) try {
  AIRFE_ENTER(c);
  return lightglue_dev(c, d_f0, d_n0, d_f1, d_n1, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr,
                       stream ? (hipStream_t)stream : c->stream);
} AIRFE_CATCH(c)

int airfe_stereo_batch_dev(airfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int B, int h, int w, int stride,
                           size_t img_stride, float* d_featL, float* d_featR, int cap, int* d_nL, int* d_nR, int32_t* d_idx,
                           float* d_score, int mcap, int* d_nmatch, void* stream) try {
  AIRFE_ENTER(c);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (c->prec != 2 && 2 * B <= c->Dmax) {        // left and right images as ONE detector batch
    if (detect_dev2(c, d_left, d_right, B, h, w, stride, img_stride, d_featL, d_featR, cap, d_nL, d_nR, st)) return 1;
  } else {
    if (detect_dev(c, d_left, B, h, w, stride, img_stride, d_featL, cap, d_nL, st)) return 1;
    if (detect_dev(c, d_right, B, h, w, stride, img_stride, d_featR, cap, d_nR, st)) return 1;
  }
  return lightglue_dev(c, d_featL, d_nL, d_featR, d_nR, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr, st);
} AIRFE_CATCH(c)

int airfe_assign_points_to_lines(airfe_ctx* c, const double* lines, int L, const float* feat, int N, int32_t* row_ptr,
                                 int32_t* pt_idx, double* pt_dist, int cap, int* total) try {
  AIRFE_ENTER(c);
  if (L < 0 || N < 0 || cap < 0 || !row_ptr || !total) return fail(c, "assign_points_to_lines: bad argument");
  *total = 0;
  if (L == 0) { row_ptr[0] = 0; return 0; }
  if (!lines || (N > 0 && !feat)) return fail(c, "assign_points_to_lines: null input");
  // staging grows on demand (lines and points per frame are a few hundred); the kernels are the batch entry's with one frame
  const size_t need = (size_t)L * 32 + (size_t)std::max(N, 1) * 259 * 4 + (size_t)(2 * L + 2) * 4 + (size_t)std::max(cap, 1) * 12 + 128;
  if (ensure_block(c, c->pl_stage, c->pl_bytes, need)) return 1;      // grows by replacing (and freeing) the previous block
  char* q = reinterpret_cast<char*>(c->pl_stage);
  double* d_lines = reinterpret_cast<double*>(q); q += (size_t)L * 32;
  double* d_dist = reinterpret_cast<double*>(q); q += (size_t)std::max(cap, 1) * 8;
  float* d_feat = reinterpret_cast<float*>(q); q += (size_t)std::max(N, 1) * 259 * 4;
  int* d_counts = reinterpret_cast<int*>(q); q += (size_t)L * 4;
  int* d_rowptr = reinterpret_cast<int*>(q); q += (size_t)(L + 1) * 4;
  int* d_idx = reinterpret_cast<int*>(q); q += (size_t)std::max(cap, 1) * 4;
  int* d_cnt = reinterpret_cast<int*>(q);                              // {nlines, npts}
  hipStream_t st = c->stream;
  const int cnt[2] = {L, N};
  HIPCHK(c, hipMemcpyAsync(d_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_lines, lines, (size_t)L * 32, hipMemcpyHostToDevice, st));
  if (N > 0) HIPCHK(c, hipMemcpyAsync(d_feat, feat, (size_t)N * 259 * 4, hipMemcpyHostToDevice, st));
  PlAssignArgs a;
  a.lines = d_lines; a.nlines = d_cnt; a.feat = d_feat; a.npts = d_cnt + 1; a.capL = L; a.cap = std::max(N, 1); a.capE = cap;
  a.counts = d_counts; a.row_ptr = d_rowptr; a.pt_idx = d_idx; a.pt_dist = d_dist;
  launch_assign_points_to_lines(a, 1, st);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(row_ptr, d_rowptr, (size_t)(L + 1) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *total = row_ptr[L];
  if (*total > cap) return fail(c, "assign_points_to_lines: output capacity too small");
  if (*total > 0) {
    HIPCHK(c, hipMemcpy(pt_idx, d_idx, (size_t)*total * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(pt_dist, d_dist, (size_t)*total * 8, hipMemcpyDeviceToHost));
  }
  return 0;
} AIRFE_CATCH(c)

// NEW (VERDICT r03 missing #5): the same over B frames whose lines and features are ALREADY on the device — the outputs of
// airfe_detect_plnet_batch_dev / airfe_stereo_plnet_batch_dev, in place (the reference calls AssignPointsToLines right after Detect: src/frame.cc:125,177)
int airfe_assign_points_to_lines_batch_dev(airfe_ctx* c, const double* d_lines, const int* d_nlines, int capL, const float* d_feat, const int* d_n,
                                           int cap, int B, int32_t* d_row_ptr, int32_t* d_pt_idx, double* d_pt_dist, int capE, int* d_total,
                                           void* stream) try {
  AIRFE_ENTER(c);
  if (B < 1 || capL < 1 || cap < 1 || capE < 1 || !d_lines || !d_nlines || !d_feat || !d_n || !d_row_ptr || !d_pt_idx || !d_pt_dist)
    return fail(c, "assign_points_to_lines_batch_dev: bad argument");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  // scratch of THIS entry (airfe_match_lines_batch_dev has its own: the two may be in flight on different streams); it grows only behind a
  // synchronisation of the stream it was last used on.  One stream at a time per entry and context — the contract of every *_dev entry (one ctx = one calling thread).
  if (ensure_block(c, c->pl_scratch, c->pl_scratch_bytes, (size_t)B * capL * 4, c->pl_scratch_stream)) return 1;
  c->pl_scratch_stream = st;
  PlAssignArgs a;
  a.lines = d_lines; a.nlines = d_nlines; a.feat = d_feat; a.npts = d_n; a.capL = capL; a.cap = cap; a.capE = capE;
  a.counts = reinterpret_cast<int*>(c->pl_scratch); a.row_ptr = d_row_ptr; a.pt_idx = d_pt_idx; a.pt_dist = d_pt_dist; a.total = d_total;
  ProfScope ps(c, ST_LINE_ASSOC, st, 0, (double)B * ((double)capL * 32 + (double)cap * 8));
  launch_assign_points_to_lines(a, B, st);
  HIPCHK(c, hipGetLastError());
  return 0;
} AIRFE_CATCH(c)

int airfe_match_lines(airfe_ctx* c, const int32_t* row_ptr0, const int32_t* pt_idx0, int L0, int point_num0, const int32_t* row_ptr1,
                      const int32_t* pt_idx1, int L1, int point_num1, const int32_t* matches, int M, int32_t* line_matches) try {
  AIRFE_ENTER(c);
  if (L0 < 0 || L1 < 0 || M < 0 || point_num0 < 0 || point_num1 < 0 || (L0 > 0 && !line_matches)) return fail(c, "match_lines: bad argument");
  for (int i = 0; i < L0; ++i) line_matches[i] = -1;                                  // line_processor.cc:127-131
  if (point_num0 == 0 || point_num1 == 0 || L0 == 0 || L1 == 0) return 0;            // :132
  if (!row_ptr0 || !row_ptr1 || (M > 0 && !matches)) return fail(c, "match_lines: null input");
  const int t0 = row_ptr0[L0], t1 = row_ptr1[L1];
  if (t0 < 0 || t1 < 0 || (t0 > 0 && !pt_idx0) || (t1 > 0 && !pt_idx1)) return fail(c, "match_lines: bad relation");
  // the device indexes with these: CSR rows must start at 0 and not decrease, point indices must be inside the frame's points
  auto csr_ok = [](const int32_t* rp, const int32_t* pi, int L, int npts) {
    if (rp[0] != 0) return false;
    for (int i = 0; i < L; ++i)
      if (rp[i + 1] < rp[i]) return false;
    for (int e = 0; e < rp[L]; ++e)
      if (pi[e] < 0 || pi[e] >= npts) return false;
    return true;
  };
  if (!csr_ok(row_ptr0, pt_idx0, L0, point_num0) || !csr_ok(row_ptr1, pt_idx1, L1, point_num1))
    return fail(c, "match_lines: relation is not a valid CSR (row_ptr must start at 0 and be non-decreasing, pt_idx within [0, point_num))");
  for (int m = 0; m < M; ++m)                                                         // the reference indexes vectors with these
    if (matches[2 * m] < 0 || matches[2 * m] >= point_num0 || matches[2 * m + 1] < 0 || matches[2 * m + 1] >= point_num1)
      return fail(c, "match_lines: point match index out of range");
  // the batch entry's kernels with one frame pair: one line capacity for both sides, the relation capacity = the larger relation
  const int capL = std::max(L0, L1), capE = std::max(std::max(t0, t1), 1), mcap = std::max(M, 1);
  const int W = (mcap + 31) / 32;
  const size_t words = 2 * (size_t)(capL + 1) + 2 * (size_t)capE + (size_t)mcap * 2 + 2 * (size_t)capL * W + 2 * (size_t)capL + 8;
  if (ensure_block(c, c->pl_stage, c->pl_bytes, words * 4 + 64)) return 1;      // grows by replacing (and freeing) the previous block
  int* q = reinterpret_cast<int*>(c->pl_stage);
  int* d_rp0 = q; q += capL + 1;
  int* d_rp1 = q; q += capL + 1;
  int* d_pi0 = q; q += capE;
  int* d_pi1 = q; q += capE;
  int* d_m = q; q += (size_t)mcap * 2;
  unsigned* d_b0 = reinterpret_cast<unsigned*>(q); q += (size_t)capL * W;
  unsigned* d_b1 = reinterpret_cast<unsigned*>(q); q += (size_t)capL * W;
  int* d_rloc = q; q += capL;
  int* d_lm = q; q += capL;
  int* d_cnt = q;                                                        // {L0, L1, point_num0, point_num1, M}
  hipStream_t st = c->stream;
  const int cnt[5] = {L0, L1, point_num0, point_num1, M};
  HIPCHK(c, hipMemcpyAsync(d_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_rp0, row_ptr0, (size_t)(L0 + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d_rp1, row_ptr1, (size_t)(L1 + 1) * 4, hipMemcpyHostToDevice, st));
  if (t0 > 0) HIPCHK(c, hipMemcpyAsync(d_pi0, pt_idx0, (size_t)t0 * 4, hipMemcpyHostToDevice, st));
  if (t1 > 0) HIPCHK(c, hipMemcpyAsync(d_pi1, pt_idx1, (size_t)t1 * 4, hipMemcpyHostToDevice, st));
  if (M > 0) HIPCHK(c, hipMemcpyAsync(d_m, matches, (size_t)M * 8, hipMemcpyHostToDevice, st));
  MlArgs a;
  a.row_ptr0 = d_rp0; a.pt_idx0 = d_pi0; a.nlines0 = d_cnt; a.npts0 = d_cnt + 2;
  a.row_ptr1 = d_rp1; a.pt_idx1 = d_pi1; a.nlines1 = d_cnt + 1; a.npts1 = d_cnt + 3;
  a.matches = d_m; a.nmatch = d_cnt + 4; a.capL = capL; a.capE = capE; a.mcap = mcap; a.W = W;
  a.bits0 = d_b0; a.bits1 = d_b1; a.row_loc = d_rloc; a.line_matches = d_lm;
  launch_match_lines(a, 1, st);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(line_matches, d_lm, (size_t)L0 * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return 0;
} AIRFE_CATCH(c)

// NEW: MatchLines over B frame pairs on the device, straight from the relations of airfe_assign_points_to_lines_batch_dev and the match lists of the
// matcher's batch entries (src/frame.cc:184 calls it right behind the stereo match).  filter3 != NULL = {min_x_diff, max_x_diff, max_y_diff}: the
// disparity band Frame::AddRightFeatures applies to the stereo matches first (src/frame.cc:147-160; the camera's numbers, host memory).
int airfe_match_lines_batch_dev(airfe_ctx* c, const int32_t* d_row_ptr0, const int32_t* d_pt_idx0, const int* d_nlines0, const int* d_n0,
                                const int32_t* d_row_ptr1, const int32_t* d_pt_idx1, const int* d_nlines1, const int* d_n1, int capL, int capE,
                                const int32_t* d_matches, const int* d_nmatch, int mcap, int B, const double* filter3, const float* d_feat0,
                                const float* d_feat1, int cap, int32_t* d_line_matches, void* stream) try {
  AIRFE_ENTER(c);
  if (B < 1 || capL < 1 || capE < 1 || mcap < 1 || !d_row_ptr0 || !d_pt_idx0 || !d_nlines0 || !d_n0 || !d_row_ptr1 || !d_pt_idx1 || !d_nlines1 ||
      !d_n1 || !d_matches || !d_nmatch || !d_line_matches || (filter3 && (!d_feat0 || !d_feat1 || cap < 1)))
    return fail(c, "match_lines_batch_dev: bad argument");
  const int W = (mcap + 31) / 32;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  // bit rows [2][B][capL][W] + row maxima [B][capL] (no vote matrix: the column pass recounts from the bit rows) — 7 MB at 64 frames x 1024 line slots
  const size_t words = 2 * (size_t)B * capL * W + (size_t)B * capL;
  if (ensure_block(c, c->ml_scratch, c->ml_scratch_bytes, words * 4, c->ml_scratch_stream)) return 1;
  c->ml_scratch_stream = st;
  unsigned* q = reinterpret_cast<unsigned*>(c->ml_scratch);
  MlArgs a;
  a.row_ptr0 = d_row_ptr0; a.pt_idx0 = d_pt_idx0; a.nlines0 = d_nlines0; a.npts0 = d_n0;
  a.row_ptr1 = d_row_ptr1; a.pt_idx1 = d_pt_idx1; a.nlines1 = d_nlines1; a.npts1 = d_n1;
  a.matches = d_matches; a.nmatch = d_nmatch; a.capL = capL; a.capE = capE; a.mcap = mcap; a.W = W; a.cap = cap;
  if (filter3) { a.filter_on = 1; a.min_x_diff = filter3[0]; a.max_x_diff = filter3[1]; a.max_y_diff = filter3[2]; a.feat0 = d_feat0; a.feat1 = d_feat1; }
  a.bits0 = q; q += (size_t)B * capL * W;
  a.bits1 = q; q += (size_t)B * capL * W;
  a.row_loc = reinterpret_cast<int*>(q);
  a.line_matches = d_line_matches;
  ProfScope ps(c, ST_LINE_ASSOC, st, 0, (double)B * (double)capL * W * 8);
  launch_match_lines(a, B, st);
  HIPCHK(c, hipGetLastError());
  return 0;
} AIRFE_CATCH(c)

int airfe_has_line_branch(const airfe_ctx* c) { return c && c->has_s0 && c->has_s1; }

// caller-supplied stage-0 tensors (golden / known-answer tests of everything downstream) -> stage slot 0 + the CHW LOI block
static int upload_stage0(airfe_ctx* c, const airfe_plnet_stage0* s0, hipStream_t st) {
  c->wf_counted = false;               // host tensors: nothing has counted their kept proposals
  const size_t NP = KEEP_CAP;
  float* d = c->s0_stage;
  HIPCHK(c, hipMemcpyAsync(d + SG_JUNCS, s0->juncs_pred, 600 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_LP, s0->lines_pred, NP * 16, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_KEEP, s0->iskeep, NP * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_MIN, s0->idx_junc_to_end_min, NP * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_MAX, s0->idx_junc_to_end_max, NP * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->s0_loi, s0->loi_features, (size_t)128 * 128 * 128 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_THIN, s0->loi_features_thin, (size_t)4 * 128 * 128 * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(d + SG_AUX, s0->loi_features_aux, (size_t)4 * 128 * 128 * 4, hipMemcpyHostToDevice, st));
  return 0;
}

int airfe_detect_plnet(airfe_ctx* c, const uint8_t* gray, int h, int w, int stride, const airfe_plnet_stage0* s0, float* feat,
                       int cap, int* n, double* lines, int capL, int* nlines, float* junc, int capJ, int* njunc,
                       int want_junctions) try {
  AIRFE_ENTER(c);
  if (nlines) *nlines = 0;
  if (njunc) *njunc = 0;
  const bool lines_on = (s0 || c->has_s0);
  if (!lines_on) return airfe_detect_points(c, gray, h, w, stride, feat, cap, n);   // no line branch available: points only (the shim says so, loudly, at build())
  if (!c->has_s1) return fail(c, "PLNet stage-1 weights were not loaded (cfg.plnet_s1_pack)");
  if (cap < c->cfg.max_keypoints) return fail(c, "feature capacity < max_keypoints");
  // Everything is QUEUED before the first synchronisation (round 4; before: the point branch was synchronised and copied out first): image up,
  // point branch (plnet.cpp:560), line branch + tail, then [count | feature rows] and the three line / junction counts come back in one wait.
  hipStream_t st = c->stream;
  if (upload_image(c, gray, h, w, stride)) return 1;
  if (detect_dev(c, c->st_img, 1, h, w, stride, (size_t)h * stride, c->st_feat0, c->Np, c->st_n0, st)) return 1;
  if (s0) {
    if (upload_stage0(c, s0, st)) return 1;
  } else if (line_branch_dev(c, st, 0, 1, false)) {   // the stage-0 line branch on the device: nothing crosses PCIe (the reference moves
    return 1;                                         // 15.5 MB D2H + 9.7 MB H2D here, plnet.cpp:237,494-509)
  }
  int* nl_d = c->d_nlines;                            // [0] kept (<= LINE_CAP: the staging holds every candidate), [Lmax] found
  int *nj_d = c->d_njunc, *njf_d = c->d_njunc + c->Lmax;
  if (line_tail_dev(c, 0, 1, s0 ? c->s0_loi : nullptr, h, w, c->d_lines, LINE_CAP, nl_d, nl_d + c->Lmax, c->junc_feat, JUNC_CAP, nj_d, njf_d,
                    want_junctions ? 1 : 0, st))
    return 1;
  const size_t fbytes = 64 + (size_t)c->cfg.max_keypoints * AIRFE_FEAT_DIM * 4;
  if (ensure_pin(c, fbytes + 64)) return 1;
  int* cnt = reinterpret_cast<int*>(c->pin + fbytes);            // pinned: {lines kept, junctions, junctions found}
  cnt[0] = cnt[1] = cnt[2] = 0;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->io_in, fbytes, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(cnt, nl_d, 4, hipMemcpyDeviceToHost, st));
  if (want_junctions) {
    HIPCHK(c, hipMemcpyAsync(cnt + 1, nj_d, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(cnt + 2, njf_d, 4, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  if (saturation_status(c)) return 1;
  const int nn = std::min(*reinterpret_cast<const int*>(c->pin), c->cfg.max_keypoints);
  if (nn > 0) memcpy(feat, c->pin + 64, (size_t)nn * AIRFE_FEAT_DIM * 4);
  *n = nn;
  const int nl = cnt[0], nj = cnt[1], njf = cnt[2];
  // the reference has no junction limit (junction_detector, plnet.cpp:425-448): more than the arena holds is an ERROR, not a shorter list
  if (njf > JUNC_CAP) return fail(c, "detect_plnet: more junctions than the device arena holds (JUNC_CAP)");
  if (nl > capL || nj > capJ) return fail(c, "detect_plnet: lines / junctions do not fit the caller's buffers (capL, capJ)");
  // second (and last) round trip: the line and junction rows, whose counts are known only now, through the pinned block
  const size_t lb = (nl > 0 && lines) ? (size_t)nl * 32 : 0, jb = (nj > 0 && junc) ? (size_t)nj * AIRFE_FEAT_DIM * 4 : 0;
  if (lb + jb) {
    if (ensure_pin(c, lb + jb)) return 1;
    if (lb) HIPCHK(c, hipMemcpyAsync(c->pin, c->d_lines, lb, hipMemcpyDeviceToHost, st));
    if (jb) HIPCHK(c, hipMemcpyAsync(c->pin + lb, c->junc_feat, jb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (lb) memcpy(lines, c->pin, lb);
    if (jb) memcpy(junc, c->pin + lb, jb);
  }
  if (nlines) *nlines = nl;
  if (njunc) *njunc = nj;
  return 0;
} AIRFE_CATCH(c)

// PLNet::infer over a device-resident batch, after the point branch ran on it (images 0 .. B-1 of the detector arena): lines of every
// image, junctions of the first nj
static int plnet_lines_batch(airfe_ctx* c, int B, int h, int w, double* d_lines, int capL, int* d_nlines, float* d_junc, int capJ,
                             int* d_njunc, int nj, int* d_found, hipStream_t st, int phase = 3) {
  if (!c->has_s0 || !c->has_s1) return fail(c, "the batched PLNet path needs the line branch (line.* in the detector pack) and cfg.plnet_s1_pack");
  if (c->prec == 2) return fail(c, "plnet_lines_batch is the 2-byte batch path (fp32 mode goes image by image: plnet_batch_f32)");
  if (B > c->Lmax) return fail(c, "batch exceeds the line-path arena");
  if (capL < 1 || !d_lines || !d_nlines) return fail(c, "detect_plnet_batch: no line output");
  if (nj < 0 || nj > B || (nj > 0 && (!d_junc || !d_njunc || capJ < 1))) return fail(c, "detect_plnet_batch: bad junction arguments");
  if ((phase & 1) && line_branch_dev(c, st, 0, B, false)) return 1;
  bool partial = false;
  if ((phase & 2) && nj > 0 && !c->desc_dense_valid) {   // the point branch ran the descriptor head on the sampled cells only: the junction images'
    dense_desc_head(c, nj, st);                   // dense maps now (the points have theirs already; c->desc is free to be rewritten)
    c->desc_dense_valid = true;
    partial = nj != c->last_B;
  }
  int* lfound = d_found ? d_found : c->d_nlines + c->Lmax;
  int* jfound = d_found ? d_found + B : c->d_njunc + c->Lmax;
  const int rc = line_tail_dev(c, 0, B, nullptr, h, w, d_lines, capL, d_nlines, lfound, d_junc, capJ, d_njunc, jfound, nj, st, phase);
  if (partial) c->desc_dense_valid = false;       // (only the first nj images' dense maps exist)
  return rc;
}

// The batched PLNet entries in fp32 mode (cfg.precision = 2, round 6): the same results through the one-image path, image by image — the fp32 line head works
// from ONE image's fp32 conv3a activations (line_branch_dev's fused form), so every image runs encoder -> point branch -> line branch -> tail before the next
// one starts.  A correctness mode: nothing here is batched for speed.  lfound / jfound: per image "found" counts (nullptr: the context's scratch words).
static int plnet_batch_f32(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, float* d_feat, int cap, int* d_n,
                           double* d_lines, int capL, int* d_nlines, float* d_junc, int capJ, int* d_njunc, int nj, int* lfound, int* jfound, hipStream_t st) {
  if (!c->has_s0 || !c->has_s1) return fail(c, "the batched PLNet path needs the line branch (line.* in the detector pack) and cfg.plnet_s1_pack");
  if (capL < 1 || !d_lines || !d_nlines) return fail(c, "detect_plnet_batch: no line output");
  if (nj < 0 || nj > B || (nj > 0 && (!d_junc || !d_njunc || capJ < 1))) return fail(c, "detect_plnet_batch: bad junction arguments");
  for (int b = 0; b < B; ++b) {
    const bool jn = b < nj;
    c->force_nms_map = jn;
    const int rc = detect_dev(c, d_gray + (size_t)b * img_stride, 1, h, w, stride, img_stride, d_feat + (size_t)b * cap * AIRFE_FEAT_DIM, cap, d_n + b, st);
    c->force_nms_map = false;
    if (rc || line_branch_dev(c, st, 0, 1, false)) return 1;
    if (line_tail_dev(c, 0, 1, nullptr, h, w, d_lines + (size_t)b * capL * 4, capL, d_nlines + b, lfound ? lfound + b : c->d_nlines + c->Lmax,
                      jn ? d_junc + (size_t)b * capJ * AIRFE_FEAT_DIM : nullptr, capJ, jn ? d_njunc + b : nullptr, jn ? (jfound ? jfound + b : c->d_njunc + c->Lmax) : nullptr,
                      jn ? 1 : 0, st, 3))
      return 1;
  }
  return 0;
}

int airfe_detect_plnet_batch_dev(airfe_ctx* c, const uint8_t* d_gray, int B, int h, int w, int stride, size_t img_stride, float* d_feat,
                                 int cap, int* d_n, double* d_lines, int capL, int* d_nlines, float* d_junc, int capJ, int* d_njunc,
                                 int junction_images, int* d_found, void* stream) try {
  AIRFE_ENTER(c);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (c->prec == 2)
    return plnet_batch_f32(c, d_gray, B, h, w, stride, img_stride, d_feat, cap, d_n, d_lines, capL, d_nlines, d_junc, capJ, d_njunc, junction_images, d_found,
                           d_found ? d_found + B : nullptr, st);
  c->force_nms_map = junction_images > 0;         // junction scores are read from the NMS'd maps
  const int rc = detect_dev(c, d_gray, B, h, w, stride, img_stride, d_feat, cap, d_n, st);
  c->force_nms_map = false;
  if (rc) return 1;
  return plnet_lines_batch(c, B, h, w, d_lines, capL, d_nlines, d_junc, capJ, d_njunc, junction_images, d_found, st);
} AIRFE_CATCH(c)

// (d_idx == nullptr: detection only — the stereo overload of Detect without the MatchingPoints that follows it)
static int stereo_plnet_dev(airfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int B, int h, int w, int stride,
                            size_t img_stride, float* d_featL, float* d_featR, int cap, int* d_nL, int* d_nR, double* d_lines,
                            int capL, int* d_nlines, float* d_juncL, int capJ, int* d_njuncL, int* d_found, int32_t* d_idx,
                            float* d_score, int mcap, int* d_nmatch, hipStream_t st, const std::function<int(hipStream_t)>* after_detect = nullptr,
                            const std::function<int(hipStream_t)>* after_lines = nullptr, const LgSecondPair* x2 = nullptr) {
  if (c->prec == 2) {            // fp32 mode: image by image (plnet_batch_f32), then the matcher over the B pairs (fp32 too unless matcher_precision says otherwise)
    if (x2) return fail(c, "stereo_plnet_batch: the second (temporal) pair rides in a 2-byte forward (precision fp16 / bf16)");
    if (plnet_batch_f32(c, d_left, B, h, w, stride, img_stride, d_featL, cap, d_nL, d_lines, capL, d_nlines, d_juncL, capJ, d_njuncL, d_juncL ? B : 0, d_found,
                        d_found ? d_found + 2 * B : nullptr, st) ||
        plnet_batch_f32(c, d_right, B, h, w, stride, img_stride, d_featR, cap, d_nR, d_lines + (size_t)B * capL * 4, capL, d_nlines + B, nullptr, 0, nullptr, 0,
                        d_found ? d_found + B : nullptr, nullptr, st))
      return 1;
    if (after_detect && (*after_detect)(st)) return 1;
    if (after_lines && (*after_lines)(st)) return 1;
    if (!d_idx) return 0;
    return lightglue_dev(c, d_featL, d_nL, d_featR, d_nR, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr, st, nullptr);
  }
  if (!(2 * B <= c->Dmax))
    return fail(c, "stereo_plnet_batch: needs the one-pass stereo detector (detector + LightGlue packs loaded, 2 B <= 2 max_batch)");
  // (With stage timers on anything behind the encoder the chains run one after the other: a stage's event pair must not span the other chain.)
  const uint32_t enc_only = (1u << ST_PREPROCESS) | (1u << ST_CONV1_FUSED) | (1u << ST_CONV3X3_C64);
  const bool overlap = c->overlap_lines && (c->prof_mask & ~enc_only) == 0;
  c->force_nms_map = d_juncL != nullptr;          // junction scores are read from the NMS'd maps
  int rc = detect_dev2(c, d_left, d_right, B, h, w, stride, img_stride, d_featL, d_featR, cap, d_nL, d_nR, st);
  c->force_nms_map = false;
  if (rc) return 1;
  // lines of the 2 B images (left 0 .. B-1, right B .. 2B-1), junctions of the left ones (feature_detector.cc:100-101).
  // The line path and the matcher share nothing but the detector's results: the line path runs on the context's second stream beside
  // LightGlue on the caller's (+2 % from filled launch ramps and tails; fork behind the point branch, join behind both).  Round 2 kept this
  // off: with the line path's workgroups beside it the matcher's scores were irreproducible in ~10 % of the steps — traced in round 3 to ONE
  // packed-math instruction form in the rotary epilogue (common.h, rotate_pairs), which also failed, 50x more rarely, on one stream.
  // With that form gone: 0 deviations in 3500 overlapped and 5000 single-stream steps (profiles/r03_matcher_trace_probe1.txt, _probe2.txt).
  // after_detect (the host entry's early copy of the feature rows): on the side stream where there is one, so that it runs beside the matcher
  if (after_detect && (!d_idx || !overlap) && (*after_detect)(st)) return 1;
  // after_lines (the host entry's copy of the first line / junction rows): behind the line path, on its stream
  if (!d_idx) {
    if (plnet_lines_batch(c, 2 * B, h, w, d_lines, capL, d_nlines, d_juncL, capJ, d_njuncL, d_juncL ? B : 0, d_found, st)) return 1;
    return after_lines ? (*after_lines)(st) : 0;
  }
  if (!overlap) {
    if (plnet_lines_batch(c, 2 * B, h, w, d_lines, capL, d_nlines, d_juncL, capJ, d_njuncL, d_juncL ? B : 0, d_found, st)) return 1;
    if (after_lines && (*after_lines)(st)) return 1;
    return lightglue_dev(c, d_featL, d_nL, d_featR, d_nR, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr, st, x2);
  }
  HIPCHK(c, hipEventRecord(c->ev_fork, st));                  // behind the point branch
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  if (after_detect && (*after_detect)(c->stream2)) rc = 1;
  // The matcher is the longer chain: at small batches its launches are queued FIRST (the host needs ~4 us per launch; the ~20 launches of the line
  // path queued ahead of it kept the GPU's main queue idle for 80 us per batch-1 keyframe, tools/kf_timeline.py), at large ones the line path's
  // (there the line path is as long as the matcher and a late start would stick out behind it).
  const bool lg_first = B <= 4;
  if (!rc && lg_first) rc = lightglue_dev(c, d_featL, d_nL, d_featR, d_nR, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr, st, x2);
  if (!rc) rc = plnet_lines_batch(c, 2 * B, h, w, d_lines, capL, d_nlines, d_juncL, capJ, d_njuncL, d_juncL ? B : 0, d_found, c->stream2);
  if (!rc && after_lines && (*after_lines)(c->stream2)) rc = 1;
  if (!rc && !lg_first) rc = lightglue_dev(c, d_featL, d_nL, d_featR, d_nR, B, cap, AIRFE_FEAT_DIM, 1, 1, d_idx, d_score, mcap, d_nmatch, nullptr, st, x2);
  HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));      // (also after an error: the caller's stream never runs ahead of the side stream)
  HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
  return rc;
}

int airfe_stereo_plnet_batch_dev(airfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, int B, int h, int w, int stride,
                                 size_t img_stride, float* d_featL, float* d_featR, int cap, int* d_nL, int* d_nR, double* d_lines,
                                 int capL, int* d_nlines, float* d_juncL, int capJ, int* d_njuncL, int* d_found, int32_t* d_idx,
                                 float* d_score, int mcap, int* d_nmatch, void* stream) try {
  AIRFE_ENTER(c);
  if (!d_idx || !d_score || !d_nmatch) return fail(c, "stereo_plnet_batch: no match output");
  return stereo_plnet_dev(c, d_left, d_right, B, h, w, stride, img_stride, d_featL, d_featR, cap, d_nL, d_nR, d_lines, capL, d_nlines, d_juncL, capJ,
                          d_njuncL, d_found, d_idx, d_score, mcap, d_nmatch, stream ? (hipStream_t)stream : c->stream);
} AIRFE_CATCH(c)

// ONE stereo keyframe through host buffers (batch 1, the regime of AirSLAM's feature thread): what map_builder.cc:85-86 does in two façade calls —
// Detect(left, right, features, lines, junctions) = PLNet::infer twice (feature_detector.cc:97-108), then MatchingPoints(left, right) — as ONE
// queue of device work: both images go up in one copy, the detector runs over them as a batch of two, the line path of both runs on the second
// stream beside LightGlue, and everything comes back in two copies (the counted rows, then the line / junction rows whose counts are known only then).
// Per image and per pair the results are the bits the separate entries return (tests/test_gpu_keyframe.py).  match_idx == NULL: detection only.
// (track_idx != NULL: also MatchingPoints(features_last_keyframe, left_features) — map_builder.cc:96 — as the SECOND pair of the same LightGlue forward)
static int stereo_keyframe_impl(airfe_ctx* c, const uint8_t* left, const uint8_t* right, int h, int w, int stride, float* featL, float* featR, int cap,
                                int* nL, int* nR, double* linesL, double* linesR, int capL, int* nlinesL, int* nlinesR, float* juncL, int capJ,
                                int* njuncL, int32_t* match_idx, float* match_score, int mcap, int* nmatch, const float* ref_feat, int n_ref,
                                int32_t* track_idx, float* track_score, int* ntrack) {
  AIRFE_ENTER(c);
  const bool track = track_idx != nullptr;
  if (track) {
    if (!match_idx || !track_score || !ntrack) return fail(c, "stereo_keyframe_tracked: bad argument");
    if (c->Pmax < 2) return fail(c, "stereo_keyframe_tracked: needs cfg.max_batch >= 2 (two pairs per LightGlue forward)");
    if (c->mprec == 2) return fail(c, "stereo_keyframe_tracked: the temporal pair rides in a 2-byte LightGlue forward (matcher_precision fp16 / bf16)");
    if (ref_feat && (n_ref < 0 || n_ref > c->cfg.max_keypoints)) return fail(c, "stereo_keyframe_tracked: reference keypoint count exceeds max_keypoints");
    if (!ref_feat && c->ref_n < 0) return fail(c, "stereo_keyframe_tracked: no reference features were ever given");
    *ntrack = 0;
  }
  if (!left || !right || h < 1 || w < 1) return fail(c, "empty image");
  if (stride < w) return fail(c, "image stride smaller than its width");
  if (!featL || !featR || !nL || !nR || !linesL || !linesR || !nlinesL || !nlinesR || capL < 1)
    return fail(c, "stereo_keyframe: bad argument");
  if (cap < c->cfg.max_keypoints) return fail(c, "feature capacity < max_keypoints");
  const bool match = match_idx != nullptr;
  if (match && (!match_score || !nmatch || mcap < c->cfg.max_keypoints)) return fail(c, "stereo_keyframe: match buffers smaller than max_keypoints");
  const bool want_j = juncL != nullptr;
  if (want_j && (!njuncL || capJ < 1)) return fail(c, "stereo_keyframe: bad junction arguments");
  if (!(c->prec != 2 && 2 <= c->Dmax)) return fail(c, "stereo_keyframe: needs a detector arena of two images (max_batch >= 2, or detector + LightGlue packs), fp16 / bf16");
  *nL = *nR = *nlinesL = *nlinesR = 0;
  if (njuncL) *njuncL = 0;
  if (nmatch) *nmatch = 0;
  hipStream_t st = c->stream;
  const int Np = c->cfg.max_keypoints, capLd = std::min(capL, LINE_CAP), capJd = std::min(std::max(capJ, 1), JUNC_CAP);
  // device block: [counts 64 B | featL | featR | idx | score] — the part that comes back in the first copy — then [lines 2 x capLd | junctions]
  const size_t fb = (size_t)Np * AIRFE_FEAT_DIM * 4, head = 64 + 2 * fb + (size_t)Np * 24;      // (idx | score of TWO pairs: stereo, then the temporal one)
  const size_t lb = (size_t)capLd * 32, total = head + 2 * lb + (size_t)capJd * AIRFE_FEAT_DIM * 4;
  if (ensure_block(c, c->kf_blk, c->kf_bytes, total)) return 1;
  int* cnt = reinterpret_cast<int*>(c->kf_blk);                 // {nL, nR, nlines[2], -, njunc, found: lines[2] junc[1], -, nmatch: stereo, temporal}
  float *d_fL = reinterpret_cast<float*>(c->kf_blk + 64), *d_fR = reinterpret_cast<float*>(c->kf_blk + 64 + fb);
  int32_t* d_idx = reinterpret_cast<int32_t*>(c->kf_blk + 64 + 2 * fb);
  float* d_sc = reinterpret_cast<float*>(c->kf_blk + 64 + 2 * fb + (size_t)Np * 16);               // [2][Np] behind idx [2][Np][2]
  double* d_ln = reinterpret_cast<double*>(c->kf_blk + head);
  float* d_jn = reinterpret_cast<float*>(c->kf_blk + head + 2 * lb);
  // both images through the pinned block in one copy (same row pitch; the right image starts at h * stride)
  const size_t ib = (size_t)(h - 1) * stride + w, pitch = (size_t)h * stride;
  // pinned block: [counts | featL | featR] — copied back on the side stream as soon as the detector is done, beside the matcher — then the final
  // [counts] and [idx | score]
  // then, from the side stream behind the line path (also beside the matcher), the FIRST capS line rows of either image and capJS junction rows:
  // the usual keyframe (a few hundred lines and junctions) needs no second round trip for them
  const int capS = std::min(capLd, c->kf_spec_lines), capJS = want_j ? std::min(capJd, c->kf_spec_juncs) : 0;
  const size_t early = 64 + 2 * fb, late = early + 64, spec = late + (size_t)Np * 24, spec_l = (size_t)capS * 32, spec_j = (size_t)capJS * AIRFE_FEAT_DIM * 4;
  if (ensure_stage_img(c, 2 * pitch)) return 1;
  // (sized for the line / junction rows of the second round trip too: the block must not move between calls, a captured graph holds its address)
  if (ensure_pin(c, std::max(std::max(pitch + ib, spec + 2 * spec_l + spec_j), 2 * lb + (size_t)capJd * AIRFE_FEAT_DIM * 4))) return 1;
  memcpy(c->pin, left, ib);
  memcpy(c->pin + pitch, right, ib);
  const std::function<int(hipStream_t)> early_copy = [&](hipStream_t s2) -> int {
    HIPCHK(c, hipMemcpyAsync(c->pin, c->kf_blk, early, hipMemcpyDeviceToHost, s2));
    HIPCHK(c, hipEventRecord(c->ev_feat, s2));
    return 0;
  };
  // the temporal pair (last keyframe -> left image) rides in the same LightGlue forward as pair 1; its reference features live in ref_blk (uploaded
  // when given, kept otherwise: airfe_track_frame shares the block)
  LgSecondPair x2{};
  size_t ref_off = 0;
  if (track) {
    if (ensure_block(c, c->ref_blk, c->ref_bytes, 64 + fb)) return 1;
    x2.f0 = reinterpret_cast<const float*>(c->ref_blk + 64); x2.n0 = reinterpret_cast<const int*>(c->ref_blk);
    x2.f1 = d_fL; x2.n1 = cnt;
    if (ref_feat) {
      ref_off = (std::max(pitch + ib, spec + 2 * spec_l + spec_j) + 63) / 64 * 64;
      if (ensure_pin(c, ref_off + 64 + fb)) return 1;
      memcpy(c->pin, left, ib);                                   // (the block may have moved)
      memcpy(c->pin + pitch, right, ib);
      *reinterpret_cast<int*>(c->pin + ref_off) = n_ref;
      if (n_ref > 0) memcpy(c->pin + ref_off + 64, ref_feat, (size_t)n_ref * AIRFE_FEAT_DIM * 4);
    }
  }
  DrainOnError drain{c};
  const std::function<int(hipStream_t)> rows_copy = [&](hipStream_t s2) -> int {
    HIPCHK(c, hipMemcpyAsync(c->pin + spec, d_ln, spec_l, hipMemcpyDeviceToHost, s2));
    HIPCHK(c, hipMemcpyAsync(c->pin + spec + spec_l, d_ln + (size_t)capLd * 4, spec_l, hipMemcpyDeviceToHost, s2));
    if (spec_j) HIPCHK(c, hipMemcpyAsync(c->pin + spec + 2 * spec_l, d_jn, spec_j, hipMemcpyDeviceToHost, s2));
    return 0;
  };
  auto queue_all = [&]() -> int {
    drain.armed = true;
    HIPCHK(c, hipMemsetAsync(cnt, 0, 64, st));
    HIPCHK(c, hipMemcpyAsync(c->st_img, c->pin, pitch + ib, hipMemcpyHostToDevice, st));
    if (track && ref_feat) {
      drain.ref_uploaded = true;       // (the reference count becomes valid with the queued upload; an error below takes it back)
      HIPCHK(c, hipMemcpyAsync(c->ref_blk, c->pin + ref_off, 64 + (size_t)n_ref * AIRFE_FEAT_DIM * 4, hipMemcpyHostToDevice, st));
      c->ref_n = n_ref;
    }
    // (match counts: cnt[10] = stereo, cnt[11] = temporal — LightGlue writes d_nmatch[pair])
    if (stereo_plnet_dev(c, c->st_img, c->st_img + pitch, 1, h, w, stride, pitch, d_fL, d_fR, Np, cnt, cnt + 1, d_ln, capLd, cnt + 2,
                         want_j ? d_jn : nullptr, capJd, want_j ? cnt + 5 : nullptr, cnt + 6, match ? d_idx : nullptr, d_sc, Np, cnt + 10, st, &early_copy, &rows_copy,
                         track ? &x2 : nullptr))
      return 1;
    HIPCHK(c, hipMemcpyAsync(c->pin + early, cnt, 64, hipMemcpyDeviceToHost, st));
    if (match) HIPCHK(c, hipMemcpyAsync(c->pin + late, d_idx, (size_t)Np * (track ? 24 : 16) , hipMemcpyDeviceToHost, st));
    if (match && !track) HIPCHK(c, hipMemcpyAsync(c->pin + late + (size_t)Np * 16, d_sc, (size_t)Np * 4, hipMemcpyDeviceToHost, st));
    return 0;
  };
  // airfe_tuning::kf_graph = 1: the whole queue (~115 launches on two streams) is captured once per (image shape, outputs, buffers) as a hipGraph and replayed
  // with one launch — the same kernels with the same arguments, so the same bits.  The first call of a configuration runs plainly (it grows blocks and
  // sets function attributes, which a capture cannot hold), the second captures, later ones replay.  Off while stage timers or the trace are on.
  // (Measured: <= 1 % per keyframe, profiles/r04_keyframe_graph_ab.txt — the queue is bound by the GPU's ~4.7 us per dependent launch, not by the
  // host's launch calls; the default stays the plain queue.)
  KfGraph& G = c->kf_graph;
  const KfGraph::Key key{h, w, stride, capLd, capJd, want_j, match, c->pin, c->kf_blk, c->st_img};
  const bool graph_ok = c->kf_graph_on && c->prof_mask == 0 && !c->trace_on && !track;
  if (!(G.key == key)) { G.reset(); G.key = key; }
  bool replay = false;
  if (graph_ok && G.exec) {
    HIPCHK(c, hipGraphLaunch(G.exec, st));
    G.state.restore(c);
    replay = true;
  } else if (graph_ok && G.seen >= 1) {
    hipGraph_t graph = nullptr;
    HIPCHK(c, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    const int qrc = queue_all();
    const hipError_t ce = hipStreamEndCapture(st, &graph);
    if (qrc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
    if (ce != hipSuccess || !graph) return fail(c, std::string("stereo_keyframe: stream capture failed: ") + hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { G.exec = nullptr; return fail(c, std::string("stereo_keyframe: hipGraphInstantiate: ") + hipGetErrorString(ie)); }
    G.state.save(c);
    HIPCHK(c, hipGraphLaunch(G.exec, st));
    replay = true;
  } else {
    if (queue_all()) return 1;
    ++G.seen;
  }
  // the feature rows leave the pinned block while the matcher is still running (a replayed graph has no event to wait on: everything at the end)
  if (replay) HIPCHK(c, hipStreamSynchronize(st));
  else HIPCHK(c, hipEventSynchronize(c->ev_feat));
  const int* hc0 = reinterpret_cast<const int*>(c->pin);
  const int n0 = std::min(hc0[0], Np), n1 = std::min(hc0[1], Np);
  if (n0 > 0) memcpy(featL, c->pin + 64, (size_t)n0 * AIRFE_FEAT_DIM * 4);
  if (n1 > 0) memcpy(featR, c->pin + 64 + fb, (size_t)n1 * AIRFE_FEAT_DIM * 4);
  *nL = n0; *nR = n1;
  if (!replay) HIPCHK(c, hipStreamSynchronize(st));
  drain.ref_uploaded = false;           // (the stream is idle: an uploaded reference is whole whatever is reported below — as airfe_track_frame does)
  if (saturation_status(c)) { *nL = *nR = 0; return 1; }
  const int* hc = reinterpret_cast<const int*>(c->pin + early);
  const int nl0 = hc[2], nl1 = hc[3], nm = std::min(hc[10], Np), nj = hc[5];
  const int fl0 = hc[6], fl1 = hc[7], fj = hc[8];
  if (match) {
    if (nm > 0) {
      memcpy(match_idx, c->pin + late, (size_t)nm * 8);
      memcpy(match_score, c->pin + late + (size_t)Np * 16, (size_t)nm * 4);
    }
    *nmatch = (n0 > 0 && n1 > 0) ? nm : 0;                        // point_matcher.cc:53-55
  }
  if (track && n0 > 0 && c->ref_n > 0) {
    const int nt = std::min(hc[11], Np);
    if (nt > 0) {
      memcpy(track_idx, c->pin + late + (size_t)Np * 8, (size_t)nt * 8);
      memcpy(track_score, c->pin + late + (size_t)Np * 20, (size_t)nt * 4);
    }
    *ntrack = nt;
  }
  drain.ref_uploaded = false;           // (everything queued has completed: the reference rows are whole whatever is reported below)
  if (want_j && fj > JUNC_CAP) return fail(c, "stereo_keyframe: more junctions than the device arena holds (JUNC_CAP)");
  if (fl0 > capLd || fl1 > capLd || (want_j && fj > capJd)) return fail(c, "stereo_keyframe: lines / junctions do not fit the caller's buffers (capL, capJ)");
  const size_t b0 = (size_t)nl0 * 32, b1 = (size_t)nl1 * 32, bj = want_j ? (size_t)nj * AIRFE_FEAT_DIM * 4 : 0;
  if (nl0 <= capS && nl1 <= capS && (!want_j || nj <= capJS)) {      // the rows are already here
    if (b0) memcpy(linesL, c->pin + spec, b0);
    if (b1) memcpy(linesR, c->pin + spec + spec_l, b1);
    if (bj) memcpy(juncL, c->pin + spec + 2 * spec_l, bj);
  } else if (b0 + b1 + bj) {
    if (ensure_pin(c, b0 + b1 + bj)) return 1;
    if (b0) HIPCHK(c, hipMemcpyAsync(c->pin, d_ln, b0, hipMemcpyDeviceToHost, st));
    if (b1) HIPCHK(c, hipMemcpyAsync(c->pin + b0, d_ln + (size_t)capLd * 4, b1, hipMemcpyDeviceToHost, st));
    if (bj) HIPCHK(c, hipMemcpyAsync(c->pin + b0 + b1, d_jn, bj, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (b0) memcpy(linesL, c->pin, b0);
    if (b1) memcpy(linesR, c->pin + b0, b1);
    if (bj) memcpy(juncL, c->pin + b0 + b1, bj);
  }
  *nlinesL = nl0; *nlinesR = nl1;
  if (njuncL) *njuncL = want_j ? nj : 0;
  drain.armed = false;
  return 0;
}

int airfe_stereo_keyframe(airfe_ctx* c, const uint8_t* left, const uint8_t* right, int h, int w, int stride, float* featL, float* featR, int cap,
                          int* nL, int* nR, double* linesL, double* linesR, int capL, int* nlinesL, int* nlinesR, float* juncL, int capJ,
                          int* njuncL, int32_t* match_idx, float* match_score, int mcap, int* nmatch) try {
  return stereo_keyframe_impl(c, left, right, h, w, stride, featL, featR, cap, nL, nR, linesL, linesR, capL, nlinesL, nlinesR, juncL, capJ, njuncL, match_idx,
                              match_score, mcap, nmatch, nullptr, 0, nullptr, nullptr, nullptr);
} AIRFE_CATCH(c)
int airfe_stereo_keyframe_tracked(airfe_ctx* c, const uint8_t* left, const uint8_t* right, int h, int w, int stride, float* featL, float* featR, int cap,
                                  int* nL, int* nR, double* linesL, double* linesR, int capL, int* nlinesL, int* nlinesR, float* juncL, int capJ,
                                  int* njuncL, int32_t* match_idx, float* match_score, int mcap, int* nmatch, const float* ref_feat, int n_ref,
                                  int32_t* track_idx, float* track_score, int* ntrack) try {
  if (c && !track_idx) return fail(c, "stereo_keyframe_tracked: no output for the temporal matches");
  return stereo_keyframe_impl(c, left, right, h, w, stride, featL, featR, cap, nL, nR, linesL, linesR, capL, nlinesL, nlinesR, juncL, capJ, njuncL, match_idx,
                              match_score, mcap, nmatch, ref_feat, n_ref, track_idx, track_score, ntrack);
} AIRFE_CATCH(c)

// ONE tracked frame through host buffers (batch 1): what map_builder.cc:94-101 does for every frame that is not a keyframe —
//     _feature_detector->Detect(image_left_rect, left_features);
//     _point_matcher->MatchingPoints(features_last_keyframe, left_features, matches, true);      (the F-RANSAC behind it stays the reference's)
// — as one queue: the last keyframe's features live on the device (uploaded when ref_feat != NULL, i.e. once per keyframe, not once per frame), the
// new frame's feature rows come back on the side stream while LightGlue runs.  Bits: those of airfe_detect_points + airfe_match_lightglue.
int airfe_track_frame(airfe_ctx* c, const uint8_t* gray, int h, int w, int stride, const float* ref_feat, int n_ref, float* feat, int cap, int* n,
                      int32_t* match_idx, float* match_score, int mcap, int* nmatch) try {
  AIRFE_ENTER(c);
  if (!gray || h < 1 || w < 1) return fail(c, "empty image");
  if (stride < w) return fail(c, "image stride smaller than its width");
  if (!feat || !n || !match_idx || !match_score || !nmatch) return fail(c, "track_frame: bad argument");
  if (cap < c->cfg.max_keypoints || mcap < c->cfg.max_keypoints) return fail(c, "feature / match capacity < max_keypoints");
  if (!c->has_lg) return fail(c, "track_frame: LightGlue weights were not loaded (cfg.lightglue_pack)");
  if (c->mprec == 2 || c->prec == 2) return fail(c, "track_frame runs in fp16 / bf16");
  const int Np = c->cfg.max_keypoints;
  if (ref_feat && (n_ref < 0 || n_ref > Np)) return fail(c, "track_frame: reference keypoint count exceeds max_keypoints");
  *n = 0; *nmatch = 0;
  c->tk_n = -1;
  hipStream_t st = c->stream;
  const size_t fb = (size_t)Np * AIRFE_FEAT_DIM * 4, early = 64 + fb, late = early + 64;
  // device block: [counts | new rows | idx | score] and the reference block [count | reference rows] (kept from call to call)
  if (ensure_block(c, c->tk_blk, c->tk_bytes, 64 + fb + (size_t)Np * 12)) return 1;
  if (ensure_block(c, c->ref_blk, c->ref_bytes, 64 + fb)) return 1;
  int* cnt = reinterpret_cast<int*>(c->tk_blk);                       // {n_new, -, nmatch}
  float* d_new = reinterpret_cast<float*>(c->tk_blk + 64);
  int32_t* d_idx = reinterpret_cast<int32_t*>(c->tk_blk + 64 + fb);
  float* d_sc = reinterpret_cast<float*>(c->tk_blk + 64 + fb + (size_t)Np * 8);
  int* d_nref = reinterpret_cast<int*>(c->ref_blk);
  float* d_ref = reinterpret_cast<float*>(c->ref_blk + 64);
  const size_t ib = (size_t)(h - 1) * stride + w, ioff = (ib + 63) / 64 * 64;
  if (ensure_stage_img(c, (size_t)h * stride)) return 1;
  if (ensure_pin(c, std::max(ioff + 64 + fb, late + (size_t)Np * 12))) return 1;
  if (!ref_feat && c->ref_n < 0) return fail(c, "track_frame: no reference features were ever given (ref_feat == NULL on the first call)");
  memcpy(c->pin, gray, ib);
  DrainOnError drain{c, true};
  HIPCHK(c, hipMemsetAsync(cnt, 0, 64, st));
  HIPCHK(c, hipMemcpyAsync(c->st_img, c->pin, ib, hipMemcpyHostToDevice, st));
  if (ref_feat) {
    *reinterpret_cast<int*>(c->pin + ioff) = n_ref;
    if (n_ref > 0) memcpy(c->pin + ioff + 64, ref_feat, (size_t)n_ref * AIRFE_FEAT_DIM * 4);
    drain.ref_uploaded = true;
    HIPCHK(c, hipMemcpyAsync(c->ref_blk, c->pin + ioff, 64 + (size_t)n_ref * AIRFE_FEAT_DIM * 4, hipMemcpyHostToDevice, st));
    c->ref_n = n_ref;
  }
  if (detect_dev(c, c->st_img, 1, h, w, stride, (size_t)h * stride, d_new, Np, cnt, st)) return 1;
  HIPCHK(c, hipEventRecord(c->ev_fork, st));                         // the new rows go home on the side stream, beside the matcher
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  HIPCHK(c, hipMemcpyAsync(c->pin, c->tk_blk, early, hipMemcpyDeviceToHost, c->stream2));
  HIPCHK(c, hipEventRecord(c->ev_feat, c->stream2));
  if (lightglue_dev(c, d_ref, d_nref, d_new, cnt, 1, Np, AIRFE_FEAT_DIM, 1, 1, d_idx, d_sc, Np, cnt + 2, nullptr, st)) return 1;
  HIPCHK(c, hipMemcpyAsync(c->pin + early, cnt, 64, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(c->pin + late, d_idx, (size_t)Np * 12, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipEventSynchronize(c->ev_feat));
  const int nn = std::min(*reinterpret_cast<const int*>(c->pin), Np);
  if (nn > 0) memcpy(feat, c->pin + 64, (size_t)nn * AIRFE_FEAT_DIM * 4);
  *n = nn;
  HIPCHK(c, hipStreamSynchronize(st));
  drain.armed = false;
  if (saturation_status(c)) { *n = 0; return 1; }
  c->tk_n = nn;                                                      // (airfe_promote_frame / airfe_adopt_reference work on these rows)
  if (nn < 1 || c->ref_n < 1) return 0;                              // point_matcher.cc:53-55
  const int nm = std::min(reinterpret_cast<const int*>(c->pin + early)[2], Np);
  if (nm > 0) {
    memcpy(match_idx, c->pin + late, (size_t)nm * 8);
    memcpy(match_score, c->pin + late + (size_t)Np * 8, (size_t)nm * 4);
  }
  *nmatch = nm;
  return 0;
} AIRFE_CATCH(c)

// The promotion of map_builder.cc:104-108 (see include/airfe.h): Detect(right) + MatchingPoints(left, right) with the left rows of the last
// airfe_track_frame still on the device; the right rows come home on the side stream while LightGlue runs (the queue of airfe_track_frame).
int airfe_promote_frame(airfe_ctx* c, const uint8_t* right, int h, int w, int stride, float* featR, int cap, int* nR, int32_t* match_idx,
                        float* match_score, int mcap, int* nmatch) try {
  AIRFE_ENTER(c);
  if (!right || h < 1 || w < 1) return fail(c, "empty image");
  if (stride < w) return fail(c, "image stride smaller than its width");
  if (!featR || !nR || !match_idx || !match_score || !nmatch) return fail(c, "promote_frame: bad argument");
  if (cap < c->cfg.max_keypoints || mcap < c->cfg.max_keypoints) return fail(c, "feature / match capacity < max_keypoints");
  if (!c->has_lg) return fail(c, "promote_frame: LightGlue weights were not loaded (cfg.lightglue_pack)");
  if (c->mprec == 2 || c->prec == 2) return fail(c, "promote_frame runs in fp16 / bf16");
  if (c->tk_n < 0 || !c->tk_blk) return fail(c, "promote_frame: no airfe_track_frame preceded it (the left features live on the device since that call)");
  const int Np = c->cfg.max_keypoints;
  *nR = 0; *nmatch = 0;
  hipStream_t st = c->stream;
  const size_t fb = (size_t)Np * AIRFE_FEAT_DIM * 4, early = 64 + fb, late = early + 64;
  if (ensure_block(c, c->pr_blk, c->pr_bytes, 64 + fb + (size_t)Np * 12)) return 1;
  int* cnt = reinterpret_cast<int*>(c->pr_blk);                       // {n_right, -, nmatch}
  float* d_right = reinterpret_cast<float*>(c->pr_blk + 64);
  int32_t* d_idx = reinterpret_cast<int32_t*>(c->pr_blk + 64 + fb);
  float* d_sc = reinterpret_cast<float*>(c->pr_blk + 64 + fb + (size_t)Np * 8);
  const int* d_nleft = reinterpret_cast<const int*>(c->tk_blk);
  const float* d_left = reinterpret_cast<const float*>(c->tk_blk + 64);
  const size_t ib = (size_t)(h - 1) * stride + w;
  if (ensure_stage_img(c, (size_t)h * stride)) return 1;
  if (ensure_pin(c, std::max(ib, late + (size_t)Np * 12))) return 1;
  memcpy(c->pin, right, ib);
  DrainOnError drain{c, true};
  HIPCHK(c, hipMemsetAsync(cnt, 0, 64, st));
  HIPCHK(c, hipMemcpyAsync(c->st_img, c->pin, ib, hipMemcpyHostToDevice, st));
  if (detect_dev(c, c->st_img, 1, h, w, stride, (size_t)h * stride, d_right, Np, cnt, st)) return 1;
  HIPCHK(c, hipEventRecord(c->ev_fork, st));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  HIPCHK(c, hipMemcpyAsync(c->pin, c->pr_blk, early, hipMemcpyDeviceToHost, c->stream2));
  HIPCHK(c, hipEventRecord(c->ev_feat, c->stream2));
  if (lightglue_dev(c, d_left, d_nleft, d_right, cnt, 1, Np, AIRFE_FEAT_DIM, 1, 1, d_idx, d_sc, Np, cnt + 2, nullptr, st)) return 1;
  HIPCHK(c, hipMemcpyAsync(c->pin + early, cnt, 64, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(c->pin + late, d_idx, (size_t)Np * 12, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipEventSynchronize(c->ev_feat));
  const int nn = std::min(*reinterpret_cast<const int*>(c->pin), Np);
  if (nn > 0) memcpy(featR, c->pin + 64, (size_t)nn * AIRFE_FEAT_DIM * 4);
  *nR = nn;
  HIPCHK(c, hipStreamSynchronize(st));
  drain.armed = false;
  if (saturation_status(c)) { *nR = 0; return 1; }
  if (nn < 1 || c->tk_n < 1) return 0;                               // point_matcher.cc:53-55
  const int nm = std::min(reinterpret_cast<const int*>(c->pin + early)[2], Np);
  if (nm > 0) {
    memcpy(match_idx, c->pin + late, (size_t)nm * 8);
    memcpy(match_score, c->pin + late + (size_t)Np * 8, (size_t)nm * 4);
  }
  *nmatch = nm;
  return 0;
} AIRFE_CATCH(c)

int airfe_adopt_reference(airfe_ctx* c) try {
  AIRFE_ENTER(c);
  if (c->tk_n < 0 || !c->tk_blk) return fail(c, "adopt_reference: no airfe_track_frame preceded it");
  const size_t fb = (size_t)c->cfg.max_keypoints * AIRFE_FEAT_DIM * 4;
  if (ensure_block(c, c->ref_blk, c->ref_bytes, 64 + fb)) return 1;
  // [count | rows] have the same layout in both blocks (the count's word 0; words 1.. of the header are scratch of the entries)
  HIPCHK(c, hipMemcpyAsync(c->ref_blk, c->tk_blk, 64 + (size_t)std::max(c->tk_n, 0) * AIRFE_FEAT_DIM * 4, hipMemcpyDeviceToDevice, c->stream));
  c->ref_n = c->tk_n;
  return 0;
} AIRFE_CATCH(c)

/* the on-device stage-0 line branch of the LAST detected image, copied out in the Appendix A.1 layouts (NULL = skip) */
int airfe_debug_plnet_stage0(airfe_ctx* c, float* juncs_pred, float* lines_pred, float* iskeep, float* idx_min, float* idx_max,
                             float* loi, float* thin, float* aux, float* jloc, float* joff) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_s0 || !c->has_s1) return fail(c, "debug_plnet_stage0: line branch / stage 1 not loaded");
  hipStream_t st = c->stream;
  if (line_branch_dev(c, st, 0, 1, true)) return 1;
  HIPCHK(c, hipStreamSynchronize(st));
  const size_t NP = KEEP_CAP;
  float* d = c->s0_stage;
  struct { float* h; const float* dv; size_t n; } cp[10] = {
      {juncs_pred, d + SG_JUNCS, 600}, {lines_pred, d + SG_LP, NP * 4}, {iskeep, d + SG_KEEP, NP}, {idx_min, d + SG_MIN, NP},
      {idx_max, d + SG_MAX, NP}, {loi, c->s0_loi, (size_t)128 * 128 * 128}, {thin, d + SG_THIN, (size_t)4 * 128 * 128},
      {aux, d + SG_AUX, (size_t)4 * 128 * 128}, {jloc, c->l_jloc, (size_t)128 * 128}, {joff, c->l_joff, (size_t)2 * 128 * 128}};
  for (auto& e : cp)
    if (e.h) HIPCHK(c, hipMemcpy(e.h, e.dv, e.n * 4, hipMemcpyDeviceToHost));
  return 0;
} AIRFE_CATCH(c)

/* the junction-to-line match of the LAST detected image as the line path runs it (fast = 1: cell search, exact where it is consumed) or
   as the inspection hook exports it (fast = 0: every proposal against every junction): iskeep, idx_junc_to_end_min / _max [3*128*128] */
int airfe_debug_plnet_j2l(airfe_ctx* c, int fast, float* iskeep, float* idx_min, float* idx_max) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_s0 || !c->has_s1) return fail(c, "debug_plnet_j2l: line branch / stage 1 not loaded");
  hipStream_t st = c->stream;
  if (line_branch_dev(c, st, 0, 1, fast == 0)) return 1;
  HIPCHK(c, hipStreamSynchronize(st));
  const size_t NP = KEEP_CAP;
  float* d = c->s0_stage;
  if (iskeep) HIPCHK(c, hipMemcpy(iskeep, d + SG_KEEP, NP * 4, hipMemcpyDeviceToHost));
  if (idx_min) HIPCHK(c, hipMemcpy(idx_min, d + SG_MIN, NP * 4, hipMemcpyDeviceToHost));
  if (idx_max) HIPCHK(c, hipMemcpy(idx_max, d + SG_MAX, NP * 4, hipMemcpyDeviceToHost));
  return 0;
} AIRFE_CATCH(c)

/* stage-1 alone on HOST stage-0 tensors: lines_adjusted [M2][4] + scores_line [M2] (parity vs the real plnet_s1.onnx) */
int airfe_debug_plnet_s1(airfe_ctx* c, const airfe_plnet_stage0* s0, float* lines_adjusted, float* scores_line, int cap, int* m2) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_s1 || !s0) return fail(c, "debug_plnet_s1: stage-1 not loaded");
  hipStream_t st = c->stream;
  if (upload_stage0(c, s0, st)) return 1;
  float* d = c->s0_stage;
  launch_wireframe(d + SG_KEEP, d + SG_MIN, d + SG_MAX, KEEP_CAP, 300, c->wf_table, c->wf_keep, c->wf_pairs, c->wf_rep, KEEP_CAP, LINE_CAP,
                   c->wf_counts, false, d + SG_JUNCS, d + SG_LP, c->s1_la, c->wf_prop, 1, SG_STRIDE, st);
  launch_plnet_s1(d + SG_JUNCS, d + SG_LP, c->wf_keep, c->wf_pairs, c->wf_rep, c->wf_counts, c->s0_loi, 0, nullptr, nullptr, d + SG_THIN, d + SG_AUX,
                  c->s1_w, c->s1_la, c->s1_sc, KEEP_CAP, LINE_CAP, 1, SG_STRIDE, st);
  int cnt[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(cnt, c->wf_counts, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const int k = std::min(cnt[1], cap);
  if (k > 0) {
    HIPCHK(c, hipMemcpy(lines_adjusted, c->s1_la, (size_t)k * 16, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(scores_line, c->s1_sc, (size_t)k * 4, hipMemcpyDeviceToHost));
  }
  *m2 = k;
  return 0;
} AIRFE_CATCH(c)

int airfe_debug_plnet_s1_last(airfe_ctx* c, float* lines_adjusted, float* scores_line, int cap, int* m2) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_s1 || !lines_adjusted || !scores_line || !m2) return fail(c, "debug_plnet_s1_last: stage-1 not loaded / bad argument");
  HIPCHK(c, hipDeviceSynchronize());
  int cnt[2] = {0, 0};
  HIPCHK(c, hipMemcpy(cnt, c->wf_counts, 8, hipMemcpyDeviceToHost));
  const int k = std::min(cnt[1], cap);
  if (k > 0) {
    HIPCHK(c, hipMemcpy(lines_adjusted, c->s1_la, (size_t)k * 16, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(scores_line, c->s1_sc, (size_t)k * 4, hipMemcpyDeviceToHost));
  }
  *m2 = k;
  return 0;
} AIRFE_CATCH(c)

// The register-resident Sinkhorn kernel raises a device word when one of its bounded rendezvous spins timed out (that pair's Z is NaN): read
// after a synchronisation, cleared once reported.
static int sinkhorn_failed(airfe_ctx* c) {
  if (!c->has_sg || !c->sg_cnt) return 0;
  unsigned f = 0;
  unsigned* flag = c->sg_cnt + (size_t)c->Pmax * 16;
  if (hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost) != hipSuccess || f == 0) return 0;
  (void)hipMemset(flag, 0, 4);
  return fail(c, "SuperGlue: a Sinkhorn rendezvous timed out (workgroups of the cooperative launch not co-resident?): the scores of that call are NaN");
}

static int sg_host(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, int32_t* idx0, int32_t* idx1, double* ms0,
                   double* ms1, float* scores_full) {
  AIRFE_ENTER(c);
  if (n0 < 1 || n1 < 1) return fail(c, "airfe_match_superglue: empty input (MatchingPoints early-outs before calling infer)");
  if (n0 > c->cfg.max_keypoints || n1 > c->cfg.max_keypoints) return fail(c, "keypoint count exceeds max_keypoints");
  HIPCHK(c, hipMemcpyAsync(c->st_feat0, f0, (size_t)n0 * 259 * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_feat1, f1, (size_t)n1 * 259 * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_n0, &n0, 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_n1, &n1, 4, hipMemcpyHostToDevice, c->stream));
  if (superglue_dev(c, c->st_feat0, c->st_n0, c->st_feat1, c->st_n1, 1, c->Np, 0, c->stream)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (sinkhorn_failed(c)) return 1;
  if (idx0) {
    std::vector<float> m0(n0), m1(n1);
    HIPCHK(c, hipMemcpy(idx0, c->sg_out0, (size_t)n0 * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(idx1, c->sg_out1, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(m0.data(), c->sg_ms0, (size_t)n0 * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(m1.data(), c->sg_ms1, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n0; ++i) ms0[i] = (double)m0[i];      // VectorXd filled from float vectors: super_glue.cpp:464-469
    for (int j = 0; j < n1; ++j) ms1[j] = (double)m1[j];
  }
  if (scores_full)
    HIPCHK(c, hipMemcpy2D(scores_full, (size_t)(n1 + 1) * 4, c->sg_Z, (size_t)c->Lz * 4, (size_t)(n1 + 1) * 4, n0 + 1, hipMemcpyDeviceToHost));
  return 0;
}

int airfe_match_superglue(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, int32_t* idx0, int32_t* idx1,
                          double* ms0, double* ms1) try {
  if (c && enter_device(c)) return 1;
  return sg_host(c, f0, n0, f1, n1, idx0, idx1, ms0, ms1, nullptr);
} AIRFE_CATCH(c)

/* decode (src/super_glue.cpp:339-367) alone on one HOST score matrix Z [n0+1][n1+1] */
int airfe_debug_sg_decode(airfe_ctx* c, const float* Z, int n0, int n1, int32_t* idx0, int32_t* idx1, double* ms0, double* ms1) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_sg) return fail(c, "debug_sg_decode: SuperGlue not loaded");
  if (n0 < 1 || n1 < 1 || n0 > c->cfg.max_keypoints || n1 > c->cfg.max_keypoints || !Z || !idx0 || !idx1 || !ms0 || !ms1)
    return fail(c, "debug_sg_decode: bad argument");
  const int lens[2] = {n0, n1};
  HIPCHK(c, hipMemcpyAsync(c->lens, lens, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->sg_Z, (size_t)c->Lz * 4, Z, (size_t)(n1 + 1) * 4, (size_t)(n1 + 1) * 4, n0 + 1, hipMemcpyHostToDevice, c->stream));
  launch_sg_decode(c->sg_Z, c->lens, 1, c->Np, c->Lz, 0.2f, c->sg_idx0, c->sg_max0, c->sg_idx1, c->sg_out0, c->sg_out1, c->sg_ms0,
                   c->sg_ms1, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<float> m0(n0), m1(n1);
  HIPCHK(c, hipMemcpy(idx0, c->sg_out0, (size_t)n0 * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(idx1, c->sg_out1, (size_t)n1 * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(m0.data(), c->sg_ms0, (size_t)n0 * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(m1.data(), c->sg_ms1, (size_t)n1 * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n0; ++i) ms0[i] = (double)m0[i];
  for (int j = 0; j < n1; ++j) ms1[j] = (double)m1[j];
  return 0;
} AIRFE_CATCH(c)

/* SuperGlue on B pairs of DEVICE feature matrices (259-float rows, original pixel coordinates; NormalizeKeypoints with scale 0.7 on
   the device): d_idx0 / d_idx1 [B][cap] (-1 = unmatched), d_ms0 / d_ms1 [B][cap] floats.  No reference counterpart (batch-1 there). */
int airfe_match_superglue_batch_dev(airfe_ctx* c, const float* d_f0, const int* d_n0, const float* d_f1, const int* d_n1, int B, int cap,
                                    int32_t* d_idx0, int32_t* d_idx1, float* d_ms0, float* d_ms1, void* stream) try {
  AIRFE_ENTER(c);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (superglue_dev(c, d_f0, d_n0, d_f1, d_n1, B, cap, 1, st)) return 1;
  const size_t sp = (size_t)c->Lz * 4, dp = (size_t)cap * 4, wb = (size_t)std::min(cap, c->Lz) * 4;
  HIPCHK(c, hipMemcpy2DAsync(d_idx0, dp, c->sg_out0, sp, wb, B, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipMemcpy2DAsync(d_idx1, dp, c->sg_out1, sp, wb, B, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipMemcpy2DAsync(d_ms0, dp, c->sg_ms0, sp, wb, B, hipMemcpyDeviceToDevice, st));
  HIPCHK(c, hipMemcpy2DAsync(d_ms1, dp, c->sg_ms1, sp, wb, B, hipMemcpyDeviceToDevice, st));
  return 0;
} AIRFE_CATCH(c)

/* full SuperGlue output `scores` [n0+1][n1+1] (binding A.5) for one HOST pair */
int airfe_debug_superglue_scores(airfe_ctx* c, const float* f0, int n0, const float* f1, int n1, float* scores) try {
  if (c && enter_device(c)) return 1;
  return sg_host(c, f0, n0, f1, n1, nullptr, nullptr, nullptr, nullptr, scores);
} AIRFE_CATCH(c)

// ---- kernel-level test hooks ------------------------------------------------------------------------------
int airfe_debug_preprocess(airfe_ctx* c, const uint8_t* gray, int h, int w, int stride, float* out) try {
  if (c && enter_device(c)) return 1;
  if (!c || !c->has_sp) return fail(c, "debug_preprocess: detector not loaded");
  const size_t bytes = (size_t)h * stride;
  if (upload_image(c, gray, h, w, stride) || ensure_tables(c, h, w)) return 1;
  const int R = AIRFE_INTERNAL_SIZE;
  launch_preprocess(c->st_img, 1, h, w, stride, bytes, c->xtab, c->ytab, c->lut, c->img32, R, R, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy2D(out, (size_t)R * 4, c->img32 + (R + 2) + 1, (size_t)(R + 2) * 4, (size_t)R * 4, R, hipMemcpyDeviceToHost));
  return 0;
} AIRFE_CATCH(c)

int airfe_debug_conv3x3(airfe_ctx* c, const float* x, int B, int cin, int H, int W, const float* w, const float* b, int cout,
                        int pool, float* y) try {
  AIRFE_ENTER(c);
  if ((cin != 64 && cin != 128) || cout % 64 || W % 16 || H % 16) return fail(c, "debug_conv3x3: unsupported shape");
  const int prec = c->prec;
  std::vector<uint16_t> xin((size_t)B * (H + 2) * (W + 2) * cin, 0);
  for (int bb = 0; bb < B; ++bb)
    for (int ci = 0; ci < cin; ++ci)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx)
          xin[(((size_t)bb * (H + 2) + yy + 1) * (W + 2) + xx + 1) * cin + ci] = cvt2(x[(((size_t)bb * cin + ci) * H + yy) * W + xx], prec);
  const int nci = cin / 64;
  auto slabs = pack_slabs(cout / 64, 9 * nci, prec, [&](int feat, int s, int k) {
    const int tap = s / nci, cc = s % nci, ci = cc * 64 + k;
    return w[((size_t)feat * cin + ci) * 9 + tap];
  });
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  uint16_t *dx = nullptr, *dw = nullptr, *dy = nullptr;
  float* db = nullptr;
  const size_t ybytes = (size_t)B * (Ho + 2) * (Wo + 2) * cout * 2;
  HIPCHK(c, hipMalloc((void**)&dx, xin.size() * 2));
  HIPCHK(c, hipMalloc((void**)&dw, slabs.size() * 2));
  HIPCHK(c, hipMalloc((void**)&dy, ybytes));
  HIPCHK(c, hipMalloc((void**)&db, (size_t)cout * 4));
  HIPCHK(c, hipMemcpy(dx, xin.data(), xin.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dw, slabs.data(), slabs.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(db, b, (size_t)cout * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(dy, 0, ybytes));
  ConvArgs a;
  a.X = dx; a.Wp = dw; a.bias = db; a.Y = dy; a.B = B; a.H = H; a.W = W; a.CIN = cin; a.COUT = cout;
  a.pool = pool; a.out_pad = 1; a.relu = 1;
  launch_conv3x3(prec, a, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<uint16_t> yo(ybytes / 2);
  HIPCHK(c, hipMemcpy(yo.data(), dy, ybytes, hipMemcpyDeviceToHost));
  for (int bb = 0; bb < B; ++bb)
    for (int co = 0; co < cout; ++co)
      for (int yy = 0; yy < Ho; ++yy)
        for (int xx = 0; xx < Wo; ++xx)
          y[(((size_t)bb * cout + co) * Ho + yy) * Wo + xx] = back2(yo[(((size_t)bb * (Ho + 2) + yy + 1) * (Wo + 2) + xx + 1) * cout + co], prec);
  (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(db);
  return 0;
} AIRFE_CATCH(c)

int airfe_debug_fail_next_launch(airfe_ctx* c, int stage) try {
  AIRFE_ENTER(c);
  if (stage < -1 || stage >= ST_COUNT) return fail(c, "debug_fail_next_launch: no such stage");
  c->fail_stage = stage;
  return 0;
} AIRFE_CATCH(c)

int airfe_debug_gemm(airfe_ctx* c, const float* x, int M, int K, const float* w, const float* b, int N, int relu, float* y) try {
  AIRFE_ENTER(c);
  if (K != 128 && K != 256 && K != 512) return fail(c, "debug_gemm: K must be 128, 256 or 512");
  const int prec = c->prec, Mp = (M + 127) / 128 * 128, Np8 = (N + 7) / 8 * 8;
  std::vector<uint16_t> xin((size_t)Mp * K, 0);
  for (size_t i = 0; i < (size_t)M * K; ++i) xin[i] = cvt2(x[i], prec);
  airfe_ctx tmp;   // only as an allocation list holder
  tmp.prec = prec;
  tmp.pack_prec = prec;
  LinW lw;
  if (!make_linear(&tmp, w, b, K, N, lw)) return fail(c, "debug_gemm: allocation failed");
  uint16_t* dx = dupload(&tmp, xin);
  float* dy = dalloc<float>(&tmp, (size_t)Mp * Np8);
  int rc = 0;
  if (!dx || !dy) rc = fail(c, "debug_gemm: allocation failed");
  if (!rc) {
    GemmArgs g;
    g.X1 = dx; g.ld1 = K; g.K1 = K; g.Wp = lw.w; g.bias = lw.b; g.M = Mp; g.N = N; g.cb_total = lw.cbt;
    g.epi = EPI_STORE_F32; g.act = relu ? ACT_RELU : ACT_NONE; g.out = dy; g.ldo = Np8;
    g.small_max = c->gemm_small_max; g.g8_min = c->gemm8_min; g.gr_min = c->gemmr_min; g.gr_wgs = c->gemmr_wgs;
    launch_gemm(prec, K, false, g, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "debug_gemm: kernel failed");
  }
  if (!rc) {
    std::vector<float> yo((size_t)Mp * Np8);
    (void)hipMemcpy(yo.data(), dy, yo.size() * 4, hipMemcpyDeviceToHost);
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) y[(size_t)m * N + n] = yo[(size_t)m * Np8 + n];
  }
  for (void* p : tmp.allocs) (void)hipFree(p);
  return rc;
} AIRFE_CATCH(c)

int airfe_debug_attention(airfe_ctx* c, const float* q, const float* k, const float* v, const int* lens, int S, int H, int n, int cross, float* out) try {
  AIRFE_ENTER(c);
  if (c->mprec == 2) return fail(c, "debug_attention drives the 2-byte kernel (matcher_precision fp16 / bf16)");
  if (S < 1 || H < 1 || n < 1 || (S * H) % 8 != 0 || (cross && (S & 1))) return fail(c, "debug_attention: S * H must be a multiple of 8 (cross: S even)");
  if (!q || !k || !v || !lens || !out) return fail(c, "debug_attention: null argument");
  for (int s = 0; s < S; ++s)
    if (lens[s] < 0 || lens[s] > n) return fail(c, "debug_attention: lens[s] must lie in 0 .. n");
  const int Np = (n + 15) / 16 * 16, prec = c->mprec;
  const size_t rows = (size_t)S * H * Np + 128;                      // (+ slack: the last key tile reads up to 63 rows past a sequence)
  std::vector<uint16_t> hq(rows * 64, 0), hk(rows * 64, 0), hvt(rows * 64, 0);
  for (int s = 0; s < S; ++s)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < n; ++i)
        for (int d = 0; d < 64; ++d) {
          const size_t src = (((size_t)s * H + h) * n + i) * 64 + d;
          hq[(((size_t)s * H + h) * Np + i) * 64 + d] = cvt2(q[src], prec);
          hk[(((size_t)s * H + h) * Np + i) * 64 + d] = cvt2(k[src], prec);
          hvt[(((size_t)s * H + h) * 64 + d) * Np + i] = cvt2(v[src], prec);       // V^T [S][H][64][Np]
        }
  airfe_ctx tmp;   // only as an allocation list holder
  uint16_t *dq = dupload(&tmp, hq), *dk = dupload(&tmp, hk), *dv = dupload(&tmp, hvt);
  uint16_t* dout = dalloc<uint16_t>(&tmp, ((size_t)S * Np + 128) * H * 64);
  std::vector<int> hl(lens, lens + S);
  int* dl = dupload(&tmp, hl);
  int rc = 0;
  if (!dq || !dk || !dv || !dout || !dl) rc = fail(c, "debug_attention: allocation failed");
  if (!rc) {
    launch_attention32(prec, dq, dk, dv, dout, dl, S, H, Np, cross, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess || launch_status(c)) rc = rc ? rc : fail(c, "debug_attention: kernel failed");
  }
  if (!rc) {
    std::vector<uint16_t> ho((size_t)S * Np * H * 64);
    (void)hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost);
    for (int s = 0; s < S; ++s)
      for (int i = 0; i < n; ++i)
        for (int f = 0; f < H * 64; ++f) {
          const uint16_t u = ho[((size_t)s * Np + i) * H * 64 + f];
          out[((size_t)s * n + i) * H * 64 + f] = back2(u, prec);
        }
  }
  for (void* p : tmp.allocs) (void)hipFree(p);
  return rc;
} AIRFE_CATCH(c)

}  // extern "C"
