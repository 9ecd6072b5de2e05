// airfe_seq — MapBuilder::ExtractFeatureThread's loop (/root/reference/src/map_builder.cc:55-147) over S sequences in lock-step, as a native driver of the
// device-resident batch entries (include/airfe_seq.h).  Host side: branch sets, keyframe policy, result records.  Device side: the airfe_*_batch_dev entries
// + ONE copy kernel (seq_copy_jobs_kernel) that does every gather / scatter / pack-to-pinned-memory of a time-step from a job list.
#include "airfe_host.h"
#include "../../include/airfe_seq.h"

#include <chrono>
#include <cmath>
#include <memory>

namespace {

// one copy: rows x row_bytes bytes from src to dst; rows = min(*cnt, cap) when cnt is given (a device count the host does not know yet), else `cap`
struct SeqJob {
  const void* src;
  void* dst;
  const int* cnt;
  uint32_t row_bytes;
  uint32_t cap;
  const unsigned long long* off;      // != nullptr: the rows go to dst + *off (airfe_pack_rows_dev: offsets computed on the device from the counts)
};

// Unit (j, c) = every `chunks`-th 4-KiB run of job j; a workgroup takes units blockIdx.x, blockIdx.x + gridDim.x, ...  (the grid is capped: copies into pinned host memory
// are PCIe-bound — a few dozen workgroups keep enough stores in flight, thousands would sit on CUs that the next step's kernels want).  Sources and destinations are
// device memory or host-mapped pinned memory (the packed results are written straight into the staging set the host reads after its synchronisation: only the valid
// rows cross PCIe).  The job list itself is read from host-mapped memory.  All sizes are multiples of 4 bytes; 16-byte accesses where both addresses allow it.
__global__ __launch_bounds__(256) void seq_copy_jobs_kernel(const SeqJob* __restrict__ jobs, int njobs, int chunks) {
  for (int u = blockIdx.x; u < njobs * chunks; u += gridDim.x) {
    const int ji = u / chunks, ck = u - ji * chunks;
    const SeqJob j = jobs[ji];
    uint32_t rows = j.cap;
    if (j.cnt) {
      const int c = *j.cnt;
      rows = c < 0 ? 0u : ((uint32_t)c < rows ? (uint32_t)c : rows);
    }
    const size_t bytes = (size_t)rows * j.row_bytes;
    const char* s = reinterpret_cast<const char*>(j.src);
    char* d = reinterpret_cast<char*>(j.dst) + (j.off ? *j.off : 0ull);
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
      const size_t q = bytes / 16;
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (size_t i = (size_t)ck * 256 + threadIdx.x; i < q; i += (size_t)chunks * 256) d4[i] = s4[i];
      const size_t done = q * 16;
      if (ck == 0 && threadIdx.x < (bytes - done) / 4)
        reinterpret_cast<uint32_t*>(d + done)[threadIdx.x] = reinterpret_cast<const uint32_t*>(s + done)[threadIdx.x];
    } else {
      const size_t q = bytes / 4;
      const uint32_t* s1 = reinterpret_cast<const uint32_t*>(s);
      uint32_t* d1 = reinterpret_cast<uint32_t*>(d);
      for (size_t i = (size_t)ck * 256 + threadIdx.x; i < q; i += (size_t)chunks * 256) d1[i] = s1[i];
    }
  }
}

// off[j] = where job j's rows start in the packed block (16-byte aligned), off[njobs] = the block's size: an exclusive scan of min(*cnt, cap) * row_bytes, one workgroup
__global__ __launch_bounds__(256) void seq_offsets_kernel(const SeqJob* __restrict__ jobs, int njobs, unsigned long long* __restrict__ off) {
  __shared__ unsigned long long part[256];
  const int tid = threadIdx.x, per = (njobs + 255) / 256, j0 = tid * per, j1 = min(j0 + per, njobs);
  auto size_of = [&](int ji) -> unsigned long long {
    const SeqJob j = jobs[ji];
    uint32_t rows = j.cap;
    if (j.cnt) {
      const int c = *j.cnt;
      rows = c < 0 ? 0u : ((uint32_t)c < rows ? (uint32_t)c : rows);
    }
    return ((unsigned long long)rows * j.row_bytes + 15ull) & ~15ull;
  };
  unsigned long long sum = 0;
  for (int ji = j0; ji < j1; ++ji) sum += size_of(ji);
  part[tid] = sum;
  __syncthreads();
  unsigned long long base = 0;
  for (int t = 0; t < tid; ++t) base += part[t];
  for (int ji = j0; ji < j1; ++ji) {
    off[ji] = base;
    base += size_of(ji);
  }
  if (tid == 255) off[njobs] = base;
}

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// `_init`, `_insert_next_keyframe`, `_last_keyframe_feature` of MapBuilder (include/map_builder.h) for one sequence; of the last keyframe's features the host keeps
// what AddKeyframeCheck reads: the count and the (x, y) of every keypoint
struct LoopState {
  bool init = false, insert_next = false, has_ref = false;
  int ref_n = 0;
  std::vector<float> ref_xy;
};

// where a sequence's left rows of THIS time-step are on the device and in the staging set
struct CurRows {
  const float* d_rows = nullptr;
  const int* d_n = nullptr;
};

struct Staging {          // one pinned block, carved: everything the host side of the loop reads
  uint8_t* base = nullptr;
  float *cur = nullptr, *kr = nullptr, *kjunc = nullptr, *ksc = nullptr, *tsc = nullptr, *pr = nullptr, *psc = nullptr;
  double* klines = nullptr;
  int32_t *kidx = nullptr, *tidx = nullptr, *pidx = nullptr;
  int* counts = nullptr;  // [cur_n S | knr S | knlines 2S | knjunc S | knm S | kfound 3S | tnm S | pnr S | pnm S]
};

}  // namespace

struct airfe_seq {
  airfe_ctx *kf = nullptr, *nf = nullptr;
  int S = 0, K = 0, CL = 0, CJ = 0, device = 0;
  airfe_seq_policy pol{};
  std::string err;
  hipStream_t stream = nullptr, stream_k = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<LoopState> st;
  std::vector<void*> dev_allocs;
  // device: per-branch batches
  uint8_t *imgL = nullptr, *imgR = nullptr;       // gathered images: [S] keyframe candidates' left | right, [S] normal frames' left (imgL second half), promotions' right (imgR second half)
  size_t img_cap = 0;                             // bytes per image slot
  float *ref = nullptr, *kl = nullptr, *kr = nullptr, *kjunc = nullptr, *ksc = nullptr, *nfeat = nullptr, *tref = nullptr, *tcur = nullptr, *tsc = nullptr, *pr = nullptr, *psc = nullptr;
  double* klines = nullptr;
  int32_t *kidx = nullptr, *tidx = nullptr, *pidx = nullptr;
  int* dcounts = nullptr;                         // [ref_n S | knl S | knr S | knlines 2S | knjunc S | knm S | kfound 3S | nn S | tref_n S | tcur_n S | tnm S | pnr S | pnm S]
  int *ref_n = nullptr, *knl = nullptr, *knr = nullptr, *knlines = nullptr, *knjunc = nullptr, *knm = nullptr, *kfound = nullptr, *nn = nullptr, *tref_n = nullptr,
      *tcur_n = nullptr, *tnm = nullptr, *pnr = nullptr, *pnm = nullptr;
  bool ext_t = false;
  // pinned: two staging sets + a ring of job lists
  Staging stg[2];
  int flip = 0;
  SeqJob* jobs_h = nullptr;
  SeqJob* jobs_d = nullptr;
  int job_slots = 0, job_cap = 0, job_slot = 0;
  std::vector<SeqJob> jl;
  // the time-step in flight
  bool in_flight = false;
  std::vector<int> kset, nset, tset, pset, newkf, kpos, tpos;
  std::vector<CurRows> cur;
  const uint8_t* R_step = nullptr;
  int h = 0, w = 0, stride = 0;
  size_t img_stride = 0;
  double t_queue = 0, t_wait = 0, t_host = 0;
  int syncs = 0, steps = 0;
};

namespace {

thread_local std::string g_seq_err;
int sfail(airfe_seq* s, const std::string& m) {
  if (s) s->err = m;
  g_seq_err = m;
  return 1;
}
int sfail_noexcept(airfe_seq* s, const char* what, const char* detail) noexcept {
  try {
    return sfail(s, std::string("airfe_seq: C++ exception at the C boundary (") + what + "): " + (detail ? detail : "unknown"));
  } catch (...) {
  }
  return 1;
}
#define SEQ_CATCH(s)                                                                       \
  catch (const std::exception& e_) { return sfail_noexcept((s), __func__, e_.what()); }    \
  catch (...) { return sfail_noexcept((s), __func__, nullptr); }
#define AIRFE_CATCH(c)                                                                                 \
  catch (const std::exception& e_) { return airfe_host::fail_noexcept((c), __func__, e_.what()); }    \
  catch (...) { return airfe_host::fail_noexcept((c), __func__, nullptr); }
#define SEQ_HIP(s, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return sfail((s), std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define SEQ_CTX(s, ctx, expr)                                                              \
  do {                                                                                     \
    if ((expr) != 0) return sfail((s), std::string(#ctx ": ") + airfe_last_error(ctx));    \
  } while (0)

constexpr size_t ROW = (size_t)AIRFE_FEAT_DIM * 4;      // one feature row: 259 floats

template <class T>
T* seq_dalloc(airfe_seq* s, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T));
  s->dev_allocs.push_back(p);
  return reinterpret_cast<T*>(p);
}

// the job list `s->jl` -> the next slot of the pinned ring -> one launch on `st`.  A slot is reused job_slots launches later: every time-step ends behind a stream
// synchronisation and queues fewer launches than that.
int launch_jobs(airfe_seq* s, hipStream_t st) {
  const int n = (int)s->jl.size();
  if (n == 0) return 0;
  if (n > s->job_cap) return sfail(s, "airfe_seq: job list overflow");
  const int slot = s->job_slot;
  s->job_slot = (slot + 1) % s->job_slots;
  memcpy(s->jobs_h + (size_t)slot * s->job_cap, s->jl.data(), (size_t)n * sizeof(SeqJob));
  size_t big = 0;
  for (const SeqJob& j : s->jl) big = std::max(big, (size_t)j.cap * j.row_bytes);
  const int chunks = (int)std::min<size_t>(16, std::max<size_t>(1, big / 32768));
  hipLaunchKernelGGL(seq_copy_jobs_kernel, dim3(std::min(n * chunks, 256)), dim3(256), 0, st, s->jobs_d + (size_t)slot * s->job_cap, n, chunks);
  s->jl.clear();
  SEQ_HIP(s, hipGetLastError());
  return 0;
}
inline void job(airfe_seq* s, const void* src, void* dst, const int* cnt, size_t row_bytes, size_t cap) {
  s->jl.push_back(SeqJob{src, dst, cnt, (uint32_t)row_bytes, (uint32_t)cap, nullptr});
}

// Frame::AddRightFeatures' count (src/frame.cc:141-172): stereo matches inside the camera's band whose signed parallax is inside it too
int good_stereo_points(const airfe_seq_policy& p, const float* fl, const float* fr, const int32_t* idx, int m) {
  int good = 0;
  for (int i = 0; i < m; ++i) {
    const float* a = fl + (size_t)idx[2 * i] * AIRFE_FEAT_DIM;
    const float* b = fr + (size_t)idx[2 * i + 1] * AIRFE_FEAT_DIM;
    const double dx = std::abs(a[1] - b[1]);             // std::abs(float - float) -> double, :150-151
    const double dy = std::abs(a[2] - b[2]);
    if (!(dx > p.min_x_diff && dx < p.max_x_diff && dy <= p.max_y_diff)) continue;      // :153
    const double parallax = a[1] - b[1];                  // :165
    if (parallax < p.max_x_diff && parallax > p.min_x_diff) ++good;                       // :167-171
  }
  return good;
}

// MapBuilder::AddKeyframeCheck (src/map_builder.cc:429-466), UseIMU() == false.  ref_xy: (x, y) per reference keypoint; cur: 259-float rows; idx [m][2]
int add_keyframe_check(const airfe_seq_policy& p, const float* ref_xy, int ref_n, const float* cur, int cur_n, const int32_t* idx, int m) {
  if (m < p.min_num_match) return 0;                                                                       // :431
  const float thr = p.tracking_point_rate;
  if ((float)m / ref_n < thr || (float)m / cur_n < thr || m < p.max_num_match) return 1;                   // :443-445
  // (parallax * parallax.transpose()).sum(): ALL four entries of the 2 x 2 product, accumulated in float (:447-458)
  float g00 = 0.f, g01 = 0.f, g11 = 0.f;
  for (int i = 0; i < m; ++i) {
    const float px = ref_xy[2 * idx[2 * i]] - cur[(size_t)idx[2 * i + 1] * AIRFE_FEAT_DIM + 1];
    const float py = ref_xy[2 * idx[2 * i] + 1] - cur[(size_t)idx[2 * i + 1] * AIRFE_FEAT_DIM + 2];
    g00 += px * px; g01 += px * py; g11 += py * py;
  }
  const double average_parallax = (double)((g00 + g01) + (g01 + g11)) / m;                                 // :458
  const double image_size = (double)(p.image_height * p.image_width);
  if (average_parallax > image_size * (double)p.tracking_parallax_rate * (double)p.tracking_parallax_rate) return 1;     // :461
  return 2;
}

int seq_enter(airfe_seq* s) {
  int d = -1;
  if (hipGetDevice(&d) == hipSuccess && d == s->device) return 0;
  if (hipSetDevice(s->device) != hipSuccess) return sfail(s, "hipSetDevice failed");
  return 0;
}

}  // namespace

extern "C" {

void airfe_seq_default_policy(airfe_seq_policy* p) {
  if (!p) return;
  *p = airfe_seq_policy{90, 30, 80, 0.65f, 0.1f, 1.0, 200.0, 5.0, 752, 480};
}

// ---- the copy kernel on its own (include/airfe.h): valid rows of device buffers -> device or host-mapped memory, one launch
static int copy_rows_impl(airfe_ctx* c, int njobs, const void* const* src, void* const* dst, void* d_packed, unsigned long long* d_off, const int* const* cnt,
                          const uint32_t* row_bytes, const uint32_t* cap, void* stream, const char* who) {
  int d = -1;
  if (!(hipGetDevice(&d) == hipSuccess && d == c->cfg.device) && hipSetDevice(c->cfg.device) != hipSuccess) return fail(c, "hipSetDevice(cfg.device) failed");
  if (njobs < 0 || (njobs > 0 && (!src || !row_bytes || !cap || (!dst && !(d_packed && d_off))))) return fail(c, std::string(who) + ": bad argument");
  if (njobs == 0) return 0;
  constexpr int SLOTS = 8;
  if (njobs > c->copy_ring_cap) {               // (grows by replacement; launches that still read the old ring are drained first)
    HIPCHK(c, hipDeviceSynchronize());
    if (c->copy_ring) (void)hipHostFree(c->copy_ring);
    c->copy_ring = nullptr; c->copy_ring_cap = 0;
    const int capn = std::max(njobs, 512);
    HIPCHK(c, hipHostMalloc(&c->copy_ring, (size_t)SLOTS * capn * sizeof(SeqJob), hipHostMallocMapped | hipHostMallocCoherent));
    c->copy_ring_cap = capn; c->copy_ring_slot = 0;
  }
  SeqJob* slot = reinterpret_cast<SeqJob*>(c->copy_ring) + (size_t)c->copy_ring_slot * c->copy_ring_cap;
  c->copy_ring_slot = (c->copy_ring_slot + 1) % SLOTS;      // a slot is rewritten 8 launches later: the caller synchronises its stream more often than that (documented)
  size_t big = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!src[j] || (dst && !dst[j]) || row_bytes[j] % 4) return fail(c, std::string(who) + ": null pointer or a row size that is not a multiple of 4");
    slot[j] = SeqJob{src[j], dst ? dst[j] : d_packed, cnt ? cnt[j] : nullptr, row_bytes[j], cap[j], dst ? nullptr : d_off + j};
    big = std::max(big, (size_t)cap[j] * row_bytes[j]);
  }
  void* dp = nullptr;
  HIPCHK(c, hipHostGetDevicePointer(&dp, slot, 0));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (!dst) hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const SeqJob*>(dp), njobs, d_off);
  const int chunks = (int)std::min<size_t>(16, std::max<size_t>(1, big / 32768));
  // a capped grid: copies into pinned memory wait on PCIe — a few dozen workgroups keep enough stores in flight (tools/microbench/pcie_kernel_write.hip: 64 reach the
  // 55 GB/s of the copy engines), thousands would sit on CUs beside the next step's kernels
  hipLaunchKernelGGL(seq_copy_jobs_kernel, dim3(std::min(njobs * chunks, std::max(dst ? c->copy_wgs : 1024, 1))), dim3(256), 0, st, reinterpret_cast<const SeqJob*>(dp), njobs,
                     chunks);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int airfe_copy_rows_dev(airfe_ctx* c, int njobs, const void* const* src, void* const* dst, const int* const* cnt, const uint32_t* row_bytes, const uint32_t* cap,
                        void* stream) try {
  if (!c) return 1;
  if (njobs > 0 && !dst) return fail(c, "airfe_copy_rows_dev: bad argument");
  return copy_rows_impl(c, njobs, src, dst, nullptr, nullptr, cnt, row_bytes, cap, stream, "airfe_copy_rows_dev");
} AIRFE_CATCH(c)

int airfe_pack_rows_dev(airfe_ctx* c, int njobs, const void* const* src, const int* const* cnt, const uint32_t* row_bytes, const uint32_t* cap, void* d_packed,
                        unsigned long long* d_offsets, void* stream) try {
  if (!c) return 1;
  if (njobs > 0 && (!d_packed || !d_offsets)) return fail(c, "airfe_pack_rows_dev: bad argument");
  return copy_rows_impl(c, njobs, src, nullptr, d_packed, d_offsets, cnt, row_bytes, cap, stream, "airfe_pack_rows_dev");
} AIRFE_CATCH(c)

int airfe_seq_add_keyframe_check(const airfe_seq_policy* p, const float* ref_feat, int ref_n, const float* cur_feat, int cur_n, const int32_t* idx, int m) try {
  if (!p || m < 0 || ref_n < 0 || cur_n < 0 || (m > 0 && (!ref_feat || !cur_feat || !idx))) return -1;
  std::vector<float> xy((size_t)2 * ref_n);
  for (int k = 0; k < ref_n; ++k) {
    xy[2 * k] = ref_feat[(size_t)k * AIRFE_FEAT_DIM + 1];
    xy[2 * k + 1] = ref_feat[(size_t)k * AIRFE_FEAT_DIM + 2];
  }
  return add_keyframe_check(*p, xy.data(), ref_n, cur_feat, cur_n, idx, m);
} catch (...) { return -1; }

int airfe_seq_good_stereo_points(const airfe_seq_policy* p, const float* feat_left, const float* feat_right, const int32_t* idx, int m) try {
  if (!p || m < 0 || (m > 0 && (!feat_left || !feat_right || !idx))) return -1;
  return good_stereo_points(*p, feat_left, feat_right, idx, m);
} catch (...) { return -1; }

const char* airfe_seq_last_error(const airfe_seq* s) { return s ? s->err.c_str() : g_seq_err.c_str(); }

void airfe_seq_destroy(airfe_seq* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->stream_k) (void)hipStreamSynchronize(s->stream_k);
  for (void* p : s->dev_allocs) (void)hipFree(p);
  for (Staging& g : s->stg)
    if (g.base) (void)hipHostFree(g.base);
  if (s->jobs_h) (void)hipHostFree(s->jobs_h);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  if (s->stream_k) (void)hipStreamDestroy(s->stream_k);
  delete s;
}

int airfe_seq_create(airfe_ctx* kf, airfe_ctx* nf, int S, const airfe_seq_policy* policy, int cap_lines, int cap_junc, int32_t* d_tidx, float* d_tscore, int* d_tn,
                     airfe_seq** out) try {
  if (!kf || !nf || !out || S < 1) return sfail(nullptr, "airfe_seq_create: null argument or S < 1");
  if (kf->cfg.max_keypoints != nf->cfg.max_keypoints) return sfail(nullptr, "airfe_seq_create: both contexts must be created with the same max_keypoints");
  if (kf->cfg.device != nf->cfg.device) return sfail(nullptr, "airfe_seq_create: both contexts must live on the same device");
  if (kf->cfg.max_batch < S || nf->cfg.max_batch < S) return sfail(nullptr, "airfe_seq_create: cfg.max_batch of both contexts must be >= S");
  if (!airfe_has_line_branch(kf)) return sfail(nullptr, "airfe_seq_create: the keyframe context needs the PLNet line branch (detector pack with line.* tensors + stage 1)");
  if (!kf->has_lg || !nf->has_lg || !nf->has_sp) return sfail(nullptr, "airfe_seq_create: both contexts need LightGlue weights, the normal-frame context a detector");
  if ((d_tidx != nullptr) != (d_tscore != nullptr) || (d_tidx != nullptr) != (d_tn != nullptr)) return sfail(nullptr, "airfe_seq_create: give all three temporal buffers or none");
  if (cap_lines < 1 || cap_junc < 1) return sfail(nullptr, "airfe_seq_create: cap_lines / cap_junc < 1");
  std::unique_ptr<airfe_seq, void (*)(airfe_seq*)> g(new airfe_seq, airfe_seq_destroy);
  airfe_seq* s = g.get();
  s->kf = kf; s->nf = nf; s->S = S; s->K = kf->cfg.max_keypoints; s->CL = cap_lines; s->CJ = cap_junc; s->device = kf->cfg.device;
  if (policy) s->pol = *policy; else airfe_seq_default_policy(&s->pol);
  if (seq_enter(s)) return 1;
  SEQ_HIP(s, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  SEQ_HIP(s, hipStreamCreateWithFlags(&s->stream_k, hipStreamNonBlocking));
  SEQ_HIP(s, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  SEQ_HIP(s, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  s->st.resize(S); s->cur.resize(S);
  const size_t K = s->K, CL = cap_lines, CJ = cap_junc;
  const size_t F = (size_t)S * K * AIRFE_FEAT_DIM;
  s->ref = seq_dalloc<float>(s, F); s->kl = seq_dalloc<float>(s, F); s->kr = seq_dalloc<float>(s, F); s->nfeat = seq_dalloc<float>(s, F);
  s->tref = seq_dalloc<float>(s, F); s->tcur = seq_dalloc<float>(s, F); s->pr = seq_dalloc<float>(s, F);
  s->kjunc = seq_dalloc<float>(s, (size_t)S * CJ * AIRFE_FEAT_DIM);
  s->klines = seq_dalloc<double>(s, (size_t)2 * S * CL * 4);
  s->kidx = seq_dalloc<int32_t>(s, (size_t)S * K * 2); s->pidx = seq_dalloc<int32_t>(s, (size_t)S * K * 2);
  s->ksc = seq_dalloc<float>(s, (size_t)S * K); s->psc = seq_dalloc<float>(s, (size_t)S * K);
  s->ext_t = d_tidx != nullptr;
  s->tidx = s->ext_t ? d_tidx : seq_dalloc<int32_t>(s, (size_t)S * K * 2);
  s->tsc = s->ext_t ? d_tscore : seq_dalloc<float>(s, (size_t)S * K);
  s->dcounts = seq_dalloc<int>(s, (size_t)17 * S);
  int* c = s->dcounts;
  s->ref_n = c; c += S; s->knl = c; c += S; s->knr = c; c += S; s->knlines = c; c += 2 * S; s->knjunc = c; c += S; s->knm = c; c += S; s->kfound = c; c += 3 * S;
  s->nn = c; c += S; s->tref_n = c; c += S; s->tcur_n = c; c += S; s->tnm = s->ext_t ? d_tn : c; c += S; s->pnr = c; c += S; s->pnm = c; c += S;
  if (!s->ref || !s->kl || !s->kr || !s->nfeat || !s->tref || !s->tcur || !s->pr || !s->kjunc || !s->klines || !s->kidx || !s->pidx || !s->ksc || !s->psc || !s->tidx ||
      !s->tsc || !s->dcounts)
    return sfail(s, "airfe_seq_create: device allocation failed");
  // two staging sets of pinned, device-visible memory
  for (Staging& t : s->stg) {
    const size_t bytes = 4 * (F * 4 + 256) + (size_t)S * CJ * ROW + (size_t)2 * S * CL * 32 + 6 * ((size_t)S * K * 8 + 256) + (size_t)14 * S * 4 + 16 * 256;
    // host-mapped, coherent (like the contexts' saturation words): the packing kernel's stores land in host memory, the host reads them after its stream
    // synchronisation; on this runtime a pinned block's device address is its host address (checked)
    SEQ_HIP(s, hipHostMalloc(reinterpret_cast<void**>(&t.base), bytes, hipHostMallocMapped | hipHostMallocCoherent));
    void* dp = nullptr;
    SEQ_HIP(s, hipHostGetDevicePointer(&dp, t.base, 0));
    if (dp != (void*)t.base) return sfail(s, "airfe_seq_create: pinned memory is not mapped at its host address");
    memset(t.base, 0, bytes);
    uint8_t* p = t.base;
    auto take = [&](size_t b) { uint8_t* r = p; p += (b + 255) & ~(size_t)255; return r; };
    t.cur = (float*)take(F * 4); t.kr = (float*)take(F * 4); t.pr = (float*)take(F * 4);
    t.kjunc = (float*)take((size_t)S * CJ * ROW); t.klines = (double*)take((size_t)2 * S * CL * 32);
    t.kidx = (int32_t*)take((size_t)S * K * 8); t.tidx = (int32_t*)take((size_t)S * K * 8); t.pidx = (int32_t*)take((size_t)S * K * 8);
    t.ksc = (float*)take((size_t)S * K * 4); t.tsc = (float*)take((size_t)S * K * 4); t.psc = (float*)take((size_t)S * K * 4);
    t.counts = (int*)take((size_t)14 * S * 4);
    if ((size_t)(p - t.base) > bytes) return sfail(s, "airfe_seq_create: staging layout overflow");
  }
  s->job_slots = 16; s->job_cap = 16 * S + 64;
  SEQ_HIP(s, hipHostMalloc(reinterpret_cast<void**>(&s->jobs_h), (size_t)s->job_slots * s->job_cap * sizeof(SeqJob), hipHostMallocMapped | hipHostMallocCoherent));
  SEQ_HIP(s, hipHostGetDevicePointer(reinterpret_cast<void**>(&s->jobs_d), s->jobs_h, 0));
  s->jl.reserve(s->job_cap);
  // seq_dalloc clears its blocks with hipMemset on the NULL stream; the driver's streams are non-blocking (they do not order against it): nothing of the driver
  // may run before those clears have finished
  SEQ_HIP(s, hipDeviceSynchronize());
  *out = g.release();
  return 0;
} SEQ_CATCH(nullptr)

void* airfe_seq_stream(airfe_seq* s) { return s ? (void*)s->stream : nullptr; }

int airfe_seq_wall_split(airfe_seq* s, double* queue_s, double* wait_s, double* host_s, int* host_syncs, int* steps) try {
  if (!s) return 1;
  if (queue_s) *queue_s = s->t_queue;
  if (wait_s) *wait_s = s->t_wait;
  if (host_s) *host_s = s->t_host;
  if (host_syncs) *host_syncs = s->syncs;
  if (steps) *steps = s->steps;
  s->t_queue = s->t_wait = s->t_host = 0;
  s->syncs = s->steps = 0;
  return 0;
} SEQ_CATCH(s)

int airfe_seq_begin(airfe_seq* s, const uint8_t* d_left, const uint8_t* d_right, int h, int w, int stride, size_t img_stride) try {
  if (!s) return 1;
  if (seq_enter(s)) return 1;
  if (s->in_flight) return sfail(s, "airfe_seq_begin: the previous time-step was not ended (airfe_seq_end)");
  if (!d_left || !d_right || h < 1 || w < 1 || stride < w) return sfail(s, "airfe_seq_begin: bad image arguments");
  const size_t ibytes = (size_t)h * stride;
  if (ibytes % 4 || img_stride < ibytes) return sfail(s, "airfe_seq_begin: h * stride must be a multiple of 4 and img_stride >= h * stride");
  const auto t_a = clk::now();
  const int S = s->S, K = s->K;
  if (ibytes > s->img_cap) {          // first step (or larger images): the gathered-image blocks
    SEQ_HIP(s, hipStreamSynchronize(s->stream));
    SEQ_HIP(s, hipStreamSynchronize(s->stream_k));
    s->imgL = seq_dalloc<uint8_t>(s, (size_t)2 * S * ibytes);
    s->imgR = seq_dalloc<uint8_t>(s, (size_t)2 * S * ibytes);
    if (!s->imgL || !s->imgR) return sfail(s, "airfe_seq_begin: device allocation failed (images)");
    SEQ_HIP(s, hipDeviceSynchronize());      // (the blocks' clears ran on the NULL stream: see airfe_seq_create — a gather queued now must not be overtaken by them)
    s->img_cap = ibytes;
  }
  s->h = h; s->w = w; s->stride = stride; s->img_stride = img_stride; s->R_step = d_right;
  s->flip ^= 1;
  Staging& G = s->stg[s->flip];
  s->kset.clear(); s->nset.clear(); s->tset.clear();
  for (int i = 0; i < S; ++i) {
    const LoopState& q = s->st[i];
    ((!q.init || q.insert_next) ? s->kset : s->nset).push_back(i);                 // map_builder.cc:83
    if (q.init) s->tset.push_back(i);
  }
  const int nk = (int)s->kset.size(), nn = (int)s->nset.size(), nt = (int)s->tset.size();
  // the images of both branches, gathered into contiguous batches: one launch
  uint8_t* Ln = s->imgL + (size_t)S * ibytes;
  for (int j = 0; j < nk; ++j) {
    job(s, d_left + (size_t)s->kset[j] * img_stride, s->imgL + (size_t)j * ibytes, nullptr, ibytes, 1);
    job(s, d_right + (size_t)s->kset[j] * img_stride, s->imgR + (size_t)j * ibytes, nullptr, ibytes, 1);
  }
  for (int j = 0; j < nn; ++j) job(s, d_left + (size_t)s->nset[j] * img_stride, Ln + (size_t)j * ibytes, nullptr, ibytes, 1);
  if (launch_jobs(s, s->stream)) return 1;
  if (nk) {     // keyframe candidates: PLNet on both images + the stereo match, one batch (map_builder.cc:85-86) — on its own stream, beside the normal frames' batch
    SEQ_HIP(s, hipEventRecord(s->ev_fork, s->stream));
    SEQ_HIP(s, hipStreamWaitEvent(s->stream_k, s->ev_fork, 0));
    SEQ_CTX(s, s->kf, airfe_stereo_plnet_batch_dev(s->kf, s->imgL, s->imgR, nk, h, w, stride, ibytes, s->kl, s->kr, K, s->knl, s->knr, s->klines, s->CL, s->knlines,
                                                   s->kjunc, s->CJ, s->knjunc, s->kfound, s->kidx, s->ksc, K, s->knm, s->stream_k));
    SEQ_HIP(s, hipEventRecord(s->ev_join, s->stream_k));
    for (int j = 0; j < nk; ++j) s->cur[s->kset[j]] = CurRows{s->kl + (size_t)j * K * AIRFE_FEAT_DIM, s->knl + j};
  }
  if (nn) {     // normal frames: SuperPoint on the left image, one batch (:94)
    SEQ_CTX(s, s->nf, airfe_detect_points_batch_dev(s->nf, Ln, nn, h, w, stride, ibytes, s->nfeat, K, s->nn, s->stream));
    for (int j = 0; j < nn; ++j) s->cur[s->nset[j]] = CurRows{s->nfeat + (size_t)j * K * AIRFE_FEAT_DIM, s->nn + j};
  }
  if (nk) SEQ_HIP(s, hipStreamWaitEvent(s->stream, s->ev_join, 0));
  if (nt) {     // the temporal match of every initialised sequence, one LightGlue batch (:100-101)
    for (int j = 0; j < nt; ++j) {
      const int i = s->tset[j];
      job(s, s->ref + (size_t)i * K * AIRFE_FEAT_DIM, s->tref + (size_t)j * K * AIRFE_FEAT_DIM, s->ref_n + i, ROW, K);
      job(s, s->ref_n + i, s->tref_n + j, nullptr, 4, 1);
      job(s, s->cur[i].d_rows, s->tcur + (size_t)j * K * AIRFE_FEAT_DIM, s->cur[i].d_n, ROW, K);
      job(s, s->cur[i].d_n, s->tcur_n + j, nullptr, 4, 1);
    }
    if (launch_jobs(s, s->stream)) return 1;
    SEQ_CTX(s, s->nf, airfe_match_lightglue_batch_dev(s->nf, s->tref, s->tref_n, s->tcur, s->tcur_n, nt, K, s->tidx, s->tsc, K, s->tnm, s->stream));
  }
  // everything the host side of the loop reads -> the pinned staging set, valid rows only: one launch
  int* C = G.counts;
  int *h_cur_n = C, *h_knr = C + S, *h_knlines = C + 2 * S, *h_knjunc = C + 4 * S, *h_knm = C + 5 * S, *h_kfound = C + 6 * S, *h_tnm = C + 9 * S;
  for (int i = 0; i < S; ++i) {
    job(s, s->cur[i].d_rows, G.cur + (size_t)i * K * AIRFE_FEAT_DIM, s->cur[i].d_n, ROW, K);
    job(s, s->cur[i].d_n, h_cur_n + i, nullptr, 4, 1);
  }
  if (nk) {
    for (int j = 0; j < nk; ++j) {
      job(s, s->kr + (size_t)j * K * AIRFE_FEAT_DIM, G.kr + (size_t)j * K * AIRFE_FEAT_DIM, s->knr + j, ROW, K);
      job(s, s->klines + (size_t)j * s->CL * 4, G.klines + (size_t)j * s->CL * 4, s->knlines + j, 32, s->CL);
      job(s, s->klines + (size_t)(nk + j) * s->CL * 4, G.klines + (size_t)(nk + j) * s->CL * 4, s->knlines + nk + j, 32, s->CL);
      job(s, s->kjunc + (size_t)j * s->CJ * AIRFE_FEAT_DIM, G.kjunc + (size_t)j * s->CJ * AIRFE_FEAT_DIM, s->knjunc + j, ROW, s->CJ);
      job(s, s->kidx + (size_t)j * K * 2, G.kidx + (size_t)j * K * 2, s->knm + j, 8, K);
      job(s, s->ksc + (size_t)j * K, G.ksc + (size_t)j * K, s->knm + j, 4, K);
    }
    job(s, s->knr, h_knr, nullptr, 4, nk);
    job(s, s->knlines, h_knlines, nullptr, 4, 2 * nk);
    job(s, s->knjunc, h_knjunc, nullptr, 4, nk);
    job(s, s->knm, h_knm, nullptr, 4, nk);
    job(s, s->kfound, h_kfound, nullptr, 4, 3 * nk);
  }
  if (nt) {
    for (int j = 0; j < nt; ++j) {
      job(s, s->tidx + (size_t)j * K * 2, G.tidx + (size_t)j * K * 2, s->tnm + j, 8, K);
      job(s, s->tsc + (size_t)j * K, G.tsc + (size_t)j * K, s->tnm + j, 4, K);
    }
    job(s, s->tnm, h_tnm, nullptr, 4, nt);
  }
  if (launch_jobs(s, s->stream)) return 1;
  s->in_flight = true;
  s->t_queue += secs(t_a, clk::now());
  return 0;
} SEQ_CATCH(s)

int airfe_seq_end(airfe_seq* s, airfe_seq_frame* out) try {
  if (!s) return 1;
  if (seq_enter(s)) return 1;
  if (!s->in_flight) return sfail(s, "airfe_seq_end: no time-step in flight (airfe_seq_begin)");
  if (!out) return sfail(s, "airfe_seq_end: null result array");
  s->in_flight = false;
  const int S = s->S, K = s->K;
  const airfe_seq_policy& pol = s->pol;
  Staging& G = s->stg[s->flip];
  int* C = G.counts;
  const int *h_cur_n = C, *h_knr = C + S, *h_knlines = C + 2 * S, *h_knjunc = C + 4 * S, *h_knm = C + 5 * S, *h_kfound = C + 6 * S, *h_tnm = C + 9 * S;
  int *h_pnr = C + 10 * S, *h_pnm = C + 11 * S;
  const int nk = (int)s->kset.size(), nt = (int)s->tset.size();
  auto t_b = clk::now();
  SEQ_HIP(s, hipStreamSynchronize(s->stream));
  auto t_c = clk::now();
  s->t_wait += secs(t_b, t_c);
  s->syncs++; s->steps++;
  // the asynchronous entries report through their contexts: an fp16 overflow of a detector (never keypoints of a poisoned score map), a failed launch
  if (saturation_status(s->kf)) return sfail(s, std::string("kf: ") + airfe_last_error(s->kf));
  if (saturation_status(s->nf)) return sfail(s, std::string("nf: ") + airfe_last_error(s->nf));
  for (int i = 0; i < S; ++i) {
    airfe_seq_frame& r = out[i];
    r = airfe_seq_frame{};
    r.enough_match = -1;
    r.n_right = r.n_lines_left = r.n_lines_right = r.n_junctions = r.n_stereo = r.n_matches = -1;
    r.n_left = std::min(std::max(h_cur_n[i], 0), K);
    r.features_left = G.cur + (size_t)i * K * AIRFE_FEAT_DIM;
  }
  for (int j = 0; j < 2 * nk; ++j)
    if (h_kfound[j] > s->CL || h_kfound[j] < 0)
      return sfail(s, "airfe_seq: line capacity overflow (cap_lines): " + std::to_string(h_kfound[j]) + " lines in the " + (j < nk ? "left" : "right") + " image of sequence " +
                          std::to_string(s->kset[j % nk]) + ", cap_lines = " + std::to_string(s->CL));
  for (int j = 0; j < nk; ++j)
    if (h_kfound[2 * nk + j] > s->CJ || h_kfound[2 * nk + j] < 0)
      return sfail(s, "airfe_seq: junction capacity overflow (cap_junc): " + std::to_string(h_kfound[2 * nk + j]) + " junctions in sequence " + std::to_string(s->kset[j]) +
                          ", cap_junc = " + std::to_string(s->CJ));
  for (int j = 0; j < nk; ++j) {
    airfe_seq_frame& r = out[s->kset[j]];
    r.candidate = 1;
    r.n_right = std::min(std::max(h_knr[j], 0), K); r.features_right = G.kr + (size_t)j * K * AIRFE_FEAT_DIM;
    r.n_lines_left = h_knlines[j]; r.lines_left = G.klines + (size_t)j * s->CL * 4;
    r.n_lines_right = h_knlines[nk + j]; r.lines_right = G.klines + (size_t)(nk + j) * s->CL * 4;
    r.n_junctions = h_knjunc[j]; r.junctions = G.kjunc + (size_t)j * s->CJ * AIRFE_FEAT_DIM;
    r.n_stereo = (r.n_left && r.n_right) ? std::min(std::max(h_knm[j], 0), K) : 0;        // MatchingPoints returns early on an empty side: src/point_matcher.cc:53-55
    r.stereo_idx = G.kidx + (size_t)j * K * 2; r.stereo_score = G.ksc + (size_t)j * K;
    r.good_stereo_point = good_stereo_points(pol, r.features_left, r.features_right, r.stereo_idx, r.n_stereo);
  }
  s->pset.clear();
  for (int j = 0; j < nt; ++j) {
    const int i = s->tset[j];
    airfe_seq_frame& r = out[i];
    const LoopState& q = s->st[i];
    r.n_matches = (q.ref_n && r.n_left) ? std::min(std::max(h_tnm[j], 0), K) : 0;
    r.matches_idx = G.tidx + (size_t)j * K * 2; r.matches_score = G.tsc + (size_t)j * K;
    r.enough_match = add_keyframe_check(pol, q.ref_xy.data(), q.ref_n, r.features_left, r.n_left, r.matches_idx, r.n_matches);      // :102
    if (!r.candidate && r.enough_match == 0) s->pset.push_back(i);
  }
  const int np = (int)s->pset.size();
  if (np) {       // promotions: SuperPoint on the right image + the stereo match, one batch each (:104-108)
    const auto t_q = clk::now();
    const size_t ibytes = (size_t)s->h * s->stride;
    uint8_t* Rp = s->imgR + (size_t)S * ibytes;
    for (int j = 0; j < np; ++j) {
      const int i = s->pset[j];
      job(s, s->R_step + (size_t)i * s->img_stride, Rp + (size_t)j * ibytes, nullptr, ibytes, 1);
      job(s, s->cur[i].d_rows, s->tcur + (size_t)j * K * AIRFE_FEAT_DIM, s->cur[i].d_n, ROW, K);
      job(s, s->cur[i].d_n, s->tcur_n + j, nullptr, 4, 1);
    }
    if (launch_jobs(s, s->stream)) return 1;
    SEQ_CTX(s, s->nf, airfe_detect_points_batch_dev(s->nf, Rp, np, s->h, s->w, s->stride, ibytes, s->pr, K, s->pnr, s->stream));
    SEQ_CTX(s, s->nf, airfe_match_lightglue_batch_dev(s->nf, s->tcur, s->tcur_n, s->pr, s->pnr, np, K, s->pidx, s->psc, K, s->pnm, s->stream));
    for (int j = 0; j < np; ++j) {
      job(s, s->pr + (size_t)j * K * AIRFE_FEAT_DIM, G.pr + (size_t)j * K * AIRFE_FEAT_DIM, s->pnr + j, ROW, K);
      job(s, s->pidx + (size_t)j * K * 2, G.pidx + (size_t)j * K * 2, s->pnm + j, 8, K);
      job(s, s->psc + (size_t)j * K, G.psc + (size_t)j * K, s->pnm + j, 4, K);
    }
    job(s, s->pnr, h_pnr, nullptr, 4, np);
    job(s, s->pnm, h_pnm, nullptr, 4, np);
    if (launch_jobs(s, s->stream)) return 1;
    const auto t_w = clk::now();
    SEQ_HIP(s, hipStreamSynchronize(s->stream));
    const auto t_e = clk::now();
    s->t_queue += secs(t_q, t_w); s->t_wait += secs(t_w, t_e);
    t_c += (t_e - t_q);                       // (the promotion pass is not host time)
    s->syncs++;
    if (saturation_status(s->nf)) return sfail(s, std::string("nf: ") + airfe_last_error(s->nf));
    for (int j = 0; j < np; ++j) {
      airfe_seq_frame& r = out[s->pset[j]];
      r.promoted = 1;                                                                       // :105-109
      r.n_right = std::min(std::max(h_pnr[j], 0), K); r.features_right = G.pr + (size_t)j * K * AIRFE_FEAT_DIM;
      r.n_stereo = (r.n_left && r.n_right) ? std::min(std::max(h_pnm[j], 0), K) : 0;
      r.stereo_idx = G.pidx + (size_t)j * K * 2; r.stereo_score = G.psc + (size_t)j * K;
      r.good_stereo_point = good_stereo_points(pol, r.features_left, r.features_right, r.stereo_idx, r.n_stereo);
    }
  }
  // map_builder.cc:99-141 per sequence
  s->newkf.clear();
  for (int i = 0; i < S; ++i) {
    airfe_seq_frame& r = out[i];
    LoopState& q = s->st[i];
    int frame_type = r.candidate ? (q.init ? 1 : 2) : 0;                                    // :88, :96
    if (q.init) {
      if (r.enough_match == 0) {
        if (r.good_stereo_point < 10) {                                                     // :111-117
          q.insert_next = true;
          frame_type = 0;
        } else {
          frame_type = 1;
          q.insert_next = false;
        }
      } else {
        q.insert_next = (r.enough_match == 1) && (frame_type == 0);                         // :119
      }
    } else {
      if (r.good_stereo_point < pol.min_init_stereo_feature) {                              // :122-125
        r.dropped = 1;
        r.frame_type = frame_type;
        continue;
      }
      q.init = true;                                                                        // :127-128
    }
    r.frame_type = frame_type;
    if (frame_type != 0) {                                                                  // :139-141 `_last_keyframe_feature = frame`
      q.has_ref = true;
      q.ref_n = r.n_left;
      q.ref_xy.resize((size_t)2 * r.n_left);
      for (int k = 0; k < r.n_left; ++k) {
        q.ref_xy[2 * k] = r.features_left[(size_t)k * AIRFE_FEAT_DIM + 1];
        q.ref_xy[2 * k + 1] = r.features_left[(size_t)k * AIRFE_FEAT_DIM + 2];
      }
      s->newkf.push_back(i);
    }
  }
  if (!s->newkf.empty()) {      // on the device, the new keyframes' rows become the reference rows
    const auto t_q = clk::now();
    for (int i : s->newkf) {
      job(s, s->cur[i].d_rows, s->ref + (size_t)i * K * AIRFE_FEAT_DIM, s->cur[i].d_n, ROW, K);
      job(s, s->cur[i].d_n, s->ref_n + i, nullptr, 4, 1);
    }
    if (launch_jobs(s, s->stream)) return 1;
    const auto t_e = clk::now();
    s->t_queue += secs(t_q, t_e);
    t_c += (t_e - t_q);
  }
  s->t_host += secs(t_c, clk::now());
  return 0;
} SEQ_CATCH(s)

int airfe_seq_step(airfe_seq* s, const uint8_t* d_left, const uint8_t* d_right, int h, int w, int stride, size_t img_stride, airfe_seq_frame* out) try {
  if (airfe_seq_begin(s, d_left, d_right, h, w, stride, img_stride)) return 1;
  return airfe_seq_end(s, out);
} SEQ_CATCH(s)

}  // extern "C"
