// airfe — weight packs: loading, slab packing (≙ TensorRT engine build, src/plnet.cpp:24-196) and the matcher arena (see airfe_host.h)
#include "airfe_host.h"

namespace airfe_host {

bool load_pack(const char* path, Pack& out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("cannot open weight pack ") + path; return false; }
  fseek(f, 0, SEEK_END);
  const long fsize = ftell(f);                     // every tensor's element count is bounded by what is left of the file
  fseek(f, 0, SEEK_SET);
  char magic[8];
  uint32_t count = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "AIRFEPK1", 8) == 0 && fread(&count, 4, 1, f) == 1;
  for (uint32_t i = 0; ok && i < count; ++i) {
    uint32_t nl = 0, nd = 0;
    ok = fread(&nl, 4, 1, f) == 1 && nl < 4096;
    if (!ok) break;
    std::string name(nl, '\0');
    ok = fread(&name[0], 1, nl, f) == nl && fread(&nd, 4, 1, f) == 1 && nd <= 8;
    if (!ok) break;
    Tensor t;
    size_t n = 1;
    for (uint32_t d = 0; d < nd; ++d) {
      uint32_t v = 0;
      ok = ok && fread(&v, 4, 1, f) == 1;
      t.dims.push_back((int)v);
      if (v > 0x7FFFFFFFu || (v != 0 && n > (size_t)0x7FFFFFFFFFFFull / v)) ok = false;      // dims are untrusted
      else n *= v;
    }
    const long pos = ftell(f);
    if (!ok || pos < 0 || fsize < pos || n > (size_t)(fsize - pos) / 4) { ok = false; break; }
    t.data.resize(n);
    ok = fread(t.data.data(), 4, n, f) == n;
    out[name] = std::move(t);
  }
  fclose(f);
  if (!ok) err = std::string("malformed weight pack ") + path;
  return ok;
}

// ---- 2-byte conversions (round to nearest even) on the host
uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t m = x & 0x7FFFFFu;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0));
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int shift = 14 - e;
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
  return (uint16_t)(sign | r);
}
float h2f(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
  int e = (h >> 10) & 31;
  uint32_t m = h & 0x3FFu;
  uint32_t u;
  if (e == 0) {
    if (!m) u = s;
    else {
      e = 1;
      while (!(m & 0x400u)) { m <<= 1; --e; }
      m &= 0x3FFu;
      u = s | ((uint32_t)(e + 112) << 23) | (m << 13);
    }
  } else if (e == 31) u = s | 0x7F800000u | (m << 13);
  else u = s | ((uint32_t)(e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---- slab packer: [cbt][nslab] slabs of [64 rows][64 k], rows in MFMA order, swz128 chunk swizzle
std::vector<uint16_t> pack_slabs(int cbt, int nslab, int prec, const std::function<float(int, int, int)>& get) {
  std::vector<uint16_t> out((size_t)cbt * nslab * 4096, 0);
  for (int cb = 0; cb < cbt; ++cb)
    for (int s = 0; s < nslab; ++s) {
      uint16_t* slab = out.data() + ((size_t)cb * nslab + s) * 4096;
      for (int rr = 0; rr < 64; ++rr) {
        const int feat = cb * 64 + slab_row_to_feature(rr);
        for (int k = 0; k < 64; ++k) {
          const int byte = rr * 128 + ((((k >> 3) ^ ((rr >> 1) & 7))) << 4) + (k & 7) * 2;
          slab[byte >> 1] = cvt2(get(feat, s, k), prec);
        }
      }
    }
  return out;
}

const Tensor* need(const Pack& p, const std::string& name, std::string& err) {
  auto it = p.find(name);
  if (it == p.end()) { err = "weight pack is missing tensor " + name; return nullptr; }
  return &it->second;
}

bool make_conv(airfe_ctx* c, const Pack& p, const std::string& name, int cin, int cout, ConvW& out, std::string& err) {
  const Tensor* w = need(p, name + ".weight", err);
  const Tensor* b = need(p, name + ".bias", err);
  if (!w || !b) return false;
  if ((int)w->data.size() != cout * cin * 9 || (int)b->data.size() != cout) { err = name + ": unexpected shape"; return false; }
  const int nci = cin / 64;
  const float* wd = w->data.data();
  auto slabs = pack_slabs(cout / 64, 9 * nci, c->prec, [&](int feat, int s, int k) {
    const int tap = s / nci, cc = s % nci, ci = cc * 64 + k;
    return wd[((size_t)feat * cin + ci) * 9 + tap];
  });
  out.w = dupload(c, slabs);
  out.b = dupload(c, b->data);
  out.cin = cin;
  out.cout = cout;
  return out.w && out.b;
}

// Linear y = W x + b with W [N][K] row-major; `src_row(feature)` lets callers permute / select output rows
bool make_linear(airfe_ctx* c, const float* W, const float* bias, int K, int N, LinW& out, float scale,
                 const std::function<int(int)>* src_row, const std::function<int(int)>* src_col) {
  const int Kp = (K + 63) / 64 * 64, cbt = (N + 63) / 64;
  const int cbp = (cbt + 3) & ~3;        // the GEMMs consume feature blocks in pairs / quads (128- / 256-feature tiles): zero pad
  auto slabs = pack_slabs(cbp, Kp / 64, c->pack_prec, [&](int feat, int s, int k) {
    const int kk = s * 64 + k;
    if (feat >= N || kk >= K) return 0.f;
    const int r = src_row ? (*src_row)(feat) : feat;
    const int cc = src_col ? (*src_col)(kk) : kk;
    return W[(size_t)r * K + cc] * scale;
  });
  std::vector<float> bp((size_t)cbp * 64, 0.f);
  for (int f = 0; f < N; ++f) bp[f] = bias[src_row ? (*src_row)(f) : f] * scale;
  out.w = dupload(c, slabs);
  out.b = dupload(c, bp);
  out.K = Kp;
  out.N = N;
  out.cbt = cbt;
  // the same bytes in FRAGMENT order for kernels that load A fragments straight from global memory (kernels_lgblockf.hip): [16-row tile T][slab s][32-wide half h]
  // [lane = g * 16 + l15][16 B] — a wave's load is 1 KB in a row instead of 16 bytes out of each of 16 lines (the slab image is laid out for LDS)
  {
    const int NS = Kp / 64;
    std::vector<uint16_t> fr(slabs.size());
    for (int T = 0; T < cbp * 4; ++T)
      for (int sl = 0; sl < NS; ++sl) {
        const uint16_t* slab = slabs.data() + ((size_t)(T >> 2) * NS + sl) * (SLAB_BYTES / 2);
        for (int h = 0; h < 2; ++h)
          for (int lane = 0; lane < 64; ++lane) {
            const int l15 = lane & 15, g = lane >> 4, rr = (T & 3) * 16 + l15;
            const int byte = rr * 128 + (((h * 4 + g) ^ ((rr >> 1) & 7)) << 4);
            uint16_t* dst = fr.data() + (((size_t)T * NS + sl) * 2 + h) * 512 + (size_t)lane * 8;
            for (int e = 0; e < 8; ++e) dst[e] = slab[(byte >> 1) + e];
          }
      }
    out.wf = dupload(c, fr);
  }
  return out.w && out.b && out.wf;
}

bool make_linear_named(airfe_ctx* c, const Pack& p, const std::string& name, int K, int N, LinW& out, std::string& err,
                       float scale = 1.f) {
  const Tensor* w = need(p, name + ".weight", err);
  const Tensor* b = need(p, name + ".bias", err);
  if (!w || !b) return false;
  if ((int)w->data.size() != N * K || (int)b->data.size() != N) { err = name + ": unexpected shape"; return false; }
  return make_linear(c, w->data.data(), b->data.data(), K, N, out, scale);
}

// airfe_tuning::fold_out_proj.  A block computes  h = W1 cat(x, msg) + b1  with  msg = Wo a + bo  (a = the attention output) and nothing between the two
// linear maps, so  h = W1x x + (W1m Wo) a + (b1 + W1m bo):  the layer is packed as ONE 512 -> 512 linear over cat(x, a) and the 256 x 256 out-projection
// (LightGlue self_attn.out_proj / cross_attn.to_out, light_glue's exported graph; SuperGlue attn.merge) is never run.  Products summed in double in a fixed
// order and rounded once to fp32 before the 2-byte packing; `wo_col` = the column order of Wo the caller's attention output has (SuperGlue: head-major).
bool make_ffn0_folded(airfe_ctx* c, const Tensor& w1, const Tensor& b1, const Tensor& wo, const Tensor& bo, LinW& out, std::string& err,
                      const std::function<int(int)>* wo_col = nullptr) {
  if (w1.data.size() != 512 * 512 || b1.data.size() != 512 || wo.data.size() != 256 * 256 || bo.data.size() != 256) { err = "fold_out_proj: unexpected shape"; return false; }
  std::vector<float> W((size_t)512 * 512), B(512);
  std::vector<double> row(256);
  for (int n = 0; n < 512; ++n) {
    const float* w1n = &w1.data[(size_t)n * 512];
    memcpy(&W[(size_t)n * 512], w1n, 256 * sizeof(float));
    std::fill(row.begin(), row.end(), 0.0);
    double bacc = b1.data[n];
    for (int j = 0; j < 256; ++j) {
      const double m = w1n[256 + j];
      bacc += m * bo.data[j];
      const float* woj = &wo.data[(size_t)j * 256];
      for (int k = 0; k < 256; ++k) row[k] += m * woj[k];
    }
    for (int k = 0; k < 256; ++k) W[(size_t)n * 512 + 256 + k] = (float)row[wo_col ? (*wo_col)(k) : k];
    B[n] = (float)bacc;
  }
  return make_linear(c, W.data(), B.data(), 512, 512, out, 1.f);
}

// OpenCV resize() INTER_LINEAR coefficient table (imgproc/src/resize.cpp) -> [d][4] = s0, s1, a0, a1
std::vector<int> resize_table(int dsize, int ssize) {
  std::vector<int> t((size_t)dsize * 4);
  const double scale = (double)ssize / dsize;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    t[d * 4 + 0] = s;
    t[d * 4 + 1] = std::min(s + 1, ssize - 1);
    t[d * 4 + 2] = (int)lrintf((1.f - f) * 2048.f);
    t[d * 4 + 3] = (int)lrintf(f * 2048.f);
  }
  return t;
}

// ---- fp32 correctness path: weights as fp32, convolutions as [9][Cin][Cout]
bool f32_conv(airfe_ctx* c, const Pack& p, const std::string& name, int cin, int cout, airfe_ctx::F32Conv& out, std::string& err) {
  const Tensor* w = need(p, name + ".weight", err);
  const Tensor* b = need(p, name + ".bias", err);
  if (!w || !b) return false;
  if ((int)w->data.size() != cout * cin * 9 || (int)b->data.size() != cout) { err = name + ": unexpected shape"; return false; }
  std::vector<float> t((size_t)9 * cin * cout);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int tap = 0; tap < 9; ++tap) t[((size_t)tap * cin + ci) * cout + co] = w->data[((size_t)co * cin + ci) * 9 + tap];
  out.w = dupload(c, t); out.b = dupload(c, b->data); out.cin = cin; out.cout = cout;
  return out.w && out.b;
}
bool f32_lin(airfe_ctx* c, const float* W, const float* bias, int K, int N, airfe_ctx::F32Lin& out, const std::function<int(int)>* src_row = nullptr) {
  std::vector<float> w((size_t)N * K), b(N);
  for (int n = 0; n < N; ++n) {
    const int r = src_row ? (*src_row)(n) : n;
    memcpy(&w[(size_t)n * K], W + (size_t)r * K, (size_t)K * 4);
    b[n] = bias[r];
  }
  out.w = dupload(c, w); out.b = dupload(c, b); out.K = K; out.N = N;
  return out.w && out.b;
}
bool f32_lin_named(airfe_ctx* c, const Pack& p, const std::string& name, int K, int N, airfe_ctx::F32Lin& out, std::string& err) {
  const Tensor* w = need(p, name + ".weight", err);
  const Tensor* b = need(p, name + ".bias", err);
  if (!w || !b) return false;
  if ((int)w->data.size() != N * K || (int)b->data.size() != N) { err = name + ": unexpected shape"; return false; }
  return f32_lin(c, w->data.data(), b->data.data(), K, N, out);
}

int load_superpoint_f32(airfe_ctx* c, const Pack& p) {
  std::string err;
  bool ok = f32_conv(c, p, "conv1b", 64, 64, c->f_c1b, err) && f32_conv(c, p, "conv2a", 64, 64, c->f_c2a, err) &&
            f32_conv(c, p, "conv2b", 64, 64, c->f_c2b, err) && f32_conv(c, p, "conv3a", 64, 128, c->f_c3a, err) &&
            f32_conv(c, p, "conv3b", 128, 128, c->f_c3b, err) && f32_conv(c, p, "conv4a", 128, 128, c->f_c4a, err) &&
            f32_conv(c, p, "conv4b", 128, 128, c->f_c4b, err) && f32_conv(c, p, "convPa", 128, 256, c->f_cPa, err) &&
            f32_conv(c, p, "convDa", 128, 256, c->f_cDa, err) && f32_lin_named(c, p, "convPb", 256, 65, c->f_cPb, err) &&
            f32_lin_named(c, p, "convDb", 256, 256, c->f_cDb, err);
  if (ok && p.count("line.conv1.weight"))
    ok = f32_conv(c, p, "line.conv1", 128, 128, c->f_cL1, err) && f32_lin_named(c, p, "line.head", 128, 145, c->f_cLh, err);
  if (!ok) return fail(c, err.empty() ? "device allocation failed while loading fp32 detector weights" : err);
  const int R = AIRFE_INTERNAL_SIZE;
  const size_t FB = c->f_B = std::min(c->Bmax, 2);
  auto sq = [](size_t n) { return n * n; };
  c->f1a = dalloc<float>(c, FB * sq(R + 2) * 64); c->f1b = dalloc<float>(c, FB * sq(R + 2) * 64);
  c->fp1 = dalloc<float>(c, FB * sq(R / 2 + 2) * 64); c->f2a = dalloc<float>(c, FB * sq(R / 2 + 2) * 64); c->f2b = dalloc<float>(c, FB * sq(R / 2 + 2) * 64);
  c->fp2 = dalloc<float>(c, FB * sq(R / 4 + 2) * 64); c->f3a = dalloc<float>(c, FB * sq(R / 4 + 2) * 128); c->f3b = dalloc<float>(c, FB * sq(R / 4 + 2) * 128);
  c->fp3 = dalloc<float>(c, FB * sq(R / 8 + 2) * 128); c->f4a = dalloc<float>(c, FB * sq(R / 8 + 2) * 128); c->f4b = dalloc<float>(c, FB * sq(R / 8 + 2) * 128);
  c->fPa = dalloc<float>(c, FB * sq(R / 8) * 256); c->fDa = dalloc<float>(c, FB * sq(R / 8) * 256);
  c->fL1 = dalloc<float>(c, sq(R / 4) * 128);
  if (!c->f1a || !c->f1b || !c->fp1 || !c->f2a || !c->f2b || !c->fp2 || !c->f3a || !c->f3b || !c->fp3 || !c->f4a || !c->f4b || !c->fPa ||
      !c->fDa || !c->fL1)
    return fail(c, "device allocation failed (fp32 detector arena)");
  return 0;
}

int alloc_matcher_arena_f32(airfe_ctx* c);

int load_lightglue_f32(airfe_ctx* c, const Pack& p, int L) {
  std::string err;
  c->f_lg.resize(L);
  bool ok = true;
  // Wqkv output index = h*192 + d*3 + {q,k,v}  ->  rows [q(h,d) | k(h,d) | v(h,d)]
  std::function<int(int)> qkv_row = [](int f) { const int sel = f >> 8, hd = f & 255; return (hd >> 6) * 192 + (hd & 63) * 3 + sel; };
  for (int i = 0; i < L && ok; ++i) {
    auto& l = c->f_lg[i];
    const std::string s = "transformers." + std::to_string(i) + ".self_attn", x = "transformers." + std::to_string(i) + ".cross_attn";
    const Tensor *wq = need(p, s + ".Wqkv.weight", err), *bq = need(p, s + ".Wqkv.bias", err);
    const Tensor *g1 = need(p, s + ".ffn.1.weight", err), *b1 = need(p, s + ".ffn.1.bias", err);
    const Tensor *g2 = need(p, x + ".ffn.1.weight", err), *b2 = need(p, x + ".ffn.1.bias", err);
    if (!wq || !bq || !g1 || !b1 || !g2 || !b2) { ok = false; break; }
    ok = f32_lin(c, wq->data.data(), bq->data.data(), 256, 768, l.qkv, &qkv_row) && f32_lin_named(c, p, s + ".out_proj", 256, 256, l.out, err) &&
         f32_lin_named(c, p, s + ".ffn.0", 512, 512, l.ffn0, err) && f32_lin_named(c, p, s + ".ffn.3", 512, 256, l.ffn3, err) &&
         f32_lin_named(c, p, x + ".to_qk", 256, 256, l.cqk, err) && f32_lin_named(c, p, x + ".to_v", 256, 256, l.cv, err) &&
         f32_lin_named(c, p, x + ".to_out", 256, 256, l.cout, err) && f32_lin_named(c, p, x + ".ffn.0", 512, 512, l.cffn0, err) &&
         f32_lin_named(c, p, x + ".ffn.3", 512, 256, l.cffn3, err);
    l.ln_g = dupload(c, g1->data); l.ln_b = dupload(c, b1->data); l.cln_g = dupload(c, g2->data); l.cln_b = dupload(c, b2->data);
  }
  ok = ok && f32_lin_named(c, p, "log_assignment." + std::to_string(L - 1) + ".final_proj", 256, 256, c->f_lgfinal, err);
  if (!ok) return fail(c, err.empty() ? "device allocation failed while loading fp32 LightGlue weights" : err);
  return alloc_matcher_arena_f32(c);
}

// the fp32 matcher's activations (LightGlue and SuperGlue share them: one matcher runs at a time)
int alloc_matcher_arena_f32(airfe_ctx* c) {
  if (c->m_qkv) return 0;
  const size_t M = (size_t)(2 * c->Pmax + 2 + 128 / c->Np) * c->Np + 256;
  c->m_qkv = dalloc<float>(c, M * 768); c->m_ctx = dalloc<float>(c, M * 256); c->m_msg = dalloc<float>(c, M * 256);
  c->m_h = dalloc<float>(c, M * 512); c->m_md = dalloc<float>(c, M * 256);
  if (!c->m_qkv || !c->m_ctx || !c->m_msg || !c->m_h || !c->m_md) return fail(c, "device allocation failed (fp32 matcher arena)");
  return 0;
}

// SuperGlue in fp32 (cfg.matcher_precision = 2): the GNN's linears as plain [N][K] fp32 matrices.  MultiHeadedAttention views its channels as
// (dim, heads) — channel = d * 4 + h — while the attention kernel wants a head's 64 features side by side: the q / k / v ROWS and merge's input
// COLUMNS are permuted to head-major here (the same permutation the 2-byte loader applies, load_superglue below), nothing else changes.
int load_superglue_f32(airfe_ctx* c, const Pack& p, int L) {
  std::string err;
  const auto hm = [](int f) { return (f & 63) * 4 + (f >> 6); };
  c->f_sg.resize(L);
  bool ok = true;
  for (int i = 0; i < L && ok; ++i) {
    auto& l = c->f_sg[i];
    const std::string g = "gnn.layers." + std::to_string(i);
    const Tensor *w[3], *b[3];
    for (int j = 0; j < 3; ++j) {
      w[j] = need(p, g + ".attn.proj." + std::to_string(j) + ".weight", err);
      b[j] = need(p, g + ".attn.proj." + std::to_string(j) + ".bias", err);
      if (!w[j] || !b[j] || w[j]->data.size() != 256 * 256 || b[j]->data.size() != 256) { ok = false; break; }
    }
    const Tensor *wm = need(p, g + ".attn.merge.weight", err), *bm = need(p, g + ".attn.merge.bias", err);
    if (!ok || !wm || !bm || wm->data.size() != 256 * 256 || bm->data.size() != 256) { ok = false; break; }
    std::vector<float> wqkv((size_t)768 * 256), bqkv(768), wmp((size_t)256 * 256);
    for (int j = 0; j < 3; ++j)
      for (int f = 0; f < 256; ++f) {
        memcpy(&wqkv[((size_t)j * 256 + f) * 256], &w[j]->data[(size_t)hm(f) * 256], 1024);
        bqkv[j * 256 + f] = b[j]->data[hm(f)];
      }
    for (int n = 0; n < 256; ++n)
      for (int f = 0; f < 256; ++f) wmp[(size_t)n * 256 + f] = wm->data[(size_t)n * 256 + hm(f)];
    ok = f32_lin(c, wqkv.data(), bqkv.data(), 256, 768, l.qkv) && f32_lin(c, wmp.data(), bm->data.data(), 256, 256, l.merge) &&
         f32_lin_named(c, p, g + ".mlp.0", 512, 512, l.mlp0, err) && f32_lin_named(c, p, g + ".mlp.3", 512, 256, l.mlp3, err);
  }
  ok = ok && f32_lin_named(c, p, "final_proj", 256, 256, c->f_sgfinal, err);
  if (!ok) return fail(c, err.empty() ? "SuperGlue (fp32): unexpected shapes or device allocation failed" : err);
  return alloc_matcher_arena_f32(c);
}

int load_superpoint(airfe_ctx* c, const char* path) {
  c->pack_prec = c->prec == 2 ? 1 : c->prec;
  Pack p;
  std::string err;
  if (!load_pack(path, p, err)) return fail(c, err);
  const Tensor* w1 = need(p, "conv1a.weight", err);
  const Tensor* b1 = need(p, "conv1a.bias", err);
  if (!w1 || !b1 || w1->data.size() != 64 * 9) return fail(c, err.empty() ? "conv1a: unexpected shape" : err);
  c->c1a_w = dupload(c, w1->data);
  c->c1a_b = dupload(c, b1->data);
  bool ok = make_conv(c, p, "conv1b", 64, 64, c->c1b, err) && make_conv(c, p, "conv2a", 64, 64, c->c2a, err) &&
            make_conv(c, p, "conv2b", 64, 64, c->c2b, err) && make_conv(c, p, "conv3a", 64, 128, c->c3a, err) &&
            make_conv(c, p, "conv3b", 128, 128, c->c3b, err) && make_conv(c, p, "conv4a", 128, 128, c->c4a, err) &&
            make_conv(c, p, "conv4b", 128, 128, c->c4b, err) && make_conv(c, p, "convPa", 128, 256, c->cPa, err) &&
            make_conv(c, p, "convDa", 128, 256, c->cDa, err) &&
            make_linear_named(c, p, "convPb", 256, 65, c->cPb, err) && make_linear_named(c, p, "convDb", 256, 256, c->cDb, err);
  if (!ok) return fail(c, err.empty() ? "device allocation failed while packing SuperPoint weights" : err);

  const int B = c->Dmax, ch = c->chunk, R = AIRFE_INTERNAL_SIZE;
  c->img32 = dalloc<float>(c, (size_t)ch * (R + 2) * (R + 2));
  c->a1b = dalloc<uint16_t>(c, (size_t)ch * (R / 2 + 2) * (R / 2 + 2) * 64);
  c->a2a = dalloc<uint16_t>(c, (size_t)ch * (R / 2 + 2) * (R / 2 + 2) * 64);
  c->a2b = dalloc<uint16_t>(c, (size_t)B * (R / 4 + 2) * (R / 4 + 2) * 64);
  c->a3a = dalloc<uint16_t>(c, (size_t)B * (R / 4 + 2) * (R / 4 + 2) * 128);
  c->a3b = dalloc<uint16_t>(c, (size_t)B * (R / 8 + 2) * (R / 8 + 2) * 128);
  c->a4a = dalloc<uint16_t>(c, (size_t)B * (R / 8 + 2) * (R / 8 + 2) * 128);
  c->a4b = dalloc<uint16_t>(c, (size_t)B * (R / 8 + 2) * (R / 8 + 2) * 128);
  const size_t cells = (size_t)B * (R / 8) * (R / 8);
  c->aPa = dalloc<uint16_t>(c, cells * 256);
  c->aDa = dalloc<uint16_t>(c, cells * 256);
  c->logits = dalloc<float>(c, cells * 72);
  c->desc = dalloc<float>(c, cells * 256);
  c->desc_idx = dalloc<int>(c, (size_t)B * 1024 * 4 + 256);
  c->heat = dalloc<float>(c, (size_t)B * R * R);
  c->heat_nms = dalloc<float>(c, (size_t)B * R * R);
  c->nms_mask = dalloc<unsigned char>(c, (size_t)2 * B * R * R);
  const bool multipass_nms = c->cfg.nms_radius > 0 && c->cfg.nms_radius != 4;
  c->nms_tmp = dalloc<float>(c, multipass_nms ? (size_t)4 * B * R * R : 1);
  c->cand = dalloc<unsigned long long>(c, (size_t)B * R * R, false);
  c->cand_cnt = dalloc<int>(c, B);
  c->xtab = dalloc<int>(c, (size_t)R * 4);
  c->ytab = dalloc<int>(c, (size_t)R * 4);
  std::vector<float> lut(256);
  for (int i = 0; i < 256; ++i) lut[i] = (float)((double)i / 255.0);
  c->lut = dupload(c, lut);
  if (!c->img32 || !c->a1b || !c->a2a || !c->a2b || !c->a3a || !c->a3b || !c->a4a || !c->a4b || !c->aPa ||
      !c->aDa || !c->logits || !c->desc || !c->heat || !c->heat_nms || !c->nms_tmp || !c->xtab || !c->ytab || !c->lut ||
      !c->cand || !c->cand_cnt)
    return fail(c, "device allocation failed (detector arena)");
  c->has_sp = true;
  if (p.count("line.conv1.weight")) {       // a PLNet stage-0 pack: the line branch rides along
    const Tensor *hw = need(p, "line.head.weight", err), *hb = need(p, "line.head.bias", err);
    if (!hw || !hb || hw->data.size() != 145 * 128 || hb->data.size() != 145) return fail(c, err.empty() ? "line.head: unexpected shape" : err);
    std::function<int(int)> dec_row = [](int f) { return 128 + f; };
    if (!make_conv(c, p, "line.conv1", 128, 128, c->cL1, err) || !make_linear(c, hw->data.data(), hb->data.data(), 128, 145, c->cLh) ||
        !make_linear(c, hw->data.data(), hb->data.data(), 128, 128, c->cLh_loi) ||
        !make_linear(c, hw->data.data(), hb->data.data(), 128, 17, c->cLh_dec, 1.f, &dec_row))
      return fail(c, err.empty() ? "device allocation failed while packing the line branch" : err);
    const size_t npx = (size_t)c->Lmax * 128 * 128;                 // one slot per image of the largest detector batch
    c->l_feat = dalloc<uint16_t>(c, npx * 128);
    c->l_head = dalloc<float>(c, (size_t)128 * 128 * 160);          // the fused head: one image (fp32 mode, inspection hook)
    c->l_dec = dalloc<float>(c, npx * 32);
    c->l_ridx = dalloc<int>(c, (size_t)c->Lmax * 1200 + 256);
    c->l_lrows = dalloc<float>(c, ((size_t)c->Lmax * 1200 + 256) * 128);
    c->l_jloc = dalloc<float>(c, npx);
    c->l_joff = dalloc<float>(c, 2 * npx);
    c->l_ta8 = dalloc<float>(c, 8 * npx);
    c->l_sel = dalloc<float>(c, (size_t)c->Lmax * 320 * AIRFE_FEAT_DIM);
    c->l_nsel = dalloc<int>(c, c->Lmax);
    c->l_cand = dalloc<unsigned long long>(c, (size_t)c->Lmax * 128 * 128, false);
    c->l_cand_cnt = dalloc<int>(c, c->Lmax);
    if (!c->l_feat || !c->l_ta8 || !c->l_head || !c->l_dec || !c->l_ridx || !c->l_lrows || !c->l_jloc || !c->l_joff || !c->l_sel || !c->l_nsel || !c->l_cand || !c->l_cand_cnt)
      return fail(c, "device allocation failed (line branch arena)");
    c->has_s0 = true;
  }
  if (c->prec == 2 && load_superpoint_f32(c, p)) return 1;
  return 0;
}

int alloc_matcher_arena(airfe_ctx* c);

int load_lightglue(airfe_ctx* c, const char* path) {
  c->pack_prec = c->mprec == 2 ? 1 : c->mprec;
  Pack p;
  std::string err;
  if (!load_pack(path, p, err)) return fail(c, err);
  int L = 0;
  while (p.count("transformers." + std::to_string(L) + ".self_attn.Wqkv.weight")) ++L;
  if (L == 0) return fail(c, "LightGlue pack has no transformer layers");
  const Tensor* wr = need(p, "posenc.Wr.weight", err);
  if (!wr || wr->data.size() != 64) return fail(c, "posenc.Wr.weight missing or wrong shape");
  c->lg_wr = dupload(c, wr->data);
  c->lg.resize(L);
  bool ok = true;
  for (int i = 0; i < L && ok; ++i) {
    LgLayer& l = c->lg[i];
    const std::string s = "transformers." + std::to_string(i) + ".self_attn";
    const std::string x = "transformers." + std::to_string(i) + ".cross_attn";
    const Tensor* wqkv = need(p, s + ".Wqkv.weight", err);
    const Tensor* bqkv = need(p, s + ".Wqkv.bias", err);
    if (!wqkv || !bqkv || wqkv->data.size() != 768 * 256) { ok = false; break; }
    // Wqkv output index = h*192 + d*3 + {q,k,v}  (qkv.unflatten(-1,(H,-1,3)))  ->  [q(h,d) | k(h,d)] and v(h,d)
    std::function<int(int)> qk_row = [](int f) { const int sel = f >> 8, hd = f & 255; return (hd >> 6) * 192 + (hd & 63) * 3 + sel; };
    std::function<int(int)> v_row = [](int f) { return (f >> 6) * 192 + (f & 63) * 3 + 2; };
    ok = ok && make_linear(c, wqkv->data.data(), bqkv->data.data(), 256, 512, l.qk, ATT_QK_FOLD, &qk_row);
    ok = ok && make_linear(c, wqkv->data.data(), bqkv->data.data(), 256, 256, l.v, 1.f, &v_row);
    ok = ok && make_linear_named(c, p, s + ".out_proj", 256, 256, l.out, err);
    if (c->fold_out) {         // (l.out / l.cout stay packed: nothing runs them in this context, airfe_debug hooks may)
      const Tensor *w1 = need(p, s + ".ffn.0.weight", err), *b1f = need(p, s + ".ffn.0.bias", err), *wo = need(p, s + ".out_proj.weight", err), *bo = need(p, s + ".out_proj.bias", err);
      ok = ok && w1 && b1f && wo && bo && make_ffn0_folded(c, *w1, *b1f, *wo, *bo, l.ffn0, err);
    } else
    ok = ok && make_linear_named(c, p, s + ".ffn.0", 512, 512, l.ffn0, err);
    ok = ok && make_linear_named(c, p, s + ".ffn.3", 512, 256, l.ffn3, err);
    ok = ok && make_linear_named(c, p, x + ".to_qk", 256, 256, l.cqk, err, ATT_QK_FOLD);
    ok = ok && make_linear_named(c, p, x + ".to_v", 256, 256, l.cv, err);
    ok = ok && make_linear_named(c, p, x + ".to_out", 256, 256, l.cout, err);
    if (c->fold_out) {
      const Tensor *w1 = need(p, x + ".ffn.0.weight", err), *b1f = need(p, x + ".ffn.0.bias", err), *wo = need(p, x + ".to_out.weight", err), *bo = need(p, x + ".to_out.bias", err);
      ok = ok && w1 && b1f && wo && bo && make_ffn0_folded(c, *w1, *b1f, *wo, *bo, l.cffn0, err);
    } else
    ok = ok && make_linear_named(c, p, x + ".ffn.0", 512, 512, l.cffn0, err);
    ok = ok && make_linear_named(c, p, x + ".ffn.3", 512, 256, l.cffn3, err);
    const Tensor *g1 = need(p, s + ".ffn.1.weight", err), *b1 = need(p, s + ".ffn.1.bias", err);
    const Tensor *g2 = need(p, x + ".ffn.1.weight", err), *b2 = need(p, x + ".ffn.1.bias", err);
    if (!g1 || !b1 || !g2 || !b2) { ok = false; break; }
    l.ln_g = dupload(c, g1->data); l.ln_b = dupload(c, b1->data);
    l.cln_g = dupload(c, g2->data); l.cln_b = dupload(c, b2->data);
  }
  const std::string a = "log_assignment." + std::to_string(L - 1);
  ok = ok && make_linear_named(c, p, a + ".final_proj", 256, 256, c->lg_final, err, 0.25f /* d^-1/4, d = 256 */);
  const Tensor *mw = need(p, a + ".matchability.weight", err), *mb = need(p, a + ".matchability.bias", err);
  if (!ok || !mw || !mb) return fail(c, err.empty() ? "LightGlue weight packing failed" : err);
  c->lg_mw = dupload(c, mw->data);
  c->lg_mb = mb->data[0];
  if (alloc_matcher_arena(c)) return 1;
  if (c->mprec == 2 && load_lightglue_f32(c, p, L)) return 1;
  c->has_lg = true;
  return 0;
}

int alloc_matcher_arena(airfe_ctx* c) {
  if (c->has_arena) return 0;
  const int S = 2 * c->Pmax, Np = c->Np;
  // Token rows: S sequences of Np, PLUS slack.  The GEMMs run over M rounded up to 128 rows and the fused block in passes of 112 on
  // top of that: the up-to-238 surplus rows are garbage tokens of "sequences" S, S+1, .. whose head-major outputs (incl. the
  // projections folded into the block) land one or more whole sequences past the real data,
  // and attention's last key tile reads up to 63 rows past a sequence.  All of it stays inside this zero-initialised slack.
  const size_t M = (size_t)(S + 2 + 128 / Np) * Np + 256;
  c->arena_rows = M;
  c->x32 = dalloc<float>(c, M * 256);
  c->xb = dalloc<uint16_t>(c, M * 256);
  c->qb = dalloc<uint16_t>(c, M * 256);
  c->kb = dalloc<uint16_t>(c, M * 256);
  c->vtb = dalloc<uint16_t>(c, M * 256);
  c->ob = dalloc<uint16_t>(c, M * 256);
  c->msg = dalloc<uint16_t>(c, M * 256);
  c->hb = dalloc<uint16_t>(c, M * 512);
  c->mdb = dalloc<uint16_t>(c, M * 256);
  c->rot_cos = dalloc<float>(c, M * 32);
  c->rot_sin = dalloc<float>(c, M * 32);
  c->zbuf = dalloc<float>(c, M);
  c->lens = dalloc<int>(c, S);
  c->simbuf = dalloc<float>(c, (size_t)c->Pmax * Np * Np);
  c->st_scores_full = dalloc<float>(c, (size_t)Np * Np);
  c->rowlse = dalloc<float>(c, (size_t)c->Pmax * Np);
  c->collse = dalloc<float>(c, (size_t)c->Pmax * Np);
  c->rowval = dalloc<float>(c, (size_t)c->Pmax * Np);
  c->rowarg = dalloc<int>(c, (size_t)c->Pmax * Np);
  c->colarg = dalloc<int>(c, (size_t)c->Pmax * Np);
  c->lg_part = dalloc<float>(c, lg_assign_part_floats(c->Pmax, Np));          // per-tile (max, sum exp) / (max, arg) partials of the fused assignment
  c->lg_argpart = dalloc<float>(c, lg_assign_part_floats(c->Pmax, Np));
  if (!c->x32 || !c->xb || !c->qb || !c->kb || !c->vtb || !c->ob || !c->msg || !c->hb || !c->mdb || !c->rot_cos ||
      !c->rot_sin || !c->zbuf || !c->lens || !c->simbuf || !c->rowlse || !c->collse || !c->rowval || !c->rowarg ||
      !c->colarg || !c->st_scores_full || !c->lg_part || !c->lg_argpart)
    return fail(c, "device allocation failed (matcher arena)");
  c->has_arena = true;
  return 0;
}

// y = W x + b stored transposed [K][N] fp32 for the thread-per-neuron VALU kernels
float* upload_transposed(airfe_ctx* c, const Tensor& w, int N, int K, int pad_rows = 0) {
  std::vector<float> t((size_t)N * (K + pad_rows), 0.f);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = w.data[(size_t)n * K + k];
  return dupload(c, t);
}

int load_superglue(airfe_ctx* c, const char* path) {
  c->pack_prec = c->mprec == 2 ? 1 : c->mprec;
  Pack p;
  std::string err;
  if (!load_pack(path, p, err)) return fail(c, err);
  int L = 0;
  while (p.count("gnn.layers." + std::to_string(L) + ".attn.merge.weight")) ++L;
  if (L == 0) return fail(c, "SuperGlue pack has no GNN layers");
  const int enc[6] = {3, 32, 64, 128, 256, 256};
  for (int i = 0; i < 5; ++i) {
    const Tensor* w = need(p, "kenc.encoder." + std::to_string(i) + ".weight", err);
    const Tensor* b = need(p, "kenc.encoder." + std::to_string(i) + ".bias", err);
    if (!w || !b || (int)w->data.size() != enc[i] * enc[i + 1]) return fail(c, err.empty() ? "kenc: unexpected shape" : err);
    c->sg_kenc[2 * i] = upload_transposed(c, *w, enc[i + 1], enc[i]);
    c->sg_kenc[2 * i + 1] = dupload(c, b->data);
  }
  // the two large layers also as packed MFMA operands (large batches: launch_sg_prepare with h128, then two GEMMs)
  if (!make_linear_named(c, p, "kenc.encoder.3", 128, 256, c->sg_k3, err) || !make_linear_named(c, p, "kenc.encoder.4", 256, 256, c->sg_k4, err))
    return fail(c, err.empty() ? "kenc: packing failed" : err);
  // MultiHeadedAttention views channels as (dim, heads): channel = d*4 + h  ->  our head-major h*64 + d
  std::function<int(int)> hm = [](int f) { return (f & 63) * 4 + (f >> 6); };
  c->sg.resize(L);
  bool ok = true;
  for (int i = 0; i < L && ok; ++i) {
    SgLayer& l = c->sg[i];
    const std::string g = "gnn.layers." + std::to_string(i);
    const Tensor *wq = need(p, g + ".attn.proj.0.weight", err), *bq = need(p, g + ".attn.proj.0.bias", err);
    const Tensor *wk = need(p, g + ".attn.proj.1.weight", err), *bk = need(p, g + ".attn.proj.1.bias", err);
    const Tensor *wv = need(p, g + ".attn.proj.2.weight", err), *bv = need(p, g + ".attn.proj.2.bias", err);
    const Tensor *wm = need(p, g + ".attn.merge.weight", err), *bm = need(p, g + ".attn.merge.bias", err);
    if (!wq || !bq || !wk || !bk || !wv || !bv || !wm || !bm) { ok = false; break; }
    std::vector<float> wqk(512 * 256), bqk(512);
    for (int f = 0; f < 256; ++f) {
      memcpy(&wqk[(size_t)f * 256], &wq->data[(size_t)hm(f) * 256], 1024);
      memcpy(&wqk[(size_t)(256 + f) * 256], &wk->data[(size_t)hm(f) * 256], 1024);
      bqk[f] = bq->data[hm(f)];
      bqk[256 + f] = bk->data[hm(f)];
    }
    ok = ok && make_linear(c, wqk.data(), bqk.data(), 256, 512, l.qk, ATT_QK_FOLD);
    ok = ok && make_linear(c, wv->data.data(), bv->data.data(), 256, 256, l.v, 1.f, &hm);
    ok = ok && make_linear(c, wm->data.data(), bm->data.data(), 256, 256, l.merge, 1.f, nullptr, &hm);
    if (c->fold_out) {
      const Tensor *w1 = need(p, g + ".mlp.0.weight", err), *b1f = need(p, g + ".mlp.0.bias", err);
      ok = ok && w1 && b1f && make_ffn0_folded(c, *w1, *b1f, *wm, *bm, l.mlp0, err, &hm);
    } else
    ok = ok && make_linear_named(c, p, g + ".mlp.0", 512, 512, l.mlp0, err);
    ok = ok && make_linear_named(c, p, g + ".mlp.3", 512, 256, l.mlp3, err);
  }
  ok = ok && make_linear_named(c, p, "final_proj", 256, 256, c->sg_final, err, 0.25f /* scores / 256^.5 split over both sides */);
  const Tensor* bs = need(p, "bin_score", err);
  if (!ok || !bs) return fail(c, err.empty() ? "SuperGlue weight packing failed" : err);
  c->sg_alpha = bs->data[0];
  if (alloc_matcher_arena(c)) return 1;
  const int P = c->Pmax;
  c->Lz = c->Np + 64;
  const size_t pl = (size_t)P * c->Lz;
  c->sg_u = dalloc<float>(c, pl); c->sg_v = dalloc<float>(c, pl); c->sg_Z = dalloc<float>(c, pl * c->Lz);
  c->sg_max0 = dalloc<float>(c, pl); c->sg_ms0 = dalloc<float>(c, pl); c->sg_ms1 = dalloc<float>(c, pl);
  c->sg_idx0 = dalloc<int>(c, pl); c->sg_idx1 = dalloc<int>(c, pl);
  c->sg_cnt = dalloc<unsigned>(c, (size_t)P * 16 + 16);      // + the Sinkhorn kernel's fail word (sg_cnt + P * 16)
  c->sg_xch = dalloc<float>(c, pl * 64);
  c->sg_out0 = dalloc<int32_t>(c, pl); c->sg_out1 = dalloc<int32_t>(c, pl);
  if (!c->sg_u || !c->sg_v || !c->sg_Z || !c->sg_max0 || !c->sg_ms0 || !c->sg_ms1 || !c->sg_idx0 || !c->sg_idx1 ||
      !c->sg_out0 || !c->sg_out1 || !c->sg_cnt || !c->sg_xch)
    return fail(c, "device allocation failed (SuperGlue arena)");
  if (c->mprec == 2 && load_superglue_f32(c, p, L)) return 1;
  c->has_sg = true;
  return 0;
}

int load_plnet_s1(airfe_ctx* c, const char* path) {
  Pack p;
  std::string err;
  if (!load_pack(path, p, err)) return fail(c, err);
  struct L { const char* name; int n, k; } ls[4] = {{"fc2.0", 128, 496}, {"fc2.2", 128, 128}, {"fc2.4", 128, 128}, {"fc2_res.0", 128, 240}};
  for (int i = 0; i < 4; ++i) {
    const Tensor* w = need(p, std::string(ls[i].name) + ".weight", err);
    const Tensor* b = need(p, std::string(ls[i].name) + ".bias", err);
    if (!w || !b || (int)w->data.size() != ls[i].n * ls[i].k) return fail(c, err.empty() ? "plnet_s1: unexpected shape" : err);
    c->s1_w[2 * i] = upload_transposed(c, *w, ls[i].n, ls[i].k, S1_WPAD);
    c->s1_w[2 * i + 1] = dupload(c, b->data);
  }
  // the same four matrices as fp16 (hi, lo) planes for the 2-byte matrix pipe (cfg.line_precision = 3, kernels_ext.hip plnet_s1h_kernel): [2][128][K], fc2.0 only
  // in its 240 thin / aux columns (its 256 LOI columns are applied per junction in fp32: s1_junc_proj_kernel); lo = fp16((w - hi) * 2^11)
  {
    struct S { const char* name; int k0, k; } ss[6] = {{"fc2.0", 256, 240}, {"fc2_res.0", 0, 240}, {"fc2.2", 0, 128}, {"fc2.4", 0, 128},
                                                     {"fc2.0", 0, 128}, {"fc2.0", 128, 128}};      // [4], [5]: the LOI columns of the two end points (s1h_junc_proj_kernel)
    for (int i = 0; i < 6; ++i) {
      const Tensor* w = need(p, std::string(ss[i].name) + ".weight", err);
      const int ld = (int)w->data.size() / 128, K = ss[i].k;
      std::vector<uint16_t> t((size_t)2 * 128 * K);      // fragment order (kernels_ext.hip, S1Wh): [feature block n / 32][step k / 16][hi | lo][lane = 32 (k % 16 / 8) + n % 32][k % 8]
      for (int n = 0; n < 128; ++n)
        for (int k = 0; k < K; ++k) {
          const float v = w->data[(size_t)n * ld + ss[i].k0 + k];
          const uint16_t hi = f2h(v);
          const size_t at = ((size_t)((n >> 5) * (K / 16) + (k >> 4)) * 2 * 64 + (size_t)(32 * ((k >> 3) & 1) + (n & 31))) * 8 + (k & 7);
          t[at] = hi;
          t[at + 512] = f2h((v - h2f(hi)) * 2048.0f);
        }
      c->s1_wsplit[i] = dupload(c, t);
      if (!c->s1_wsplit[i]) return fail(c, "device allocation failed (plnet_s1 split weights)");
    }
  }
  const Tensor *wh = need(p, "fc2_head.weight", err), *bh = need(p, "fc2_head.bias", err), *tt = need(p, "sample_t", err);
  if (!wh || !bh || !tt || wh->data.size() != 256 || tt->data.size() != 30) return fail(c, err.empty() ? "plnet_s1 head: unexpected shape" : err);
  c->s1_w[8] = dupload(c, wh->data);
  c->s1_w[9] = dupload(c, bh->data);
  c->s1_w[10] = dupload(c, tt->data);
  const size_t L = (size_t)c->Lmax;
  c->wf_table = dalloc<int>(c, L * 300 * 300, false);
  if (c->wf_table) HIPCHK(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->wf_table), 0x7FFFFFFF, L * 300 * 300, c->stream));
  c->wf_keep = dalloc<int>(c, L * KEEP_CAP);
  c->wf_pairs = dalloc<int>(c, L * LINE_CAP * 2);
  c->wf_rep = dalloc<int>(c, L * LINE_CAP);
  c->wf_prop = dalloc<float>(c, L * LINE_CAP * 4);
  c->wf_counts = dalloc<int>(c, L * LINE_CNT_LD);      // per image: M1, M2, then the per-workgroup counts of wf_count_kernel
  c->s1_la = dalloc<float>(c, L * LINE_CAP * 4);
  c->s1_sc = dalloc<float>(c, L * LINE_CAP);
  c->s1_jfeat = dalloc<float>(c, L * 300 * 256);
  c->s0_stage = dalloc<float>(c, L * SG_STRIDE);
  c->s0_loi = dalloc<float>(c, (size_t)128 * 128 * 128);
  c->jmap = dalloc<unsigned char>(c, L * AIRFE_INTERNAL_SIZE * AIRFE_INTERNAL_SIZE);
  c->d_lines = dalloc<double>(c, (size_t)LINE_CAP * 4);
  c->d_nlines = dalloc<int>(c, 2 * L);
  c->d_njunc = dalloc<int>(c, L * (2 + 64));
  c->junc_feat = dalloc<float>(c, (size_t)JUNC_CAP * AIRFE_FEAT_DIM);
  for (int i = 0; i < 11; ++i) if (!c->s1_w[i]) return fail(c, "device allocation failed (plnet_s1 weights)");
  if (!c->wf_table || !c->wf_keep || !c->wf_pairs || !c->wf_rep || !c->wf_prop || !c->wf_counts || !c->s1_la || !c->s1_sc || !c->s1_jfeat || !c->s0_stage || !c->s0_loi ||
      !c->jmap || !c->d_lines || !c->d_nlines || !c->d_njunc || !c->junc_feat)
    return fail(c, "device allocation failed (line path arena)");
  c->has_s1 = true;
  return 0;
}

}  // namespace airfe_host
