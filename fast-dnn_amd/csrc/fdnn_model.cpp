// fdnn_model.cpp -- .bin loader, quantizer and weight-blob packer (host only).
#include "fdnn_model.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../../include/fdnn.h"

namespace fdnn {

namespace {

inline uint32_t load_be32(const uint8_t *p) {
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

struct Cursor {
  const std::vector<uint8_t> &buf;
  size_t off = 0;
  bool ok = true;
  bool need(size_t n) {
    if (off + n > buf.size()) ok = false;
    return ok;
  }
  int32_t i32() {
    if (!need(4)) return 0;
    uint32_t v = load_be32(&buf[off]);
    off += 4;
    return int32_t(v);
  }
  // n big-endian floats into dst
  void f32(float *dst, size_t n) {
    if (!need(4 * n)) return;
    const uint8_t *p = &buf[off];
    for (size_t i = 0; i < n; ++i, p += 4) {
      uint32_t v = load_be32(p);
      std::memcpy(&dst[i], &v, 4);
    }
    off += 4 * n;
  }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// static_cast<char>(float) as the reference's x86 build executes it
// (cvttss2si to int32, "integer indefinite" 0x80000000 when NaN / out of
// range, low byte kept) -- dnn.cc:499.
inline int8_t float_to_char_x86(float v) {
  int32_t i;
  if (!(v > -2147483904.0f && v < 2147483648.0f))
    i = std::numeric_limits<int32_t>::min();
  else
    i = static_cast<int32_t>(v);
  return static_cast<int8_t>(static_cast<uint8_t>(i & 0xff));
}

// RN(1/c) for a positive finite float c.
float rounded_reciprocal(float c) {
  if (!(c > 0.0f) || std::isinf(c)) return 0.0f;
  float y = static_cast<float>(1.0 / static_cast<double>(c));
  float best = y;
  double best_err = std::fabs(std::fma(static_cast<double>(c), static_cast<double>(y), -1.0));
  const float cand[2] = {std::nextafterf(y, 0.0f), std::nextafterf(y, INFINITY)};
  for (float t : cand) {
    double e = std::fabs(std::fma(static_cast<double>(c), static_cast<double>(t), -1.0));
    if (e < best_err) {
      best_err = e;
      best = t;
    }
  }
  return best;
}

}  // namespace

void build_sigmoid_lut(uint8_t *out) {
  // dnn.cc:100-115: float k = i/100.0f; sigmoid = 1.0f/(1+exp(-k)) in float
  // (exp resolves to the float overload); round() is std::round(float).
  for (int i = -kLutHalf; i < kLutHalf; ++i) {
    float k = static_cast<float>(i) / 100.0f;
    float sig = 1.0f / (1.0f + std::exp(-k));
    out[i + kLutHalf] = static_cast<uint8_t>(std::round(sig * 255.0f));
  }
}

void quantize_layer(const float *w, int rows, int cols, float cutoff, int8_t *out, float *multiplier) {
  const float hi = cutoff, lo = -cutoff;
  // abs-max over the weights clamped to [-cutoff, cutoff] (dnn.cc:148-160, :468-476)
  float amax = -FLT_MAX;
  const size_t total = size_t(rows) * size_t(cols);
  for (size_t i = 0; i < total; ++i) {
    float f = w[i];
    if (f < lo) f = lo;
    if (f > hi) f = hi;
    float a = std::fabs(f);
    if (a > amax) amax = a;
  }
  const float mult = std::round(127.0f / amax);  // dnn.cc:98, :479
  // only the lower clamp is live when quantizing (dnn.cc:492-498)
  for (size_t i = 0; i < total; ++i) {
    float f = w[i];
    if (f < lo) f = lo;
    out[i] = float_to_char_x86(std::round(f * mult));
  }
  *multiplier = mult;
}

namespace {

struct RawLayer {
  int in_dim = 0, in_pad = 0, out_dim = 0;
  std::vector<float> w;  // out_dim x in_pad
  std::vector<float> bias;
};

int pack(const std::vector<RawLayer> &layers, const std::vector<float> &shift, const std::vector<float> &scale,
         int in_dim_file, float cutoff, HostModel *hm, std::string *msg) {
  const int n_affine = int(layers.size());
  const int H = layers[0].out_dim;
  const int D = layers[0].in_pad;
  const int O = layers.back().out_dim;
  BlobHeader h{};
  h.magic = kBlobMagic;
  h.version = kBlobVersion;
  h.n_affine = n_affine;
  h.in_dim_file = in_dim_file;
  h.in_dim = D;
  h.hidden = H;
  h.out_dim = O;
  h.n_q = n_affine - 1;
  h.cutoff = cutoff;

  // quantize first: section sizes depend on the fix-up lists
  struct QTmp {
    std::vector<int8_t> wq;
    std::vector<int32_t> wsum, fix_grp;
    std::vector<FixEntry> ent;
    float mult = 0;
    int cols_pad = 0;
  };
  std::vector<QTmp> qt(size_t(h.n_q));
  for (int qi = 0; qi < h.n_q; ++qi) {
    const RawLayer &L = layers[size_t(qi) + 1];
    QTmp &t = qt[size_t(qi)];
    const int rows = L.out_dim, cols = L.in_pad;
    const int rows_pad = int(align_up(size_t(rows), kRowPad));
    // row stride: k extent padded to the GEMM k-step, plus the skew that keeps the rows
    // of a tile off the same L2 channels (see qgemm_kernel)
    const int cols_pad = int(align_up(size_t(cols), kColPad)) + kRowSkew;
    t.cols_pad = cols_pad;
    t.wq.assign(size_t(rows_pad) * cols_pad, 0);
    {
      std::vector<int8_t> dense(size_t(rows) * cols);
      quantize_layer(L.w.data(), rows, cols, cutoff, dense.data(), &t.mult);
      for (int r = 0; r < rows; ++r) std::memcpy(&t.wq[size_t(r) * cols_pad], &dense[size_t(r) * cols], size_t(cols));
    }
    t.wsum.assign(size_t(rows_pad), 0);
    t.fix_grp.assign(size_t(rows_pad) / 64 + 1, 0);
    for (int r = 0; r < rows; ++r) {
      const int8_t *wr = &t.wq[size_t(r) * cols_pad];
      int32_t s = 0;
      const size_t first = t.ent.size();
      for (int k = 0; k < cols; k += 2) {
        const int w0 = wr[k], w1 = wr[k + 1];
        s += w0 + w1;
        // the pair sum a0*w0 + a1*w1 (a in 0..255) can leave int16 for SOME activation
        // iff 255*(w0^+ + w1^+) > 32767 or 255*(w0^- + w1^-) < -32768
        const int pos = (w0 > 0 ? w0 : 0) + (w1 > 0 ? w1 : 0);
        const int neg = (w0 < 0 ? w0 : 0) + (w1 < 0 ? w1 : 0);
        if (255 * pos > 32767 || 255 * neg < -32768) t.ent.push_back(FixEntry{uint16_t(k), int8_t(w0), int8_t(w1), r});
      }
      t.wsum[size_t(r)] = 128 * s;
      (void)first;
      if ((r & 63) == 63) t.fix_grp[size_t(r) / 64 + 1] = int32_t(t.ent.size());
    }
    // close the ranges of a partial last group and of the padding groups
    for (size_t gi = size_t(rows + 63) / 64; gi < t.fix_grp.size(); ++gi) t.fix_grp[gi] = int32_t(t.ent.size());
    // inside a 64-node group the GEMM consumes the entries in k order (as its k-loop
    // brings the columns through LDS)
    for (size_t gi = 0; gi + 1 < t.fix_grp.size(); ++gi)
      std::stable_sort(t.ent.begin() + t.fix_grp[gi], t.ent.begin() + t.fix_grp[gi + 1],
                       [](const FixEntry &a, const FixEntry &b) { return a.k < b.k; });
  }

  size_t off = align_up(sizeof(BlobHeader), 256);
  auto place = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return uint64_t(o);
  };
  h.off_w0 = place(sizeof(float) * size_t(H) * D);
  h.off_b0 = place(sizeof(float) * size_t(H));
  h.off_shift = place(sizeof(float) * size_t(D));
  h.off_scale = place(sizeof(float) * size_t(D));
  h.off_lut = place(size_t(kLutExt) + 15);
  h.off_lut2 = place(size_t(kLut2Size) + 15);
  for (int qi = 0; qi < h.n_q; ++qi) {
    const RawLayer &L = layers[size_t(qi) + 1];
    QTmp &t = qt[size_t(qi)];
    QLayerDesc &d = h.q[qi];
    d.rows = L.out_dim;
    d.cols = L.in_pad;
    d.cols_pad = t.cols_pad;
    d.rows_pad = int32_t(t.wsum.size());
    d.n_fix = int32_t(t.ent.size());
    d.mult = t.mult;
    d.coef = t.mult * 255.0f;
    d.rcp_coef = rounded_reciprocal(d.coef);
    d.fastdiv_ok = 0;
    d.off_w = place(t.wq.size());
    d.off_bias = place(sizeof(float) * size_t(d.rows_pad));
    d.off_wsum = place(sizeof(int32_t) * size_t(d.rows_pad));
    d.off_fix_grp = place(sizeof(int32_t) * t.fix_grp.size());
    d.off_fix_ent = place(sizeof(FixEntry) * (t.ent.size() + 1));
  }
  h.total_bytes = off;

  hm->blob.assign(off, 0);
  uint8_t *b = hm->blob.data();
  std::memcpy(b + h.off_w0, layers[0].w.data(), sizeof(float) * size_t(H) * D);
  std::memcpy(b + h.off_b0, layers[0].bias.data(), sizeof(float) * size_t(H));
  std::memcpy(b + h.off_shift, shift.data(), sizeof(float) * size_t(D));
  std::memcpy(b + h.off_scale, scale.data(), sizeof(float) * size_t(D));
  {
    uint8_t lut[kLutSize];
    build_sigmoid_lut(lut);
    uint8_t *ext = b + h.off_lut;
    // device table, index clamp(k,-640,640)+640: k <= -640 -> 0, k >= 640 -> 255
    // (dnn.h:38-41); stored XOR 0x80 because activations travel as s8 = u8 - 128.
    ext[0] = 0 ^ 0x80;
    for (int i = 1; i < kLutSize; ++i) ext[i] = lut[i] ^ 0x80;
    ext[kLutSize] = 255 ^ 0x80;
    // Half-step table for the int8-layer epilogue.  k = (int)round(t) (half away from
    // zero) equals sign(u) * ((|u| + 1) >> 1) with u = trunc(2t), and 2t = RN(lin*200)
    // exactly, so QuantizedSigmoid::get(lin) = table2[clamp(u, -kLut2Half, kLut2Half)]:
    // one multiply, one convert, one clamp, one byte gather.
    uint8_t *t2 = b + h.off_lut2;
    for (int u = -kLut2Half; u <= kLut2Half; ++u) {
      const int mag = ((u < 0 ? -u : u) + 1) >> 1;
      const int k = u < 0 ? -mag : mag;
      const uint8_t v = k <= -kLutHalf ? 0 : (k >= kLutHalf ? 255 : lut[k + kLutHalf]);  // dnn.h:38-42
      t2[u + kLut2Half] = v ^ 0x80;
    }
  }
  // The half-step table path converts lin*200 to int32: valid when no |lin| can reach
  // 2^31/200 (the x86 build would turn such values into INT_MIN -> 0, dnn.h:37) and no
  // bias is NaN/inf.  |acc| <= cols*255*128, so bound |lin| per layer.
  for (int qi = 0; qi < h.n_q; ++qi) {
    const RawLayer &L = layers[size_t(qi) + 1];
    QLayerDesc &d = h.q[qi];
    float bmax = 0.0f;
    bool finite = std::isfinite(d.coef) && d.coef > 0.0f;
    for (float bv : L.bias) {
      if (!std::isfinite(bv)) finite = false;
      bmax = std::max(bmax, std::fabs(bv));
    }
    const double bound = finite ? double(d.cols) * 255.0 * 128.0 / double(d.coef) + double(bmax) : 1e300;
    d.lin_bounded = bound * 200.0 < 2.0e9 ? 1 : 0;
  }
  for (int qi = 0; qi < h.n_q; ++qi) {
    const RawLayer &L = layers[size_t(qi) + 1];
    const QTmp &t = qt[size_t(qi)];
    const QLayerDesc &d = h.q[qi];
    std::memcpy(b + d.off_w, t.wq.data(), t.wq.size());
    std::memcpy(b + d.off_bias, L.bias.data(), sizeof(float) * size_t(d.rows));
    std::memcpy(b + d.off_wsum, t.wsum.data(), sizeof(int32_t) * t.wsum.size());
    std::memcpy(b + d.off_fix_grp, t.fix_grp.data(), sizeof(int32_t) * t.fix_grp.size());
    if (!t.ent.empty()) std::memcpy(b + d.off_fix_ent, t.ent.data(), sizeof(FixEntry) * t.ent.size());
  }
  std::memcpy(b, &h, sizeof(h));
  hm->hdr = h;
  (void)msg;
  return FDNN_OK;
}

}  // namespace

int load_host_model(const std::string &path, float cutoff, HostModel *out, std::string *msg) {
  if (!(cutoff > 0.0f)) {  // QuantizedDnn.java:55-57
    *msg = "weight cut-off must be positive";
    return FDNN_E_ARG;
  }
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) {
    *msg = "cannot open model file " + path;
    return FDNN_E_IO;
  }
  std::fseek(fp, 0, SEEK_END);
  long sz = std::ftell(fp);
  std::rewind(fp);
  std::vector<uint8_t> buf(sz > 0 ? size_t(sz) : 0);
  size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), fp);
  std::fclose(fp);
  if (got != buf.size() || buf.size() < 4) {
    *msg = "short read on " + path;
    return FDNN_E_IO;
  }
  Cursor c{buf};
  const int n_affine = c.i32();
  if (n_affine < 4 || n_affine > kMaxQLayers + 1) {
    // dnn.cc:199 takes hidden_node_count_ from layers()[1]: that must be a hidden layer
    *msg = "model has " + std::to_string(n_affine) + " affine layers; need 4.." + std::to_string(kMaxQLayers + 1);
    return FDNN_E_FORMAT;
  }
  std::vector<RawLayer> layers(static_cast<size_t>(n_affine));
  int in_dim_file = 0;
  for (int j = 0; j < n_affine; ++j) {
    RawLayer &L = layers[size_t(j)];
    L.in_dim = c.i32();
    L.out_dim = c.i32();
    if (!c.ok || L.in_dim <= 0 || L.out_dim <= 0 || L.in_dim > (1 << 20) || L.out_dim > (1 << 19)) {  // (2^19 output nodes = 2048 soft-max tiles of 256: normalize_row's tree)
      *msg = "bad layer header at layer " + std::to_string(j);
      return FDNN_E_FORMAT;
    }
    if (j == 0) in_dim_file = L.in_dim;
    L.in_pad = j == 0 ? int(align_up(size_t(L.in_dim), 4)) : L.in_dim;  // float_dnn.cc:32-33
    if (!c.need(4 * (size_t(L.in_dim) * L.out_dim + size_t(L.out_dim)))) break;
    L.w.assign(size_t(L.out_dim) * L.in_pad, 0.0f);
    for (int o = 0; o < L.out_dim; ++o) c.f32(&L.w[size_t(o) * L.in_pad], size_t(L.in_dim));
    L.bias.resize(size_t(L.out_dim));
    c.f32(L.bias.data(), size_t(L.out_dim));
  }
  std::vector<float> shift(size_t(layers[0].in_pad), 0.0f), scale(size_t(layers[0].in_pad), 0.0f);
  c.f32(shift.data(), size_t(in_dim_file));  // zero padded, float_dnn.cc:60-66
  c.f32(scale.data(), size_t(in_dim_file));
  if (!c.ok) {
    *msg = "model file truncated: " + path;
    return FDNN_E_FORMAT;
  }
  const int H = layers[0].out_dim;
  if (H % 16) {
    *msg = "hidden width must be a multiple of 16 (README.md:10)";
    return FDNN_E_FORMAT;
  }
  for (int j = 1; j < n_affine; ++j) {
    const RawLayer &L = layers[size_t(j)];
    if (L.in_dim != H || (j < n_affine - 1 && L.out_dim != H)) {
      *msg = "all hidden layers must have the same width (dnn.cc:199-208)";
      return FDNN_E_FORMAT;
    }
  }
  if (H > 32768) {
    *msg = "hidden width above 32768 overflows the int32 accumulator bound";
    return FDNN_E_FORMAT;
  }
  return pack(layers, shift, scale, in_dim_file, cutoff, out, msg);
}

int adopt_blob(std::vector<uint8_t> &&bytes, HostModel *out, std::string *msg) {
  if (bytes.size() < sizeof(BlobHeader)) {
    *msg = "blob smaller than its header";
    return FDNN_E_FORMAT;
  }
  BlobHeader h;
  std::memcpy(&h, bytes.data(), sizeof(h));
  if (h.magic != kBlobMagic || h.version != kBlobVersion || h.total_bytes != bytes.size() || h.n_q < 3 ||
      h.n_q > kMaxQLayers) {
    *msg = "not a fast-dnn weight blob (magic/version/size mismatch)";
    return FDNN_E_FORMAT;
  }
  // Everything below drives device buffer descriptors and the scalar walk of the fix lists: a
  // truncated or foreign blob must be refused here, not read out of bounds on the GPU.
  const uint64_t total = h.total_bytes;
  auto inside = [&](uint64_t off, uint64_t len) { return off >= sizeof(BlobHeader) && off <= total && len <= total - off && off % 16 == 0; };
  auto bad = [&](const std::string &what) {
    *msg = "weight blob rejected: " + what;
    return FDNN_E_FORMAT;
  };
  if (h.n_affine != h.n_q + 1 || h.in_dim <= 0 || h.in_dim % 4 || h.hidden <= 0 || h.hidden % 16 || h.out_dim <= 0 ||
      h.in_dim > (1 << 20) || h.hidden > 32768 || h.out_dim > (1 << 19))
    return bad("inconsistent dimensions");
  const uint64_t H = uint64_t(h.hidden), D = uint64_t(h.in_dim);
  if (!inside(h.off_w0, 4 * H * D) || !inside(h.off_b0, 4 * H) || !inside(h.off_shift, 4 * D) || !inside(h.off_scale, 4 * D) ||
      !inside(h.off_lut, uint64_t(kLutExt) + 15) || !inside(h.off_lut2, uint64_t(kLut2Size) + 15))
    return bad("layer-0 / table sections outside the blob");
  for (int qi = 0; qi < h.n_q; ++qi) {
    const QLayerDesc &d = h.q[qi];
    const std::string L = "int8 layer " + std::to_string(qi + 1) + ": ";
    const bool last = qi == h.n_q - 1;
    if (d.rows <= 0 || d.cols != h.hidden || d.rows != (last ? h.out_dim : h.hidden) || d.rows_pad < d.rows ||
        d.rows_pad % kRowPad || d.rows_pad - d.rows >= kRowPad || d.cols_pad != int32_t(align_up(size_t(d.cols), kColPad)) + kRowSkew ||
        d.n_fix < 0)
      return bad(L + "inconsistent dimensions");
    const uint64_t rp = uint64_t(d.rows_pad), groups = rp / 64 + 1;
    if (!inside(d.off_w, rp * uint64_t(d.cols_pad)) || !inside(d.off_bias, 4 * rp) || !inside(d.off_wsum, 4 * rp) ||
        !inside(d.off_fix_grp, 4 * groups) || !inside(d.off_fix_ent, sizeof(FixEntry) * (uint64_t(d.n_fix) + 1)))
      return bad(L + "section outside the blob");
    const int32_t *grp = reinterpret_cast<const int32_t *>(bytes.data() + d.off_fix_grp);
    if (grp[0] != 0 || grp[groups - 1] != d.n_fix) return bad(L + "fix-up group table does not cover the entry list");
    for (uint64_t gi = 0; gi + 1 < groups; ++gi)
      if (grp[gi] > grp[gi + 1]) return bad(L + "fix-up group table not monotone");
    const FixEntry *ent = reinterpret_cast<const FixEntry *>(bytes.data() + d.off_fix_ent);
    for (uint64_t gi = 0; gi + 1 < groups; ++gi)
      for (int32_t e = grp[gi]; e < grp[gi + 1]; ++e) {
        // the GEMM derives an accumulator register from node - 64*group and reads LDS at column k
        if (ent[e].node < int32_t(gi) * 64 || ent[e].node >= int32_t(gi + 1) * 64 || ent[e].node >= d.rows || (ent[e].k & 1) ||
            int(ent[e].k) + 1 >= d.cols || (e > grp[gi] && ent[e].k < ent[e - 1].k))
          return bad(L + "fix-up entry out of range or out of order");
      }
  }
  out->hdr = h;
  out->blob = std::move(bytes);
  return FDNN_OK;
}

}  // namespace fdnn
