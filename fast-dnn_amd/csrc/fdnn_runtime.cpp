// fdnn_runtime.cpp -- device model, calculation contexts and the C-ABI (include/fdnn.h).
//
// Host-side counterpart of the reference's QuantizedDnn (dnn.h:106-142) and
// CalculationContext (dnn.h:144-208, dnn.cc:194-215, :402-454), re-designed for
// one MI355X: the model is one immutable packed blob in HBM; a context is a set
// of persistent device scratch buffers sized for n frames plus its own stream;
// fdnn_calculate draws contexts from a per-model pool so that concurrent callers
// (MultiThreadedStressTest.java:48-69) never share scratch.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <emmintrin.h>
#include <fcntl.h>
#include <cerrno>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdnn_internal.hpp"

namespace {
thread_local std::string g_err;
}  // namespace

namespace fdnn {
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
}  // namespace fdnn

using fdnn::DeviceGuard;
using fdnn::fail;
using fdnn::round_up;
using fdnn::Taps;
using fdnn::ctx_enter;
using fdnn::ctx_leave;
using fdnn::destroy_ctx;
using fdnn::make_ctx;
using fdnn::run_hidden;
using fdnn::run_output;

struct fdnn_host_model {
  fdnn::HostModel hm;
};

namespace fdnn {

using fdnn::BlobHeader;
using fdnn::QLayerDesc;

int build_l0_image(fdnn_model *m);

int upload_model(fdnn_model *m) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(FDNN_E_DEVICE, "no HIP device available: this library has no CPU path");
  if (m->device < 0 || m->device >= count) return fail(FDNN_E_ARG, "device index out of range");
  DeviceGuard g(m->device);
  if (!g.ok) return fail(FDNN_E_DEVICE, "hipSetDevice failed");
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_blob), m->hm.blob.size()));
  // exhaustive validation of the 3-op division per layer (see dequant() in fdnn_kernels.hip)
  unsigned long long *d_bad = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_bad), sizeof(unsigned long long) * fdnn::kMaxQLayers));
  HIP_TRY(hipMemset(d_bad, 0, sizeof(unsigned long long) * fdnn::kMaxQLayers));
  BlobHeader &h = m->hm.hdr;
  for (int qi = 0; qi < h.n_q; ++qi) {
    // same coefficient as an earlier layer -> same verdict, skip the sweep
    int same = -1;
    for (int pj = 0; pj < qi; ++pj)
      if (h.q[pj].coef == h.q[qi].coef) same = pj;
    if (same >= 0) continue;
    fdnn::launch_fastdiv_check(h.q[qi].coef, h.q[qi].rcp_coef, d_bad + qi, nullptr);
  }
  unsigned long long bad[fdnn::kMaxQLayers];
  HIP_TRY(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(d_bad));
  for (int qi = 0; qi < h.n_q; ++qi) {
    int src = qi;
    for (int pj = 0; pj < qi; ++pj)
      if (h.q[pj].coef == h.q[qi].coef) {
        src = pj;
        break;
      }
    // the fast epilogue = validated 3-op division + half-step table (needs bounded |lin|)
    h.q[qi].fastdiv_ok = (bad[src] == 0 && h.q[qi].lin_bounded) ? 1 : 0;
  }
  std::memcpy(m->hm.blob.data(), &h, sizeof(h));
  HIP_TRY(hipMemcpy(m->d_blob, m->hm.blob.data(), m->hm.blob.size(), hipMemcpyHostToDevice));
  return build_l0_image(m);
}

// Layer-0 weight image for the chain-pass kernel, built on the device from the blob's [H][D] rows.
int build_l0_image(fdnn_model *m) {
  const BlobHeader &h = m->hm.hdr;
  m->l0_jc = fdnn::l0_chunk_rows(h.in_dim);
  m->l0_j_pad = round_up(h.in_dim / 4, m->l0_jc);
  m->l0_h_ld = round_up(h.hidden, 128);
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_w0t), sizeof(float) * 4 * size_t(m->l0_j_pad) * m->l0_h_ld));
  fdnn::launch_l0_weight_image(reinterpret_cast<const float *>(m->d_blob + h.off_w0), m->d_w0t, h.hidden, h.in_dim, m->l0_j_pad,
                               m->l0_h_ld, nullptr);
  HIP_TRY(hipGetLastError());
  // ||w_n||_2 per layer-0 node, in double, rounded up to float: with the frame norms it bounds sum_k |x_k w_k| of every
  // output (Cauchy-Schwarz) for the screened path (fdnn_l0.hip)
  {
    const float *w0 = m->hm.w0();
    std::vector<float> wn(size_t(h.hidden));
    for (int i = 0; i < h.hidden; ++i) {
      double acc = 0.0;
      for (int k = 0; k < h.in_dim; ++k) acc += double(w0[size_t(i) * h.in_dim + k]) * double(w0[size_t(i) * h.in_dim + k]);
      const double up = std::sqrt(acc) * (1.0 + 1e-6);
      float f = float(up);
      if (double(f) < up) f = std::nextafter(f, std::numeric_limits<float>::infinity());
      wn[size_t(i)] = f;  // inf / NaN weights stay inf / NaN: every output of that node is then recomputed exactly
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_w0norm), sizeof(float) * wn.size()));
    HIP_TRY(hipMemcpy(m->d_w0norm, wn.data(), sizeof(float) * wn.size(), hipMemcpyHostToDevice));
    if (fdnn::l0_split_ok(h.in_dim, h.hidden)) {  // the node half of the int8 screening (fdnn_l0s.hip): digit planes + constants
      std::vector<int8_t> planes;
      std::vector<float> stat;
      std::vector<uint32_t> pairs;
      fdnn::l0_split_build_weights(w0, wn.data(), m->hm.blob.data() + h.off_lut2, h.hidden, h.in_dim, m->l0_h_ld, &planes, &stat, &pairs);
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_lutpair), sizeof(uint32_t) * pairs.size()));
      HIP_TRY(hipMemcpy(m->d_lutpair, pairs.data(), sizeof(uint32_t) * pairs.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_w0d), planes.size()));
      HIP_TRY(hipMemcpy(m->d_w0d, planes.data(), planes.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_w0stat), sizeof(float) * stat.size()));
      HIP_TRY(hipMemcpy(m->d_w0stat, stat.data(), sizeof(float) * stat.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_l0_stats), 32 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(m->d_l0_stats, 0, 32 * sizeof(unsigned long long)));
    if (hipHostMalloc(reinterpret_cast<void **>(&m->h_fuse_fault), sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
      *m->h_fuse_fault = 0;
      if (hipHostGetDevicePointer(reinterpret_cast<void **>(&m->d_fuse_fault), m->h_fuse_fault, 0) != hipSuccess) m->d_fuse_fault = nullptr;
    } else {
      (void)hipGetLastError();
      m->h_fuse_fault = nullptr;
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  return FDNN_OK;
}

void destroy_ctx(fdnn_ctx *c) {
  if (!c) return;
  DeviceGuard g(c->m->device);
  if (c->stream) {
    fuse_chain_retire_stream(c->m->device, c->stream);
    hipStreamSynchronize(c->stream);
  }
  hipFree(c->d_x);
  hipFree(c->d_xt);
  hipFree(c->d_l0park);
  hipFree(c->d_scr_count);
  hipFree(c->d_xd);
  hipFree(c->d_xstat);
  hipFree(c->d_comp);
  hipFree(c->d_glist);
  hipFree(c->d_glist_count);
  hipFree(c->d_scr_list);
  hipFree(c->d_act[0]);
  hipFree(c->d_act[1]);
  hipFree(c->d_out);
  hipFree(c->d_partial);
  hipFree(c->d_mask);
  hipFree(c->d_mask_bits);
  hipFree(c->d_fuse_s);
  hipFree(c->d_fuse_cnt);
  hipFree(c->d_fuse_flag);
  hipFree(c->d_chain_ctl);
  hipFree(c->d_chain_done);
  hipFree(c->d_chain_clk);
  if (c->h_chain_fault) hipHostFree(c->h_chain_fault);
  hipFree(c->d_l0_dbg_t);
  hipFree(c->d_l0_dbg_dd);
  if (c->h_mask_pin) hipHostFree(c->h_mask_pin);
  if (c->h_out_pin) hipHostFree(c->h_out_pin);
  if (c->done) hipEventDestroy(c->done);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int make_ctx(fdnn_model *m, int n, fdnn_ctx **out, bool lean) {
  DeviceGuard g(m->device);
  if (!g.ok) return fail(FDNN_E_DEVICE, "hipSetDevice failed");
  const BlobHeader &h = m->hm.hdr;
  fdnn_ctx *c = new fdnn_ctx();
  c->m = m;
  c->n = n;
  c->cap = round_up(std::max(n, 1), 64);
  c->act_ld = round_up(h.hidden, fdnn::kColPad) + fdnn::kRowSkew;
  int max_rows_pad = 0;
  for (int qi = 0; qi < h.n_q; ++qi) {
    max_rows_pad = std::max(max_rows_pad, h.q[qi].rows_pad);
  }
  const size_t np = size_t(c->cap);
  // the GEMMs work on whole frame tiles: every frame-indexed scratch carries one
  // tile of slack rows (a launch covers [first, first + round_up(count, tile)))
  const size_t npt = np + fdnn::kMaxFrameTile;
  hipError_t e = hipSuccess;
  auto alloc = [&](void **p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16);
  };
  if (!lean) alloc(reinterpret_cast<void **>(&c->d_x), sizeof(float) * np * h.in_dim);
  c->xt_ld = round_up(c->cap, 128);
  alloc(reinterpret_cast<void **>(&c->d_xt), sizeof(float) * 4 * size_t(m->l0_j_pad) * c->xt_ld);
  {  // screened layer-0 path: the per-tile lists of outputs to recompute exactly
    const size_t tiles = size_t(c->xt_ld / 64) * size_t((h.hidden + 127) / 128);  // 64- or 128-frame x 128-node screening tiles
    alloc(reinterpret_cast<void **>(&c->d_scr_count), sizeof(uint32_t) * tiles);
    alloc(reinterpret_cast<void **>(&c->d_scr_list), sizeof(uint16_t) * tiles * fdnn::kL0ScreenCap);
    if (e == hipSuccess) e = hipMemset(c->d_scr_count, 0, sizeof(uint32_t) * tiles);
  }
  if (m->d_w0d) {  // int8 screening: the frames' digit planes and row constants
    alloc(reinterpret_cast<void **>(&c->d_xd), fdnn::l0_split_plane_bytes(h.in_dim, c->xt_ld));
    alloc(reinterpret_cast<void **>(&c->d_xstat), sizeof(float) * 3 * size_t(c->xt_ld));
    c->glist_cap = int(std::min<size_t>(size_t(c->xt_ld) * size_t(m->l0_h_ld) / 16, size_t(1) << 26));  // 6 % of the outputs
    if (m->l0_list_cap > 0) c->glist_cap = std::min(c->glist_cap, m->l0_list_cap);  // (tests: fdnn_debug_set_l0_list_cap)
    alloc(reinterpret_cast<void **>(&c->d_glist), sizeof(uint2) * size_t(c->glist_cap));
    alloc(reinterpret_cast<void **>(&c->d_glist_count), sizeof(uint32_t) * 2);
    if (e == hipSuccess) e = hipMemset(c->d_glist, 0, sizeof(uint2) * size_t(c->glist_cap));
    if (e == hipSuccess) e = hipMemset(c->d_glist_count, 0, sizeof(uint32_t) * 2);
  }
  if (fdnn::l0_chain_node_tile() == 128)  // the 64-node tile keeps its partial sums in registers
    alloc(reinterpret_cast<void **>(&c->d_l0park), sizeof(float) * size_t(c->xt_ld) * m->l0_h_ld);
  alloc(reinterpret_cast<void **>(&c->d_act[0]), npt * c->act_ld);
  alloc(reinterpret_cast<void **>(&c->d_act[1]), npt * c->act_ld);
  if (!lean) alloc(reinterpret_cast<void **>(&c->d_out), sizeof(float) * np * h.out_dim);
  alloc(reinterpret_cast<void **>(&c->d_partial), sizeof(float) * npt * (max_rows_pad / fdnn::kPartialNodes));
  if (!lean) alloc(reinterpret_cast<void **>(&c->d_mask), np * h.out_dim);
  alloc(reinterpret_cast<void **>(&c->d_mask_bits), sizeof(uint64_t) * np * size_t((h.out_dim + 63) / 64));
  {  // fused soft-max (large dense batches): row sums per 256-node tile, counters and flags per tile (kept zero between launches)
    const size_t mt = size_t(max_rows_pad / 256), tiles = npt / 128 + 2;  // frame tiles of 128 frames and up
    alloc(reinterpret_cast<void **>(&c->d_fuse_s), sizeof(float) * npt * mt);
    alloc(reinterpret_cast<void **>(&c->d_fuse_cnt), sizeof(uint32_t) * 8 * tiles);
    alloc(reinterpret_cast<void **>(&c->d_fuse_flag), sizeof(uint32_t) * tiles * mt);
    c->fuse_cnt_bytes = sizeof(uint32_t) * 8 * tiles;
    c->fuse_flag_bytes = sizeof(uint32_t) * tiles * mt;
    if (e == hipSuccess) e = hipMemset(c->d_fuse_cnt, 0, sizeof(uint32_t) * 8 * tiles);
    if (e == hipSuccess) e = hipMemset(c->d_fuse_flag, 0, sizeof(uint32_t) * tiles * mt);
  }
  {  // chained hidden layers: queue heads + leave counter, per frame tile and layer the finished node tiles (zero between launches)
    const size_t tiles = npt / 256 + 2;
    alloc(reinterpret_cast<void **>(&c->d_chain_ctl), sizeof(uint32_t) * 16);
    alloc(reinterpret_cast<void **>(&c->d_chain_done), sizeof(uint32_t) * tiles * fdnn::kMaxChainLayers);
    c->chain_done_bytes = sizeof(uint32_t) * tiles * fdnn::kMaxChainLayers;
    if (e == hipSuccess) e = hipMemset(c->d_chain_ctl, 0, sizeof(uint32_t) * 16);
    if (e == hipSuccess) e = hipMemset(c->d_chain_done, 0, sizeof(uint32_t) * tiles * fdnn::kMaxChainLayers);
    if (e == hipSuccess && hipHostMalloc(reinterpret_cast<void **>(&c->h_chain_fault), sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
      *c->h_chain_fault = 0;
      if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_chain_fault), c->h_chain_fault, 0) != hipSuccess) c->d_chain_fault = nullptr;
    }
  }
  if (e == hipSuccess && !lean)  // (at least one padded row of slack)
    e = hipHostMalloc(reinterpret_cast<void **>(&c->h_mask_pin), std::max(size_t(kPinFrames) * h.out_dim, size_t(max_rows_pad)), hipHostMallocMapped);
  if (e == hipSuccess && !lean) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_mask_pin), c->h_mask_pin, 0);
  if (e == hipSuccess && !lean)
    e = hipHostMalloc(reinterpret_cast<void **>(&c->h_out_pin), sizeof(float) * kPinFrames * h.out_dim, hipHostMallocMapped);
  if (e == hipSuccess && !lean) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_out_pin), c->h_out_pin, 0);
  // The hipMemsets above are ordered on the NULL stream and return before they have run; the context's kernels go to
  // non-blocking streams, which do not wait for it.  Without this wait a new context's first kernels could start first and
  // have their counters (flagged-output list, fused soft-max arrivals) zeroed under them: a partly walked list -- a few
  // layer-0 bytes left at their screened value -- or a frame tile waiting for arrivals that were wiped.  Seen as one failure
  // in ten of the many-streams test (contexts are created while other callers' kernels run), never in a single-stream run.
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming | hipEventDisableSystemFence);  // (orders streams of one device only: no system-scope flush per record)
  if (e != hipSuccess) {
    std::string msg = std::string("context allocation for ") + std::to_string(n) + " frames: " + hipGetErrorString(e);
    destroy_ctx(c);
    return fail(e == hipErrorOutOfMemory ? FDNN_E_NOMEM : FDNN_E_DEVICE, msg);
  }
  *out = c;
  return FDNN_OK;
}

// Brackets one kernel launch with HIP events on the launch stream when profiling is on.
struct ProfScope {
  fdnn_model *m;
  hipStream_t s;
  int kind;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(fdnn_model *m_, hipStream_t s_, int kind_) : m(m_), s(s_), kind(kind_) {
    if (!m->profiling) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      a = b = nullptr;
      return;
    }
    hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!a) return;
    hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(m->mu);
    m->prof.push_back({kind, a, b});
  }
};

// one device-side u8 snapshot of the s8 activation buffer (taps only)
int snapshot_acts(const fdnn_ctx *c, int buf, uint8_t *d_dst, hipStream_t s) {
  const BlobHeader &h = c->m->hm.hdr;
  // compact [n_pad][act_ld] -> [n][H] while flipping bit 7
  if (c->act_ld == h.hidden) {
    fdnn::launch_xor80(c->d_act[buf], d_dst, size_t(c->n) * h.hidden, s);
  } else {
    for (int f = 0; f < c->n; ++f)
      fdnn::launch_xor80(c->d_act[buf] + size_t(f) * c->act_ld, d_dst + size_t(f) * h.hidden, size_t(h.hidden), s);
  }
  return FDNN_OK;
}

// GEMM descriptor of one int8 layer over `n` frames starting at activation row
// `act` (frame tile chosen for this n).
fdnn::QGemmParams prepare_qlayer(fdnn_ctx *c, const QLayerDesc &d, const int8_t *act, int n, hipStream_t s, bool output = false) {
  fdnn_model *m = c->m;
  const BlobHeader &h = m->hm.hdr;
  const uint8_t *B = m->d_blob;
  fdnn::QGemmParams g{};
  g.small = fdnn::qgemm_small_pick(d.rows_pad, d.cols_pad - fdnn::kRowSkew, n, d.fastdiv_ok, output) ? 1 : 0;
  g.frame_tile = g.small ? 32 : d.fastdiv_ok ? fdnn::qgemm_frame_tile(d.rows_pad, n) : 128;  // the true-divide kernel has one shape
  g.node_tile = (!g.small && d.fastdiv_ok) ? fdnn::qgemm_node_tile(d.rows_pad, n, output) : 256;
  if (g.node_tile == 128) g.frame_tile = 128;
  g.debug = fdnn::qgemm_debug_flags();
  g.n = n;
  g.n_pad = round_up(n, g.frame_tile);
  if (d.n_fix > 0) {
    g.fix_grp = reinterpret_cast<const int32_t *>(B + d.off_fix_grp);
    g.fix_ent = B + d.off_fix_ent;
  }
  g.w = reinterpret_cast<const int8_t *>(B + d.off_w);
  g.a = act;
  g.bias = reinterpret_cast<const float *>(B + d.off_bias);
  g.wsum = reinterpret_cast<const int32_t *>(B + d.off_wsum);
  g.lut = B + h.off_lut;
  g.lut2 = B + h.off_lut2;
  g.rows = d.rows;
  g.rows_pad = d.rows_pad;
  g.K = d.cols_pad - fdnn::kRowSkew;
  g.ldw = d.cols_pad;
  g.lda = c->act_ld;
  g.coef = d.coef;
  g.rcp_coef = d.rcp_coef;
  g.fastdiv = d.fastdiv_ok;
  return g;
}

// Layer 0 of the context's n frames into d_act[0] (shift/scale, fp32 affine, bias, table).
void run_layer0(fdnn_ctx *c, const float *d_x, hipStream_t s, const Taps *taps) {
  fdnn_model *m = c->m;
  const BlobHeader &h = m->hm.hdr;
  const uint8_t *B = m->d_blob;
  fdnn::L0Params l0{};
  l0.x = d_x;
  l0.shift = reinterpret_cast<const float *>(B + h.off_shift);
  l0.scale = reinterpret_cast<const float *>(B + h.off_scale);
  l0.w = reinterpret_cast<const float *>(B + h.off_w0);
  l0.bias = reinterpret_cast<const float *>(B + h.off_b0);
  l0.lut = B + h.off_lut;
  l0.act_out = c->d_act[0];
  l0.act_ld = c->act_ld;
  l0.tap_lin = taps ? taps->l0_lin : nullptr;
  l0.n = c->n;
  l0.n_rows = c->n;
  l0.D = h.in_dim;
  l0.H = h.hidden;
  l0.fma = m->l0_fma;
  l0.kernel = (c->l0_chain_only && m->l0_kernel == 0) ? 1 : m->l0_kernel;
  l0.xt = c->d_xt;
  l0.wt = m->d_w0t;
  l0.park = c->d_l0park;
  l0.wnorm = m->d_w0norm;
  l0.scr_count = c->d_scr_count;
  l0.scr_list = c->d_scr_list;
  l0.scr_stats = m->d_l0_stats;
  l0.xd = c->d_xd;
  l0.xstat = c->d_xstat;
  l0.wd = m->d_w0d;
  l0.wstat = m->d_w0stat;
  l0.luthalf = m->d_lutpair;
  l0.glist = c->d_glist;
  l0.glist_count = c->d_glist_count;
  l0.glist_cap = c->glist_cap;
  l0.dbg_t = c->d_l0_dbg_t;
  l0.dbg_dd = c->d_l0_dbg_dd;
  l0.j_pad = m->l0_j_pad;
  l0.jc = m->l0_jc;
  l0.n_ld = c->xt_ld;
  l0.h_ld = m->l0_h_ld;
  {
    ProfScope ps(m, s, FDNN_PROF_L0);
    fdnn::launch_l0(l0, s);
  }
}

// CalculateUntilLastHiddenLayer (dnn.cc:402-424): layer 0, then every int8
// hidden layer, layer-major over the whole frame batch.
int run_hidden(fdnn_ctx *c, const float *d_x, hipStream_t s, const Taps *taps) {
  fdnn_model *m = c->m;
  const BlobHeader &h = m->hm.hdr;
  run_layer0(c, d_x, s, taps);
  int cur = 0;
  if (taps && taps->u8_acts) snapshot_acts(c, cur, taps->u8_acts, s);
  // Large batches, no taps: the int8 hidden layers as ONE persistent launch (fdnn_chain.hip) -- tasks (layer, frame tile,
  // node tile) drawn from per-XCD queues, each waiting only for its own frame tile's node tiles of the layer before.
  const int n_hid = h.n_q - 1;
  // (advisor, round 5) a chained launch of this context ran into its wait bound: its counters are dirty and its results
  // were wrong.  Re-zero the counters in stream order and never chain on this context again; the host-synchronising dense
  // call re-runs its pass (calculate_on_one_device), the others observe fdnn_model_chain_faults.
  if (c->h_chain_fault && __atomic_load_n(c->h_chain_fault, __ATOMIC_RELAXED) != 0 && !c->chain_broken) {
    c->chain_broken = true;
    (void)hipMemsetAsync(c->d_chain_ctl, 0, sizeof(uint32_t) * 16, s);
    (void)hipMemsetAsync(c->d_chain_done, 0, c->chain_done_bytes, s);
  }
  bool chain = !taps && n_hid >= 2 && c->d_chain_ctl != nullptr && !c->chain_broken &&
               fdnn::qchain_ok(h.q[0].rows_pad, h.q[0].cols_pad - fdnn::kRowSkew, c->n, std::min(n_hid, fdnn::kMaxChainLayers));
  for (int qi = 0; chain && qi < n_hid; ++qi)
    chain = h.q[qi].fastdiv_ok && h.q[qi].rows == h.q[0].rows && h.q[qi].rows_pad == h.q[0].rows_pad && h.q[qi].cols_pad == h.q[0].cols_pad;
  if (chain) {
    const uint8_t *B = m->d_blob;
    for (int q0 = 0; q0 < n_hid; q0 += fdnn::kMaxChainLayers) {  // (nets deeper than kMaxChainLayers + 1: several chains)
      const int nl = std::min(fdnn::kMaxChainLayers, n_hid - q0);
      fdnn::QChainParams g{};
      for (int i = 0; i < nl; ++i) {
        const QLayerDesc &d = h.q[q0 + i];
        fdnn::QChainLayer &L = g.layer[i];
        L.w = reinterpret_cast<const int8_t *>(B + d.off_w);
        L.bias = reinterpret_cast<const float *>(B + d.off_bias);
        L.wsum = reinterpret_cast<const int32_t *>(B + d.off_wsum);
        L.fix_grp = d.n_fix > 0 ? reinterpret_cast<const int32_t *>(B + d.off_fix_grp) : nullptr;
        L.fix_ent = d.n_fix > 0 ? B + d.off_fix_ent : nullptr;
        L.coef = d.coef;
        L.rcp_coef = d.rcp_coef;
      }
      g.n_layers = nl;
      g.act[0] = c->d_act[cur];
      g.act[1] = c->d_act[cur ^ 1];
      g.lut2 = B + h.off_lut2;
      g.rows = h.q[0].rows;
      g.rows_pad = h.q[0].rows_pad;
      g.K = h.q[0].cols_pad - fdnn::kRowSkew;
      g.ldw = h.q[0].cols_pad;
      g.lda = c->act_ld;
      g.n = c->n;
      g.frame_tile = fdnn::qchain_frame_tile(g.rows_pad, c->n);
      g.n_pad = round_up(c->n, g.frame_tile);
      g.ctl = c->d_chain_ctl;
      g.done = c->d_chain_done;
      g.faults = m->d_l0_stats ? m->d_l0_stats + 3 : nullptr;
      g.fault_flag = c->d_chain_fault;
      g.clk = c->d_chain_clk;
      g.clk_cap = c->chain_clk_cap;
      {
        ProfScope ps(m, s, FDNN_PROF_HIDDEN);
        fdnn::launch_qchain(g, s);
      }
      cur ^= nl & 1;
    }
    c->last = cur;
    HIP_TRY(hipGetLastError());
    return FDNN_OK;
  }
  for (int qi = 0; qi < h.n_q - 1; ++qi) {
    fdnn::QGemmParams g = prepare_qlayer(c, h.q[qi], c->d_act[cur], c->n, s);
    g.act_out = c->d_act[cur ^ 1];
    g.act_ld = c->act_ld;
    g.tap_acc = (taps && taps->acc_hid) ? taps->acc_hid + size_t(qi) * c->n * h.hidden : nullptr;
    // Large batches of the production shape: the role-split kernel (fdnn_pp.hip) -- one wave of each SIMD in the k-loop,
    // its partner staging that tile's operands and running the epilogue of the tile before.  Identical bytes.
    static const int pp_only = [] { const char *e = std::getenv("FDNN_PP_ONLY"); return e ? std::atoi(e) : -1; }();
    const bool pp = !g.tap_acc && fdnn::qpp_ok(g.rows_pad, g.K, c->n, g.fastdiv != 0, g.fix_ent != nullptr) && (pp_only < 0 || pp_only == qi);
    if (pp) {
      g.small = 0;
      g.frame_tile = fdnn::qpp_frame_tile();
      g.n_pad = round_up(c->n, g.frame_tile);
    }
    {
      ProfScope ps(m, s, FDNN_PROF_HIDDEN);
      if (pp) fdnn::launch_qpp_hidden(g, s);
      else fdnn::launch_qgemm_hidden(g, s);
    }
    cur ^= 1;
    if (taps && taps->u8_acts) snapshot_acts(c, cur, taps->u8_acts + size_t(qi + 1) * c->n * h.hidden, s);
  }
  c->last = cur;
  HIP_TRY(hipGetLastError());
  return FDNN_OK;
}

// CalculateOutput (dnn.cc:428-454) / LazyOutputActivations (dnn.cc:355-392)
// over frames [first, first+count) of the context's last hidden activations.
// Rows [first+count, first+n_pad) are read by the GEMM as padding frames; the
// activation buffers carry one tile of slack rows for that.
// Two PROCESSES on one GPU.  Inside a process the fused soft-max launches of a device are chained (FuseChain below); two
// processes cannot be, and their fused kernels' workgroups can hold each other's CUs while every one of them sits out its
// bounded wait (correct results, ~100 x the latency).  So the first process to load a model on a device takes an advisory
// lock on /dev/shm/fdnn-gpu-<pci bus id> and keeps it for its lifetime; a process that finds the lock taken runs the
// UNFUSED output path (output kernel + scale pass: nothing in it waits for another workgroup) and says so once on stderr.
// The chained hidden-layer kernel needs no such care: its tasks only ever wait for tasks drawn earlier (fdnn_chain.hip).
// FDNN_FUSE_NORM=0 / 1 in the environment forces the unfused / fused path regardless.  Two containers that share a GPU but
// not /dev/shm cannot see each other: set FDNN_FUSE_NORM=0 there.
static int device_marker_state(int device) {  // 1 = this process owns the device's marker (or cannot tell), 0 = another process does
  static std::mutex mu;
  static int state[64];
  static bool known[64];
  std::lock_guard<std::mutex> lk(mu);
  const int d = device & 63;
  if (known[d]) return state[d];
  known[d] = true;
  state[d] = 1;
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess || !bus[0]) std::snprintf(bus, sizeof(bus), "dev%d", device);
  for (char *q = bus; *q; ++q)
    if (*q == ':' || *q == '/') *q = '-';
  // The marker is ADVISORY (advisor, round 5): a world-writable lock file in a sticky directory.  Never follow a planted
  // symlink (O_NOFOLLOW), only touch the mode of a regular file this user owns, and do not wander to another directory when
  // /dev/shm is unusable -- two processes looking in different places would both believe they are alone: "cannot tell" is
  // treated as SHARED (the unfused soft-max: correct, a little slower) and said once.
  const std::string path = std::string("/dev/shm/fdnn-gpu-") + bus;
  const int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
  if (fd < 0) {
    state[d] = 0;
    std::fprintf(stderr, "fast-dnn: cannot open the device marker %s (%s): assuming GPU %s is shared -- unfused soft-max (FDNN_FUSE_NORM=1 overrides)\n",
                 path.c_str(), std::strerror(errno), bus);
    return state[d];
  }
  struct stat st {};
  if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid()) fchmod(fd, 0666);
  if (flock(fd, LOCK_EX | LOCK_NB) == 0) return state[d];  // (kept open: the lock lives as long as this process)
  close(fd);
  state[d] = 0;
  std::fprintf(stderr, "fast-dnn: another process is scoring on GPU %s: this one runs the unfused soft-max (fdnn_device_shared)\n", bus);
  return state[d];
}
static std::atomic<int> g_fuse_override{-1};  // fdnn_debug_set_fuse: -1 = by environment / device marker, 0 = never, 1 = always
// The model's own evidence (see fdnn_model::h_fuse_fault): once a fused launch of this model has given up, it does not fuse again.
static bool model_may_fuse(fdnn_model *m) {
  if (!m->h_fuse_fault || __atomic_load_n(m->h_fuse_fault, __ATOMIC_RELAXED) == 0) return true;
  if (!m->fuse_fault_said) {
    m->fuse_fault_said = true;
    std::fprintf(stderr,
                 "fast-dnn: a fused soft-max launch on GPU %d sat out its bounded wait (another process scoring on this GPU that the device marker "
                 "did not show?): this model runs the unfused soft-max from here on (fdnn_model_fuse_giveups counts; FDNN_FUSE_NORM=1 overrides)\n",
                 m->device);
  }
  static const bool forced_on = [] {
    const char *e = std::getenv("FDNN_FUSE_NORM");
    return e && std::atoi(e) != 0;
  }();
  return forced_on || g_fuse_override.load(std::memory_order_relaxed) == 1;
}

static bool process_may_fuse(int device) {
  const int o = g_fuse_override.load(std::memory_order_relaxed);
  if (o >= 0) return o == 1;
  static const int forced = [] {
    const char *e = std::getenv("FDNN_FUSE_NORM");
    return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
  }();
  if (forced >= 0) return forced == 1;
  return device_marker_state(device) == 1;
}

// One chain of fused soft-max launches per device (see run_output).
struct FuseChain {
  std::mutex mu;
  hipEvent_t ev = nullptr;  // recorded after the device's latest fused output launch -- at once on a caller's stream;
  bool recorded = false;    // on a durable stream (stream_is_durable) only when a launch on ANOTHER stream needs it:
  bool pending = false;     // `pending` = the latest launch went to last_stream and `ev` does not cover it yet
  hipStream_t last_stream = nullptr;
};
static FuseChain &fuse_chain(int device) {
  static FuseChain chains[64];
  return chains[device & 63];
}
// REQUIREMENT (advisor, round 4): whoever owns a stream that stream_is_durable() answers true for -- a context's own stream,
// a scoring loop's -- must call this before destroying it (destroy_ctx and fdnn_server_free do): the chain keeps the raw
// handle of the stream its last launch went to and records its event there later.  A caller-created stream is never
// remembered (its record is made at once).
void fuse_chain_retire_stream(int device, hipStream_t s) {
  FuseChain &fc = fuse_chain(device);
  std::lock_guard<std::mutex> lk(fc.mu);
  if (fc.pending && fc.last_stream == s && fc.ev) {
    fc.recorded = hipEventRecord(fc.ev, s) == hipSuccess;
    fc.pending = false;
  }
}

int run_output(fdnn_ctx *c, int first, int count, const int8_t *d_masks, float *d_out, hipStream_t s, const Taps *taps,
               float *d_final, hipStream_t tail, hipEvent_t gemm_done, const uint64_t *d_bits) {  // d_final: where the probabilities go (default: in place in d_out)
  fdnn_model *m = c->m;
  const BlobHeader &h = m->hm.hdr;
  const QLayerDesc &d = h.q[h.n_q - 1];
  if (c->last < 0) return fail(FDNN_E_STATE, "output requested before the hidden layers were computed");
  if (first < 0 || count < 0 || first + count > c->n) return fail(FDNN_E_ARG, "frame range outside the context");
  if (count == 0) return FDNN_OK;
  // (One frame and decoder-sized blocks take the small-batch GEMM kernels as well: a row-by-row kernel that skips the
  // masked-out nodes as the reference does, dnn.cc:361-365, was measured against them -- DESIGN.md section 6 -- and lost
  // at every block size from 40 % active nodes up: the call is two launches of latency either way.)
  fdnn::QGemmParams g = prepare_qlayer(c, d, c->d_act[c->last] + size_t(first) * c->act_ld, count, s, true);
  g.out = d_out;
  g.partial = c->d_partial;
  g.partial_ld = g.n_pad;
  if (d_bits && !d_masks) {  // bit-mask entry points
    if (g.small || (taps && taps->acc_out)) {
      if (!c->d_mask) return fail(FDNN_E_STATE, "this context has no byte-mask scratch for a small bit-mask batch");
      ProfScope ps(m, s, FDNN_PROF_OUTPUT);
      fdnn::launch_mask_unpack(d_bits, c->d_mask, count, d.rows, s);
      d_masks = c->d_mask;
      d_bits = nullptr;
    } else {
      d_masks = reinterpret_cast<const int8_t *>(d_bits);  // (non-null = the masked instances; they read mask_bits only)
    }
  }
  g.mask = d_masks;
  if (d_bits) {
    g.mask_bits = d_bits;
    g.mask_wpr = (d.rows + 63) / 64;
  } else if (d_masks && !g.small && !(taps && taps->acc_out)) {
    // large-batch production instances: the mask travels as bits (one pass over the caller's bytes at HBM speed)
    ProfScope ps(m, s, FDNN_PROF_OUTPUT);
    fdnn::launch_mask_pack(d_masks, c->d_mask_bits, count, d.rows, s);
    g.mask_bits = c->d_mask_bits;
    g.mask_wpr = (d.rows + 63) / 64;
    c->mask_bits_packed = true;  // (fdnn_ctx_lazy_output_batch needs the same bits for its compacted return: not twice)
  }
  g.tap_acc = taps ? taps->acc_out : nullptr;
  g.tap_logit = taps ? taps->logits : nullptr;
  g.acc_probe = taps ? taps->acc_probe : nullptr;
  g.probe_stride = taps ? std::max(1, taps->probe_stride) : 1;
  const bool fused = !c->no_fuse && process_may_fuse(m->device) && model_may_fuse(m) && fdnn::qgemm_fused_ok(g);  // (taps exclude it; the accumulator probe of the parity tests does not)
  // the role-split fused kernel (fdnn_ppo.hip): dense, unprobed, the production shape, enough frames for a steady state
  const bool ppo = fused && !g.mask && !g.mask_bits && !g.acc_probe && fdnn::qppo_ok(d.rows, g.rows_pad, g.K, count, g.fastdiv != 0, g.fix_ent != nullptr);
  if (ppo) {
    g.small = 0;
    g.frame_tile = fdnn::qppo_frame_tile();
    g.n_pad = round_up(count, g.frame_tile);
  }
  if (fused && m->h_fuse_fault && __atomic_load_n(m->h_fuse_fault, __ATOMIC_RELAXED) != 0) {
    // (fusing although a launch of this model gave up before -- FDNN_FUSE_NORM=1 / fdnn_debug_set_fuse(1): a workgroup that
    // gave up may have left its exchange counters half counted; they are zeroed in stream order before every such launch)
    HIP_TRY(hipMemsetAsync(c->d_fuse_cnt, 0, c->fuse_cnt_bytes, s));
    HIP_TRY(hipMemsetAsync(c->d_fuse_flag, 0, c->fuse_flag_bytes, s));
  }
  if (fused) {
    g.final = d_final ? d_final : d_out;
    g.fuse_s = c->d_fuse_s;
    g.fuse_cnt = c->d_fuse_cnt;
    g.fuse_flag = c->d_fuse_flag;
    g.fuse_giveups = m->d_l0_stats ? m->d_l0_stats + 2 : nullptr;
    g.fuse_fault = m->d_fuse_fault;
    static const int stagger = [] {
      const char *e = FDNN_TUNE_ENV("FDNN_FUSE_STAGGER");
      return e ? std::atoi(e) : 0;
    }();
    g.fuse_stagger = stagger;
  }
  {
    ProfScope ps(m, s, FDNN_PROF_OUTPUT);
    if (fused) {
      // The fused kernel's workgroups wait for their frame tile's other node tiles, which is safe while ONE such kernel is
      // being dispatched (in block order: the oldest unfinished frame tile always has all its workgroups resident) and a
      // latency cliff when several are -- nine partially dispatched frame tiles fill the 256 CUs and every one of them sits
      // in its bounded wait.  So the fused launches of a DEVICE form one chain, whatever stream, context, model or entry
      // point they come from: each waits for the previous one's event and records its own.  Other kernels overlap them
      // freely (they wait for nothing).  Two PROCESSES on one GPU cannot be chained: FDNN_FUSE_NORM=0.
      FuseChain &fc = fuse_chain(m->device);
      std::lock_guard<std::mutex> lk(fc.mu);
      if (!fc.ev) HIP_TRY(hipEventCreateWithFlags(&fc.ev, hipEventDisableTiming | hipEventDisableSystemFence));
      static const bool eager = FDNN_TUNE_ENV("FDNN_EAGER_EVENTS") != nullptr;
      if (fc.last_stream != s || !(fc.pending || fc.recorded)) {  // (same stream as the previous fused launch: in order already)
        if (fc.pending) {  // the deferred record: the tail of the previous launch's stream is behind that launch
          fc.pending = false;
          fc.recorded = false;
          HIP_TRY(hipEventRecord(fc.ev, fc.last_stream));
          fc.recorded = true;
        }
        if (fc.recorded) HIP_TRY(hipStreamWaitEvent(s, fc.ev, 0));
      }
      if (ppo) fdnn::launch_qppo_output(g, s);
      else fdnn::launch_qgemm_output(g, s);
      fc.last_stream = s;
      if (stream_is_durable(c, s) && !eager) {
        fc.pending = true;
      } else {
        fc.pending = false;
        fc.recorded = false;
        HIP_TRY(hipEventRecord(fc.ev, s));
        fc.recorded = true;
      }
    } else {
      fdnn::launch_qgemm_output(g, s);
    }
  }
  hipStream_t ns = s;
  if (tail && gemm_done) {  // the scale pass goes to the tail stream, behind the GEMM (fused: nothing is left to run
    HIP_TRY(hipEventRecord(gemm_done, s));  // there, but the caller records its completion event on the tail stream)
    HIP_TRY(hipStreamWaitEvent(tail, gemm_done, 0));
    ns = tail;
  }
  if (!fused) {
    ProfScope ps(m, ns, FDNN_PROF_NORMALIZE);
    fdnn::launch_normalize(d_out, d_final ? d_final : d_out, c->d_partial, count, g.partial_ld, d.rows, d.rows_pad / fdnn::kPartialNodes, ns,
                           ns != s);
  }
  HIP_TRY(hipGetLastError());
  return FDNN_OK;
}

bool output_will_fuse(fdnn_ctx *c, int count, const int8_t *d_masks) {
  const BlobHeader &h = c->m->hm.hdr;
  const QLayerDesc &d = h.q[h.n_q - 1];
  fdnn::QGemmParams g = prepare_qlayer(c, d, c->d_act[0], count, nullptr, true);
  g.mask = d_masks;
  if (d_masks && !g.small) g.mask_bits = c->d_mask_bits;  // (what run_output will do)
  return process_may_fuse(c->m->device) && model_may_fuse(c->m) && fdnn::qgemm_fused_ok(g);
}

// Device -> pageable host memory for large results (the 8000-float rows of a whole batch:
// 320 MB for 10 000 frames).  hipMemcpy into resident pageable memory runs at ~45 GB/s, but a
// result array that was just allocated (numpy's np.empty; a JVM float[] is already zeroed) is
// not resident: the copy then takes every first-touch page fault on one thread, 31 of the
// call's 34 ms.  So the destination is faulted in first, by a few host threads, one byte per
// page (it is about to be overwritten anyway) -- while the GPU is still computing, the
// kernels are already enqueued -- and the copy itself stays the runtime's.
// Measured, 10 000 frames: fresh array 33 -> 24.5 ms, resident array 7.1 ms either way.  (A
// hand-made pipeline over pinned bounce buffers with parallel host memcpy: 7.9 / 23.5 ms.)
// Synchronises the stream.
int copy_out(void *dst, const void *d_src, size_t bytes, hipStream_t s) {
  static const bool plain = FDNN_TUNE_ENV("FDNN_PLAIN_COPY_OUT") != nullptr;
  if (bytes >= (size_t(64) << 20) && !plain) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = static_cast<int>(std::min<unsigned>(16, std::max<unsigned>(1, hw / 2)));
    const size_t page = 4096;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + page - 1) & ~(page - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(page - 1);
    if (hi > lo) {
      const size_t pages = (hi - lo) / page;
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t)
        pool.emplace_back([=] {
          const size_t b = pages * size_t(t) / size_t(T), e = pages * size_t(t + 1) / size_t(T);
          if (e == b) return;
          volatile char *q = reinterpret_cast<volatile char *>(lo);
          for (size_t i = b; i < e; ++i) q[i * page] = 0;  // (MADV_POPULATE_WRITE and MADV_HUGEPAGE were both slower)
        });
      for (auto &th : pool) th.join();
    }
  }
  HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FDNN_OK;
}

// pool: idle contexts with capacity >= n.  The hand-over event orders a reuse
// on another stream behind the previous user's kernels.
int acquire_ctx(fdnn_model *m, int n, fdnn_ctx **out) {
  fdnn_ctx *c = nullptr;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    int best = -1;
    for (size_t i = 0; i < m->pool.size(); ++i)
      if (m->pool[i]->cap >= n && (best < 0 || m->pool[i]->cap < m->pool[size_t(best)]->cap)) best = int(i);
    if (best >= 0) {
      c = m->pool[size_t(best)];
      m->pool.erase(m->pool.begin() + best);
    }
  }
  if (!c) {
    int rc = make_ctx(m, n, &c);
    if (rc) return rc;
    c->pooled = true;
  }
  c->n = n;
  c->last = -1;
  *out = c;
  return FDNN_OK;
}

// A context's scratch is touched from two kinds of streams: the caller's (the *_device entry
// points) and the context's own (the host-pointer entry points; created non-blocking, so not even
// the NULL stream orders it).  Every entry point therefore starts by making its stream wait for
// the context's last enqueued work and ends by recording it: calculateUntilOutputDevice(stream)
// followed by calculateForOutputNodes() or hiddenActivations() reads finished activations
// without the caller synchronising anything.  On one stream both calls are no-ops for the device.
// Would run_hidden chain the int8 hidden layers of an n-frame batch of this model?  (The context-independent part of its
// decision: taps and a context whose chain faulted are the caller's business.)
bool hidden_layers_chain(const fdnn_model *m, int n) {
  const BlobHeader &h = m->hm.hdr;
  const int n_hid = h.n_q - 1;
  if (n_hid < 2) return false;
  for (int qi = 0; qi < n_hid; ++qi)
    if (!(h.q[qi].fastdiv_ok && h.q[qi].rows == h.q[0].rows && h.q[qi].rows_pad == h.q[0].rows_pad && h.q[qi].cols_pad == h.q[0].cols_pad)) return false;
  return fdnn::qchain_ok(h.q[0].rows_pad, h.q[0].cols_pad - fdnn::kRowSkew, n, std::min(n_hid, fdnn::kMaxChainLayers));
}

std::vector<std::pair<int, int>> frame_chunks(int n, const fdnn_model *m, bool assume_chained) {
  std::vector<std::pair<int, int>> out;
  static const int kChunk = [] {  // FDNN_CHUNK_FRAMES: measurement switch (0 = never chunk; otherwise whole rounds)
    const char *e = std::getenv("FDNN_CHUNK_FRAMES");
    const int v = e ? std::atoi(e) : kChunkFrames;
    return v <= 0 ? 0 : std::max(kRoundFrames, v / kRoundFrames * kRoundFrames);
  }();
  // A batch whose hidden layers run as a launch per layer (chaining off, fewer than two int8 hidden layers, a layer without
  // the validated division: hidden_layers_chain says) pays one more, nearly empty round of workgroups in every
  // layer for a few frames past a whole round -- 11 000 frames as one batch 977 us, as 10 240 + 760: 754 + 150 (rounds 2-4;
  // advisor, round 5: the split had been dropped for every configuration).  Such a tail (up to kChunkTailSplit frames) goes
  // as a small batch of its own.
  auto split_tail = [&](int off, int cnt) {
    const int tail = cnt % kRoundFrames;
    const bool chained = m ? hidden_layers_chain(m, cnt) : assume_chained;
    if (!chained && cnt > kRoundFrames && tail > 0 && tail <= kChunkTailSplit) {
      out.emplace_back(off, cnt - tail);
      out.emplace_back(off + cnt - tail, tail);
    } else {
      out.emplace_back(off, cnt);
    }
  };
  if (kChunk <= 0 || n <= kChunk) {
    split_tail(0, n);
    return out;
  }
  // Chunks of kChunk frames, what is left over as one more batch.  (Rounds 2-4 kept a batch to whole rounds of workgroups and
  // split a small tail off as a batch of its own -- 11 000 frames as one batch cost a second, nearly empty round in every
  // hidden layer: 977 us against 754 + 150.  From ~9 800 frames up the hidden layers now run as one chained launch whose
  // tasks flow across the layers (fdnn_chain.hip), the partial round is gone -- 11 000 frames: layer 0 + hidden layers 534 ->
  // 459 us -- and a tail costs less inside the batch than as a call of its own: tools/chain_sweep.py, tools/batch_sweep.py.)
  int off = 0;
  while (n - off > kChunk) {
    out.emplace_back(off, kChunk);
    off += kChunk;
  }
  split_tail(off, n - off);
  return out;
}


bool stream_is_durable(const fdnn_ctx *c, hipStream_t s) {
  return s == nullptr || s == c->stream || s == c->durable[0] || s == c->durable[1] || s == c->durable[2];
}
hipError_t ctx_enter(fdnn_ctx *c, hipStream_t s) {
  if (c->done_stream == s && (c->done_valid || c->done_pending)) return hipSuccess;  // same stream: already in order
  if (c->done_pending) {  // the deferred record: the tail of done_stream covers everything the context enqueued there
    c->done_pending = false;
    const hipError_t e = hipEventRecord(c->done, c->done_stream);
    c->done_valid = e == hipSuccess;
    if (e != hipSuccess) return e;
  }
  return hipStreamWaitEvent(s, c->done, 0);
}
void ctx_leave(fdnn_ctx *c, hipStream_t s) {
  static const bool eager = FDNN_TUNE_ENV("FDNN_EAGER_EVENTS") != nullptr;  // (measurements: every record made at once, as before round 4)
  c->done_stream = s;
  if (stream_is_durable(c, s) && !eager) {
    c->done_pending = true;
    c->done_valid = false;
    return;
  }
  c->done_pending = false;
  c->done_valid = hipEventRecord(c->done, s) == hipSuccess;
}
void ctx_wait_host(fdnn_ctx *c) {
  if (c->done_pending)
    hipStreamSynchronize(c->done_stream);
  else if (c->done_valid)
    hipEventSynchronize(c->done);
}

void release_ctx(fdnn_ctx *c, hipStream_t s) {
  ctx_leave(c, s);
  fdnn_model *m = c->m;
  fdnn_ctx *victim = nullptr;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->pool.push_back(c);
    if (m->pool.size() > 8) {  // keep a handful of the largest contexts
      auto it = std::min_element(m->pool.begin(), m->pool.end(),
                                 [](const fdnn_ctx *a, const fdnn_ctx *b) { return a->cap < b->cap; });
      victim = *it;
      m->pool.erase(it);
    }
  }
  if (victim) {
    ctx_wait_host(victim);
    destroy_ctx(victim);
  }
}

// fdnn_calculate on the model's own device (the group path calls this per shard).
int calculate_on_one_device(fdnn_model *m, const float *x, int n, int dim, int batch_hint, float *out) {
  (void)batch_hint;
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;  // QuantizedDnn.java:154-156
  if (!x || !out) return fail(FDNN_E_ARG, "null buffer");
  const BlobHeader &h = m->hm.hdr;
  if (dim != h.in_dim)  // QuantizedDnn.java:157-161
    return fail(FDNN_E_ARG, "input vector size " + std::to_string(dim) + " must be equal with network input size " +
                                std::to_string(h.in_dim));
  if (m->batcher) {  // coalesced with the other callers' utterances (fdnn_server.cpp)
    uint64_t ticket = 0;
    int brc = fdnn_server_submit(m->batcher, x, n, nullptr, out, &ticket);
    if (!brc) brc = fdnn_server_wait(m->batcher, ticket);
    return brc;
  }
  DeviceGuard g(m->device);
  fdnn_ctx *c = nullptr;
  int rc = acquire_ctx(m, n, &c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  hipError_t e = ctx_enter(c, s);
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x, sizeof(float) * size_t(n) * dim, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const unsigned long long unwritten_before = m->h_fuse_fault ? __atomic_load_n(m->h_fuse_fault, __ATOMIC_RELAXED) >> 32 : 0ull;
    rc = run_hidden(c, c->d_x, s, nullptr);
    if (!rc) rc = run_output(c, 0, n, nullptr, c->d_out, s, nullptr);
    if (!rc) rc = copy_out(out, c->d_out, sizeof(float) * size_t(n) * h.out_dim, s);
    if (rc) hipStreamSynchronize(s);
    // copy_out has synchronised: did a fused soft-max workgroup of THIS pass sit out its bounded wait?  fdnn_gemm.hip's
    // tiles finish such a frame tile after the fact, fdnn_ppo.hip's leave the half's rows unwritten (and say so through the
    // same word): the output layer runs again -- unfused now, model_may_fuse has seen the word -- over the activations that
    // are still in the context.  (The word's upper half counts such halves.  Callers of the *_device entry points observe
    // fdnn_model_fuse_giveups after their own synchronisation: INTEGRATION.md.)
    if (!rc && m->h_fuse_fault && (__atomic_load_n(m->h_fuse_fault, __ATOMIC_RELAXED) >> 32) != unwritten_before) {
      const bool saved = c->no_fuse;
      c->no_fuse = true;  // (whatever FDNN_FUSE_NORM says)
      rc = run_output(c, 0, n, nullptr, c->d_out, s, nullptr);
      c->no_fuse = saved;
      if (!rc) rc = copy_out(out, c->d_out, sizeof(float) * size_t(n) * h.out_dim, s);
      if (rc) hipStreamSynchronize(s);
    }
    // copy_out has synchronised: did this pass's chained launch run into its wait bound?  Then what it computed on may not have
    // been written -- run the pass again, layer by layer (run_hidden sees the flag, re-zeroes the counters, stops chaining)
    if (!rc && c->h_chain_fault && __atomic_load_n(c->h_chain_fault, __ATOMIC_RELAXED) != 0 && !c->chain_broken) {
      rc = run_hidden(c, c->d_x, s, nullptr);
      if (!rc) rc = run_output(c, 0, n, nullptr, c->d_out, s, nullptr);
      if (!rc) rc = copy_out(out, c->d_out, sizeof(float) * size_t(n) * h.out_dim, s);
      if (rc) hipStreamSynchronize(s);
    }
  }
  release_ctx(c, s);
  if (rc) return rc;
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("fdnn_calculate: ") + hipGetErrorString(e));
  return FDNN_OK;
}

// The host half of a compacted lazy return (lazy_compact_kernel): a compacted row is the row's inactive value followed by
// its active nodes' probabilities in node order; expanding it = every node reads the next active value or the inactive one.
// With AVX-512 that is one expanding load per 16 nodes (the mask's 16 bits select which lanes take the next values from
// memory); checked at run time, scalar otherwise (whole words of inactive / active nodes as runs).  100 frames x 8000 nodes
// at 40 %: 0.55 ms scalar bit by bit (round 5's first form), ~0.05 ms with the expanding loads -- this runs on the caller's
// thread for every utterance, next to a 1.3 MB transfer.
static void expand_row_scalar(float *row, const float *vals, const uint64_t *brow, size_t O) {
  const size_t wpr = (O + 63) / 64;
  const uint64_t tail_mask = (O & 63) ? ((uint64_t(1) << (O & 63)) - 1) : ~uint64_t(0);
  const float inact = vals[0];
  const float *src = vals + 1;
  for (size_t w = 0; w < wpr; ++w) {
    const size_t width = std::min<size_t>(64, O - 64 * w);
    uint64_t word = brow[w];
    if (w + 1 == wpr) word &= tail_mask;
    float *dst = row + 64 * w;
    if (word == 0) {
      std::fill(dst, dst + width, inact);
    } else if (width == 64 && word == ~uint64_t(0)) {
      std::memcpy(dst, src, 64 * sizeof(float));
      src += 64;
    } else {
      for (size_t b = 0; b < width; ++b) dst[b] = ((word >> b) & 1u) ? *src++ : inact;
    }
  }
}

__attribute__((target("avx512f,popcnt"))) static void expand_row_avx512(float *row, const float *vals, const uint64_t *brow, size_t O) {
  const __m512 inact = _mm512_set1_ps(vals[0]);
  const float *src = vals + 1;
  const size_t full = O / 64;
  for (size_t w = 0; w < full; ++w) {
    const uint64_t word = brow[w];
    float *dst = row + 64 * w;
    for (int q = 0; q < 4; ++q) {
      const __mmask16 mk = static_cast<__mmask16>(word >> (16 * q));
      _mm512_storeu_ps(dst + 16 * q, _mm512_mask_expandloadu_ps(inact, mk, src));
      src += __builtin_popcount(static_cast<unsigned>(mk));
    }
  }
  if (O & 63) {  // the last, partial word
    const uint64_t word = brow[full] & ((uint64_t(1) << (O & 63)) - 1);
    float *dst = row + 64 * full;
    for (size_t b = 0; b < (O & 63); ++b) dst[b] = ((word >> b) & 1u) ? *src++ : vals[0];
  }
}

static std::atomic<bool> g_expand_scalar{false};  // fdnn_debug_lazy_expand, mode 1
static void expand_row(float *row, const float *vals, const uint64_t *brow, size_t O) {
  static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("popcnt");
  if (wide && !g_expand_scalar.load(std::memory_order_relaxed))
    expand_row_avx512(row, vals, brow, O);
  else
    expand_row_scalar(row, vals, brow, O);
}

// `count` compacted rows of `stride` floats sit in the TAIL of the caller's [count][O] block and are expanded front to back
// (row f's place never reaches the compacted rows of later frames; its own is copied aside first).  bits: the rows' masks.
void lazy_expand_rows(float *out, int count, size_t O, size_t stride, const uint64_t *bits) {
  const size_t wpr = (O + 63) / 64;
  const float *land = out + size_t(count) * O - size_t(count) * stride;
  std::vector<float> mine(stride);
  for (int f = 0; f < count; ++f) {
    std::memcpy(mine.data(), land + size_t(f) * stride, sizeof(float) * stride);
    expand_row(out + size_t(f) * O, mine.data(), bits + size_t(f) * wpr, O);
  }
}

// The same from a separate buffer of compacted rows (the scoring loop's pinned landing area).
void lazy_expand_rows_from(float *out, const float *comp, int count, size_t O, size_t stride, const uint64_t *bits) {
  const size_t wpr = (O + 63) / 64;
  for (int f = 0; f < count; ++f) expand_row(out + size_t(f) * O, comp + size_t(f) * stride, bits + size_t(f) * wpr, O);
}

}  // namespace fdnn

using namespace fdnn;

// =====================================================================  C-ABI
extern "C" {

const char *fdnn_last_error(void) { return g_err.c_str(); }
const char *fdnn_version(void) { return "fast-dnn_amd 0.1 (gfx950)"; }

int fdnn_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int fdnn_model_load_on(const char *path, float cutoff, int device, fdnn_model **out) {
  if (!path || !out) return fail(FDNN_E_ARG, "null argument");
  *out = nullptr;
  fdnn_model *m = new fdnn_model();
  std::string msg;
  int rc = fdnn::load_host_model(path, cutoff, &m->hm, &msg);
  if (rc) {
    delete m;
    return fail(rc, msg);
  }
  m->device = device;
  rc = upload_model(m);
  if (!rc) (void)fdnn_device_shared(device);  // take (or find taken) the device's process marker now, not at the first large call
  if (rc) {
    if (m->d_blob) hipFree(m->d_blob);
    if (m->d_w0t) hipFree(m->d_w0t);
    if (m->d_w0norm) hipFree(m->d_w0norm);
    if (m->d_w0d) hipFree(m->d_w0d);
    if (m->d_w0stat) hipFree(m->d_w0stat);
    if (m->d_lutpair) hipFree(m->d_lutpair);
    if (m->d_l0_stats) hipFree(m->d_l0_stats);
    if (m->h_fuse_fault) hipHostFree(m->h_fuse_fault);
    delete m;
    return rc;
  }
  if (const char *env = std::getenv("FDNN_BATCHER")) {  // max_frames[:depth[:linger_us]]
    int mf = 0, depth = 2, linger = 0;
    if (std::sscanf(env, "%d:%d:%d", &mf, &depth, &linger) >= 1 && mf > 0) {
      rc = fdnn_model_enable_batcher(m, mf, depth, linger);
      if (rc) {
        fdnn_model_free(m);
        return rc;
      }
    }
  }
  *out = m;
  return FDNN_OK;
}

int fdnn_model_load(const char *path, float cutoff, fdnn_model **out) {
  // FDNN_DEVICES="0,1,2,3" or "all": one replica per listed device, weights distributed at load,
  // fdnn_calculate on the returned handle shards its frames over them (fdnn_group.cpp) -- how the
  // unmodified Java class reaches every GPU of the node.
  if (const char *env = std::getenv("FDNN_DEVICES")) {
    std::vector<int> devs;
    if (std::strcmp(env, "all") == 0) {
      for (int d = 0; d < fdnn_device_count(); ++d) devs.push_back(d);
    } else {
      for (const char *q = env; *q;) {
        char *end = nullptr;
        const long v = std::strtol(q, &end, 10);
        if (end == q) break;
        devs.push_back(int(v));
        q = (*end == ',') ? end + 1 : end;
      }
    }
    if (devs.size() > 1) {
      if (!out) return fail(FDNN_E_ARG, "null argument");
      fdnn_group *g = nullptr;
      int rc = fdnn_group_load(path, cutoff, devs.data(), int(devs.size()), &g);
      if (rc) return rc;
      fdnn_group_attach(g);
      *out = fdnn_group_model(g, 0);
      return FDNN_OK;
    }
    if (devs.size() == 1) return fdnn_model_load_on(path, cutoff, devs[0], out);
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return fdnn_model_load_on(path, cutoff, dev, out);
}

int fdnn_model_enable_batcher(fdnn_model *m, int max_frames, int depth, int linger_us) {
  if (!m) return fail(FDNN_E_ARG, "null model");
  if (m->batcher) return fail(FDNN_E_STATE, "the model already has a batcher");
  fdnn_server *srv = nullptr;
  int rc = fdnn_server_create(m, max_frames, depth, &srv);
  if (!rc) rc = fdnn_server_set_linger_us(srv, linger_us);
  if (rc) {
    fdnn_server_free(srv);
    return rc;
  }
  m->batcher = srv;
  return FDNN_OK;
}

void fdnn_model_free(fdnn_model *m) {
  if (!m) return;
  if (m->group) {  // the leader of an attached group: the group owns every replica, this one included
    fdnn_group_free(m->group);
    return;
  }
  if (m->batcher) fdnn_server_free(m->batcher);
  m->batcher = nullptr;
  for (fdnn_ctx *c : m->pool) destroy_ctx(c);
  m->pool.clear();
  {
    DeviceGuard g(m->device);
    hipFree(m->d_blob);
    hipFree(m->d_w0t);
    hipFree(m->d_w0norm);
    hipFree(m->d_w0d);
    hipFree(m->d_w0stat);
    hipFree(m->d_lutpair);
    hipFree(m->d_l0_stats);
    if (m->h_fuse_fault) hipHostFree(m->h_fuse_fault);
  }
  delete m;
}

int fdnn_model_input_dim(const fdnn_model *m) { return m ? m->hm.hdr.in_dim : -1; }
int fdnn_model_output_dim(const fdnn_model *m) { return m ? m->hm.hdr.out_dim : -1; }
int fdnn_model_hidden_dim(const fdnn_model *m) { return m ? m->hm.hdr.hidden : -1; }
int fdnn_model_layer_count(const fdnn_model *m) { return m ? m->hm.hdr.n_q + 1 : -1; }  // jni_dnn.cc:155
int fdnn_model_device(const fdnn_model *m) { return m ? m->device : -1; }

int fdnn_model_layer_dim(const fdnn_model *m, int index) {
  if (!m) return -1;
  const BlobHeader &h = m->hm.hdr;
  if (index < 0) return -1;
  if (index == 0) return h.hidden;           // input_layer()->node_count(), jni_dnn.cc:144-146
  if (index >= h.n_q) return -1;             // layers()[index] must exist (see fdnn.h)
  return h.q[index].rows;                    // layers()[index]->node_count(), jni_dnn.cc:147
}

int fdnn_model_set_l0_fma(fdnn_model *m, int on) {
  if (!m) return fail(FDNN_E_ARG, "null model");
  m->l0_fma = on ? 1 : 0;
  return FDNN_OK;
}

int fdnn_debug_set_l0_kernel(fdnn_model *m, int kind) {
  if (!m) return fail(FDNN_E_ARG, "null model");
  if (kind < 0 || kind > 4) return fail(FDNN_E_ARG, "layer-0 kernel kind must be 0 .. 4");
  m->l0_kernel = kind;
  return FDNN_OK;
}

int fdnn_device_shared(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return fail(FDNN_E_ARG, "no such device");
  return device_marker_state(device) == 1 ? 0 : 1;
}

int fdnn_debug_set_fuse(int mode) {
  if (mode < -1 || mode > 1) return fail(FDNN_E_ARG, "fuse mode must be -1, 0 or 1");
  g_fuse_override.store(mode, std::memory_order_relaxed);
  return FDNN_OK;
}

int fdnn_debug_lazy_expand(float *out, const float *comp, int count, int O, int stride, const uint64_t *bits, int mode) {
  if (!out || !comp || !bits || count < 0 || O <= 0 || stride <= 0 || stride > O + 1 || mode < 0 || mode > 2) return fail(FDNN_E_ARG, "bad argument");
  g_expand_scalar.store(mode == 1, std::memory_order_relaxed);
  if (mode == 2) {
    if (stride > O) return fail(FDNN_E_ARG, "the in-place form needs stride <= O");
    std::memmove(out + size_t(count) * size_t(O) - size_t(count) * size_t(stride), comp, sizeof(float) * size_t(count) * size_t(stride));
    fdnn::lazy_expand_rows(out, count, size_t(O), size_t(stride), bits);
  } else {
    fdnn::lazy_expand_rows_from(out, comp, count, size_t(O), size_t(stride), bits);
  }
  g_expand_scalar.store(false, std::memory_order_relaxed);
  return FDNN_OK;
}

int fdnn_debug_set_l0_list_cap(fdnn_model *m, int cap) {
  if (!m || cap < 0) return fail(FDNN_E_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(m->mu);
  for (fdnn_ctx *c : m->pool) destroy_ctx(c);  // pooled contexts carry the old capacity
  m->pool.clear();
  m->l0_list_cap = cap;
  return FDNN_OK;
}

int fdnn_debug_set_pp(int mode, int min_frames) {
  if (mode < -1 || mode > 1) return fail(FDNN_E_ARG, "pp mode must be -1, 0 or 1");
  fdnn::qpp_set_mode(mode, min_frames);
  return FDNN_OK;
}

int fdnn_debug_raise_fuse_fault(fdnn_model *m, int value) {
  if (!m) return fail(FDNN_E_ARG, "null model");
  if (!m->h_fuse_fault) return fail(FDNN_E_STATE, "this model has no fault word");
  __atomic_store_n(m->h_fuse_fault, value ? 1ull : 0ull, __ATOMIC_RELAXED);
  if (!value) m->fuse_fault_said = false;
  return FDNN_OK;
}

int fdnn_debug_set_ppo(int mode) {
  if (mode < -1 || mode > 1) return fail(FDNN_E_ARG, "ppo mode must be -1, 0 or 1");
  fdnn::qppo_set_mode(mode);
  return FDNN_OK;
}

int fdnn_debug_set_chain(int mode, int min_frames) {
  if (mode < -1 || mode > 1) return fail(FDNN_E_ARG, "chain mode must be -1, 0 or 1");
  fdnn::qchain_set_mode(mode, min_frames);
  return FDNN_OK;
}

int fdnn_debug_chain_clocks(fdnn_ctx *c, long long *out, int cap_tasks) {
  if (!c || cap_tasks <= 0) return fail(FDNN_E_ARG, "bad argument");
  DeviceGuard g(c->m->device);
  const size_t words = 8 + size_t(cap_tasks) * 10;
  if (!out) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_chain_clk) hipFree(c->d_chain_clk);
    c->d_chain_clk = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_chain_clk), words * sizeof(long long)));
    HIP_TRY(hipMemset(c->d_chain_clk, 0, words * sizeof(long long)));
    HIP_TRY(hipDeviceSynchronize());
    c->chain_clk_cap = cap_tasks;
    return FDNN_OK;
  }
  if (!c->d_chain_clk || cap_tasks > c->chain_clk_cap) return fail(FDNN_E_STATE, "no clock buffer of that size");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, c->d_chain_clk, words * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(c->d_chain_clk, 0, 8 * sizeof(long long)));
  HIP_TRY(hipDeviceSynchronize());
  return FDNN_OK;
}

// ---------------------------------------------------------------- contexts
int fdnn_ctx_create(fdnn_model *m, int n, int batch_hint, fdnn_ctx **out) {
  (void)batch_hint;  // frame blocking is a CPU cache device; results never depend on it
  if (!m || !out) return fail(FDNN_E_ARG, "null argument");
  if (n < 0) return fail(FDNN_E_ARG, "negative frame count");
  return make_ctx(m, n, out);
}

void fdnn_ctx_free(fdnn_ctx *c) {
  if (!c) return;
  destroy_ctx(c);
}

int fdnn_ctx_frame_count(const fdnn_ctx *c) { return c ? c->n : -1; }
int fdnn_ctx_output_dim(const fdnn_ctx *c) { return c ? c->m->hm.hdr.out_dim : -1; }

int fdnn_ctx_forward_hidden_device(fdnn_ctx *c, const float *d_x, void *stream) {
  if (!c || (!d_x && c->n)) return fail(FDNN_E_ARG, "null argument");
  if (c->n == 0) {
    c->last = 0;
    return FDNN_OK;
  }
  DeviceGuard g(c->m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(ctx_enter(c, s));
  int rc = run_hidden(c, d_x, s, nullptr);
  ctx_leave(c, s);
  return rc;
}

int fdnn_ctx_forward_hidden(fdnn_ctx *c, const float *x) {
  if (!c || (!x && c->n)) return fail(FDNN_E_ARG, "null argument");
  if (c->n == 0) {
    c->last = 0;
    return FDNN_OK;
  }
  DeviceGuard g(c->m->device);
  const BlobHeader &h = c->m->hm.hdr;
  HIP_TRY(ctx_enter(c, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_x, x, sizeof(float) * size_t(c->n) * h.in_dim, hipMemcpyHostToDevice, c->stream));
  int rc = run_hidden(c, c->d_x, c->stream, nullptr);
  ctx_leave(c, c->stream);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return FDNN_OK;
}

int fdnn_ctx_lazy_output_batch_device(fdnn_ctx *c, int first, int count, const int8_t *d_masks, float *d_out,
                                      void *stream) {
  if (!c || !d_out) return fail(FDNN_E_ARG, "null argument");
  DeviceGuard g(c->m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(ctx_enter(c, s));
  int rc = run_output(c, first, count, d_masks, d_out, s, nullptr);
  ctx_leave(c, s);
  return rc;
}

static int lazy_copy_out(fdnn_ctx *c, int count, const uint64_t *d_bits, const uint64_t *bits, float *out, hipStream_t s);

// masks [count][O] bytes (non-zero = active, dnn.cc:361) -> bits [count][ceil(O / 64)], 16 bytes per step
static void pack_mask_rows(const int8_t *masks, int count, size_t O, uint64_t *bits) {
  const size_t wpr = (O + 63) / 64;
  const __m128i zero = _mm_setzero_si128();
  for (int f = 0; f < count; ++f) {
    const int8_t *mrow = masks + size_t(f) * O;
    uint64_t *brow = bits + size_t(f) * wpr;
    size_t k = 0;
    for (size_t w = 0; w < wpr; ++w) {
      uint64_t word = 0;
      for (int q = 0; q < 4 && k + 16 <= O; ++q, k += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(mrow + k));
        word |= uint64_t(uint32_t(~_mm_movemask_epi8(_mm_cmpeq_epi8(v, zero))) & 0xffffu) << (16 * q);
      }
      const size_t base = 64 * w;
      for (; k < O && k < base + 64; ++k) word |= uint64_t(mrow[k] != 0) << (k - base);
      brow[w] = word;
    }
  }
}

int fdnn_ctx_lazy_output_batch(fdnn_ctx *c, int first, int count, const int8_t *masks, float *out) {
  if (!c || !out || !masks) return fail(FDNN_E_ARG, "null argument");
  if (c->last < 0) return fail(FDNN_E_STATE, "calculateLazy before calculateUntilOutput");
  if (first < 0 || count < 0 || first + count > c->n) return fail(FDNN_E_ARG, "frame index outside the context");
  if (count == 0) return FDNN_OK;
  DeviceGuard g(c->m->device);
  const BlobHeader &h = c->m->hm.hdr;
  const size_t O = size_t(h.out_dim);
  HIP_TRY(ctx_enter(c, c->stream));
  if (count <= kPinFrames) {  // the per-frame protocol: no copy commands (see fdnn_ctx)
    std::memcpy(c->h_mask_pin, masks, size_t(count) * O);
    // (blocks of 1..8 frames go through the small GEMM kernel's 32-frame tile, which reads the mask bytes straight from the
    // host-mapped staging; nets the small kernel cannot take -- K > 2048, no validated fast division -- fall to the large
    // tiles behind a mask_pack pass over the same staging)
    const int rc = run_output(c, first, count, c->d_mask_pin, c->d_out, c->stream, nullptr, c->d_out_pin);
    ctx_leave(c, c->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::memcpy(out, c->h_out_pin, sizeof(float) * size_t(count) * O);
    return FDNN_OK;
  }
  HIP_TRY(hipMemcpyAsync(c->d_mask, masks, size_t(count) * O, hipMemcpyHostToDevice, c->stream));
  c->mask_bits_packed = false;
  int rc = run_output(c, first, count, c->d_mask, c->d_out, c->stream, nullptr);
  if (!rc) {
    // the same masks as bits, for the compacted return (lazy_copy_out): on the host while the GPU computes, on the device
    // by the pack kernel (a large batch's output kernel has run it already)
    const size_t wpr = (O + 63) / 64;
    std::vector<uint64_t> hb(size_t(count) * wpr, 0);
    pack_mask_rows(masks, count, O, hb.data());
    if (!c->mask_bits_packed) fdnn::launch_mask_pack(c->d_mask, c->d_mask_bits, count, int(O), c->stream);  // (a large batch's output kernel has)
    rc = lazy_copy_out(c, count, c->d_mask_bits, hb.data(), out, c->stream);
  }
  ctx_leave(c, c->stream);
  return rc;
}

int fdnn_ctx_lazy_output_batch_bits_device(fdnn_ctx *c, int first, int count, const uint64_t *d_bits, float *d_out, void *stream) {
  if (!c || !d_out || !d_bits) return fail(FDNN_E_ARG, "null argument");
  DeviceGuard g(c->m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(ctx_enter(c, s));
  int rc = run_output(c, first, count, nullptr, d_out, s, nullptr, nullptr, nullptr, nullptr, d_bits);
  ctx_leave(c, s);
  return rc;
}

// Lazy results to a host caller.  Every inactive node of a row reads the same 1 / total (dnn.cc:366-369, :389), so what
// crosses PCIe is the active nodes' probabilities and that one value per frame (lazy_compact_kernel); the rows are
// rebuilt on the host inside the caller's array: the compacted block lands in its tail, and the rows are expanded front to
// back (row f's place never reaches the compacted rows of later frames; its own is copied aside first).  d_bits: the
// masks of the `count` frames on the device; bits: the same on the host.  With mostly active masks (> 3/4) the plain copy
// is used.  Synchronises the stream.
static int lazy_copy_out(fdnn_ctx *c, int count, const uint64_t *d_bits, const uint64_t *bits, float *out, hipStream_t s) {
  const size_t O = size_t(c->m->hm.hdr.out_dim), wpr = (O + 63) / 64;
  static const bool no_compact = FDNN_TUNE_ENV("FDNN_LAZY_NO_COMPACT") != nullptr;
  size_t most = 0;
  const uint64_t tail_mask = (O & 63) ? ((uint64_t(1) << (O & 63)) - 1) : ~uint64_t(0);
  for (int f = 0; f < count; ++f) {
    size_t k = 0;
    const uint64_t *row = bits + size_t(f) * wpr;
    for (size_t w = 0; w + 1 < wpr; ++w) k += size_t(__builtin_popcountll(row[w]));
    k += size_t(__builtin_popcountll(row[wpr - 1] & tail_mask));
    most = std::max(most, k);
  }
  const size_t stride = most + 1;
  if (no_compact || stride * 4 > O * 3) return copy_out(out, c->d_out, sizeof(float) * size_t(count) * O, s);
  if (c->comp_floats < size_t(count) * stride) {
    if (c->d_comp) HIP_TRY(hipFree(c->d_comp));
    c->d_comp = nullptr;
    c->comp_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_comp), sizeof(float) * size_t(count) * stride));
    c->comp_floats = size_t(count) * stride;
  }
  fdnn::launch_lazy_compact(c->d_out, d_bits, c->d_comp, count, int(O), int(stride), s);
  float *land = out + size_t(count) * O - size_t(count) * stride;
  HIP_TRY(hipMemcpyAsync(land, c->d_comp, sizeof(float) * size_t(count) * stride, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  fdnn::lazy_expand_rows(out, count, O, stride, bits);
  return FDNN_OK;
}

int fdnn_ctx_lazy_output_batch_bits(fdnn_ctx *c, int first, int count, const uint64_t *bits, float *out) {
  if (!c || !out || !bits) return fail(FDNN_E_ARG, "null argument");
  if (c->last < 0) return fail(FDNN_E_STATE, "calculateLazy before calculateUntilOutput");
  if (first < 0 || count < 0 || first + count > c->n) return fail(FDNN_E_ARG, "frame index outside the context");
  if (count == 0) return FDNN_OK;
  DeviceGuard g(c->m->device);
  const BlobHeader &h = c->m->hm.hdr;
  const size_t O = size_t(h.out_dim), wpr = (O + 63) / 64;
  HIP_TRY(ctx_enter(c, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_mask_bits, bits, sizeof(uint64_t) * size_t(count) * wpr, hipMemcpyHostToDevice, c->stream));
  int rc = run_output(c, first, count, nullptr, c->d_out, c->stream, nullptr, nullptr, nullptr, nullptr, c->d_mask_bits);
  if (!rc) rc = lazy_copy_out(c, count, c->d_mask_bits, bits, out, c->stream);
  ctx_leave(c, c->stream);
  return rc;
}

int fdnn_ctx_lazy_output(fdnn_ctx *c, int frame, const int8_t *mask, float *out) {
  return fdnn_ctx_lazy_output_batch(c, frame, 1, mask, out);
}

int fdnn_ctx_output_device(fdnn_ctx *c, float *d_out, void *stream) {
  if (!c || !d_out) return fail(FDNN_E_ARG, "null argument");
  DeviceGuard g(c->m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(ctx_enter(c, s));
  int rc = run_output(c, 0, c->n, nullptr, d_out, s, nullptr);
  ctx_leave(c, s);
  return rc;
}

int fdnn_ctx_output(fdnn_ctx *c, float *out) {
  if (!c || !out) return fail(FDNN_E_ARG, "null argument");
  if (c->n == 0) return FDNN_OK;
  DeviceGuard g(c->m->device);
  HIP_TRY(ctx_enter(c, c->stream));
  int rc = run_output(c, 0, c->n, nullptr, c->d_out, c->stream, nullptr);
  ctx_leave(c, c->stream);
  if (rc) return rc;
  return copy_out(out, c->d_out, sizeof(float) * size_t(c->n) * c->m->hm.hdr.out_dim, c->stream);
}

int fdnn_ctx_read_hidden(fdnn_ctx *c, uint8_t *out) {
  if (!c || !out) return fail(FDNN_E_ARG, "null argument");
  if (c->last < 0) return fail(FDNN_E_STATE, "hidden layers not computed yet");
  if (c->n == 0) return FDNN_OK;
  DeviceGuard g(c->m->device);
  const int H = c->m->hm.hdr.hidden;
  std::vector<int8_t> tmp(size_t(c->n) * c->act_ld);
  HIP_TRY(ctx_enter(c, c->stream));  // the hidden layers may have been enqueued on a caller's stream
  HIP_TRY(hipMemcpyAsync(tmp.data(), c->d_act[c->last], tmp.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int f = 0; f < c->n; ++f)
    for (int i = 0; i < H; ++i) out[size_t(f) * H + i] = uint8_t(tmp[size_t(f) * c->act_ld + i]) ^ 0x80;
  return FDNN_OK;
}

// ---------------------------------------------------------------- dense path
int fdnn_calculate_device(fdnn_model *m, const float *d_x, int n, float *d_out, void *stream) {
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;
  if (!d_x || !d_out) return fail(FDNN_E_ARG, "null buffer");
  DeviceGuard g(m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  fdnn_ctx *c = nullptr;
  const auto chunks = fdnn::frame_chunks(n, m);
  int cap = 0;  // the scratch only has to hold the largest chunk
  for (const auto &ch : chunks) cap = std::max(cap, ch.second);
  int rc = acquire_ctx(m, cap, &c);
  if (rc) return rc;
  const hipError_t e = ctx_enter(c, s);
  if (e == hipSuccess) {
    const size_t D = size_t(m->hm.hdr.in_dim), O = size_t(m->hm.hdr.out_dim);
    for (const auto &ch : chunks) {  // frames are independent: a chunk is a batch of its own
      c->n = ch.second;
      rc = run_hidden(c, d_x + size_t(ch.first) * D, s, nullptr);
      if (!rc) rc = run_output(c, 0, ch.second, nullptr, d_out + size_t(ch.first) * O, s, nullptr);
      if (rc) break;
    }
  }
  release_ctx(c, s);  // also on the error paths: the context goes back to the pool
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("fdnn_calculate_device: ") + hipGetErrorString(e));
  return rc;
}

int fdnn_calculate(fdnn_model *m, const float *x, int n, int dim, int batch_hint, float *out) {
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;  // QuantizedDnn.java:154-156
  if (!x || !out) return fail(FDNN_E_ARG, "null buffer");
  if (m->group) return fdnn_group_calculate(m->group, x, n, dim, batch_hint, out);  // sharded over the node's devices
  return fdnn::calculate_on_one_device(m, x, n, dim, batch_hint, out);
}

// One-call lazy scoring: hidden layers + masked output + compacted return in ONE call and ONE stream synchronisation
// (a LazyContext costs two calls and two synchronisations per utterance: calculateUntilOutput, then the masked rows).
int fdnn_calculate_lazy_bits(fdnn_model *m, const float *x, int n, int dim, const uint64_t *bits, float *out) {
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;
  if (!x || !out || !bits) return fail(FDNN_E_ARG, "null buffer");
  const BlobHeader &h = m->hm.hdr;
  if (dim != h.in_dim)
    return fail(FDNN_E_ARG, "input vector size " + std::to_string(dim) + " must be equal with network input size " + std::to_string(h.in_dim));
  if (m->batcher) {  // coalesced with the other callers' lazy utterances (fdnn_server.cpp), rows back compacted
    uint64_t ticket = 0;
    int brc = fdnn_server_submit_lazy_bits(m->batcher, x, n, bits, out, &ticket);
    if (!brc) brc = fdnn_server_wait(m->batcher, ticket);
    return brc;
  }
  DeviceGuard g(m->device);
  const size_t wpr = (size_t(h.out_dim) + 63) / 64;
  int rc = FDNN_OK;
  for (int first = 0; first < n && !rc; first += fdnn::kChunkFrames) {  // (very large calls: chunk by chunk, as the dense call)
    const int cnt = std::min(fdnn::kChunkFrames, n - first);
    fdnn_ctx *c = nullptr;
    rc = acquire_ctx(m, cnt, &c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    hipError_t e = ctx_enter(c, s);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x + size_t(first) * dim, sizeof(float) * size_t(cnt) * dim, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_mask_bits, bits + size_t(first) * wpr, sizeof(uint64_t) * size_t(cnt) * wpr, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
      rc = run_hidden(c, c->d_x, s, nullptr);
      if (!rc) rc = run_output(c, 0, cnt, nullptr, c->d_out, s, nullptr, nullptr, nullptr, nullptr, c->d_mask_bits);
      if (!rc) rc = lazy_copy_out(c, cnt, c->d_mask_bits, bits + size_t(first) * wpr, out + size_t(first) * h.out_dim, s);
      if (rc) hipStreamSynchronize(s);
    }
    release_ctx(c, s);
    if (!rc && e != hipSuccess) rc = fail(FDNN_E_DEVICE, std::string("fdnn_calculate_lazy_bits: ") + hipGetErrorString(e));
  }
  return rc;
}

int fdnn_calculate_lazy(fdnn_model *m, const float *x, int n, int dim, const int8_t *masks, float *out) {
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;
  if (!masks) return fail(FDNN_E_ARG, "null buffer");
  const size_t O = size_t(m->hm.hdr.out_dim), wpr = (O + 63) / 64;
  std::vector<uint64_t> hb(size_t(n) * wpr);
  pack_mask_rows(masks, n, O, hb.data());
  return fdnn_calculate_lazy_bits(m, x, n, dim, hb.data(), out);
}

int fdnn_calculate_lazy_bits_device(fdnn_model *m, const float *d_x, int n, const uint64_t *d_bits, float *d_out, void *stream) {
  if (!m || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;
  if (!d_x || !d_out || !d_bits) return fail(FDNN_E_ARG, "null buffer");
  DeviceGuard g(m->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  fdnn_ctx *c = nullptr;
  const auto chunks = fdnn::frame_chunks(n, m);
  int cap = 0;
  for (const auto &ch : chunks) cap = std::max(cap, ch.second);
  int rc = acquire_ctx(m, cap, &c);
  if (rc) return rc;
  const hipError_t e = ctx_enter(c, s);
  if (e == hipSuccess) {
    const size_t D = size_t(m->hm.hdr.in_dim), O = size_t(m->hm.hdr.out_dim), wpr = (O + 63) / 64;
    for (const auto &ch : chunks) {
      c->n = ch.second;
      rc = run_hidden(c, d_x + size_t(ch.first) * D, s, nullptr);
      if (!rc) rc = run_output(c, 0, ch.second, nullptr, d_out + size_t(ch.first) * O, s, nullptr, nullptr, nullptr, nullptr, d_bits + size_t(ch.first) * wpr);
      if (rc) break;
    }
  }
  release_ctx(c, s);
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("fdnn_calculate_lazy_bits_device: ") + hipGetErrorString(e));
  return rc;
}

// ---------------------------------------------------------------- taps
int fdnn_debug_forward_taps(fdnn_model *m, const float *x, int n, const int8_t *masks, float *l0_lin, uint8_t *u8_acts,
                            int32_t *acc_hid, int32_t *acc_out, float *logits, float *probs) {
  if (!m || !x || n <= 0) return fail(FDNN_E_ARG, "bad argument");
  DeviceGuard g(m->device);
  const BlobHeader &h = m->hm.hdr;
  const size_t H = size_t(h.hidden), O = size_t(h.out_dim), N = size_t(n);
  const int n_hidden = h.n_q;  // fp32 layer + (n_q - 1) int8 hidden layers
  fdnn_ctx *c = nullptr;
  int rc = make_ctx(m, n, &c);
  if (rc) return rc;
  Taps t{};
  hipError_t e = hipSuccess;
  auto alloc = [&](void **p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes);
  };
  alloc(reinterpret_cast<void **>(&t.l0_lin), sizeof(float) * N * H);
  alloc(reinterpret_cast<void **>(&t.u8_acts), size_t(n_hidden) * N * H);
  alloc(reinterpret_cast<void **>(&t.acc_hid), sizeof(int32_t) * size_t(std::max(n_hidden - 1, 1)) * N * H);
  alloc(reinterpret_cast<void **>(&t.acc_out), sizeof(int32_t) * N * O);
  alloc(reinterpret_cast<void **>(&t.logits), sizeof(float) * N * O);
  hipStream_t s = c->stream;
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x, sizeof(float) * N * h.in_dim, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && masks) e = hipMemcpyAsync(c->d_mask, masks, N * O, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    rc = run_hidden(c, c->d_x, s, &t);
    if (!rc) rc = run_output(c, 0, n, masks ? c->d_mask : nullptr, c->d_out, s, &t);
  }
  auto fetch = [&](void *dst, const void *src, size_t bytes) {
    if (dst && e == hipSuccess && !rc) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
  };
  fetch(l0_lin, t.l0_lin, sizeof(float) * N * H);
  fetch(u8_acts, t.u8_acts, size_t(n_hidden) * N * H);
  fetch(acc_hid, t.acc_hid, sizeof(int32_t) * size_t(n_hidden - 1) * N * H);
  fetch(acc_out, t.acc_out, sizeof(int32_t) * N * O);
  fetch(logits, t.logits, sizeof(float) * N * O);
  fetch(probs, c->d_out, sizeof(float) * N * O);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(t.l0_lin);
  hipFree(t.u8_acts);
  hipFree(t.acc_hid);
  hipFree(t.acc_out);
  hipFree(t.logits);
  fdnn_ctx_free(c);
  if (rc) return rc;
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("taps: ") + hipGetErrorString(e));
  return FDNN_OK;
}

int fdnn_debug_frame_chunks(int n, int *chunks, int cap) { return fdnn_debug_frame_chunks_for(n, 1, chunks, cap); }

int fdnn_debug_frame_chunks_for(int n, int chained, int *chunks, int cap) {
  if (n <= 0 || !chunks || cap <= 0) return -1;
  const auto v = fdnn::frame_chunks(n, nullptr, chained != 0);
  if (static_cast<int>(v.size()) > cap) return -1;
  for (size_t i = 0; i < v.size(); ++i) {
    chunks[2 * i] = v[i].first;
    chunks[2 * i + 1] = v[i].second;
  }
  return static_cast<int>(v.size());
}

int fdnn_debug_production_acc_out(fdnn_model *m, const float *x, int n, int stride, const int8_t *masks, int32_t *acc, float *probs) {
  if (!m || !x || !acc || n <= 0 || stride <= 0) return fail(FDNN_E_ARG, "bad argument");
  DeviceGuard g(m->device);
  const BlobHeader &h = m->hm.hdr;
  const size_t O = size_t(h.out_dim), N = size_t(n), NP = size_t((n + stride - 1) / stride);
  fdnn_ctx *c = nullptr;
  int rc = make_ctx(m, n, &c);
  if (rc) return rc;
  Taps t{};  // only the probe: hidden layers and output layer run their production instances
  t.probe_stride = stride;
  hipStream_t s = c->stream;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&t.acc_probe), sizeof(int32_t) * NP * O);
  if (e == hipSuccess) e = hipMemsetAsync(t.acc_probe, 0xff, sizeof(int32_t) * NP * O, s);
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x, sizeof(float) * N * h.in_dim, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && masks) e = hipMemcpyAsync(c->d_mask, masks, N * O, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    rc = run_hidden(c, c->d_x, s, nullptr);
    if (!rc) rc = run_output(c, 0, n, masks ? c->d_mask : nullptr, c->d_out, s, &t);
  }
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(acc, t.acc_probe, sizeof(int32_t) * NP * O, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && !rc && probs) e = hipMemcpyAsync(probs, c->d_out, sizeof(float) * N * O, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(t.acc_probe);
  fdnn_ctx_free(c);
  if (rc) return rc;
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("acc probe: ") + hipGetErrorString(e));
  return FDNN_OK;
}

int fdnn_debug_device_counters(fdnn_model *m, unsigned long long *out, int n) {  // raw device counter words (kernel clock stamps of timing builds live at [4..])
  if (!m || !out || n < 0 || n > 32) return fail(FDNN_E_ARG, "bad argument");
  if (!m->d_l0_stats) return fail(FDNN_E_STATE, "no counters");
  DeviceGuard g(m->device);
  const hipError_t e = hipMemcpy(out, m->d_l0_stats, sizeof(unsigned long long) * size_t(n), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("device counters: ") + hipGetErrorString(e));
  return FDNN_OK;
}

int fdnn_model_chain_faults(fdnn_model *m, unsigned long long *faults) {
  if (!m || !faults) return fail(FDNN_E_ARG, "null argument");
  *faults = 0;
  if (!m->d_l0_stats) return FDNN_OK;
  DeviceGuard g(m->device);
  const hipError_t e = hipMemcpy(faults, m->d_l0_stats + 3, sizeof(*faults), hipMemcpyDeviceToHost);  // (synchronizes with the device)
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("chain_faults: ") + hipGetErrorString(e));
  return FDNN_OK;
}

int fdnn_model_fuse_giveups(fdnn_model *m, unsigned long long *tiles) {
  if (!m || !tiles) return fail(FDNN_E_ARG, "null argument");
  *tiles = 0;
  if (!m->d_l0_stats) return FDNN_OK;
  DeviceGuard g(m->device);
  const hipError_t e = hipMemcpy(tiles, m->d_l0_stats + 2, sizeof(*tiles), hipMemcpyDeviceToHost);  // (synchronizes with the device)
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("fuse_giveups: ") + hipGetErrorString(e));
  return FDNN_OK;
}

int fdnn_debug_layer0(fdnn_model *m, const float *x, int n, uint8_t *u8_out, unsigned long long *recomputed) {
  if (!m || !x || !u8_out || n <= 0) return fail(FDNN_E_ARG, "bad argument");
  DeviceGuard g(m->device);
  const BlobHeader &h = m->hm.hdr;
  fdnn_ctx *c = nullptr;
  int rc = make_ctx(m, n, &c);
  if (rc) return rc;
  unsigned long long before[2] = {0, 0}, after[2] = {0, 0};
  hipError_t e = hipMemcpy(before, m->d_l0_stats, sizeof(before), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x, sizeof(float) * size_t(n) * h.in_dim, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    run_layer0(c, c->d_x, c->stream, nullptr);  // the PRODUCTION instance (no taps): screened path for large batches
    e = hipGetLastError();
  }
  const size_t act_ld = size_t(c->act_ld);
  std::vector<int8_t> tmp(size_t(n) * act_ld);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), c->d_act[0], tmp.size(), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(after, m->d_l0_stats, sizeof(after), hipMemcpyDeviceToHost);
  fdnn_ctx_free(c);
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("layer 0: ") + hipGetErrorString(e));
  for (int f = 0; f < n; ++f)
    for (int i = 0; i < h.hidden; ++i) u8_out[size_t(f) * h.hidden + i] = uint8_t(tmp[size_t(f) * act_ld + i]) ^ 0x80;
  if (recomputed) *recomputed = after[1] - before[1];
  return FDNN_OK;
}

int fdnn_debug_layer0_screen(fdnn_model *m, const float *x, int n, uint8_t *u8_out, float *t_out, float *dd_out, unsigned long long *recomputed) {
  if (!m || !x || !u8_out || !t_out || !dd_out || n <= 0) return fail(FDNN_E_ARG, "bad argument");
  if (!m->d_w0d) return fail(FDNN_E_STATE, "this model's input layer has no int8 screening (input width outside 64..496)");
  DeviceGuard g(m->device);
  const BlobHeader &h = m->hm.hdr;
  fdnn_ctx *c = nullptr;
  int rc = make_ctx(m, n, &c);
  if (rc) return rc;
  const size_t outs = size_t(n) * h.hidden;
  unsigned long long before[2] = {0, 0}, after[2] = {0, 0};
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_l0_dbg_t), outs * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_l0_dbg_dd), outs * sizeof(float));
  if (e == hipSuccess) e = hipMemset(c->d_l0_dbg_t, 0xff, outs * sizeof(float));  // NaN: an output the screening kernel did not visit
  if (e == hipSuccess) e = hipMemset(c->d_l0_dbg_dd, 0xff, outs * sizeof(float));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(before, m->d_l0_stats, sizeof(before), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_x, x, sizeof(float) * size_t(n) * h.in_dim, hipMemcpyHostToDevice, c->stream);
  const int kernel_before = m->l0_kernel;
  if (e == hipSuccess) {
    if (m->l0_kernel == 0) m->l0_kernel = 4;  // the int8 screening whatever the batch size
    run_layer0(c, c->d_x, c->stream, nullptr);
    m->l0_kernel = kernel_before;
    e = hipGetLastError();
  }
  const size_t act_ld = size_t(c->act_ld);
  std::vector<int8_t> tmp(size_t(n) * act_ld);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), c->d_act[0], tmp.size(), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(t_out, c->d_l0_dbg_t, outs * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dd_out, c->d_l0_dbg_dd, outs * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(after, m->d_l0_stats, sizeof(after), hipMemcpyDeviceToHost);
  fdnn_ctx_free(c);
  if (e != hipSuccess) return fail(FDNN_E_DEVICE, std::string("layer 0 (screen debug): ") + hipGetErrorString(e));
  for (int f = 0; f < n; ++f)
    for (int i = 0; i < h.hidden; ++i) u8_out[size_t(f) * h.hidden + i] = uint8_t(tmp[size_t(f) * act_ld + i]) ^ 0x80;
  if (recomputed) *recomputed = after[1] - before[1];
  return FDNN_OK;
}

// ---------------------------------------------------------------- per-kernel timing
int fdnn_profile_begin(fdnn_model *m) {
  if (!m) return fail(FDNN_E_ARG, "null model");
  std::lock_guard<std::mutex> lk(m->mu);
  for (auto &r : m->prof) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  m->prof.clear();
  m->profiling = true;
  return FDNN_OK;
}

int fdnn_profile_end(fdnn_model *m, double *ms, int *launches) {
  if (!m || !ms || !launches) return fail(FDNN_E_ARG, "null argument");
  DeviceGuard g(m->device);
  std::vector<fdnn_model::ProfRec> recs;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->profiling = false;
    recs.swap(m->prof);
  }
  for (int k = 0; k < FDNN_PROF_KINDS; ++k) {
    ms[k] = 0.0;
    launches[k] = 0;
  }
  int rc = FDNN_OK;
  for (auto &r : recs) {
    float t = 0.0f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
    if (e == hipSuccess) {
      ms[r.kind] += t;
      launches[r.kind]++;
    } else {
      rc = fail(FDNN_E_DEVICE, std::string("profile: ") + hipGetErrorString(e));
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  return rc;
}

// ---------------------------------------------------------------- weight blob exchange
int fdnn_model_blob_size(const fdnn_model *m, size_t *bytes) {
  if (!m || !bytes) return fail(FDNN_E_ARG, "null argument");
  *bytes = m->hm.blob.size();
  return FDNN_OK;
}

int fdnn_model_export_blob(const fdnn_model *m, void *d_dst, size_t capacity, void *stream) {
  if (!m || !d_dst) return fail(FDNN_E_ARG, "null argument");
  if (capacity < m->hm.blob.size()) return fail(FDNN_E_ARG, "destination smaller than the blob");
  DeviceGuard g(m->device);
  HIP_TRY(hipMemcpyAsync(d_dst, m->d_blob, m->hm.blob.size(), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return FDNN_OK;
}

int fdnn_model_import_blob(const void *d_src, size_t bytes, int device, fdnn_model **out) {
  if (!d_src || !out) return fail(FDNN_E_ARG, "null argument");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(FDNN_E_DEVICE, "no HIP device available: this library has no CPU path");
  if (device < 0 || device >= count) return fail(FDNN_E_ARG, "device index out of range");
  DeviceGuard g(device);
  std::vector<uint8_t> host(bytes);
  HIP_TRY(hipMemcpy(host.data(), d_src, bytes, hipMemcpyDeviceToHost));
  fdnn_model *m = new fdnn_model();
  std::string msg;
  int rc = fdnn::adopt_blob(std::move(host), &m->hm, &msg);
  if (rc) {
    delete m;
    return fail(rc, msg);
  }
  m->device = device;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_blob), bytes);
  if (e == hipSuccess) e = hipMemcpy(m->d_blob, d_src, bytes, hipMemcpyDeviceToDevice);
  if (e != hipSuccess) {
    if (m->d_blob) hipFree(m->d_blob);
    delete m;
    return fail(FDNN_E_DEVICE, std::string("blob import: ") + hipGetErrorString(e));
  }
  rc = build_l0_image(m);
  if (rc) {
    hipFree(m->d_blob);
    if (m->d_w0t) hipFree(m->d_w0t);
    if (m->d_w0norm) hipFree(m->d_w0norm);
    if (m->d_w0d) hipFree(m->d_w0d);
    if (m->d_w0stat) hipFree(m->d_w0stat);
    if (m->d_lutpair) hipFree(m->d_lutpair);
    if (m->d_l0_stats) hipFree(m->d_l0_stats);
    delete m;
    return rc;
  }
  *out = m;
  return FDNN_OK;
}

// ---------------------------------------------------------------- host-only helpers
int fdnn_host_model_load(const char *path, float cutoff, fdnn_host_model **out) {
  if (!path || !out) return fail(FDNN_E_ARG, "null argument");
  *out = nullptr;
  fdnn_host_model *hm = new fdnn_host_model();
  std::string msg;
  int rc = fdnn::load_host_model(path, cutoff, &hm->hm, &msg);
  if (rc) {
    delete hm;
    return fail(rc, msg);
  }
  *out = hm;
  return FDNN_OK;
}

void fdnn_host_model_free(fdnn_host_model *hm) { delete hm; }
int fdnn_host_model_layers(const fdnn_host_model *hm) { return hm ? hm->hm.hdr.n_affine : -1; }

int fdnn_host_model_layer_in(const fdnn_host_model *hm, int j) {
  if (!hm || j < 0 || j >= hm->hm.hdr.n_affine) return -1;
  return j == 0 ? hm->hm.hdr.in_dim : hm->hm.hdr.q[j - 1].cols;
}

int fdnn_host_model_layer_out(const fdnn_host_model *hm, int j) {
  if (!hm || j < 0 || j >= hm->hm.hdr.n_affine) return -1;
  return j == 0 ? hm->hm.hdr.hidden : hm->hm.hdr.q[j - 1].rows;
}

float fdnn_host_model_multiplier(const fdnn_host_model *hm, int j) {
  if (!hm || j < 1 || j >= hm->hm.hdr.n_affine) return 0.0f;
  return hm->hm.hdr.q[j - 1].mult;
}

int fdnn_host_model_weights_q(const fdnn_host_model *hm, int j, int8_t *out) {
  if (!hm || !out || j < 1 || j >= hm->hm.hdr.n_affine) return fail(FDNN_E_ARG, "bad layer index");
  const QLayerDesc &d = hm->hm.hdr.q[j - 1];
  const int8_t *w = hm->hm.wq(j - 1);
  for (int r = 0; r < d.rows; ++r) std::memcpy(out + size_t(r) * d.cols, w + size_t(r) * d.cols_pad, size_t(d.cols));
  return FDNN_OK;
}

int fdnn_host_model_bias(const fdnn_host_model *hm, int j, float *out) {
  if (!hm || !out || j < 0 || j >= hm->hm.hdr.n_affine) return fail(FDNN_E_ARG, "bad layer index");
  if (j == 0)
    std::memcpy(out, hm->hm.b0(), sizeof(float) * size_t(hm->hm.hdr.hidden));
  else
    std::memcpy(out, hm->hm.bias(j - 1), sizeof(float) * size_t(hm->hm.hdr.q[j - 1].rows));
  return FDNN_OK;
}

int fdnn_host_model_wsum128(const fdnn_host_model *hm, int j, int32_t *out) {
  if (!hm || !out || j < 1 || j >= hm->hm.hdr.n_affine) return fail(FDNN_E_ARG, "bad layer index");
  std::memcpy(out, hm->hm.wsum(j - 1), sizeof(int32_t) * size_t(hm->hm.hdr.q[j - 1].rows));
  return FDNN_OK;
}

long long fdnn_host_model_risky_pairs(const fdnn_host_model *hm, int j) {
  if (!hm || j < 1 || j >= hm->hm.hdr.n_affine) return -1;
  return hm->hm.hdr.q[j - 1].n_fix;
}

size_t fdnn_host_model_blob_size(const fdnn_host_model *hm) { return hm ? hm->hm.blob.size() : 0; }

int fdnn_host_model_blob(const fdnn_host_model *hm, void *out, size_t capacity) {
  if (!hm || !out) return fail(FDNN_E_ARG, "null argument");
  if (capacity < hm->hm.blob.size()) return fail(FDNN_E_ARG, "destination smaller than the blob");
  std::memcpy(out, hm->hm.blob.data(), hm->hm.blob.size());
  return FDNN_OK;
}

int fdnn_host_blob_check(const void *bytes, size_t size, int *input_dim, int *hidden_dim, int *output_dim,
                         int *n_affine) {
  if (!bytes) return fail(FDNN_E_ARG, "null argument");
  std::vector<uint8_t> copy(static_cast<const uint8_t *>(bytes), static_cast<const uint8_t *>(bytes) + size);
  fdnn::HostModel hm;
  std::string msg;
  int rc = fdnn::adopt_blob(std::move(copy), &hm, &msg);
  if (rc) return fail(rc, msg);
  if (input_dim) *input_dim = hm.hdr.in_dim;
  if (hidden_dim) *hidden_dim = hm.hdr.hidden;
  if (output_dim) *output_dim = hm.hdr.out_dim;
  if (n_affine) *n_affine = hm.hdr.n_affine;
  return FDNN_OK;
}

int fdnn_host_sigmoid_lut(uint8_t *out) {
  if (!out) return fail(FDNN_E_ARG, "null argument");
  fdnn::build_sigmoid_lut(out);
  return FDNN_OK;
}

int fdnn_host_quantize(const float *w, int rows, int cols, float cutoff, int8_t *out, float *multiplier) {
  if (!w || !out || !multiplier || rows <= 0 || cols <= 0) return fail(FDNN_E_ARG, "bad argument");
  fdnn::quantize_layer(w, rows, cols, cutoff, out, multiplier);
  return FDNN_OK;
}

}  // extern "C"
