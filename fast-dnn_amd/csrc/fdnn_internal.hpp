// fdnn_internal.hpp -- what the runtime's translation units share: the model / context objects
// behind the opaque C-ABI handles and the enqueue helpers over them (fdnn_runtime.cpp), used by
// the multi-stream server loop (fdnn_server.cpp).  Not installed; the boundary is include/fdnn.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fdnn.h"
#include "fdnn_kernels.hpp"
#include "fdnn_model.hpp"

namespace fdnn {

int fail(int code, const std::string &msg);  // sets fdnn_last_error() of the calling thread, returns code

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return ::fdnn::fail(FDNN_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

struct DeviceGuard {
  int prev = 0;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() {
    if (ok) hipSetDevice(prev);
  }
};

}  // namespace fdnn

struct fdnn_model {
  fdnn::HostModel hm;  // header + host copy of the blob (kept: export, host queries)
  int device = 0;
  uint8_t *d_blob = nullptr;
  float *d_w0t = nullptr;  // layer-0 weights as a chain-major image [4][l0_j_pad][l0_h_ld] (fdnn_l0.hip)
  float *d_w0norm = nullptr;  // [H] ||w_n||_2 rounded up: the node half of the screened layer-0 path's bound
  int8_t *d_w0d = nullptr;    // int8 screening (fdnn_l0s.hip): the layer-0 weights as three int8 digit planes, MFMA fragment order
  float *d_w0stat = nullptr;  // [3][l0_h_ld]: 2^8 / c_n, ||w_n||_2, 2^8 b_n
  uint32_t *d_lutpair = nullptr;  // the sigmoid table as (round-down, round-up) byte pairs
  // the device side of "is somebody else scoring on this GPU?".  A fused soft-max workgroup that
  // sits out its bounded wait (tens of milliseconds: its frame tile's siblings were kept off the chip) raises this
  // host-mapped word; from the next call on the model runs the scale-pass path, for good, and says so once.
  unsigned long long *h_fuse_fault = nullptr, *d_fuse_fault = nullptr;
  bool fuse_fault_said = false;
  unsigned long long *d_l0_stats = nullptr;  // [4] device counters: [1] layer-0 outputs recomputed exactly, [2] fused soft-max tiles that gave up waiting
  int l0_jc = 0, l0_j_pad = 0, l0_h_ld = 0;
  int l0_fma = 0;
  int l0_kernel = 0;  // fdnn_debug_set_l0_kernel
  int l0_list_cap = 0;  // fdnn_debug_set_l0_list_cap: > 0 caps the flagged-output list of contexts created afterwards (tests)
  std::mutex mu;
  std::vector<fdnn_ctx *> pool;  // idle contexts owned by the model (fdnn_calculate*)
  struct fdnn_server *batcher = nullptr;  // fdnn_model_enable_batcher: fdnn_calculate goes through it
  struct fdnn_group *group = nullptr;     // fdnn_group_attach: this model leads a device group, fdnn_calculate shards over it
  // per-kernel HIP-event timing (fdnn_profile_begin/end); off in production
  bool profiling = false;
  struct ProfRec {
    int kind;
    hipEvent_t a, b;
  };
  std::vector<ProfRec> prof;
};

struct fdnn_ctx {
  fdnn_model *m = nullptr;
  int n = 0;              // frames in use
  int cap = 0;            // frames the scratch was allocated for (padded)
  int act_ld = 0;
  hipStream_t stream = nullptr;   // own stream for the host-pointer entry points
  hipEvent_t done = nullptr;      // last enqueued work (pool hand-over between streams, ctx_enter/ctx_leave)
  hipStream_t done_stream = nullptr;  // the stream `done` was last recorded on
  bool done_valid = false;
  bool done_pending = false;      // work went to done_stream -- a stream that outlives this context's use of it (see
                                  // stream_is_durable) -- after the last record: recorded when ANOTHER stream asks (ctx_enter)
  hipStream_t durable[3] = {nullptr, nullptr, nullptr};  // streams of the context's owner (the scoring loop's), besides `stream`
  float *d_x = nullptr;           // [n][D]
  float *d_xt = nullptr;          // [4][l0_j_pad][xt_ld] layer-0 frame image (shifted, scaled, chain-major)
  int xt_ld = 0;
  float *d_l0park = nullptr;      // [xt_ld][l0_h_ld] partial chain sums parked by the layer-0 kernel
  uint32_t *d_scr_count = nullptr;  // [frame tiles x node tiles] flagged outputs per tile (kept zero between launches)
  uint16_t *d_scr_list = nullptr;   // [tiles][kL0ScreenCap]
  int8_t *d_xd = nullptr;           // int8 screening: the frames' digit planes [chunks][3][xt_ld / 32][1024]
  float *d_xstat = nullptr;         // [3][xt_ld] row constants written by the pre-pass
  uint2 *d_glist = nullptr;         // int8 screening: {frame, node} of the launch's flagged outputs
  uint32_t *d_glist_count = nullptr;  // [2] entries appended, tiles in the whole-tile path
  int glist_cap = 0;
  int8_t *d_act[2] = {nullptr, nullptr};  // [n_pad][act_ld] ping/pong, s8 = u8-128
  float *d_out = nullptr;         // [n][O]
  float *d_partial = nullptr;     // [rows_pad/64][n_pad]
  int8_t *d_mask = nullptr;       // [n][O]
  float *d_fuse_s = nullptr;        // fused soft-max: per-tile row sums [n_pad / tile][rows_pad / 256][tile] floats
  uint32_t *d_fuse_cnt = nullptr;   // {arrived part 0 .., left at [7]} per frame tile; zero between launches
  size_t fuse_cnt_bytes = 0, fuse_flag_bytes = 0;
  uint32_t *d_fuse_flag = nullptr;  // per tile: parts a workgroup left unscaled ("gave up waiting"); zero between launches
  uint32_t *d_chain_ctl = nullptr;   // chained hidden layers (fdnn_chain.hip): queue heads [0..7], workgroups that left [8]
  uint32_t *d_chain_done = nullptr;  // [frame tiles][layers of the chain] node tiles finished; zero between launches
  long long *d_chain_clk = nullptr;  // measurement builds only: per-task phase clocks
  // A chained launch whose wait ran into its bound has computed on rows that may not have been written: it raises this
  // host-visible word (fine-grained pinned memory, plain store from the kernel).  run_hidden looks at it before every
  // launch -- counters re-zeroed, the context never chains again -- and the calls that synchronise anyway re-run the pass.
  unsigned long long *h_chain_fault = nullptr, *d_chain_fault = nullptr;
  bool chain_broken = false;
  size_t chain_done_bytes = 0;
  int chain_clk_cap = 0;
  float *d_l0_dbg_t = nullptr, *d_l0_dbg_dd = nullptr;  // fdnn_debug_layer0_screen only: the int8 screening's t~ and Dd per output
  uint64_t *d_mask_bits = nullptr;  // [n][ceil(O/64)] the batched lazy call's mask as bits (launch_mask_pack)
  bool mask_bits_packed = false;    // run_output has packed the current call's byte masks into d_mask_bits
  float *d_comp = nullptr;          // host lazy batches: compacted result rows (allocated on first use)
  size_t comp_floats = 0;
  int last = -1;                  // d_act index holding the last hidden layer, -1 = not computed
  bool pooled = false;
  bool no_fuse = false;           // this call must not use the fused soft-max
  bool l0_chain_only = false;     // scoring loop, large batches: the soft-max scale of the previous batch runs under this
                                  // batch's layer 0, which must then be the vector-pipe chain kernel (the matrix-pipe
                                  // screened path fills the register file: nothing can run beside it)
  // per-frame lazy calls (the JNI contract): host-mapped pinned staging for kPinFrames masks and
  // result rows -- the output kernel reads the mask and the scale kernel writes the probabilities
  // straight through these, so a call is two launches and one stream sync, no copy commands
  int8_t *h_mask_pin = nullptr, *d_mask_pin = nullptr;
  float *h_out_pin = nullptr, *d_out_pin = nullptr;
};
constexpr int kPinFrames = 8;


namespace fdnn {

struct Taps {
  float *l0_lin = nullptr;
  uint8_t *u8_acts = nullptr;   // device [n_hidden][n][H]
  int32_t *acc_hid = nullptr;   // device [n_hidden-1][n][H]
  int32_t *acc_out = nullptr;   // device [n][O]
  float *logits = nullptr;      // device [n][O]
  int32_t *acc_probe = nullptr; // device [ceil(n / probe_stride)][O]: accumulators of the PRODUCTION output instance (no tap kernels)
  int probe_stride = 1;
};

// lean: a scoring-loop slot -- no frame / result / mask buffers and no per-frame pinned staging of its own (device
// submissions bring their buffers; the host path allocates what it needs in alloc_host_side)
int make_ctx(fdnn_model *m, int n, fdnn_ctx **out, bool lean = false);
void destroy_ctx(fdnn_ctx *c);
// CalculateUntilLastHiddenLayer (dnn.cc:402-424) enqueued on s.
int run_hidden(fdnn_ctx *c, const float *d_x, hipStream_t s, const Taps *taps);
// CalculateOutput / LazyOutputActivations (dnn.cc:428-454, :355-392) over frames [first, first+count):
// the output GEMM (exp(z) rows + partial sums) on s, then the soft-max scale -- on s, or, when
// `tail` is given, on that stream behind an event (`gemm_done`) recorded on s, so that the
// HBM-bound scale pass can run under the next batch's layer 0.
int run_output(fdnn_ctx *c, int first, int count, const int8_t *d_masks, float *d_out, hipStream_t s, const Taps *taps,
               float *d_final = nullptr, hipStream_t tail = nullptr, hipEvent_t gemm_done = nullptr, const uint64_t *d_bits = nullptr);
// (d_bits: the masks of the same frames as BITS, [count][ceil(O / 64)], bit b of word w = node 64 w + b; then d_masks may be
// null -- large batches read the words as they are, small ones unpack them into the context's byte mask first)
// Will run_output scale the soft-max inside the output kernel for such a call (dense, large batch)?  Then there is no
// scale pass to hide under the next batch's layer 0.
bool output_will_fuse(fdnn_ctx *c, int count, const int8_t *d_masks);
// Host half of a compacted lazy return: expands `count` compacted rows sitting in the tail of out[count][O] (fdnn_runtime.cpp).
void lazy_expand_rows(float *out, int count, size_t O, size_t stride, const uint64_t *bits);
void lazy_expand_rows_from(float *out, const float *comp, int count, size_t O, size_t stride, const uint64_t *bits);  // rows in a buffer of their own
// A pass over a very large batch runs as chunks (frames are independent: a chunk is a batch of its own, and the scratch
// context only has to hold one).  kRoundFrames = 32 frame tiles of 320 = one workgroup per CU in the 2048-wide hidden
// layers; a chunk is two rounds.  Measured, 125 000 frames (the 8-GPU shard of BASELINE configs[4]), fused soft-max:
// 12.09 M frames/s as one batch (4 GB of result rows, 512 MB of activations per layer: the address-translation and Infinity
// caches stop covering the working set), 12.87 M in chunks of one round, 13.14 M in chunks of two (tools/chunk_bench.py).
// Returns (offset, count) pairs: chunks of kChunkFrames, then what is left as one more batch.  (Up to round 4 a small tail
// past a whole round was always split off as a batch of its own; with chained hidden layers that is unnecessary: frame_chunks.)
constexpr int kRoundFrames = 10240;
constexpr int kChunkFrames = 2 * kRoundFrames;
constexpr int kChunkTailSplit = 2048;
// With a model: a chunk whose hidden layers will not run as one chained launch (hidden_layers_chain) gives up to
// kChunkTailSplit frames past a whole round away as a batch of their own.  Without: as assume_chained says.
std::vector<std::pair<int, int>> frame_chunks(int n, const fdnn_model *m = nullptr, bool assume_chained = true);
bool hidden_layers_chain(const fdnn_model *m, int n);
// Ordering between the streams a context is used on.  An event record costs 3-4 us of queue time behind the kernel it
// follows, so a context whose work went to a stream that is certain to exist later -- its own, its owner's, or the null
// stream -- only NOTES that (done_pending); the record is made when a different stream next needs the order (ctx_enter),
// or never.  Work on a caller's stream, which the caller may destroy, is recorded at once as before.
hipError_t ctx_enter(fdnn_ctx *c, hipStream_t s);
void ctx_leave(fdnn_ctx *c, hipStream_t s);
bool stream_is_durable(const fdnn_ctx *c, hipStream_t s);
void ctx_wait_host(fdnn_ctx *c);  // host-side wait for everything ctx_leave covered
// The per-device chain of fused soft-max launches (run_output) notes its last launch the same way: a stream that is about
// to be destroyed must be retired from it first.
void fuse_chain_retire_stream(int device, hipStream_t s);

}  // namespace fdnn
