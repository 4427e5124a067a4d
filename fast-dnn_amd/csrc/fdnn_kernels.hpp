// fdnn_kernels.hpp -- launch interface of the gfx950 kernels (fdnn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

// Run-time switches.  The DEPLOYMENT switches (FDNN_BATCHER, FDNN_DEVICES, FDNN_FUSE_NORM,
// FDNN_GROUP_*, FDNN_JNI_KEEP_MB, FDNN_CHAIN, FDNN_CHUNK_FRAMES) are read with std::getenv.  Everything else -- tile-shape
// overrides, thresholds and kernel choices that the sweeps under tools/ turn -- exists only in measurement builds
// (-DFDNN_ABLATION: tools/build_variant.sh): in the shipped library FDNN_TUNE_ENV is a null pointer, the defaults are
// constants, and the compile-time ablation branches (FDNN_GEMM_DEBUG, FDNN_L0S_DEBUG, FDNN_CHAIN_CLK, ...) cannot be set.
#ifdef FDNN_ABLATION
#include <cstdlib>
#define FDNN_TUNE_ENV(name) std::getenv(name)
#else
#define FDNN_TUNE_ENV(name) static_cast<const char *>(nullptr)
#if (defined(FDNN_GEMM_DEBUG) && FDNN_GEMM_DEBUG) || (defined(FDNN_L0S_DEBUG) && FDNN_L0S_DEBUG) || (defined(FDNN_L0_DEBUG) && FDNN_L0_DEBUG) || \
    (defined(FDNN_CHAIN_CLK) && FDNN_CHAIN_CLK) || defined(FDNN_L0S_CLK) || defined(FDNN_GEMM_PAD)
#error "ablation / clock builds need -DFDNN_ABLATION (tools/build_variant.sh)"
#endif
#endif

namespace fdnn {

constexpr int kMaxFrameTile = 320;  // largest GEMM frame tile; scratch rows carry this much slack
constexpr int kPartialNodes = 64;   // nodes covered by one soft-max partial sum

// Layer 0: shift/scale + fp32 affine + bias + sigmoid LUT -> s8 activations.
struct L0Params {
  const float *x;      // [n][D] caller frames (never written)
  const float *shift;  // [D]
  const float *scale;  // [D]
  const float *w;      // [H][D]
  const float *bias;   // [H]
  const uint8_t *lut;  // [kLutExt] XOR 0x80
  int8_t *act_out;     // [n_rows][act_ld]
  int act_ld;          // row stride of act_out (H padded to the GEMM k-step)
  float *tap_lin;      // [n][H] or null
  int n, D, H;
  int n_rows;          // rows of act_out to fill (>= n, the next GEMM's padded frame count)
  int fma;             // 0: mul then add (canonical), 1: fused
  int kernel;          // canonical flavour: 0 pick by batch size, 1 chain-pass kernel, 2 64 x 64-tile kernel
  // canonical flavour: chain-major operand images  img[(c + 2) % 4][j][col] = src[col][4j + c]
  float *xt;           // [4][j_pad][n_ld] frame image, scratch, rewritten by every launch
  const float *wt;     // [4][j_pad][h_ld] weight image (launch_l0_weight_image at model load)
  float *park;         // [n_ld/128][h_ld/128][64 KB] scratch: l2 + l3 of a tile while l0, l1 run (128-node tiles only, else null)
  int j_pad, jc;       // D/4 rounded up to the chunk depth jc = l0_chunk_rows(D)
  int n_ld, h_ld;      // image row lengths: frame capacity and H, both rounded up to 128
  // canonical flavour, screened path (large batches, no taps): the FUSED chains on the fp32 MFMA for every output,
  // a rigorous bound on |fused - unfused| per output, and the exact unfused chains for the few outputs whose
  // table index the difference could change (fdnn_l0.hip: "screened").  All null/0 = path not available.
  const float *wnorm;  // [H] upper bound of ||w_n||_2 (model load)
  uint32_t *scr_count; // [tiles] flagged outputs per 128 x 128 tile; zero between launches
  uint16_t *scr_list;  // [tiles][kL0ScreenCap] tile-local indices frame_row * 128 + node_col
  unsigned long long *scr_stats;  // [2] running totals: outputs screened, outputs recomputed (may be null)
  // round 4, the screening on the INT8 matrix pipe (fdnn_l0s.hip): 24-bit integer images of both operands as three
  // int8 digit planes in MFMA fragment order + per-row constants of the error bound.  All null = path not available.
  int8_t *xd;          // [chunks][3][n_ld / 32][1024] frame planes, scratch, rewritten by every launch
  float *xstat;        // [3][n_ld]: 2^16 / c_f, ||x_f||_2 (rounded up), a_f
  const int8_t *wd;    // [chunks][3][h_ld / 32][1024] node planes (model load)
  const float *wstat;  // [3][h_ld]: 2^8 / c_n, ||w_n||_2 (rounded up), 2^8 b_n
  uint2 *glist;        // [glist_cap] {frame, node} of every flagged output of the launch (tiles append with one atomic each)
  uint32_t *glist_count;  // [2]: entries appended; tiles that overflowed into the whole-tile path.  Zeroed by the pre-pass
  int glist_cap;
  const uint32_t *luthalf;  // [kLut2Size (+pad)] the half-step table, 4 bytes per entry: table byte + the 'same byte across the boundary' gate
  // parity tests only (fdnn_debug_layer0_screen; null in production): what the int8 screening saw per output, [n][H] each --
  // t~ = 100 lin~ and the half-width Dd of the interval it vouches for; |100 lin_ref - t~| <= Dd is the bound's claim
  float *dbg_t;
  float *dbg_dd;
};
constexpr int kL0ScreenCap = 4096;  // listed outputs per tile (25 %); a tile that overflows is recomputed whole
void launch_l0(const L0Params &p, hipStream_t s);
// fdnn_l0s.hip: is the int8 screening available for this layer shape; bytes of one operand's digit planes; the node half
// (host code, model load); pre-pass + matrix kernel (launch_l0 follows with the exact recomputation of the flagged outputs)
bool l0_split_ok(int D, int H);
size_t l0_split_plane_bytes(int D, int rows_ld);
void launch_l0_split(const L0Params &p, hipStream_t s);
void l0_split_build_weights(const float *w, const float *wnorm, const uint8_t *lut2, int H, int D, int h_ld, std::vector<int8_t> *planes,
                            std::vector<float> *stat, std::vector<uint32_t> *half);
int l0_chunk_rows(int D);
int l0_chain_node_tile();  // 64 (default: no park scratch needed) or 128 (L0Params::park must be allocated)
void launch_l0_weight_image(const float *w, float *wt, int H, int D, int j_pad, int h_ld, hipStream_t s);

// Frame tile (32/64/128/160/256/320) the int8 GEMM should use for `n` frames of a layer
// with rows_pad padded nodes; n_pad = n rounded up to it.
int qgemm_frame_tile(int rows_pad, int n);
int qgemm_node_tile(int rows_pad, int n, bool output);  // 256, or 128 where the 128 x 128 shape (frame tile 128) is the better one
int qgemm_debug_flags();

// int8 layer: C[node][frame] = sum_k W[node][k] * (A[frame][k] + 128), then the
// layer's epilogue.  W rows are padded to 256, A rows to the frame tile.
struct QGemmParams {
  const int8_t *w;        // [rows_pad][K]  (K = padded input width, pad columns are zero)
  const int8_t *a;        // [n_pad][K]  s8 = u8 - 128 (pad columns: anything)
  const float *bias;      // [rows_pad]
  const int32_t *wsum;    // [rows_pad]  128*sum_k w
  const int32_t *fix_grp; // [rows_pad/64 + 1] entry range of every 64-node group
  const void *fix_ent;    // FixEntry[n_fix] sorted by node; null when the layer has no risky pairs
  const uint8_t *lut;     // [kLutExt]
  const uint8_t *lut2;    // [kLut2Size] half-step table (fast epilogue)
  int rows, rows_pad, K, n, n_pad;
  int ldw, lda;           // row strides (bytes) of w and a: K plus the anti-channel-conflict skew
  int frame_tile;         // 32 / 64 / 128 / 160 / 256 / 320, n_pad is a multiple of it
  int small;              // 1: the small-batch kernel (fdnn_small.hip; frame_tile = 32)
  int node_tile;          // 256 (default) or 128: with frame_tile 128, the 128 x 128 four-wave shape for mid-size batches
  int debug;              // timing experiments only (FDNN_GEMM_DEBUG): 1 no staging, 2 no MFMA, 4 no LDS reads
  float coef, rcp_coef;
  int fastdiv;
  // hidden-layer output
  int8_t *act_out;        // [n_pad][act_ld]
  int act_ld;
  // output-layer output
  float *out;             // [n][rows] un-normalised exp, row stride = rows
  float *partial;         // [rows_pad/kPartialNodes][partial_ld]
  int partial_ld;
  const int8_t *mask;     // [n][rows] or null (lazy contract)
  const uint64_t *mask_bits;  // [n][mask_wpr] the same mask, one bit per node (launch_mask_pack), or null: the large-batch
  int mask_wpr;               // production instances read one 64-bit word per frame row and 64-node group instead of 64 bytes
  // fused soft-max (large-batch dense output instance): where the probabilities go, the per-tile row sums S
  // [n_pad / frame_tile][rows_pad / 256][frame_tile], the per-frame-tile {arrived, left} counters (zero between launches)
  // and the per-tile "gave up waiting" flags (the frame tile's last workgroup scales such blocks); all null = the unfused path
  float *final;
  float *fuse_s;
  uint32_t *fuse_cnt;
  uint32_t *fuse_flag;
  int fuse_stagger;       // fused soft-max: start delay of the first round's frame tile j = (j % 8) * this many 512-cycle naps (see qgemm_kernel)
  unsigned long long *fuse_giveups;  // per model: tiles that had to be scaled after the fact (a workgroup gave up waiting); null = not counted
  unsigned long long *fuse_fault;    // per model, HOST memory: raised with the first give-up -- the host then stops fusing for this model (run_output); null = nobody listens
  // accumulator probe of the PRODUCTION output instances (parity tests only; null otherwise): the int32 accumulators of
  // every probe_stride-th frame, [ceil(n / probe_stride)][rows] -- a wave-uniform branch in front of the epilogue
  int32_t *acc_probe;
  int probe_stride;
  // taps (null in production)
  int32_t *tap_acc;       // [n][rows]
  float *tap_logit;       // [n][rows]
};
// Small batches (fdnn_small.hip): does this layer (K bytes per row, exact-division epilogue validated) have the
// small-batch shape, and should a batch of n frames take it?
bool qgemm_small_ok(int K, int fastdiv);
bool qgemm_small_pick(int rows_pad, int K, int n, int fastdiv, bool output);
void launch_qgemm_small_hidden(const QGemmParams &p, hipStream_t s);
void launch_qgemm_small_output(const QGemmParams &p, hipStream_t s);
// Fused soft-max available for this launch?  (dense production call, 8-wave shapes, the row sums of all node tiles fit the
// epilogue's LDS)
bool qgemm_fused_ok(const QGemmParams &p);
void launch_qgemm_hidden(const QGemmParams &p, hipStream_t s);
void launch_qgemm_output(const QGemmParams &p, hipStream_t s);

// An int8 hidden layer of a large batch with the two waves of every SIMD in different roles (fdnn_pp.hip): one computes
// (fragment reads + MFMAs only) while its partner stages that tile's operands and runs the epilogue of the tile it computed
// before.  256-node x 320-frame tiles (n_pad a multiple of qpp_frame_tile()), K = 2048, validated 3-operation division.
bool qpp_ok(int rows_pad, int K, int n, bool fastdiv, bool has_fix);
void qpp_set_mode(int mode, int min_frames);  // fdnn_debug_set_pp
int qpp_frame_tile();
void launch_qpp_hidden(const QGemmParams &p, hipStream_t s);
// fdnn_ppo.hip: the same role split for the OUTPUT layer of a large dense batch, soft-max scaled inside the kernel (the
// caller holds the device's chain of fused launches: run_output)
bool qppo_ok(int rows, int rows_pad, int K, int n, bool fastdiv, bool has_fix);
void qppo_set_mode(int mode);  // -1 default (FDNN_PPO in the environment, else by size), 0 never, 1 whenever the shape allows
int qppo_frame_tile();
void launch_qppo_output(const QGemmParams &p, hipStream_t s);

// The int8 HIDDEN layers of a pass in one persistent launch (fdnn_chain.hip): tasks (layer, frame tile, node tile) drawn
// from per-XCD queues, a task waits only for its own frame tile's node tiles of the layer before.  All hidden layers of
// a net have the same shape (README.md:10), so one set of sizes serves every layer.
constexpr int kMaxChainLayers = 8;
struct QChainLayer {
  const int8_t *w;         // [rows_pad][ldw]
  const float *bias;       // [rows_pad]
  const int32_t *wsum;     // [rows_pad]
  const int32_t *fix_grp;  // as QGemmParams
  const void *fix_ent;     // null: no risky pairs in this layer
  float coef, rcp_coef;
};
struct QChainParams {
  QChainLayer layer[kMaxChainLayers];
  int n_layers;
  int8_t *act[2];          // layer i reads act[i & 1] and writes act[(i & 1) ^ 1]; rows [n_pad][lda]
  const uint8_t *lut2;     // half-step sigmoid table
  int rows, rows_pad, K, ldw, lda, n, n_pad;
  int frame_tile;          // 320 or 256
  uint32_t *ctl;           // [16]: queue heads [0..7], workgroups that have left [8]; zero between launches
  uint32_t *done;          // [n_pad / frame_tile][n_layers] node tiles finished; zero between launches
  unsigned long long *faults;  // waits that ran into their bound (never in a healthy setup): counted, fdnn_model_chain_faults
  unsigned long long *fault_flag;  // host-visible word of the launching context, raised with the count (null: not raised)
  long long *clk;          // measurement builds (FDNN_CHAIN_CLK): [8 + clk_cap * 10] phase clocks per task, else null
  int clk_cap;
};
bool qchain_ok(int rows_pad, int K, int n, int n_layers);
void qchain_set_mode(int mode, int min_frames);  // fdnn_debug_set_chain
int qchain_frame_tile(int rows_pad, int n);
void launch_qchain(const QChainParams &p, hipStream_t s);

// bits[f][w] bit b = mask[f][64 w + b] != 0  (words per row = ceil(rows / 64); bits past the row are zero).  The lazy
// contract's byte masks (80 MB for 10 000 frames x 8000 nodes) are read once here, at HBM speed, instead of inside the
// output GEMM's epilogue.
void launch_mask_pack(const int8_t *mask, uint64_t *bits, int n, int rows, hipStream_t s);
void launch_mask_unpack(const uint64_t *bits, int8_t *mask, int n, int rows, hipStream_t s);
// comp[f][0] = row f's inactive value, comp[f][1 + r] = probability of its r-th active node (rows of `stride` floats)
void launch_lazy_compact(const float *out, const uint64_t *bits, float *comp, int n, int rows, int stride, hipStream_t s);

// dst[f][:] = out[f][:] / sum_t partial[t][f]   (dst == out: in place; dst may be host-mapped)
// background: a small fixed grid walking the rows (server loop: runs under the next batch's layer 0)
void launch_normalize(float *out, float *dst, const float *partial, int n, int partial_ld, int rows, int n_partial, hipStream_t s,
                      bool background = false);

// Exhaustive check of the 3-op division against IEEE division for every int32
// accumulator in [-2^26, 2^26]; *d_mismatch receives the count.
void launch_fastdiv_check(float coef, float rcp, unsigned long long *d_mismatch, hipStream_t s);

// s8 <-> u8 view of activations for taps / read-back.
void launch_xor80(const int8_t *in, uint8_t *out, size_t count, hipStream_t s);

}  // namespace fdnn
