// fdnn_kernels.hpp -- launch interface of the gfx950 kernels (fdnn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fdnn {

constexpr int kFrameTile = 128;  // frames per GEMM workgroup tile; scratch rows are padded to this
constexpr int kNodeTile = 128;   // nodes per GEMM workgroup tile
constexpr int kPartialNodes = 64;  // nodes covered by one soft-max partial sum

// Layer 0: shift/scale + fp32 affine + bias + sigmoid LUT -> s8 activations.
struct L0Params {
  const float *x;      // [n][D] caller frames (never written)
  const float *shift;  // [D]
  const float *scale;  // [D]
  const float *w;      // [H][D]
  const float *bias;   // [H]
  const uint8_t *lut;  // [kLutExt] XOR 0x80
  int8_t *act_out;     // [n_pad][act_ld]
  int act_ld;          // row stride of act_out (H padded to the GEMM k-step)
  float *tap_lin;      // [n][H] or null
  int n, D, H;
  int fma;             // 0: mul then add (canonical), 1: fused
};
void launch_l0(const L0Params &p, hipStream_t s);

// int8 layer: C[node][frame] = sum_k W[node][k] * (A[frame][k] + 128), then the
// layer's epilogue.  W rows are padded to kNodeTile, A rows to kFrameTile.
struct QGemmParams {
  const int8_t *w;        // [rows_pad][K]  (K = padded input width, pad columns are zero)
  const int8_t *a;        // [n_pad][K]  s8 = u8 - 128 (pad columns: anything)
  const float *bias;      // [rows_pad]
  const int32_t *wsum;    // [rows_pad]  128*sum_k w
  const int32_t *slot;    // [rows_pad]  or null when the layer has no risky pairs
  const int32_t *corr;    // [n_slots][n_pad] saturation corrections, or null
  const uint8_t *lut;     // [kLutExt]
  int rows, rows_pad, K, n, n_pad;
  float coef, rcp_coef;
  int fastdiv;
  // hidden-layer output
  int8_t *act_out;        // [n_pad][act_ld]
  int act_ld;
  // output-layer output
  float *out;             // [n][rows] un-normalised exp, row stride = rows
  float *partial;         // [rows_pad/kPartialNodes][n_pad]
  const int8_t *mask;     // [n][rows] or null (lazy contract)
  // taps (null in production)
  int32_t *tap_acc;       // [n][rows]
  float *tap_logit;       // [n][rows]
};
void launch_qgemm_hidden(const QGemmParams &p, hipStream_t s);
void launch_qgemm_output(const QGemmParams &p, hipStream_t s);

// pmaddubsw saturation corrections for one layer: corr[slot][f] =
// sum over the slot's risky pairs of sat16(p) - p.
struct FixParams {
  const int8_t *a;         // [n_pad][K] s8 activations feeding the layer
  const int32_t *fix_ptr;  // [n_slots+1]
  const void *fix_ent;     // FixEntry[n_fix]
  int32_t *corr;           // [n_slots][n_pad]
  int n_slots, n, n_pad, K;
};
void launch_fix(const FixParams &p, hipStream_t s);

// out[f][:] /= sum_t partial[t][f]
void launch_normalize(float *out, const float *partial, int n, int n_pad, int rows, int n_partial, hipStream_t s);

// Exhaustive check of the 3-op division against IEEE division for every int32
// accumulator in [-2^26, 2^26]; *d_mismatch receives the count.
void launch_fastdiv_check(float coef, float rcp, unsigned long long *d_mismatch, hipStream_t s);

// s8 <-> u8 view of activations for taps / read-back.
void launch_xor80(const int8_t *in, uint8_t *out, size_t count, hipStream_t s);

}  // namespace fdnn
