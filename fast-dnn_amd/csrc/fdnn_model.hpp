// fdnn_model.hpp -- host-side model: .bin loader, quantizer, packed weight blob.
//
// Replaces, for the MI355X path, the reference's load-time half:
//   FloatDnn / BinaryLoader      (src/cpp/float_dnn.cc:18-69, :166-212)
//   QuantizedSimdLayer ctor      (src/cpp/dnn.cc:460-509)
//   FloatSimdLayer ctor          (src/cpp/dnn.cc:123-144)
//   QuantizedSigmoid table       (src/cpp/dnn.cc:100-115)
// and adds what only the GPU kernels need (per-node 128*sum(w) offsets for the
// u8->s8 trick, the list of weight pairs whose pmaddubsw sum can saturate).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace fdnn {

constexpr int kLutSize = 1280;      // dnn.h:26
constexpr int kLutHalf = 640;       // dnn.h:27
constexpr int kLutExt = 1281;       // device table: index clamp(k,-640,640)+640
constexpr int kLut2Half = 1281;     // half-step table: index clamp(trunc(2t),-1281,1281)+1281
constexpr int kLut2Size = 2 * kLut2Half + 1;
constexpr int kMaxQLayers = 30;
constexpr uint32_t kBlobMagic = 0x4e4e4446u;  // "FDNN"
constexpr uint32_t kBlobVersion = 6;
constexpr int kRowPad = 256;        // int8 weight rows padded to the GEMM node tile
constexpr int kColPad = 128;        // int8 weight columns (and activation rows) padded to the GEMM k-step
constexpr int kRowSkew = 0;         // extra bytes on int8 row strides (0: measured slower with a 64-byte skew)
// No cap on the saturating pairs a layer may carry: the reference runs any weights, so does this
// path (a layer of 2^24 x 2^15 / 2 pairs still indexes with int32).  Each listed pair costs a
// gather per frame tile, so a net whose weights sit near +-127 everywhere runs correctly but at a
// fraction of the MFMA rate.

// One (node, adjacent weight pair) entry whose pmaddubsw pair sum can leave
// int16 for some activation (dnn.cc:337-340): k is the even column.
struct FixEntry {
  uint16_t k;
  int8_t w0;
  int8_t w1;
  int32_t node;
};

struct QLayerDesc {       // lives in the blob header, read by host and device
  uint64_t off_w;         // int8 [rows_pad][cols_pad], pad rows / pad columns are zero
  uint64_t off_bias;      // f32 [rows_pad]
  uint64_t off_wsum;      // i32 [rows_pad]  128 * sum_k w[row][k]
  uint64_t off_fix_grp;   // i32 [rows_pad/64 + 1]  entry range of each 64-node group
  uint64_t off_fix_ent;   // FixEntry [n_fix], sorted by node
  int32_t rows, rows_pad, cols;
  int32_t n_fix;          // risky (node, pair) entries in this layer
  int32_t lin_bounded;    // |sum/coef + bias| * 200 provably < 2^31 and all biases finite
  float mult;             // QuantizedSimdLayer::multiplier_
  float coef;             // mult * 255.0f  (dnn.cc:298-299)
  float rcp_coef;         // RN(1/coef) for the 3-op exact division
  int32_t fastdiv_ok;     // set after the exhaustive device check at load
  int32_t cols_pad;       // row stride of w: cols padded to the GEMM k-step (kColPad) + kRowSkew
};

struct BlobHeader {
  uint32_t magic, version;
  uint64_t total_bytes;
  int32_t n_affine;       // affine layers incl. fp32 layer 0
  int32_t in_dim_file;    // as stored in the .bin
  int32_t in_dim;         // padded to x4 (float_dnn.cc:32-33)
  int32_t hidden;         // H (all hidden layers)
  int32_t out_dim;        // O
  int32_t n_q;            // quantized layers = n_affine - 1
  float cutoff;
  int32_t pad_;
  uint64_t off_w0;        // f32 [H][in_dim]
  uint64_t off_b0;        // f32 [H]
  uint64_t off_shift;     // f32 [in_dim]
  uint64_t off_scale;     // f32 [in_dim]
  uint64_t off_lut;       // u8  [kLutExt] (+pad), already XOR 0x80 (s8 activations)
  uint64_t off_lut2;      // u8  [kLut2Size] (+pad) half-step table, XOR 0x80
  QLayerDesc q[kMaxQLayers];
};

// Host image of a loaded + quantized net.
struct HostModel {
  BlobHeader hdr{};
  std::vector<uint8_t> blob;  // hdr copy at offset 0, then the sections
  // convenience views into blob
  // NB: rows of wq() are cols_pad apart
  const int8_t *wq(int qi) const { return reinterpret_cast<const int8_t *>(blob.data() + hdr.q[qi].off_w); }
  const float *bias(int qi) const { return reinterpret_cast<const float *>(blob.data() + hdr.q[qi].off_bias); }
  const int32_t *wsum(int qi) const { return reinterpret_cast<const int32_t *>(blob.data() + hdr.q[qi].off_wsum); }
  const float *w0() const { return reinterpret_cast<const float *>(blob.data() + hdr.off_w0); }
  const float *b0() const { return reinterpret_cast<const float *>(blob.data() + hdr.off_b0); }
};

// QuantizedSigmoid table (1280 bytes), dnn.cc:100-115.
void build_sigmoid_lut(uint8_t *out1280);

// QuantizedSimdLayer quantizer, dnn.cc:460-509 (row pointers not needed: w is
// rows x cols contiguous).
void quantize_layer(const float *w, int rows, int cols, float cutoff, int8_t *out, float *multiplier);

// Loads + quantizes + packs.  Returns 0 or a negative fdnn_status; msg gets the reason.
int load_host_model(const std::string &path, float cutoff, HostModel *out, std::string *msg);

// Validates a blob received from another rank and rebuilds the header view.
int adopt_blob(std::vector<uint8_t> &&bytes, HostModel *out, std::string *msg);

}  // namespace fdnn
