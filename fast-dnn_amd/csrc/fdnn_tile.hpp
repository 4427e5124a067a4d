// fdnn_tile.hpp -- the int8 GEMM tile shared by the per-layer kernel (fdnn_gemm.hip) and the chained hidden-layer
// kernel (fdnn_chain.hip): tile geometry / LDS map, the chunk swizzle and the fragment read.
#pragma once
#include "fdnn_device.hpp"

namespace fdnn {

// WM waves along the nodes (64 each): 4 = the 256-node tile of every large shape; 2 / 1 = the 128- / 64-node tiles of
// the smallest batches, where one workgroup's operand stream is what a launch waits for (see launch_qgemm).
template <int NF, int WN, int BK, int STAGES, int WM = 4>
struct GemmCfg {
  static constexpr int G_BM = 64 * WM;               // nodes per workgroup tile
  static constexpr int NW = WM * WN;                 // waves
  static constexpr int THREADS = 64 * NW;
  static constexpr int FT = 32 * NF * WN;            // frames per workgroup tile
  static constexpr int RPI = 1024 / BK;              // rows per 1-KiB wave instruction
  static constexpr int LPR = BK / 16;                // lanes per row
  static constexpr int W_BYTES = G_BM * BK;
  static constexpr int A_BYTES = FT * BK;
  static constexpr int STAGE = W_BYTES + A_BYTES;
  static constexpr int W_SLABS = G_BM / RPI;
  static constexpr int A_SLABS = FT / RPI;
  static constexpr int MIN_LOADS = W_SLABS / NW + A_SLABS / NW;  // fewest loads any wave issues per stage
  // Behind the ring: the sigmoid table (3 KiB window) and this tile's 256 biases, both
  // LDS-DMA'd before the first stage so the epilogue starts without a load phase.  After the k-loop the ring is reused for the s8
  // output tile FT x (256+16) bytes (hidden layers) / the per-wave e tiles (output layer).
  static constexpr int EPI = 8192 + FT * (G_BM + 16);
  static constexpr int FIX_OFF = STAGE * STAGES;
  static constexpr int AUX_OFF = FIX_OFF;  // table at +0, biases at +3072
  static constexpr int RING = AUX_OFF + 4096;
  static constexpr int LDS = RING > EPI ? RING : EPI;
  static_assert(EPI <= FIX_OFF, "the epilogue tile must not reach the table/biases");
  static_assert(W_SLABS % NW == 0, "weight slabs must split evenly over the waves");
  static_assert(LDS <= 160 * 1024, "LDS ring exceeds the CU");
};

template <int BK>
__device__ __forceinline__ int swz(int row) {
  return BK == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7);
}

template <int BK>
__device__ __forceinline__ v4i read_frag(const char *tile, int row, int chunk) {
  return *reinterpret_cast<const v4i *>(tile + row * BK + ((chunk ^ swz<BK>(row)) << 4));
}

}  // namespace fdnn
