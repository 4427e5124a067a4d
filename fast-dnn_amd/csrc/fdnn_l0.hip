// fdnn_l0.hip -- layer 0 on gfx950: shift/scale + fp32 affine + bias + sigmoid table.
//
//   ApplyShiftAndScale + InputActivations + AddBias + QuantizedSigmoid
//     (dnn.cc:175-192, :219-286)
//
// The reference accumulates FOUR partial sums per (frame, node) -- SSE lane l takes
// k = l mod 4 -- each a sequential chain over k, combined (l0+l1)+(l2+l3)
// (dnn.cc:233-238, :168-172).  The u8 output goes through round(100*x) and a table,
// so one ulp matters: both kernels below reproduce those chains exactly.
//
//   l0_valu_kernel  canonical numerics (reference built -O2 -msse4 -ffp-contract=off):
//                   multiply and add are separate roundings -> VALU v_pk_mul_f32 +
//                   v_pk_add_f32 (packed by hand), two lane-ops per MAC; bound by the
//                   fp32 vector instruction rate (83 % of what a bare mul/add loop reaches).
//   l0_mfma_kernel  the reference as its own Makefile builds it on an FMA host
//                   (-march=native contracts mul+add): each chain is an fmaf chain,
//                   which is bit-for-bit what one block of v_mfma_f32_32x32x1_2b_f32
//                   computes, so the four chains become four accumulator blocks fed
//                   k = c, c+4, c+8, ...  (validated bit-for-bit against the FMA build).
#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {
namespace {

// Workgroup b runs on XCD b % 8.  All node tiles of a frame tile go to ONE XCD (frame tile =
// xcd + 8 * (slot / node_tiles)), so a frame tile's input rows are pulled into one L2 instead of
// all eight (PMC: 150 MB of HBM+MALL reads per launch for a 17 MB input with the plain 2-D grid).
__device__ __forceinline__ bool l0_tile_of_block(int node_tiles, int frame_tiles, int &bx, int &by) {
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  bx = slot % node_tiles;
  by = xcd + 8 * (slot / node_tiles);
  return by < frame_tiles;
}
inline unsigned l0_grid(int node_tiles, int frame_tiles) { return static_cast<unsigned>(node_tiles) * ((frame_tiles + 7) / 8) * 8; }

// ---------------------------------------------------------------- VALU, order-faithful, unfused
// 256 threads, tile (16*TI frames) x 64 nodes, thread (tx, ty) owns frames ty*TI+i and
// nodes tx+16j (j<4): 4*TI outputs x 4 partial sums.  LDS rows are padded to BK+4
// floats, which makes the per-node ds_read_b128 (rows tx+16j) conflict free.
template <int TI, int BK, bool FMA, bool TAP>
__global__ __launch_bounds__(256, TI <= 4 ? 4 : 2) void l0_valu_kernel(L0Params p) {
  constexpr int TF = 16 * TI, TN = 64, LD = BK + 4;
  constexpr int XQ = TF * BK / 4 / 256;  // float4 of x each thread stages per chunk
  constexpr int WQ = TN * BK / 4 / 256;
  static_assert(XQ >= 1 && WQ >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) float smem[2 * TF * LD + 2 * TN * LD + (kLutExt + 15) / 4 + 4];
  float *xs = smem;
  float *ws = smem + 2 * TF * LD;
  uint8_t *lut = reinterpret_cast<uint8_t *>(smem + 2 * TF * LD + 2 * TN * LD);
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  int bx, by;
  if (!l0_tile_of_block((p.H + TN - 1) / TN, (p.n_rows + TF - 1) / TF, bx, by)) return;
  const int f0 = by * TF, n0 = bx * TN;
  // one 16-byte load per thread (the blob pads the table to kLutExt + 15 bytes, 256-aligned)
  if (tid < (kLutExt + 15) / 16) reinterpret_cast<uint4 *>(lut)[tid] = reinterpret_cast<const uint4 *>(p.lut)[tid];

  constexpr int QPR = BK / 4;  // float4 per row
  float acc[TI][4][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int l = 0; l < 4; ++l) acc[i][j][l] = 0.0f;

  const int nchunk = (p.D + BK - 1) / BK;
  float4 xr[XQ], wr[WQ];
  auto gload = [&](int c) {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int item = tid + q * 256, row = item / QPR, k = c * BK + (item % QPR) * 4;
      const int f = f0 + row;
      xr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < p.D && f < p.n) {  // D is a multiple of 4
        const float4 v = *reinterpret_cast<const float4 *>(p.x + static_cast<size_t>(f) * p.D + k);
        const float4 sh = *reinterpret_cast<const float4 *>(p.shift + k);
        const float4 sc = *reinterpret_cast<const float4 *>(p.scale + k);
        // ApplyShiftAndScale: add, then multiply (dnn.cc:184-187)
        xr[q].x = (v.x + sh.x) * sc.x;
        xr[q].y = (v.y + sh.y) * sc.y;
        xr[q].z = (v.z + sh.z) * sc.z;
        xr[q].w = (v.w + sh.w) * sc.w;
      }
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int item = tid + q * 256, row = item / QPR, k = c * BK + (item % QPR) * 4;
      const int node = n0 + row;
      wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < p.D && node < p.H) wr[q] = *reinterpret_cast<const float4 *>(p.w + static_cast<size_t>(node) * p.D + k);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int item = tid + q * 256;
      *reinterpret_cast<float4 *>(xs + (buf * TF + item / QPR) * LD + (item % QPR) * 4) = xr[q];
    }
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int item = tid + q * 256;
      *reinterpret_cast<float4 *>(ws + (buf * TN + item / QPR) * LD + (item % QPR) * 4) = wr[q];
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(c + 1);
    const float *xb = xs + buf * TF * LD;
    const float *wb = ws + buf * TN * LD;
#pragma unroll 2
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      float4 xv[TI], wv[4];
#pragma unroll
      for (int i = 0; i < TI; ++i) xv[i] = *reinterpret_cast<const float4 *>(xb + (ty * TI + i) * LD + k4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const float4 *>(wb + (tx + 16 * j) * LD + k4 * 4);
      if (!FMA) {
        // InputActivations, canonical flavour: four lane partial sums over k mod 4, multiply and
        // add rounded separately (dnn.cc:233-238).  Packed fp32 by hand: chains (0,1) and (2,3)
        // sit in even-aligned register pairs, as do the halves of the ds_read_b128 operands ->
        // v_pk_mul_f32 + v_pk_add_f32 without a single shuffle; the eight products of a frame
        // first, so that no add waits on the multiply right before it (0.403 -> 0.391 ms; the
        // SLP vectorizer's own packing of the scalar code is slower: keep -fno-slp-vectorize)
        typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const v2f x01 = {xv[i].x, xv[i].y}, x23 = {xv[i].z, xv[i].w};
          v2f pr[4][2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pr[j][0] = x01 * v2f{wv[j].x, wv[j].y};
            pr[j][1] = x23 * v2f{wv[j].z, wv[j].w};
          }
          asm volatile("" : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[1][0]), "+v"(pr[1][1]), "+v"(pr[2][0]), "+v"(pr[2][1]),
                       "+v"(pr[3][0]), "+v"(pr[3][1]));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const v2f a01 = v2f{acc[i][j][0], acc[i][j][1]} + pr[j][0], a23 = v2f{acc[i][j][2], acc[i][j][3]} + pr[j][1];
            acc[i][j][0] = a01.x; acc[i][j][1] = a01.y; acc[i][j][2] = a23.x; acc[i][j][3] = a23.y;
          }
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // fused flavour on the VALU (FDNN_L0_FMA_VALU=1; production uses l0_mfma_kernel)
          acc[i][j][0] = fmaf(xv[i].x, wv[j].x, acc[i][j][0]);
          acc[i][j][1] = fmaf(xv[i].y, wv[j].y, acc[i][j][1]);
          acc[i][j][2] = fmaf(xv[i].z, wv[j].z, acc[i][j][2]);
          acc[i][j][3] = fmaf(xv[i].w, wv[j].w, acc[i][j][3]);
        }
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int f = f0 + ty * TI + i;
    if (f >= p.n_rows) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int node = n0 + tx + 16 * j;
      if (node < p.H) {
        // horizontalSum: (l0+l1)+(l2+l3) (dnn.cc:168-172), then AddBias
        const float s = (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        const float lin = s + p.bias[node];
        if (TAP && f < p.n) p.tap_lin[static_cast<size_t>(f) * p.H + node] = lin;
        p.act_out[static_cast<size_t>(f) * p.act_ld + node] = static_cast<int8_t>(lut[lut_index(lin)]);
      }
    }
  }
}

template <int TI, int BK>
void launch_valu(const L0Params &p, hipStream_t s) {
  dim3 grid(l0_grid((p.H + 63) / 64, (p.n_rows + 16 * TI - 1) / (16 * TI)));
  if (p.tap_lin) {
    if (p.fma)
      hipLaunchKernelGGL((l0_valu_kernel<TI, BK, true, true>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((l0_valu_kernel<TI, BK, false, true>), grid, dim3(256), 0, s, p);
  } else {
    if (p.fma)
      hipLaunchKernelGGL((l0_valu_kernel<TI, BK, true, false>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((l0_valu_kernel<TI, BK, false, false>), grid, dim3(256), 0, s, p);
  }
}

// ---------------------------------------------------------------- MFMA, fused flavour
// v_mfma_f32_32x32x1_2b_f32 performs, per output element and per block, exactly one
// acc = fma(a, b, acc).  Lanes 0-31 feed block 0 and lanes 32-63 block 1, so one
// instruction advances two of the four k-mod-4 chains by one k each:
//   instr P: block 0 <- k = 4g+0, block 1 <- k = 4g+1
//   instr Q: block 0 <- k = 4g+2, block 1 <- k = 4g+3
// and the same lane holds element (i, j) of all four chains -> the (l0+l1)+(l2+l3)
// combine is register-local.  LDS rows keep every k quad in the order [k0 k2 k1 k3] so
// that lane half h fetches its P and Q operands (k = h, h+2) with one ds_read_b64.
//
// WFR x 2 waves (frames x nodes); block tile 32*WFR frames x 128 nodes, wave tile 32 frames
// x 64 nodes = 2 node subtiles x {P, Q} x 32 accumulators (2 waves per SIMD).  With WFR = 2
// two 256-thread blocks share a CU, so one block's prologue (cold loads) and epilogue
// (table + stores) overlap the other's MFMA stream.
typedef float v32f __attribute__((ext_vector_type(32)));

#ifndef FDNN_L0_DEBUG
#define FDNN_L0_DEBUG 0  // kernel-ablation timing builds only (tools/build_variant.sh): 1 no MFMA, 2 no loads in the k-loop
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// buffer descriptor held in SGPRs (a descriptor the compiler cannot prove wave-uniform
// turns every load into a readfirstlane waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float *base, int bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint64_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a)));
  const uint64_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32)));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#endif

template <int BK, int WFR>
struct L0MfmaCfg {
  static constexpr int TF = 32 * WFR, TN = 128, LD = BK + 2, QPR = BK / 4, THREADS = 128 * WFR;
  static constexpr int RPQ = THREADS / QPR;  // rows between a thread's staging items
  static constexpr int XPT = TF / RPQ, WPT = TN / RPQ;  // float4 a thread stages per chunk
  static constexpr int TS = TN + 16;         // byte tile row stride (16-B aligned, skewed)
  static constexpr int kStage = (TF + TN) * LD;
  static constexpr int LDS = (2 * kStage + (kLutExt + 15) / 4 + 4) * 4;
  static_assert(TF * TS <= 2 * kStage * 4, "epilogue tile must fit in the staging ring");
  static_assert(XPT >= 1 && XPT * RPQ == TF && WPT * RPQ == TN, "chunk does not divide over the threads");
};

template <int BK, int WFR, bool TAP>
__global__ __launch_bounds__(128 * WFR, 2) void l0_mfma_kernel(L0Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = L0MfmaCfg<BK, WFR>;
  constexpr int TF = Cfg::TF, TN = Cfg::TN, LD = Cfg::LD, QPR = Cfg::QPR, TS = Cfg::TS, kStage = Cfg::kStage;
  constexpr int THREADS = Cfg::THREADS, RPQ = Cfg::RPQ, XPT = Cfg::XPT, WPT = Cfg::WPT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint8_t *lut = reinterpret_cast<uint8_t *>(smem + 2 * kStage);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int wf = wave % WFR, wn = wave / WFR;
  int bx, by;
  if (!l0_tile_of_block((p.H + TN - 1) / TN, (p.n_rows + TF - 1) / TF, bx, by)) return;
  const int f0 = by * TF, n0 = bx * TN;
  if (tid < (kLutExt + 15) / 16) reinterpret_cast<uint4 *>(lut)[tid] = reinterpret_cast<const uint4 *>(p.lut)[tid];

  v32f acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 32; ++r) acc[s][c][r] = 0.0f;

  const int nchunk = (p.D + BK - 1) / BK;
  // staged raw: the shift/scale arithmetic is done when the registers are written to LDS,
  // one chunk later, so that no s_waitcnt sits between the loads and the MFMA block
  float4 sx[XPT], sw[WPT], sh, sc;  // item q is row srow + q * RPQ, same k quad for every q
  const int srow = tid / QPR, skq = (tid % QPR) * 4;
  // buffer loads: rows past the end of the tile's window and k >= D (poisoned offset) read
  // as zero without a branch; the windows start at the tile so 32-bit offsets suffice
  const int x_rows = max(0, min(TF, p.n - f0)), w_rows = max(0, min(TN, p.H - n0));
  const auto x_rsrc = uniform_rsrc(p.x + static_cast<size_t>(f0) * p.D, x_rows * p.D * 4);
  const auto w_rsrc = uniform_rsrc(p.w + static_cast<size_t>(n0) * p.D, w_rows * p.D * 4);
  const auto sh_rsrc = uniform_rsrc(p.shift, p.D * 4);
  const auto sc_rsrc = uniform_rsrc(p.scale, p.D * 4);
  const int row_off = (srow * p.D + skq) * 4;
  // rows past the frame count read x = 0, so (0 + shift) * scale is finite garbage that stays
  // inside its own (never returned) row
  auto gload = [&](int c) {
    const int k = c * BK + skq;  // D is a multiple of 4
    const int off = k < p.D ? row_off + c * BK * 4 : 0x7ffffff0;
#pragma unroll
    for (int q = 0; q < XPT; ++q)
      sx[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off + q * RPQ * p.D * 4, 0, 0));
#pragma unroll
    for (int q = 0; q < WPT; ++q)
      sw[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off + q * RPQ * p.D * 4, 0, 0));
    sh = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sh_rsrc, k * 4, 0, 0));
    sc = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc, k * 4, 0, 0));
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      float *dst = smem + buf * kStage + (srow + q * RPQ) * LD + skq;
      // ApplyShiftAndScale: add, then multiply (dnn.cc:184-187); quad order [k0 k2 k1 k3]
      *reinterpret_cast<float2 *>(dst) = make_float2((sx[q].x + sh.x) * sc.x, (sx[q].z + sh.z) * sc.z);
      *reinterpret_cast<float2 *>(dst + 2) = make_float2((sx[q].y + sh.y) * sc.y, (sx[q].w + sh.w) * sc.w);
    }
#pragma unroll
    for (int q = 0; q < WPT; ++q) {
      float *dst = smem + buf * kStage + (TF + srow + q * RPQ) * LD + skq;
      *reinterpret_cast<float2 *>(dst) = make_float2(sw[q].x, sw[q].z);
      *reinterpret_cast<float2 *>(dst + 2) = make_float2(sw[q].y, sw[q].w);
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (!(FDNN_L0_DEBUG & 2) && c + 1 < nchunk) gload(c + 1);
    const float *xb = smem + buf * kStage + (wf * 32 + l32) * LD + 2 * h;
    const float *wb = smem + buf * kStage + (TF + wn * 64 + l32) * LD + 2 * h;
    // a zero-filled tail quad adds fma(0, 0, acc) = acc
    float2 a[QPR], b0[QPR], b1[QPR];
#pragma unroll
    for (int g = 0; g < QPR; ++g) {
      a[g] = *reinterpret_cast<const float2 *>(xb + 4 * g);
      b0[g] = *reinterpret_cast<const float2 *>(wb + 4 * g);
      b1[g] = *reinterpret_cast<const float2 *>(wb + 32 * LD + 4 * g);
    }
#pragma unroll
    for (int g = 0; g < QPR; ++g) {
      if (FDNN_L0_DEBUG & 1) {  // ablation: operands consumed, no matrix work
        acc[0][0][0] += a[g].x + b0[g].x + b1[g].y;
        acc[1][1][1] += a[g].y + b0[g].y + b1[g].x;
        continue;
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[g].x, b0[g].x, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[g].x, b1[g].x, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[g].y, b0[g].y, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[g].y, b1[g].y, acc[1][1], 0, 0, 0);
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: combine the chains, bias, table; bytes go through LDS so that each frame
  // row leaves as one 128-B segment
  uint8_t *tile = reinterpret_cast<uint8_t *>(smem);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int col = wn * 64 + s * 32 + l32;
    const int node = n0 + col;
    const float bias = node < p.H ? p.bias[node] : 0.0f;
    uint8_t act[16];  // table gathers first, tile writes after: the two alias in LDS as far as the compiler knows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
      // horizontalSum: (l0+l1)+(l2+l3) (dnn.cc:168-172), then AddBias
      const float sum = (acc[s][0][r] + acc[s][0][16 + r]) + (acc[s][1][r] + acc[s][1][16 + r]);
      const float lin = sum + bias;
      if (TAP && f0 + row < p.n && node < p.H) p.tap_lin[static_cast<size_t>(f0 + row) * p.H + node] = lin;
      act[r] = lut[lut_index(lin)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * TS + col] = act[r];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TF * 8 / THREADS; ++q) {
    const int item = tid + q * THREADS, row = item >> 3, c16 = (item & 7) * 16;
    const int f = f0 + row;
    if (f < p.n_rows && n0 + c16 < p.H)  // H is a multiple of 16
      *reinterpret_cast<uint4 *>(p.act_out + static_cast<size_t>(f) * p.act_ld + n0 + c16) =
          *reinterpret_cast<const uint4 *>(tile + row * TS + c16);
  }
#endif
}

template <int BK, int WFR>
void launch_mfma(const L0Params &p, hipStream_t s) {
  using Cfg = L0MfmaCfg<BK, WFR>;
  auto k_prod = l0_mfma_kernel<BK, WFR, false>;
  auto k_tap = l0_mfma_kernel<BK, WFR, true>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tap), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set = true;
  }
  dim3 grid(l0_grid((p.H + Cfg::TN - 1) / Cfg::TN, (p.n_rows + Cfg::TF - 1) / Cfg::TF));
  hipLaunchKernelGGL(p.tap_lin ? k_tap : k_prod, grid, dim3(Cfg::THREADS), Cfg::LDS, s, p);
}

}  // namespace

void launch_l0(const L0Params &p, hipStream_t s) {
  static const bool fma_on_valu = std::getenv("FDNN_L0_FMA_VALU") != nullptr;
  if (p.fma && !fma_on_valu) {
    static const int bk = [] {
      const char *e = std::getenv("FDNN_L0_MFMA_BK");
      return e ? std::atoi(e) : 324;
    }();
    switch (bk) {  // BK * 10 + WFR
      case 164: launch_mfma<16, 4>(p, s); break;
      case 644: launch_mfma<64, 4>(p, s); break;
      case 162: launch_mfma<16, 2>(p, s); break;
      case 642: launch_mfma<64, 2>(p, s); break;
      case 322: launch_mfma<32, 2>(p, s); break;
      default: launch_mfma<32, 4>(p, s); break;
    }
    return;
  }
  static const int variant = [] {
    const char *e = std::getenv("FDNN_L0_VARIANT");
    return e ? std::atoi(e) : 0;
  }();
  switch (variant) {
    case 1: launch_valu<4, 32>(p, s); break;
    case 2: launch_valu<8, 16>(p, s); break;
    case 3: launch_valu<8, 32>(p, s); break;
    default: launch_valu<4, 16>(p, s); break;
  }
}

}  // namespace fdnn
