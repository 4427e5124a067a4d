// fdnn_l0.hip -- layer 0 on gfx950: shift/scale + fp32 affine + bias + sigmoid table.
//
//   ApplyShiftAndScale + InputActivations + AddBias + QuantizedSigmoid
//     (dnn.cc:175-192, :219-286)
//
// The reference accumulates FOUR partial sums per (frame, node) -- SSE lane l takes
// k = l mod 4 -- each a sequential chain over k, combined (l0+l1)+(l2+l3)
// (dnn.cc:233-238, :168-172).  The u8 output goes through round(100*x) and a table,
// so one ulp matters: all kernels below reproduce those chains exactly.
//
//   l0_chain_kernel canonical numerics (reference built -O2 -msse4 -ffp-contract=off): multiply
//                   and add are separate roundings -> v_pk_mul_f32 + v_pk_add_f32, two lane-ops
//                   per MAC; one chain per pass over chain-major operand images, 8 x 8 outputs
//                   per thread (81 % of what a bare packed mul/add loop reaches).  Batches
//                   that fill the chip with 128 x 128 tiles.
//   l0_valu_kernel  same numerics, 64 x 64 tiles, all four chains at once: smaller batches.
//   l0_mfma_kernel  the reference as its own Makefile builds it on an FMA host
//                   (-march=native contracts mul+add): each chain is an fmaf chain,
//                   which is bit-for-bit what one block of v_mfma_f32_32x32x1_2b_f32
//                   computes, so the four chains become four accumulator blocks fed
//                   k = c, c+4, c+8, ...  (validated bit-for-bit against the FMA build).
#include <atomic>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {
namespace {

// Workgroup b runs on XCD b % 8 (mostly: a locality hint only).  All node tiles of a frame tile go to ONE XCD (frame tile =
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

#ifndef FDNN_L0_SCHED_MASK
#define FDNN_L0_SCHED_MASK 0x3f4  // what may cross the products | adds fence of the chain kernel: memory and scalar ops
#endif
#ifndef FDNN_L0_SCREEN_EVERY
#define FDNN_L0_SCREEN_EVERY 2  // screened path: the fused partial sums are sampled every this many 8-step chunks
#endif
#ifndef FDNN_L0_TN64_WAVES
#define FDNN_L0_TN64_WAVES 3  // waves per SIMD the 64-node chain kernel is compiled for (134 VGPRs as written)
#endif
#ifndef FDNN_L0_DEBUG
#define FDNN_L0_DEBUG 0  // kernel-ablation timing builds only (tools/build_variant.sh): 1 no MFMA, 2 no loads in the k-loop;
                         // chain kernel: 4 operands from registers (no LDS reads), 8 no LDS-DMA in the loop
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

// ---------------------------------------------------------------- small batches: the whole K extent at once
// One utterance (100 frames) is 88 M MACs -- 3 us of vector work spread over the chip -- and l0_valu_kernel spends
// 20 us on it: seven k-chunks, each a global -> register -> LDS round trip behind a barrier with one wave per SIMD to
// hide it.  Here a workgroup owns 32 frames x 32 nodes and loads ALL of its operand rows up front (LDS-DMA, 2 x 32 rows
// x D floats = 110 KB at D = 432), waits once, applies ApplyShiftAndScale (dnn.cc:175-192: add, then multiply) to its
// frame rows in LDS, and then runs the four k-mod-4 chains of InputActivations (dnn.cc:219-247) straight through:
// the same multiplies, adds and combine as l0_valu_kernel, bit for bit.  LDS rows are padded to an ODD number of
// 16-byte chunks, so the 16 lanes of a ds_read_b128 group (16 different node rows, same k) hit 16 different slots.
// 256 threads: thread (tx, ty) owns nodes tx, tx + 16 and frames ty, ty + 16.
template <bool TAP>
__global__ __launch_bounds__(512) void l0_small_kernel(L0Params p, int sc, unsigned sc_magic, unsigned ch_magic) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem_l0s[];
  constexpr int TF = 32, TN = 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // plain tile order (node tiles fastest): at these sizes the grid is about one workgroup per CU and every XCD
  // must get its share; the frame rows (a few hundred KB in all) reach every L2 anyway
  const int node_tiles = (p.H + TN - 1) / TN;
  const int bx = blockIdx.x % node_tiles, by = blockIdx.x / node_tiles;
  const int f0 = by * TF, n0 = bx * TN;
#ifdef FDNN_L0S_CLK
  long long tc[6];
  tc[0] = __builtin_readcyclecounter();
#define L0S_TS(i) tc[i] = __builtin_readcyclecounter()
#else
#define L0S_TS(i)
#endif
  const int ch = p.D >> 2;                    // 16-byte chunks per operand row
  const int tile_chunks = 32 * sc;            // per operand tile, pad chunks included
  const int n_ld = (tile_chunks + 63) >> 6;   // 1-KiB LDS-DMA pieces per operand tile
  char *xs = smem_l0s;                        // [32][sc] chunks: frames
  char *ws = xs + n_ld * 1024;                // [32][sc] chunks: nodes
  char *ss = ws + n_ld * 1024;                // shift [D] | scale [D]
  uint8_t *lut = reinterpret_cast<uint8_t *>(ss + 2 * ((p.D * 4 + 1023) & ~1023));
  const int row_bytes = p.D * 4;
  // rows past the batch / past the layer read as zeros: the descriptors end there
  const __amdgpu_buffer_rsrc_t rsrc_x = uniform_rsrc(p.x + static_cast<size_t>(f0) * p.D, max(0, min(TF, p.n - f0)) * row_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = uniform_rsrc(p.w + static_cast<size_t>(n0) * p.D, max(0, min(TN, p.H - n0)) * row_bytes);
  // all eight waves issue (an LDS-DMA piece costs its wave 100-300 cycles of issue time; 110 pieces over four waves
  // were 9.4 k cycles of a workgroup's 37 k); c / sc by multiplication (magic checked on the host for every c used)
  for (int i = wave; i < 2 * n_ld; i += 8) {
    const bool is_w = i >= n_ld;
    const int c = (is_w ? i - n_ld : i) * 64 + lane;
    const int row = static_cast<int>(__umulhi(static_cast<unsigned>(c), sc_magic)), col = c - row * sc;
    const int off = (col < ch && row < 32) ? row * row_bytes + col * 16 : 0x7ffffff0;
    if (is_w)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(ws + (i - n_ld) * 1024), 16, off, 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, FDNN_LDS_PTR(xs + i * 1024), 16, off, 0, 0, 0);
  }
  {  // shift | scale | table: a few more pieces, spread over the waves
    const int sp = (row_bytes + 1023) >> 10;
    const __amdgpu_buffer_rsrc_t rsrc_sh = uniform_rsrc(p.shift, row_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_sc = uniform_rsrc(p.scale, row_bytes);
    for (int i = 7 - wave; i < 2 * sp; i += 8) {
      if (i < sp)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_sh, FDNN_LDS_PTR(ss + i * 1024), 16, lane * 16, i * 1024, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_sc, FDNN_LDS_PTR(ss + i * 1024), 16, lane * 16, (i - sp) * 1024, 0, 0);
    }
    // (the blob pads the table to kLutExt + 15 bytes; the last 16-byte piece must be inside the descriptor as a whole)
    const __amdgpu_buffer_rsrc_t rsrc_lut =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.lut), 0, (kLutExt + 15) & ~15, 0x00020000);
    if (wave >= 6) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lut, FDNN_LDS_PTR(lut + (wave - 6) * 1024), 16, lane * 16, (wave - 6) * 1024, 0, 0);
  }
  // this thread's two biases, requested before the wait (first use is the epilogue)
  const int tx = tid & 15, ty = (tid >> 4) & 15;
  float bias2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias2[j] = (tid < 256 && n0 + tx + 16 * j < p.H) ? p.bias[n0 + tx + 16 * j] : 0.0f;
  L0S_TS(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  L0S_TS(2);
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  {  // ApplyShiftAndScale on the frame tile, in place (all 512 threads)
    const char *scs = ss + ((row_bytes + 1023) & ~1023);
    for (int c = tid; c < 32 * ch; c += 512) {
      const int row = static_cast<int>(__umulhi(static_cast<unsigned>(c), ch_magic)), col = c - row * ch;
      v4f *px = reinterpret_cast<v4f *>(xs + (row * sc + col) * 16);
      const v4f sh = *reinterpret_cast<const v4f *>(ss + col * 16), scl = *reinterpret_cast<const v4f *>(scs + col * 16);
      v4f v = *px;
      v.x = (v.x + sh.x) * scl.x;
      v.y = (v.y + sh.y) * scl.y;
      v.z = (v.z + sh.z) * scl.z;
      v.w = (v.w + sh.w) * scl.w;
      // a frame past the batch stays all-zero (l0_valu_kernel stages zeros there): its outputs are padding rows
      if (f0 + row < p.n) *px = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (tid >= 256) return;  // the helper waves are done: one wave per SIMD runs the chains
  L0S_TS(3);
  float acc[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int l = 0; l < 4; ++l) acc[i][j][l] = 0.0f;
  const char *xr0 = xs + (ty * sc) * 16, *xr1 = xs + ((ty + 16) * sc) * 16;
  const char *wr0 = ws + (tx * sc) * 16, *wr1 = ws + ((tx + 16) * sc) * 16;
  // InputActivations, canonical flavour: four lane partial sums over k mod 4, multiply and add rounded
  // separately (dnn.cc:233-238); chains (0,1) and (2,3) ride v_pk_mul_f32 / v_pk_add_f32.  Operand reads run
  // three steps ahead of the arithmetic (one wave per SIMD: nothing else hides the LDS latency); reads past the
  // last step land in the next row / the next LDS region and are never used.
  auto rd = [&](v4f (&xv)[2], v4f (&wv)[2], int k4) {
    xv[0] = *reinterpret_cast<const v4f *>(xr0 + k4 * 16);
    xv[1] = *reinterpret_cast<const v4f *>(xr1 + k4 * 16);
    wv[0] = *reinterpret_cast<const v4f *>(wr0 + k4 * 16);
    wv[1] = *reinterpret_cast<const v4f *>(wr1 + k4 * 16);
  };
  auto mac = [&](const v4f (&xv)[2], const v4f (&wv)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const v2f x01 = {xv[i].x, xv[i].y}, x23 = {xv[i].z, xv[i].w};
      v2f pr[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        pr[j][0] = x01 * v2f{wv[j].x, wv[j].y};
        pr[j][1] = x23 * v2f{wv[j].z, wv[j].w};
      }
      asm volatile("" : "+v"(pr[0][0]), "+v"(pr[0][1]), "+v"(pr[1][0]), "+v"(pr[1][1]));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const v2f a01 = v2f{acc[i][j][0], acc[i][j][1]} + pr[j][0], a23 = v2f{acc[i][j][2], acc[i][j][3]} + pr[j][1];
        acc[i][j][0] = a01.x; acc[i][j][1] = a01.y; acc[i][j][2] = a23.x; acc[i][j][3] = a23.y;
      }
    }
  };
  v4f x0[2], w0[2], x1[2], w1[2], x2[2], w2[2], x3[2], w3[2];
  rd(x0, w0, 0);
  rd(x1, w1, 1);
  rd(x2, w2, 2);
  int k4 = 0;
  for (; k4 + 3 < ch; k4 += 4) {
    rd(x3, w3, k4 + 3);
    mac(x0, w0);
    rd(x0, w0, k4 + 4);
    mac(x1, w1);
    rd(x1, w1, k4 + 5);
    mac(x2, w2);
    rd(x2, w2, k4 + 6);
    mac(x3, w3);
  }
  if (k4 < ch) mac(x0, w0);
  if (k4 + 1 < ch) mac(x1, w1);
  if (k4 + 2 < ch) mac(x2, w2);
  L0S_TS(4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = f0 + ty + 16 * i;
    if (f >= p.n_rows) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int node = n0 + tx + 16 * j;
      if (node < p.H) {
        // horizontalSum: (l0+l1)+(l2+l3) (dnn.cc:168-172), then AddBias
        const float s = (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        const float lin = s + bias2[j];
        if (TAP && f < p.n) p.tap_lin[static_cast<size_t>(f) * p.H + node] = lin;
        p.act_out[static_cast<size_t>(f) * p.act_ld + node] = static_cast<int8_t>(lut[lut_index(lin)]);
      }
    }
  }
#ifdef FDNN_L0S_CLK
  L0S_TS(5);
  if (tid == 0 && (blockIdx.x % 61) == 0)
    printf("l0s blk %d: issue %lld  land %lld  scale %lld  loop %lld  epi %lld  total %lld\n", blockIdx.x, tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2],
           tc[4] - tc[3], tc[5] - tc[4], tc[5] - tc[0]);
#endif
#endif
}

// LDS the small-batch kernel needs for input width D (0: does not fit, or the division magics are not exact)
inline unsigned l0_div_magic(int d, int max_c) {
  const unsigned m = static_cast<unsigned>((0x100000000ull + static_cast<unsigned>(d) - 1) / static_cast<unsigned>(d));
  for (int c = 0; c <= max_c; ++c)
    if (static_cast<int>((static_cast<unsigned long long>(c) * m) >> 32) != c / d) return 0;
  return m;
}
struct L0SmallGeom {
  int sc = 0, lds = 0;
  unsigned sc_magic = 0, ch_magic = 0;
};
inline L0SmallGeom l0_small_geom(int D) {
  L0SmallGeom g;
  const int ch = D / 4, sc = ch | 1;
  const int n_ld = (32 * sc + 63) / 64;
  const int bytes = 2 * n_ld * 1024 + 2 * ((D * 4 + 1023) & ~1023) + 2048 + 64;
  if (ch < 2 || bytes > 160 * 1024) return g;  // (ch = 1: the magic 2^32 does not fit 32 bits)
  g.sc_magic = l0_div_magic(sc, n_ld * 64 + 64);
  g.ch_magic = l0_div_magic(ch, 32 * ch + 512);
  if (!g.sc_magic || !g.ch_magic) return g;
  g.sc = sc;
  g.lds = bytes;
  return g;
}

void launch_l0_small(const L0Params &p, hipStream_t s) {
  const L0SmallGeom g = l0_small_geom(p.D);
  auto k = l0_small_kernel<false>;
  auto kt = l0_small_kernel<true>;
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid(static_cast<unsigned>((p.H + 31) / 32) * static_cast<unsigned>((p.n_rows + 31) / 32));
  if (p.tap_lin)
    hipLaunchKernelGGL(kt, grid, dim3(512), g.lds, s, p, g.sc, g.sc_magic, g.ch_magic);
  else
    hipLaunchKernelGGL(k, grid, dim3(512), g.lds, s, p, g.sc, g.sc_magic, g.ch_magic);
}

// ---------------------------------------------------------------- VALU, one chain per pass (canonical flavour)
// The four k-mod-4 chains of an output are independent until (l0+l1)+(l2+l3), so a tile can
// run them one after the other: pass c walks k = c, c+4, c+8, ... with ONE accumulator per
// output instead of four.  That buys an 8 x 8 output block per thread at 64 accumulators, i.e.
// 4 ds_read_b128 per 64 MACs = 1 LDS byte per MAC; the 4 x 4 x (4 chains) kernel above needs 2,
// which at the packed-fp32 rate (64 MAC/clk/CU) is the whole 128 B/clk of the LDS.
//
// Operands come as chain-major images  img[c][j][column] = src[column][4j + c]  (column = frame
// or node): a stage is JC rows of 128 frames and JC rows of 128 nodes, every row 512 contiguous
// bytes -> LDS-DMA, no VGPR round trip, no VALU work outside the MACs.  The frame image (with
// ApplyShiftAndScale folded in) is written by l0_image_kernel before every launch (17 MB in,
// 17 MB out), the weight image once at model load.
//
// 256 threads = 16 (tx: nodes) x 16 (ty: frames); thread owns frames {4ty..4ty+3, 64+4ty..} and
// nodes {4tx..4tx+3, 64+4tx..}; frame pairs ride v_pk_mul_f32 / v_pk_add_f32.
// TN = 128: 8 x 8 outputs per thread, t = l2 + l3 parked in global scratch (two 64-register sets
// live).  TN = 64: 8 frames x 4 nodes per thread -- 32 accumulators per chain, so l2 + l3 STAYS IN
// REGISTERS (three 32-register sets live), no park traffic at all (the 128-wide tile moves 82 MB
// out and back per launch, 8x the kernel's algorithmic bytes), and the smaller register
// footprint admits three workgroups per CU; the price is 1.5 LDS bytes per MAC instead of 1.
template <int JC, int TN, bool TAP>
__global__ __launch_bounds__(256, TN == 128 ? 2 : FDNN_L0_TN64_WAVES) void l0_chain_kernel(L0Params p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins are device-only
  constexpr int TF = 128, NST = 3;
  constexpr int NA = TN / 16;              // nodes per thread: 8 or 4
  constexpr int STAGE_F = JC * (TF + TN);  // floats per stage
  constexpr int NX = JC / 2;               // 1-KiB wave loads per stage: frame rows, two per load
  constexpr int NW = TN == 128 ? JC / 2 : JC / 4;  // node rows: two (512 B) or four (256 B) per load
  constexpr int NI = NX + NW;
  constexpr int NLD = NI / 4;              // FEWEST loads any of the four waves issues per stage (vmcnt bound)
  static_assert(JC % 4 == 0, "a stage is split over four waves");
  static_assert(TN == 128 || TN == 64, "node tile");
  typedef float v2f __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float smem[];  // NST stages, then the sigmoid table
  uint8_t *lut = reinterpret_cast<uint8_t *>(smem + NST * STAGE_F);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx = tid & 15, ty = tid >> 4;
  int bx, by;
  if (!l0_tile_of_block(p.h_ld / TN, (p.n_rows + TF - 1) / TF, bx, by)) return;
  const int f0 = by * TF, n0 = bx * TN;
  if (tid < (kLutExt + 15) / 16) reinterpret_cast<uint4 *>(lut)[tid] = reinterpret_cast<const uint4 *>(p.lut)[tid];

  const int NQ = p.j_pad / JC, NS = 4 * NQ;  // stage s = (pass s / NQ, chunk s % NQ): image row s * JC
  const int voff_x = ((lane >> 5) * p.n_ld + (lane & 31) * 4) * 4;
  const int voff_w = TN == 128 ? ((lane >> 5) * p.h_ld + (lane & 31) * 4) * 4 : ((lane >> 4) * p.h_ld + (lane & 15) * 4) * 4;
  const size_t x_ld = static_cast<size_t>(p.n_ld), w_ld = static_cast<size_t>(p.h_ld);
  auto issue = [&](int s, int buf) {
    const size_t row0 = static_cast<size_t>(s) * JC;
    const auto rsrc_x = uniform_rsrc(p.xt + row0 * x_ld + f0, JC * p.n_ld * 4);
    const auto rsrc_w = uniform_rsrc(p.wt + row0 * w_ld + n0, JC * p.h_ld * 4);
    float *base = smem + buf * STAGE_F;
#pragma unroll
    for (int i = 0; i < (NI + 3) / 4; ++i) {
      const int u = i * 4 + wave;
      if (u < NX)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, FDNN_LDS_PTR(base + u * 256), 16, voff_x, u * 2 * p.n_ld * 4, 0, 0);
      else if (u < NI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(base + JC * TF + (u - NX) * 256), 16, voff_w,
                                                 (u - NX) * (TN == 128 ? 2 : 4) * p.h_ld * 4, 0, 0);
    }
  };

  // Pass order 2, 3, 0, 1 (the images store their planes in that order): l2 waits in registers
  // while l3 runs, t = l2 + l3 is parked (TN = 128: in global scratch, 64 KB per tile, read back
  // from L2 two passes later; TN = 64: in 32 registers), l0 waits while l1 runs.
  v2f acc[NA][4], held[NA][4], tsum[TN == 64 ? NA : 1][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = held[a][b] = v2f{0.0f, 0.0f};
  v2f *park = reinterpret_cast<v2f *>(p.park) + (static_cast<size_t>(by) * (p.h_ld / 128) + bx) * (32 * 256) + tid;  // TN = 128 only

  issue(0, 0);
  if (NS > 1) issue(1, 1);
#if FDNN_L0_DEBUG & 4
  typedef float v4f_dbg __attribute__((ext_vector_type(4)));
  const v4f_dbg dbg_x = {float(tid), 1.0f, 2.0f, 3.0f}, dbg_w = {0.5f, float(lane), 0.25f, 2.0f};
#endif
  int buf = 0, q = 0, pass = 0;
  for (int s = 0; s < NS; ++s) {
    // this wave's share of stage s has landed (stage s+1 may still be in flight) ...
    if (s + 1 < NS)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and everybody's has; every wave is also done with stage s-1, whose buffer stage s+2 takes
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!(FDNN_L0_DEBUG & 8) && s + 2 < NS) issue(s + 2, buf >= 1 ? buf - 1 : NST - 1);
    const float *xb = smem + buf * STAGE_F, *wb = xb + JC * TF;
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      typedef float v4f __attribute__((ext_vector_type(4)));
#if FDNN_L0_DEBUG & 4
      static_assert(sizeof(v4f) == 16, "");
      v4f xa = dbg_x, xc = dbg_x, wa = dbg_w, wc = dbg_w;
      asm volatile("" : "+v"(xa), "+v"(xc), "+v"(wa), "+v"(wc));
#else
      const v4f xa = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(xb) + (j * TF + ty * 4) * 4);
      const v4f xc = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(xb) + (j * TF + 64 + ty * 4) * 4);
      const v4f wa = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(wb) + (j * TN + tx * 4) * 4);
      v4f wc = wa;
      if (TN == 128) wc = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(wb) + (j * TN + 64 + tx * 4) * 4);
#endif
      const v2f xp[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xc.x, xc.y}, {xc.z, xc.w}};
      const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wc.x, wc.y, wc.z, wc.w};
#pragma unroll
      for (int g = 0; g < NA / 2; ++g) {
        // InputActivations, canonical flavour: multiply and add rounded separately (dnn.cc:233-238);
        // eight products first, so that no add waits on the multiply right before it
        v2f pr[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) pr[e][b] = v2f{wv[2 * g + e], wv[2 * g + e]} * xp[b];
        __builtin_amdgcn_sched_barrier(FDNN_L0_SCHED_MASK);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[2 * g + e][b] += pr[e][b];
        __builtin_amdgcn_sched_barrier(FDNN_L0_SCHED_MASK);
      }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    if (++q == NQ) {  // end of a pass; horizontalSum is (l0+l1)+(l2+l3) (dnn.cc:168-172)
      q = 0;
      if (pass == 0 || pass == 2) {  // l2 / l0 done: hold it, start l3 / l1 from zero
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            held[a][b] = acc[a][b];
            acc[a][b] = v2f{0.0f, 0.0f};
          }
      } else if (pass == 1) {  // t = l2 + l3 -> scratch (TN = 128) / registers (TN = 64)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const v2f t = held[a][b] + acc[a][b];
            if (TN == 64)
              tsum[TN == 64 ? a : 0][b] = t;
            else if (FDNN_WT & 4)  // read back from L2 either way; written through, it is not dirty when the kernel ends
              store_wt(park + (a * 4 + b) * 256, v2f_t{t.x, t.y});
            else
              park[(a * 4 + b) * 256] = t;
            acc[a][b] = v2f{0.0f, 0.0f};
          }
      }
      ++pass;
    }
  }
  // held = l0, acc = l1.  AddBias + QuantizedSigmoid; node quads go out as one dword per frame.
  float bias[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int node = n0 + (a >> 2) * 64 + tx * 4 + (a & 3);
    bias[a] = node < p.H ? p.bias[node] : 0.0f;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    v2f lin2[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) lin2[a] = (held[a][b] + acc[a][b]) + (TN == 64 ? tsum[TN == 64 ? a : 0][b] : park[(a * 4 + b) * 256]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = f0 + (b >> 1) * 64 + ty * 4 + (b & 1) * 2 + h;
      if (f >= p.n_rows) continue;
#pragma unroll
      for (int grp = 0; grp < NA / 4; ++grp) {
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int a = grp * 4 + e;
          const float lin = (h ? lin2[a].y : lin2[a].x) + bias[a];
          const int node = n0 + grp * 64 + tx * 4 + e;
          if (TAP && f < p.n && node < p.H) p.tap_lin[static_cast<size_t>(f) * p.H + node] = lin;
          packed |= static_cast<uint32_t>(lut[lut_index(lin)]) << (8 * e);
        }
        // act_ld == h_ld: the pad nodes of the last tile land in the row's pad columns
        if (FDNN_WT & 8)
          store_wt(p.act_out + static_cast<size_t>(f) * p.act_ld + n0 + grp * 64 + tx * 4, packed);
        else
          *reinterpret_cast<uint32_t *>(p.act_out + static_cast<size_t>(f) * p.act_ld + n0 + grp * 64 + tx * 4) = packed;
      }
    }
  }
#endif
}

// Chain-major image of a row-major matrix: dst[(c + 2) % 4][j][col] = src[col][4j + c] (+ shift, * scale
// when given: ApplyShiftAndScale, add then multiply, dnn.cc:184-187); rows j >= D/4 and columns
// >= n_src are zero (a zero product leaves a chain untouched).  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void l0_image_kernel(const float *src, const float *shift, const float *scale, float *dst,
                                                       int n_src, int D, int j_pad, int ld) {
  __shared__ float t[64][65];
  const int tid = threadIdx.x, lo = tid & 15, hi = tid >> 4;
  const int col0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = hi + 16 * i, k = k0 + lo * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col0 + r < n_src && k < D) {  // D is a multiple of 4
      v = *reinterpret_cast<const float4 *>(src + static_cast<size_t>(col0 + r) * D + k);
      if (shift) {
        const float4 sh = *reinterpret_cast<const float4 *>(shift + k);
        const float4 sc = *reinterpret_cast<const float4 *>(scale + k);
        v.x = (v.x + sh.x) * sc.x;
        v.y = (v.y + sh.y) * sc.y;
        v.z = (v.z + sh.z) * sc.z;
        v.w = (v.w + sh.w) * sc.w;
      }
    }
    t[r][lo * 4 + 0] = v.x;
    t[r][lo * 4 + 1] = v.y;
    t[r][lo * 4 + 2] = v.z;
    t[r][lo * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = hi + 16 * i, c = ((kl & 3) + 2) & 3, j = (k0 >> 2) + (kl >> 2);  // planes in pass order: chains 2, 3, 0, 1
    if (j >= j_pad) continue;
    const float4 v = make_float4(t[lo * 4 + 0][kl], t[lo * 4 + 1][kl], t[lo * 4 + 2][kl], t[lo * 4 + 3][kl]);
    if (FDNN_WT & 16)
      store_wt(dst + (static_cast<size_t>(c) * j_pad + j) * ld + col0 + lo * 4, v4f_t{v.x, v.y, v.z, v.w});
    else
      *reinterpret_cast<float4 *>(dst + (static_cast<size_t>(c) * j_pad + j) * ld + col0 + lo * 4) = v;
  }
}

template <int JC, int TN>
void launch_chain_tn(const L0Params &p, hipStream_t s) {
  const int cols = (p.n_rows + 127) / 128 * 128;
  hipLaunchKernelGGL(l0_image_kernel, dim3(cols / 64, (p.j_pad * 4 + 63) / 64), dim3(256), 0, s, p.x, p.shift, p.scale, p.xt, p.n,
                     p.D, p.j_pad, p.n_ld);
  dim3 grid(l0_grid(p.h_ld / TN, cols / 128));
  constexpr size_t lds = sizeof(float) * (3 * JC * (128 + TN) + (kLutExt + 15) / 16 * 4);
  if (p.tap_lin)
    hipLaunchKernelGGL((l0_chain_kernel<JC, TN, true>), grid, dim3(256), lds, s, p);
  else
    hipLaunchKernelGGL((l0_chain_kernel<JC, TN, false>), grid, dim3(256), lds, s, p);
}

template <int JC>
void launch_chain(const L0Params &p, hipStream_t s) {
  if (l0_chain_node_tile() == 64)
    launch_chain_tn<JC, 64>(p, s);
  else
    launch_chain_tn<JC, 128>(p, s);
}

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

// SCREEN (canonical flavour on the matrix pipe): the reference's canonical numerics round every product and every
// add (dnn.cc:233-238); this kernel's chains round once per step (fma).  The two agree to a few ulp, and the layer's
// OUTPUT is only the 8-bit table entry of round(100 * lin): so the fused result is used wherever the difference
// provably cannot move 100 * lin across a rounding boundary of the table index, and the handful of outputs where it
// could are listed for l0_fix_kernel, which recomputes them with the exact unfused chains.
//
// Bound (u = 2^-24; t_k = x_k * w_k exact; S = sum_k |t_k| over all four chains; s_j the computed partial sums):
//   fused chain    s_j  = fl(s_{j-1} + t_j)              =>  s_n  - sum t = sum_j eps_j,            |eps_j| <= u |s_j| / (1-u)
//   unfused chain  s'_j = fl(s'_{j-1} + fl(t_j))         =>  s'_n - sum t = sum_j eps'_j + sum_j rho_j, |rho_j| <= u |t_j|
//   (telescoping, exact).  The kernel samples the fused partial sums at the start of every window of W = 8 kScreenEvery
//   steps: inside a window |s_j| <= |s_start| + S_window, so  sum_j |s_j| <= W A + W S  with  A = sum over windows and chains of
//   |s_start|  (accumulated in registers; the kernel keeps one A per frame row and pair of node columns: an upper bound of
//   each), and |s'_j| <= |s_j| + 218 u S  (the classical gamma_n bound).  Hence
//     |lin_fused - lin_unfused| <= E = 1.0002 u (2W A + (2W + 1 + c2) S + 8 (F + |bias|)),   c2 = (D^2/2 + 2D) u  [the
//   second-order term: sum_j u |s'_j - s_j| over the D steps of the four chains],
//   the last term for the three adds of (l0+l1)+(l2+l3) and the bias add (the same operations on both sides, each can
//   widen the gap by one rounding of a value <= F + |bias|, F = sum of the four final |chain values|), and S <=
//   ||x||_2 ||w||_2 (Cauchy-Schwarz, both norms rounded up).  |fl(100 lin) - fl(100 lin')| <= D = 100.001 E + 4 u |fl(100 lin)|;
//   round() can differ only if a half-integer lies within D of fl(100 lin) (ties included): those outputs are flagged --
//   unless both neighbouring indices hold the same table byte.  Products below 2^-126 add at most 432 * 2^-150, covered by
//   the 1e-30 in D; a NaN anywhere flags.  (With the classical bound 218 u S alone 1.9 % of the outputs were flagged;
//   A is small because partial sums are random-walk sized, S is not.)
[[maybe_unused]] constexpr int kScreenEvery = FDNN_L0_SCREEN_EVERY;
template <int BK, int WFR, bool TAP, bool SCREEN = false>
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
  // SCREEN: ||(x_f + shift) * scale||^2 of the rows this thread stages, one k quad per chunk (the frame half of the
  // bound; every element passes through lstore exactly once).  Computed here, by all 16 node tiles of a frame tile over
  // again, it costs eight fma per chunk and thread; as a kernel of its own (round 2's first version) 8 us + a launch.
  float nsq[SCREEN ? XPT : 1];
#pragma unroll
  for (int q = 0; q < (SCREEN ? XPT : 1); ++q) nsq[q] = 0.0f;
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      float *dst = smem + buf * kStage + (srow + q * RPQ) * LD + skq;
      // ApplyShiftAndScale: add, then multiply (dnn.cc:184-187); quad order [k0 k2 k1 k3]
      const float v0 = (sx[q].x + sh.x) * sc.x, v1 = (sx[q].y + sh.y) * sc.y, v2 = (sx[q].z + sh.z) * sc.z, v3 = (sx[q].w + sh.w) * sc.w;
      *reinterpret_cast<float2 *>(dst) = make_float2(v0, v2);
      *reinterpret_cast<float2 *>(dst + 2) = make_float2(v1, v3);
      if (SCREEN) nsq[q] = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, nsq[q]))));
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
  // SCREEN: A of the bound above.  One register per frame row of this lane, shared by its two node columns (the sum
  // over both columns bounds each: 16 registers instead of 32 -- with 32 the k-loop spilled).
  float absacc[SCREEN ? 16 : 1];
  if (SCREEN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) absacc[SCREEN ? r : 0] = 0.0f;
  }
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (!(FDNN_L0_DEBUG & 2) && c + 1 < nchunk) gload(c + 1);
    if (SCREEN && c > 0 && c % kScreenEvery == 0) {  // the chains' partial sums at the start of this window (zero before the first)
      // fenced: left to itself the scheduler overlaps these reads with the next MFMAs by copying the accumulators they
      // are about to overwrite (289 spilled registers)
      __builtin_amdgcn_sched_barrier(0);
      // the reads below are inline asm: the compiler's hazard recognizer does not see that they consume matrix-pipe
      // results (a 16-pass MFMA's destination may not be read by the vector ALU for 18 wait states).  The previous
      // chunk's last MFMA is a staging pass and a barrier away already; these 20 idle slots make that independent of
      // how the code in between is scheduled.
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // eight in-place v_add_f32 with |.| on the accumulator operand (written as asm: through fabsf() the compiler
        // copied the four 32-register accumulator tuples before reading them)
        float a = absacc[SCREEN ? r : 0];
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
          asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a) : "v"(acc[sc >> 1][sc & 1][r]));
          asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(a) : "v"(acc[sc >> 1][sc & 1][16 + r]));
        }
        absacc[SCREEN ? r : 0] = a;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
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
  // SCREEN: the tile's list of flagged outputs lives behind the byte tile (TF * TS bytes) in the dead staging ring
  uint32_t *scr_n = reinterpret_cast<uint32_t *>(tile + TF * TS);
  uint16_t *scr_l = reinterpret_cast<uint16_t *>(tile + TF * TS + 16);
  float *xn_s = reinterpret_cast<float *>(tile + TF * TS + 16 + 2 * kL0ScreenCap);  // [TF] frame norms, rounded up
  static_assert(!SCREEN || TF * TS + 16 + 2 * kL0ScreenCap + TF * 4 <= 2 * kStage * 4, "flag list and norms must fit in the staging ring");
  static_assert((TF * TS + 16 + 2 * kL0ScreenCap) % 4 == 0, "norm array alignment");
  if (SCREEN) {
    if (tid == 0) *scr_n = 0;
    // the QPR threads of a row sit in adjacent lanes: sum their partial squares, one of them publishes the norm.  The
    // float (fma) sum of D squares is within D u of the true one, the square root and these adds a few u more: rounded
    // up by (1 + 2 D u + 1e-5)  (D <= 2^20 by the loader)
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      float s2 = nsq[SCREEN ? q : 0];
#pragma unroll
      for (int off = 1; off < QPR; off <<= 1) s2 += __shfl_xor(s2, off);
      if (tid % QPR == 0) xn_s[srow + q * RPQ] = sqrtf(s2) * (1.00001f + 2.0f * 5.9604645e-8f * static_cast<float>(p.D));
    }
    __syncthreads();
  }
  // SCREEN, pass 1 (branch free): one flag bit per output of this lane (bit 16 s + r).  The frame norms of the lane's
  // 16 rows are fetched once.
  uint32_t scr_mask = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int col = wn * 64 + s * 32 + l32;
    const int node = n0 + col;
    const float bias = node < p.H ? p.bias[node] : 0.0f;
    const float wn_bound = (SCREEN && node < p.H) ? p.wnorm[node] : 0.0f;
    constexpr float kScreenE = 1.0002f * 5.9604645e-8f, kScreenT = 4.0f * 5.9604645e-8f;  // 1.0002 u (covers 1/(1-u) and float evaluation of E), 4 u
    // second order in u: the unfused chain's own partial sums differ from the sampled (fused) ones, |s'_j - s_j| <= (D/2 + 2) u S,
    // which over the D steps of the four chains adds (D^2 / 2 + 2 D) u^2 S -- 0.006 u S at D = 432, but growing with D^2
    // (it used to ride on the 1.0002's slack, which only holds up to D ~ 470)
    const float c2 = 1.001f * (0.5f * static_cast<float>(p.D) * static_cast<float>(p.D) + 2.0f * static_cast<float>(p.D)) * 5.9604645e-8f;
    uint8_t act[16];  // table gathers first, tile writes after: the two alias in LDS as far as the compiler knows
    uint8_t nb0[16], nb1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
      // horizontalSum: (l0+l1)+(l2+l3) (dnn.cc:168-172), then AddBias
      const float sum = (acc[s][0][r] + acc[s][0][16 + r]) + (acc[s][1][r] + acc[s][1][16 + r]);
      const float lin = sum + bias;
      if (TAP && f0 + row < p.n && node < p.H) p.tap_lin[static_cast<size_t>(f0 + row) * p.H + node] = lin;
      act[r] = lut[lut_index(lin)];
      if (SCREEN) {  // the table bytes on both sides of the half-integer nearest to 100 lin
        const float t = lin * 100.0f;
        const float fl = floorf(t);
        const int lo = fabsf(t) < 1.0e6f ? static_cast<int>(fl) : 0;
        nb0[r] = lut[max(-kLutHalf, min(kLutHalf, lo)) + kLutHalf];
        nb1[r] = lut[max(-kLutHalf, min(kLutHalf, lo + 1)) + kLutHalf];
      }
    }
    if (SCREEN) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        const float t = (((acc[s][0][r] + acc[s][0][16 + r]) + (acc[s][1][r] + acc[s][1][16 + r])) + bias) * 100.0f;  // as above
        const float xn = f0 + row < p.n ? xn_s[row] : 0.0f;
        const float F = (fabsf(acc[s][0][r]) + fabsf(acc[s][0][16 + r])) + (fabsf(acc[s][1][r]) + fabsf(acc[s][1][16 + r]));
        constexpr float kW = 2.0f * (BK / 4) * kScreenEvery;  // 2 x steps per window: both chains' partial sums
        const float E = fmaf(kW, absacc[SCREEN ? r : 0], fmaf((kW + 1.0f + c2) * xn, wn_bound, 8.0f * (F + fabsf(bias)))) * kScreenE;
        const float D = fmaf(100.001f, E, fabsf(t) * kScreenT) + 1e-30f;
        const float fr = fabsf((t - floorf(t)) - 0.5f);  // distance to the nearest half-integer (exact below 2^23)
        const bool near = !(fr > D) && !(fabsf(t) - D >= 641.0f);     // written so that a NaN flags
        const bool one_boundary = D < 0.25f && fabsf(t) < 1.0e6f;     // else: several boundaries in reach, always recompute
        const bool flag = near && !(one_boundary && nb0[r] == nb1[r]) && f0 + row < p.n && node < p.H;
        scr_mask |= flag ? (1u << (16 * s + r)) : 0u;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * TS + col] = act[r];
  }
  if (SCREEN) {
    // pass 2: the wave reserves room for all its flagged outputs with ONE LDS atomic, every lane then writes its own
    // entries (a push per output cost an atomic round trip in 3 of 4 loop iterations: 413 us instead of 188)
    const int mine = __popc(scr_mask);
    int incl = mine;  // inclusive prefix sum over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const int wave_total = __shfl(incl, 63);
    uint32_t base = 0;
    if (wave_total) {
      if (lane == 63) base = atomicAdd(scr_n, static_cast<uint32_t>(wave_total));
      base = __shfl(base, 63);
      uint32_t at = base + static_cast<uint32_t>(incl - mine);
      uint32_t m = scr_mask;
      while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        const int r = i & 15, sb = i >> 4;
        const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3), col = wn * 64 + sb * 32 + l32;
        if (at < static_cast<uint32_t>(kL0ScreenCap)) scr_l[at] = static_cast<uint16_t>(row * TN + col);
        ++at;
      }
    }
  }
  __syncthreads();
  if (SCREEN) {
    const int tile_id = by * ((p.H + TN - 1) / TN) + bx;
    const uint32_t cnt = *scr_n;
    if (tid == 0) p.scr_count[tile_id] = cnt;  // > kL0ScreenCap: the fix kernel recomputes the whole tile
    const uint32_t listed = min(cnt, static_cast<uint32_t>(kL0ScreenCap));
    for (uint32_t i = tid; i < listed; i += THREADS) p.scr_list[static_cast<size_t>(tile_id) * kL0ScreenCap + i] = scr_l[i];
  }
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

// element `src` (0..3) of every aligned group of four lanes, to all four: a DPP quad_perm move, no LDS traffic
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), SRC * 0x55, 0xf, 0xf, true));
}

// The exact canonical chains (four k-mod-4 partial sums, multiply and add rounded separately, (l0+l1)+(l2+l3), bias,
// table: dnn.cc:219-286) for the outputs l0_mfma_kernel<SCREEN> listed -- all 128 x 128 of them when a tile's list
// overflowed.  One workgroup per tile, FOUR lanes per listed output: lane c is chain c.  The four lanes walk
// the two operand rows 64 contiguous bytes per load instruction -- lane c fetches the quad of k-step 4i + c (x with shift
// and scale applied, w) -- and each step's quad is then handed round with quad broadcasts, chain c keeping element c.
// (One lane per output reading 16 bytes of its own rows per load is address-processing bound: every (output, k-quad) is
// its own 16-byte segment -- 66 us for 0.4 % of the outputs.)
#ifndef FDNN_L0_FIX_DEPTH
#define FDNN_L0_FIX_DEPTH 3  // operand quads per lane in flight per round of the exact recomputation
#endif
// One listed output, LPO lanes (all of them must call this together): lane c fetches the quads of k-steps LPO i + c and keeps
// chain c & 3 (LPO = 8: lanes 4..7 repeat chains 0..3, so that every quad of lanes ends with all four chain sums).
template <int NB = FDNN_L0_FIX_DEPTH, int LPO = 4>
__device__ __forceinline__ void fix_one_output(const L0Params &p, const float *sh_s, const float *sc_s, float *tr_w, int f, int node, bool live, int c, int quads) {
  static_assert(LPO == 4 || LPO == 8, "lanes per output");
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float *xr = p.x + static_cast<size_t>(live ? f : 0) * p.D, *wr = p.w + static_cast<size_t>(live ? node : 0) * p.D;
  float acc = 0.0f;
  // LPO NB k-steps per round: the NB quads per lane (and operand) are all requested before the first is used -- the walk
  // is a chain of L2 round trips, one per round (LPO = 4, NB = 3: nine of them for D = 432).
  // Chain e needs element e of EVERY step's quad: the owner of a quad forms its four products (multiply rounds on its
  // own, -ffp-contract=off), and the block of products of LPO consecutive steps changes hands through the wave's 1 KiB of
  // LDS per b -- four conflict-free dword writes (element e of lane l to row e, slot l), then 16-byte reads of row c & 3,
  // slots of the output's lanes = its steps in order.  Sixteen DPP broadcasts and twelve selects per four steps before:
  // the chain was vector-instruction bound, not L2 bound.
  const int lane = threadIdx.x & 63, l0 = lane & ~(LPO - 1);
  for (int q0 = 0; q0 < quads; q0 += LPO * NB) {
    v4f xq[NB], wq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q = min(q0 + LPO * b + c, quads - 1);  // (a clamped quad is never consumed: its step is skipped below)
      xq[b] = *reinterpret_cast<const v4f *>(xr + 4 * q);
      wq[b] = *reinterpret_cast<const v4f *>(wr + 4 * q);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q = min(q0 + LPO * b + c, quads - 1);
      xq[b] = (xq[b] + *reinterpret_cast<const v4f *>(sh_s + 4 * q)) * *reinterpret_cast<const v4f *>(sc_s + 4 * q);  // add, then multiply
      const v4f pr = xq[b] * wq[b];  // product and sum round separately (dnn.cc:233-238)
      float *t = tr_w + b * 256;
      t[0 * 64 + lane] = pr.x;
      t[1 * 64 + lane] = pr.y;
      t[2 * 64 + lane] = pr.z;
      t[3 * 64 + lane] = pr.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (compiler ordering only: the exchange is between lanes of one wave,
    __builtin_amdgcn_wave_barrier();                        //  whose LDS operations complete in order)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int h = 0; h < LPO / 4; ++h) {
        const v4f r = *reinterpret_cast<const v4f *>(tr_w + b * 256 + (c & 3) * 64 + l0 + 4 * h);
        const int st = q0 + LPO * b + 4 * h;
        if (st + 0 < quads) acc = acc + r.x;
        if (st + 1 < quads) acc = acc + r.y;
        if (st + 2 < quads) acc = acc + r.z;
        if (st + 3 < quads) acc = acc + r.w;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const float c0 = quad_bcast<0>(acc), c1 = quad_bcast<1>(acc), c2 = quad_bcast<2>(acc), c3 = quad_bcast<3>(acc);
  if (live && c == 0) {
    const float lin = ((c0 + c1) + (c2 + c3)) + p.bias[node];  // horizontalSum (dnn.cc:168-172), AddBias
    if (p.tap_lin) p.tap_lin[static_cast<size_t>(f) * p.H + node] = lin;
    p.act_out[static_cast<size_t>(f) * p.act_ld + node] = static_cast<int8_t>(p.lut[lut_index(lin)]);
  }
}

#ifndef FDNN_L0_FIX_THREADS
#define FDNN_L0_FIX_THREADS 512  // 128 outputs per pass: 40.1 us against 42.6 (256), 45.4 (128), 71 (1024) for 82 000 outputs
#endif
constexpr int kFixThreads = FDNN_L0_FIX_THREADS;
__global__ __launch_bounds__(kFixThreads) void l0_fix_kernel(L0Params p, int TF) {  // TF: the screening kernel's frame tile (128 or 64)
  constexpr int TN = 128;
  const int node_tiles = (p.H + TN - 1) / TN;
  const int tile_id = blockIdx.x, by = tile_id / node_tiles, bx = tile_id % node_tiles;
  const uint32_t count = p.scr_count[tile_id];
  if (count == 0) return;
  const int f0 = by * TF, n0 = bx * TN;
  const bool all = count > static_cast<uint32_t>(kL0ScreenCap);
  const int total = all ? TF * TN : static_cast<int>(count);
  const int tid = threadIdx.x, c = tid & 3;
  const int quads = p.D / 4;  // D is a multiple of 4
  extern __shared__ __attribute__((aligned(16))) float fix_smem[];  // shift[D], scale[D]
  float *sh_s = fix_smem, *sc_s = fix_smem + p.D;
  float *tr_w = fix_smem + 2 * p.D + (tid >> 6) * (FDNN_L0_FIX_DEPTH * 256);  // this wave's product blocks (fix_one_output)
  for (int k = tid; k < p.D; k += kFixThreads) {
    sh_s[k] = p.shift[k];
    sc_s[k] = p.scale[k];
  }
  __syncthreads();
  for (int base = 0; base < total; base += kFixThreads / 4) {  // (uniform trip count: all four lanes of an output must be present)
    const int o = base + (tid >> 2);
    const bool valid = o < total;
    const int local = valid ? (all ? o : p.scr_list[static_cast<size_t>(tile_id) * kL0ScreenCap + o]) : 0;
    const int f = f0 + local / TN, node = n0 + local % TN;
    fix_one_output(p, sh_s, sc_s, tr_w, f, node, valid && f < p.n && node < p.H, c, quads);
  }
  __syncthreads();  // every thread has read the count and its entries
  if (tid == 0) {
    p.scr_count[tile_id] = 0;  // ready for the next launch
    // ONE atomic per tile (one per recomputed output -- 80 000 on one address -- cost more than the recomputation)
    if (p.scr_stats) atomicAdd(p.scr_stats + 1, static_cast<unsigned long long>(all ? TF * TN : count));
  }
}

// Round 4 (int8 screening): the flagged outputs of ALL tiles in one list (the matrix kernel reserves room with one global
// atomic per tile), recomputed 128 per workgroup pass.  With a workgroup per tile every workgroup ran one pass for its
// ~70 outputs -- 1280 latency-bound passes, 2.5 rounds of them on the chip; the same outputs are 570 full passes, all
// resident at once.  A tile whose own count overflowed the per-tile list (> 25 % flagged) is still recomputed whole.
template <int NB, int THREADS, int LPO>
__global__ __launch_bounds__(THREADS) void l0_fix_list_kernel(L0Params p, int tiles) {
  constexpr int TN = 128, TF = 128;
  constexpr int kFixThreads = THREADS;  // (shadows the per-tile kernel's constant)
  const int tid = threadIdx.x, c = tid & (LPO - 1);
  const int quads = p.D / 4;
  extern __shared__ __attribute__((aligned(16))) float fix_smem[];  // shift[D], scale[D]
  float *sh_s = fix_smem, *sc_s = fix_smem + p.D;
  float *tr_w = fix_smem + 2 * p.D + (tid >> 6) * (NB * 256);  // this wave's product blocks (fix_one_output)
  const uint32_t total = min(p.glist_count[0], static_cast<uint32_t>(p.glist_cap));
  const bool any_overflow = p.glist_count[1] != 0u;  // some tile kept its outputs to itself (scr_count / whole-tile path)
  // Which piece of the list is whose: the list is (nearly) in tile order, frame tiles major, so a contiguous eighth of it
  // touches an eighth of the frame rows -- and workgroup b runs on XCD b % 8, whose L2 is its own.  Workgroup b takes piece
  // (b % 8) * pieces_per_xcd + b / 8 (+ gridDim / 8, ...): an XCD's L2 then holds ITS frame rows beside the weight rows
  // (2.1 + 3.5 MB on the bench batch) instead of every XCD streaming all 17 MB of rows through 4 MB (FDNN_L0_FIX_XCD=0: b -> piece b).
  const uint32_t per = kFixThreads / LPO, pieces = (total + per - 1) / per, ppx = (pieces + 7) / 8;
  const uint32_t xcd = blockIdx.x & 7u, wj = blockIdx.x >> 3, wpx = max(1u, gridDim.x >> 3);
  if (wj >= ppx && !any_overflow) return;
  for (int k = tid; k < p.D; k += kFixThreads) {
    sh_s[k] = p.shift[k];
    sc_s[k] = p.scale[k];
  }
  __syncthreads();
  for (uint32_t pj = wj; pj < ppx; pj += wpx) {
    const uint32_t piece = xcd * ppx + pj;
    if (piece >= pieces) break;
    const uint32_t o = piece * per + (tid / LPO);
    const bool valid = o < total;
    const uint2 ent = valid ? p.glist[o] : make_uint2(0u, 0u);
    fix_one_output<NB, LPO>(p, sh_s, sc_s, tr_w, static_cast<int>(ent.x), static_cast<int>(ent.y), valid && ent.x < static_cast<uint32_t>(p.n) && ent.y < static_cast<uint32_t>(p.H), c, quads);
  }
  if (any_overflow) {
    const int node_tiles = (p.H + TN - 1) / TN;
    for (int tile_id = blockIdx.x; tile_id < tiles; tile_id += gridDim.x) {
      const uint32_t count = p.scr_count[tile_id];
      if (count == 0) continue;
      const int f0 = (tile_id / node_tiles) * TF, n0 = (tile_id % node_tiles) * TN;
      for (int base = 0; base < TF * TN; base += kFixThreads / LPO) {
        const int o = base + (tid / LPO);
        const int f = f0 + o / TN, node = n0 + o % TN;
        fix_one_output<NB, LPO>(p, sh_s, sc_s, tr_w, f, node, f < p.n && node < p.H, c, quads);
      }
      __syncthreads();
      if (tid == 0) {
        p.scr_count[tile_id] = 0;
        if (p.scr_stats) atomicAdd(p.scr_stats + 1, static_cast<unsigned long long>(TF * TN));
      }
    }
  }
  if (blockIdx.x == 0 && tid == 0 && p.scr_stats) atomicAdd(p.scr_stats + 1, static_cast<unsigned long long>(total));
  // (glist_count is zeroed by the next launch's pre-pass, stream-ordered before its matrix kernel)
}

template <int BK, int WFR>
void launch_mfma(const L0Params &p, hipStream_t s) {
  using Cfg = L0MfmaCfg<BK, WFR>;
  auto k_prod = l0_mfma_kernel<BK, WFR, false>;
  auto k_tap = l0_mfma_kernel<BK, WFR, true>;
  // the attribute is per device: a process may hold models on several GPUs
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tap), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid(l0_grid((p.H + Cfg::TN - 1) / Cfg::TN, (p.n_rows + Cfg::TF - 1) / Cfg::TF));
  hipLaunchKernelGGL(p.tap_lin ? k_tap : k_prod, grid, dim3(Cfg::THREADS), Cfg::LDS, s, p);
}

// Canonical numerics through the screened path: fused chains + screening (frame norms included), exact recomputation of the flagged.
// WFR = 4: 128 x 128 tiles, one 512-thread workgroup per CU; WFR = 2: 64 x 128 tiles, two 256-thread workgroups per CU
// (one's screening epilogue under the other's matrix stream, and half the batch-size staircase).
template <int WFR>
void launch_screened_cfg(const L0Params &p, hipStream_t s) {
  using Cfg = L0MfmaCfg<32, WFR>;
  static_assert(Cfg::TN == 128, "l0_fix_kernel assumes 128-node tiles");
  auto k_scr = l0_mfma_kernel<32, WFR, false, true>;
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_scr), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  const int node_tiles = (p.H + 127) / 128, frame_tiles = (p.n_rows + Cfg::TF - 1) / Cfg::TF;
  hipLaunchKernelGGL(k_scr, dim3(l0_grid(node_tiles, frame_tiles)), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  hipLaunchKernelGGL(l0_fix_kernel, dim3(node_tiles * frame_tiles), dim3(kFixThreads), 2 * sizeof(float) * p.D + (kFixThreads / 64) * FDNN_L0_FIX_DEPTH * 1024, s, p, Cfg::TF);
}
int l0_screen_wfr() {
  static const int wfr = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_L0_SCREEN_WFR");
    return e && std::atoi(e) == 2 ? 2 : 4;
  }();
  return wfr;
}
void launch_screened(const L0Params &p, hipStream_t s) {
  if (l0_screen_wfr() == 2)
    launch_screened_cfg<2>(p, s);
  else
    launch_screened_cfg<4>(p, s);
}

}  // namespace

void launch_l0(const L0Params &p, hipStream_t s) {
  static const bool fma_on_valu = FDNN_TUNE_ENV("FDNN_L0_FMA_VALU") != nullptr;
  if (p.fma && !fma_on_valu) {
    // 32-float chunks, 4 x 2 waves (128 x 128 tile).  Measured alternatives at 10 000 frames:
    // 16-float chunks 0.221 ms, 64-float 0.205, 256-thread workgroups (two per CU) 0.205.
    launch_mfma<32, 4>(p, s);
    return;
  }
  // Small batches (canonical flavour): the whole-K kernel, 32 x 32 tiles (100 frames: 20 -> 6 us)
  static const int small_max = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_L0_SMALL_MAX");
    return e ? std::atoi(e) : 128;  // one round of 32 x 32 tiles on a 2048-node layer; beyond, the (16 | 32) x 64 tiles are as fast
  }();
  if (!p.fma && p.kernel == 0 && p.n_rows <= small_max && l0_small_geom(p.D).lds > 0) {
    launch_l0_small(p, s);
    return;
  }
  // Three bit-identical candidates, chosen by modelled time (432 -> 2048 layer, tools/l0_kind_sweep.py; all three
  // come in rounds of 256 tiles, one per CU, and a round costs the same full or not):
  //   screened   fused chains on the matrix pipe + exact recomputation of the few outputs the fusion could change
  //              (128 x 128 tiles): 74 / 122 / 174 / 234 / 285 us for 1..5 rounds; large batches without taps only
  //   chain      all-VALU chain kernel (128 frames x 64 nodes): 57 / 82 / 110 / 142 / 171 / 203 us for 1..6 rounds
  //   tile64     (16 | 32 | 64) x 64 tiles, 4 outputs x 4 chains per thread and frame: 19-23 us up to 400 frames, 33-49 us up to
  //              1200, then 20 us + 35.5 ns per frame
  // e.g. 2560 frames: 120 screened, 109 chain; 3000: 110 chain, 125 the others; 6000: 174 screened, 204 chain.
  static const bool no_screen = FDNN_TUNE_ENV("FDNN_L0_NO_SCREEN") != nullptr;
  static const bool classic = FDNN_TUNE_ENV("FDNN_L0_CLASSIC") != nullptr;
  const double work = static_cast<double>(p.D) / 432.0;
  const long ft128 = (p.n_rows + 127) / 128;
  const double screened_us = 18.0 + 54.0 * work * static_cast<double>((ft128 * ((p.H + 127) / 128) + 255) / 256);
  const double chain_us = 27.0 + 30.0 * work * static_cast<double>((ft128 * (p.h_ld / l0_chain_node_tile()) + 255) / 256);
  const double tile64_us = p.n_rows <= 320    ? 23.0 * work
                           : p.n_rows <= 1200 ? (17.0 + 0.032 * p.n_rows) * work
                                              : 20.0 + 0.0355 * work * (p.H / 2048.0) * p.n_rows;
  const bool can_screen = !p.fma && !no_screen && (p.kernel == 0 || p.kernel == 3) && !p.tap_lin && p.wnorm && p.scr_count && p.scr_list && p.n >= 2048 &&
                          p.D <= 4096;  // l0_fix_kernel: 8 D bytes of shift / scale + 24 KB of product blocks in dynamic LDS (64 KB without an attribute)
  const bool can_chain = !p.fma && p.xt && p.wt && p.kernel != 2 && !(classic && p.kernel == 0);
  // Round 4: the screening on the int8 matrix pipe (fdnn_l0s.hip: exact 24-bit integer images of both operands, eight
  // int8 MFMA products) + the same exact recomputation of the flagged outputs.  128 x 128 tiles, 1.85 us of matrix-pipe
  // time per tile and CU at peak: from FDNN_L0_SPLIT_MIN frames up it replaces all of the above.
  static const bool no_split = FDNN_TUNE_ENV("FDNN_L0_NO_SPLIT") != nullptr;
  static const int split_min = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_L0_SPLIT_MIN");
    return e ? std::atoi(e) : 560;  // whole call at 512 / 600 frames: 118 / 140 us with the 64 x 64 tiles, 121 / 136 with the screening
  }();
  const bool can_split = !p.fma && !no_split && !no_screen && (p.kernel == 0 || p.kernel == 4) && !p.tap_lin && p.xd && p.xstat && p.wd && p.wstat && p.luthalf && p.glist && p.glist_count &&
                         p.scr_count && p.scr_list && l0_split_ok(p.D, p.H);
  if (can_split && (p.kernel == 4 || p.n >= split_min)) {
    launch_l0_split(p, s);
    const int node_tiles = p.h_ld / 128, frame_tiles = (p.n_rows + 127) / 128;
    // Variant by batch size (rocprofv3, us at 1 000 / 4 000 / 10 000 frames; LABBOOK): four lanes per output, three quads per
    // lane and operand in flight, 256 threads: 8.7 / 11.4 / 23.8; EIGHT lanes per output (a whole 128-byte line per output
    // and load, five round trips instead of nine): 7.3 / 13.8 / 25.0 -- fewer flagged outputs = a latency chain, many = L2
    // gathers, where the second set of lanes only costs registers.  FDNN_L0_FIX_NB / _T / _LPO force a variant.
    static const int force_nb = [] {
      const char *e = FDNN_TUNE_ENV("FDNN_L0_FIX_NB");
      return e ? std::atoi(e) : 0;
    }();
    static const int force_t = [] {
      const char *e = FDNN_TUNE_ENV("FDNN_L0_FIX_T");
      return e ? std::atoi(e) : 0;
    }();
    static const int force_lpo = [] {
      const char *e = FDNN_TUNE_ENV("FDNN_L0_FIX_LPO");
      return e ? std::atoi(e) : 0;
    }();
    const int nb = force_nb ? force_nb : 3;
    const int thr = force_t == 512 ? 512 : 256;
    const int lpo = force_lpo ? (force_lpo == 8 ? 8 : 4) : (p.n_rows < 3000 ? 8 : 4);
    // enough workgroups for 1.5 % flagged in one pass each (0.35 % on the bench batch); more is walked in further passes
    const long expect = static_cast<long>(p.n_rows) * p.H * 3 / 200 / (thr / lpo) + 8;
    const int grid = static_cast<int>(std::min<long>((expect + 7) / 8 * 8, 8192));  // (a multiple of 8: l0_fix_list_kernel deals its pieces per XCD)
    const int tiles = node_tiles * frame_tiles;
#define FDNN_FIX_LAUNCH(NB_, T_, L_) hipLaunchKernelGGL((l0_fix_list_kernel<NB_, T_, L_>), dim3(grid), dim3(T_), 2 * sizeof(float) * p.D + (T_ / 64) * NB_ * 1024, s, p, tiles)
#define FDNN_FIX_LAUNCH_T(T_)                                                        \
  do {                                                                               \
    if (lpo == 8) {                                                                  \
      if (nb >= 3) FDNN_FIX_LAUNCH(3, T_, 8); else FDNN_FIX_LAUNCH(2, T_, 8);        \
    } else {                                                                         \
      if (nb >= 5) FDNN_FIX_LAUNCH(5, T_, 4); else FDNN_FIX_LAUNCH(3, T_, 4);        \
    }                                                                                \
  } while (0)
    if (thr == 512) FDNN_FIX_LAUNCH_T(512); else FDNN_FIX_LAUNCH_T(256);
#undef FDNN_FIX_LAUNCH_T
#undef FDNN_FIX_LAUNCH
    return;
  }
  if (can_screen && (p.kernel == 3 || (screened_us < (can_chain ? chain_us : tile64_us) && screened_us < tile64_us))) {
    launch_screened(p, s);
    return;
  }
  const bool chain = can_chain && (p.kernel == 1 || chain_us < tile64_us);
  if (chain) {
    if (p.jc == 12)
      launch_chain<12>(p, s);
    else
      launch_chain<16>(p, s);
    return;
  }
  // 64 x 64 tile, 16-float chunks, 4 x 4 outputs per thread.  Measured alternatives: 32-float
  // chunks 0.448 ms, 8 x 4 outputs per thread 0.468 / 0.477 ms (occupancy 2) against 0.388.
  static const int t64_bk = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_L0_T64_BK");
    return e ? std::atoi(e) : 0;
  }();
  // Few frames: a 64 x 64 tile is 30-40 us of dependent vector work for ONE workgroup however few of them there are, so
  // small batches take 16- / 32-frame tiles (more, shorter workgroups).  Measured (tools/l0_kind_sweep.py), 8 / 100 / 256 /
  // 512 / 1000 frames: 16 x 64 tiles 19 / 21 / 23 / 38 / 60 us, 32 x 64 27 / 30 / 30 / 33 / 49, 64 x 64 43 / 45 / 45 / 45 / 55.
  if (t64_bk == 164 || (t64_bk == 0 && p.n_rows <= 320))
    launch_valu<1, 64>(p, s);
  else if (t64_bk == 232 || (t64_bk == 0 && p.n_rows <= 1200))
    launch_valu<2, 32>(p, s);
  else
    launch_valu<4, 16>(p, s);
}

// Node tile of the chain kernel: 64 (8 frames x 4 nodes per thread, every partial sum in registers,
// three workgroups per CU) or 128 (8 x 8, l2 + l3 parked in a global scratch buffer).  Both run
// at the same speed -- the kernel is bound by vector-instruction issue, 329.8 vs 331.7 us at
// 10 000 frames (rocprofv3) -- but the 64-wide tile moves 164 MB less through HBM per launch and
// needs no scratch, which is what the soft-max scale running underneath it in the server loop
// competes for.  FDNN_L0_TN=128 selects the round-1 shape.
int l0_chain_node_tile() {
  static const int tn = [] {
    const char *e = std::getenv("FDNN_L0_TN") /* test hook: 128 = the round-1 tile shape, kept selectable */;
    return (e && std::atoi(e) == 128) ? 128 : 64;
  }();
  return tn;
}

int l0_chunk_rows(int D) {
  const int J = D / 4, p12 = (J + 11) / 12 * 12, p16 = (J + 15) / 16 * 16;
  return p12 <= p16 ? 12 : 16;
}

void launch_l0_weight_image(const float *w, float *wt, int H, int D, int j_pad, int h_ld, hipStream_t s) {
  hipLaunchKernelGGL(l0_image_kernel, dim3(h_ld / 64, (j_pad * 4 + 63) / 64), dim3(256), 0, s, w, nullptr, nullptr, wt, H, D, j_pad,
                     h_ld);
}

}  // namespace fdnn
