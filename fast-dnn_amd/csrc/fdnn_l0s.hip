// fdnn_l0s.hip -- layer 0, canonical numerics, large batches: screening on the INT8 matrix pipe.
//
//   ApplyShiftAndScale + InputActivations + AddBias + QuantizedSigmoid   (dnn.cc:175-192, :219-286)
//
// The layer's output is the table byte of round(100 lin), lin = (l0 + l1) + (l2 + l3) + bias with l_c the reference's
// four k-mod-4 fp32 chains (every product and every add rounded, dnn.cc:233-238, :168-172).  For all but a few
// outputs ANY sufficiently accurate value of lin gives the same byte.  Round 2 took that value from fused chains on
// the fp32 matrix pipe (1/16 of the bf16 rate, 235 us per 10 000 frames).  Here it comes from EXACT integer
// arithmetic on the int8 pipe (2 x the bf16 rate):
//
//   frame row f:  X_k = rint(x_k c_f)  with  c_f = 8355000 / max_k |x_k|    (24-bit signed integers, |X_k| <= 8355711)
//   node row n:   W_k = rint(w_k c_n)  likewise (model load, in double)
//   X = 65536 X1 + 256 X2 + X3,  W likewise, balanced digits in [-128, 127]   -> three int8 planes per operand
//   sum_k X_k W_k = 2^32 [P0 + 2^-8 P1 + 2^-16 P2 + 2^-24 P3 + 2^-32 P4],   P_o = sum over digit pairs of order o
//   P0 = X1.W1, P1 = X1.W2 + X2.W1, P2 = X1.W3 + X2.W2 + X3.W1: six int8 MFMA products, int32 accumulators, no rounding
//   anywhere; P3 = X2.W3 + X3.W2 and P4 = X3.W3 are dropped and bounded (|X2|, |X3| <= 128: |P3| <= 128 (||W2||_1 +
//   ||W3||_1), |P4| <= 128 ||W3||_1).  (With P3 kept -- eight products -- 0.27 % of the outputs are flagged instead of
//   0.35 %, at a quarter more matrix work and a fourth accumulator set: profiles/LABBOOK.md.)
//
// lin~ = sigma V + bias,  V = P0 + 2^-8 P1 + 2^-16 P2,  sigma = 2^32 / (c_f c_n).  What separates lin~ from the
// reference's lin is then (a) the quantisation x c_f - X, w c_n - W, (b) P3 + P4, (c) a few float roundings of the final
// evaluation -- all bounded per row at no cost -- and (d) the REFERENCE's own rounding errors
//     |lin_ref - (sum_k x_k w_k + bias)|  <=  u (sum_{c,j} |s'_{c,j}| + S + 3 sum_c |l_c| + |lin_ref|),   S = sum_k |x_k w_k|,
// s'_{c,j} its partial sums (telescoping, exact; u = 2^-24).  Partial sums are random-walk sized and S is not, so as in
// round 2 the kernel SAMPLES them: the k order of the planes is chain-major, every chain padded with zeros to a whole
// number of 32-position chunk pairs (432 -> 4 x 128 positions), P0 is accumulated chain by chain, and after every MFMA
// of P0 the accumulator is added, as |.|, to A.  Between two samples a, b of a chain, at most w = 32 steps apart,
//     |s'_j| <= min(|s'_a| + F_j, |s'_b| + B_j),  F_j + B_j = (window's sum of |fl(t)| + |eps|)   =>
//     sum_j |s'_j| <= (w/2)(|s'_a| + |s'_b|) + (w/2) S_window       (two-sided; round 2 used the one-sided form)
// and a sample of P0 differs from the reference's partial sum by the low digits: |X W - 2^32 X1 W1| <= 2^32 0.502
// (|X1| + |W| / 65536), i.e. per sample at most 0.502 (||X1||_1 + ||W||_1 / 65536) of the chain's rows.  Hence
//     E = sigma (kA A + a_f + b_n) + kS ||x||_2 ||w||_2 + 10 u |bias|
//       kA = (w + 3) u        A = sum over samples of |P0 partial|     (+3: the three combining adds and the bias add)
//       kS = (w/2 + 2) u      (+1: the products' own roundings; +1: the x c_f multiply's rounding, u |X W|)
//       a_f = u (w+3) m 0.502 ||X1||_1 + 2^-32 (0.5 ||X||_1 + 0.5 D) + u 2^8 D                          (per frame, pre-pass)
//       b_n = u (w+3) m 0.502 ||W||_1 / 65536 + 2^-32 0.5 ||W||_1 + 2^-24 128 (||W2||_1 + ||W3||_1) + 2^-32 128 ||W3||_1   (per node, model load)
//       m = samples per chain = chunks per chain + 1
// all evaluated with upward slack; |fl(100 lin_ref) - fl(100 lin~)| <= Dd = 100.001 E + 12 u |100 lin~|.  An output is
// flagged when a half-integer lies within Dd of 100 lin~ and the table bytes on both sides differ (ties included; NaN /
// degenerate rows flag everything); l0_fix_list_kernel (fdnn_l0.hip) recomputes the flagged outputs with the exact chains.
// Measured on the 10 000-frame bench batch: DESIGN.md section 5, profiles/LABBOOK.md.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <type_traits>
#include <vector>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {

namespace {

#ifndef FDNN_L0S_DEBUG
#define FDNN_L0S_DEBUG 0  // kernel-ablation timing builds only (tools/build_variant.sh): 1 no sampling, 2 no lower-order MFMAs, 4 no screening arithmetic in the epilogue, 8 no staging in the loop, 16 no barrier in the loop, 32 no fragment reads in the loop
#endif
constexpr float kU = 5.9604645e-8f;  // 2^-24
constexpr int kSplitW = 32;          // steps between samples at most (one MFMA = 32 k)

// same workgroup -> tile map as the other layer-0 kernels (fdnn_l0.hip): all node tiles of a frame tile on one XCD
[[maybe_unused]] __device__ __forceinline__ bool split_tile_of_block(int node_tiles, int frame_tiles, int &bx, int &by) {
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  bx = slot % node_tiles;
  by = xcd + 8 * (slot / node_tiles);
  return by < frame_tiles;
}

// ---------------------------------------------------------------- pre-pass: frames -> digit planes + row constants
// One 256-thread workgroup per 8 frames (1280 workgroups for 10 240 frames: short dependent chains, many in flight).  The
// rows are staged (shift, scale applied: add, then multiply, dnn.cc:184-187) in LDS with an odd row stride, 32 lanes per
// row reduce max |x|, sum x^2 and sum |x|, then every (row, 16-position half chunk) item turns its 16 values into 3 x 16
// digit bytes.  Plane layout in memory = the MFMA fragment order, so that the matrix kernel's LDS-DMA copies are
// lane-linear on both sides and its fragment reads are lane-linear too:
//   xd[chunk][plane][row block of 32][half h][row r][16 bytes]  =  positions 32 chunk + 16 h + 0..15 of row 32 block + r
constexpr int kDigFrames = 8;
__global__ __launch_bounds__(256) void l0_digits_kernel(L0Params p, int KC, int J, int JP) {
  extern __shared__ __attribute__((aligned(16))) float dig_smem[];
  const int D = p.D, ld = D + 1;
  float *xs = dig_smem;                      // [8][D + 1]
  float *cf_s = dig_smem + kDigFrames * ld;  // [8]
  const int tid = threadIdx.x, f0 = blockIdx.x * kDigFrames;
  if (blockIdx.x == 0 && tid < 2) p.glist_count[tid] = 0u;  // this launch's list of flagged outputs starts empty
  const int quads = D >> 2;
  {
    // all of a thread's loads first (at most four 16-byte pieces of the rows, D <= 496), then the arithmetic: with a load,
    // its wait and its LDS stores per iteration the loop was four HBM round trips long (11 us per workgroup)
    float4 raw[4], sh[4], sc[4];
    int rowq[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 256;
      const int row = i / quads, q = i - row * quads, f = f0 + row;
      const bool in = i < kDigFrames * quads;
      rowq[it] = in ? row * ld + 4 * q : -1;
      raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      sh[it] = raw[it];
      sc[it] = raw[it];
      if (in && f < p.n) {
        raw[it] = *reinterpret_cast<const float4 *>(p.x + static_cast<size_t>(f) * D + 4 * q);
        sh[it] = *reinterpret_cast<const float4 *>(p.shift + 4 * q);
        sc[it] = *reinterpret_cast<const float4 *>(p.scale + 4 * q);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (rowq[it] < 0) continue;
      float *dst = xs + rowq[it];  // rows past the batch: zeros ((0 + 0) * 0)
      dst[0] = (raw[it].x + sh[it].x) * sc[it].x;  // ApplyShiftAndScale: add, then multiply (dnn.cc:184-187)
      dst[1] = (raw[it].y + sh[it].y) * sc[it].y;
      dst[2] = (raw[it].z + sh[it].z) * sc[it].z;
      dst[3] = (raw[it].w + sh[it].w) * sc[it].w;
    }
  }
  __syncthreads();
  {
    const int row = tid >> 5, sub = tid & 31;
    float mx = 0.0f, s2 = 0.0f, s1 = 0.0f;
    int bad = 0;
    float vals[16];  // (D <= 496: at most 16 per lane; all LDS reads go out before the first is used)
#pragma unroll
    for (int it = 0; it < 16; ++it) vals[it] = sub + 32 * it < D ? xs[row * ld + sub + 32 * it] : 0.0f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const float v = vals[it], a = fabsf(v);
      bad |= !(a < 3.0e38f);  // inf or NaN
      mx = fmaxf(mx, a);
      s2 = fmaf(v, v, s2);
      s1 += a;
    }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off));
      s2 += __shfl_xor(s2, off);
      s1 += __shfl_xor(s1, off);
      bad |= __shfl_xor(bad, off);
    }
    if (sub == 0) {
      // rows the integer image cannot represent to 2^-24 of their own scale take the exact path as a whole
      const bool degenerate = bad || mx > 1.0e18f || (mx != 0.0f && mx < 1.0e-18f);
      const float cf = (mx == 0.0f || degenerate) ? 1.0f : 8355000.0f / mx;
      cf_s[row] = degenerate ? 0.0f : cf;  // (digits of a degenerate row: zeros)
      const float fD = static_cast<float>(D);
      const float slack = 1.001f + 2.0f * kU * fD;
      // ||X||_1 <= c_f sum|x| + D  (|X_k| <= |x_k| c_f + 1),  ||X1||_1 <= ||X||_1 / 65536 + 0.502 D
      const float n1X = (cf * s1 * slack + fD) * 1.001f;
      const float n1X1 = n1X * (1.0f / 65536.0f) + 0.502f * fD;
      const float m = static_cast<float>(JP / 32 + 1);  // samples per chain (one per chunk) + slack
      const float a_f = kU * (kSplitW + 3) * m * 0.502f * n1X1 + 2.3283064e-10f * (0.5f * n1X + 0.5f * fD) + kU * 256.0f * fD;
      const int f = f0 + row;
      if (f < p.n_ld) {
        p.xstat[f] = 65536.0f / cf;                                       // r_f
        p.xstat[p.n_ld + f] = sqrtf(s2) * (1.00001f + 2.0f * kU * fD);    // ||x||_2, rounded up (as the fp32 screen did)
        p.xstat[2 * p.n_ld + f] = degenerate ? __builtin_inff() : 256.0f * a_f * 1.002f;  // (x 256: the kernel's sigma is 2^-8 sigma*)
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < kDigFrames * KC * 2; idx += 256) {
    const int row = idx & 7, hh = (idx >> 3) & 1, kc = idx >> 4;
    const float cf = cf_s[row];
    uint32_t d1[4], d2[4], d3[4];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      uint32_t w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pp = kc * 32 + hh * 16 + e4 * 4 + e;  // chain c owns positions c JP .. c JP + J - 1, zeros up to (c + 1) JP
        const int c = (pp >= JP) + (pp >= 2 * JP) + (pp >= 3 * JP), j = pp - c * JP;
        const float v = j < J ? xs[row * ld + 4 * j + c] : 0.0f;
        const int X = static_cast<int>(rintf(v * cf));  // |X| <= 8355000 (1 + u) + 0.5
        const int x3 = (X << 24) >> 24, r = (X - x3) >> 8, x2 = (r << 24) >> 24, x1 = (r - x2) >> 8;
        w1 |= static_cast<uint32_t>(x1 & 0xff) << (8 * e);
        w2 |= static_cast<uint32_t>(x2 & 0xff) << (8 * e);
        w3 |= static_cast<uint32_t>(x3 & 0xff) << (8 * e);
      }
      d1[e4] = w1; d2[e4] = w2; d3[e4] = w3;
    }
    const int f = f0 + row;
    if (f >= p.n_ld) continue;
    const size_t blocks = static_cast<size_t>(p.n_ld >> 5);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const uint32_t *d = pl == 0 ? d1 : pl == 1 ? d2 : d3;
      char *base = reinterpret_cast<char *>(p.xd) + ((static_cast<size_t>(kc) * 3 + pl) * blocks + (f >> 5)) * 1024 + hh * 512 + (f & 31) * 16;
      *reinterpret_cast<uint4 *>(base) = make_uint4(d[0], d[1], d[2], d[3]);
    }
  }
}

// ---------------------------------------------------------------- the matrix kernel
// 512 threads = 4 x 2 waves (frames x nodes), tile 128 frames x 128 nodes, wave tile 32 frames x 64 nodes = two
// 32 x 32 MFMA tiles; per 32-position chunk a wave reads 3 frame fragments + 6 node fragments (into the register set
// the previous chunk is not using) and issues 12 MFMAs.  Registers per lane: P0, P1, P2 and A, 32 each, + 2 x 36 of
// fragments.  A finished chain's P0 is folded into P1 as 256 P0 (exact in int32 for D <= 496), so no fourth set.
// Chains are padded to whole chunk pairs, so a chunk never straddles a chain boundary and the chunk body is branch free.
// Staging: per chunk 24 lane-linear 1-KiB LDS-DMA pieces (3 planes x 4 row blocks, both operands), three per wave,
// 3-stage ring, one barrier per chunk -- placed between the chunk's two MFMA groups, so that the matrix pipe has work
// while the waves meet, issue the next pieces and fetch the next fragments.
constexpr int kSTF = 128, kSStages = 3;
template <int WN>  // node waves: 2 = 128-node tiles, 512 threads, one workgroup per CU; 1 = 64-node tiles, 256 threads, two per CU
struct SplitCfg {
  static constexpr int TN = 64 * WN, NW = 4 * WN, THREADS = 64 * NW;
  static constexpr int PIECES = 12 + 6 * WN;           // 1-KiB pieces per stage: 3 planes x 4 frame blocks + 3 planes x 2 WN node blocks
  static constexpr int STAGE = PIECES * 1024;
  static constexpr int RING = STAGE * kSStages;
  static constexpr int HALF_OFF = RING;                // half-step table: u32 [kLut2Size] (+ pad), 11 KiB
  static constexpr int STAT_OFF = RING + 11 * 1024;    // three 1-KiB slots: r_f, ||x||_2, a_f of the tile's 128 frames (512 bytes each + the DMA piece's zero tail)
  static constexpr int LDS = STAT_OFF + 3 * 1024;
  static constexpr int TS = TN + 16;                   // byte tile row stride
  static constexpr int PPW = (PIECES + NW - 1) / NW;   // pieces per wave and stage, at most
  static_assert(kSTF * TS + 16 + 2 * kL0ScreenCap <= RING, "epilogue tile and flag list must fit in the dead ring");
};
static_assert(4 * kLut2Size <= 11 * 1024, "half-step table area");

template <int WN, bool DBG = false>  // DBG (parity tests only: fdnn_debug_layer0_screen): t~ and Dd of every output leave as well
__global__ __launch_bounds__(256 * WN, 3 - WN) void l0_split_kernel(L0Params p, int KC, int CPC) {  // CPC: chunks per chain (even)
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = SplitCfg<WN>;
  constexpr int kSTN = Cfg::TN, kSStage = Cfg::STAGE, kSTS = Cfg::TS, kSHalfOff = Cfg::HALF_OFF, kSStatOff = Cfg::STAT_OFF, NW = Cfg::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int wf = wave & 3, wn = wave >> 2;
  const int node_tiles = p.h_ld / kSTN, frame_tiles = (p.n_rows + kSTF - 1) / kSTF;
  int bx, by;
  if (!split_tile_of_block(node_tiles, frame_tiles, bx, by)) return;
  const int f0 = by * kSTF, n0 = bx * kSTN;
  const int xblocks = p.n_ld >> 5, wblocks = p.h_ld >> 5;
#ifdef FDNN_L0S_CLK
  long long tc[12];
  tc[0] = __builtin_readcyclecounter();
#define L0S_TS(i) tc[i] = __builtin_readcyclecounter()
#else
#define L0S_TS(i)
#endif

  {  // half-step table and this tile's frame constants into the aux area (LDS-DMA, ahead of the ring): 11 + 3 pieces
    const __amdgpu_buffer_rsrc_t rsrc_half =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.luthalf), 0, (4 * kLut2Size + 15) & ~15, 0x00020000);
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      if (i % NW != wave) continue;
      if (i < 11) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_half, FDNN_LDS_PTR(smem + kSHalfOff + i * 1024), 16, lane * 16, i * 1024, 0, 0);
      } else {  // 128 floats = 512 bytes per constant: lanes 0..31 carry them, lanes 32..63 read past num_records: zeros into the slot's own tail
        const int q = i - 11;
        const __amdgpu_buffer_rsrc_t rsrc_st =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.xstat + static_cast<size_t>(q) * p.n_ld + f0), 0, kSTF * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_st, FDNN_LDS_PTR(smem + kSStatOff + q * 1024), 16, lane * 16, 0, 0, 0);
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.xd), 0, static_cast<unsigned>(KC) * 3u * static_cast<unsigned>(xblocks) * 1024u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(p.wd), 0, static_cast<unsigned>(KC) * 3u * static_cast<unsigned>(wblocks) * 1024u, 0x00020000);
  const int voff = lane * 16;
  // my pieces of a stage: piece wave + NW t.  X pieces 0..11 = plane i / 4, frame block i % 4; W pieces 12.. = plane j / (2 WN),
  // node block j % (2 WN).  (WN = 1: 18 pieces over 4 waves -- waves 0, 1 carry five, waves 2, 3 four.)
  const int my_pieces = (Cfg::PIECES - wave + NW - 1) / NW;
  auto stage_piece = [&](int kc, int buf, int t) {
    char *base = smem + buf * kSStage;
    const int i = wave + NW * t;
    if (i >= Cfg::PIECES) return;
    if (i < 12) {
      const int pl = i >> 2, q = i & 3;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, FDNN_LDS_PTR(base + i * 1024), 16, voff, ((kc * 3 + pl) * xblocks + (f0 >> 5) + q) * 1024, 0, 0);
    } else {
      const int j = i - 12, pl = j / (2 * WN), q = j % (2 * WN);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(base + i * 1024), 16, voff, ((kc * 3 + pl) * wblocks + (n0 >> 5) + q) * 1024, 0, 0);
    }
  };
  auto stage = [&](int kc, int buf) {
#pragma unroll
    for (int t = 0; t < Cfg::PPW; ++t) stage_piece(kc, buf, t);
  };
  // wait until at most `stages` of my stages are still in flight
  auto wait_stages = [&](int stages) {
    if (stages == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else if (Cfg::PIECES % NW == 0) {
      if (stages == 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Cfg::PPW) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * Cfg::PPW) : "memory");
    } else if (my_pieces == Cfg::PPW) {
      if (stages == 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Cfg::PPW) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * Cfg::PPW) : "memory");
    } else {
      if (stages == 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Cfg::PPW - 1) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (Cfg::PPW - 1)) : "memory");
    }
  };
  stage(0, 0);
  stage(1, 1);
  stage(2, 2);
  asm volatile("" ::: "memory");

  // P0 lives at an offset of 2^30: every partial sum of a chain (|.| <= 2^14 D) is then a positive integer, and |P0 - 2^30|
  // joins A in ONE instruction (v_sad_u32: unsigned |a - b| + c).  The offset costs nothing at a chain's end: 2^30 << 8 = 0 mod 2^32.
  constexpr int kP0Off = 0x40000000;
  v16i P0[2], P1[2], P2[2];
  uint32_t A[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      P0[s][r] = kP0Off; P1[s][r] = 0; P2[s][r] = 0;
      A[s][r] = 0u;
    }
  auto chain_end = [&](int s) {  // the finished chain's total joins P1 at its weight 2^8 (int32, exact); the next chain starts from zero
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      P1[s][r] += P0[s][r] << 8;  // (the offset drops out: 2^30 << 8 = 2^38)
      P0[s][r] = kP0Off;
    }
  };
  v4i xa[2][3], wb[2][3][2];  // fragment sets, double buffered over the chunks
  auto load_plane = [&](int buf, int set, int pl) {  // (set, pl are literals at every call site: the lambdas are inlined)
    const char *sb = smem + buf * kSStage;
    xa[set][pl] = *reinterpret_cast<const v4i *>(sb + (pl * 4 + wf) * 1024 + lane * 16);
#pragma unroll
    for (int s = 0; s < 2; ++s) wb[set][pl][s] = *reinterpret_cast<const v4i *>(sb + 12288 + (pl * 2 * WN + wn * 2 + s) * 1024 + lane * 16);
  };
  auto load_frags = [&](int buf, int set) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) load_plane(buf, set, pl);
  };
  // The samples of node half ss (its P0 is complete: one MFMA group ago) go into the issue slots between the MFMAs of
  // half s -- fenced, or the scheduler runs the 16 vector instructions before or after the MFMAs instead of beside them.
  auto sample_part = [&](int ss, int r0, int r1, bool on) {
    if ((FDNN_L0S_DEBUG & 1) || !on) return;
    // inline asm (no builtin for v_sad_u32; 2.0 is the inline constant 0x40000000 = the offset).  The compiler's hazard
    // recognizer does not see an MFMA result being read here: P0[ss] was written one whole MFMA group (six MFMAs and a
    // barrier or the loop edge) ago, far beyond the 18 wait states a 16-pass MFMA needs.
#pragma unroll
    for (int r = r0; r < r1; ++r) asm volatile("v_sad_u32 %0, %1, 2.0, %0" : "+v"(A[ss][r]) : "v"(P0[ss][r]));
  };
#define FDNN_L0S_FENCE __builtin_amdgcn_sched_barrier(0)
  // (x0 .. x4: what else goes between the MFMAs -- the refill pieces and the next chunk's fragment reads in the second
  // group of a chunk: an LDS-DMA piece costs its wave 60-180 cycles of issue, which the matrix pipe spends on the other
  // wave's MFMA instead of idling behind the barrier)
  auto mfma_group = [&](int set, int s, int ss, bool on, auto x0, auto x1, auto x2, auto x3, auto x4) {
    FDNN_L0S_FENCE;
    P0[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][0], wb[set][0][s], P0[s], 0, 0, 0);
    if (FDNN_L0S_DEBUG & 2) {
      sample_part(ss, 0, 16, on);
      x0(); x1(); x2(); x3(); x4();
      return;
    }
    FDNN_L0S_FENCE;
    sample_part(ss, 0, 3, on);
    x0();
    FDNN_L0S_FENCE;
    P1[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][0], wb[set][1][s], P1[s], 0, 0, 0);
    FDNN_L0S_FENCE;
    sample_part(ss, 3, 6, on);
    x1();
    FDNN_L0S_FENCE;
    P2[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][0], wb[set][2][s], P2[s], 0, 0, 0);
    FDNN_L0S_FENCE;
    sample_part(ss, 6, 9, on);
    x2();
    FDNN_L0S_FENCE;
    P1[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][1], wb[set][0][s], P1[s], 0, 0, 0);
    FDNN_L0S_FENCE;
    sample_part(ss, 9, 12, on);
    x3();
    FDNN_L0S_FENCE;
    P2[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][1], wb[set][1][s], P2[s], 0, 0, 0);
    FDNN_L0S_FENCE;
    sample_part(ss, 12, 16, on);
    x4();
    FDNN_L0S_FENCE;
    P2[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa[set][2], wb[set][0][s], P2[s], 0, 0, 0);
    FDNN_L0S_FENCE;
  };
  auto nothing = [] {};

  wait_stages(2);  // chunk 0 (and the aux pieces before it) landed; chunks 1, 2 may be in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(0, 0);
  L0S_TS(1);
  // One chunk: [MFMA group of node half 0 + the previous chunk's samples of half 1] -- meet, refill, next fragments --
  // [MFMA group of half 1 + this chunk's samples of half 0].  At the first chunk of a later chain the finished chain's P0
  // joins P1: half 0 before its MFMA, half 1 after its last sample.
  auto chunk = [&](int kc, auto set_c, auto first_c, bool chain_start) {
    constexpr int set = decltype(set_c)::value;  // (compile time: the fragment sets are registers)
    constexpr bool first = decltype(first_c)::value;
    if (chain_start) chain_end(0);  // (wave-uniform, three times per tile)
    mfma_group(set, 0, 1, !first, nothing, nothing, nothing, nothing, nothing);
    const bool more = kc + 1 < KC, refill = kc + 3 < KC && !(FDNN_L0S_DEBUG & 8);
    if (more) {
      wait_stages(kc + 2 < KC ? 1 : 0);  // my pieces of chunk kc + 2 may still be in flight
      if (!(FDNN_L0S_DEBUG & 16)) __builtin_amdgcn_s_barrier();  // chunk kc + 1 landed for everyone, everyone has read chunk kc (its fragments are in registers)
      asm volatile("" ::: "memory");
    }
    if (chain_start) chain_end(1);
    // second group: the refill of the buffer chunk kc has just left (chunk kc + 3) and the fragments of chunk kc + 1 go
    // out between its MFMAs
    const int nb = (kc + 1) % 3;
    mfma_group(
        set, 1, 0, true, [&] { if (more && !(FDNN_L0S_DEBUG & 32)) load_plane(nb, set ^ 1, 0); if (refill) stage_piece(kc + 3, kc % 3, 0); },
        [&] { if (more && !(FDNN_L0S_DEBUG & 32)) load_plane(nb, set ^ 1, 1); if (refill) stage_piece(kc + 3, kc % 3, 1); },
        [&] { if (more && !(FDNN_L0S_DEBUG & 32)) load_plane(nb, set ^ 1, 2); if (refill) stage_piece(kc + 3, kc % 3, 2); },
        [&] { if (refill && Cfg::PPW > 3) stage_piece(kc + 3, kc % 3, 3); }, [&] { if (refill && Cfg::PPW > 4) stage_piece(kc + 3, kc % 3, 4); });
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  chunk(0, C0{}, std::true_type{}, false);
  chunk(1, C1{}, std::false_type{}, false);
  for (int kc = 2, in_chain = 2; kc < KC; kc += 2, in_chain += 2) {  // (KC and CPC are even)
    const bool chain_start = in_chain == CPC;
    if (chain_start) in_chain = 0;
    chunk(kc, C0{}, std::false_type{}, chain_start);
    chunk(kc + 1, C1{}, std::false_type{}, false);
  }
  sample_part(1, 0, 16, true);  // (half 1's samples trail by one group; the last MFMA group is long done: the epilogue's barrier follows)
  chain_end(0);
  chain_end(1);
  L0S_TS(2);

  // ------------------------------------------------------------ epilogue
  __syncthreads();  // the ring is dead: byte tile + flag list
  const char *half_b = smem + kSHalfOff + 4 * kLut2Half;  // entry 0 of the half-step table
  const float *rf_s = reinterpret_cast<const float *>(smem + kSStatOff);
  const float *xn_s = rf_s + 256, *af_s = rf_s + 512;
  uint8_t *tile = reinterpret_cast<uint8_t *>(smem);
  uint32_t *scr_n = reinterpret_cast<uint32_t *>(tile + kSTF * kSTS);
  uint16_t *scr_l = reinterpret_cast<uint16_t *>(tile + kSTF * kSTS + 16);
  if (tid == 0) *scr_n = 0;
  const float fD = static_cast<float>(p.D);
  const float c2 = 1.001f * (0.5f * fD * fD + 2.0f * fD) * kU;  // second order in u over the D steps (as the fp32 screen)
  // everything below works on t = 100 lin: sig100 = 100 sigma, and the bound's constants carry the 100.001 / 100 = 1.00001
  const float kA = 1.00001f * 256.0f * 1.002f * kU * (kSplitW + 3);  // (x 256: sigma is 2^-8 sigma*)
  const float kS = 100.001f * 1.002f * kU * (kSplitW / 2 + 2 + c2);
  // this lane's node constants (two columns)
  float bias100[2], rn100[2], kSw2[2], bn2[2], eb2[2];
  bool node_in[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int node = n0 + wn * 64 + s * 32 + l32;
    node_in[s] = node < p.H;
    const float b = node_in[s] ? p.bias[node] : 0.0f;
    bias100[s] = 100.0f * b;
    rn100[s] = 100.0f * (node_in[s] ? p.wstat[node] : 0.0f);
    kSw2[s] = kS * (node_in[s] ? p.wstat[p.h_ld + node] : 0.0f);
    bn2[s] = 1.00001f * (node_in[s] ? p.wstat[2 * p.h_ld + node] : 0.0f);
    eb2[s] = 100.001f * 10.0f * kU * fabsf(b);
  }
  // which of this lane's 32 outputs exist (frame inside the batch, node inside the layer): applied to the flag bits at the end
  uint32_t valid = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool row_in = f0 + wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3) < p.n;
    valid |= (row_in && node_in[0] ? 1u : 0u) << r;
    valid |= (row_in && node_in[1] ? 1u : 0u) << (16 + r);
  }
  // pass 1: t = 100 lin~ and sigma for all 32 outputs, and their table entries requested -- all the gathers are in flight
  // before the first is needed (one gather and its wait per output was 32 exposed LDS round trips per wave)
  float tt[2][16], sg[2][16];
  uint32_t ent[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float rf = rf_s[wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3)];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // V = 2^-8 (P1' + 2^-8 P2), P1' = 256 P0 + P1 in int32; sig = 100 * 2^24 / (c_f c_n);  t ~ 100 lin (dnn.h:37)
      const float v = fmaf(static_cast<float>(P2[s][r]), 0.00390625f, static_cast<float>(P1[s][r]));
      sg[s][r] = rf * rn100[s];
      tt[s][r] = fmaf(v, sg[s][r], bias100[s]);
      // QuantizedSigmoid::get through the half-step table (fdnn_model.cpp): index trunc(2 t), clamped; the entry carries the
      // table byte and, in its upper half, a gate: 0.25f where the byte on the other side of the nearest half-integer is
      // the same, else 0
      const int u = max(-kLut2Half, min(kLut2Half, static_cast<int>(tt[s][r] + tt[s][r])));  // (v_cvt_i32_f32 truncates, saturates, NaN -> 0)
      ent[s][r] = *reinterpret_cast<const uint32_t *>(half_b + 4 * u);
    }
  }
  // pass 2: the table bytes into the tile, and the screening
  uint32_t scr_mask = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
    const float xn = xn_s[row], af = af_s[row];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float t = tt[s][r], sig = sg[s][r];
      tile[row * kSTS + wn * 64 + s * 32 + l32] = static_cast<uint8_t>(ent[s][r]);
      if (!(FDNN_L0S_DEBUG & 4)) {
        const float e = __builtin_amdgcn_fractf(t) - 0.5f;  // against the half-integer above floor(t): the nearest one
        const float Dd = fmaf(fabsf(t), 12.0f * kU, fmaf(sig, fmaf(kA, static_cast<float>(A[s][r]), af + bn2[s]), fmaf(kSw2[s], xn, eb2[s]))) + 1e-30f;
        // flagged: a half-integer within Dd of t, AND (different bytes on its two sides OR Dd >= 0.25: several boundaries in
        // reach), AND not both clamped to the same end of the table (|t| - Dd >= 641)  ==  max(|e|, gate, |t| - 641) <= Dd,
        // written so that a NaN flags.  x86 float -> int turns NaN and |t| >= 2^31 into INT_MIN (entry 0 after the clamp,
        // lut_index): anything near that takes the exact path as well.
        const float gate = __builtin_bit_cast(float, ent[s][r] & 0xffff0000u);
        const float m = fmaxf(fmaxf(fabsf(e), gate), fabsf(t) - 641.0f);
        const bool flag = !(m > Dd) | !(fabsf(t) < 1.0e9f);
        scr_mask |= flag ? (1u << (16 * s + r)) : 0u;
        if (DBG && ((valid >> (16 * s + r)) & 1u)) {
          const size_t o = static_cast<size_t>(f0 + row) * p.H + (n0 + wn * 64 + s * 32 + l32);
          p.dbg_t[o] = t;
          p.dbg_dd[o] = Dd;
        }
      }
    }
  }
  scr_mask &= valid;
  L0S_TS(3);
  __syncthreads();  // (scr_n = 0 is visible; the byte tile is complete)
  L0S_TS(4);
  {  // the wave reserves room for all its flagged outputs with ONE LDS atomic, every lane then writes its own entries
    const int mine = __popc(scr_mask);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const int wave_total = __shfl(incl, 63);
    if (wave_total) {
      uint32_t base = 0;
      if (lane == 63) base = atomicAdd(scr_n, static_cast<uint32_t>(wave_total));
      base = __shfl(base, 63);
      uint32_t at = base + static_cast<uint32_t>(incl - mine);
      uint32_t m = scr_mask;
      while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        const int r = i & 15, sb2 = i >> 4;
        const int row = wf * 32 + 8 * (r >> 2) + 4 * h + (r & 3), col = wn * 64 + sb2 * 32 + l32;
        if (at < static_cast<uint32_t>(kL0ScreenCap)) scr_l[at] = static_cast<uint16_t>(row * kSTN + col);
        ++at;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kSTF * (kSTN / 16) / Cfg::THREADS; ++q) {  // the byte tile leaves as whole row segments
    const int item = tid + q * Cfg::THREADS, row = item / (kSTN / 16), c16 = (item % (kSTN / 16)) * 16;
    const int f = f0 + row;
    if (f < p.n_rows && n0 + c16 < p.H)  // H is a multiple of 16
      *reinterpret_cast<uint4 *>(p.act_out + static_cast<size_t>(f) * p.act_ld + n0 + c16) = *reinterpret_cast<const uint4 *>(tile + row * kSTS + c16);
  }
  __syncthreads();
  {  // the tile's entries join the launch's list: one global atomic per tile reserves the room
    const uint32_t cnt = *scr_n;
    uint32_t *gbase_s = scr_n + 1;
    if (tid == 0) {
      uint32_t gb = 0xffffffffu, fill_from = 0xffffffffu;
      if (cnt != 0 && cnt <= static_cast<uint32_t>(kL0ScreenCap)) {
        // One add reserves [gb, gb + cnt).  A tile that no longer fits takes the whole-tile path -- and fills what it reserved
        // below the capacity with entries no output matches ({~0, ~0}: the fix kernel's range test drops them), so that no
        // walked entry is ever left unwritten (round-4 advisor finding: such entries were stale memory).  (A compare-and-swap
        // loop that reserves only what fits serialises the 256 tiles that finish together: 97 -> 1500 us per launch, measured.)
        gb = atomicAdd(p.glist_count, cnt);
        if (gb + cnt > static_cast<uint32_t>(p.glist_cap)) {
          fill_from = gb;
          gb = 0xffffffffu;
        }
      }
      if (cnt != 0 && gb == 0xffffffffu) {  // too many for either list: the fix kernel recomputes the whole tile
        p.scr_count[by * (p.h_ld / 128) + (n0 >> 7)] = cnt;  // (the whole-tile path works on 128 x 128 tiles)
        atomicAdd(p.glist_count + 1, 1u);
      }
      *gbase_s = gb;
      gbase_s[1] = fill_from;
    }
    __syncthreads();
    const uint32_t gb = *gbase_s, fill_from = gbase_s[1];
    if (fill_from < static_cast<uint32_t>(p.glist_cap))  // (rare) my reservation straddles the end of the list
      for (uint32_t i = fill_from + tid; i < static_cast<uint32_t>(p.glist_cap); i += Cfg::THREADS) p.glist[i] = make_uint2(0xffffffffu, 0xffffffffu);
    if (gb != 0xffffffffu)
      for (uint32_t i = tid; i < cnt; i += Cfg::THREADS) {
        const uint32_t e = scr_l[i];
        p.glist[gb + i] = make_uint2(static_cast<uint32_t>(f0) + e / kSTN, static_cast<uint32_t>(n0) + e % kSTN);
      }
  }
#ifdef FDNN_L0S_CLK
  L0S_TS(5);
  if (p.scr_stats && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3)) {
    unsigned long long *o = p.scr_stats + 4 + (blockIdx.x == 0 ? 0 : 12);
    for (int i = 0; i < 6; ++i) o[i] = static_cast<unsigned long long>(tc[i] - tc[0]);
  }
#endif
#endif
}

}  // namespace

// D <= 496: 256 P0 + P1 stays inside int32 (2^22 D + 2^15 D < 2^31) and the pre-pass rows fit its LDS
bool l0_split_ok(int D, int H) { return D >= 64 && D <= 496 && (D & 3) == 0 && (H & 15) == 0; }
// chains are padded to whole chunk PAIRS (the kernel alternates two fragment register sets): 432 -> 4 x 128 positions, 16 chunks
int l0_split_chain_pad(int D) { return (D / 4 + 63) / 64 * 64; }
int l0_split_chunks(int D) { return 4 * l0_split_chain_pad(D) / 32; }
size_t l0_split_plane_bytes(int D, int rows_ld) { return static_cast<size_t>(l0_split_chunks(D)) * 3 * static_cast<size_t>(rows_ld / 32) * 1024; }
size_t l0_split_half_bytes() { return (4 * size_t(kLut2Size) + 15) & ~size_t(15); }

// Model load: the node half.  w [H][D] fp32 -> digit planes in fragment order + per-node constants {r_n, ||w||_2 (the
// caller's), b_n}, and the table as pairs.  Host code, double arithmetic: w c_n is exact in double, so |w c_n - W| <= 0.5.
void l0_split_build_weights(const float *w, const float *wnorm, const uint8_t *lut2, int H, int D, int h_ld, std::vector<int8_t> *planes,
                            std::vector<float> *stat, std::vector<uint32_t> *half) {
  const int J = D / 4, JP = l0_split_chain_pad(D);
  planes->assign(l0_split_plane_bytes(D, h_ld), 0);
  stat->assign(static_cast<size_t>(3) * h_ld, 0.0f);
  // The blob's half-step table (index trunc(2 t), fdnn_model.cpp) widened to 4 bytes per entry: the table byte, and in the
  // upper half the float bits of the gate -- 0.25f where the entry on the other side of the nearest half-integer
  // (|u| ^ 1 with the sign of u; both neighbours for u = 0) holds the same byte, else 0.
  half->assign(l0_split_half_bytes() / 4, 0);
  auto at = [&](int u) { return lut2[std::max(-kLut2Half, std::min(kLut2Half, u)) + kLut2Half]; };
  for (int u = -kLut2Half; u <= kLut2Half; ++u) {
    const int mag = u < 0 ? -u : u;
    bool same;
    if (u == 0)
      same = at(0) == at(1) && at(0) == at(-1);
    else
      same = at(u) == at(u < 0 ? -(mag ^ 1) : (mag ^ 1));
    (*half)[u + kLut2Half] = at(u) | (same ? 0x3e800000u : 0u);  // 0.25f = 0x3e800000
  }
  const size_t blocks = static_cast<size_t>(h_ld / 32);
  const double u = std::ldexp(1.0, -24);
  const double m = static_cast<double>(JP / 32 + 1);
  for (int n = 0; n < H; ++n) {
    const float *row = w + static_cast<size_t>(n) * D;
    double mx = 0.0;
    bool bad = false;
    for (int k = 0; k < D; ++k) {
      const double a = std::fabs(static_cast<double>(row[k]));
      if (!(a < 3.0e38)) bad = true;
      if (a > mx) mx = a;
    }
    const bool degenerate = bad || mx > 1.0e18 || (mx != 0.0 && mx < 1.0e-18);
    const float cn = (mx == 0.0 || degenerate) ? 1.0f : static_cast<float>(8355000.0 / mx);
    double n1W = 0.0, n1W2 = 0.0, n1W3 = 0.0;
    for (int pp = 0; pp < 4 * JP; ++pp) {
      const int c = pp / JP, j = pp - c * JP, k = 4 * j + c;
      if (j >= J) continue;  // (pad positions stay zero)
      const long W = degenerate ? 0 : std::lrint(static_cast<double>(row[k]) * static_cast<double>(cn));
      n1W += static_cast<double>(W < 0 ? -W : W);
      const int x3 = static_cast<int8_t>(W & 0xff), r = static_cast<int>((W - x3) >> 8), x2 = static_cast<int8_t>(r & 0xff), x1 = (r - x2) >> 8;
      n1W2 += std::abs(x2);
      n1W3 += std::abs(x3);
      const int digit[3] = {x1, x2, x3};
      const int kc = pp >> 5, hh = (pp >> 4) & 1, e = pp & 15;
      for (int pl = 0; pl < 3; ++pl)
        (*planes)[((static_cast<size_t>(kc) * 3 + pl) * blocks + (n >> 5)) * 1024 + hh * 512 + (n & 31) * 16 + e] = static_cast<int8_t>(digit[pl]);
    }
    // sampling resolution + quantisation + the dropped orders: |P3| <= 128 (||W2||_1 + ||W3||_1) (|X2|, |X3| <= 128), |P4| <= 128 ||W3||_1
    const double b_n = (u * (kSplitW + 3) * m * 0.502 * n1W / 65536.0 + std::ldexp(0.5 * n1W, -32) + std::ldexp(128.0 * (n1W2 + n1W3), -24) +
                        std::ldexp(128.0 * n1W3, -32)) * 1.002;
    auto up = [](double v) {
      float f = static_cast<float>(v);
      if (static_cast<double>(f) < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
      return f;
    };
    (*stat)[n] = 256.0f / cn;  // sigma = r_f r_n = 2^24 / (c_f c_n) = 2^-8 sigma*
    (*stat)[static_cast<size_t>(h_ld) + n] = wnorm[n];
    (*stat)[static_cast<size_t>(2) * h_ld + n] = degenerate ? std::numeric_limits<float>::infinity() : up(256.0 * b_n);
  }
}

// pre-pass + matrix kernel; the caller (fdnn_l0.hip: launch_l0) follows with l0_fix_kernel on the same tile lists
void launch_l0_split(const L0Params &p, hipStream_t s) {
  const int KC = l0_split_chunks(p.D), J = p.D / 4, JP = l0_split_chain_pad(p.D);
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  const int dig_lds = (kDigFrames * (p.D + 1) + kDigFrames) * 4;
  // 128-node tiles, one 512-thread workgroup per CU; batches so small that those would leave half the chip idle (up to 128
  // tiles: 1 024 frames on a 2048-node layer) take 64-node tiles, twice as many workgroups of half the size.  Measured
  // equal both where both fill the chip (99.2 vs 97.7 us at 10 000 frames) and below (layer 0 at 1 000 frames 40.2 vs 40.4 us:
  // one tile's latency -- 16 chunks and a 32-output-per-lane epilogue per wave -- either way).  FDNN_L0S_WN=1|2 forces one.
  static const int wn_forced = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_L0S_WN");
    return e ? std::atoi(e) : 0;
  }();
  const int tiles128 = ((p.n_rows + kSTF - 1) / kSTF) * (p.h_ld / 128);
  const int wn_cfg = wn_forced == 1 || wn_forced == 2 ? wn_forced : (tiles128 <= 128 ? 1 : 2);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(l0_split_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, SplitCfg<1>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(l0_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, SplitCfg<2>::LDS);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  const int frame_tiles = (p.n_rows + kSTF - 1) / kSTF;
  const int dig_rows = frame_tiles * kSTF;  // every row a matrix tile will read (<= n_ld)
  hipLaunchKernelGGL(l0_digits_kernel, dim3(dig_rows / kDigFrames), dim3(256), dig_lds, s, p, KC, J, JP);
  if (p.dbg_t != nullptr && p.dbg_dd != nullptr) {  // the tests' instance (128-node tiles)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(l0_split_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SplitCfg<2>::LDS);
    hipLaunchKernelGGL((l0_split_kernel<2, true>), dim3(static_cast<unsigned>(p.h_ld / 128) * ((frame_tiles + 7) / 8) * 8), dim3(512), SplitCfg<2>::LDS, s, p, KC, JP / 32);
    return;
  }
  if (wn_cfg == 2)
    hipLaunchKernelGGL(l0_split_kernel<2>, dim3(static_cast<unsigned>(p.h_ld / 128) * ((frame_tiles + 7) / 8) * 8), dim3(512), SplitCfg<2>::LDS, s, p, KC, JP / 32);
  else
    hipLaunchKernelGGL(l0_split_kernel<1>, dim3(static_cast<unsigned>(p.h_ld / 64) * ((frame_tiles + 7) / 8) * 8), dim3(256), SplitCfg<1>::LDS, s, p, KC, JP / 32);
}

}  // namespace fdnn
