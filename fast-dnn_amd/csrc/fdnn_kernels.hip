// fdnn_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the quantized
// feed-forward scorer.  Built with -ffp-contract=off: every float operation
// below rounds exactly where the reference's scalar/SSE code rounds, fused
// multiply-adds appear only where written as fmaf().
//
// Path (reference -> kernel):
//   ApplyShiftAndScale + InputActivations + AddBias + QuantizedSigmoid
//     (dnn.cc:175-192, :219-286)                      -> l0_kernel
//   QuantizedLayerActivations/quantizedNodeSum + AddBias + QuantizedSigmoid
//     (dnn.cc:289-349, :250-286)                      -> qgemm_kernel<.., OUTPUT=false>
//   CalculateOutput / LazyOutputActivations + SoftMax
//     (dnn.cc:428-454, :355-392, :534-544)            -> qgemm_kernel<.., OUTPUT=true> + normalize_kernel
//   pmaddubsw int16 pair saturation (dnn.cc:337-340)  -> sparse exact correction in the GEMM epilogue
//
// u8 x s8 on signed MFMA: activations travel as s8 = u8 - 128 (bit 7 flipped),
// so sum_k u8*w = sum_k s8*w + 128*sum_k w; the second term is a per-node int32
// precomputed at load.  Exact: |sum| <= 2^15 * 255 * 128 < 2^31.
#include "fdnn_kernels.hpp"

#include <climits>

#include "fdnn_model.hpp"

namespace fdnn {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define FDNN_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define FDNN_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// 16 bytes per lane straight from global memory into LDS at
// wave-uniform base + lane*16 (global_load_lds_dwordx4).
__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds(FDNN_GLOBAL_PTR(g), FDNN_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// float(sum) / (multiplier * 255)   -- dnn.cc:298-299, :311.  The fast form is
// the Markstein sequence q = x*y, r = fma(-q, c, x), q' = fma(r, y, q) with
// y = RN(1/c); it is enabled per layer only after launch_fastdiv_check has
// compared it with IEEE division for every possible accumulator.
__device__ __forceinline__ float dequant(int acc, float coef, float rcp, int fast) {
  const float x = static_cast<float>(acc);  // v_cvt_f32_i32, RNE like cvtsi2ss
  if (fast) {
    const float q = x * rcp;
    const float r = fmaf(-q, coef, x);
    return fmaf(r, rcp, q);
  }
  return x / coef;
}

// QuantizedSigmoid::get -- dnn.h:36-43: k = (int)round(x*100), table index
// clamp(k,-640,640)+640 into the extended table.  round() is half away from
// zero; the x86 build turns NaN / |t| >= 2^31 into INT_MIN (-> entry 0).
__device__ __forceinline__ int lut_index(float lin) {
  const float t = lin * 100.0f;
  float r = truncf(t);
  if (fabsf(t - r) >= 0.5f) r += copysignf(1.0f, t);
  int k = (fabsf(t) < 2147483648.0f) ? static_cast<int>(r) : INT_MIN;
  k = max(-kLutHalf, min(kLutHalf, k));
  return k + kLutHalf;
}

// ---------------------------------------------------------------- layer 0 (fp32, order-faithful)
constexpr int L0_TF = 64;   // frames per block
constexpr int L0_TN = 64;   // nodes per block
constexpr int L0_BK = 16;   // k per LDS chunk
constexpr int L0_LD = 20;   // padded LDS row (floats): conflict-free ds_read_b128 for rows tx+16j

template <bool FMA, bool TAP>
__global__ __launch_bounds__(256, 4) void l0_kernel(L0Params p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * L0_TF * L0_LD + 2 * L0_TN * L0_LD + (kLutExt + 15) / 4 + 4];
  float *xs = smem;                               // [2][64][20]
  float *ws = smem + 2 * L0_TF * L0_LD;           // [2][64][20]
  uint8_t *lut = reinterpret_cast<uint8_t *>(smem + 2 * L0_TF * L0_LD + 2 * L0_TN * L0_LD);
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int f0 = blockIdx.y * L0_TF, n0 = blockIdx.x * L0_TN;
  for (int i = tid; i < kLutExt; i += 256) lut[i] = p.lut[i];

  // each thread stages one float4 of x and one of w per chunk
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const int xf = f0 + lrow, wn_ = n0 + lrow;
  const bool xok = xf < p.n, wok = wn_ < p.H;
  const float *xrow = p.x + static_cast<size_t>(xok ? xf : 0) * p.D;
  const float *wrow = p.w + static_cast<size_t>(wok ? wn_ : 0) * p.D;

  float acc[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int l = 0; l < 4; ++l) acc[i][j][l] = 0.0f;

  const int nchunk = (p.D + L0_BK - 1) / L0_BK;
  float4 xr, wr;
  auto gload = [&](int c) {
    const int k = c * L0_BK + lk;
    xr = make_float4(0.f, 0.f, 0.f, 0.f);
    wr = xr;
    if (k < p.D) {  // D is a multiple of 4
      if (xok) {
        const float4 v = *reinterpret_cast<const float4 *>(xrow + k);
        const float4 sh = *reinterpret_cast<const float4 *>(p.shift + k);
        const float4 sc = *reinterpret_cast<const float4 *>(p.scale + k);
        // ApplyShiftAndScale: add, then multiply (dnn.cc:184-187)
        xr.x = (v.x + sh.x) * sc.x;
        xr.y = (v.y + sh.y) * sc.y;
        xr.z = (v.z + sh.z) * sc.z;
        xr.w = (v.w + sh.w) * sc.w;
      }
      if (wok) wr = *reinterpret_cast<const float4 *>(wrow + k);
    }
  };
  auto lstore = [&](int buf) {
    *reinterpret_cast<float4 *>(xs + (buf * L0_TF + lrow) * L0_LD + lk) = xr;
    *reinterpret_cast<float4 *>(ws + (buf * L0_TN + lrow) * L0_LD + lk) = wr;
  };
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(c + 1);
    const float *xb = xs + buf * L0_TF * L0_LD;
    const float *wb = ws + buf * L0_TN * L0_LD;
#pragma unroll 2
    for (int k4 = 0; k4 < L0_BK / 4; ++k4) {
      float4 xv[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(xb + (ty * 4 + i) * L0_LD + k4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const float4 *>(wb + (tx + 16 * j) * L0_LD + k4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // InputActivations: four lane partial sums over k mod 4 (dnn.cc:233-238)
          if (FMA) {
            acc[i][j][0] = fmaf(xv[i].x, wv[j].x, acc[i][j][0]);
            acc[i][j][1] = fmaf(xv[i].y, wv[j].y, acc[i][j][1]);
            acc[i][j][2] = fmaf(xv[i].z, wv[j].z, acc[i][j][2]);
            acc[i][j][3] = fmaf(xv[i].w, wv[j].w, acc[i][j][3]);
          } else {
            acc[i][j][0] = acc[i][j][0] + xv[i].x * wv[j].x;
            acc[i][j][1] = acc[i][j][1] + xv[i].y * wv[j].y;
            acc[i][j][2] = acc[i][j][2] + xv[i].z * wv[j].z;
            acc[i][j][3] = acc[i][j][3] + xv[i].w * wv[j].w;
          }
        }
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int node = n0 + tx + 16 * j;
      if (node < p.H) {
        // horizontalSum: (l0+l1)+(l2+l3) (dnn.cc:168-172), then AddBias
        const float s = (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        const float lin = s + p.bias[node];
        if (TAP && f < p.n) p.tap_lin[static_cast<size_t>(f) * p.H + node] = lin;
        // rows >= n are scratch padding; writing them keeps later loads defined
        p.act_out[static_cast<size_t>(f) * p.act_ld + node] = static_cast<int8_t>(lut[lut_index(lin)]);
      }
    }
  }
}

// ---------------------------------------------------------------- int8 GEMM (MFMA 32x32x32 i8)
// C[node][frame] = sum_k W[node][k] * A[frame][k].  Both operands are
// K-contiguous byte rows, so both MFMA fragments are plain 16-byte row slices.
//
// Workgroup tile: 256 nodes x FT = 32*NF frames, k-step 64 bytes.  4 waves, wave
// w owns nodes [64w, 64w+64) x all FT frames = 2 x NF MFMA 32x32 tiles
// (32*NF accumulator registers).  Two workgroups per CU, so one workgroup's
// VALU epilogue overlaps the other's MFMA main loop.
//
// HBM/L2 -> LDS: global_load_lds_dwordx4, one wave instruction = 16 rows x 64 B,
// into a ring of STAGES buffers; loads stay in flight across the (raw) barrier
// with a counted s_waitcnt vmcnt, one barrier per k-step.
//
// LDS image: rows of 64 B (four rows per 256-B bank row).  16-byte chunk c of row
// r is stored at chunk c ^ ((r>>2)&3); the XOR is applied to the per-lane GLOBAL
// address while the LDS destination stays lane-linear, and again on the
// ds_read_b128 fragment reads, which makes every 16-lane read group hit 16
// distinct 16-byte slots (conflict free).
constexpr int G_BM = 256;
constexpr int G_BK = 64;
constexpr int G_W_BYTES = G_BM * G_BK;  // 16 KiB

template <int NF>
struct GemmCfg {
  static constexpr int FT = 32 * NF;
  static constexpr int A_BYTES = FT * G_BK;
  static constexpr int STAGE = G_W_BYTES + A_BYTES;
  static constexpr int A_SLABS = FT / 16;                 // 1-KiB wave instructions per activation stage
  static constexpr int MIN_LOADS = 4 + A_SLABS / 4;       // fewest loads any wave issues per stage
};

template <int NF, int STAGES>
constexpr int gemm_lds_bytes() {
  return GemmCfg<NF>::STAGE * STAGES > ((kLutExt + 15) & ~15) ? GemmCfg<NF>::STAGE * STAGES : ((kLutExt + 15) & ~15);
}

__device__ __forceinline__ v4i read_frag(const char *tile, int row, int chunk) {
  return *reinterpret_cast<const v4i *>(tile + row * G_BK + ((chunk ^ ((row >> 2) & 3)) << 4));
}

template <int NF, int STAGES, bool OUTPUT, bool TAP>
__global__ __launch_bounds__(256, 2) void qgemm_kernel(QGemmParams p) {
  using Cfg = GemmCfg<NF>;
  constexpr int FT = Cfg::FT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Workgroup b runs on XCD b%8.  Give each XCD a contiguous band of frame tiles and
  // walk the node tiles fastest inside it: co-resident workgroups then share the
  // same activation rows (and all of them share the weights) in that XCD's L2.
  const int MT = p.rows_pad / G_BM;
  const int NT = p.n_pad / FT;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int per_xcd = (NT + 7) / 8;
  const int mt = j % MT;
  const int nt = xcd * per_xcd + j / MT;
  if (j / MT >= per_xcd || nt >= NT) return;
  const int m0 = mt * G_BM, f0 = nt * FT;

  // Row strides carry a 64-byte skew on top of the power-of-two width (kRowSkew): with
  // a 2048-byte stride every row of a tile -- and every workgroup, since they all
  // walk k in lockstep -- would map to the same couple of L2 channels.
  const size_t ldw = static_cast<size_t>(p.ldw), lda = static_cast<size_t>(p.lda);
  // per-lane source offsets of the staging loads (row = slab*16 + lane/4, swizzled chunk)
  const int srow = lane >> 2;
  const int schunk = ((lane & 3) ^ ((srow >> 2) & 3)) << 4;
  const int8_t *gw = p.w + static_cast<size_t>(m0 + wave * 64 + srow) * ldw + schunk;  // + slab*16 rows
  const int8_t *ga = p.a + static_cast<size_t>(f0 + wave * 16 + srow) * lda + schunk;  // + 64 rows per extra slab

  const int KT = p.K / G_BK;
  auto stage = [&](int kt, int buf) {
    char *base = smem + buf * Cfg::STAGE;
    const int koff = kt * G_BK;
#pragma unroll
    for (int s = 0; s < 4; ++s)  // weight slabs 4w .. 4w+3
      glds16(gw + static_cast<size_t>(s * 16) * ldw + koff, base + (wave * 4 + s) * 1024);
#pragma unroll
    for (int s = 0; s < (Cfg::A_SLABS + 3) / 4; ++s) {  // activation slabs w, w+4, ...
      if (s * 4 + wave < Cfg::A_SLABS)
        glds16(ga + static_cast<size_t>(s * 64) * lda + koff, base + G_W_BYTES + (s * 4 + wave) * 1024);
    }
  };

  v16i acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < KT) stage(s, s);

  const int frow = lane & 31, fch = lane >> 5;

  // pmaddubsw saturation (dnn.cc:337-340).  The MFMA sum is exact; the reference
  // saturates every ADJACENT pair a[2j]*w[2j] + a[2j+1]*w[2j+1] to int16.  Only the
  // few (node, pair) entries listed at load time can saturate at all.  Those of this
  // wave's 64 nodes are sorted by k; when the k-step holding an entry's columns is in
  // LDS, the pair is recomputed from the staged activation bytes and sat16(p) - p is
  // added to the accumulator that holds (node, frame).  The entry walk and the register
  // select are wave-uniform; a layer without risky pairs has fix_k_next = INT_MAX.
  const FixEntry *ent = reinterpret_cast<const FixEntry *>(p.fix_ent);
  int fix_e = 0, fix_end = 0, fix_k_next = INT_MAX;
  if (ent) {
    const int grp = (m0 >> 6) + wave;
    fix_e = p.fix_grp[grp];
    fix_end = p.fix_grp[grp + 1];
    if (fix_e < fix_end) fix_k_next = ent[fix_e].k;
  }

  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // stage kt has landed once at most (STAGES-2) younger stages are outstanding
    if (STAGES > 2 && kt + STAGES - 2 < KT) {
      if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::MIN_LOADS) : "memory");
      if (STAGES == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Cfg::MIN_LOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // everyone's share of stage kt landed; everyone is done reading stage kt-1
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < KT) {
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage(kt + STAGES - 1, nb);
    }
    const char *wt = smem + buf * Cfg::STAGE;
    const char *at = wt + G_W_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      v4i a[2], b[NF];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = read_frag(wt, 64 * wave + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) b[ni] = read_frag(at, 32 * ni + frow, kk * 2 + fch);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    while (fix_k_next < (kt + 1) * G_BK) {  // rare: a risky pair lives in this k-step
      const FixEntry t = ent[fix_e];
      const int node = __builtin_amdgcn_readfirstlane(t.node) - (m0 + 64 * wave);  // 0..63
      const int kl = __builtin_amdgcn_readfirstlane(t.k) - kt * G_BK;               // even, 0..62
      const int w0 = __builtin_amdgcn_readfirstlane(t.w0), w1 = __builtin_amdgcn_readfirstlane(t.w1);
      const int rr = node & 31;
      const int idx = (node >> 5) * 16 + (rr & 3) + 4 * (rr >> 3);  // mi*16 + reg
      const bool mine = (lane >> 5) == ((rr >> 2) & 1);
      int c[NF];
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int row = 32 * ni + frow;
        const uint32_t pair = *reinterpret_cast<const uint16_t *>(at + row * G_BK + ((((kl >> 4) ^ ((row >> 2) & 3))) << 4) + (kl & 15));
        const int a0 = static_cast<int>((pair & 0xff) ^ 0x80), a1 = static_cast<int>((pair >> 8) ^ 0x80);  // back to u8
        const int prod = a0 * w0 + a1 * w1;
        c[ni] = mine ? max(-32768, min(32767, prod)) - prod : 0;
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (idx == i) {
#pragma unroll
          for (int ni = 0; ni < NF; ++ni) acc[i >> 4][ni][i & 15] += c[ni];
        }
      }
      ++fix_e;
      fix_k_next = fix_e < fix_end ? static_cast<int>(ent[fix_e].k) : INT_MAX;
    }
    if (++buf == STAGES) buf = 0;
  }

  // ------------------------------------------------------------ epilogue
  // the ring is free now: drop the sigmoid table into LDS
  uint8_t *lut = reinterpret_cast<uint8_t *>(smem);
  if (!OUTPUT) {
    __syncthreads();
    for (int i = tid; i < kLutExt; i += 256) lut[i] = p.lut[i];
    __syncthreads();
  }
  // D layout (32x32): column (frame) = lane&31, row (node) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  const int half = lane >> 5;
  const bool vec4 = (p.rows & 3) == 0;

  float psum[NF];
#pragma unroll
  for (int ni = 0; ni < NF; ++ni) psum[ni] = 0.0f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = m0 + 64 * wave + 32 * mi + 8 * g + 4 * half;  // 4 consecutive nodes nb..nb+3
      const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + nb);
      const int4 ws4 = *reinterpret_cast<const int4 *>(p.wsum + nb);
      const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
      const int wj[4] = {ws4.x, ws4.y, ws4.z, ws4.w};
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int f = f0 + 32 * ni + frow;
        int av[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[q] = acc[mi][ni][g * 4 + q] + wj[q];
          if (TAP && f < p.n && nb + q < p.rows) p.tap_acc[static_cast<size_t>(f) * p.rows + nb + q] = av[q];
        }
        if (!OUTPUT) {
          uint32_t packed = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lin = dequant(av[q], p.coef, p.rcp_coef, p.fastdiv) + bj[q];
            packed |= static_cast<uint32_t>(lut[lut_index(lin)]) << (8 * q);
          }
          if (nb < p.rows) *reinterpret_cast<uint32_t *>(p.act_out + static_cast<size_t>(f) * p.act_ld + nb) = packed;
        } else {
          float e[4];
          const bool live = f < p.n;
          uint32_t mbits = 0x01010101u;
          if (p.mask && live && nb < p.rows) {
            const int8_t *mp = p.mask + static_cast<size_t>(f) * p.rows + nb;
            if (vec4) {
              mbits = *reinterpret_cast<const uint32_t *>(mp);
            } else {
              mbits = 0;
              for (int q = 0; q < 4; ++q)
                if (nb + q < p.rows && mp[q]) mbits |= 0xffu << (8 * q);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // sum/coef, then += bias (dnn.cc:311, :446); masked-out nodes keep z = 0 (dnn.cc:366-369)
            float z = dequant(av[q], p.coef, p.rcp_coef, p.fastdiv) + bj[q];
            if (((mbits >> (8 * q)) & 0xffu) == 0) z = 0.0f;
            if (TAP && live && nb + q < p.rows) p.tap_logit[static_cast<size_t>(f) * p.rows + nb + q] = z;
            e[q] = (nb + q < p.rows) ? expf(z) : 0.0f;
            psum[ni] += e[q];
          }
          if (live) {
            float *op = p.out + static_cast<size_t>(f) * p.rows + nb;
            if (vec4) {
              if (nb < p.rows) *reinterpret_cast<float4 *>(op) = make_float4(e[0], e[1], e[2], e[3]);
            } else {
              for (int q = 0; q < 4; ++q)
                if (nb + q < p.rows) op[q] = e[q];
            }
          }
        }
      }
    }
  }
  if (OUTPUT) {
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const float tot = psum[ni] + __shfl_xor(psum[ni], 32);
      const int f = f0 + 32 * ni + frow;
      if (half == 0) p.partial[static_cast<size_t>(mt * 4 + wave) * p.partial_ld + f] = tot;
    }
  }
}

// ---------------------------------------------------------------- soft-max normalisation
// SoftMax::apply second loop (dnn.cc:541-543): p_i = e_i / total.  total is the
// sum of the output kernel's per-64-node partials in a fixed order.
__global__ __launch_bounds__(256) void normalize_kernel(float *out, const float *partial, int n, int partial_ld, int rows,
                                                        int n_partial) {
  __shared__ float red[4];
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  float s = 0.0f;
  for (int t = tid; t < n_partial; t += 256) s += partial[static_cast<size_t>(t) * partial_ld + f];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float total = (red[0] + red[1]) + (red[2] + red[3]);
  float *row = out + static_cast<size_t>(f) * rows;
  if ((rows & 3) == 0) {
    float4 *r4 = reinterpret_cast<float4 *>(row);
    for (int i = tid; i < rows / 4; i += 256) {
      float4 v = r4[i];
      v.x = v.x / total;
      v.y = v.y / total;
      v.z = v.z / total;
      v.w = v.w / total;
      r4[i] = v;
    }
  } else {
    for (int i = tid; i < rows; i += 256) row[i] = row[i] / total;
  }
}

// ---------------------------------------------------------------- load-time check of the fast division
__global__ __launch_bounds__(256) void fastdiv_check_kernel(float coef, float rcp, unsigned long long *mismatch) {
  const long long lo = -(1ll << 26), hi = (1ll << 26);
  unsigned long long bad = 0;
  for (long long a = lo + blockIdx.x * 256ll + threadIdx.x; a <= hi; a += static_cast<long long>(gridDim.x) * 256ll) {
    const float fast = dequant(static_cast<int>(a), coef, rcp, 1);
    const float ieee = dequant(static_cast<int>(a), coef, rcp, 0);
    if (__float_as_uint(fast) != __float_as_uint(ieee)) ++bad;
  }
  if (bad) atomicAdd(mismatch, bad);
}

__global__ __launch_bounds__(256) void xor80_kernel(const int8_t *in, uint8_t *out, size_t count) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < count) out[i] = static_cast<uint8_t>(in[i]) ^ 0x80;
}

}  // namespace

// ---------------------------------------------------------------- launchers
void launch_l0(const L0Params &p, hipStream_t s) {
  dim3 grid((p.H + L0_TN - 1) / L0_TN, (p.n_rows + L0_TF - 1) / L0_TF);
  if (p.tap_lin) {
    if (p.fma)
      hipLaunchKernelGGL((l0_kernel<true, true>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((l0_kernel<false, true>), grid, dim3(256), 0, s, p);
  } else {
    if (p.fma)
      hipLaunchKernelGGL((l0_kernel<true, false>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((l0_kernel<false, false>), grid, dim3(256), 0, s, p);
  }
}

int qgemm_frame_tile(int rows_pad, int n) {
  // Pick the frame tile (128 / 160 / 192) that minimises rounds x tile cost, where a
  // round is 2 workgroups on each of the 256 CUs; ties go to the larger tile (more
  // reuse per byte staged).
  const int mt = rows_pad / G_BM;
  int best = 128;
  long best_cost = -1;
  for (int ft : {128, 160, 192}) {
    const long blocks = static_cast<long>(mt) * ((n + ft - 1) / ft);
    const long rounds = (blocks + 511) / 512;
    const long cost = rounds * ft;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && ft > best)) {
      best_cost = cost;
      best = ft;
    }
  }
  return best;
}

namespace {

template <int NF, bool OUTPUT>
void launch_qgemm_nf(const QGemmParams &p, hipStream_t s) {
  constexpr int STAGES = 3;
  constexpr int lds = gemm_lds_bytes<NF, STAGES>();
  const int MT = p.rows_pad / G_BM, NT = p.n_pad / (32 * NF);
  const int blocks = 8 * MT * ((NT + 7) / 8);
  auto k_prod = qgemm_kernel<NF, STAGES, OUTPUT, false>;
  auto k_tap = qgemm_kernel<NF, STAGES, OUTPUT, true>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tap), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  if (p.tap_acc)
    hipLaunchKernelGGL(k_tap, dim3(blocks), dim3(256), lds, s, p);
  else
    hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(256), lds, s, p);
}

template <bool OUTPUT>
void launch_qgemm(const QGemmParams &p, hipStream_t s) {
  switch (p.frame_tile) {
    case 128: launch_qgemm_nf<4, OUTPUT>(p, s); break;
    case 160: launch_qgemm_nf<5, OUTPUT>(p, s); break;
    default: launch_qgemm_nf<6, OUTPUT>(p, s); break;
  }
}

}  // namespace

void launch_qgemm_hidden(const QGemmParams &p, hipStream_t s) { launch_qgemm<false>(p, s); }
void launch_qgemm_output(const QGemmParams &p, hipStream_t s) { launch_qgemm<true>(p, s); }

void launch_normalize(float *out, const float *partial, int n, int partial_ld, int rows, int n_partial, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(normalize_kernel, dim3(n), dim3(256), 0, s, out, partial, n, partial_ld, rows, n_partial);
}

void launch_fastdiv_check(float coef, float rcp, unsigned long long *d_mismatch, hipStream_t s) {
  hipLaunchKernelGGL(fastdiv_check_kernel, dim3(2048), dim3(256), 0, s, coef, rcp, d_mismatch);
}

void launch_xor80(const int8_t *in, uint8_t *out, size_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(xor80_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s, in, out, count);
}

}  // namespace fdnn
