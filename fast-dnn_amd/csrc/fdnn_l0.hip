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
//                   v_pk_add_f32, two lane-ops per MAC; bound by the fp32 vector rate.
//   l0_mfma_kernel  the reference as its own Makefile builds it on an FMA host
//                   (-march=native contracts mul+add): each chain is an fmaf chain,
//                   which is bit-for-bit what v_mfma_f32_32x32x2_f32 computes, so the
//                   four chains become four accumulator tiles fed k = c, c+4, c+8, ...
#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {
namespace {

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
  const int f0 = blockIdx.y * TF, n0 = blockIdx.x * TN;
  for (int i = tid; i < kLutExt; i += 256) lut[i] = p.lut[i];

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
#pragma unroll
      for (int i = 0; i < TI; ++i)
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
  dim3 grid((p.H + 63) / 64, (p.n_rows + 16 * TI - 1) / (16 * TI));
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

}  // namespace

void launch_l0(const L0Params &p, hipStream_t s) {
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
