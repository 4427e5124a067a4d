// fdnn_kernels.hip -- layer 0, soft-max normalisation and small helper kernels
// for gfx950 (the int8 layer kernel lives in fdnn_gemm.hip).
//
//   ApplyShiftAndScale + InputActivations + AddBias + QuantizedSigmoid
//     (dnn.cc:175-192, :219-286)                      -> l0_kernel
//   SoftMax::apply second loop (dnn.cc:541-543)       -> normalize_kernel
#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {
namespace {

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
    const float fast = dequant<true>(static_cast<int>(a), coef, rcp);
    const float ieee = dequant<false>(static_cast<int>(a), coef, rcp);
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
