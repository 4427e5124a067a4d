// fdnn_kernels.hip -- soft-max normalisation and small helper kernels for gfx950
// (layer 0 lives in fdnn_l0.hip, the int8 layer kernel in fdnn_gemm.hip).
//
//   SoftMax::apply second loop (dnn.cc:541-543)       -> normalize_kernel
#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

namespace fdnn {
namespace {

// ---------------------------------------------------------------- soft-max normalisation
// SoftMax::apply second loop (dnn.cc:541-543): p_i = e_i / total.  total is the
// sum of the output kernel's per-64-node partials in a fixed order.
// dst == out scales in place; the per-frame lazy call passes a host-mapped dst instead, so the
// probabilities land in the caller's pinned buffer without a copy command.
__global__ __launch_bounds__(256) void normalize_kernel(const float *out, float *dst, const float *partial, int n, int partial_ld,
                                                        int rows, int n_partial) {
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
  const float *row = out + static_cast<size_t>(f) * rows;
  float *drow = dst + static_cast<size_t>(f) * rows;
  if ((rows & 3) == 0) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
    float4 *d4 = reinterpret_cast<float4 *>(drow);
    for (int i = tid; i < rows / 4; i += 256) {
      float4 v = r4[i];
      v.x = v.x / total;
      v.y = v.y / total;
      v.z = v.z / total;
      v.w = v.w / total;
      d4[i] = v;
    }
  } else {
    for (int i = tid; i < rows; i += 256) drow[i] = row[i] / total;
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

void launch_normalize(float *out, float *dst, const float *partial, int n, int partial_ld, int rows, int n_partial, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(normalize_kernel, dim3(n), dim3(256), 0, s, out, dst, partial, n, partial_ld, rows, n_partial);
}

void launch_fastdiv_check(float coef, float rcp, unsigned long long *d_mismatch, hipStream_t s) {
  hipLaunchKernelGGL(fastdiv_check_kernel, dim3(2048), dim3(256), 0, s, coef, rcp, d_mismatch);
}

void launch_xor80(const int8_t *in, uint8_t *out, size_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(xor80_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s, in, out, count);
}

}  // namespace fdnn
