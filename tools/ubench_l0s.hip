// ubench_l0s.hip -- layer-0 (unfused fp32, four k-mod-4 chains) with the WEIGHTS IN SGPRs:
// lane = frame, the frame tile's shifted/scaled input lives in LDS for the whole block, each wave
// walks node groups and streams their weights through the scalar cache (s_load), so the inner
// loop is v_mul_f32 v, s, v + v_add_f32 with one ds_read_b128 per 32 (NT=8) multiply-adds.
// Timing only (synthetic data): is the scalar-cache path fast enough to feed the VALU?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_l0s ubench_l0s.hip && ./ubench_l0s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int D = 432, H = 2048, LDX = D + 4;

// wp: weights packed [H/NT][D/4][NT][4]
template <int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void l0s_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                            const float *__restrict__ bias, unsigned char *__restrict__ out,
                                                            int n) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [64][LDX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f0 = blockIdx.x * 64;
  for (int i = tid; i < 64 * (D / 4); i += 64 * WAVES) {
    const int r = i / (D / 4), q = i % (D / 4);
    float4 v = make_float4(0, 0, 0, 0);
    if (f0 + r < n) v = *reinterpret_cast<const float4 *>(x + (size_t)(f0 + r) * D + q * 4);
    *reinterpret_cast<float4 *>(xs + r * LDX + q * 4) = v;
  }
  __syncthreads();
  const float *xrow = xs + lane * LDX;
  constexpr int GROUPS = H / NT, GPW = GROUPS / WAVES;
  for (int g = wave * GPW; g < (wave + 1) * GPW; ++g) {
    float acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[j][c] = 0.0f;
    const float *wg = wp + (size_t)g * (D / 4) * NT * 4;
#pragma unroll 2
    for (int k4 = 0; k4 < D / 4; ++k4) {
      const float4 xv = *reinterpret_cast<const float4 *>(xrow + k4 * 4);
      const float *w = wg + k4 * NT * 4;  // wave-uniform: scalar loads
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[j][0] = acc[j][0] + xv.x * w[j * 4 + 0];
        acc[j][1] = acc[j][1] + xv.y * w[j * 4 + 1];
        acc[j][2] = acc[j][2] + xv.z * w[j * 4 + 2];
        acc[j][3] = acc[j][3] + xv.w * w[j * 4 + 3];
      }
    }
    unsigned char b[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float s = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]) + bias[g * NT + j];
      b[j] = (unsigned char)(int)(s * 100.0f);
    }
    if (f0 + lane < n) {
      unsigned char *o = out + (size_t)(f0 + lane) * H + g * NT;
#pragma unroll
      for (int j = 0; j < NT; j += 4) *reinterpret_cast<unsigned *>(o + j) = b[j] | b[j + 1] << 8 | b[j + 2] << 16 | b[j + 3] << 24;
    }
  }
}

template <int NT, int WAVES>
void run(const float *x, const float *wp, const float *bias, unsigned char *out, int n) {
  const int lds = 64 * LDX * 4;
  hipFuncSetAttribute((const void *)l0s_kernel<NT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((l0s_kernel<NT, WAVES>), dim3((n + 63) / 64), dim3(64 * WAVES), lds, 0, x, wp, bias, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("NT=%2d waves=%2d: %.3f ms  (%.1f lane-MAC/clk/CU at 2.4 GHz)\n", NT, WAVES, best,
         double(n) * H * D / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  const int n = 10000;
  float *x, *wp, *bias; unsigned char *out;
  hipMalloc(&x, (size_t)n * D * 4); hipMalloc(&wp, (size_t)H * D * 4); hipMalloc(&bias, H * 4); hipMalloc(&out, (size_t)n * H);
  hipMemset(x, 0, (size_t)n * D * 4); hipMemset(wp, 0, (size_t)H * D * 4); hipMemset(bias, 0, H * 4);
  run<8, 4>(x, wp, bias, out, n);
  run<8, 8>(x, wp, bias, out, n);
  run<8, 16>(x, wp, bias, out, n);
  run<16, 8>(x, wp, bias, out, n);
  run<4, 8>(x, wp, bias, out, n);
  run<4, 16>(x, wp, bias, out, n);
  return 0;
}
