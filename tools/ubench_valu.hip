// ubench_valu.hip -- fp32 vector rate on gfx950: scalar v_mul_f32/v_add_f32 pairs vs packed
// v_pk_mul_f32/v_pk_add_f32 (the unfused multiply-add of the canonical layer-0 numerics).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_valu ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float *out, float x0, float w0, int iters) {
  // 16 independent accumulator pairs per lane
  v2f acc[16], x = {x0, x0 * 1.5f}, w = {w0, w0 * 0.75f};
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = v2f{float(i), float(threadIdx.x)};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) {  // scalar: 2 mul + 2 add
        float p0, p1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(x.x), "v"(w.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(x.y), "v"(w.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i].x) : "v"(acc[i].x), "v"(p0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[i].y) : "v"(acc[i].y), "v"(p1));
      } else {  // packed: 1 pk_mul + 1 pk_add
        v2f pr;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(x), "v"(w));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(acc[i]), "v"(pr));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out, int per_cu = 8) {
  const int iters = 2000, blocks = 256 * per_cu;
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double macs = double(blocks) * 256 * iters * 32;  // unfused multiply-adds
  printf("%-8s %d waves/SIMD  %.3f ms  %.2f T unfused-MAC/s  (%.1f lane-MAC/clk/CU at 2.4 GHz)\n", name, per_cu, best, macs / best / 1e9,
         macs / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int per_cu : {1, 2, 3, 4, 8}) {  // 256-thread blocks per CU = waves per SIMD
    run<0>("scalar", out, per_cu);
    run<1>("packed", out, per_cu);
  }
  return 0;
}
