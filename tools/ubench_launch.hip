// ubench_launch.hip -- fixed cost of a kernel launch by shape (threads, LDS, VGPRs).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int T>
__global__ __launch_bounds__(T, 2) void empty_kernel(int *p, int n) {
  extern __shared__ char smem[];
  if (n == -1) { smem[threadIdx.x] = 1; p[0] = smem[0]; }
}
template <int T>
__global__ __launch_bounds__(T, 2) void fatreg_kernel(int *p, int n) {
  extern __shared__ char smem[];
  int v[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) v[i] = n * i;
  if (n == -1) {
#pragma unroll
    for (int i = 0; i < 200; ++i) asm volatile("" : "+v"(v[i]));
    int s = 0;
#pragma unroll
    for (int i = 0; i < 200; ++i) s ^= v[i];
    p[0] = s + smem[0];
  }
}
template <typename K>
void timeit(const char *name, K k, int blocks, int threads, size_t lds, int *d) {
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s blocks %5d threads %4d lds %7zu : %7.2f us per launch\n", name, blocks, threads, lds, ms * 1000 / reps);
}
int main() {
  int *d; hipMalloc(&d, 64);
  timeit("empty 256thr", empty_kernel<256>, 512, 256, 0, d);
  timeit("empty 256thr lds80k", empty_kernel<256>, 512, 256, 80 * 1024, d);
  timeit("empty 512thr lds147k", empty_kernel<512>, 256, 512, 147 * 1024, d);
  timeit("empty 512thr lds147k x1024", empty_kernel<512>, 1024, 512, 147 * 1024, d);
  timeit("fatreg 512thr lds147k", fatreg_kernel<512>, 256, 512, 147 * 1024, d);
  timeit("fatreg 512thr lds0", fatreg_kernel<512>, 256, 512, 0, d);
  timeit("fatreg 256thr lds80k x512", fatreg_kernel<256>, 512, 256, 80 * 1024, d);
  timeit("empty 256thr x2048", empty_kernel<256>, 2048, 256, 0, d);
  return 0;
}
