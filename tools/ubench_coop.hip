// ubench_coop.hip -- what a cooperative launch costs next to ordinary launches on the same stream, and what a
// hand-made grid barrier (one agent-scope atomic counter, write-through / L2-bypassing accesses) costs inside a kernel.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_coop ubench_coop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void empty_kernel(int *p, int n) {
  if (n == -1) p[0] = 1;
}
// `rounds` grid barriers; every workgroup writes a value before each and reads its neighbour's after
__global__ __launch_bounds__(512) void barrier_kernel(unsigned *counter, unsigned *data, int rounds, long long *cycles) {
  const unsigned nb = gridDim.x;
  long long t0 = __builtin_readcyclecounter();
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(&data[blockIdx.x], unsigned(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope by default for HIP)
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = unsigned(r + 1) * nb;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      const unsigned v = __hip_atomic_load(&data[(blockIdx.x + 37) % nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v < unsigned(r + 1)) ++bad;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cycles[0] = __builtin_readcyclecounter() - t0;
    cycles[1] = bad;
  }
}
// two-level barrier: 8 group counters (workgroup b -> group b % 8, i.e. mostly its XCD), the last arrival of a group
// bumps the top counter, everybody polls the top counter
__global__ __launch_bounds__(512) void barrier2_kernel(unsigned *group_cnt, unsigned *top, int rounds, long long *cycles) {
  const unsigned nb = gridDim.x, g = blockIdx.x & 7, per = nb / 8;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      const unsigned prev = __hip_atomic_fetch_add(&group_cnt[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((prev + 1) % per == 0) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = unsigned(r + 1) * 8;
      while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = __builtin_readcyclecounter() - t0;
}
int main() {
  int *d; hipMalloc(&d, 64);
  unsigned *counter, *data; long long *cyc;
  hipMalloc(&counter, 4); hipMalloc(&data, 4 * 1024); hipMalloc(&cyc, 16);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 100;
  float ms;
  int n = 0;
  void *args[] = {&d, &n};
  for (int mode = 0; mode < 3; ++mode) {
    for (int w = 0; w < 2; ++w) {
      if (w) hipEventRecord(a, s);
      for (int i = 0; i < reps; ++i) {
        if (mode == 0 || (mode == 2 && (i & 1)))
          hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, d, 0);
        else
          hipLaunchCooperativeKernel((const void *)empty_kernel, dim3(256), dim3(512), args, 0, s);
      }
      if (w) hipEventRecord(b, s);
      hipStreamSynchronize(s);
    }
    hipEventElapsedTime(&ms, a, b);
    printf("%-40s %7.2f us per launch\n", mode == 0 ? "ordinary launches" : mode == 1 ? "cooperative launches" : "alternating cooperative / ordinary", ms * 1000 / reps);
  }
  for (int blocks : {64, 128, 256}) {
    const int rounds = 200;
    hipMemsetAsync(counter, 0, 4, s);
    hipMemsetAsync(data, 0, 4096, s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(512), 0, s, counter, data, rounds, cyc);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("grid barrier, %3d workgroups: %6.2f us per barrier (%lld cycles), stale reads %lld\n", blocks, ms * 1000 / rounds, h[0] / rounds, h[1]);
  }
  unsigned *gc, *top;
  hipMalloc(&gc, 4 * 32 * 8); hipMalloc(&top, 4);
  for (int blocks : {64, 128, 256}) {
    const int rounds = 200;
    hipMemsetAsync(gc, 0, 4 * 32 * 8, s);
    hipMemsetAsync(top, 0, 4, s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(barrier2_kernel, dim3(blocks), dim3(512), 0, s, gc, top, rounds, cyc);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, a, b);
    printf("two-level grid barrier, %3d workgroups: %6.2f us per barrier\n", blocks, ms * 1000 / rounds);
  }
  return 0;
}
