// ubench_dma_waves.hip -- does a workgroup's LDS-DMA operand stream go faster when more waves issue it?
// One workgroup per CU streams 16 k-steps of 36 slabs x 1 KiB (256 weight rows + 32 activation rows of 128 B: the
// small-batch GEMM tile) through a 3-stage ring with W waves issuing the loads.  Prints cycles per k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
struct P { const char *w; int ld; int KT; int slabs; long long *out; };
template <int W>
__global__ __launch_bounds__(64 * W) void k(P p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.w), 0, 0x7fffffff, 0x00020000);
  const int voff = (lane >> 3) * p.ld + ((lane & 7) << 4);
  const int row0 = (blockIdx.x & 7) * 256;  // 8 node tiles share weights like the real launch
  auto stage = [&](int kt, int buf) {
    for (int s = wave; s < p.slabs; s += W)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(smem + buf * 40960 + s * 1024), 16, voff, (row0 + s * 8) * p.ld + kt * 128, 0, 0);
  };
  long long t0 = __builtin_readcyclecounter();
  stage(0, 0);
  stage(1, 1);
  for (int kt = 0; kt < p.KT; ++kt) {
    if (kt + 2 < p.KT) stage(kt + 2, (kt + 2) % 3);
    // wait for stage kt: everything but the two younger stages (each wave issued ceil/floor(slabs/W) loads per stage;
    // conservative: wait for all when near the end)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) p.out[blockIdx.x] = t1 - t0;
}
// the same stream through registers: buffer_load_dwordx4 -> VGPR -> ds_write_b128 (one stage ahead)
template <int W, int PER>
__global__ __launch_bounds__(64 * W) void kreg(P p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.w), 0, 0x7fffffff, 0x00020000);
  const int voff = (lane >> 3) * p.ld + ((lane & 7) << 4);
  const int row0 = (blockIdx.x & 7) * 256;
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i r[PER];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = wave + i * W;
      r[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, s < p.slabs ? voff : 0x7ffffff0, (row0 + s * 8) * p.ld + kt * 128, 0));
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int s = wave + i * W;
      if (s < p.slabs) *reinterpret_cast<v4i *>(smem + buf * 40960 + s * 1024 + lane * 16) = r[i];
    }
  };
  long long t0 = __builtin_readcyclecounter();
  gload(0);
  for (int kt = 0; kt < p.KT; ++kt) {
    lstore(kt & 1);
    if (kt + 1 < p.KT) gload(kt + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) p.out[blockIdx.x] = t1 - t0 + (r[0].x == 12345);
}
template <int W, int PER>
void runreg(const P &p, int blocks) {
  hipFuncSetAttribute((const void *)kreg<W, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
  long long h[256];
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((kreg<W, PER>), dim3(blocks), dim3(64 * W), 3 * 40960, 0, p);
    hipDeviceSynchronize();
    hipMemcpy(h, p.out, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    s /= blocks; if (s < best) best = s;
  }
  printf("reg blocks %3d waves %2d: %7.0f cycles per k-step (%d slabs)\n", blocks, W, best / p.KT, p.slabs);
}
template <int W>
void run(const P &p, int blocks, const char *tag) {
  hipFuncSetAttribute((const void *)k<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960);
  long long h[256];
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64 * W), 3 * 40960, 0, p);
    hipDeviceSynchronize();
    hipMemcpy(h, p.out, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    s /= blocks; if (s < best) best = s;
  }
  printf("%s blocks %3d waves %2d: %7.0f cycles per k-step (%d slabs)\n", tag, blocks, W, best / p.KT, p.slabs);
}
int main() {
  const int ld = 2048, KT = 16;
  char *w; long long *out;
  hipMalloc(&w, (size_t)4096 * ld); hipMalloc(&out, 256 * 8);
  hipMemset(w, 1, (size_t)4096 * ld);
  for (int blocks : {8, 32, 256})
    for (int slabs : {36, 12}) {
      P p{w, ld, KT, slabs, out};
      run<1>(p, blocks, "dma"); run<2>(p, blocks, "dma"); run<4>(p, blocks, "dma"); run<8>(p, blocks, "dma"); run<16>(p, blocks, "dma");
      if (slabs == 36) { runreg<4, 9>(p, blocks); runreg<8, 5>(p, blocks); runreg<16, 3>(p, blocks); }
    }
  return 0;
}
