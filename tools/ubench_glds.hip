// ubench_glds.hip -- how fast can a CU pull L2-resident rows into LDS?
// Variants: row segment bytes (64/128/256) per k-step, LDS-DMA vs VGPR loads.
// Every block streams the same `rows x 2048 B` matrix (L2 resident) K-step by K-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void *)(p))
template <int SEG, bool DMA, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char *base, int rows_per_block, int ld, int iters, int *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SEG / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per wave instruction
  const int r = lane / LPR, c = (lane % LPR) * 16;
  const char *g = base + (size_t)((blockIdx.x * 7) % 16) * 0 + (size_t)(wave * RPI + r) * ld + c;
  const int slabs = rows_per_block / (4 * RPI);  // instrs per wave per k-step
  const int KT = ld / SEG;
  int4 accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (int kt = 0; kt < KT; ++kt) {
      char *lds = smem + (kt % DEPTH) * rows_per_block * SEG;
      for (int s = 0; s < slabs; ++s) {
        const char *src = g + (size_t)(s * 4 * RPI) * ld + kt * SEG;
        if (DMA) {
          __builtin_amdgcn_global_load_lds(GLBP(src), LDSP(lds + (s * 4 + wave) * 1024), 16, 0, 0);
        } else {
          int4 v = *(const int4 *)src;
          accv.x ^= v.x; accv.y += v.y; accv.z ^= v.z; accv.w += v.w;
        }
      }
      if (DMA && (kt % DEPTH) == DEPTH - 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
  }
  if (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); accv.x = smem[threadIdx.x * 16]; }
  if (accv.x == 0x12345678 && accv.y == 7) sink[0] = accv.z + accv.w;
}
template <int SEG, bool DMA, int DEPTH>
void run(const char *name, const char *d, int rows, int ld, int blocks, int *sink) {
  const int iters = 8;
  size_t lds = (size_t)DEPTH * rows * SEG;
  hipFuncSetAttribute((const void *)stream_kernel<SEG, DMA, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  stream_kernel<SEG, DMA, DEPTH><<<blocks, 256, lds>>>(d, rows, ld, 1, sink);
  hipEventRecord(a);
  stream_kernel<SEG, DMA, DEPTH><<<blocks, 256, lds>>>(d, rows, ld, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * rows * ld * iters;
  printf("%-34s rows/blk %4d blocks %4d lds %6zu : %8.3f ms  %7.2f TB/s  %6.1f GB/s/CU (per-CU %5.1f B/clk @2.4)\n", name, rows, blocks, lds, ms,
         bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
}
int main() {
  const int ld = 2048, total_rows = 4096;  // 8 MB matrix, L2/MALL resident
  char *d; hipMalloc(&d, (size_t)total_rows * ld); hipMemset(d, 1, (size_t)total_rows * ld);
  int *sink; hipMalloc(&sink, 4);
  for (int blocks : {256, 512}) {
    run<64, true, 2>("glds  64B rows depth2", d, 384, ld, blocks, sink);
    run<128, true, 2>("glds 128B rows depth2", d, 192, ld, blocks, sink);
    run<256, true, 2>("glds 256B rows depth2", d, 96, ld, blocks, sink);
    run<64, true, 3>("glds  64B rows depth3", d, 384, ld, blocks, sink);
    run<128, true, 1>("glds 128B rows depth1", d, 384, ld, blocks, sink);
    run<64, false, 1>("vgpr  64B rows", d, 384, ld, blocks, sink);
    run<128, false, 1>("vgpr 128B rows", d, 192, ld, blocks, sink);
    run<256, false, 1>("vgpr 256B rows", d, 96, ld, blocks, sink);
  }
  return 0;
}
