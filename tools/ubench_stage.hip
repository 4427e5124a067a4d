// ubench_stage.hip -- the GEMM's staging pattern alone (no MFMA): 256 blocks x 512 threads, per k-step
// 256 weight rows (shared by all blocks with the same node tile) + 320 activation rows (per frame tile),
// 128 B per row, double buffered, wait + barrier each step.  Prints cycles per k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void *)(p))
struct P { const char *w; const char *a; int ld; int KT; int mode; long long *out; };
template <int AUX>
__global__ __launch_bounds__(512, 2) void stage_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;  // 32 blocks per XCD: 8 node tiles x 4 frame tiles
  const int mt = j & 7, nt = xcd * 4 + (j >> 3);
  const int srow = lane >> 3, sch = (lane & 7) << 4;
  const char *gw = p.w + (size_t)(mt * 256 + wave * 8 + srow) * p.ld + sch;
  const char *ga = p.a + (size_t)(nt * 320 + wave * 8 + srow) * p.ld + sch;
  const int rot = (p.mode & 1) ? ((mt + (j >> 3)) & 7) * (p.KT >> 3) : 0;
  long long t0 = __builtin_readcyclecounter();
  auto stage = [&](int kt, int buf) {
    char *base = smem + buf * 73728;
    int kr = kt + rot; if (kr >= p.KT) kr -= p.KT;
    const int koff = kr * 128;
    if (!(p.mode & 2))
      for (int s = 0; s < 4; ++s) __builtin_amdgcn_global_load_lds(GLBP(gw + (size_t)(s * 64) * p.ld + koff), LDSP(base + (s * 8 + wave) * 1024), 16, 0, AUX);
    if (!(p.mode & 4))
      for (int s = 0; s < 5; ++s) __builtin_amdgcn_global_load_lds(GLBP(ga + (size_t)(s * 64) * p.ld + koff), LDSP(base + 32768 + (s * 8 + wave) * 1024), 16, 0, AUX);
  };
  stage(0, 0);
  for (int kt = 0; kt < p.KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < p.KT) stage(kt + 1, (kt + 1) & 1);
    if (p.mode & 8) __builtin_amdgcn_s_sleep(20);  // ~1280 cycles of "compute"
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) p.out[blockIdx.x] = t1 - t0;
}
int main() {
  const int ld = 2048, KT = 16;
  char *w, *a; long long *out;
  hipMalloc(&w, (size_t)2048 * ld); hipMalloc(&a, (size_t)10240 * ld); hipMalloc(&out, 256 * 8);
  hipMemset(w, 1, (size_t)2048 * ld); hipMemset(a, 1, (size_t)10240 * ld);
  char *flush; hipMalloc(&flush, 512u << 20);
  hipFuncSetAttribute((const void *)stage_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void *)stage_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  for (int cold = 0; cold < 2; ++cold)
    for (int mode : {0, 1, 2, 4, 8, 9, 16}) {
      double best = 1e30, sum = 0;
      float ms_best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        if (cold) hipMemset(flush, rep, 512u << 20);  // evict L2 + MALL
        P p{w, a, ld, KT, mode & 15, out};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode & 16) hipLaunchKernelGGL(stage_kernel<2>, dim3(256), dim3(512), 147456, 0, p);
        else hipLaunchKernelGGL(stage_kernel<0>, dim3(256), dim3(512), 147456, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < ms_best) ms_best = ms;
        long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256; sum += m; if (m < best) best = m;
      }
      printf("%s mode %2d (1=rotate 2=noW 4=noA 8=sleep 16=nt): kernel %6.1f us best; cycles/k-step avg %7.0f best %7.0f  -> %5.1f B/clk/CU\n",
             cold ? "cold" : "warm", mode, ms_best * 1000, sum / 5 / KT, best / KT, 73728.0 / (best / KT));
    }
  return 0;
}
