// ubench_pipe.hip -- the hidden GEMM's main loop in isolation: LDS-DMA staging + barriers + the real
// number of int8 MFMAs (register operands, no ds_read), to compare pipeline shapes before touching
// the product kernel.  256 blocks x 512 threads, tile 256 nodes x 320 frames, K = 2048.
//   layouts   row-major (rows of a 2048-B pitch)  |  k-slab-major ([k/BK][row][BK], 1-KB contiguous wave-loads)
//   MODE bits 1 = loads interleaved with the MFMAs (else all loads first)
//             2 = no loads in the k-loop (MFMA + barrier floor)      4 = no MFMA (staging floor)
//   hipcc --offload-arch=gfx950 -O3 -o ubench_pipe ubench_pipe.hip && ./ubench_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void *)(p))
struct P { const char *w; const char *a; int K; long long *out; };

template <int BK, int STAGES, bool SLAB, int MODE>
__global__ __launch_bounds__(512, 2) void pipe_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WROWS = 256, AROWS = 320, STAGE = (WROWS + AROWS) * BK;
  constexpr int WL = WROWS * BK / 1024, AL = AROWS * BK / 1024;  // 1-KB wave-loads per stage
  constexpr int NL = (WL + AL + 7) / 8;                          // per wave (max)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int mt = j >> 2, nt = xcd * 4 + (j & 3);  // frame tile fastest inside an XCD
  const int KT = p.K / BK;
  v16i acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  v4i fa = {lane, 1, 2, 3}, fb = {3, lane, 1, 0};
  auto stage_one = [&](int kt, int buf, int s) {
    char *base = smem + buf * STAGE;
    const int i = s * 8 + wave;  // wave-load index in the stage
    if (i >= WL + AL) return;
    const bool isw = i < WL;
    const int li = isw ? i : i - WL;
    const char *g;
    if (SLAB) {
      const char *mat = isw ? p.w : p.a;
      const size_t rows = isw ? 2048 : 10240;
      g = mat + (size_t)kt * rows * BK + (size_t)(isw ? mt * WROWS : nt * AROWS) * BK + li * 1024 + lane * 16;
    } else {
      constexpr int LPR = BK / 16, RPL = 64 / LPR;  // lanes per row, rows per wave-load
      const int row = (isw ? mt * WROWS : nt * AROWS) + li * RPL + lane / LPR;
      g = (isw ? p.w : p.a) + (size_t)row * p.K + kt * BK + (lane % LPR) * 16;
    }
    __builtin_amdgcn_global_load_lds(GLBP(g), LDSP(base + i * 1024), 16, 0, 0);
  };
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int s = 0; s < NL; ++s) stage_one(kt, buf, s);
  };
  long long t0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < STAGES - 1; ++s) stage(s, s);
  for (int kt = 0; kt < KT; ++kt) {
    // stage kt must have landed; up to STAGES-2 younger stages may stay in flight
    const int younger = (MODE & 2) ? 0 : min(STAGES - 2, KT - 1 - kt);
    const int mine = (wave < (WL + AL) - (NL - 1) * 8) ? NL : NL - 1;
    const int allow = younger * mine;
    if (allow >= 2 * NL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL > 63 ? 63 : 2 * NL) : "memory");
    else if (allow >= 2 * (NL - 1) && NL > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NL - 1)) : "memory");
    else if (allow >= NL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else if (allow >= NL - 1 && NL > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL - 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = !(MODE & 2) && kt + STAGES - 1 < KT;
    const int nkt = kt + STAGES - 1, nbuf = nkt % STAGES;
    if (more && (!(MODE & 1) || (MODE & 4))) stage(nkt, nbuf);
    if (!(MODE & 4)) {
      constexpr int NM = BK / 32 * 10, PER = NM / NL;  // MFMAs between two loads
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if ((MODE & 1) && m % PER == 0 && m / PER < NL)
          if (more) stage_one(nkt, nbuf, m / PER);
        acc[m % 10] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[m % 10], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) sum += acc[i][3];
  if (threadIdx.x == 0) { p.out[blockIdx.x] = (t1 - t0) + (sum == 12345 ? 1 : 0); p.out[256 + blockIdx.x] = r1 - r0; }
}

template <int BK, int STAGES, bool SLAB, int MODE>
void run(const char *name, const char *w, const char *a, long long *out, char *flush) {
  constexpr int LDS = (256 + 320) * BK * STAGES;
  hipFuncSetAttribute((const void *)pipe_kernel<BK, STAGES, SLAB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  for (int cold = 0; cold < 2; ++cold) {
    float best = 1e9;
    double cyc = 0, rt = 0;
    for (int rep = 0; rep < 6; ++rep) {
      if (cold) hipMemset(flush, rep, 512u << 20);
      P p{w, a, 2048, out};
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL((pipe_kernel<BK, STAGES, SLAB, MODE>), dim3(256), dim3(512), LDS, 0, p);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) {
        best = ms;
        long long h[512]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); rt = 0; for (int i = 0; i < 256; ++i) rt += h[256 + i]; rt /= 256;
        cyc = 0; for (int i = 0; i < 256; ++i) cyc += h[i]; cyc /= 256;
      }
    }
    printf("%-30s %-22s %s: %6.1f us, %7.0f cycles/block, %5.0f per 128-k, %.2f us in-kernel, clock %.2f GHz\n", name,
           (MODE & 4) ? "loads only" : (MODE & 2) ? "mfma+barrier only" : (MODE & 1) ? "loads interleaved" : "loads first",
           cold ? "cold" : "warm", best * 1e3, cyc, cyc / 16, rt / 100.0, cyc / (rt * 10.0));
  }
}

template <int BK, int STAGES, bool SLAB>
void run_all(const char *name, const char *w, const char *a, long long *out, char *flush) {
  run<BK, STAGES, SLAB, 0>(name, w, a, out, flush);
  run<BK, STAGES, SLAB, 1>(name, w, a, out, flush);
  run<BK, STAGES, SLAB, 2>(name, w, a, out, flush);
  run<BK, STAGES, SLAB, 4>(name, w, a, out, flush);
}

int main() {
  char *w, *a, *flush; long long *out;
  hipMalloc(&w, (size_t)2048 * 2048); hipMalloc(&a, (size_t)10240 * 2048); hipMalloc(&out, 512 * 8);
  hipMemset(w, 1, (size_t)2048 * 2048); hipMemset(a, 1, (size_t)10240 * 2048);
  hipMalloc(&flush, 512u << 20);
  run_all<128, 2, false>("row-major BK=128 x2 (ships)", w, a, out, flush);
  run_all<128, 2, true>("slab-major BK=128 x2", w, a, out, flush);
  run_all<64, 4, true>("slab-major BK=64 x4", w, a, out, flush);
  run_all<64, 4, false>("row-major BK=64 x4", w, a, out, flush);
  return 0;
}
