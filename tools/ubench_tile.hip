// ubench_tile.hip -- the GEMM k-loop with its REAL LDS fragment reads, for two wave layouts of
// the same 256-node x 320-frame workgroup tile (BK = 128, 2-stage LDS-DMA ring, loads
// interleaved with the MFMA sub-steps):
//   8 waves (4 x 2), wave tile  64 x 160: 28 ds_read_b128 + 40 MFMA per wave per k-step  (ships)
//   4 waves (2 x 2), wave tile 128 x 160: 36 ds_read_b128 + 80 MFMA per wave per k-step
// The second re-reads 36 % fewer operand bytes from LDS (147 vs 229 KB per k-step) but leaves one
// wave per SIMD.  Timing only; no epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_tile ubench_tile.hip && ./ubench_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
struct P { const char *w; const char *a; int K; long long *out; };

__device__ __forceinline__ v4i read_frag(const char *tile, int row, int chunk) {
  return *reinterpret_cast<const v4i *>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int WM, int MI, int DIST, int FLAGS>  // DIST: loads issued before sub-steps 0..3 as decimal digits; FLAGS: 1 no barrier, 2 no setprio
// WM waves along nodes, each MI x 32 nodes; 2 waves along frames, 5 x 32 frames each
__global__ __launch_bounds__(128 * WM, 1) void tile_kernel(P p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 2 * WM, NF = 5, BK = 128, WROWS = 256, AROWS = 320, STAGE = (WROWS + AROWS) * BK;
  constexpr int SLABS = (WROWS + AROWS) / 8, NLD = SLABS / NW;  // 8 rows of 128 B per 1-KB wave-load
  static_assert(WM * MI * 32 == WROWS && SLABS % NW == 0, "layout");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int mt = j >> 2, nt = xcd * 4 + (j & 3);
  const int KT = p.K / BK;
  const int srow = lane >> 3, schunk = ((lane & 7) ^ (((wave * 8 + srow) >> 1) & 7)) << 4;
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.w + (size_t)mt * WROWS * p.K), 0, WROWS * p.K, 0x00020000);
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.a + (size_t)nt * AROWS * p.K), 0, AROWS * p.K, 0x00020000);
  const int voff = srow * p.K + schunk;
  auto stage_load = [&](int kt, int buf, int i) {
    const int slab = i * NW + wave;
    char *dst = smem + buf * STAGE + slab * 1024;
    if (slab < WROWS / 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, LDSP(dst), 16, voff, slab * 8 * p.K + kt * BK, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, LDSP(dst), 16, voff, (slab - WROWS / 8) * 8 * p.K + kt * BK, 0, 0);
  };
  v16i acc[MI][NF];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
  const int frow = lane & 31, fch = lane >> 5;
  long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int i = 0; i < NLD; ++i) stage_load(0, 0, i);
  if (FLAGS & 4) {
    // ROTATED barrier: the step's one barrier sits before its LAST sub-step.  By then every
    // wave has all of this stage's fragments in registers, so the stage buffer is free and the
    // next stage (loaded a whole step ago) is verified; the next step's first fragments are
    // requested right after the barrier and their LDS latency hides behind sub-step 3's MFMAs.
    v4i a[2][MI], b[2][NF];
    auto load_frags = [&](int kt, int kk, int set) {
      const char *wt = smem + (kt & 1) * STAGE, *at = wt + WROWS * BK;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[set][mi] = read_frag(wt, 32 * MI * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) b[set][ni] = read_frag(at, 160 * wn + 32 * ni + frow, kk * 2 + fch);
    };
    auto mfmas = [&](int set) {
      if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[set][mi], b[set][ni], acc[mi][ni], 0, 0, 0);
      if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(0);
      if (FLAGS & 8) {  // round 6 (tools/ubench_role.hip): one fragment read pinned behind each of the first MI + NF MFMAs
#pragma unroll
        for (int i = 0; i < MI + NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, MI * NF - MI - NF, 0);
      }
    };
    constexpr int d0 = DIST / 1000, d1 = DIST / 100 % 10, d2 = DIST / 10 % 10;  // loads after the barrier, before ss0, ss1, (ss2)
    if (KT > 1) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) stage_load(1, 1, i);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");  // stage 0 landed (stage 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
      const bool refill = kt >= 1 && kt + 1 < KT;  // stage kt+1 goes into the buffer freed at the previous barrier
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        if (FLAGS & 8) __builtin_amdgcn_sched_barrier(0);
        load_frags(kt, kk + 1, (kk + 1) & 1);
        if (refill) {
          const int lo = kk == 0 ? d0 : kk == 1 ? d0 + d1 : d0 + d1 + d2;
          const int cnt = kk == 0 ? d1 : kk == 1 ? d2 : DIST % 10;
#pragma unroll
          for (int i = 0; i < NLD; ++i)
            if (i >= lo && i < lo + cnt) stage_load(kt + 1, (kt + 1) & 1, i);
        }
        mfmas(kk & 1);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my share of stage kt+1 landed; my reads of stage kt done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < KT) load_frags(kt + 1, 0, 0);
      if (kt + 2 < KT) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          if (i < d0) stage_load(kt + 2, kt & 1, i);
      }
      mfmas(1);
    }
  } else
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(FLAGS & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = kt + 1 < KT;
    const int buf = kt & 1;
    const char *wt = smem + buf * STAGE, *at = wt + WROWS * BK;
    v4i a[2][MI], b[2][NF];
    auto load_frags = [&](int kk, int set) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[set][mi] = read_frag(wt, 32 * MI * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) b[set][ni] = read_frag(at, 160 * wn + 32 * ni + frow, kk * 2 + fch);
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) load_frags(kk + 1, (kk + 1) & 1);
      if (more) {
        constexpr int d0 = DIST / 1000, d1 = DIST / 100 % 10, d2 = DIST / 10 % 10;
        const int lo = kk == 0 ? 0 : kk == 1 ? d0 : kk == 2 ? d0 + d1 : d0 + d1 + d2;
        const int cnt = kk == 0 ? d0 : kk == 1 ? d1 : kk == 2 ? d2 : DIST % 10;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          if (i >= lo && i < lo + cnt) stage_load(kt + 1, buf ^ 1, i);
      }
      if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni], 0, 0, 0);
      if (!(FLAGS & 2)) __builtin_amdgcn_s_setprio(0);
    }
  }
  long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  int sum = 0;
#pragma unroll
  for (int a2 = 0; a2 < MI; ++a2)
#pragma unroll
    for (int b2 = 0; b2 < NF; ++b2) sum += acc[a2][b2][3];
  if (threadIdx.x == 0) {
    p.out[blockIdx.x] = (t1 - t0) + (sum == 12345 ? 1 : 0);
    p.out[256 + blockIdx.x] = r1 - r0;
  }
#endif
}

template <int WM, int MI, int DIST, int FLAGS>
void run(const char *name, const char *w, const char *a, long long *out) {
  constexpr int LDS = (256 + 320) * 128 * 2;
  hipFuncSetAttribute((const void *)tile_kernel<WM, MI, DIST, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  float best = 1e9;
  double cyc = 0, rt = 0;
  for (int rep = 0; rep < 8; ++rep) {
    P p{w, a, 2048, out};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tile_kernel<WM, MI, DIST, FLAGS>), dim3(256), dim3(128 * WM), LDS, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      long long h[512]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      cyc = rt = 0;
      for (int i = 0; i < 256; ++i) { cyc += h[i]; rt += h[256 + i]; }
      cyc /= 256; rt /= 256;
    }
  }
  printf("%-40s %6.1f us launch, %6.2f us in-kernel, %7.0f cycles/block, %5.0f per k-step, clock %.2f GHz\n", name, best * 1e3,
         rt / 100.0, cyc, cyc / 16, cyc / (rt * 10.0));
}

int main() {
  char *w, *a; long long *out;
  hipMalloc(&w, (size_t)2048 * 2048); hipMalloc(&a, (size_t)10240 * 2048); hipMalloc(&out, 512 * 8);
  hipMemset(w, 1, (size_t)2048 * 2048); hipMemset(a, 2, (size_t)10240 * 2048);
  run<4, 2, 3330, 6>("8 waves, ROTATED, 3/3/3/0, no setprio (ships)", w, a, out);
  run<4, 2, 3330, 14>("8 waves, ROTATED, 3/3/3/0, reads pinned behind the MFMAs", w, a, out);
  run<4, 2, 5400, 14>("8 waves, ROTATED, 5/4/0/0, reads pinned", w, a, out);
  run<4, 2, 9000, 14>("8 waves, ROTATED, 9/0/0/0, reads pinned", w, a, out);
  run<4, 2, 2340, 14>("8 waves, ROTATED, 2/3/4/0, reads pinned", w, a, out);
  run<4, 2, 3330, 12>("8 waves, ROTATED, 3/3/3/0, reads pinned, setprio", w, a, out);
  run<4, 2, 3222, 0>("8 waves 64x160, loads 3/2/2/2 (classic)", w, a, out);
  run<4, 2, 9000, 0>("8 waves, loads 9/0/0/0", w, a, out);
  run<4, 2, 5400, 0>("8 waves, loads 5/4/0/0", w, a, out);
  run<4, 2, 3330, 0>("8 waves, loads 3/3/3/0", w, a, out);
  run<4, 2, 4320, 0>("8 waves, loads 4/3/2/0", w, a, out);
  run<4, 2, 2223, 0>("8 waves, loads 2/2/2/3", w, a, out);
  run<4, 2, 342, 0>("8 waves, loads 0/3/4/2", w, a, out);
  run<4, 2, 3222, 1>("8 waves, 3/2/2/2, NO barrier", w, a, out);
  run<4, 2, 3222, 2>("8 waves, 3/2/2/2, no setprio", w, a, out);
  run<4, 2, 5400, 4>("8 waves, ROTATED barrier, loads 5/4/0/0", w, a, out);
  run<4, 2, 3330, 4>("8 waves, ROTATED barrier, loads 3/3/3/0", w, a, out);
  run<4, 2, 9000, 4>("8 waves, ROTATED barrier, loads 9/0/0/0", w, a, out);
  run<4, 2, 3222, 4>("8 waves, ROTATED barrier, loads 3/2/2/2", w, a, out);
  run<4, 2, 2340, 4>("8 waves, ROTATED barrier, loads 2/3/4/0", w, a, out);
  run<4, 2, 5400, 6>("8 waves, ROTATED, 5/4/0/0, no setprio", w, a, out);
  run<4, 2, 3330, 6>("8 waves, ROTATED, 3/3/3/0, no setprio", w, a, out);
  run<4, 2, 2340, 6>("8 waves, ROTATED, 2/3/4/0, no setprio", w, a, out);
  run<4, 2, 4500, 6>("8 waves, ROTATED, 4/5/0/0, no setprio", w, a, out);
  run<4, 2, 2430, 6>("8 waves, ROTATED, 2/4/3/0, no setprio", w, a, out);
  run<4, 2, 1440, 6>("8 waves, ROTATED, 1/4/4/0, no setprio", w, a, out);
  run<4, 2, 450, 6>("8 waves, ROTATED, 0/4/5/0, no setprio", w, a, out);
  run<4, 2, 5400, 2>("8 waves, classic, 5/4/0/0, no setprio", w, a, out);
  run<4, 2, 6300, 6>("8 waves, ROTATED, 6/3/0/0, no setprio", w, a, out);
  run<4, 2, 7200, 6>("8 waves, ROTATED, 7/2/0/0, no setprio", w, a, out);
  run<4, 2, 9000, 6>("8 waves, ROTATED, 9/0/0/0, no setprio", w, a, out);
  run<4, 2, 3600, 6>("8 waves, ROTATED, 3/6/0/0, no setprio", w, a, out);
  run<4, 2, 5220, 6>("8 waves, ROTATED, 5/2/2/0, no setprio", w, a, out);
  run<4, 2, 5400, 7>("8 waves, ROTATED, 5/4/0/0, no setprio, NO barrier", w, a, out);
  run<2, 4, 5544, 0>("4 waves, wave tile 128x160", w, a, out);
  return 0;
}
