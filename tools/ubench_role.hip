// ubench_role.hip -- the prerequisite of every overlap design: can ONE wave per SIMD keep the
// matrix pipe busy if its partner on the SIMD takes everything that is not a fragment read or an MFMA?
//
// One 512-thread workgroup per CU, two GROUPS of four waves (waves w and w + 4 share a SIMD).  The groups ALTERNATE at
// tile granularity: while group A runs the 16 k-steps of its 256-node x 160-frame tile (28 ds_read_b128 + 40 MFMA per
// wave and k-step, nothing else), group B is in the SUPPORT role -- it issues ALL LDS-DMA loads of A's next stage (13
// one-KiB pieces per wave and k-step) and runs V vector instructions per k-step (the stand-in for its own tile's
// epilogue); one s_barrier per k-step, before the compute waves' last sub-step (the rotated form of fdnn_gemm.hip).  Because
// only one group stages operands at a time the ring holds ONE group's operands: 2 x 52 KB.
//
// Printed: cycles per k-step ("tick") in steady state against the 1 280 cycles of MFMA issue a tick contains, for a
// range of V -- how much vector work rides for free under the partner's k-loop.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_role ubench_role.hip && ./ubench_role
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void *)(p))
struct P {
  const char *w;  // [layers][2048][K]
  const char *a;  // [rows][K]
  int K, rows, layers, phases;
  long long *out;  // [256][64]: wave 0's clock at each phase start
};

__device__ __forceinline__ v4i read_frag(const char *tile, int row, int chunk) {
  return *reinterpret_cast<const v4i *>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// FLAGS: 1 setprio(1) around each MFMA block of the compute role; 2 the support role spreads its loads over the vector
// piece (one load every V/13 instructions) instead of issuing them as a burst; 4 no loads at all (pure co-issue);
// 8 no MFMAs (support alone); 16 the vector piece contains a transcendental every 8th instruction and an LDS write/read
// pair every 16th (closer to an epilogue); 32 static setprio(1) for whoever is in the compute role
template <int V, int FLAGS>
__global__ __launch_bounds__(512, 2) void role_kernel(P p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NF = 5, BK = 128, WROWS = 256, AROWS = 160;
  constexpr bool W3 = (FLAGS & 64) != 0;  // weights three stages deep (issued two ticks ahead), activation rows two
  constexpr int NWS = W3 ? 3 : 2;
  // separate static arrays: the compiler then knows that the vector piece's LDS traffic cannot alias the LDS-DMA destinations
  // (through one extern block every ds_read waits for every load in flight: 52 k cycles per tick, first run)
  __shared__ __attribute__((aligned(16))) char ringW[NWS][WROWS * BK];
  __shared__ __attribute__((aligned(16))) char ringA[2][AROWS * BK];
  __shared__ float scratch_all[8 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 3, grp = wave >> 2;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int mt = j >> 2, nt0 = xcd * 4 + (j & 3);
  const int NTILES = p.rows / 320;
  const int srow = lane >> 3, schunk = ((lane & 7) ^ (((wm * 8 + srow) >> 1) & 7)) << 4;
  const int voff = srow * p.K + schunk;
  float *scratch = scratch_all + wave * 64;

  // the tile group g computes in phase ph (ph & 1 == g): layer = (ph >> 1) % layers, frame tile moves on every cycle
  auto rsrc_w_of = [&](int ph) {
    const int layer = (ph >> 1) % p.layers;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.w + ((size_t)layer * 2048 + (size_t)mt * WROWS) * p.K), 0, WROWS * p.K, 0x00020000);
  };
  auto rsrc_a_of = [&](int ph) {
    const int nt = (nt0 + 5 * (ph >> 1)) % NTILES;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.a + ((size_t)nt * 320 + (size_t)(ph & 1) * AROWS) * p.K), 0, AROWS * p.K, 0x00020000);
  };
  // piece i of stage kt: i < 8 weights (into weight buffer wbuf), else activation rows (into buffer kt & 1)
  auto stage_load = [&](__amdgpu_buffer_rsrc_t rw, __amdgpu_buffer_rsrc_t ra, int kt, int wbuf, int i) {
    if (FLAGS & 4) return;
    if (i < 8) {
      const int slab = i * 4 + wm;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDSP(&ringW[wbuf][slab * 1024]), 16, voff, slab * 8 * p.K + kt * BK, 0, 0);
    } else {
      const int slab = (i - 8) * 4 + wm;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDSP(&ringA[kt & 1][slab * 1024]), 16, voff, slab * 8 * p.K + kt * BK, 0, 0);
    }
  };

  v16i acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NF; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
  const int frow = lane & 31, fch = lane >> 5;
  v4i fa[2][2], fb[2][NF];
  auto load_frags = [&](int kt, int kk, int set) {
    const char *wt = ringW[kt % NWS], *at = ringA[kt & 1];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) fa[set][mi] = read_frag(wt, 64 * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) fb[set][ni] = read_frag(at, 32 * ni + frow, kk * 2 + fch);
  };
  auto mfmas = [&](int set) {
    if (FLAGS & 8) return;
    if (FLAGS & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][mi], fb[set][ni], acc[mi][ni], 0, 0, 0);
    if (FLAGS & 1) __builtin_amdgcn_s_setprio(0);
  };
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 0.001f + i;
  const float c0 = 0.999f, c1 = 0.0001f;

  // prologue: stage 0 of phase 0 (group 0's tile) -- W3: weights of stages 0 and 1 -- issued by group 1
  if (grp == 1) {
    const auto rw = rsrc_w_of(0), ra = rsrc_a_of(0);
#pragma unroll
    for (int i = 0; i < 13; ++i) stage_load(rw, ra, 0, 0, i);
    if (W3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stage_load(rw, ra, 1, 1, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int T_end = p.phases * 16;
  for (int ph = 0; ph < p.phases; ++ph) {
    if (wave == 0 && lane == 0) p.out[blockIdx.x * 64 + ph] = __builtin_readcyclecounter();
    if ((ph & 1) == grp) {
      // ------------------------------------------------ compute role: fragment reads and MFMAs only
      if (FLAGS & 32) __builtin_amdgcn_s_setprio(1);
      load_frags(ph * 16, 0, 0);
      const auto ra_c = rsrc_a_of(ph), ra_cn = rsrc_a_of(ph + 1);
      const auto rw_c = rsrc_w_of(ph);
      for (int kt = 0; kt < 16; ++kt) {
        const int T = ph * 16 + kt;
        if ((FLAGS & 512) && T + 1 < T_end) {  // the COMPUTE wave requests its own next activation rows (5 of the tick's 13 pieces)
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            if (kt == 15) stage_load(rw_c, ra_cn, 0, 0, 8 + i);
            else stage_load(rw_c, ra_c, kt + 1, 0, 8 + i);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          if (FLAGS & 128) __builtin_amdgcn_sched_barrier(0);
          load_frags(T, kk + 1, (kk + 1) & 1);
          if (FLAGS & 128) __builtin_amdgcn_sched_barrier(0);  // the next sub-step's reads go out BEFORE this sub-step's MFMAs
          mfmas(kk & 1);
          if (FLAGS & 256) {  // one read behind each of the first seven MFMAs
#pragma unroll
            for (int i = 0; i < 7; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
          }
        }
        if (FLAGS & 384) __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 512) __builtin_amdgcn_s_waitcnt(0x0f70);  // my rows of the next stage have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage's fragments are all in registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt < 15) load_frags(T + 1, 0, 0);
        if (FLAGS & 128) __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        if (FLAGS & 256) {
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
        }
        if (FLAGS & 384) __builtin_amdgcn_sched_barrier(0);
      }
      if (FLAGS & 32) __builtin_amdgcn_s_setprio(0);
    } else {
      // ------------------------------------------------ support role: the partner's loads + my vector work
      const auto rw = rsrc_w_of(ph), ra = rsrc_a_of(ph);
      const auto rw_n = rsrc_w_of(ph + 1), ra_n = rsrc_a_of(ph + 1);
      for (int kt = 0; kt < 16; ++kt) {
        const int T = ph * 16 + kt;
        // 2 stages: pieces 0..12 = stage T+1 (weights, then rows).  W3: pieces 0..4 = the ROWS of stage T+1, 5..12 = the WEIGHTS of stage T+2
        auto piece = [&](int i) {
          if (!W3) {
            if (T + 1 >= T_end) return;
            if (kt == 15) stage_load(rw_n, ra_n, 0, (T + 1) & 1, i);
            else stage_load(rw, ra, kt + 1, (T + 1) & 1, i);
          } else if (i < 5) {
            if (T + 1 >= T_end) return;
            if (kt == 15) stage_load(rw_n, ra_n, 0, 0, 8 + i);
            else stage_load(rw, ra, kt + 1, 0, 8 + i);
          } else {
            if (T + 2 >= T_end) return;
            if (kt >= 14) stage_load(rw_n, ra_n, kt - 14, (T + 2) % 3, i - 5);
            else stage_load(rw, ra, kt + 2, (T + 2) % 3, i - 5);
          }
        };
        if (!(FLAGS & 2)) {
#pragma unroll
          for (int i = (FLAGS & 512) ? 5 : 0; i < 13; ++i) piece(i);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if ((FLAGS & 2) && V >= 13 && v % (V / 13) == 0 && v / (V / 13) < 13) piece(v / (V / 13));
          if ((FLAGS & 16) && v % 8 == 7) {
            x[v & 7] = __builtin_amdgcn_exp2f(x[v & 7]);
          } else if ((FLAGS & 16) && v % 16 == 3) {
            scratch[lane] = x[v & 7];
            asm volatile("" ::: "memory");
            x[v & 7] += scratch[lane ^ 1];
          } else {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c0), "v"(c1));
          }
        }
        if (W3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but the youngest weight stage has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  int sum = 0;
#pragma unroll
  for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
    for (int b2 = 0; b2 < NF; ++b2) sum += acc[a2][b2][3];
  float xs = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) xs += x[i];
  if (wave == 0 && lane == 0) p.out[blockIdx.x * 64 + p.phases] = t1 + ((sum == 12345 || xs == 1.2345f) ? 1 : 0);
  if (sum == 12345 || xs == 1.2345f) p.out[0] = 0;
#endif
}

template <int V, int FLAGS>
void run(const char *name, const char *w, const char *a, long long *out) {
  constexpr int LDS = 0;  // (static arrays)
  const int phases = 12;
  float best = 1e9;
  double tick_all = 0, tick_steady = 0;
  for (int rep = 0; rep < 5; ++rep) {
    P p{w, a, 2048, 10240, 6, phases, out};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((role_kernel<V, FLAGS>), dim3(256), dim3(512), LDS, 0, p);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      static long long h[256 * 64];
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      tick_all = tick_steady = 0;
      for (int b = 0; b < 256; ++b) {
        tick_all += double(h[b * 64 + phases] - h[b * 64]) / (phases * 16);
        tick_steady += double(h[b * 64 + phases - 2] - h[b * 64 + 2]) / ((phases - 4) * 16);
      }
      tick_all /= 256;
      tick_steady /= 256;
    }
  }
  // one tick = one group's k-step: 40 MFMAs per SIMD = 1 280 cycles of issue; a 320-frame tile costs 32 ticks
  printf("%-64s %7.1f us | tick %5.0f cycles (steady %5.0f) | MFMA issue %.2f | 320-frame tile %6.0f cycles\n", name, best * 1e3, tick_all, tick_steady,
         1280.0 / tick_steady, 32 * tick_steady);
}

int main() {
  char *w, *a;
  long long *out;
  hipMalloc(&w, (size_t)6 * 2048 * 2048);
  hipMalloc(&a, (size_t)10240 * 2048);
  hipMalloc(&out, 256 * 64 * 8);
  hipMemset(w, 1, (size_t)6 * 2048 * 2048);
  hipMemset(a, 2, (size_t)10240 * 2048);
  run<0, 4>("compute alone, compiler's order (ceiling)", w, a, out);
  run<0, 132>("compute alone, reads pinned before the MFMAs", w, a, out);
  run<0, 260>("compute alone, one read behind each MFMA", w, a, out);
  run<0, 64>("W3/A2 loads by the partner, compiler's order", w, a, out);
  run<0, 192>("W3/A2 loads, reads pinned first", w, a, out);
  run<0, 320>("W3/A2 loads, reads interleaved", w, a, out);
  run<104, 192>("W3/A2, pinned, V = 104", w, a, out);
  run<156, 192>("W3/A2, pinned, V = 156", w, a, out);
  run<208, 192>("W3/A2, pinned, V = 208", w, a, out);
  run<104, 320>("W3/A2, interleaved, V = 104", w, a, out);
  run<156, 320>("W3/A2, interleaved, V = 156", w, a, out);
  run<208, 320>("W3/A2, interleaved, V = 208", w, a, out);
  run<156, 321>("W3/A2, interleaved, V = 156, setprio around MFMA blocks", w, a, out);
  run<156, 352>("W3/A2, interleaved, V = 156, static setprio", w, a, out);
  run<156, 128>("2 stages, pinned, V = 156", w, a, out);
  run<0, 832>("rows requested by the COMPUTE wave, W by the partner, V = 0", w, a, out);
  run<104, 832>("rows requested by the COMPUTE wave, V = 104", w, a, out);
  run<156, 832>("rows requested by the COMPUTE wave, V = 156", w, a, out);
  run<208, 832>("rows requested by the COMPUTE wave, V = 208", w, a, out);
  return 0;
}
