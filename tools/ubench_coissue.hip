// ubench_coissue.hip -- do a wave's int8 MFMAs and ANOTHER wave's vector instructions on the same SIMD overlap on gfx950?
//
// Rounds 3-4 concluded from ablations of the product kernels that "vector time and MFMA time add"; the guide
// (MI355X_MICROARCH.md, "Two waves per SIMD") says a MFMA-only wave and a VALU-only wave run concurrently.  The product
// kernels' waves are barrier-locked into the same phase and their vector work reads the accumulators it follows, so
// they cannot tell.  This does, with nothing else in the way:
//
//   one workgroup of 512 threads per CU = two waves per SIMD (waves w and w + 4 share SIMD w % 4);
//   role A (waves 0-3): N back-to-back independent v_mfma_i32_32x32x32_i8 on four rotating accumulators, never read;
//   role B (waves 4-7): M independent vector instructions of one kind (v_fma_f32, v_pk_fma_f32, v_exp_f32,
//                       v_cvt_f32_i32, ds_read_b128, an epilogue-like mix);
//   timed per wave with s_memtime: A alone (B waves leave at once), B alone, A and B together; optionally s_setprio on
//   either role; and the one-wave form: K vector instructions between consecutive MFMAs of ONE wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_coissue ubench_coissue.hip && ./ubench_coissue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)

enum { B_FMA = 0, B_PKFMA = 1, B_EXP = 2, B_CVT = 3, B_DSREAD = 4, B_MIX = 5, B_KINDS = 6 };
static const char *kBName[B_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_f32_i32", "ds_read_b128", "epilogue mix"};

// One loop iteration of role A: 32 MFMAs (4 accumulators, each written 8 times; dependent MFMAs on one accumulator are
// 4 apart = 128 cycles > the 32x32x32 i8 latency of 16 passes x 4 = 64 cycles: issue-bound, not latency-bound).
__device__ __forceinline__ void mfma_block(v16i &c0, v16i &c1, v16i &c2, v16i &c3, v4i a, v4i b) {
  asm volatile(REP8("v_mfma_i32_32x32x32_i8 %0, %4, %5, %0\n\t"
                    "v_mfma_i32_32x32x32_i8 %1, %4, %5, %1\n\t"
                    "v_mfma_i32_32x32x32_i8 %2, %4, %5, %2\n\t"
                    "v_mfma_i32_32x32x32_i8 %3, %4, %5, %3\n\t")
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
               : "v"(a), "v"(b));
}

// One loop iteration of role B: 64 independent instructions of the kind (8 registers x 8).
template <int KIND>
__device__ __forceinline__ void valu_block(float (&r)[8], v2f (&p)[4], v4i (&d)[4], int (&iv)[8], float k0, float k1, uint32_t lds_addr) {
  if (KIND == B_FMA) {
    asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                      "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t")
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "v"(k0), "v"(k1));
  } else if (KIND == B_PKFMA) {
    v2f kk0 = {k0, k0}, kk1 = {k1, k1};
    asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t")
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                 : "v"(kk0), "v"(kk1));
  } else if (KIND == B_EXP) {
    asm volatile(REP8("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                      "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  } else if (KIND == B_CVT) {
    asm volatile(REP8("v_cvt_f32_i32 %0, %8\n\tv_cvt_f32_i32 %1, %9\n\tv_cvt_f32_i32 %2, %10\n\tv_cvt_f32_i32 %3, %11\n\t"
                      "v_cvt_f32_i32 %4, %12\n\tv_cvt_f32_i32 %5, %13\n\tv_cvt_f32_i32 %6, %14\n\tv_cvt_f32_i32 %7, %15\n\t")
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "v"(iv[0]), "v"(iv[1]), "v"(iv[2]), "v"(iv[3]), "v"(iv[4]), "v"(iv[5]), "v"(iv[6]), "v"(iv[7]));
  } else if (KIND == B_DSREAD) {  // 64 conflict-free 16-byte reads, drained every 16
    asm volatile(REP4(REP4("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t")
                      "s_waitcnt lgkmcnt(0)\n\t")
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(lds_addr));
  } else {  // what an output value costs in the hidden epilogue: cvt, mul, fma, fma, add, mul, cvt, med3 -- 8 values
    asm volatile(REP8("v_cvt_f32_i32 %0, %8\n\tv_mul_f32 %1, %0, %16\n\tv_fma_f32 %2, %1, %17, %0\n\tv_fma_f32 %3, %2, %16, %1\n\t"
                      "v_add_f32 %4, %3, %17\n\tv_mul_f32 %5, %4, %16\n\tv_cvt_i32_f32 %6, %5\n\tv_med3_i32 %7, %6, %9, %10\n\t")
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "v"(iv[0]), "v"(iv[1]), "v"(iv[2]), "v"(iv[3]), "v"(iv[4]), "v"(iv[5]), "v"(iv[6]), "v"(iv[7]), "v"(k0), "v"(k1));
  }
}

struct Rec {
  long long a_cycles, b_cycles;
};

// mode bits: 1 = role A runs, 2 = role B runs; prio_a / prio_b = s_setprio value of the role
template <int KIND>
__global__ __launch_bounds__(512, 2) void coissue_kernel(long long *out, int iters_a, int iters_b, int mode, int prio_a, int prio_b) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_a = wave < 4;
  for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<int *>(lds)[i] = i;
  __syncthreads();
  long long t0 = 0, t1 = 0;
  if (role_a) {
    if (!(mode & 1)) return;
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v4i a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
    if (prio_a == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_a == 2) __builtin_amdgcn_s_setprio(2);
    if (prio_a == 3) __builtin_amdgcn_s_setprio(3);
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters_a; ++it) mfma_block(c0, c1, c2, c3, a, b);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    t1 = __builtin_readcyclecounter();
    int keep = 0;
    for (int i = 0; i < 16; ++i) keep ^= c0[i] ^ c1[i] ^ c2[i] ^ c3[i];
    if (keep == 0x7fffffff) out[0] = keep;
  } else {
    if (!(mode & 2)) return;
    float r[8];
    v2f p[4];
    v4i d[4] = {};
    int iv[8];
    for (int i = 0; i < 8; ++i) {
      r[i] = 0.001f * (lane + i);
      iv[i] = lane * 3 + i;
    }
    for (int i = 0; i < 4; ++i) p[i] = v2f{0.01f * lane, 0.02f * i};
    const uint32_t lds_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds)) + (wave - 4) * 4096 + lane * 16;
    if (prio_b == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_b == 2) __builtin_amdgcn_s_setprio(2);
    if (prio_b == 3) __builtin_amdgcn_s_setprio(3);
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters_b; ++it) valu_block<KIND>(r, p, d, iv, 0.999f, 0.0001f, lds_addr);
    t1 = __builtin_readcyclecounter();
    float keep = 0;
    for (int i = 0; i < 8; ++i) keep += r[i];
    for (int i = 0; i < 4; ++i) keep += p[i].x + p[i].y + static_cast<float>(d[i].x);
    if (keep == 1234.5f) out[1] = 1;
  }
  if (lane == 0) out[8 + (static_cast<size_t>(blockIdx.x) * 8 + wave) * 2 + 0] = t1 - t0;
}

// ONE wave per SIMD (256-thread workgroup): K vector instructions between consecutive MFMAs
template <int K>
__global__ __launch_bounds__(256, 1) void interleave_kernel(long long *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
  v4i a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 0.001f * (lane + i);
  const float k0 = 0.999f, k1 = 0.0001f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define FILL(n)                                                                                          \
  if (K > n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[n & 7]) : "v"(k0), "v"(k1));
#define ONE(cx)                                                                                          \
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(cx) : "v"(a), "v"(b));                      \
  FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) FILL(7) FILL(8) FILL(9) FILL(10) FILL(11)
    ONE(c0) ONE(c1) ONE(c2) ONE(c3) ONE(c0) ONE(c1) ONE(c2) ONE(c3)
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  int keep = 0;
  for (int i = 0; i < 16; ++i) keep ^= c0[i] ^ c1[i] ^ c2[i] ^ c3[i];
  float kf = 0;
  for (int i = 0; i < 8; ++i) kf += r[i];
  if (keep == 0x7fffffff || kf == 1234.5f) out[0] = keep;
  if (lane == 0) out[8 + (static_cast<size_t>(blockIdx.x) * 4 + wave) * 2] = t1 - t0;
}

static long long *d_out;
static std::vector<long long> h_out;

template <int KIND>
static void run_pair(int blocks, int prio_a, int prio_b) {
  const int ia = 400, ib = 400;  // A: 400 x 32 MFMAs x 32 cycles = 410 k cycles alone
  double res[3][2] = {};
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d_out, 0, h_out.size() * 8);
      hipLaunchKernelGGL(coissue_kernel<KIND>, dim3(blocks), dim3(512), 0, 0, d_out, ia, ib, mode, prio_a, prio_b);
      hipDeviceSynchronize();
    }
    hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += double(h_out[8 + (size_t(b) * 8 + w) * 2]);
    res[mode - 1][0] = sa / (blocks * 4);
    res[mode - 1][1] = sb / (blocks * 4);
  }
  const double a_alone = res[0][0], b_alone = res[1][1], a_both = res[2][0], b_both = res[2][1];
  // if the two add, the longer role ends at a_alone + b_alone; if they overlap, at max(a_alone, b_alone)
  const double longer = a_both > b_both ? a_both : b_both, mx = a_alone > b_alone ? a_alone : b_alone, sum = a_alone + b_alone;
  printf("%-14s blocks %3d prio A/B %d/%d | A alone %7.0f (%.1f cyc/MFMA)  B alone %7.0f (%.2f cyc/inst) | together: A %7.0f  B %7.0f | overlap %.2f (1 = max, 0 = sum)\n",
         kBName[KIND], blocks, prio_a, prio_b, a_alone, a_alone / (ia * 32.0), b_alone, b_alone / (ib * 64.0), a_both, b_both, (sum - longer) / (sum - mx));
}

template <int K>
static void run_interleave(int blocks) {
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d_out, 0, h_out.size() * 8);
    hipLaunchKernelGGL(interleave_kernel<K>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 4; ++w) s += double(h_out[8 + (size_t(b) * 4 + w) * 2]);
  printf("one wave per SIMD, %2d v_fma_f32 between MFMAs: %.1f cycles per MFMA\n", K, s / (blocks * 4) / (iters * 8.0));
}

int main() {
  h_out.resize(8 + 256 * 8 * 2);
  hipMalloc(&d_out, h_out.size() * 8);
  for (int blocks : {1, 256}) {
    run_pair<B_FMA>(blocks, 0, 0);
    run_pair<B_PKFMA>(blocks, 0, 0);
    run_pair<B_EXP>(blocks, 0, 0);
    run_pair<B_CVT>(blocks, 0, 0);
    run_pair<B_DSREAD>(blocks, 0, 0);
    run_pair<B_MIX>(blocks, 0, 0);
  }
  run_pair<B_MIX>(256, 1, 0);
  run_pair<B_MIX>(256, 0, 1);
  run_pair<B_FMA>(256, 1, 0);
  run_pair<B_FMA>(256, 0, 1);
  run_interleave<0>(256);
  run_interleave<1>(256);
  run_interleave<2>(256);
  run_interleave<3>(256);
  run_interleave<4>(256);
  run_interleave<5>(256);
  run_interleave<6>(256);
  run_interleave<8>(256);
  run_interleave<12>(256);
  return 0;
}
