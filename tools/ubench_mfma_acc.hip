// ubench_mfma_acc.hip -- how v_mfma_f32_32x32x16_f16 accumulates: what the screened layer 0's error bound may assume.
//   hipcc --offload-arch=gfx950 -O2 -o ubench_mfma_acc ubench_mfma_acc.hip
// For many (C, a[16], b[16]) per output it compares D with the exactly rounded C + sum a_k b_k (long double: the
// generated exponent spreads keep the exact sum inside 64 bits) and with candidate evaluation orders, and prints the
// largest |D - exact| in units of u * M for M = |C| + sum |a_k b_k| and for M = max(|C|, max |a_k b_k|, |exact|).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A [32][16] halfs (row-major), B [32][16] halfs (column n, k contiguous), C / D [32][32] floats; one wave per problem
__global__ __launch_bounds__(64) void mfma_probe(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
  const int lane = threadIdx.x, l32 = lane & 31, h = lane >> 5;
  const size_t pb = blockIdx.x;
  const v8h a = *reinterpret_cast<const v8h *>(A + pb * 512 + l32 * 16 + 8 * h);
  const v8h b = *reinterpret_cast<const v8h *>(B + pb * 512 + l32 * 16 + 8 * h);
  v16f c;
  for (int r = 0; r < 16; ++r) c[r] = C[pb * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32];
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[pb * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l32] = c[r];
}

static float h2f(_Float16 h) { return static_cast<float>(h); }

int main() {
  const int P = 4096;
  std::vector<_Float16> A(P * 512), B(P * 512);
  std::vector<float> C(P * 1024), D(P * 1024);
  _Float16 *dA, *dB;
  float *dC, *dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  std::mt19937_64 rng(12345);
  std::normal_distribution<float> gauss(0.f, 1.f);
  const char *names[] = {"gauss, C ~ sum", "gauss, C = 0", "wide exponents", "cancel: C = -sum p", "C huge", "ties: p = 1, C = 2^24", "fp16 denormal inputs", "layer-0 like: x1 w1 pieces"};
  for (int mode = 0; mode < 8; ++mode) {
    for (int p = 0; p < P; ++p) {
      for (int i = 0; i < 512; ++i) {
        float a = gauss(rng), b = gauss(rng);
        if (mode == 2) { a = std::ldexp(a, int(rng() % 21) - 10); b = std::ldexp(b, int(rng() % 21) - 10); }
        if (mode == 5) { a = 1.0f; b = 1.0f; }
        if (mode == 6) { a = std::ldexp(a, -18); b = std::ldexp(b, 4); }
        if (mode == 7) { a = std::ldexp(a, 12); b = std::ldexp(b, 12); }
        A[p * 512 + i] = static_cast<_Float16>(a);
        B[p * 512 + i] = static_cast<_Float16>(b);
      }
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          long double s = 0;
          for (int k = 0; k < 16; ++k) s += (long double)h2f(A[p * 512 + i * 16 + k]) * h2f(B[p * 512 + j * 16 + k]);
          float c = 0;
          switch (mode) {
            case 0: c = 4.0f * gauss(rng); break;
            case 1: c = 0; break;
            case 2: c = std::ldexp(gauss(rng), int(rng() % 31) - 15); break;
            case 3: c = -static_cast<float>(s) * (1.0f + 1e-3f * gauss(rng)); break;
            case 4: c = std::ldexp(gauss(rng), 20); break;
            case 5: c = 16777216.0f + 2.0f * float(rng() % 8); break;
            case 6: c = std::ldexp(gauss(rng), -16); break;
            case 7: c = std::ldexp(gauss(rng), 27); break;
          }
          C[p * 1024 + i * 32 + j] = c;
        }
    }
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(P), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    const double u = std::ldexp(1.0, -24);
    double worst_sum = 0, worst_max = 0, worst_res = 0;
    long eq_exact_rn = 0, eq_exact_rz = 0, eq_seq = 0, eq_seq4 = 0, eq_seq8 = 0, eq_half = 0, total = 0;
    for (int p = 0; p < P; ++p)
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          const float c = C[p * 1024 + i * 32 + j], d = D[p * 1024 + i * 32 + j];
          long double ex = c, mag = std::fabs((long double)c), mx = std::fabs((long double)c);
          float seq = c;
          long double g4[4] = {0, 0, 0, 0}, g8[2] = {0, 0};
          for (int k = 0; k < 16; ++k) {
            const float a = h2f(A[p * 512 + i * 16 + k]), b = h2f(B[p * 512 + j * 16 + k]);
            const long double pr = (long double)a * b;
            ex += pr; mag += std::fabs(pr); mx = std::fmax(mx, std::fabs(pr));
            seq = std::fmaf(a, b, seq);
            g4[k / 4] += pr; g8[k / 8] += pr;
          }
          mx = std::fmax(mx, std::fabs(ex));
          const float rn = static_cast<float>(ex);  // long double -> float: round to nearest even
          float rz = rn;
          if (std::fabs((long double)rz) > std::fabs(ex)) rz = std::nextafterf(rz, 0.0f);
          float s4 = c; for (int g = 0; g < 4; ++g) s4 = static_cast<float>((long double)s4 + g4[g]);
          float s8 = c; for (int g = 0; g < 2; ++g) s8 = static_cast<float>((long double)s8 + g8[g]);
          // the two lane halves (k 0..7, 8..15) as two exactly summed groups added to C one after the other = s8; and
          // as: products summed exactly first, rounded, then added to C
          const float hsum = static_cast<float>((long double)c + (long double)static_cast<float>(g8[0] + g8[1]));
          ++total;
          eq_exact_rn += d == rn; eq_exact_rz += d == rz; eq_seq += d == seq; eq_seq4 += d == s4; eq_seq8 += d == s8; eq_half += d == hsum;
          const double err = std::fabs((double)((long double)d - ex));
          if (mag > 0) worst_sum = std::fmax(worst_sum, err / (u * (double)mag));
          if (mx > 0) worst_max = std::fmax(worst_max, err / (u * (double)mx));
          if (d != 0) worst_res = std::fmax(worst_res, err / (u * std::fabs((double)d)));
        }
    std::printf("mode %d (%s): n=%ld  == exact RN %.4f  == exact RZ %.4f  == fmaf chain %.4f  == 4 exact groups of 4 %.4f  == 2 groups of 8 %.4f  == C + fl(sum) %.4f\n"
                "    worst |D-exact| / (u (|C| + sum|p|)) = %.4f   / (u max(|C|, max|p|, |exact|)) = %.4f   / (u |D|) = %.4f\n",
                mode, names[mode], total, double(eq_exact_rn) / total, double(eq_exact_rz) / total, double(eq_seq) / total, double(eq_seq4) / total,
                double(eq_seq8) / total, double(eq_half) / total, worst_sum, worst_max, worst_res);
  }
  return 0;
}
