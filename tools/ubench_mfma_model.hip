// ubench_mfma_model.hip -- which evaluation model reproduces v_mfma_f32_32x32x16_f16 bit for bit?
//   hipcc --offload-arch=gfx950 -O2 -o ubench_mfma_model ubench_mfma_model.hip
// Hypotheses: the 16 products are consumed in `groups` (1 x 16, 2 x 8, 4 x 4) one after the other; inside a group the
// accumulator and the group's exact products are aligned to the largest exponent among them (the product's exponent
// taken as the true one or as exp(a) + exp(b)), each term cut to g bits below the fp32 ulp of that exponent (toward
// zero or toward -inf), the cut terms summed exactly and the sum rounded to fp32 (nearest-even or toward zero).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

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

static float to_f32(double v, bool rz) {
  float f = static_cast<float>(v);  // round to nearest even
  if (rz && std::fabs(static_cast<double>(f)) > std::fabs(v)) f = std::nextafterf(f, 0.0f);
  return f;
}

struct Hyp { int groups, g; bool sum_exp, floor_cut, rz; };

static float eval(const Hyp &h, float c, const float *a, const float *b) {
  const int per = 16 / h.groups;
  double acc = c;
  for (int gi = 0; gi < h.groups; ++gi) {
    double t[17];
    int e[17], n = 0;
    if (acc != 0) { t[n] = acc; e[n] = std::ilogb(acc); ++n; }
    for (int k = gi * per; k < (gi + 1) * per; ++k) {
      const double p = static_cast<double>(a[k]) * b[k];
      if (p == 0) continue;
      t[n] = p;
      e[n] = h.sum_exp ? std::ilogb(a[k]) + std::ilogb(b[k]) : std::ilogb(p);
      ++n;
    }
    if (n == 0) { acc = 0; continue; }
    int emax = e[0];
    for (int i = 1; i < n; ++i) emax = e[i] > emax ? e[i] : emax;
    const double q = std::ldexp(1.0, emax - 23 - h.g);
    double s = 0;  // integers below 2^53: exact
    for (int i = 0; i < n; ++i) s += h.floor_cut ? std::floor(t[i] / q) : std::trunc(t[i] / q);
    acc = to_f32(s * q, h.rz);
  }
  return static_cast<float>(acc);
}

int main() {
  const int P = 256;
  std::vector<_Float16> A(P * 512), B(P * 512);
  std::vector<float> C(P * 1024), D(P * 1024), Af(P * 512), Bf(P * 512);
  _Float16 *dA, *dB;
  float *dC, *dD;
  (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
  std::mt19937_64 rng(777);
  std::normal_distribution<float> gauss(0.f, 1.f);
  std::vector<Hyp> hyps;
  for (int groups : {1, 2, 4})
    for (int g = 0; g <= 8; ++g)
      for (int se = 0; se < 2; ++se)
        for (int fc = 0; fc < 2; ++fc)
          for (int rz = 0; rz < 2; ++rz) hyps.push_back({groups, g, se != 0, fc != 0, rz != 0});
  hyps.push_back({2, 40, false, false, false});  // exact groups of 8
  hyps.push_back({1, 40, false, false, false});  // exact
  std::vector<long> total_hits(hyps.size(), 0);
  long total = 0;
  const char *names[] = {"gauss, C ~ sum", "wide exponents", "cancellation", "layer-0 like", "C tiny"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int p = 0; p < P; ++p) {
      for (int i = 0; i < 512; ++i) {
        float a = gauss(rng), b = gauss(rng);
        if (mode == 1) { a = std::ldexp(a, int(rng() % 17) - 8); b = std::ldexp(b, int(rng() % 17) - 8); }
        if (mode == 3) { a = std::ldexp(a, 12); b = std::ldexp(b, 12); }
        A[p * 512 + i] = static_cast<_Float16>(a);
        B[p * 512 + i] = static_cast<_Float16>(b);
        // (no fp16 denormals: |a|, |b| >= 2^-14 or zero)
        if (std::fabs(static_cast<float>(A[p * 512 + i])) < 6.2e-5f) A[p * 512 + i] = 0;
        if (std::fabs(static_cast<float>(B[p * 512 + i])) < 6.2e-5f) B[p * 512 + i] = 0;
        Af[p * 512 + i] = static_cast<float>(A[p * 512 + i]);
        Bf[p * 512 + i] = static_cast<float>(B[p * 512 + i]);
      }
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0;
          for (int k = 0; k < 16; ++k) s += static_cast<double>(Af[p * 512 + i * 16 + k]) * Bf[p * 512 + j * 16 + k];
          float c = 0;
          switch (mode) {
            case 0: c = 4.0f * gauss(rng); break;
            case 1: c = std::ldexp(gauss(rng), int(rng() % 25) - 12); break;
            case 2: c = -static_cast<float>(s) * (1.0f + 1e-3f * gauss(rng)); break;
            case 3: c = std::ldexp(gauss(rng), 27); break;
            case 4: c = std::ldexp(gauss(rng), -10); break;
          }
          C[p * 1024 + i * 32 + j] = c;
        }
    }
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(P), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    if (FILE *f = std::fopen("gpurun_out/mfma_samples.bin", mode == 0 ? "wb" : "ab")) {
      for (int p = 0; p < 32; ++p)
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            std::fwrite(&Af[p * 512 + i * 16], 4, 16, f);
            std::fwrite(&Bf[p * 512 + j * 16], 4, 16, f);
            std::fwrite(&C[p * 1024 + i * 32 + j], 4, 1, f);
            std::fwrite(&D[p * 1024 + i * 32 + j], 4, 1, f);
          }
      std::fclose(f);
    }
    std::vector<long> hits(hyps.size(), 0);
    long n = 0;
    for (int p = 0; p < P; ++p)
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          const float c = C[p * 1024 + i * 32 + j], d = D[p * 1024 + i * 32 + j];
          ++n;
          for (size_t hI = 0; hI < hyps.size(); ++hI) hits[hI] += eval(hyps[hI], c, &Af[p * 512 + i * 16], &Bf[p * 512 + j * 16]) == d;
        }
    total += n;
    std::printf("mode %d (%s), n = %ld: best hypotheses\n", mode, names[mode], n);
    std::vector<size_t> order(hyps.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return hits[x] > hits[y]; });
    for (int r = 0; r < 6; ++r) {
      const Hyp &h = hyps[order[r]];
      std::printf("   %.5f  groups %d  guard bits %d  exponent %s  cut %s  final %s\n", double(hits[order[r]]) / n, h.groups, h.g,
                  h.sum_exp ? "ea+eb" : "true", h.floor_cut ? "floor" : "trunc", h.rz ? "RZ" : "RNE");
    }
    for (size_t hI = 0; hI < hyps.size(); ++hI) total_hits[hI] += hits[hI];
  }
  std::printf("all modes: best hypotheses\n");
  std::vector<size_t> order(hyps.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return total_hits[x] > total_hits[y]; });
  for (int r = 0; r < 10; ++r) {
    const Hyp &h = hyps[order[r]];
    std::printf("   %.5f  groups %d  guard bits %d  exponent %s  cut %s  final %s\n", double(total_hits[order[r]]) / total, h.groups, h.g,
                h.sum_exp ? "ea+eb" : "true", h.floor_cut ? "floor" : "trunc", h.rz ? "RZ" : "RNE");
  }
  return 0;
}
