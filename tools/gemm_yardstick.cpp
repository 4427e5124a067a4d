// gemm_yardstick.cpp -- what the vendor libraries reach on the hidden / output layer GEMM shapes of the
// 432 -> 7 x 2048 -> 8000 net, on the same box and the same kind of data as the product's own kernels.
//
// A MEASURING TOOL ONLY: never linked into libfast-dnn.so, never on the product path.  It answers one
// question of the round-2 review: is 45 us (2048 nodes) / 177 us (8000 nodes) at 10 240 frames what a
// tuned library I8 x I8 -> I32 GEMM reaches on an MI355X, or is there head-room?
//
//   D[frame][node] = sum_k W[node][k] * A[frame][k]     (both operands K-contiguous = BLAS "TN")
//
// Plain GEMM: no dequantisation, no sigmoid table, no exp -- i.e. LESS work than qgemm_kernel does per launch.
//
// build:  hipcc -O2 --offload-arch=gfx950 tools/gemm_yardstick.cpp -o tools/gemm_yardstick -lhipblaslt -lrocblas
// run:    tools/gemm_yardstick [frames=10240]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    auto e_ = (x);                                                             \
    if (e_ != 0) {                                                             \
      std::fprintf(stderr, "%s failed: %d (line %d)\n", #x, int(e_), __LINE__); \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

static float time_loop(hipStream_t s, int iters, const std::function<void()> &f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) f();
  CK(hipStreamSynchronize(s));
  std::vector<float> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    t.push_back(ms * 1000.0f / iters);
  }
  std::sort(t.begin(), t.end());
  hipEventDestroy(a);
  hipEventDestroy(b);
  return t[t.size() / 2];  // median of five passes, us per call
}

int main(int argc, char **argv) {
  const int frames = argc > 1 ? std::atoi(argv[1]) : 10240;
  const int K = 2048;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipblasLtHandle_t lt;
  CK(hipblasLtCreate(&lt));
  rocblas_handle rb;
  CK(rocblas_create_handle(&rb));
  CK(rocblas_set_stream(rb, s));

  const size_t ws_bytes = size_t(256) << 20;
  void *ws = nullptr;
  CK(hipMalloc(&ws, ws_bytes));

  std::mt19937 rng(7);
  // the product's data: weights ~ N(0, 0.05) quantised with multiplier ~ 127/(4 sigma), activations = sigmoid bytes - 128
  std::normal_distribution<float> wn(0.0f, 32.0f);
  std::uniform_int_distribution<int> an(-128, 127);

  for (int nodes : {2048, 8000}) {
    const int M = nodes, N = frames;
    std::vector<int8_t> hw(size_t(M) * K), ha(size_t(N) * K);
    for (auto &v : hw) v = int8_t(std::max(-127.0f, std::min(127.0f, std::round(wn(rng)))));
    for (auto &v : ha) v = int8_t(an(rng));
    int8_t *dw, *da;
    int32_t *dd;
    CK(hipMalloc(&dw, hw.size()));
    CK(hipMalloc(&da, ha.size()));
    CK(hipMalloc(&dd, size_t(M) * N * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice));
    const double ops = 2.0 * M * double(N) * K;

    // ---- hipBLASLt: every algorithm the heuristic returns, best one reported
    {
      hipblasLtMatmulDesc_t desc;
      hipblasLtMatrixLayout_t la, lb, lc;
      CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32I, HIP_R_32I));
      hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
      CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_8I, K, M, K));   // A = W stored k x m column-major (rows of K bytes)
      CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_8I, K, N, K));   // B = activations k x n
      CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32I, M, N, M));  // D m x n column-major = [frame][node]
      hipblasLtMatmulPreference_t pref;
      CK(hipblasLtMatmulPreferenceCreate(&pref));
      CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
      hipblasLtMatmulHeuristicResult_t res[32];
      int got = 0;
      hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(lt, desc, la, lb, lc, lc, pref, 32, res, &got);
      if (st != HIPBLAS_STATUS_SUCCESS || got == 0) {
        std::printf("hipblaslt  nodes %5d frames %6d K %d : no algorithm (status %d)\n", M, N, K, int(st));
      } else {
        const int32_t alpha = 1, beta = 0;
        float best = 1e30f;
        int best_i = -1;
        for (int i = 0; i < got; ++i) {
          auto call = [&] {
            CK(hipblasLtMatmul(lt, desc, &alpha, dw, la, da, lb, &beta, dd, lc, dd, lc, &res[i].algo, ws, ws_bytes, s));
          };
          const float us = time_loop(s, 50, call);
          std::printf("hipblaslt  nodes %5d frames %6d K %d : algo %2d/%d  %8.2f us  %7.1f TOP/s\n", M, N, K, i, got, us,
                      ops / us * 1e-6);
          if (us < best) best = us, best_i = i;
        }
        std::printf("hipblaslt  nodes %5d frames %6d K %d : BEST algo %d  %8.2f us  %7.1f TOP/s  (%.3f of 5000)\n", M, N, K, best_i,
                    best, ops / best * 1e-6, ops / best * 1e-6 / 5000.0);
      }
    }
    // ---- rocBLAS gemm_ex I8 -> I32
    {
      const int32_t alpha = 1, beta = 0;
      auto call = [&] {
        CK(rocblas_gemm_ex(rb, rocblas_operation_transpose, rocblas_operation_none, M, N, K, &alpha, dw, rocblas_datatype_i8_r, K, da,
                           rocblas_datatype_i8_r, K, &beta, dd, rocblas_datatype_i32_r, M, dd, rocblas_datatype_i32_r, M,
                           rocblas_datatype_i32_r, rocblas_gemm_algo_standard, 0, 0));
      };
      const float us = time_loop(s, 50, call);
      std::printf("rocblas    nodes %5d frames %6d K %d :            %8.2f us  %7.1f TOP/s  (%.3f of 5000)\n", M, N, K, us,
                  ops / us * 1e-6, ops / us * 1e-6 / 5000.0);
    }
    // spot check of the last result against the host (transposition-detecting: random data, 64 entries)
    {
      std::vector<int32_t> hd(size_t(M) * N);
      CK(hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int t = 0; t < 64; ++t) {
        const int f = int(rng() % unsigned(N)), nd = int(rng() % unsigned(M));
        long ref = 0;
        for (int k = 0; k < K; ++k) ref += long(hw[size_t(nd) * K + k]) * long(ha[size_t(f) * K + k]);
        if (ref != hd[size_t(f) * M + nd]) ++bad;
      }
      std::printf("check      nodes %5d : %d of 64 sampled entries differ from the host sum\n", M, bad);
    }
    hipFree(dw);
    hipFree(da);
    hipFree(dd);
  }
  return 0;
}
