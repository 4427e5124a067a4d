// serve_bench -- the serving shape, natively: T host threads, each scoring U utterances of F frames
// (100 frames = 1 s of speech) through the C-ABI, buffers reused, no interpreter in the loop.
//   serve_bench model.bin threads utts_per_thread frames mode [max_frames depth linger_us]
//   mode: percall  = fdnn_calculate per utterance (what the JNI calculate() does)
//         server   = fdnn_server_submit + fdnn_server_wait (coalesced host submissions)
//         batcher  = fdnn_calculate on a model with fdnn_model_enable_batcher
//         fresh    = percall into a newly allocated, zero-filled result block per call (the reference shim's shape)
//         lazy       = fdnn_calculate_lazy_bits per utterance (one-call lazy scoring, 40 % of the nodes active per frame)
//         lazyserver = fdnn_server_submit_lazy_bits + fdnn_server_wait (bit-mask utterances coalesced, rows back compacted)
// INFLIGHT=k (environment, server modes): every caller keeps k utterances in flight (k result blocks per thread) instead of
// waiting for each one before submitting the next.
// Prints one JSON line.  Build: see tools/serve_bench.sh.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../include/fdnn.h"

int main(int argc, char **argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: serve_bench model threads utts frames percall|server|batcher [max_frames depth linger_us]\n");
    return 2;
  }
  const char *path = argv[1];
  const int T = std::atoi(argv[2]), U = std::atoi(argv[3]), F = std::atoi(argv[4]);
  const std::string mode = argv[5];
  const int max_frames = argc > 6 ? std::atoi(argv[6]) : 6400, depth = argc > 7 ? std::atoi(argv[7]) : 3,
            linger = argc > 8 ? std::atoi(argv[8]) : 100;
  fdnn_model *m = nullptr;
  if (fdnn_model_load(path, 3.0f, &m)) {
    std::fprintf(stderr, "load: %s\n", fdnn_last_error());
    return 1;
  }
  const int D = fdnn_model_input_dim(m), O = fdnn_model_output_dim(m);
  fdnn_server *srv = nullptr;
  const bool lazy = mode == "lazy" || mode == "lazyserver";
  if ((mode == "server" || mode == "lazyserver") && fdnn_server_create(m, max_frames, depth, &srv)) {
    std::fprintf(stderr, "server: %s\n", fdnn_last_error());
    return 1;
  }
  if (srv) fdnn_server_set_linger_us(srv, linger);
  if (mode == "batcher" && fdnn_model_enable_batcher(m, max_frames, depth, linger)) {
    std::fprintf(stderr, "batcher: %s\n", fdnn_last_error());
    return 1;
  }
  const int inflight = std::max(1, std::getenv("INFLIGHT") ? std::atoi(std::getenv("INFLIGHT")) : 1);
  std::vector<std::vector<float>> xs(static_cast<size_t>(T)), outs(static_cast<size_t>(T));
  std::vector<std::vector<std::vector<float>>> more(static_cast<size_t>(T));  // result blocks 1 .. inflight - 1 of a caller
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.0f, 15.0f);
  for (int t = 0; t < T; ++t) {
    xs[size_t(t)].resize(size_t(F) * D);
    for (float &v : xs[size_t(t)]) v = nd(rng);
    outs[size_t(t)].assign(size_t(F) * O, 0.0f);  // resident, like a reused JVM float[]
    if (srv) more[size_t(t)].assign(size_t(inflight - 1), std::vector<float>(size_t(F) * O, 0.0f));
  }
  // lazy modes: per thread, masks with 40 % of the nodes active in every frame and 3 % churn from frame to frame
  // (FuncTest.java:121-154's statistics), as bits
  const size_t wpr = (size_t(O) + 63) / 64;
  std::vector<std::vector<uint64_t>> bits(static_cast<size_t>(T));
  if (lazy) {
    for (int t = 0; t < T; ++t) {
      std::vector<char> on(size_t(O), 0);
      std::vector<int> perm(static_cast<size_t>(O));
      for (int i = 0; i < O; ++i) perm[size_t(i)] = i;
      std::shuffle(perm.begin(), perm.end(), rng);
      const int active = int(O * 0.40), churn = int(O * 0.03);
      for (int i = 0; i < active; ++i) on[size_t(perm[size_t(i)])] = 1;
      bits[size_t(t)].assign(size_t(F) * wpr, 0);
      std::uniform_int_distribution<int> pick(0, O - 1);
      for (int f = 0; f < F; ++f) {
        if (f) {
          for (int c = 0, done = 0; done < churn && c < 100 * churn; ++c) { const int i = pick(rng); if (!on[size_t(i)]) { on[size_t(i)] = 1; ++done; } }
          for (int c = 0, done = 0; done < churn && c < 100 * churn; ++c) { const int i = pick(rng); if (on[size_t(i)]) { on[size_t(i)] = 0; ++done; } }
        }
        for (int i = 0; i < O; ++i)
          if (on[size_t(i)]) bits[size_t(t)][size_t(f) * wpr + size_t(i >> 6)] |= uint64_t(1) << (i & 63);
      }
    }
  }
  std::atomic<int> failed{0};
  static thread_local volatile float sink = 0.0f;
  auto body = [&](int t, int utts) {
    if (srv && inflight > 1) {  // a window of `inflight` tickets per caller
      std::vector<uint64_t> tk(size_t(inflight), 0);
      for (int u = 0; u < utts + inflight; ++u) {
        const size_t slot = size_t(u % inflight);
        if (u >= inflight && fdnn_server_wait(srv, tk[slot])) ++failed;
        if (u >= utts) continue;
        float *o = slot == 0 ? outs[size_t(t)].data() : more[size_t(t)][slot - 1].data();
        const int rc = lazy ? fdnn_server_submit_lazy_bits(srv, xs[size_t(t)].data(), F, bits[size_t(t)].data(), o, &tk[slot])
                            : fdnn_server_submit(srv, xs[size_t(t)].data(), F, nullptr, o, &tk[slot]);
        if (rc) ++failed;
      }
      return;
    }
    for (int u = 0; u < utts; ++u) {
      int rc;
      if (srv && lazy) {
        uint64_t ticket = 0;
        rc = fdnn_server_submit_lazy_bits(srv, xs[size_t(t)].data(), F, bits[size_t(t)].data(), outs[size_t(t)].data(), &ticket);
        if (!rc) rc = fdnn_server_wait(srv, ticket);
      } else if (lazy) {
        rc = fdnn_calculate_lazy_bits(m, xs[size_t(t)].data(), F, D, bits[size_t(t)].data(), outs[size_t(t)].data());
      } else if (srv) {
        uint64_t ticket = 0;
        rc = fdnn_server_submit(srv, xs[size_t(t)].data(), F, nullptr, outs[size_t(t)].data(), &ticket);
        if (!rc) rc = fdnn_server_wait(srv, ticket);
      } else if (mode == "fresh") {  // the reference shim's shape: a new zero-filled result block per call (jni_dnn.cc:49-57)
        std::vector<float> fresh(size_t(F) * O);
        rc = fdnn_calculate(m, xs[size_t(t)].data(), F, D, 10, fresh.data());
        sink = sink + fresh[size_t(u) % fresh.size()];
      } else {
        rc = fdnn_calculate(m, xs[size_t(t)].data(), F, D, 10, outs[size_t(t)].data());
      }
      if (rc) ++failed;
    }
  };
  auto run = [&](int utts) {
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; ++t) th.emplace_back(body, t, utts);
    for (auto &x : th) x.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  run(std::max(2, U / 10));  // warm-up: contexts, pinned staging, clocks
  const double sec = run(U);
  double sum = 0.0;
  for (int i = 0; i < O; ++i) sum += outs[0][size_t(i)];
  uint64_t batches = 0, frames = 0, reqs = 0, coal = 0;
  if (srv) fdnn_server_stats(srv, &batches, &frames, &reqs, &coal);
  std::printf("{\"mode\": \"%s\", \"threads\": %d, \"frames_per_utt\": %d, \"utts\": %d, \"seconds\": %.4f, \"utts_per_s\": %.1f, "
              "\"frames_per_s\": %.1f, \"row0_sum\": %.6f, \"failed\": %d, \"batches\": %llu, \"requests\": %llu}\n",
              mode.c_str(), T, F, T * U, sec, T * U / sec, double(T) * U * F / sec, sum, failed.load(), (unsigned long long)batches,
              (unsigned long long)reqs);
  if (srv) fdnn_server_free(srv);
  fdnn_model_free(m);
  return failed.load() ? 1 : 0;
}
