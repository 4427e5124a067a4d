// ref_tap.cpp -- tap harness around the UNMODIFIED reference sources.
//
// TEST INFRASTRUCTURE ONLY (see oracle/fdnn_oracle.c header).  This file holds
// no reference code: it #includes /root/reference/src/cpp/dnn.cc where it lies
// (oracle/Makefile passes -I/root/reference/src/cpp) so that the reference's
// own functions run, and exports C entry points that expose their results and
// intermediate state.  It is only buildable where /root/reference exists; the
// output goes to oracle/_ref/ (git-ignored, travels to the GPU box as a .so).
//
// Two tricks, both compile-time only: `main` in dnn.cc is renamed so the CLI
// entry does not clash, and `private` is made public for the reference's own
// headers so the harness can read CalculationContext's scratch buffers.
#include <pthread.h>
#include <time.h>
#include <x86intrin.h>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#define private public
#define main fastdnn_reference_cli_main
#include "dnn.cc"
#undef main
#undef private

extern "C" {

void *ref_model_load(const char *path, float cutoff) {
  const dnn::FloatDnn floatDnn{std::string(path)};
  return new dnn::QuantizedDnn(floatDnn, cutoff);
}
void ref_model_free(void *h) { delete reinterpret_cast<dnn::QuantizedDnn *>(h); }

int ref_input_dim(void *h) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->input_dimension(); }
int ref_output_dim(void *h) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->output_dimension(); }
int ref_q_layer_count(void *h) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->layer_count(); }
int ref_hidden_dim(void *h) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->input_layer()->node_count(); }
const float *ref_l0_weights(void *h) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->input_layer()->weights(); }
const float *ref_l0_bias(void *h) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->input_layer()->bias(); }
float ref_q_mult(void *h, int j) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->layers()[j]->multiplier(); }
const char *ref_q_weights(void *h, int j) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->layers()[j]->weights(); }
const float *ref_q_bias(void *h, int j) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->layers()[j]->bias(); }
int ref_q_out(void *h, int j) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->layers()[j]->node_count(); }
int ref_q_in(void *h, int j) { return (int)reinterpret_cast<dnn::QuantizedDnn *>(h)->layers()[j]->input_dimension(); }
const float *ref_shift(void *h) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->shift_; }
const float *ref_scale(void *h) { return reinterpret_cast<dnn::QuantizedDnn *>(h)->scale_; }

void ref_lut(unsigned char *out) { std::memcpy(out, dnn::qSigmoid->lookup_, dnn::SIGMOID_LOOKUP_SIZE); }
unsigned char ref_sigmoid_get(float x) { return dnn::qSigmoid->get(x); }

// Quantize one free-standing layer through the reference constructor.
void ref_quantize(const float *w, int rows, int cols, float cutoff, char *out, float *mult) {
  float **rowsp = new float *[rows];
  for (int i = 0; i < rows; ++i) {
    rowsp[i] = new float[cols];
    std::memcpy(rowsp[i], w + (size_t)i * cols, sizeof(float) * cols);
  }
  float *bias = new float[rows]();
  {
    dnn::FloatLayer fl(rowsp, bias, (size_t)cols, (size_t)rows);  // owns rows + bias
    dnn::QuantizedSimdLayer q(fl, cutoff);
    std::memcpy(out, q.weights(), (size_t)rows * cols);
    *mult = q.multiplier();
  }
  delete[] rowsp;
}

// The public entry: CalculationContext::Calculate on a private copy of x (the
// reference shifts/scales its input in place).
void ref_calculate(void *h, const float *x, int n, int dim, int batch, float *out) {
  dnn::QuantizedDnn *q = reinterpret_cast<dnn::QuantizedDnn *>(h);
  float *copy = dnn::AlignedAlloc<float>((size_t)n * dim);
  std::memcpy(copy, x, sizeof(float) * (size_t)n * dim);
  dnn::BatchData in(copy, (size_t)n, (size_t)dim, true);
  dnn::CalculationContext ctx(q, (size_t)n, (size_t)batch);
  dnn::BatchData *res = ctx.Calculate(in);
  std::memcpy(out, res->data(), sizeof(float) * (size_t)n * q->output_dimension());
  delete res;
}

// CalculateUntilLastHiddenLayer, returning the last hidden layer's u8 rows.
void ref_hidden(void *h, const float *x, int n, int dim, int batch, unsigned char *act) {
  dnn::QuantizedDnn *q = reinterpret_cast<dnn::QuantizedDnn *>(h);
  float *copy = dnn::AlignedAlloc<float>((size_t)n * dim);
  std::memcpy(copy, x, sizeof(float) * (size_t)n * dim);
  dnn::BatchData in(copy, (size_t)n, (size_t)dim, true);
  dnn::CalculationContext ctx(q, (size_t)n, (size_t)batch);
  ctx.CalculateUntilLastHiddenLayer(in);
  std::memcpy(act, ctx.quantized_activations_, (size_t)n * ctx.hidden_node_count_);
}

// Lazy path exactly as QuantizedDnn.LazyContext drives it: hidden layers once,
// then LazyOutputActivations(frame, mask) per frame.
void ref_lazy(void *h, const float *x, int n, int dim, int batch, const char *masks, float *out) {
  dnn::QuantizedDnn *q = reinterpret_cast<dnn::QuantizedDnn *>(h);
  const size_t O = q->output_dimension();
  float *copy = dnn::AlignedAlloc<float>((size_t)n * dim);
  std::memcpy(copy, x, sizeof(float) * (size_t)n * dim);
  dnn::BatchData in(copy, (size_t)n, (size_t)dim, true);
  dnn::CalculationContext ctx(q, (size_t)n, (size_t)batch);
  ctx.CalculateUntilLastHiddenLayer(in);
  for (int f = 0; f < n; ++f) {
    float *res = ctx.LazyOutputActivations((size_t)f, masks + (size_t)f * O);
    std::memcpy(out + (size_t)f * O, res, sizeof(float) * O);
  }
}

// Layer-by-layer drive through the reference's PUBLIC per-stage methods, in the
// order of CalculateUntilLastHiddenLayer / CalculateOutput, snapshotting state
// between stages.  Any output pointer may be null.
//   l0_lin   [n][H]            layer-0 activation after AddBias
//   u8_acts  [n_hidden][n][H]  u8 activations after every hidden layer
//   acc_hid  [n_hidden-1][n][H] quantizedNodeSum value (float(int32)) per int8 hidden layer
//   acc_out  [n][O]            same for the output layer
//   logits   [n][O]            output-layer value after bias, before soft-max
//   probs    [n][O]
void ref_forward_taps(void *h, const float *x, int n, int dim, int batch, float *l0_lin, unsigned char *u8_acts,
                      float *acc_hid, float *acc_out, float *logits, float *probs) {
  dnn::QuantizedDnn *q = reinterpret_cast<dnn::QuantizedDnn *>(h);
  const size_t O = q->output_dimension();
  float *copy = dnn::AlignedAlloc<float>((size_t)n * dim);
  std::memcpy(copy, x, sizeof(float) * (size_t)n * dim);
  dnn::BatchData in(copy, (size_t)n, (size_t)dim, true);
  dnn::CalculationContext ctx(q, (size_t)n, (size_t)batch);
  const size_t H = ctx.hidden_node_count_;
  const size_t B = ctx.batch_size_;
  q->ApplyShiftAndScale(in);
  for (size_t i = 0; i < (size_t)n; i += B) {
    ctx.InputActivations(in, i);
    ctx.AddBias(q->input_layer()->bias());
    if (l0_lin)
      for (size_t k = 0; k < B && i + k < (size_t)n; ++k)
        std::memcpy(l0_lin + (i + k) * H, ctx.activations_ + k * H, sizeof(float) * H);
    ctx.QuantizedSigmoid(i);
  }
  if (u8_acts) std::memcpy(u8_acts, ctx.quantized_activations_, (size_t)n * H);
  for (size_t j = 0; j < q->layer_count() - 1; ++j) {
    const dnn::QuantizedSimdLayer &layer = *q->layers()[j];
    if (acc_hid)
      for (size_t f = 0; f < (size_t)n; ++f)
        for (size_t i = 0; i < H; ++i)
          acc_hid[(j * n + f) * H + i] =
              dnn::quantizedNodeSum(H, &ctx.quantized_activations_[f * H], &layer.weights()[i * H]);
    for (size_t i = 0; i < (size_t)n; i += B) {
      ctx.QuantizedLayerActivations(layer, i, ctx.activations_);
      ctx.AddBias(layer.bias());
      ctx.QuantizedSigmoid(i);
    }
    if (u8_acts) std::memcpy(u8_acts + (j + 1) * n * H, ctx.quantized_activations_, (size_t)n * H);
  }
  dnn::QuantizedSimdLayer &ol = *q->output_layer();
  if (acc_out)
    for (size_t f = 0; f < (size_t)n; ++f)
      for (size_t i = 0; i < O; ++i)
        acc_out[f * O + i] = dnn::quantizedNodeSum(H, &ctx.quantized_activations_[f * H], &ol.weights()[i * H]);
  if (logits) {
    for (size_t i = 0; i < (size_t)n; i += B) ctx.QuantizedLayerActivations(ol, i, &logits[i * O]);
    for (size_t f = 0; f < (size_t)n; ++f)
      for (size_t i = 0; i < O; ++i) logits[f * O + i] += ol.bias()[i];
  }
  if (probs) {
    dnn::BatchData *res = ctx.CalculateOutput();
    std::memcpy(probs, res->data(), sizeof(float) * (size_t)n * O);
    delete res;
  }
}

// Multi-thread timing harness over the reference itself (bench.py's cpu_baseline, kind "reference"): T native
// threads share one immutable QuantizedDnn, each scores `utts` independent utterances of n frames with a fresh
// CalculationContext per call -- exactly what the JNI shim does per calculate() (jni_dnn.cc:49-51), under the
// concurrency model of MultiThreadedStressTest.java:48-61.  Timed region per call = context construction +
// Calculate (the reference CLI's, dnn.cc:64-71) plus the private input copy the in-place ApplyShiftAndScale needs.
// Returns wall seconds from the common start to the last finish (per-thread seconds in per_thread[T], may be
// null), or -1 when a thread could not be started.
namespace {
struct RefMtShared {
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int state;  // 0 = wait, 1 = go, 2 = abort (a thread failed to start: nobody computes)
};
struct RefMtArg {
  dnn::QuantizedDnn *q;
  const float *x;
  int n, dim, batch, utts;
  RefMtShared *sh;
  double seconds;
};
double ref_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}
void *ref_mt_worker(void *p) {
  RefMtArg *a = reinterpret_cast<RefMtArg *>(p);
  pthread_mutex_lock(&a->sh->mu);
  while (a->sh->state == 0) pthread_cond_wait(&a->sh->cv, &a->sh->mu);
  const int state = a->sh->state;
  pthread_mutex_unlock(&a->sh->mu);
  if (state != 1) return nullptr;
  const double t0 = ref_now();
  for (int u = 0; u < a->utts; ++u) {
    float *copy = dnn::AlignedAlloc<float>((size_t)a->n * a->dim);
    std::memcpy(copy, a->x, sizeof(float) * (size_t)a->n * a->dim);
    dnn::BatchData in(copy, (size_t)a->n, (size_t)a->dim, true);
    dnn::CalculationContext ctx(a->q, (size_t)a->n, (size_t)a->batch);
    dnn::BatchData *res = ctx.Calculate(in);
    delete res;
  }
  a->seconds = ref_now() - t0;
  return nullptr;
}
}  // namespace

double ref_bench_threads(void *h, const float *x, int n, int dim, int batch, int threads, int utts, double *per_thread) {
  if (threads < 1 || utts < 1) return -1.0;
  RefMtShared sh;
  pthread_mutex_init(&sh.mu, nullptr);
  pthread_cond_init(&sh.cv, nullptr);
  sh.state = 0;
  std::vector<pthread_t> th((size_t)threads);
  std::vector<RefMtArg> args((size_t)threads);
  int made = 0;
  for (int t = 0; t < threads; ++t) {
    args[(size_t)t] = RefMtArg{reinterpret_cast<dnn::QuantizedDnn *>(h), x, n, dim, batch, utts, &sh, 0.0};
    if (pthread_create(&th[(size_t)t], nullptr, ref_mt_worker, &args[(size_t)t]) != 0) break;
    ++made;
  }
  pthread_mutex_lock(&sh.mu);
  sh.state = made == threads ? 1 : 2;  // every started thread is released either way and joined below
  pthread_cond_broadcast(&sh.cv);
  pthread_mutex_unlock(&sh.mu);
  const double t0 = ref_now();
  for (int t = 0; t < made; ++t) pthread_join(th[(size_t)t], nullptr);
  const double wall = ref_now() - t0;
  pthread_cond_destroy(&sh.cv);
  pthread_mutex_destroy(&sh.mu);
  if (made != threads) return -1.0;
  if (per_thread)
    for (int t = 0; t < threads; ++t) per_thread[t] = args[(size_t)t].seconds;
  return wall;
}

// Reference CLI (dnn.cc:20-84) callable in-process.
int ref_cli(int argc, char **argv) { return fastdnn_reference_cli_main(argc, argv); }

}  // extern "C"
