/*
 * fdnn_oracle.c -- CPU restatement of the fast-dnn quantized forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under fast-dnn_amd/ links, imports or calls it, and the
 * product library fails loudly without its HIP device code.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (int8
 * weights, multipliers, u8 activations, int32 accumulators) and to <= 1e-6
 * (logits, soft-max) against the reference itself, compiled from
 * /root/reference/src/cpp by oracle/Makefile into oracle/_ref/ (see
 * oracle/ref_tap.cpp), and against the committed fixtures in tests/golden/
 * that were generated from that build (tests/golden/make_golden.py).
 *
 * Canonical numerics = the reference built `-O2 -msse4 -ffp-contract=off`
 * (SURVEY.md 8(c)): layer-0 multiply and add are NOT fused.  Build this file
 * with -ffp-contract=off as well (oracle/Makefile does).  orc_set_l0_fma(1)
 * switches layer 0 to the fused flavour the reference shows when built with
 * its own Makefile's -march=native on an FMA host (vfmadd231ps in
 * InputActivations).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <smmintrin.h>
#include <tmmintrin.h>

#define ORC_LUT_SIZE 1280 /* dnn.h:26 SIGMOID_LOOKUP_SIZE */
#define ORC_LUT_HALF 640  /* dnn.h:27 */

typedef struct {
  int in_dim;  /* padded for layer 0 */
  int out_dim;
  float *w;    /* fp32 rows (layer 0 only, kept for all for convenience) */
  float *bias;
  int8_t *wq;  /* int8 rows (layers >= 1) */
  float mult;  /* per-layer multiplier (layers >= 1) */
} orc_layer;

typedef struct {
  int n_layers; /* affine layers, including fp32 layer 0 */
  int in_dim_file;
  orc_layer *layers;
  float *shift;
  float *scale;
  float cutoff;
} orc_model;

static uint8_t g_lut[ORC_LUT_SIZE];
static int g_lut_ready = 0;
static int g_l0_fma = 0;

void orc_set_l0_fma(int on) { g_l0_fma = on; }

/* QuantizedSigmoid::QuantizedSigmoid -- dnn.cc:100-115.  `exp(-k)` resolves to
 * the float overload (using namespace std), the sum and quotient are float,
 * round() is std::round(float). */
static void lut_build(void) {
  if (g_lut_ready) return;
  for (int i = -ORC_LUT_HALF; i < ORC_LUT_HALF; ++i) {
    float k = i / 100.0f;
    float sigmoid = 1.0f / (1 + expf(-k));
    g_lut[i + ORC_LUT_HALF] = (uint8_t)roundf(sigmoid * 255.0f);
  }
  g_lut_ready = 1;
}

void orc_lut(uint8_t *out) {
  lut_build();
  memcpy(out, g_lut, ORC_LUT_SIZE);
}

/* QuantizedSigmoid::get -- dnn.h:36-43.  In that header `round` is the global
 * C round(double); the float product converts exactly, so the integer is the
 * same as roundf's. */
static inline uint8_t lut_get(float x) {
  int k = (int)round((double)(x * 100));
  if (k <= -ORC_LUT_HALF) return 0;
  if (k >= ORC_LUT_HALF) return 255;
  return g_lut[k + ORC_LUT_HALF];
}

uint8_t orc_sigmoid_q(float x) {
  lut_build();
  return lut_get(x);
}

/* float -> char as x86 gcc compiles static_cast<char>(float): cvttss2si to a
 * 32-bit int (0x80000000 "integer indefinite" when NaN / out of range), low
 * byte kept.  dnn.cc:499. */
static inline int8_t float_to_char_x86(float v) {
  int32_t i;
  if (!(v > -2147483904.0f && v < 2147483648.0f)) /* NaN or out of int32 range */
    i = (int32_t)0x80000000u;
  else
    i = (int32_t)v;
  return (int8_t)(uint8_t)(i & 0xff);
}

/* absMax -- dnn.cc:148-160 (both-sided clamp only here). */
static float abs_max(const float *f, size_t n, float tmin, float tmax) {
  float max = -FLT_MAX;
  for (size_t i = 0; i < n; ++i) {
    float v = f[i];
    if (v < tmin) v = tmin;
    if (v > tmax) v = tmax;
    float a = (float)fabs(v);
    if (a > max) max = a;
  }
  return max;
}

/* QuantizedSimdLayer::QuantizedSimdLayer -- dnn.cc:460-509.  One multiplier
 * per layer; when quantizing, only the LOWER clamp is live (the upper one at
 * :496-498 tests minWeight > maxWeight and never fires). */
void orc_quantize(const float *w, int rows, int cols, float cutoff, int8_t *out, float *mult_out) {
  float maxW = cutoff, minW = -cutoff;
  float max = -FLT_MAX;
  for (int i = 0; i < rows; ++i) {
    float m = abs_max(w + (size_t)i * cols, (size_t)cols, minW, maxW);
    if (m > max) max = m;
  }
  float mult = roundf(127.0f / max); /* WEIGHT_MULTIPLIER dnn.cc:98 */
  for (int i = 0; i < rows; ++i) {
    for (int k = 0; k < cols; ++k) {
      float f = w[(size_t)i * cols + k];
      if (f < minW) f = minW;
      if (minW > maxW) f = maxW;
      out[(size_t)i * cols + k] = float_to_char_x86(roundf(f * mult));
    }
  }
  *mult_out = mult;
}

/* ------------------------------------------------------------------ .bin loader
 * FloatDnn::FloatDnn -- float_dnn.cc:18-69; BinaryLoader float_dnn.cc:166-212. */
static uint32_t be32(const unsigned char *p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static float bef32(const unsigned char *p) {
  uint32_t u = be32(p);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static int padded(int n, int d) { /* float_dnn.cc:77-83 */
  int dif = d - n % d;
  return dif == d ? n : n + dif;
}

void orc_model_free(orc_model *m);

orc_model *orc_model_load(const char *path, float cutoff) {
  lut_build();
  FILE *fp = fopen(path, "rb");
  if (!fp) return NULL;
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  rewind(fp);
  unsigned char *buf = (unsigned char *)malloc((size_t)sz);
  if (fread(buf, 1, (size_t)sz, fp) != (size_t)sz) {
    fclose(fp);
    free(buf);
    return NULL;
  }
  fclose(fp);
  size_t off = 0;
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  m->cutoff = cutoff;
  m->n_layers = (int)be32(buf + off);
  off += 4;
  m->layers = (orc_layer *)calloc((size_t)m->n_layers, sizeof(orc_layer));
  for (int j = 0; j < m->n_layers; ++j) {
    int in_dim = (int)be32(buf + off);
    off += 4;
    if (j == 0) m->in_dim_file = in_dim;
    int pin = j == 0 ? padded(in_dim, 4) : in_dim; /* float_dnn.cc:32-33 */
    int out_dim = (int)be32(buf + off);
    off += 4;
    orc_layer *L = &m->layers[j];
    L->in_dim = pin;
    L->out_dim = out_dim;
    L->w = (float *)calloc((size_t)pin * out_dim, sizeof(float));
    for (int o = 0; o < out_dim; ++o)
      for (int i = 0; i < in_dim; ++i, off += 4) L->w[(size_t)o * pin + i] = bef32(buf + off);
    L->bias = (float *)malloc(sizeof(float) * (size_t)out_dim);
    for (int o = 0; o < out_dim; ++o, off += 4) L->bias[o] = bef32(buf + off);
  }
  int pin0 = m->layers[0].in_dim;
  m->shift = (float *)calloc((size_t)pin0, sizeof(float)); /* float_dnn.cc:60-66 */
  m->scale = (float *)calloc((size_t)pin0, sizeof(float));
  for (int i = 0; i < m->in_dim_file; ++i, off += 4) m->shift[i] = bef32(buf + off);
  for (int i = 0; i < m->in_dim_file; ++i, off += 4) m->scale[i] = bef32(buf + off);
  free(buf);
  /* QuantizedDnn::QuantizedDnn -- dnn.cc:511-531: layers 1.. are quantized. */
  for (int j = 1; j < m->n_layers; ++j) {
    orc_layer *L = &m->layers[j];
    if (posix_memalign((void **)&L->wq, 16, (size_t)L->in_dim * L->out_dim)) {
      orc_model_free(m);
      return NULL;
    }
    orc_quantize(L->w, L->out_dim, L->in_dim, cutoff, L->wq, &L->mult);
    free(L->w);
    L->w = NULL;
  }
  return m;
}

void orc_model_free(orc_model *m) {
  if (!m) return;
  for (int j = 0; j < m->n_layers; ++j) {
    free(m->layers[j].w);
    free(m->layers[j].bias);
    free(m->layers[j].wq);
  }
  free(m->layers);
  free(m->shift);
  free(m->scale);
  free(m);
}

int orc_n_layers(const orc_model *m) { return m->n_layers; }
int orc_layer_in(const orc_model *m, int j) { return m->layers[j].in_dim; }
int orc_layer_out(const orc_model *m, int j) { return m->layers[j].out_dim; }
float orc_layer_mult(const orc_model *m, int j) { return m->layers[j].mult; }
const int8_t *orc_layer_wq(const orc_model *m, int j) { return m->layers[j].wq; }
const float *orc_layer_bias(const orc_model *m, int j) { return m->layers[j].bias; }
const float *orc_layer_w0(const orc_model *m) { return m->layers[0].w; }
const float *orc_shift(const orc_model *m) { return m->shift; }
const float *orc_scale(const orc_model *m) { return m->scale; }

/* ------------------------------------------------------------------ kernels */

/* quantizedNodeSum -- dnn.cc:323-349, scalar statement: every ADJACENT pair
 * a[2j]*w[2j] + a[2j+1]*w[2j+1] is saturated to int16 (pmaddubsw), then the
 * eight int16 are sign-extended and summed in int32 (no overflow at K<=2^15). */
static inline int32_t node_sum_scalar(int K, const uint8_t *a, const int8_t *w) {
  int32_t sum = 0;
  for (int k = 0; k < K; k += 2) {
    int32_t p = (int32_t)a[k] * w[k] + (int32_t)a[k + 1] * w[k + 1];
    if (p > 32767) p = 32767;
    if (p < -32768) p = -32768;
    sum += p;
  }
  return sum;
}

/* The same loop with the reference's instruction sequence (SSSE3 pmaddubsw +
 * SSE4.1 pmovsxwd), used for the CPU baseline timing. */
static inline int32_t node_sum_sse(int K, const uint8_t *a, const int8_t *w) {
  __m128i sum = _mm_setzero_si128();
  for (int j = 0; j < K; j += 16) {
    const __m128i in = _mm_load_si128((const __m128i *)(a + j));
    const __m128i wt = _mm_load_si128((const __m128i *)(w + j));
    const __m128i c = _mm_maddubs_epi16(in, wt);
    const __m128i lo = _mm_cvtepi16_epi32(c);
    const __m128i hi = _mm_cvtepi16_epi32(_mm_shuffle_epi32(c, 0x4e));
    sum = _mm_add_epi32(_mm_add_epi32(lo, hi), sum);
  }
  sum = _mm_hadd_epi32(sum, sum); /* horizontalSum dnn.cc:395-399 */
  sum = _mm_hadd_epi32(sum, sum);
  return _mm_extract_epi32(sum, 0);
}

/* exact (unsaturated) dot product, for counting saturation events */
static inline int32_t node_sum_exact(int K, const uint8_t *a, const int8_t *w) {
  int32_t sum = 0;
  for (int k = 0; k < K; ++k) sum += (int32_t)a[k] * w[k];
  return sum;
}

typedef struct {
  uint8_t *u8_acts;  /* [n_hidden][n][H] u8 activations after each hidden layer, or NULL */
  int32_t *acc_hid;  /* [n_hidden-1][n][H] int32 sums of the int8 hidden layers, or NULL */
  int32_t *acc_out;  /* [n][O] int32 sums of the output layer, or NULL */
  float *logits;     /* [n][O] z = sum/coef + bias, or NULL */
  float *l0_lin;     /* [n][H] layer-0 linear activation incl. bias, or NULL */
  long long sat_events; /* out: pairs whose int16 saturation fired */
} orc_taps;

/* InputActivations -- dnn.cc:219-247 (+ horizontalSum dnn.cc:168-172): four
 * lane partial sums over k mod 4, each a sequential chain `sum = sum + x*w`,
 * combined (l0+l1)+(l2+l3). */
static inline float l0_dot(int D, const float *x, const float *w) {
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (g_l0_fma) {
    for (int k = 0; k < D; k += 4) {
      s0 = fmaf(x[k], w[k], s0);
      s1 = fmaf(x[k + 1], w[k + 1], s1);
      s2 = fmaf(x[k + 2], w[k + 2], s2);
      s3 = fmaf(x[k + 3], w[k + 3], s3);
    }
  } else {
    for (int k = 0; k < D; k += 4) {
      s0 = s0 + x[k] * w[k];
      s1 = s1 + x[k + 1] * w[k + 1];
      s2 = s2 + x[k + 2] * w[k + 2];
      s3 = s3 + x[k + 3] * w[k + 3];
    }
  }
  return (s0 + s1) + (s2 + s3);
}

/* The same chains as the reference runs them (dnn.cc:232-240): one __m128 accumulator, mulps then addps
 * (this file is built -ffp-contract=off, so they stay two roundings), hadd, hadd = (l0+l1)+(l2+l3).
 * Bit-identical to l0_dot's unfused branch; the timing harness uses it so that the port's layer 0 costs what
 * the reference's does (the scalar loop made the port 15 % slower than the compiled reference on one thread). */
static inline float l0_dot_sse(int D, const float *x, const float *w) {
  __m128 sum = _mm_setzero_ps();
  for (int k = 0; k < D; k += 4) sum = _mm_add_ps(sum, _mm_mul_ps(_mm_loadu_ps(x + k), _mm_loadu_ps(w + k)));
  sum = _mm_hadd_ps(sum, sum);
  sum = _mm_hadd_ps(sum, sum);
  return _mm_cvtss_f32(sum);
}

/* SoftMax::apply -- dnn.cc:534-544: expf, sequential fp32 total, no max
 * subtraction. */
static void softmax_apply(float *z, int n, float *scratch) {
  float total = 0;
  for (int i = 0; i < n; ++i) {
    float d = expf(z[i]);
    scratch[i] = d;
    total += d;
  }
  for (int i = 0; i < n; ++i) z[i] = scratch[i] / total;
}

static int check_model(const orc_model *m) {
  if (m->n_layers < 4) return -1; /* dnn.cc:199 needs layers()[1] to be hidden */
  int H = m->layers[0].out_dim;
  for (int j = 1; j < m->n_layers - 1; ++j)
    if (m->layers[j].out_dim != H || m->layers[j].in_dim != H) return -2;
  if (m->layers[m->n_layers - 1].in_dim != H) return -2;
  if (H % 16) return -3;
  return 0;
}

/* CalculationContext::CalculateUntilLastHiddenLayer -- dnn.cc:402-424.
 * `x` is n x in_dim (padded) and is NOT modified (the reference mutates the
 * caller's buffer in ApplyShiftAndScale, dnn.cc:175-192; we work on a copy).
 * `act` receives the last hidden layer's n x H u8 activations.
 * The frame blocking of the reference (batch_size_) never changes any
 * per-element arithmetic, so `batch` only shapes the loop nest (node-outer,
 * frame-inner per block, dnn.cc:289-318) for the timing legs. */
int orc_hidden(const orc_model *m, const float *x_in, int n, int batch, int use_sse, uint8_t *act,
               orc_taps *taps) {
  int rc = check_model(m);
  if (rc) return rc;
  lut_build();
  if (batch < 1) batch = 1;
  const int D = m->layers[0].in_dim, H = m->layers[0].out_dim;
  float *x = (float *)malloc(sizeof(float) * (size_t)n * D);
  /* ApplyShiftAndScale dnn.cc:175-192: add, then multiply */
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < D; ++k) {
      float v = x_in[(size_t)i * D + k] + m->shift[k];
      x[(size_t)i * D + k] = v * m->scale[k];
    }
  float *lin = (float *)malloc(sizeof(float) * (size_t)batch * H);
  uint8_t *cur;
  if (posix_memalign((void **)&cur, 16, (size_t)n * H)) return -9;
  /* layer 0: InputActivations / AddBias / QuantizedSigmoid per frame block */
  const orc_layer *L0 = &m->layers[0];
  for (int b0 = 0; b0 < n; b0 += batch) {
    int nb = n - b0 < batch ? n - b0 : batch;
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < nb; ++j)
        lin[(size_t)j * H + i] = (use_sse && !g_l0_fma) ? l0_dot_sse(D, x + (size_t)(b0 + j) * D, L0->w + (size_t)i * D)
                                                       : l0_dot(D, x + (size_t)(b0 + j) * D, L0->w + (size_t)i * D);
    for (int j = 0; j < nb; ++j)
      for (int i = 0; i < H; ++i) {
        float v = lin[(size_t)j * H + i] + L0->bias[i]; /* AddBias dnn.cc:250-264 */
        if (taps && taps->l0_lin) taps->l0_lin[(size_t)(b0 + j) * H + i] = v;
        cur[(size_t)(b0 + j) * H + i] = lut_get(v); /* dnn.cc:267-286 */
      }
  }
  if (taps && taps->u8_acts) memcpy(taps->u8_acts, cur, (size_t)n * H);
  /* int8 hidden layers, in place on `cur` block by block exactly as the
   * reference overwrites quantized_activations_ (each block only reads its
   * own rows, so in-place is safe). */
  for (int j = 1; j < m->n_layers - 1; ++j) {
    const orc_layer *L = &m->layers[j];
    const float coef = L->mult * 255.0f; /* dnn.cc:298-299 */
    for (int b0 = 0; b0 < n; b0 += batch) {
      int nb = n - b0 < batch ? n - b0 : batch;
      for (int i = 0; i < H; ++i) {
        const int8_t *w = L->wq + (size_t)i * H;
        for (int k = 0; k < nb; ++k) {
          const uint8_t *a = cur + (size_t)(b0 + k) * H;
          int32_t s = use_sse ? node_sum_sse(H, a, w) : node_sum_scalar(H, a, w);
          if (taps) {
            if (taps->acc_hid) taps->acc_hid[((size_t)(j - 1) * n + b0 + k) * H + i] = s;
            if (s != node_sum_exact(H, a, w)) {
              for (int q = 0; q < H; q += 2) {
                int32_t p = (int32_t)a[q] * w[q] + (int32_t)a[q + 1] * w[q + 1];
                if (p > 32767 || p < -32768) taps->sat_events++;
              }
            }
          }
          lin[(size_t)k * H + i] = (float)s / coef;
        }
      }
      for (int k = 0; k < nb; ++k)
        for (int i = 0; i < H; ++i) cur[(size_t)(b0 + k) * H + i] = lut_get(lin[(size_t)k * H + i] + L->bias[i]);
    }
    if (taps && taps->u8_acts) memcpy(taps->u8_acts + (size_t)j * n * H, cur, (size_t)n * H);
  }
  memcpy(act, cur, (size_t)n * H);
  free(cur);
  free(lin);
  free(x);
  return 0;
}

/* CalculationContext::CalculateOutput -- dnn.cc:428-454 over precomputed last
 * hidden activations. */
int orc_output(const orc_model *m, const uint8_t *act_in, int n, int batch, int use_sse, float *out, orc_taps *taps) {
  int rc = check_model(m);
  if (rc) return rc;
  if (batch < 1) batch = 1;
  const orc_layer *L = &m->layers[m->n_layers - 1];
  const int H = L->in_dim, O = L->out_dim;
  const float coef = L->mult * 255.0f;
  uint8_t *act;
  if (posix_memalign((void **)&act, 16, (size_t)n * H)) return -9;
  memcpy(act, act_in, (size_t)n * H);
  for (int b0 = 0; b0 < n; b0 += batch) {
    int nb = n - b0 < batch ? n - b0 : batch;
    for (int i = 0; i < O; ++i) {
      const int8_t *w = L->wq + (size_t)i * H;
      for (int k = 0; k < nb; ++k) {
        const uint8_t *a = act + (size_t)(b0 + k) * H;
        int32_t s = use_sse ? node_sum_sse(H, a, w) : node_sum_scalar(H, a, w);
        if (taps && taps->acc_out) taps->acc_out[(size_t)(b0 + k) * O + i] = s;
        out[(size_t)(b0 + k) * O + i] = (float)s / coef;
      }
    }
  }
  float *scratch = (float *)malloc(sizeof(float) * (size_t)O);
  for (int f = 0; f < n; ++f) {
    float *o = out + (size_t)f * O;
    for (int j = 0; j < O; ++j) o[j] += L->bias[j];
    if (taps && taps->logits) memcpy(taps->logits + (size_t)f * O, o, sizeof(float) * (size_t)O);
    softmax_apply(o, O, scratch);
  }
  free(scratch);
  free(act);
  return 0;
}

/* CalculationContext::Calculate -- dnn.cc:162-165 */
int orc_calculate(const orc_model *m, const float *x, int n, int batch, int use_sse, float *out, orc_taps *taps) {
  const int H = m->layers[0].out_dim;
  uint8_t *act = (uint8_t *)malloc((size_t)n * H + 16);
  int rc = orc_hidden(m, x, n, batch, use_sse, act, taps);
  if (!rc) rc = orc_output(m, act, n, batch, use_sse, out, taps);
  free(act);
  return rc;
}

/* CalculationContext::LazyOutputActivations -- dnn.cc:355-392 for one frame:
 * masked-out nodes keep z = 0 and therefore contribute exp(0) = 1 to the
 * soft-max denominator and come back as 1/total. */
int orc_lazy_output(const orc_model *m, const uint8_t *act_frame, const int8_t *mask, int use_sse, float *out,
                    int32_t *acc_tap) {
  int rc = check_model(m);
  if (rc) return rc;
  const orc_layer *L = &m->layers[m->n_layers - 1];
  const int H = L->in_dim, O = L->out_dim;
  const float coef = L->mult * 255.0f;
  uint8_t *a;
  if (posix_memalign((void **)&a, 16, (size_t)H)) return -9;
  memcpy(a, act_frame, (size_t)H);
  for (int i = 0; i < O; ++i) {
    if (mask[i] == 0) {
      out[i] = 0;
      if (acc_tap) acc_tap[i] = 0;
      continue;
    }
    const int8_t *w = L->wq + (size_t)i * H;
    int32_t s = use_sse ? node_sum_sse(H, a, w) : node_sum_scalar(H, a, w);
    if (acc_tap) acc_tap[i] = s;
    out[i] = (float)s / coef + L->bias[i];
  }
  float *scratch = (float *)malloc(sizeof(float) * (size_t)O);
  softmax_apply(out, O, scratch);
  free(scratch);
  free(a);
  return 0;
}

/* Batched form of the lazy path: frame f uses masks[f*O .. ). */
int orc_lazy_batch(const orc_model *m, const uint8_t *act, int n, const int8_t *masks, int use_sse, float *out) {
  const orc_layer *L = &m->layers[m->n_layers - 1];
  for (int f = 0; f < n; ++f) {
    int rc = orc_lazy_output(m, act + (size_t)f * L->in_dim, masks + (size_t)f * L->out_dim, use_sse,
                             out + (size_t)f * L->out_dim, NULL);
    if (rc) return rc;
  }
  return 0;
}

/* Count (node, pair) entries of an int8 layer that CAN saturate for some
 * activation: 255*(w0^+ + w1^+) > 32767 or 255*(w0^- + w1^-) < -32768. */
long long orc_risky_pairs(const orc_model *m, int j) {
  const orc_layer *L = &m->layers[j];
  long long cnt = 0;
  for (size_t i = 0; i < (size_t)L->out_dim; ++i)
    for (int k = 0; k < L->in_dim; k += 2) {
      int w0 = L->wq[i * L->in_dim + k], w1 = L->wq[i * L->in_dim + k + 1];
      int pos = (w0 > 0 ? w0 : 0) + (w1 > 0 ? w1 : 0);
      int neg = (w0 < 0 ? w0 : 0) + (w1 < 0 ? w1 : 0);
      if (255 * pos > 32767 || 255 * neg < -32768) cnt++;
    }
  return cnt;
}

/* ---------------------------------------------------------------- multi-thread timing harness
 * bench.py's cpu_baseline leg: T native threads, each with its own output buffer, each scoring
 * `utts` independent utterances of n frames with orc_calculate (context construction + Calculate,
 * the reference CLI's timed region dnn.cc:64-71; concurrency model of
 * MultiThreadedStressTest.java:48-61: one shared immutable model, one private context per call).
 * No interpreter in the timed region.  Returns wall seconds from a common start barrier to the
 * last thread's finish, and per-thread seconds in per_thread[T] (may be NULL). */
#include <pthread.h>
#include <time.h>

typedef struct {
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int state; /* 0 wait, 1 go, 2 abort */
} orc_mt_start;

typedef struct {
  const orc_model *m;
  const float *x;
  int n, batch, sse, utts;
  orc_mt_start *start;
  double seconds;
  int rc;
} orc_mt_arg;

static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *orc_mt_worker(void *p) {
  orc_mt_arg *a = (orc_mt_arg *)p;
  const int O = a->m->layers[a->m->n_layers - 1].out_dim;
  float *out = (float *)malloc(sizeof(float) * (size_t)a->n * (size_t)O);
  pthread_mutex_lock(&a->start->mu);
  while (a->start->state == 0) pthread_cond_wait(&a->start->cv, &a->start->mu);
  const int go = a->start->state == 1;
  pthread_mutex_unlock(&a->start->mu);
  if (!go) {
    free(out);
    return NULL;
  }
  const double t0 = orc_now();
  a->rc = out ? 0 : -1;
  for (int u = 0; u < a->utts && !a->rc; ++u) a->rc = orc_calculate(a->m, a->x, a->n, a->batch, a->sse, out, NULL);
  a->seconds = orc_now() - t0;
  free(out);
  return NULL;
}

double orc_bench_threads(const orc_model *m, const float *x, int n, int batch, int use_sse, int threads, int utts,
                         double *per_thread) {
  if (threads < 1 || utts < 1) return -1.0;
  pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  orc_mt_arg *args = (orc_mt_arg *)calloc((size_t)threads, sizeof(orc_mt_arg));
  orc_mt_start start;
  pthread_mutex_init(&start.mu, NULL);
  pthread_cond_init(&start.cv, NULL);
  start.state = 0;
  int made = 0;
  for (int t = 0; t < threads; ++t) {
    args[t] = (orc_mt_arg){m, x, n, batch, use_sse, utts, &start, 0.0, 0};
    if (pthread_create(&th[t], NULL, orc_mt_worker, &args[t]) != 0) break;
    ++made;
  }
  /* every started thread is released (go, or abort when one could not be started) and joined:
   * nothing is cancelled and nothing is freed under a running thread */
  pthread_mutex_lock(&start.mu);
  start.state = made == threads ? 1 : 2;
  pthread_cond_broadcast(&start.cv);
  pthread_mutex_unlock(&start.mu);
  const double t0 = orc_now();
  int rc = made == threads ? 0 : -1;
  for (int t = 0; t < made; ++t) {
    pthread_join(th[t], NULL);
    rc |= args[t].rc;
    if (per_thread) per_thread[t] = args[t].seconds;
  }
  const double wall = orc_now() - t0;
  pthread_cond_destroy(&start.cv);
  pthread_mutex_destroy(&start.mu);
  free(th);
  free(args);
  return rc ? -1.0 : wall;
}
