// fdnn_jni.cpp -- Java_suskun_nn_QuantizedDnn_* over the C-ABI (include/fdnn.h).
//
// Drop-in for the reference shim src/cpp/jni_dnn.cc: same symbols, same argument
// meaning, same ownership (handles are raw pointers in a jlong, freed only by
// delete()/deleteLazyContext()).  Differences, all on the safe side:
//   * inputs are treated as const (the reference shifts/scales the pinned Java
//     array in place and releases it with JNI_ABORT, dnn.cc:175-192);
//   * a failure throws java.lang.RuntimeException (IllegalArgumentException for
//     argument errors) with fdnn_last_error() instead of crashing / exit(3)
//     (float_dnn.cc:171, :185-188).
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/fdnn.h"
#include "../../include/fdnn_jni.h"

namespace {

template <typename Fn>
inline Fn slot(JNIEnv *env, int index) {
  return reinterpret_cast<Fn>(const_cast<void *>(env->functions[index]));
}

using FindClassFn = jclass (*)(JNIEnv *, const char *);
using ThrowNewFn = jint (*)(JNIEnv *, jclass, const char *);
using GetStringUTFCharsFn = const char *(*)(JNIEnv *, jstring, jboolean *);
using ReleaseStringUTFCharsFn = void (*)(JNIEnv *, jstring, const char *);
using GetArrayLengthFn = jsize (*)(JNIEnv *, jarray);
using NewFloatArrayFn = jfloatArray (*)(JNIEnv *, jsize);
using GetByteArrayElementsFn = jbyte *(*)(JNIEnv *, jbyteArray, jboolean *);
using GetFloatArrayElementsFn = jfloat *(*)(JNIEnv *, jfloatArray, jboolean *);
using ReleaseByteArrayElementsFn = void (*)(JNIEnv *, jbyteArray, jbyte *, jint);
using ReleaseFloatArrayElementsFn = void (*)(JNIEnv *, jfloatArray, jfloat *, jint);
using SetFloatArrayRegionFn = void (*)(JNIEnv *, jfloatArray, jsize, jsize, const jfloat *);

void throw_status(JNIEnv *env, int rc, const char *msg = nullptr) {
  const char *cls_name = rc == FDNN_E_ARG ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException";
  auto find = slot<FindClassFn>(env, FDNN_JNI_FindClass);
  auto thr = slot<ThrowNewFn>(env, FDNN_JNI_ThrowNew);
  if (!find || !thr) return;
  jclass cls = find(env, cls_name);
  if (cls) thr(env, cls, msg ? msg : fdnn_last_error());
}

// Native staging of a call's results: one block per calling thread, reused, so that its pages are resident.  A fresh
// zero-filled block per call, as the reference's shim has it (jni_dnn.cc:49-57), costs a page fault per 4 KB and the fill
// before the copy even starts: -15 % on a single thread scoring 100-frame utterances (tools/serve_bench.cpp, mode
// "fresh").  Deliberately NOT page-locked: into page-locked memory the rows leave as one DMA, which is 2 % faster for one
// caller and 35-40 % slower for 8-16 concurrent ones (the copies queue on the DMA engines instead of spreading over the
// callers' cores; measured, same tool).  Blocks above kScratchMax are not kept between calls.
struct Scratch {
  float *p = nullptr;
  size_t cap = 0;
  ~Scratch() { release(); }
  void release() {
    delete[] p;
    p = nullptr;
    cap = 0;
  }
  float *get(size_t floats) {
    if (floats > cap) {
      release();
      p = new (std::nothrow) float[floats];
      cap = p ? floats : 0;
    }
    return p;
  }
};
// Bytes a calling thread keeps between calls (a 100-frame utterance of an 8000-output net is 3.2 MB; 64 Java threads
// that each once scored a 256 MB batch would otherwise pin 16 GB for their lifetime).  FDNN_JNI_KEEP_MB overrides.
size_t scratch_keep_bytes() {
  static const size_t v = [] {
    const char *e = std::getenv("FDNN_JNI_KEEP_MB");
    const long mb = e ? std::atol(e) : 32;
    return static_cast<size_t>(mb < 0 ? 0 : mb) << 20;
  }();
  return v;
}
thread_local Scratch t_scratch;

jfloatArray to_java(JNIEnv *env, const float *data, size_t len) {
  jfloatArray result = slot<NewFloatArrayFn>(env, FDNN_JNI_NewFloatArray)(env, static_cast<jsize>(len));
  if (result) slot<SetFloatArrayRegionFn>(env, FDNN_JNI_SetFloatArrayRegion)(env, result, 0, static_cast<jsize>(len), data);
  return result;
}

}  // namespace

extern "C" {

jlong Java_suskun_nn_QuantizedDnn_initialize(JNIEnv *env, jobject, jstring path, jfloat cutoff) {
  const char *chars = slot<GetStringUTFCharsFn>(env, FDNN_JNI_GetStringUTFChars)(env, path, nullptr);
  fdnn_model *m = nullptr;
  int rc = fdnn_model_load(chars, cutoff, &m);
  slot<ReleaseStringUTFCharsFn>(env, FDNN_JNI_ReleaseStringUTFChars)(env, path, chars);
  if (rc) {
    throw_status(env, rc);
    return 0;
  }
  return reinterpret_cast<jlong>(m);
}

jint Java_suskun_nn_QuantizedDnn_inputDimension(JNIEnv *, jobject, jlong handle) {
  return fdnn_model_input_dim(reinterpret_cast<fdnn_model *>(handle));
}

jint Java_suskun_nn_QuantizedDnn_outputDimension(JNIEnv *, jobject, jlong handle) {
  return fdnn_model_output_dim(reinterpret_cast<fdnn_model *>(handle));
}

jfloatArray Java_suskun_nn_QuantizedDnn_calculate(JNIEnv *env, jobject, jlong handle, jfloatArray flat, jint n, jint dim,
                                                  jint batch) {
  fdnn_model *m = reinterpret_cast<fdnn_model *>(handle);
  const size_t len = static_cast<size_t>(n < 0 ? 0 : n) * static_cast<size_t>(fdnn_model_output_dim(m));
  if (n < 0 || len > static_cast<size_t>(INT32_MAX)) {  // a Java float[] holds at most 2^31 - 1 elements (jsize is a 32-bit int)
    throw_status(env, FDNN_E_ARG, "frames x output nodes does not fit a Java float[] (2^31 - 1 elements): score the batch in pieces");
    return nullptr;
  }
  jfloat *elements = slot<GetFloatArrayElementsFn>(env, FDNN_JNI_GetFloatArrayElements)(env, flat, nullptr);
  float *out = t_scratch.get(len);
  int rc = out ? fdnn_calculate(m, elements, n, dim, batch, out) : FDNN_E_NOMEM;
  slot<ReleaseFloatArrayElementsFn>(env, FDNN_JNI_ReleaseFloatArrayElements)(env, flat, elements, FDNN_JNI_ABORT);
  jfloatArray result = nullptr;
  if (rc)
    throw_status(env, rc, out ? nullptr : "out of host memory for the result block");
  else
    result = to_java(env, out, len);
  if (len * sizeof(float) > scratch_keep_bytes()) t_scratch.release();
  return result;
}

jlong Java_suskun_nn_QuantizedDnn_getContext(JNIEnv *env, jobject, jlong handle, jint n, jint batch) {
  fdnn_ctx *c = nullptr;
  int rc = fdnn_ctx_create(reinterpret_cast<fdnn_model *>(handle), n, batch, &c);
  if (rc) {
    throw_status(env, rc);
    return 0;
  }
  return reinterpret_cast<jlong>(c);
}

void Java_suskun_nn_QuantizedDnn_calculateUntilOutput(JNIEnv *env, jobject, jlong handle, jfloatArray input) {
  fdnn_ctx *c = reinterpret_cast<fdnn_ctx *>(handle);
  jfloat *elements = slot<GetFloatArrayElementsFn>(env, FDNN_JNI_GetFloatArrayElements)(env, input, nullptr);
  int rc = fdnn_ctx_forward_hidden(c, elements);
  slot<ReleaseFloatArrayElementsFn>(env, FDNN_JNI_ReleaseFloatArrayElements)(env, input, elements, FDNN_JNI_ABORT);
  if (rc) throw_status(env, rc);
}

jfloatArray Java_suskun_nn_QuantizedDnn_calculateLazy(JNIEnv *env, jobject, jlong handle, jint index, jbyteArray mask) {
  fdnn_ctx *c = reinterpret_cast<fdnn_ctx *>(handle);
  jbyte *bytes = slot<GetByteArrayElementsFn>(env, FDNN_JNI_GetByteArrayElements)(env, mask, nullptr);
  // the reference sizes the result by the mask length (jni_dnn.cc:111-113)
  const jsize len = slot<GetArrayLengthFn>(env, FDNN_JNI_GetArrayLength)(env, mask);
  std::vector<float> out(static_cast<size_t>(len));
  int rc = FDNN_E_ARG;
  if (len == fdnn_ctx_output_dim(c))
    rc = fdnn_ctx_lazy_output(c, index, bytes, out.data());
  else
    fdnn_ctx_lazy_output(c, -1, nullptr, nullptr);  // sets the argument-error text
  slot<ReleaseByteArrayElementsFn>(env, FDNN_JNI_ReleaseByteArrayElements)(env, mask, bytes, FDNN_JNI_ABORT);
  if (rc) {
    throw_status(env, rc);
    return nullptr;
  }
  return to_java(env, out.data(), out.size());
}

// EXTENSION (not in the reference's header): the lazy contract for a whole utterance in one native call --
//   private native float[] calculateLazyBatch(long handle, float[] flatInput, int n, int dim, byte[] flatMasks);
// flatMasks = n x outputDimension bytes, row-major, non-zero = active (the per-frame calculateLazy's mask, jni_dnn.cc:97-117,
// n times).  Returns n x outputDimension floats.  What LazyContext.calculateUntilOutput + n x calculateForOutputNodes do
// (QuantizedDnn.java:72-107) without the per-frame JNI round trips (README.md:45).  Uses the same nine JNI services.
jfloatArray Java_suskun_nn_QuantizedDnn_calculateLazyBatch(JNIEnv *env, jobject, jlong handle, jfloatArray flat, jint n, jint dim,
                                                           jbyteArray masks) {
  fdnn_model *m = reinterpret_cast<fdnn_model *>(handle);
  const size_t O = static_cast<size_t>(fdnn_model_output_dim(m));
  const size_t len = static_cast<size_t>(n < 0 ? 0 : n) * O;
  if (n < 0 || len > static_cast<size_t>(INT32_MAX)) {
    throw_status(env, FDNN_E_ARG, "frames x output nodes does not fit a Java float[] (2^31 - 1 elements): score the batch in pieces");
    return nullptr;
  }
  if (static_cast<size_t>(slot<GetArrayLengthFn>(env, FDNN_JNI_GetArrayLength)(env, masks)) != len) {
    throw_status(env, FDNN_E_ARG, "mask array length must equal frames x output nodes");
    return nullptr;
  }
  jfloat *elements = slot<GetFloatArrayElementsFn>(env, FDNN_JNI_GetFloatArrayElements)(env, flat, nullptr);
  jbyte *bytes = slot<GetByteArrayElementsFn>(env, FDNN_JNI_GetByteArrayElements)(env, masks, nullptr);
  float *out = t_scratch.get(len);
  int rc = out ? fdnn_calculate_lazy(m, elements, n, dim, bytes, out) : FDNN_E_NOMEM;
  slot<ReleaseByteArrayElementsFn>(env, FDNN_JNI_ReleaseByteArrayElements)(env, masks, bytes, FDNN_JNI_ABORT);
  slot<ReleaseFloatArrayElementsFn>(env, FDNN_JNI_ReleaseFloatArrayElements)(env, flat, elements, FDNN_JNI_ABORT);
  jfloatArray result = nullptr;
  if (rc)
    throw_status(env, rc, out ? nullptr : "out of host memory for the result block");
  else
    result = to_java(env, out, len);
  if (len * sizeof(float) > scratch_keep_bytes()) t_scratch.release();
  return result;
}

void Java_suskun_nn_QuantizedDnn_deleteLazyContext(JNIEnv *, jobject, jlong handle) {
  fdnn_ctx_free(reinterpret_cast<fdnn_ctx *>(handle));
}

void Java_suskun_nn_QuantizedDnn_delete(JNIEnv *, jobject, jlong handle) {
  fdnn_model_free(reinterpret_cast<fdnn_model *>(handle));
}

jint Java_suskun_nn_QuantizedDnn_layerDimension(JNIEnv *, jobject, jlong handle, jint index) {
  return fdnn_model_layer_dim(reinterpret_cast<fdnn_model *>(handle), index);
}

jint Java_suskun_nn_QuantizedDnn_layerCount(JNIEnv *, jobject, jlong handle) {
  return fdnn_model_layer_count(reinterpret_cast<fdnn_model *>(handle));
}

}  // extern "C"
