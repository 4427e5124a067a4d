/*
 * fdnn_jni.h -- the JNI surface libfast-dnn.so exports for suskun.nn.QuantizedDnn.
 *
 * These eleven symbols are exactly the ones the reference's shim exports
 * (src/cpp/jni_dnn.cc, declared in src/cpp/suskun_nn_QuantizedDnn.h:15-96), so
 * the unmodified Java class (src/java/suskun/nn/QuantizedDnn.java:46,:109-127)
 * binds to this library as a drop-in.  Each one is a thin wrapper over the
 * C-ABI in fdnn.h.
 *
 * No JDK is needed to build: the shim uses nine slots of the standard
 * 235-entry JNINativeInterface_ function table (plus FindClass/ThrowNew to
 * surface errors as java.lang.RuntimeException instead of crashing), addressed
 * by their spec-fixed indices below.  The JNI primitive types are the Linux
 * x86-64 ones (jni_md.h of any JDK).
 */
#ifndef FDNN_JNI_H
#define FDNN_JNI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef float jfloat;
typedef jint jsize;
typedef uint8_t jboolean;
typedef void *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jfloatArray;
typedef jarray jbyteArray;

/* A JNIEnv is a pointer to a pointer to the function table. */
typedef const void *const *fdnn_jni_table;
typedef struct fdnn_JNIEnv_ {
  fdnn_jni_table functions;
} JNIEnv;

enum {
  FDNN_JNI_FindClass = 6,
  FDNN_JNI_ThrowNew = 14,
  FDNN_JNI_GetStringUTFChars = 169,
  FDNN_JNI_ReleaseStringUTFChars = 170,
  FDNN_JNI_GetArrayLength = 171,
  FDNN_JNI_NewFloatArray = 181,
  FDNN_JNI_GetByteArrayElements = 184,
  FDNN_JNI_GetFloatArrayElements = 189,
  FDNN_JNI_ReleaseByteArrayElements = 192,
  FDNN_JNI_ReleaseFloatArrayElements = 197,
  FDNN_JNI_SetFloatArrayRegion = 213,
  FDNN_JNI_TABLE_SIZE = 235
};
#define FDNN_JNI_ABORT 2

#define FDNN_JNIEXPORT __attribute__((visibility("default")))

/* native long initialize(String fileName, float weightCutOffValue)   jni_dnn.cc:7-18 */
FDNN_JNIEXPORT jlong Java_suskun_nn_QuantizedDnn_initialize(JNIEnv *, jobject, jstring, jfloat);
/* native int inputDimension(long)                                     jni_dnn.cc:20-25 */
FDNN_JNIEXPORT jint Java_suskun_nn_QuantizedDnn_inputDimension(JNIEnv *, jobject, jlong);
/* native int outputDimension(long)                                    jni_dnn.cc:27-33 */
FDNN_JNIEXPORT jint Java_suskun_nn_QuantizedDnn_outputDimension(JNIEnv *, jobject, jlong);
/* native float[] calculate(long, float[], int, int, int)              jni_dnn.cc:35-62 */
FDNN_JNIEXPORT jfloatArray Java_suskun_nn_QuantizedDnn_calculate(JNIEnv *, jobject, jlong, jfloatArray, jint, jint, jint);
/* native long getContext(long, int, int)                              jni_dnn.cc:64-77 */
FDNN_JNIEXPORT jlong Java_suskun_nn_QuantizedDnn_getContext(JNIEnv *, jobject, jlong, jint, jint);
/* native void calculateUntilOutput(long, float[])                     jni_dnn.cc:79-95 */
FDNN_JNIEXPORT void Java_suskun_nn_QuantizedDnn_calculateUntilOutput(JNIEnv *, jobject, jlong, jfloatArray);
/* native float[] calculateLazy(long, int, byte[])                     jni_dnn.cc:97-117 */
FDNN_JNIEXPORT jfloatArray Java_suskun_nn_QuantizedDnn_calculateLazy(JNIEnv *, jobject, jlong, jint, jbyteArray);
/* native void deleteLazyContext(long)                                 jni_dnn.cc:119-126 */
/* EXTENSION, not in suskun_nn_QuantizedDnn.h: the lazy contract for n frames in one call --
 * private native float[] calculateLazyBatch(long handle, float[] flatInput, int n, int dim, byte[] flatMasks); */
FDNN_JNIEXPORT jfloatArray Java_suskun_nn_QuantizedDnn_calculateLazyBatch(JNIEnv *, jobject, jlong, jfloatArray, jint, jint, jbyteArray);
FDNN_JNIEXPORT void Java_suskun_nn_QuantizedDnn_deleteLazyContext(JNIEnv *, jobject, jlong);
/* native void delete(long)                                            jni_dnn.cc:128-133 */
FDNN_JNIEXPORT void Java_suskun_nn_QuantizedDnn_delete(JNIEnv *, jobject, jlong);
/* native int layerDimension(long, int)                                jni_dnn.cc:135-148 */
FDNN_JNIEXPORT jint Java_suskun_nn_QuantizedDnn_layerDimension(JNIEnv *, jobject, jlong, jint);
/* native int layerCount(long)                                         jni_dnn.cc:150-156 */
FDNN_JNIEXPORT jint Java_suskun_nn_QuantizedDnn_layerCount(JNIEnv *, jobject, jlong);

#ifdef __cplusplus
}
#endif
#endif /* FDNN_JNI_H */
