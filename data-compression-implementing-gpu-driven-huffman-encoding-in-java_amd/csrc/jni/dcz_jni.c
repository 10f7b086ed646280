/*
 * dcz_jni.c -- JNI shim between com.datacomp.service.hip.HipNative and the C ABI of include/dcz.h.
 * Built only where a JDK is present (needs jni.h):
 *   gcc -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../../include \
 *       -o libdczjni.so dcz_jni.c -L.. -ldczhip -Wl,-rpath,'$ORIGIN'
 * NOT COMPILED in the authoring image (no JDK / jni.h there).  Java owns every byte[]; nothing native
 * outlives a call (the reference's ownership rule).
 */
#include <jni.h>
#include <stdint.h>

#include "dcz.h"

#define CTX(h) ((dcz_ctx*)(intptr_t)(h))

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_deviceCount(JNIEnv* env, jclass cls) {
    (void)env; (void)cls;
    return dcz_device_count();
}

JNIEXPORT jlong JNICALL Java_com_datacomp_service_hip_HipNative_ctxCreate(JNIEnv* env, jclass cls, jint device) {
    (void)env; (void)cls;
    dcz_ctx* c = NULL;
    return dcz_ctx_create(device, &c) == DCZ_OK ? (jlong)(intptr_t)c : 0;
}

JNIEXPORT void JNICALL Java_com_datacomp_service_hip_HipNative_ctxDestroy(JNIEnv* env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    dcz_ctx_destroy(CTX(ctx));
}

JNIEXPORT jstring JNICALL Java_com_datacomp_service_hip_HipNative_strerror(JNIEnv* env, jclass cls, jint st) {
    (void)cls;
    return (*env)->NewStringUTF(env, dcz_strerror(st));
}

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_histogram(JNIEnv* env, jclass cls, jlong ctx,
                                                                        jbyteArray data, jint offset, jint length,
                                                                        jlongArray hist) {
    (void)cls;
    int64_t h[256];
    jbyte* p = (*env)->GetPrimitiveArrayCritical(env, data, NULL);
    if (!p) return DCZ_E_INVALID;
    int st = dcz_histogram(CTX(ctx), (const uint8_t*)p, (size_t)offset, (size_t)length, h);
    (*env)->ReleasePrimitiveArrayCritical(env, data, p, JNI_ABORT);
    if (st == DCZ_OK) (*env)->SetLongArrayRegion(env, hist, 0, 256, (const jlong*)h);
    return st;
}

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_encodeBlock(JNIEnv* env, jclass cls, jlong ctx,
                                                                          jbyteArray data, jint length,
                                                                          jintArray lengths, jbyteArray out) {
    (void)cls;
    int32_t len[256];
    size_t n_out = 0;
    jsize cap = (*env)->GetArrayLength(env, out);
    jbyte* src = (*env)->GetPrimitiveArrayCritical(env, data, NULL);
    if (!src) return DCZ_E_INVALID;
    jbyte* dst = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    if (!dst) {
        (*env)->ReleasePrimitiveArrayCritical(env, data, src, JNI_ABORT);
        return DCZ_E_INVALID;
    }
    int st = dcz_encode_block(CTX(ctx), (const uint8_t*)src, (size_t)length, len, (uint8_t*)dst, (size_t)cap, &n_out);
    (*env)->ReleasePrimitiveArrayCritical(env, out, dst, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, data, src, JNI_ABORT);
    if (st != DCZ_OK) return st;
    (*env)->SetIntArrayRegion(env, lengths, 0, 256, (const jint*)len);
    return (jint)n_out;
}

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_decodeBlock(JNIEnv* env, jclass cls, jlong ctx,
                                                                          jbyteArray comp, jint compSize,
                                                                          jintArray lengths, jbyteArray out,
                                                                          jint outSize, jlongArray errPos) {
    (void)cls;
    int32_t len[256];
    int64_t ep = -1;
    (*env)->GetIntArrayRegion(env, lengths, 0, 256, (jint*)len);
    jbyte* src = (*env)->GetPrimitiveArrayCritical(env, comp, NULL);
    if (!src) return DCZ_E_INVALID;
    jbyte* dst = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    if (!dst) {
        (*env)->ReleasePrimitiveArrayCritical(env, comp, src, JNI_ABORT);
        return DCZ_E_INVALID;
    }
    int st = dcz_decode_block(CTX(ctx), (const uint8_t*)src, (size_t)compSize, len, (uint8_t*)dst, (size_t)outSize, &ep);
    (*env)->ReleasePrimitiveArrayCritical(env, out, dst, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, comp, src, JNI_ABORT);
    if (st == DCZ_E_BADSTREAM) {
        jlong v = (jlong)ep;
        (*env)->SetLongArrayRegion(env, errPos, 0, 1, &v);
    }
    return st;
}
