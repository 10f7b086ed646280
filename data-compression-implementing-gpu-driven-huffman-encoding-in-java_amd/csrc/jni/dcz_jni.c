/*
 * dcz_jni.c -- JNI shim between com.datacomp.service.hip.HipNative and the C ABI of include/dcz.h.
 * Built only where a JDK is present (needs jni.h):
 *   gcc -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../../include \
 *       -o libdczjni.so dcz_jni.c -L.. -ldczhip -Wl,-rpath,'$ORIGIN'
 * NOT COMPILED in the authoring image (no JDK / jni.h there).  Java owns every byte[] / ByteBuffer; nothing native
 * outlives a call (the reference's ownership rule).
 *
 * Rules this file keeps (JNI spec, "GetPrimitiveArrayCritical"): no blocking call and no HIP call inside a critical
 * region -- in fact no critical regions at all: byte[] arguments are copied with Get/SetByteArrayRegion into the
 * context's pinned staging (dcz_ctx_pinned), which is also what makes the H2D/D2H copies run at PCIe rate; every
 * offset / length / array size coming from Java is checked against GetArrayLength / GetDirectBufferCapacity before
 * native code touches memory.  The batched entry points take DIRECT ByteBuffers (native memory, optionally page-locked
 * once through hostRegister) and bind dcz_compress_host / dcz_decompress_host: one JNI crossing per batch of chunks.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dcz.h"

#define CTX(h) ((dcz_ctx*)(intptr_t)(h))

static int array_ok(JNIEnv* env, jarray a, jlong need) {
    return a != NULL && need >= 0 && (jlong)(*env)->GetArrayLength(env, a) >= need;
}

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

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_ctxReserve(JNIEnv* env, jclass cls, jlong ctx, jlong n,
                                                                         jlong blockBytes) {
    (void)env; (void)cls;
    if (n < 0 || blockBytes <= 0) return DCZ_E_INVALID;
    return dcz_ctx_reserve(CTX(ctx), (size_t)n, (size_t)blockBytes);
}

JNIEXPORT jstring JNICALL Java_com_datacomp_service_hip_HipNative_strerror(JNIEnv* env, jclass cls, jint st) {
    (void)cls;
    return (*env)->NewStringUTF(env, dcz_strerror(st));
}

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_hostRegister(JNIEnv* env, jclass cls, jobject buf) {
    (void)cls;
    void* p = buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL;
    jlong cap = buf ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
    if (!p || cap <= 0) return DCZ_E_INVALID;
    return dcz_host_register(p, (size_t)cap);
}

JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_hostUnregister(JNIEnv* env, jclass cls, jobject buf) {
    (void)cls;
    void* p = buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL;
    return p ? dcz_host_unregister(p) : DCZ_E_INVALID;
}

/* dcz_histogram(ctx, data, offset, length, hist[256]) */
JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_histogram(JNIEnv* env, jclass cls, jlong ctx,
                                                                        jbyteArray data, jint offset, jint length,
                                                                        jlongArray hist) {
    (void)cls;
    int64_t h[256];
    if (offset < 0 || length < 0 || !array_ok(env, data, (jlong)offset + length) || !array_ok(env, hist, 256))
        return DCZ_E_INVALID;
    uint8_t* stage = (uint8_t*)dcz_ctx_pinned(CTX(ctx), 0, (size_t)length + 16);
    if (!stage) return DCZ_E_HIP;
    (*env)->GetByteArrayRegion(env, data, offset, length, (jbyte*)stage);  /* a copy, no critical region */
    int st = dcz_histogram(CTX(ctx), stage, 0, (size_t)length, h);
    if (st == DCZ_OK) (*env)->SetLongArrayRegion(env, hist, 0, 256, (const jlong*)h);
    return st;
}

/* dcz_encode_block: returns the compressed size (>= 0) or a negative status; fills lengths256 */
JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_encodeBlock(JNIEnv* env, jclass cls, jlong ctx,
                                                                          jbyteArray data, jint length,
                                                                          jintArray lengths, jbyteArray out) {
    (void)cls;
    int32_t len[256];
    size_t n_out = 0;
    if (length < 0 || !array_ok(env, data, length) || !array_ok(env, lengths, 256) || out == NULL) return DCZ_E_INVALID;
    const jsize cap = (*env)->GetArrayLength(env, out);
    uint8_t* src = (uint8_t*)dcz_ctx_pinned(CTX(ctx), 0, (size_t)length + 16);
    uint8_t* dst = (uint8_t*)dcz_ctx_pinned(CTX(ctx), 1, (size_t)cap + 16);
    if (!src || !dst) return DCZ_E_HIP;
    (*env)->GetByteArrayRegion(env, data, 0, length, (jbyte*)src);
    int st = dcz_encode_block(CTX(ctx), src, (size_t)length, len, dst, (size_t)cap, &n_out);
    if (st != DCZ_OK) return st;
    (*env)->SetByteArrayRegion(env, out, 0, (jsize)n_out, (const jbyte*)dst);
    (*env)->SetIntArrayRegion(env, lengths, 0, 256, (const jint*)len);
    return (jint)n_out;
}

/* dcz_decode_block: DCZ_OK or a negative status; errPos[0] = symbol index on DCZ_E_BADSTREAM */
JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_decodeBlock(JNIEnv* env, jclass cls, jlong ctx,
                                                                          jbyteArray comp, jint compSize,
                                                                          jintArray lengths, jbyteArray out,
                                                                          jint outSize, jlongArray errPos) {
    (void)cls;
    int32_t len[256];
    int64_t ep = -1;
    if (compSize < 0 || outSize < 0 || !array_ok(env, comp, compSize) || !array_ok(env, lengths, 256) ||
        !array_ok(env, out, outSize) || !array_ok(env, errPos, 1))
        return DCZ_E_INVALID;
    (*env)->GetIntArrayRegion(env, lengths, 0, 256, (jint*)len);
    uint8_t* src = (uint8_t*)dcz_ctx_pinned(CTX(ctx), 0, (size_t)compSize + 16);
    uint8_t* dst = (uint8_t*)dcz_ctx_pinned(CTX(ctx), 1, (size_t)outSize + 16);
    if (!src || !dst) return DCZ_E_HIP;
    (*env)->GetByteArrayRegion(env, comp, 0, compSize, (jbyte*)src);
    int st = dcz_decode_block(CTX(ctx), src, (size_t)compSize, len, dst, (size_t)outSize, &ep);
    if (st == DCZ_OK) (*env)->SetByteArrayRegion(env, out, 0, outSize, (const jbyte*)dst);
    if (st == DCZ_E_BADSTREAM) {
        jlong v = (jlong)ep;
        (*env)->SetLongArrayRegion(env, errPos, 0, 1, &v);
    }
    return st;
}

/* dcz_compress_host over direct ByteBuffers: returns the payload bytes (>= 0) or a negative status.
 * in: n bytes from position 0; out: capacity >= n; per-chunk columns of K = ceil(n / blockBytes) entries. */
JNIEXPORT jlong JNICALL Java_com_datacomp_service_hip_HipNative_compressBlocks(
    JNIEnv* env, jclass cls, jlong ctx, jobject in, jlong n, jint blockBytes, jobject out, jintArray compSize,
    jlongArray compOff, jbyteArray lens, jintArray status, jbyteArray sha) {
    (void)cls;
    if (n < 0 || blockBytes <= 0 || !in || !out) return DCZ_E_INVALID;
    uint8_t* pin = (uint8_t*)(*env)->GetDirectBufferAddress(env, in);
    uint8_t* pout = (uint8_t*)(*env)->GetDirectBufferAddress(env, out);
    if (!pin || !pout || (*env)->GetDirectBufferCapacity(env, in) < n) return DCZ_E_INVALID;
    const jlong ocap = (*env)->GetDirectBufferCapacity(env, out);
    const jlong K = (n + blockBytes - 1) / blockBytes;
    if (K > 0x7FFFFFFF || !array_ok(env, compSize, K) || !array_ok(env, compOff, K) || !array_ok(env, lens, K * 256) ||
        !array_ok(env, status, K) || (sha != NULL && !array_ok(env, sha, K * 32)))
        return DCZ_E_INVALID;
    if (K == 0) return 0;
    /* small host columns: one allocation */
    uint8_t* cols = (uint8_t*)calloc((size_t)K, 8 + 4 + 4 + 256 + 32);  /* cleared: an early failure must not hand heap bytes to Java */
    if (!cols) return DCZ_E_HIP;
    uint64_t* c_off = (uint64_t*)cols;
    uint32_t* c_size = (uint32_t*)(cols + 8 * K);
    int32_t* c_st = (int32_t*)(cols + 12 * K);
    uint8_t* c_len = cols + 16 * K;
    uint8_t* c_sha = c_len + 256 * K;
    uint64_t total = 0;
    int st = dcz_compress_host(CTX(ctx), pin, (size_t)n, (size_t)blockBytes, pout, (size_t)ocap, c_size, c_off, c_len, c_st,
                               &total, sha ? c_sha : NULL);
    (*env)->SetIntArrayRegion(env, status, 0, (jsize)K, (const jint*)c_st);
    if (st == DCZ_OK) {
        (*env)->SetIntArrayRegion(env, compSize, 0, (jsize)K, (const jint*)c_size);
        (*env)->SetLongArrayRegion(env, compOff, 0, (jsize)K, (const jlong*)c_off);
        (*env)->SetByteArrayRegion(env, lens, 0, (jsize)(K * 256), (const jbyte*)c_len);
        if (sha) (*env)->SetByteArrayRegion(env, sha, 0, (jsize)(K * 32), (const jbyte*)c_sha);
    }
    free(cols);
    return st == DCZ_OK ? (jlong)total : (jlong)st;
}

/* dcz_decompress_host over direct ByteBuffers: DCZ_OK, 1 (decoded, digests not computed) or a negative status;
 * per-chunk status / errPos are filled either way. */
JNIEXPORT jint JNICALL Java_com_datacomp_service_hip_HipNative_decompressBlocks(
    JNIEnv* env, jclass cls, jlong ctx, jobject comp, jlong compBytes, jlongArray compOff, jintArray compSize,
    jintArray origSize, jbyteArray lens, jint K, jlong outStride, jobject out, jintArray status, jlongArray errPos,
    jbyteArray sha) {
    (void)cls;
    if (compBytes < 0 || K < 0 || outStride <= 0 || !comp || !out) return DCZ_E_INVALID;
    uint8_t* pc = (uint8_t*)(*env)->GetDirectBufferAddress(env, comp);
    uint8_t* po = (uint8_t*)(*env)->GetDirectBufferAddress(env, out);
    if (!pc || !po || (*env)->GetDirectBufferCapacity(env, comp) < compBytes ||
        (*env)->GetDirectBufferCapacity(env, out) < (jlong)K * outStride)
        return DCZ_E_INVALID;
    if (!array_ok(env, compOff, K) || !array_ok(env, compSize, K) || !array_ok(env, origSize, K) ||
        !array_ok(env, lens, (jlong)K * 256) || !array_ok(env, status, K) || !array_ok(env, errPos, K) ||
        (sha != NULL && !array_ok(env, sha, (jlong)K * 32)))
        return DCZ_E_INVALID;
    if (K == 0) return DCZ_OK;
    uint8_t* cols = (uint8_t*)malloc((size_t)K * (8 + 8 + 4 + 4 + 4 + 256 + 32));
    if (!cols) return DCZ_E_HIP;
    uint64_t* c_off = (uint64_t*)cols;
    int64_t* c_ep = (int64_t*)(cols + 8 * (size_t)K);
    uint32_t* c_size = (uint32_t*)(cols + 16 * (size_t)K);
    uint32_t* c_orig = c_size + K;
    int32_t* c_st = (int32_t*)(c_orig + K);
    uint8_t* c_len = (uint8_t*)(c_st + K);
    uint8_t* c_sha = c_len + 256 * (size_t)K;
    (*env)->GetLongArrayRegion(env, compOff, 0, K, (jlong*)c_off);
    (*env)->GetIntArrayRegion(env, compSize, 0, K, (jint*)c_size);
    (*env)->GetIntArrayRegion(env, origSize, 0, K, (jint*)c_orig);
    (*env)->GetByteArrayRegion(env, lens, 0, K * 256, (jbyte*)c_len);
    memset(c_st, 0, 4 * (size_t)K);
    memset(c_ep, 0, 8 * (size_t)K);
    int st = dcz_decompress_host(CTX(ctx), pc, (size_t)compBytes, c_off, c_size, c_orig, c_len, (size_t)K, (size_t)outStride,
                                 po, c_st, c_ep, sha ? c_sha : NULL);
    (*env)->SetIntArrayRegion(env, status, 0, K, (const jint*)c_st);
    (*env)->SetLongArrayRegion(env, errPos, 0, K, (const jlong*)c_ep);
    if (st == DCZ_OK && sha) (*env)->SetByteArrayRegion(env, sha, 0, K * 32, (const jbyte*)c_sha);
    free(cols);
    return st;
}
