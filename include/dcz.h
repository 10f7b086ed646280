/*
 * dcz.h -- C ABI of libdczhip.so: the MI355X (gfx950) hot path of the DataComp compressor.
 *
 * This is the drop-in boundary.  The reference (vuyraj/Data-Compression-...-Huffman-Encoding-in-Java)
 * has no native interface; its seam is two Java interfaces picked by a factory
 * (service/FrequencyService.java:6-27, service/CompressionService.java:11-66,
 * service/ServiceFactory.java:21-70).  Every entry point below names the reference method it
 * replaces; INTEGRATION.md shows the JNI binding (`com.datacomp.service.hip.HipNative`) that a
 * maintainer adds on the Java side.  Paths are relative to app/src/main/java/com/datacomp/.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; `int` status, 0 = DCZ_OK, < 0 = error;
 * caller owns every buffer; a dcz_ctx is bound to one device and may be used by one thread at a
 * time (create one per worker thread -- the reference calls the seam from 1..8 pool threads,
 * service/cpu/CpuCompressionService.java:42-44, service/gpu/GpuCompressionService.java:103);
 * `stream` is a hipStream_t passed as void* (NULL = the ctx's own stream).  Everything is
 * computed by HIP kernels: there is no CPU fallback inside the library (the Java/host service
 * keeps the reference's own fallback-to-CpuCompressionService contract, GpuCompressionService.java:145-167).
 */
#ifndef DCZ_H
#define DCZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCZ_OK 0
#define DCZ_E_INVALID (-1)   /* bad argument (IllegalArgumentException in the reference, core/CanonicalHuffman.java:20-22) */
#define DCZ_E_NODEVICE (-2)  /* no usable gfx950 device: isAvailable() == false */
#define DCZ_E_HIP (-3)       /* HIP runtime failure; text via dcz_last_error() */
#define DCZ_E_CAPACITY (-4)  /* output buffer too small */
#define DCZ_E_BADSTREAM (-5) /* "Huffman decode error at position i" (core/TableBasedHuffmanDecoder.java:109-111) */
#define DCZ_E_CODELEN (-6)   /* a code length would exceed 32 (ArrayIndexOutOfBounds at core/CanonicalHuffman.java:106) */
#define DCZ_E_BADTABLE (-7)  /* stored length table is not a prefix code (length > 32 or Kraft sum > 1) */

#define DCZ_SEGMENT_BYTES 32768u /* intra-block work unit of the histogram and encode kernels */

typedef struct dcz_ctx dcz_ctx;

/* ---- lifetime --------------------------------------------------------------------------- */

/* GpuFrequencyService.isAvailable (service/gpu/GpuFrequencyService.java:255-283) without launching
 * a probe kernel: number of visible gfx950 devices, 0 if none. */
int dcz_device_count(void);

/* new GpuCompressionService(...) device acquisition (service/gpu/GpuCompressionService.java:46-73). */
int dcz_ctx_create(int device, dcz_ctx** out);

/* CompressionService.close (service/gpu/GpuCompressionService.java:1563-1593). */
void dcz_ctx_destroy(dcz_ctx* ctx);

/* The context's own stream (a hipStream_t, created non-blocking): what every entry point runs on when its `stream`
 * argument is NULL.  Hosts that mix library calls with their own HIP work order the two with events on this handle
 * (the Python mirror wraps it in torch.cuda.ExternalStream). */
void* dcz_ctx_stream(dcz_ctx* ctx);

const char* dcz_strerror(int status);
const char* dcz_last_error(const dcz_ctx* ctx);

/* Pre-allocate the device workspace (encoder AND decoder) for inputs of up to n bytes in blocks of block_bytes, so
 * that later dcz_compress_blocks / dcz_decompress_blocks calls allocate nothing and never synchronise with the host:
 * they can then be captured in a hipGraph (tests/test_gpu_parity.py::test_compress_and_decompress_are_capturable...).
 * Without it the first call of a larger geometry grows the workspace (one stream synchronisation + hipMalloc). */
int dcz_ctx_reserve(dcz_ctx* ctx, size_t n, size_t block_bytes);

/* ---- single-block primitives, host pointers ----------------------------------------------- */

/* FrequencyService.computeHistogram(byte[] data, int offset, int length) -> long[256]
 * (service/FrequencyService.java:16; CPU service/cpu/CpuFrequencyService.java:29-46;
 * TornadoVM service/gpu/GpuFrequencyService.java:87-149). */
int dcz_histogram(dcz_ctx* ctx, const uint8_t* data, size_t offset, size_t length, int64_t hist[256]);

/* CanonicalHuffman.buildCanonicalCodes(long[256]) (core/CanonicalHuffman.java:19-50), exact
 * java.util.PriorityQueue tie order (core/CanonicalHuffman.java:55-92, core/HuffmanNode.java:52-58).
 * len[s] = 0 and code[s] = 0 for absent symbols (null HuffmanCode). */
int dcz_build_codes(dcz_ctx* ctx, const int64_t hist[256], int32_t len[256], uint32_t code[256]);

/* CanonicalHuffman.generateCanonicalCodesFromLengths(int[256]) (core/CanonicalHuffman.java:99-146). */
int dcz_codes_from_lengths(dcz_ctx* ctx, const int32_t len[256], uint32_t code[256]);

/* One chunk of CpuCompressionService.processChunk, hot stages only
 * (service/cpu/CpuCompressionService.java:233-258: histogram -> buildCanonicalCodes -> encodeChunk
 * service/cpu/CpuCompressionService.java:303-315, BitOutputStream :711-737).
 * Writes ceil(bits/8) payload bytes and the 256 code lengths the footer stores. */
int dcz_encode_block(dcz_ctx* ctx, const uint8_t* data, size_t n, int32_t len_out[256], uint8_t* out,
                     size_t cap, size_t* out_len);

/* One chunk of CpuCompressionService.decodeChunkParallel, hot stages only
 * (service/cpu/CpuCompressionService.java:513-532: rebuildCodes :582-586 ->
 * TableBasedHuffmanDecoder.decode core/TableBasedHuffmanDecoder.java:103-152).
 * Decodes exactly out_size symbols; bits past comp_size read as zero (TableBasedHuffmanDecoder.java:204-208).
 * On DCZ_E_BADSTREAM *err_pos is the symbol index i of the reference's exception text. */
int dcz_decode_block(dcz_ctx* ctx, const uint8_t* comp, size_t comp_size, const int32_t len[256],
                     uint8_t* out, size_t out_size, int64_t* err_pos);

/* ---- batched, device-resident pipeline (what bench.py times) -------------------------------- */

/* The compress loop of CpuCompressionService.compress (service/cpu/CpuCompressionService.java:90-151)
 * for K = ceil(n / block_bytes) chunks already resident in HBM.
 *   d_in        n input bytes
 *   d_out       payloads, concatenated in chunk order with no gaps (CpuCompressionService.java:160-163);
 *               out_cap >= n is always enough (a Huffman code never beats 8 bits/symbol the wrong way)
 *   d_comp_size K x u32  compressedSize of each chunk            (core/CompressionHeader.java:76)
 *   d_comp_off  K x u64  byte offset of each payload in d_out     (core/CompressionHeader.java:75, minus the rank's base)
 *   d_len       K x 256 x u8 code lengths                         (core/CompressionHeader.java:80-83)
 *   d_status    K x i32  DCZ_OK / DCZ_E_CODELEN / DCZ_E_CAPACITY per chunk
 *   d_total     1 x u64  sum of compressedSize (may be NULL)
 * All pointers are device pointers.  Asynchronous on `stream`; no host synchronisation.
 * d_out[0, min(out_cap, n)) may be written anywhere by the call, not only inside [0, *d_total): when the calls before had
 * nothing but chunks whose code is 256 symbols of 8 bits (payload = input), the histogram pass also stores the input at
 * its own offsets of d_out (valid payload if this call is like them; overwritten by the encoder where it is not).  That
 * needs out_cap >= n and d_in - d_out a multiple of 16; DCZ_NO_IN_PLACE=1 turns it off. */
int dcz_compress_blocks(dcz_ctx* ctx, const void* d_in, size_t n, size_t block_bytes, void* d_out,
                        size_t out_cap, uint32_t* d_comp_size, uint64_t* d_comp_off, uint8_t* d_len,
                        int32_t* d_status, uint64_t* d_total, void* stream);

/* The decode loop of CpuCompressionService.decompress (service/cpu/CpuCompressionService.java:400-479)
 * for K chunks whose payloads are resident in HBM.
 *   d_comp       payload bytes; comp_bytes = total size of that buffer
 *   d_comp_off / d_comp_size / d_orig_size / d_len   per-chunk footer fields (K entries)
 *   d_out        chunk k is written at d_out + k * out_stride (originalOffset, CompressionHeader.java:73)
 *   d_status     K x i32  DCZ_OK / DCZ_E_BADSTREAM / DCZ_E_BADTABLE / DCZ_E_INVALID
 *   d_errpos     K x i64  symbol index of the decode error (may be NULL)
 * The footer fields are untrusted: a chunk with comp_off + comp_size > comp_bytes or orig_size > out_stride gets
 * DCZ_E_INVALID and is neither read nor written.  Alignment: d_len 16 bytes; the payload may start at any byte, but
 * the 16-byte aligned units that overlap [d_comp, d_comp + comp_bytes) must be readable (hipMalloc'ed buffers are;
 * a sub-range of a larger buffer is).  Asynchronous on `stream`; after dcz_ctx_reserve no host synchronisation and no
 * allocation.  With fewer than 128 chunks of >= 256 KiB of payload on average a chunk is decoded by many workgroups
 * (csrc/k4_split.hip), so a single 16-32 MiB chunk -- the reference's own chunk sizes -- uses the whole chip. */
int dcz_decompress_blocks(dcz_ctx* ctx, const void* d_comp, size_t comp_bytes, const uint64_t* d_comp_off,
                          const uint32_t* d_comp_size, const uint32_t* d_orig_size, const uint8_t* d_len,
                          size_t K, size_t out_stride, void* d_out, int32_t* d_status, int64_t* d_errpos,
                          void* stream);

/* ---- batched pipeline on host buffers (one call per batch of chunks; the JNI twins bind these) ---- */

/* Page-lock a caller-owned host range (a direct ByteBuffer) so that the copies below run at full PCIe rate. */
int dcz_host_register(void* p, size_t n);
int dcz_host_unregister(void* p);
/* Library-owned pinned staging (grow-only, slot 0 or 1), for callers whose bytes live in movable memory (byte[]). */
void* dcz_ctx_pinned(dcz_ctx* ctx, int slot, size_t bytes);

/* dcz_compress_blocks for n bytes in host memory: H2D, K1-K3 (and K5 when sha256 != NULL: 32 bytes per chunk,
 * ChecksumUtil.computeSha256, util/ChecksumUtil.java:11-27), D2H of the footer columns and the payload.  Everything in
 * host memory; comp_size / comp_off / len / status as in dcz_compress_blocks.  Synchronous.  Returns the first failing
 * chunk's status, DCZ_E_CAPACITY if the payload exceeds out_cap. */
int dcz_compress_host(dcz_ctx* ctx, const uint8_t* in, size_t n, size_t block_bytes, uint8_t* out, size_t out_cap,
                      uint32_t* comp_size, uint64_t* comp_off, uint8_t* len, int32_t* status, uint64_t* total,
                      uint8_t* sha256);

/* dcz_decompress_blocks for payloads in host memory; chunk k is written at out + k * out_stride.  sha256 != NULL asks
 * for the digests of the decoded chunks (computed on the device when every chunk but the last fills its stride;
 * otherwise the call returns 1 = decoded, digests not computed).  Per-chunk status / errpos as in dcz_decompress_blocks. */
int dcz_decompress_host(dcz_ctx* ctx, const uint8_t* comp, size_t comp_bytes, const uint64_t* comp_off,
                        const uint32_t* comp_size, const uint32_t* orig_size, const uint8_t* len, size_t K,
                        size_t out_stride, uint8_t* out, int32_t* status, int64_t* errpos, uint8_t* sha256);

/* ---- measurement hooks --------------------------------------------------------------------- */

enum {
    DCZ_K_HISTOGRAM = 0, /* K1 */
    DCZ_K_CODEBUILD = 1, /* K2 */
    DCZ_K_OFFSETS = 2,   /* payload offset scan */
    DCZ_K_ENCODE = 3,    /* K3 */
    DCZ_K_DECODE = 4,    /* K4 */
    DCZ_K_HISTOGRAM_COPY = 5, /* K1 that also stores the input at the same offsets of the output: launched instead of K1
                               * when the calls before had nothing but blocks whose code is 256 symbols of 8 bits (payload =
                               * input, the reference's high-entropy case); K3 then finds those blocks in place */
    DCZ_K_COUNT = 6
};

/* When on, every kernel launch is bracketed by hipEvents on its own stream (the per-stage timers
 * of model/StageMetrics.java:45-49).  dcz_ctx_kernel_time synchronises the pending events. */
int dcz_ctx_set_profiling(dcz_ctx* ctx, int on);
int dcz_ctx_reset_profiling(dcz_ctx* ctx);
int dcz_ctx_kernel_time(dcz_ctx* ctx, int kernel, double* total_ms, uint64_t* launches);
/* Launch shapes chosen since the context was created or dcz_ctx_reset_profiling was called (counted whether or not
 * profiling is on): counts[0] decode calls that gave k4_fixed its flat grid, [1] the persistent grid; [2] compress calls
 * that gave k3_copy_identity its flat grid, [3] the persistent grid (which includes every call counted under
 * DCZ_K_HISTOGRAM_COPY).  The choice follows what the last COMPLETED calls met, never the number of calls queued: a
 * benchmark loop that queues many calls must see one shape throughout (tests/test_gpu_parity.py asserts it). */
int dcz_ctx_launch_shapes(dcz_ctx* ctx, uint64_t counts[4]);

/* ---- checksums ------------------------------------------------ */

/* SHA-256 of each of the K = ceil(n / block_bytes) blocks of a device-resident buffer, 32 bytes per block into
 * d_digests (device): ChecksumUtil.computeSha256(byte[],int,int) (util/ChecksumUtil.java:11-27) as called per chunk by
 * CpuCompressionService.processChunk (service/cpu/CpuCompressionService.java:224-231) and by the decompressor's
 * verification (:536-550).  One lane per block: worth using when there are >= ~1000 blocks. */
int dcz_sha256_blocks(dcz_ctx* ctx, const void* d_in, size_t n, size_t block_bytes, void* d_digests, void* stream);

/* ---- reproducible inputs (util/TestDataGenerator.java:26-73) on the device ------------------- */

/* java.util.Random(seed).nextBytes stream, bytes [start, start+n); start must be a multiple of 4. */
int dczu_fill_java_random(dcz_ctx* ctx, void* d_buf, size_t n, int64_t seed, uint64_t start, void* stream);
/* SURVEY.md section 8(d) config-4 (order-0 English-like) and config-5 (zeros + 1 % noise) streams. */
int dczu_fill_text(dcz_ctx* ctx, void* d_buf, size_t n, uint64_t seed, uint64_t start, void* stream);
int dczu_fill_lowentropy(dcz_ctx* ctx, void* d_buf, size_t n, uint64_t seed, uint64_t start, void* stream);

#ifdef __cplusplus
}
#endif
#endif
