/*
 * dcz_oracle.h -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (libdczhip.so) never includes, links or calls anything in oracle/.
 *
 * Every function names the reference file:line it restates.  Shorthands (relative to
 * /root/reference/app/src/main/java/com/datacomp/):
 *   CH  = core/CanonicalHuffman.java           HN  = core/HuffmanNode.java
 *   TBD = core/TableBasedHuffmanDecoder.java   CCS = service/cpu/CpuCompressionService.java
 *   CFS = service/cpu/CpuFrequencyService.java HDR = core/CompressionHeader.java
 *   TDG = util/TestDataGenerator.java
 * java.util.PriorityQueue / java.util.Random are JDK classes that are not under /root/reference;
 * their published algorithms (OpenJDK 8..21, identical) are restated and cited in the .c file.
 */
#ifndef DCZ_ORACLE_H
#define DCZ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* CFS:37-46 computeHistogramSequential (offset/length window, bytes taken unsigned). */
void orc_histogram(const uint8_t* data, size_t offset, size_t length, int64_t hist[256]);

/* CH:19-50 buildCanonicalCodes: returns number of symbols with freq>0; len[s]=0 when absent.
 * 0 symbols -> all zero; 1 symbol -> len 1 code 0; else CH:55-92 + CH:99-132.
 * Returns -1 if a length exceeds 32 (CH:102/106 would throw ArrayIndexOutOfBounds). */
int orc_build_canonical_codes(const int64_t freq[256], int32_t len[256], uint32_t code[256]);

/* CH:55-92 buildCodeLengths + extractLengths (literal java.util.PriorityQueue emulation). */
void orc_build_code_lengths(const int64_t freq[256], int32_t len[256]);

/* CH:99-132 generateCanonicalCodes / CH:141-146 generateCanonicalCodesFromLengths.
 * Returns max length, or -1 if some length is outside 0..32. */
int orc_canonical_codes(const int32_t len[256], uint32_t code[256]);

/* CCS:303-315 encodeChunk + CCS:711-737 BitOutputStream (bit-serial, MSB first).
 * Returns number of bytes written (ceil(bits/8)); -1 if cap is too small. */
int64_t orc_encode_block(const uint8_t* data, size_t n, const int32_t len[256],
                         const uint32_t code[256], uint8_t* out, size_t cap);

/* Number of payload bytes encodeChunk would produce (no output). */
int64_t orc_encoded_size(const int64_t hist[256], const int32_t len[256]);

/* TBD:36-97 table build + TBD:103-152 decode + TBD:165-232 FastBitReader + CH:161-229 fallback map.
 * Decodes exactly out_size symbols.  Returns 0 on success, or -(i+1) where i is the symbol
 * index of "Huffman decode error at position i" (TBD:109-111). */
int64_t orc_decode_block(const uint8_t* comp, size_t comp_size, const int32_t len[256],
                         uint8_t* out, size_t out_size);

/* TBD:66-97: expose the 1024-entry table for table-level tests (symbol or -1, codeLength). */
void orc_build_lookup_table(const int32_t len[256], int32_t sym_out[1024], int32_t len_out[1024]);

/* java.util.Random(seed).nextBytes over consecutive buffers, as TDG:26-50 uses it. */
void orc_java_random_bytes(int64_t seed, uint8_t* buf, size_t n);

/* Synthetic distributions of SURVEY.md section 8(d), configs 4 and 5 (counter-based, keyed by
 * absolute byte index so any sub-range can be regenerated on the device). */
void orc_gen_text(uint64_t seed, uint64_t start, uint8_t* buf, size_t n);
void orc_gen_lowentropy(uint64_t seed, uint64_t start, uint8_t* buf, size_t n);

/* FIPS 180-4 SHA-256 (the JDK MessageDigest the reference calls at util/ChecksumUtil.java:11-27). */
void orc_sha256(const uint8_t* data, size_t n, uint8_t digest[32]);

/* Whole-job CPU baseline over K independent blocks with `threads` chunk workers
 * (CCS:42-44 uses max(2,min(nproc,8))): histogram -> codes -> encode, then table decode.
 * Returns 0 on success and fills seconds for encode and decode legs; verifies the round trip. */
int orc_roundtrip_blocks_mt(const uint8_t* data, size_t n, size_t block_bytes, int threads,
                            double* enc_seconds, double* dec_seconds, uint64_t* comp_total);

#ifdef __cplusplus
}
#endif
#endif
