/*
 * dcz_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 * See dcz_oracle.h for the citation shorthands.  This file is the *checker*: it is never
 * linked into the product library and the product never falls back to it.
 *
 * Pinning status (tests/test_oracle_golden.py).  PINNED against every known-answer vector the
 * reference holds for this path: the exact-histogram KATs of
 * test/.../CpuFrequencyServiceTest.java:25-91, the MSB-first concatenation KATs of
 * test/.../ReductionBasedEncodingTest.java:27-163, the three root .bin files (SHA-256 + payload),
 * and the payload sizes the reference itself logged for its own test inputs
 * (app/logs/datacomp-2025-11-14.log:109-386), which pin the Java LCG, the canonical-code rule,
 * the single-symbol rule and the bit order end to end.
 * PARITY UNPINNED for one thing only: the choice among equal-weight internal nodes
 * (HN:52-58 returns 0).  Payload sizes cannot pin it (every Huffman tree has the same cost), no
 * reference test asserts code lengths, the reference holds no compressed file, and there is no
 * JVM in this image to run it.  For those ties the oracle rests on the literal restatement of
 * OpenJDK's published java.util.PriorityQueue (siftUp/siftDown below), nothing more.
 */
#include "dcz_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* Histogram: CFS:37-46.                                                                      */
/* ------------------------------------------------------------------------------------------ */
void orc_histogram(const uint8_t* data, size_t offset, size_t length, int64_t hist[256]) {
    memset(hist, 0, 256 * sizeof(int64_t));
    size_t end = offset + length;
    for (size_t i = offset; i < end; i++) hist[data[i] & 0xFF]++;
}

/* ------------------------------------------------------------------------------------------ */
/* HuffmanNode (HN:6-59) and java.util.PriorityQueue (OpenJDK, array binary heap).            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t freq;
    int32_t symbol; /* -1 for internal nodes (HN:26) */
    int32_t left, right;
} orc_node;

/* HN:52-58 compareTo: frequency, then symbol. */
static int node_cmp(const orc_node* a, const orc_node* b) {
    if (a->freq != b->freq) return a->freq < b->freq ? -1 : 1;
    if (a->symbol != b->symbol) return a->symbol < b->symbol ? -1 : 1;
    return 0;
}

typedef struct {
    int32_t q[512];
    int size;
    const orc_node* nodes;
} orc_pq;

/* PriorityQueue.offer -> siftUpComparable: ties do not move up. */
static void pq_offer(orc_pq* pq, int32_t x) {
    int k = pq->size++;
    while (k > 0) {
        int parent = (k - 1) >> 1;
        int32_t e = pq->q[parent];
        if (node_cmp(&pq->nodes[x], &pq->nodes[e]) >= 0) break;
        pq->q[k] = e;
        k = parent;
    }
    pq->q[k] = x;
}

/* PriorityQueue.poll -> siftDownComparable: ties prefer the left child; x stops on <=. */
static int32_t pq_poll(orc_pq* pq) {
    int32_t result = pq->q[0];
    int n = --pq->size;
    int32_t x = pq->q[n];
    if (n > 0) {
        int k = 0, half = n >> 1;
        while (k < half) {
            int child = (k << 1) + 1;
            int32_t c = pq->q[child];
            int right = child + 1;
            if (right < n && node_cmp(&pq->nodes[c], &pq->nodes[pq->q[right]]) > 0)
                c = pq->q[child = right];
            if (node_cmp(&pq->nodes[x], &pq->nodes[c]) <= 0) break;
            pq->q[k] = c;
            k = child;
        }
        pq->q[k] = x;
    }
    return result;
}

/* CH:85-92 extractLengths (iterative form of the same depth-first walk). */
static void extract_lengths(const orc_node* nodes, int32_t root, int32_t len[256]) {
    int32_t stack_n[512];
    int32_t stack_d[512];
    int sp = 0;
    stack_n[sp] = root;
    stack_d[sp++] = 0;
    while (sp > 0) {
        int32_t n = stack_n[--sp];
        int32_t d = stack_d[sp];
        if (nodes[n].left < 0 && nodes[n].right < 0) {
            len[nodes[n].symbol] = d;
        } else {
            stack_n[sp] = nodes[n].left;
            stack_d[sp++] = d + 1;
            stack_n[sp] = nodes[n].right;
            stack_d[sp++] = d + 1;
        }
    }
}

/* CH:55-80 buildCodeLengths. */
void orc_build_code_lengths(const int64_t freq[256], int32_t len[256]) {
    orc_node nodes[512];
    orc_pq pq;
    int nn = 0;
    pq.size = 0;
    pq.nodes = nodes;
    memset(len, 0, 256 * sizeof(int32_t));
    for (int i = 0; i < 256; i++) { /* CH:59-63 */
        if (freq[i] > 0) {
            nodes[nn].freq = freq[i];
            nodes[nn].symbol = i;
            nodes[nn].left = nodes[nn].right = -1;
            pq_offer(&pq, nn++);
        }
    }
    while (pq.size > 1) { /* CH:66-70 */
        int32_t l = pq_poll(&pq);
        int32_t r = pq_poll(&pq);
        nodes[nn].freq = nodes[l].freq + nodes[r].freq;
        nodes[nn].symbol = -1;
        nodes[nn].left = l;
        nodes[nn].right = r;
        pq_offer(&pq, nn++);
    }
    if (pq.size > 0) { /* CH:74-77 */
        int32_t root = pq_poll(&pq);
        extract_lengths(nodes, root, len);
    }
}

/* CH:99-132 generateCanonicalCodes. Java ints wrap; uint32_t arithmetic reproduces that. */
int orc_canonical_codes(const int32_t len[256], uint32_t code[256]) {
    int max_len = 0;
    uint32_t length_counts[33];
    uint32_t next_code[34];
    memset(length_counts, 0, sizeof length_counts);
    memset(code, 0, 256 * sizeof(uint32_t));
    for (int s = 0; s < 256; s++) {
        int l = len[s];
        if (l < 0 || l > 32) return -1; /* CH:106 ArrayIndexOutOfBounds */
        if (l > 0) {
            length_counts[l]++;
            if (l > max_len) max_len = l;
        }
    }
    uint32_t c = 0;
    next_code[0] = 0;
    for (int l = 1; l <= max_len; l++) { /* CH:112-117 */
        c = (c + length_counts[l - 1]) << 1;
        next_code[l] = c;
    }
    for (int s = 0; s < 256; s++) { /* CH:123-129 */
        int l = len[s];
        if (l > 0) code[s] = next_code[l]++;
    }
    return max_len;
}

/* CH:19-50 buildCanonicalCodes. */
int orc_build_canonical_codes(const int64_t freq[256], int32_t len[256], uint32_t code[256]) {
    int num = 0;
    for (int i = 0; i < 256; i++)
        if (freq[i] > 0) num++;
    memset(len, 0, 256 * sizeof(int32_t));
    memset(code, 0, 256 * sizeof(uint32_t));
    if (num == 0) return 0; /* CH:30-32 */
    if (num == 1) {         /* CH:35-45 */
        for (int i = 0; i < 256; i++)
            if (freq[i] > 0) {
                len[i] = 1;
                code[i] = 0;
                break;
            }
        return 1;
    }
    orc_build_code_lengths(freq, len);
    if (orc_canonical_codes(len, code) < 0) return -1;
    return num;
}

/* ------------------------------------------------------------------------------------------ */
/* Encode: CCS:303-315 encodeChunk + CCS:711-737 BitOutputStream.                             */
/* ------------------------------------------------------------------------------------------ */
int64_t orc_encode_block(const uint8_t* data, size_t n, const int32_t len[256],
                         const uint32_t code[256], uint8_t* out, size_t cap) {
    uint32_t current = 0;
    int nbits = 0;
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        int sym = data[i] & 0xFF;
        int l = len[sym];
        if (l == 0) continue; /* codes[symbol] == null (CCS:309) */
        uint32_t bits = code[sym];
        for (int b = l - 1; b >= 0; b--) { /* CCS:717-727 */
            current = (current << 1) | ((bits >> b) & 1u);
            if (++nbits == 8) {
                if (w >= cap) return -1;
                out[w++] = (uint8_t)current;
                current = 0;
                nbits = 0;
            }
        }
    }
    if (nbits > 0) { /* CCS:731-734 */
        if (w >= cap) return -1;
        out[w++] = (uint8_t)(current << (8 - nbits));
    }
    return (int64_t)w;
}

int64_t orc_encoded_size(const int64_t hist[256], const int32_t len[256]) {
    int64_t bits = 0;
    for (int s = 0; s < 256; s++) bits += hist[s] * (int64_t)len[s];
    return (bits + 7) / 8;
}

/* ------------------------------------------------------------------------------------------ */
/* Decode: TBD.  FastBitReader keeps (bytePos, bitPos); advance() stops moving bytePos at the  */
/* end of data and peek() zero-pads, which is equivalent to an absolute bit position over an   */
/* infinitely zero-extended buffer (TBD:180-231).                                             */
/* ------------------------------------------------------------------------------------------ */
#define ORC_TABLE_BITS 10
#define ORC_TABLE_SIZE (1 << ORC_TABLE_BITS)

void orc_build_lookup_table(const int32_t len[256], int32_t sym_out[1024], int32_t len_out[1024]) {
    uint32_t code[256];
    for (int i = 0; i < ORC_TABLE_SIZE; i++) { /* TBD:68-70 */
        sym_out[i] = -1;
        len_out[i] = 0;
    }
    if (orc_canonical_codes(len, code) < 0) return;
    for (int s = 0; s < 256; s++) { /* TBD:73-96 */
        int l = len[s];
        if (l == 0) continue;
        uint32_t v = code[s];
        if (l <= ORC_TABLE_BITS) {
            uint32_t nsuf = 1u << (ORC_TABLE_BITS - l);
            uint32_t base = v << (ORC_TABLE_BITS - l);
            for (uint32_t suf = 0; suf < nsuf; suf++) {
                uint32_t idx = (base | suf) & (ORC_TABLE_SIZE - 1);
                sym_out[idx] = s;
                len_out[idx] = l;
            }
        } else {
            uint32_t prefix = (v >> (l - ORC_TABLE_BITS)) & (ORC_TABLE_SIZE - 1);
            if (sym_out[prefix] == -1) len_out[prefix] = ORC_TABLE_BITS;
        }
    }
}

static inline uint32_t peek_bits(const uint8_t* d, size_t nbytes, uint64_t bitpos, int n) {
    uint32_t r = 0; /* TBD:180-211: one bit at a time, zero once past the end */
    for (int i = 0; i < n; i++) {
        uint64_t p = bitpos + (uint64_t)i;
        uint64_t byte = p >> 3;
        uint32_t bit = 0;
        if (byte < nbytes) bit = (d[byte] >> (7 - (p & 7))) & 1u;
        r = (r << 1) | bit;
    }
    return r;
}

int64_t orc_decode_block(const uint8_t* comp, size_t comp_size, const int32_t len[256],
                         uint8_t* out, size_t out_size) {
    int32_t tsym[ORC_TABLE_SIZE], tlen[ORC_TABLE_SIZE];
    uint32_t code[256];
    int max_len = orc_canonical_codes(len, code);
    if (max_len < 0) return -1;
    orc_build_lookup_table(len, tsym, tlen);
    uint64_t pos = 0;
    for (size_t i = 0; i < out_size; i++) { /* TBD:107-113 */
        uint32_t look = peek_bits(comp, comp_size, pos, ORC_TABLE_BITS);
        int sym = tsym[look];
        if (sym != -1) { /* TBD:126-129 */
            pos += (uint64_t)tlen[look];
        } else { /* TBD:140-152 decodeWithFallback; CH:219-228 decodeSymbol (HashMap: last put wins) */
            uint32_t c = 0;
            sym = -1;
            for (int l = 1; l <= max_len && sym == -1; l++) {
                c = (c << 1) | peek_bits(comp, comp_size, pos, 1);
                pos += 1;
                for (int s = 0; s < 256; s++)
                    if (len[s] == l && code[s] == c) sym = s;
            }
            if (sym == -1) return -((int64_t)i + 1); /* TBD:109-111 */
        }
        out[i] = (uint8_t)sym;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* java.util.Random (OpenJDK): 48-bit LCG; nextBytes emits each nextInt() low byte first.      */
/* ------------------------------------------------------------------------------------------ */
void orc_java_random_bytes(int64_t seed, uint8_t* buf, size_t n) {
    const uint64_t mask = (1ULL << 48) - 1;
    uint64_t s = ((uint64_t)seed ^ 0x5DEECE66DULL) & mask;
    size_t i = 0;
    while (i < n) {
        s = (s * 0x5DEECE66DULL + 0xBULL) & mask;
        int32_t rnd = (int32_t)(s >> 16);
        for (int k = 0; k < 4 && i < n; k++) {
            buf[i++] = (uint8_t)rnd;
            rnd >>= 8;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic config-4/5 distributions; integer-only so the device     */
/* generator in the product library reproduces them bit for bit.                              */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

#define ORC_TEXT_SYMS 97
static void text_tables(uint8_t order[ORC_TEXT_SYMS], uint32_t cum[ORC_TEXT_SYMS]) {
    static const char lower[] = " etaoinshrdlcumwfgypbvkjxqz";
    int n = 0;
    uint8_t used[256];
    memset(used, 0, sizeof used);
    for (int i = 0; lower[i]; i++) { order[n++] = (uint8_t)lower[i]; used[(uint8_t)lower[i]] = 1; }
    for (int i = 1; lower[i]; i++) { uint8_t u = (uint8_t)(lower[i] - 32); order[n++] = u; used[u] = 1; }
    for (int c = '0'; c <= '9'; c++) { order[n++] = (uint8_t)c; used[c] = 1; }
    for (int c = 33; c < 127; c++) if (!used[c]) order[n++] = (uint8_t)c;
    order[n++] = '\n';
    uint32_t acc = 0;
    for (int r = 0; r < ORC_TEXT_SYMS; r++) { /* Zipf head, geometric tail of rare symbols (long codes) */
        uint32_t w = (r < 64) ? 1000000u / (uint32_t)(r + 1) : (15625u >> ((r - 62) / 2));
        acc += w ? w : 1u;
        cum[r] = acc;
    }
}

void orc_gen_text(uint64_t seed, uint64_t start, uint8_t* buf, size_t n) {
    uint8_t order[ORC_TEXT_SYMS];
    uint32_t cum[ORC_TEXT_SYMS];
    text_tables(order, cum);
    uint64_t total = cum[ORC_TEXT_SYMS - 1];
    for (size_t i = 0; i < n; i++) {
        uint64_t x = splitmix64(seed + (start + i) * 0x9E3779B97F4A7C15ULL);
        uint32_t t = (uint32_t)(((x >> 32) * total) >> 32);
        int r = 0;
        while (cum[r] <= t) r++;
        buf[i] = order[r];
    }
}

void orc_gen_lowentropy(uint64_t seed, uint64_t start, uint8_t* buf, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint64_t x = splitmix64(seed + (start + i) * 0x9E3779B97F4A7C15ULL);
        uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
        buf[i] = (hi % 100u == 0u) ? (uint8_t)(1u + lo % 255u) : 0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SHA-256 (FIPS 180-4).                                                                      */
/* ------------------------------------------------------------------------------------------ */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void orc_sha256(const uint8_t* data, size_t n, uint8_t digest[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = n / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, data + 64 * i);
    uint8_t tail[128];
    size_t rem = n - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        digest[4 * i] = (uint8_t)(h[i] >> 24);
        digest[4 * i + 1] = (uint8_t)(h[i] >> 16);
        digest[4 * i + 2] = (uint8_t)(h[i] >> 8);
        digest[4 * i + 3] = (uint8_t)h[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* CPU baseline: K independent chunks on a fixed pool of chunk workers, the shape of          */
/* CCS:90-98 (compress) and CCS:400-441 (decompress), hot-path stages only (no I/O, no SHA).   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* data;
    size_t n, block_bytes, nblocks;
    uint8_t** comp;
    int64_t* comp_size;
    int32_t (*lens)[256];
    uint8_t* decoded;
    volatile long next;
    int phase; /* 0 = encode, 1 = decode */
    int failed;
    pthread_mutex_t mu;
} rt_job;

static long rt_take(rt_job* j) {
    pthread_mutex_lock(&j->mu);
    long b = j->next < (long)j->nblocks ? j->next++ : -1;
    pthread_mutex_unlock(&j->mu);
    return b;
}

static void* rt_worker(void* arg) {
    rt_job* j = (rt_job*)arg;
    for (;;) {
        long b = rt_take(j);
        if (b < 0) break;
        size_t off = (size_t)b * j->block_bytes;
        size_t len = j->n - off < j->block_bytes ? j->n - off : j->block_bytes;
        if (j->phase == 0) {
            int64_t hist[256];
            uint32_t code[256];
            orc_histogram(j->data, off, len, hist);
            if (orc_build_canonical_codes(hist, j->lens[b], code) < 0) { j->failed = 1; continue; }
            int64_t cs = orc_encoded_size(hist, j->lens[b]);
            j->comp[b] = (uint8_t*)malloc((size_t)cs + 1);
            j->comp_size[b] = orc_encode_block(j->data + off, len, j->lens[b], code, j->comp[b], (size_t)cs + 1);
            if (j->comp_size[b] != cs) j->failed = 1;
        } else {
            if (orc_decode_block(j->comp[b], (size_t)j->comp_size[b], j->lens[b], j->decoded + off, len) != 0)
                j->failed = 1;
        }
    }
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_roundtrip_blocks_mt(const uint8_t* data, size_t n, size_t block_bytes, int threads,
                            double* enc_seconds, double* dec_seconds, uint64_t* comp_total) {
    rt_job j;
    memset(&j, 0, sizeof j);
    j.data = data;
    j.n = n;
    j.block_bytes = block_bytes;
    j.nblocks = (n + block_bytes - 1) / block_bytes;
    j.comp = (uint8_t**)calloc(j.nblocks ? j.nblocks : 1, sizeof(uint8_t*));
    j.comp_size = (int64_t*)calloc(j.nblocks ? j.nblocks : 1, sizeof(int64_t));
    j.lens = (int32_t(*)[256])calloc(j.nblocks ? j.nblocks : 1, sizeof(int32_t[256]));
    j.decoded = (uint8_t*)malloc(n ? n : 1);
    pthread_mutex_init(&j.mu, NULL);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    double secs[2] = {0.0, 0.0};
    for (int phase = 0; phase < 2; phase++) {
        j.phase = phase;
        j.next = 0;
        double t0 = now_s();
        for (int i = 0; i < threads; i++) pthread_create(&tid[i], NULL, rt_worker, &j);
        for (int i = 0; i < threads; i++) pthread_join(tid[i], NULL);
        secs[phase] = now_s() - t0;
    }
    if (enc_seconds) *enc_seconds = secs[0];
    if (dec_seconds) *dec_seconds = secs[1];
    uint64_t tot = 0;
    for (size_t b = 0; b < j.nblocks; b++) tot += (uint64_t)j.comp_size[b];
    if (comp_total) *comp_total = tot;
    int ok = !j.failed && (n == 0 || memcmp(j.decoded, data, n) == 0);
    for (size_t b = 0; b < j.nblocks; b++) free(j.comp[b]);
    free(j.comp);
    free(j.comp_size);
    free(j.lens);
    free(j.decoded);
    pthread_mutex_destroy(&j.mu);
    return ok ? 0 : -1;
}
