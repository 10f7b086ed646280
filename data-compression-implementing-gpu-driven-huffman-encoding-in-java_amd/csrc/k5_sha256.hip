// k5_sha256.hip -- K5: SHA-256 of every block (FIPS 180-4), the per-chunk checksum of the DCZF container (gfx950).
//
// Replaces ChecksumUtil.computeSha256(byte[], int, int) = MessageDigest("SHA-256")
// (util/ChecksumUtil.java:11-27) as called once per chunk by CpuCompressionService.processChunk
// (service/cpu/CpuCompressionService.java:224-231) and by the decompressor's verification (:536-550).
// The global checksum (SHA-256 over the concatenated chunk digests, :106-109, :126) is 32*K bytes and stays on the host.
//
// SHA-256 is a serial chain over the 64-byte pieces of one message, so the parallelism is across blocks: ONE LANE PER
// BLOCK, 64 rounds fully unrolled with the message schedule in 16 rotating registers (all indices compile-time).
// Rotates are v_alignbit, the three-input XORs v_xor3, Ch/Maj one v_bitop3/v_bfi each, sums v_add3.
// Every lane streams its own block with 16-byte loads (4 per 64-byte piece; the lanes of a wave are block_bytes
// apart, so each load instruction touches 64 lines and uses each of them completely).
// Throughput therefore scales with the NUMBER of blocks in flight (K lanes of the chip's 16384 x occupancy): it is a
// checksum stage beside the codec, not part of the timed hot path (SURVEY.md 8(d) excludes CHECKSUM_* stages).
#include "dcz_internal.h"

namespace dcz {

namespace {

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, (uint32_t)n); }
__device__ __forceinline__ uint32_t big_sigma0(uint32_t x) { return rotr(x, 2) ^ rotr(x, 13) ^ rotr(x, 22); }
__device__ __forceinline__ uint32_t big_sigma1(uint32_t x) { return rotr(x, 6) ^ rotr(x, 11) ^ rotr(x, 25); }
__device__ __forceinline__ uint32_t small_sigma0(uint32_t x) { return rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3); }
__device__ __forceinline__ uint32_t small_sigma1(uint32_t x) { return rotr(x, 17) ^ rotr(x, 19) ^ (x >> 10); }
__device__ __forceinline__ uint32_t ch(uint32_t e, uint32_t f, uint32_t g) { return (e & f) ^ (~e & g); }
__device__ __forceinline__ uint32_t maj(uint32_t a, uint32_t b, uint32_t c) { return (a & b) ^ (a & c) ^ (b & c); }

__constant__ const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

// One 64-byte piece: w[16] = its big-endian words (destroyed), h[8] updated.
__device__ __forceinline__ void compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        if (t >= 16) {
            w[t & 15] = small_sigma1(w[(t - 2) & 15]) + w[(t - 7) & 15] + small_sigma0(w[(t - 15) & 15]) + w[t & 15];
        }
        const uint32_t t1 = hh + big_sigma1(e) + ch(e, f, g) + K256[t] + w[t & 15];
        const uint32_t t2 = big_sigma0(a) + maj(a, b, c);
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}

}  // namespace

__global__ __launch_bounds__(64) void k5_sha256(const uint8_t* __restrict__ in, size_t n, size_t block_bytes, uint32_t K,
                                                uint8_t* __restrict__ digests) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= K) return;
    const uint64_t start = (uint64_t)b * block_bytes;
    const uint64_t end = (start + block_bytes < n) ? start + block_bytes : (uint64_t)n;
    const uint64_t len = end > start ? end - start : 0;
    const uint8_t* p = in + start;
    const bool aligned = (((uintptr_t)p) & 15u) == 0u;

    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t w[16];
    uint64_t off = 0;
    if (aligned) {
        for (; off + 64 <= len; off += 64) {
            const uint4* q = reinterpret_cast<const uint4*>(p + off);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 v = q[i];
                w[4 * i + 0] = bswap32(v.x);
                w[4 * i + 1] = bswap32(v.y);
                w[4 * i + 2] = bswap32(v.z);
                w[4 * i + 3] = bswap32(v.w);
            }
            compress(h, w);
        }
    } else {
        for (; off + 64 <= len; off += 64) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint8_t* s = p + off + 4 * i;
                w[i] = ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | (uint32_t)s[3];
            }
            compress(h, w);
        }
    }
    // padding: the remaining < 64 bytes, 0x80, zeros, the bit length as a big-endian u64 (one or two more pieces)
    const uint32_t rem = (uint32_t)(len - off);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t j = 4u * (uint32_t)i + (uint32_t)k;
            uint32_t by = 0;
            if (j < rem) by = p[off + j];
            else if (j == rem) by = 0x80u;
            v |= by << (24 - 8 * k);
        }
        w[i] = v;
    }
    const uint64_t bitlen = len * 8ull;
    if (rem >= 56u) {  // no room for the length: it goes into a piece of its own
        compress(h, w);
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = 0;
    }
    w[14] = (uint32_t)(bitlen >> 32);
    w[15] = (uint32_t)bitlen;
    compress(h, w);
    uint8_t* o = digests + (uint64_t)b * 32u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        o[4 * i + 0] = (uint8_t)(h[i] >> 24);
        o[4 * i + 1] = (uint8_t)(h[i] >> 16);
        o[4 * i + 2] = (uint8_t)(h[i] >> 8);
        o[4 * i + 3] = (uint8_t)h[i];
    }
}

void launch_sha256(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t K, uint8_t* d_digests, hipStream_t s) {
    if (K == 0) return;
    hipLaunchKernelGGL(k5_sha256, dim3((K + 63u) / 64u), dim3(64), 0, s, d_in, n, block_bytes, K, d_digests);
}

}  // namespace dcz
