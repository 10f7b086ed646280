// gen.hip -- reproducible input generators on the device (util/TestDataGenerator.java:26-73 analogue).
//
// dczu_fill_java_random reproduces java.util.Random(seed).nextBytes over consecutive buffers exactly
// (util/TestDataGenerator.java:30-40): 48-bit LCG, each nextInt() emitted low byte first.  Every
// thread jumps to its position with the affine power (A^k, C_k) of the LCG, so any sub-range of
// the stream can be produced in parallel.  The text / low-entropy streams are the integer-only
// recipes of SURVEY.md section 8(d), configs 4 and 5 (same arithmetic as oracle/dcz_oracle.c).
#include "dcz_internal.h"

namespace dcz {

constexpr unsigned long long LCG_A = 0x5DEECE66DULL;
constexpr unsigned long long LCG_C = 0xBULL;
constexpr unsigned long long LCG_MASK = (1ULL << 48) - 1;

__global__ __launch_bounds__(256) void gen_java_random(uint8_t* __restrict__ d, size_t n, long long seed,
                                                       unsigned long long start) {
    // thread t produces ints [q0, q0 + 4) of the stream = bytes [start + 16 t, +16)
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long byte0 = 16ull * t;
    if (byte0 >= n) return;
    const unsigned long long q0 = (start >> 2) + 4ull * t;
    // state after k = q0 steps: s_k = A^k s_0 + C_k
    unsigned long long acc_a = 1, acc_c = 0, cur_a = LCG_A, cur_c = LCG_C;
    for (unsigned long long k = q0; k != 0; k >>= 1) {
        if (k & 1) {
            acc_a = (acc_a * cur_a) & LCG_MASK;
            acc_c = (acc_c * cur_a + cur_c) & LCG_MASK;
        }
        cur_c = (cur_c * (cur_a + 1)) & LCG_MASK;
        cur_a = (cur_a * cur_a) & LCG_MASK;
    }
    unsigned long long s = (((unsigned long long)seed ^ LCG_A) & LCG_MASK);
    s = (acc_a * s + acc_c) & LCG_MASK;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s = (s * LCG_A + LCG_C) & LCG_MASK;
        w[i] = (uint32_t)(s >> 16);
    }
    if (byte0 + 16 <= n && (((uintptr_t)d) & 15u) == 0) {
        *reinterpret_cast<uint4*>(d + byte0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int i = 0; i < 16; i++)
            if (byte0 + i < n) d[byte0 + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

constexpr int TEXT_SYMS = 97;

__global__ __launch_bounds__(256) void gen_text(uint8_t* __restrict__ d, size_t n, unsigned long long seed,
                                                unsigned long long start) {
    __shared__ uint32_t cum[TEXT_SYMS];
    __shared__ uint8_t order[TEXT_SYMS];
    if (threadIdx.x == 0) {
        const char lower[] = " etaoinshrdlcumwfgypbvkjxqz";
        bool used[256];
        for (int i = 0; i < 256; i++) used[i] = false;
        int m = 0;
        for (int i = 0; lower[i]; i++) { order[m++] = (uint8_t)lower[i]; used[(uint8_t)lower[i]] = true; }
        for (int i = 1; lower[i]; i++) { const uint8_t u = (uint8_t)(lower[i] - 32); order[m++] = u; used[u] = true; }
        for (int c = '0'; c <= '9'; c++) { order[m++] = (uint8_t)c; used[c] = true; }
        for (int c = 33; c < 127; c++) if (!used[c]) order[m++] = (uint8_t)c;
        order[m++] = (uint8_t)'\n';
        uint32_t acc = 0;
        for (int r = 0; r < TEXT_SYMS; r++) {  // Zipf head, geometric tail of rare symbols (long codes)
            const uint32_t w = (r < 64) ? 1000000u / (uint32_t)(r + 1) : (15625u >> ((r - 62) / 2));
            acc += w ? w : 1u;
            cum[r] = acc;
        }
    }
    __syncthreads();
    const unsigned long long total = cum[TEXT_SYMS - 1];
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long x = splitmix64(seed + (start + i) * 0x9E3779B97F4A7C15ULL);
        const uint32_t t = (uint32_t)(((x >> 32) * total) >> 32);
        int lo = 0, hi = TEXT_SYMS - 1;  // smallest r with cum[r] > t
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] > t) hi = mid; else lo = mid + 1;
        }
        d[i] = order[lo];
    }
}

__global__ __launch_bounds__(256) void gen_lowentropy(uint8_t* __restrict__ d, size_t n, unsigned long long seed,
                                                      unsigned long long start) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long x = splitmix64(seed + (start + i) * 0x9E3779B97F4A7C15ULL);
        const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
        d[i] = (hi % 100u == 0u) ? (uint8_t)(1u + lo % 255u) : (uint8_t)0;
    }
}

static uint32_t stride_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

void launch_fill_java_random(uint8_t* d, size_t n, int64_t seed, uint64_t start, hipStream_t s) {
    if (n == 0) return;
    const size_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(gen_java_random, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, d, n, (long long)seed,
                       (unsigned long long)start);
}

void launch_fill_text(uint8_t* d, size_t n, uint64_t seed, uint64_t start, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(gen_text, dim3(stride_grid(n)), dim3(256), 0, s, d, n, (unsigned long long)seed,
                       (unsigned long long)start);
}

void launch_fill_lowentropy(uint8_t* d, size_t n, uint64_t seed, uint64_t start, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(gen_lowentropy, dim3(stride_grid(n)), dim3(256), 0, s, d, n, (unsigned long long)seed,
                       (unsigned long long)start);
}

}  // namespace dcz
