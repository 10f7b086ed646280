// k2_codebuild.hip -- K2: per-block canonical Huffman code build on the device (gfx950).
//
// Replaces CanonicalHuffman.buildCanonicalCodes (core/CanonicalHuffman.java:19-50):
//   buildCodeLengths  core/CanonicalHuffman.java:55-80   (java.util.PriorityQueue of HuffmanNode)
//   extractLengths    core/CanonicalHuffman.java:85-92
//   generateCanonicalCodes core/CanonicalHuffman.java:99-132
// and the host-side prefix sum of GpuCompressionService.executePacketEncoding
// (service/gpu/GpuCompressionService.java:771-779: int[N] bit positions built serially on the CPU) --
// here only one bit offset per 32 KiB segment is needed, and it comes from the segment histograms:
//   bits(segment) = sum_s seg_hist[segment][s] * len[s].
//
// One workgroup per block.  The PriorityQueue is emulated literally (array binary heap, OpenJDK
// siftUp/siftDown tie rules) by ONE lane because the code lengths depend on the heap's tie order when
// equal-weight internal nodes coexist (core/HuffmanNode.java:52-58 compares them equal).  Heap
// entries are packed u64 = weight<<18 | (symbol+1)<<9 | node id; entries compare by (entry >> 9), so
// internal nodes (symbol -1 -> 0) sort before leaves of equal weight and two internal nodes of equal
// weight compare EQUAL, exactly as compareTo does.  The heap is stored shifted by one slot so that
// the two children of slot k sit in one aligned 16-byte pair (a single ds_read_b128 per level).
// This kernel is latency-bound (a few hundred dependent heap operations per block), so it runs ONE WAVE per
// block: 20+ blocks are resident per CU and the serial heap walks of different blocks overlap.
#include "dcz_internal.h"

namespace dcz {

constexpr int K2_T = 64;  // threads per block-workgroup (one wave)

// 3.6 KiB per block-wave: all 32 wave slots of a CU can hold a block (the block histogram lives in registers,
// lane t owns symbols 4t..4t+3, and is only staged inside the heap array for the serial build).
struct CodeLds {
    __attribute__((aligned(16))) unsigned long long heap[260];  // slot i+1 holds PriorityQueue.queue[i]
    uint32_t cnt[34];
    uint32_t first[34];
    uint16_t parent[512];
    uint8_t len[256];
    int nsym;
    int maxlen;
};

#define HKEY(e) ((e) >> 9)

// PriorityQueue.offer -> siftUp: ties do not move up.
__device__ __forceinline__ void heap_offer(unsigned long long* q, int& size, unsigned long long x) {
    int k = size++;
    const unsigned long long xk = HKEY(x);
    while (k > 0) {
        const int parent = (k - 1) >> 1;
        const unsigned long long e = q[parent + 1];
        if (xk >= HKEY(e)) break;
        q[k + 1] = e;
        k = parent;
    }
    q[k + 1] = x;
}

// PriorityQueue.poll -> siftDown: ties prefer the left child; x stops when x <= child.
// (Measured with DCZ_K2_PROF / tools/k2prof.py: the serial build is 90-97 % of a block's K2 time, ~250 cycles per heap
// level.  Fetching children and grandchildren together -- two levels per LDS round trip -- changed nothing: a level's cost
// is its chain of ~30 dependent 64-bit vector/scalar instructions and branches on a SIMD that holds one wave, not the read.)
__device__ __forceinline__ unsigned long long heap_poll(unsigned long long* q, int& size) {
    const unsigned long long result = q[1];
    const int n = --size;
    const unsigned long long x = q[n + 1];
    if (n > 0) {
        const unsigned long long xk = HKEY(x);
        int k = 0;
        const int half = n >> 1;
        while (k < half) {
            int child = 2 * k + 1;
            const ulonglong2 pr = *reinterpret_cast<const ulonglong2*>(&q[child + 1]);  // slots 2k+2, 2k+3
            unsigned long long c = pr.x;
            if (child + 1 < n && HKEY(c) > HKEY(pr.y)) {
                c = pr.y;
                child = child + 1;
            }
            if (xk <= HKEY(c)) break;
            q[k + 1] = c;
            k = child;
        }
        q[k + 1] = x;
    }
    return result;
}

// Code lengths for the block histogram hf[i] = count of symbol 4*tid+i -> L.len, L.maxlen, L.nsym.
__device__ void build_lengths(CodeLds& L, const unsigned long long (&hf)[4]) {
    const int tid = (int)threadIdx.x;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        mine += (hf[i] > 0) ? 1 : 0;
        L.len[4 * tid + i] = 0;
        // Stage the weights for the serial build at heap slot s+2: offering symbol s reads slot s+2 and writes only
        // slots <= s+1 (the heap holds at most s entries before it), so the staging is consumed before it is reused.
        L.heap[4 * tid + i + 2] = hf[i];
    }
    const int nsym = (int)wave_reduce_add_u32((uint32_t)mine);
    if (tid == 0) {
        L.nsym = nsym;
        L.maxlen = 0;
    }
    __syncthreads();
    if (nsym == 0) return;  // core/CanonicalHuffman.java:30-32
    if (nsym == 1) {        // core/CanonicalHuffman.java:35-45: the single symbol gets length 1, code 0
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (hf[i] > 0) {
                L.len[4 * tid + i] = 1;
                L.maxlen = 1;
            }
        __syncthreads();
        return;
    }
    // Balanced case, decided without the queue: 2^k symbols whose largest count is below twice the smallest.  Then the
    // two lightest leaves together outweigh every leaf, so buildCodeLengths (core/CanonicalHuffman.java:66-70) pairs up all
    // leaves before it polls an internal node, the 2^(k-1) internal nodes again satisfy max < 2 * min, and by induction the
    // tree is complete: every symbol gets length k whatever order equal weights are polled in.  This is the reference's
    // own high-entropy case (256 symbols, all lengths 8) and skips the ~1000 dependent heap operations below.
    {
        unsigned long long mn = ~0ull, mx = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (hf[i] > 0 && hf[i] < mn) mn = hf[i];
            if (hf[i] > mx) mx = hf[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = __shfl_xor(mn, o, 64), c = __shfl_xor(mx, o, 64);
            mn = a < mn ? a : mn;
            mx = c > mx ? c : mx;
        }
        if ((nsym & (nsym - 1)) == 0 && mx < 2ull * mn) {  // wave-uniform
            const int k = __builtin_ctz((unsigned)nsym);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (hf[i] > 0) L.len[4 * tid + i] = (uint8_t)k;
            if (tid == 0) L.maxlen = k;
            __syncthreads();
            return;
        }
    }
    if (tid == 0) {
        int size = 0;
        for (int s = 0; s < 256; s++) {  // core/CanonicalHuffman.java:59-63: leaves in symbol order
            const unsigned long long w = L.heap[s + 2];
            if (w > 0) heap_offer(L.heap, size, (w << 18) | ((unsigned long long)(s + 1) << 9) | (unsigned long long)s);
        }
        int nn = 256;
        while (size > 1) {  // core/CanonicalHuffman.java:66-70
            const unsigned long long l = heap_poll(L.heap, size);
            const unsigned long long r = heap_poll(L.heap, size);
            L.parent[l & 511u] = (uint16_t)nn;
            L.parent[r & 511u] = (uint16_t)nn;
            const unsigned long long w = (l >> 18) + (r >> 18);
            heap_offer(L.heap, size, (w << 18) | (unsigned long long)nn);
            nn++;
        }
        L.parent[L.heap[1] & 511u] = 0xFFFFu;
    }
    __syncthreads();
    // core/CanonicalHuffman.java:85-92 extractLengths: depth of each leaf
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (hf[i] > 0) {
            int d = 0, x = 4 * tid + i;
            while (true) {
                const int p = L.parent[x];
                if (p == 0xFFFF) break;
                x = p;
                d++;
            }
            L.len[4 * tid + i] = (uint8_t)(d > 255 ? 255 : d);
            atomicMax(&L.maxlen, d);
        }
    }
    __syncthreads();
}

// core/CanonicalHuffman.java:99-132 generateCanonicalCodes, from L.len (all <= 32) -> out[256] (global memory).
__device__ void canonical_codes(CodeLds& L, uint32_t* __restrict__ out, bool zero_all) {
    const int tid = (int)threadIdx.x;
    if (tid < 34) L.cnt[tid] = 0;
    __syncthreads();
    for (int s = tid; s < 256; s += K2_T) {
        const int l = L.len[s];
        if (l > 0) atomicAdd(&L.cnt[l], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0;
        L.first[0] = 0;
        for (int i = 1; i <= 32; i++) {  // core/CanonicalHuffman.java:112-117
            c = (c + L.cnt[i - 1]) << 1;
            L.first[i] = c;
        }
    }
    __syncthreads();
    // core/CanonicalHuffman.java:123-129: ascending symbol order within a length.  Symbols are visited in chunks of
    // K2_T consecutive symbols; a ballot per distinct length ranks the lanes of a chunk, L.first[] carries over.
    for (int base = 0; base < 256; base += K2_T) {
        const int sym = base + tid;
        const uint32_t l = L.len[sym];
        uint32_t code = 0;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(l > 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t ll = (uint32_t)__builtin_amdgcn_readlane((int)l, leader);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(l == ll);
            if (l == ll) code = L.first[ll] + (uint32_t)__builtin_popcountll(same & ((1ull << tid) - 1ull));
            __syncthreads();
            if (tid == leader) L.first[ll] += (uint32_t)__builtin_popcountll(same);
            __syncthreads();
            todo &= ~same;
        }
        out[sym] = zero_all ? 0u : code;
    }
    __syncthreads();
}

#if DCZ_K2_PROF
__device__ unsigned long long k2_prof[8];  // cycles of wave 0's lane 0 per phase, summed over blocks; [7] = blocks
#define K2_T_(i)                                           \
    do {                                                   \
        const unsigned long long t_ = clock64();           \
        if (tid == 0) atomicAdd(&k2_prof[i], t_ - plast);  \
        plast = t_;                                        \
    } while (0)
#else
#define K2_T_(i) do { } while (0)
#endif
__global__ __launch_bounds__(K2_T) void k2_codebuild(const uint16_t* __restrict__ seg_hist,
                                                     const long long* __restrict__ hist_in, size_t n,
                                                     size_t block_bytes, uint32_t spb, uint32_t K,
                                                     uint8_t* __restrict__ d_len, uint32_t* __restrict__ d_code,
                                                     uint8_t* __restrict__ d_maxlen, uint32_t* __restrict__ d_comp_size,
                                                     unsigned long long* __restrict__ d_seg_bitoff,
                                                     int32_t* __restrict__ d_status, uint32_t* __restrict__ hint,
                                                     uint32_t epoch) {
    __shared__ CodeLds L;
    const uint32_t b = blockIdx.x;
    const int tid = (int)threadIdx.x;

#if DCZ_K2_PROF
    unsigned long long plast = clock64();
    if (tid == 0) atomicAdd(&k2_prof[7], 1ull);
#endif
    // number of segments this block really has (the last block may be short)
    uint32_t nsb = 0;
    if (seg_hist) {
        const uint64_t bstart = (uint64_t)b * block_bytes;
        const uint64_t bend = (bstart + block_bytes < n) ? bstart + block_bytes : (uint64_t)n;
        nsb = (uint32_t)((bend - bstart + SEG - 1) / SEG);
    }
    const uint16_t* rows = seg_hist ? seg_hist + (uint64_t)b * spb * 256u : nullptr;

    // block histogram = sum of its segment rows (lane t owns bins 4t..4t+3: one 8-byte load per row)
    unsigned long long hf[4] = {0, 0, 0, 0};
    if (seg_hist) {
#pragma unroll 16
        for (uint32_t j = 0; j < nsb; j++) {  // (unrolled: 16 independent loads in flight; a 32 MiB block has 1024 rows)
            const uint2 v = *reinterpret_cast<const uint2*>(rows + (uint64_t)j * 256u + 4u * (uint32_t)tid);
            hf[0] += v.x & 0xFFFFu;
            hf[1] += v.x >> 16;
            hf[2] += v.y & 0xFFFFu;
            hf[3] += v.y >> 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) hf[i] = (unsigned long long)hist_in[(uint64_t)b * 256u + 4 * tid + i];
    }

    K2_T_(0);
    build_lengths(L, hf);
    K2_T_(1);
    const int maxlen = L.maxlen;
    const bool too_long = maxlen > 32;  // core/CanonicalHuffman.java:102-106 would throw
    if (too_long) {
#pragma unroll
        for (int i = 0; i < 4; i++) L.len[4 * tid + i] = 0;
    }
    __syncthreads();
    canonical_codes(L, d_code + (uint64_t)b * 256u, false);
    K2_T_(2);
    unsigned long long bits = 0;
    {
        uint32_t l4 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t l = L.len[4 * tid + i];
            l4 |= l << (8 * i);
            bits += hf[i] * (unsigned long long)l;
        }
        reinterpret_cast<uint32_t*>(d_len + (uint64_t)b * 256u)[tid] = l4;
    }
    bits = wave_reduce_add_u64(bits);
    if (tid == 0) {
        // bit 7: 256 symbols of 8 bits, i.e. codeword(s) = s and the payload is a copy of the input (K3 copies it)
        const bool identity = !too_long && maxlen == 8 && L.nsym == 256;
        d_maxlen[b] = (uint8_t)(too_long ? 0 : (maxlen | (identity ? 0x80 : 0)));
        if (hint) hint[identity ? 0 : 1] = epoch;  // (ShapeHint: calls like this one have / do not only have identity blocks)
        d_comp_size[b] = too_long ? 0u : (uint32_t)((bits + 7) >> 3);
        d_status[b] = too_long ? DCZ_E_CODELEN : DCZ_OK;
    }

    K2_T_(3);
    // per-segment bit offsets inside the block: exclusive scan of bits(segment) = sum_s seg_hist[seg][s] * len[s]
    if (seg_hist && d_seg_bitoff) {
        unsigned long long carry = 0;
        for (uint32_t c0 = 0; c0 < spb; c0 += K2_T) {
            const uint32_t j = c0 + (uint32_t)tid;
            unsigned long long sb = 0;
            if (j < nsb) {
                const uint16_t* r = rows + (uint64_t)j * 256u;
                uint32_t acc = 0;
                for (int s = 0; s < 256; s += 2) {
                    const uint32_t pr = *reinterpret_cast<const uint32_t*>(r + s);
                    acc += (pr & 0xFFFFu) * L.len[s] + (pr >> 16) * L.len[s + 1];
                }
                sb = acc;
            }
            unsigned long long inc = sb;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long t = __shfl_up(inc, o, 64);
                if (tid >= o) inc += t;
            }
            const unsigned long long tot = __shfl(inc, 63, 64);
            if (j < spb) d_seg_bitoff[(uint64_t)b * spb + j] = carry + inc - sb;
            carry += tot;
        }
    }
    K2_T_(4);
    (void)K;
}

// CanonicalHuffman.generateCanonicalCodesFromLengths (core/CanonicalHuffman.java:141-146) for one table.
__global__ __launch_bounds__(K2_T) void k2_codes_from_lengths(const int32_t* __restrict__ len32,
                                                              uint32_t* __restrict__ d_code,
                                                              int32_t* __restrict__ d_status) {
    __shared__ CodeLds L;
    const int tid = (int)threadIdx.x;
    bool mybad = false;
    for (int s = tid; s < 256; s += K2_T) {
        const int32_t l = len32[s];
        mybad |= (l < 0 || l > 32);  // core/CanonicalHuffman.java:106 would throw
    }
    const bool bad = __builtin_amdgcn_ballot_w64(mybad) != 0ull;
    for (int s = tid; s < 256; s += K2_T) L.len[s] = bad ? 0 : (uint8_t)len32[s];
    __syncthreads();
    canonical_codes(L, d_code, bad);
    if (tid == 0) d_status[0] = bad ? DCZ_E_BADTABLE : DCZ_OK;
}

// Payload offsets: exclusive scan of comp_size over the K blocks of this call, total, capacity check.
__global__ __launch_bounds__(1024) void k2_offsets(const uint32_t* __restrict__ comp_size, uint32_t K,
                                                   unsigned long long* __restrict__ comp_off,
                                                   unsigned long long* __restrict__ d_total,
                                                   const unsigned long long* __restrict__ carry_in,
                                                   unsigned long long out_cap, int32_t* __restrict__ d_status) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry_s = carry_in ? *carry_in : 0ull;  // payload bytes of the blocks before this range
    __syncthreads();
    for (uint32_t c0 = 0; c0 < K; c0 += 1024) {
        const uint32_t k = c0 + (uint32_t)tid;
        const unsigned long long v = (k < K) ? comp_size[k] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if ((tid & 63) >= o) inc += t;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < (tid >> 6)) wbase += wsum[w];
            tot += wsum[w];
        }
        const unsigned long long off = carry_s + wbase + inc - v;
        if (k < K) {
            comp_off[k] = off;
            if (off + v > out_cap && d_status[k] == DCZ_OK) d_status[k] = DCZ_E_CAPACITY;
        }
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0 && d_total) *d_total = carry_s;
}

void launch_codebuild(const uint16_t* seg_hist, const int64_t* d_hist_in, size_t n, size_t block_bytes,
                      uint32_t segs_per_block, uint32_t K, uint8_t* d_len, uint32_t* d_code, uint8_t* d_maxlen,
                      uint32_t* d_comp_size, uint64_t* d_seg_bitoff, int32_t* d_status, hipStream_t s,
                      const ShapeHint& hint) {
    if (K == 0) return;
    hipLaunchKernelGGL(k2_codebuild, dim3(K), dim3(K2_T), 0, s, seg_hist, reinterpret_cast<const long long*>(d_hist_in),
                       n, block_bytes, segs_per_block, K, d_len, d_code, d_maxlen, d_comp_size,
                       reinterpret_cast<unsigned long long*>(d_seg_bitoff), d_status, hint.dev, hint.epoch);
}

void launch_codes_from_lengths(const int32_t* d_len32, uint32_t* d_code, int32_t* d_status, hipStream_t s) {
    hipLaunchKernelGGL(k2_codes_from_lengths, dim3(1), dim3(K2_T), 0, s, d_len32, d_code, d_status);
}

void launch_offsets(const uint32_t* d_comp_size, uint32_t K, uint64_t* d_comp_off, uint64_t* d_total,
                    const uint64_t* d_carry_in, size_t out_cap, int32_t* d_status, hipStream_t s) {
    hipLaunchKernelGGL(k2_offsets, dim3(1), dim3(1024), 0, s, d_comp_size, K,
                       reinterpret_cast<unsigned long long*>(d_comp_off),
                       reinterpret_cast<unsigned long long*>(d_total),
                       reinterpret_cast<const unsigned long long*>(d_carry_in), (unsigned long long)out_cap, d_status);
}

}  // namespace dcz

#if DCZ_K2_PROF
extern "C" void dcz_debug_k2_prof(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dcz::k2_prof), sizeof(dcz::k2_prof));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(dcz::k2_prof), z, sizeof(z));
    }
}
#endif
