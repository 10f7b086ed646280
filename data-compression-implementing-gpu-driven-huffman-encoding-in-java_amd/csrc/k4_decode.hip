// k4_decode.hip -- K4: parallel table-walk Huffman decode without side information (gfx950).
//
// Replaces TableBasedHuffmanDecoder (core/TableBasedHuffmanDecoder.java:36-152: 10-bit table, bit-serial
// peek(), HashMap fallback for long codes core/CanonicalHuffman.java:161-229) as called from
// CpuCompressionService.decodeChunkParallel (service/cpu/CpuCompressionService.java:511-556).  The
// reference's GPU decode kernels run ONE work-item per chunk and are dead code
// (service/gpu/GpuCompressionService.java:1340-1469); GpuCompressionService.decompress delegates to the CPU (:858).
//
// The frozen format stores no intra-block offsets, so parallelism inside a
// block comes from self-synchronisation: the workgroup walks its block in windows of W subsequences
// of 32 bytes.  Per window:
//   A. every thread decodes its subsequence from a guessed entry bit and reports where the codeword
//      that crosses its end finishes; guesses are replaced by the left neighbour's exit until nothing
//      changes (the first subsequence's entry is exact, so the fixed point is the true parse; Huffman
//      streams resynchronise within a few codewords, typically 2-3 rounds);
//   B. a workgroup scan of the symbol counts gives every thread its output offset; threads decode
//      again, writing bytes into an LDS staging tile that is flushed with coalesced 16-byte stores.
// The compressed window is staged once in LDS (16-byte aligned global loads, byte-swapped so that a
// 64-bit window read gives MSB-first bits), so HBM traffic is the algorithmic C + N per block.
// Decode table: 2^11 entries (len<<8 | symbol) in LDS; longer codes take the canonical
// first-code/count search, which for a prefix code returns exactly what the reference's
// 10-bit-table-then-HashMap path returns.  Bits past the end of the payload read as zero
// (TableBasedHuffmanDecoder.java:204-208); a missing code is "Huffman decode error at position i" (:109-111).
#include "dcz_internal.h"

namespace dcz {

constexpr int SUB_BYTES = 32;
constexpr int SUB_BITS = SUB_BYTES * 8;
constexpr int TB = 11;
constexpr int OC = 16384;  // output staging bytes per flush

template <int W>
struct DecLds {
    __attribute__((aligned(16))) uint32_t cbuf[W * (SUB_BYTES / 4) + 8];
    __attribute__((aligned(16))) uint8_t outbuf[OC + 16];
    uint16_t table[1 << TB];
    uint16_t exits[W];
    uint32_t first[34];
    uint32_t cnt[34];
    uint32_t offs[34];
    uint32_t wsum[W / 64];
    uint8_t symtab[256];
    uint8_t len8[256];
    uint32_t maxlen;
    uint32_t err_idx;
    int bad_table;
};

struct Sym {
    uint32_t sym;
    uint32_t len;  // 0 = no codeword matches
};

template <int W>
__device__ __forceinline__ Sym dec_lookup(const DecLds<W>& L, uint32_t pos) {
    const uint32_t wi = pos >> 5, sh = pos & 31u;
    const unsigned long long two = ((unsigned long long)L.cbuf[wi] << 32) | (unsigned long long)L.cbuf[wi + 1];
    const uint32_t win = (uint32_t)((two << sh) >> 32);
    const uint32_t e = L.table[win >> (32 - TB)];
    Sym r;
    if (e != 0) {
        r.sym = e & 0xFFu;
        r.len = e >> 8;
        return r;
    }
    r.sym = 0;
    r.len = 0;
    const uint32_t maxlen = L.maxlen;
    for (uint32_t l = TB + 1; l <= maxlen; l++) {
        const uint32_t c = win >> (32u - l);
        const uint32_t f = L.first[l];
        if (c >= f && c - f < L.cnt[l]) {
            r.sym = L.symtab[L.offs[l] + (c - f)];
            r.len = l;
            break;
        }
    }
    return r;
}

template <int W>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, DecLds<W>& L, uint32_t& total) {
    const uint32_t inc = wave_inclusive_scan_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) L.wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < W / 64; w++) {
        const uint32_t s = L.wsum[w];
        if (w < (int)(threadIdx.x >> 6)) base += s;
        tot += s;
    }
    total = tot;
    return base + inc - v;
}

template <int W>
__global__ __launch_bounds__(W) void k4_decode(const uint8_t* __restrict__ comp, size_t comp_bytes,
                                               const unsigned long long* __restrict__ d_comp_off,
                                               const uint32_t* __restrict__ d_comp_size,
                                               const uint32_t* __restrict__ d_orig_size,
                                               const uint8_t* __restrict__ d_len, size_t out_stride,
                                               uint8_t* __restrict__ out, int32_t* __restrict__ d_status,
                                               long long* __restrict__ d_errpos) {
    __shared__ DecLds<W> L;
    const uint32_t b = blockIdx.x;
    const int tid = (int)threadIdx.x;

    // ---- per-block tables (rebuildCodes: CpuCompressionService.java:582-586 -> CanonicalHuffman.java:99-132) ----
    if (tid < 34) L.cnt[tid] = 0;
    if (tid == 0) {
        L.bad_table = 0;
        L.err_idx = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t mylen = 0;
    if (tid < 256) {
        mylen = d_len[(uint64_t)b * 256u + tid];
        L.len8[tid] = (uint8_t)mylen;
        if (mylen > 32) L.bad_table = 1;
        else if (mylen > 0) atomicAdd(&L.cnt[mylen], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0, o = 0, mx = 0;
        unsigned long long kraft = 0;
        L.first[0] = 0;
        L.offs[0] = 0;
        for (int l = 1; l <= 32; l++) {
            c = (c + L.cnt[l - 1]) << 1;
            L.first[l] = c;
            L.offs[l] = o;
            o += L.cnt[l];
            if (L.cnt[l]) mx = (uint32_t)l;
            kraft += (unsigned long long)L.cnt[l] << (32 - l);
        }
        L.maxlen = mx;
        if (kraft > (1ull << 32)) L.bad_table = 1;  // not a prefix code
    }
    __syncthreads();
    if (L.bad_table) {
        if (tid == 0) {
            d_status[b] = DCZ_E_BADTABLE;
            if (d_errpos) d_errpos[b] = 0;
        }
        return;
    }
    if (tid < 256 && mylen > 0) {
        uint32_t rank = 0;
        for (int s = 0; s < tid; s++) rank += (L.len8[s] == mylen) ? 1u : 0u;
        L.symtab[L.offs[mylen] + rank] = (uint8_t)tid;
    }
    __syncthreads();
    for (int idx = tid; idx < (1 << TB); idx += W) {
        uint32_t e = 0;
        for (uint32_t l = 1; l <= TB; l++) {
            const uint32_t c = (uint32_t)idx >> (TB - l);
            const uint32_t f = L.first[l];
            if (c >= f && c - f < L.cnt[l]) {
                e = (l << 8) | L.symtab[L.offs[l] + (c - f)];
                break;
            }
        }
        L.table[idx] = (uint16_t)e;
    }
    __syncthreads();

    // ---- block geometry ----
    const uint32_t orig = d_orig_size[b];
    const unsigned long long coff = d_comp_off[b];
    const uint32_t csize = d_comp_size[b];
    uint8_t* const oblk = out + (uint64_t)b * out_stride;
    const bool out_aligned = (((uintptr_t)oblk) & 15u) == 0u;
    // virtual byte 0 = 16-byte aligned address at or below the payload start
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)coff;
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew;                               // first valid virtual byte
    const unsigned long long vhi = (unsigned long long)skew + csize;   // one past the last valid virtual byte
    (void)comp_bytes;

    unsigned long long ventry = 8ull * skew;  // virtual bit of the next codeword boundary
    uint32_t produced = 0;
    int status = DCZ_OK;
    long long errpos = 0;

    while (produced < orig) {
        const unsigned long long wchunk0 = ventry >> 7;
        const uint32_t g0 = (uint32_t)(ventry - (wchunk0 << 7));

        // stage the window (+16 bytes of look-ahead), zero outside the payload, MSB-first dwords
        for (int c = tid; c < 2 * W + 1; c += W) {
            const unsigned long long vb = (wchunk0 + (unsigned long long)c) << 4;  // virtual byte of this chunk
            uint4 v = make_uint4(0, 0, 0, 0);
            if (vb + 16 > vlo && vb < vhi) {
                v = *reinterpret_cast<const uint4*>(vbase + vb);
                if (vb < vlo || vb + 16 > vhi) {
                    uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        uint32_t m = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const unsigned long long bb = vb + 4 * d + k;
                            if (bb >= vlo && bb < vhi) m |= 0xFFu << (8 * k);
                        }
                        wds[d] &= m;
                    }
                    v = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                }
            }
            uint4 sw;
            sw.x = bswap32(v.x);
            sw.y = bswap32(v.y);
            sw.z = bswap32(v.z);
            sw.w = bswap32(v.w);
            *reinterpret_cast<uint4*>(&L.cbuf[c * 4]) = sw;
        }
        __syncthreads();

        // ---- phase A: self-synchronisation ----
        const uint32_t limit = (uint32_t)(tid + 1) * SUB_BITS;
        uint32_t g = (tid == 0) ? g0 : 0u;
        uint32_t x = 0, nsym = 0;
        bool bad = false;
        bool need = true;
        while (true) {
            if (need) {
                uint32_t pos = (uint32_t)tid * SUB_BITS + g;
                nsym = 0;
                bad = false;
                while (pos < limit) {
                    const Sym s = dec_lookup<W>(L, pos);
                    if (s.len == 0) {
                        bad = true;
                        break;
                    }
                    pos += s.len;
                    nsym++;
                }
                x = bad ? 0u : pos - limit;
            }
            L.exits[tid] = (uint16_t)x;
            __syncthreads();
            const uint32_t ng = (tid == 0) ? g0 : (uint32_t)L.exits[tid - 1];
            need = (ng != g);
            g = ng;
            if (!__syncthreads_or(need)) break;
        }

        // ---- offsets, errors ----
        uint32_t tw = 0;
        const uint32_t o = block_exclusive_scan<W>(nsym, L, tw);
        const uint32_t remaining = orig - produced;
        if (bad) atomicMin(&L.err_idx, o + nsym);
        __syncthreads();
        const uint32_t err_idx = L.err_idx;
        if (err_idx < remaining) {
            status = DCZ_E_BADSTREAM;
            errpos = (long long)produced + (long long)err_idx;
            break;
        }
        const uint32_t lim = (tw < remaining) ? tw : remaining;

        // ---- phase B: decode into the staging tile, flush coalesced ----
        uint32_t k = 0;
        uint32_t bpos = (uint32_t)tid * SUB_BITS + g;
        for (uint32_t cb = 0; cb < lim;) {
            const uint32_t gstart = produced + cb;
            const uint32_t a = gstart & 15u;
            uint32_t cc = lim - cb;
            if (cc > (uint32_t)OC - a) cc = (uint32_t)OC - a;
            const uint32_t cend = cb + cc;
            while (k < nsym && o + k < cend) {
                const Sym s = dec_lookup<W>(L, bpos);
                L.outbuf[a + (o + k - cb)] = (uint8_t)s.sym;
                bpos += s.len;
                k++;
            }
            __syncthreads();
            uint8_t* const dst = oblk + (gstart - a);
            const uint32_t nunits = (a + cc + 15u) >> 4;
            for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                const uint32_t lo = u << 4;
                if (out_aligned && lo >= a && lo + 16u <= a + cc) {
                    *reinterpret_cast<uint4*>(dst + lo) = *reinterpret_cast<const uint4*>(&L.outbuf[lo]);
                } else {
                    for (uint32_t q = 0; q < 16u; q++) {
                        const uint32_t i = lo + q;
                        if (i >= a && i < a + cc) dst[i] = L.outbuf[i];
                    }
                }
            }
            __syncthreads();
            cb = cend;
        }
        produced += lim;
        ventry = (wchunk0 << 7) + (unsigned long long)W * SUB_BITS + (unsigned long long)L.exits[W - 1];
        __syncthreads();
    }

    if (tid == 0) {
        d_status[b] = status;
        if (d_errpos) d_errpos[b] = errpos;
    }
}

void launch_decode(const uint8_t* d_comp, size_t comp_bytes, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                   const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                   int32_t* d_status, int64_t* d_errpos, hipStream_t s) {
    if (K == 0) return;
    const unsigned long long* off = reinterpret_cast<const unsigned long long*>(d_comp_off);
    long long* ep = reinterpret_cast<long long*>(d_errpos);
    if (K >= 1024) {
        hipLaunchKernelGGL(k4_decode<256>, dim3(K), dim3(256), 0, s, d_comp, comp_bytes, off, d_comp_size, d_orig_size,
                           d_len, out_stride, d_out, d_status, ep);
    } else {
        hipLaunchKernelGGL(k4_decode<1024>, dim3(K), dim3(1024), 0, s, d_comp, comp_bytes, off, d_comp_size,
                           d_orig_size, d_len, out_stride, d_out, d_status, ep);
    }
}

}  // namespace dcz
