// k3_encode.hip -- K3: codeword lookup + variable-length MSB-first bit packing (gfx950).
//
// Replaces CpuCompressionService.encodeChunk + BitOutputStream (service/cpu/CpuCompressionService.java:303-315,
// :711-737: one bit per loop iteration) and the TornadoVM path GpuCompressionService.executePacketEncoding +
// TornadoKernels.encodePacketKernel (service/gpu/GpuCompressionService.java:758-831, service/gpu/TornadoKernels.java:115-205:
// one work-item per OUTPUT word, binary search over an int[N] bit-position array built on the host).
//
// Output format (frozen): symbols in input order, codeword MSB first, stream bit i
// lives in byte i>>3 at bit 7-(i&7), last byte zero padded; blocks are byte aligned and concatenated.
//
// Design.  One WAVE owns one 32 KiB segment of a block; its first bit offset comes from K2
// (d_seg_bitoff), so waves never wait for each other and there is no inter-workgroup traffic.
//   * input: 16 B/lane coalesced loads (1 KiB per wave instruction), next chunk prefetched;
//   * codebook: LDS, 256 entries of code<<16|len (code<<6|len when maxlen>16), replicated K3_COPIES times across the banks (entry s of
//     replica r at dword s*COPIES+r, lane l reads replica l % COPIES) so that lookups of equal or different
//     symbols by different lanes rarely collide, whatever the data;
//   * each lane concatenates G consecutive codewords in registers (G=8 when maxlen<=8, G=4 when maxlen<=16,
//     G=2 when maxlen<=26; a wide path with 64-bit entries covers maxlen<=32), a DPP wave scan of the lane
//     bit counts gives every lane its bit offset (no LDS, no ballot needed);
//   * lanes OR their strings into a per-wave 4 KiB LDS ring (ds_or_b32), indexed by stream position
//     relative to a 16-byte-aligned origin of the OUTPUT address, so that complete 16-byte chunks
//     are byte-swapped and stored with one coalesced global_store_dwordx4 per lane;
//   * a byte belongs to the segment that holds its first bit: the owner completes its last byte
//     by encoding up to 7 look-ahead symbols of the next segment, and never writes the byte its
//     first bits fall into.  Every output byte has exactly one writer: no global atomics.
#include "dcz_internal.h"

namespace dcz {

#ifndef DCZ_K3_WAVES
#define DCZ_K3_WAVES 4
#endif
#ifndef DCZ_K3_COPIES
#define DCZ_K3_COPIES 4
#endif
#ifndef DCZ_K3_LAZY_FLUSH
#define DCZ_K3_LAZY_FLUSH 1  // flush the ring only when the next step might not fit (0: after every step)
#endif
constexpr int K3_WAVES = DCZ_K3_WAVES;    // waves (= segments) per workgroup
constexpr int K3_COPIES = DCZ_K3_COPIES;  // codebook replicas: 32 would be conflict-free; 8 (a few 2-way conflicts) leaves room
                                          // for 6 workgroups per CU and measures ~10 % faster; 4 for 7 (what the registers
                                          // allow): text 1.75 -> 1.68 ms per 4 GiB half, zeros + noise 1.18 -> 1.11; 16: 2.06 / 1.43
constexpr int K3_CSHIFT = (K3_COPIES == 32) ? 5 : (K3_COPIES == 16) ? 4 : (K3_COPIES == 8) ? 3 : 2;
constexpr int RING_WORDS = 1024;     // 4 KiB per wave = 32768 bits
constexpr uint32_t RING_MASK = RING_WORDS - 1;
constexpr int RING_STRIDE = RING_WORDS + 4;  // + 2 slack dwords (a string that starts in the last dwords spills into
                                             // them and is folded back to dwords 0/1 by the next flush) + 2 pad

struct EncState {
    uint32_t* ring;   // per-wave LDS ring (big-endian bit order inside each dword)
    uint8_t* gbase;   // global address of relative byte 0 (16-byte aligned)
    uint32_t rpos;    // next free bit, relative to the origin
    uint32_t rflush;  // next byte to store, relative to the origin
};

// OR the low `l` bits of g (0 <= l <= 64) into the ring at relative bit position p.  The string is left-aligned in 64
// bits and spread over three consecutive dwords with funnel shifts; the dwords are OR-ed unconditionally (a zero does
// nothing), their addresses are the first one plus immediates -- the ring has two slack dwords past its end.
__device__ __forceinline__ void ring_or(uint32_t* ring, uint32_t p, unsigned long long g, uint32_t l) {
    const unsigned long long ga = l ? (g << (64u - l)) : 0ull;
    const uint32_t hi = (uint32_t)(ga >> 32), lo = (uint32_t)ga;
    const uint32_t sh = p;  // v_alignbit uses the low 5 bits
    const uint32_t x0 = hi >> (sh & 31u);
    const uint32_t x1 = __builtin_amdgcn_alignbit(hi, lo, sh);
    const uint32_t x2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
    uint32_t* q = ring + ((p >> 5) & RING_MASK);
    __hip_atomic_fetch_or(q, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(q + 1, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(q + 2, x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ uint32_t ring_byte(const uint32_t* ring, uint32_t rb) {
    const uint32_t w = ring[(rb >> 2) & RING_MASK];
    return (w >> (24u - 8u * (rb & 3u))) & 0xFFu;
}

// Store every relative byte in [st.rflush, upto) and zero the ring chunks that are completely done.
// `upto` is wave-uniform.  Whole 16-byte chunks go out as one dwordx4 per lane.
__device__ __forceinline__ void ring_flush(EncState& st, uint32_t upto, int lane) {
    wave_lds_fence();
    if (lane < 2) {  // fold the slack dwords (spill of strings that started in the ring's last two dwords) back in
        const uint32_t v = st.ring[RING_WORDS + lane];
        if (v) {
            __hip_atomic_fetch_or(&st.ring[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            st.ring[RING_WORDS + lane] = 0u;
        }
    }
    wave_lds_fence();
    while (st.rflush < upto) {
        const uint32_t f = st.rflush;
        if ((f & 15u) != 0u || upto - f < 16u) {
            // ragged head or tail: byte stores (at most 15 bytes)
            uint32_t e = (f | 15u) + 1u;
            if (e > upto) e = upto;
            if ((uint32_t)lane < e - f) st.gbase[f + lane] = (uint8_t)ring_byte(st.ring, f + (uint32_t)lane);
            wave_lds_fence();
            if ((e & 15u) == 0u && (uint32_t)lane < 4u) st.ring[((f >> 4) * 4u + (uint32_t)lane) & RING_MASK] = 0u;
            st.rflush = e;
        } else {
            uint32_t nc = (upto - f) >> 4;
            if (nc > 64u) nc = 64u;
            if ((uint32_t)lane < nc) {
                const uint32_t c = (f >> 4) + (uint32_t)lane;
                uint4* rp = reinterpret_cast<uint4*>(&st.ring[(c * 4u) & RING_MASK]);
                uint4 v = *rp;
                v.x = bswap32(v.x);
                v.y = bswap32(v.y);
                v.z = bswap32(v.z);
                v.w = bswap32(v.w);
                *reinterpret_cast<uint4*>(st.gbase + (size_t)c * 16u) = v;
                *rp = make_uint4(0, 0, 0, 0);
            }
            st.rflush = f + nc * 16u;
        }
        wave_lds_fence();
    }
}

// Read the 16 bytes of this lane for the chunk at `p` (valid bytes: nb, 0..16). `fast` = 16-byte aligned source.
__device__ __forceinline__ uint4 load_lane16(const uint8_t* p, int nb, bool fast) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nb >= 16 && fast) {
        v = *reinterpret_cast<const uint4*>(p);
    } else if (nb > 0) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++)
            if (i < nb) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

// A lane's N strings go into the ring at bit p.  The strings a lane builds are what ALWAYS fits 64 bits (2 codewords of
// <= 26 bits, 4 of <= 16, 8 of <= 8); on real data neighbours usually fit together as well (text: 4.9 bits per symbol,
// zero pages with noise: 1.1): while that holds for every lane of the wave, adjacent strings are joined -- half as many
// ring_or's (3 LDS atomics and ~10 vector instructions each) per level.
template <int N>
__device__ __forceinline__ void join_emit(uint32_t* ring, uint32_t p, const unsigned long long (&gs)[N], const uint32_t (&gl)[N]) {
    if constexpr (N > 1) {
        uint32_t jl[N / 2];
        bool fit = true;
#pragma unroll
        for (int q = 0; q < N / 2; q++) {
            jl[q] = gl[2 * q] + gl[2 * q + 1];
            fit = fit && jl[q] <= 64u;
        }
        if (__builtin_amdgcn_ballot_w64(!fit) == 0ull) {  // wave-uniform
            unsigned long long js[N / 2];
#pragma unroll
            for (int q = 0; q < N / 2; q++)  // (a 64-bit right part means an empty left part: the shift amount wraps to 0)
                js[q] = (gs[2 * q] << (gl[2 * q + 1] & 63u)) | gs[2 * q + 1];
            join_emit<N / 2>(ring, p, js, jl);
            return;
        }
    }
#pragma unroll
    for (int q = 0; q < N; q++) {
        ring_or(ring, p, gs[q], gl[q]);
        p += gl[q];
    }
}

// Packed 32-bit entries, G symbols per register string; FULL = every lane holds 16 valid bytes.
//   G == 2 (maxlen <= 26): entry = code << 6 | len, strings built by 64-bit shifts;
//   G == 4 (maxlen <= 16), G == 8 (maxlen <= 8): entry = code << 16 | len; adjacent codewords are first joined
//   in 32-bit registers (a pair is <= 32 bits, and for maxlen <= 8 so is a quad), then once in 64 bits.
template <int G, bool FULL>
__device__ __forceinline__ void encode_chunk_packed(EncState& st, const uint32_t* lut, uint32_t col, const uint4& d,
                                                    int nb, int lane) {
    const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
    unsigned long long gs[16 / G];
    uint32_t gl[16 / G];
    uint32_t total = 0;
    if constexpr (G == 2) {
#pragma unroll
        for (int q = 0; q < 16 / G; q++) {
            unsigned long long g = 0;
            uint32_t l = 0;
#pragma unroll
            for (int k = 0; k < G; k++) {
                const int i = q * G + k;
                const uint32_t sym = (dw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                uint32_t e = lut[(sym << K3_CSHIFT) + col];
                if (!FULL && i >= nb) e = 0;
                const uint32_t li = e & 63u;
                g = (g << li) | (unsigned long long)(e >> 6);
                l += li;
            }
            gs[q] = g;
            gl[q] = l;
            total += l;
        }
    } else {
        uint32_t pc[8], pl[8];  // pairs
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i0 = 2 * j, i1 = 2 * j + 1;
            const uint32_t s0 = (dw[i0 >> 2] >> (8 * (i0 & 3))) & 0xFFu;
            const uint32_t s1 = (dw[i1 >> 2] >> (8 * (i1 & 3))) & 0xFFu;
            uint32_t e0 = lut[(s0 << K3_CSHIFT) + col];
            uint32_t e1 = lut[(s1 << K3_CSHIFT) + col];
            if (!FULL && i0 >= nb) e0 = 0;
            if (!FULL && i1 >= nb) e1 = 0;
            // entry = code << 16 | len with len <= 16: a shift takes its amount from the low 5 (6) bits of the register,
            // and the low 16 bits of a SUM of entries are the sum of the lengths -- no masking until the strings are final
            pc[j] = ((e0 >> 16) << (e1 & 31u)) | (e1 >> 16);
            pl[j] = e0 + e1;
        }
        if constexpr (G == 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                gs[q] = ((unsigned long long)pc[2 * q] << (pl[2 * q + 1] & 63u)) | (unsigned long long)pc[2 * q + 1];
                gl[q] = (pl[2 * q] + pl[2 * q + 1]) & 0xFFFFu;
                total += gl[q];
            }
        } else {
            uint32_t qc[4], ql[4];  // quads still fit 32 bits
#pragma unroll
            for (int q = 0; q < 4; q++) {
                qc[q] = (pc[2 * q] << (pl[2 * q + 1] & 31u)) | pc[2 * q + 1];
                ql[q] = pl[2 * q] + pl[2 * q + 1];
            }
#pragma unroll
            for (int o = 0; o < 2; o++) {
                gs[o] = ((unsigned long long)qc[2 * o] << (ql[2 * o + 1] & 63u)) | (unsigned long long)qc[2 * o + 1];
                gl[o] = (ql[2 * o] + ql[2 * o + 1]) & 0xFFFFu;
                total += gl[o];
            }
        }
    }
    const uint32_t inc = wave_inclusive_scan_u32(total);
    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    join_emit<16 / G>(st.ring, st.rpos + inc - total, gs, gl);
    st.rpos += wave_total;
    (void)lane;
}

// Wide entries (code<<8 | len as u64, unreplicated): any length up to 32. 8 symbols per lane per step.
__device__ __forceinline__ void encode_chunk_wide(EncState& st, const unsigned long long* lut64, uint32_t lo,
                                                  uint32_t hi, int nb) {
    const uint32_t dw[2] = {lo, hi};
    unsigned long long cs[8];
    uint32_t ls[8];
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t sym = (dw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        unsigned long long e = lut64[sym];
        if (i >= nb) e = 0;
        ls[i] = (uint32_t)(e & 0xFFu);
        cs[i] = e >> 8;
        total += ls[i];
    }
    const uint32_t inc = wave_inclusive_scan_u32(total);
    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    join_emit<8>(st.ring, st.rpos + inc - total, cs, ls);
    st.rpos += wave_total;
}

__global__ __launch_bounds__(K3_WAVES * 64) void k3_encode(const uint8_t* __restrict__ in, size_t n, size_t block_bytes,
                                                            uint32_t spb, uint32_t groups_per_block,
                                                            const uint8_t* __restrict__ d_len,
                                                            const uint32_t* __restrict__ d_code,
                                                            const uint8_t* __restrict__ d_maxlen,
                                                            const unsigned long long* __restrict__ d_comp_off,
                                                            const unsigned long long* __restrict__ d_seg_bitoff,
                                                            const int32_t* __restrict__ d_status,
                                                            uint8_t* __restrict__ out) {
    // one array: the codebook (256 entries x K3_COPIES replicas), then one ring of 1024 (+ 4) dwords per wave
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * K3_COPIES + K3_WAVES * RING_STRIDE];
    const uint32_t b = blockIdx.x / groups_per_block;
    const uint32_t grp = blockIdx.x % groups_per_block;
    const int tid = (int)threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;

    if (d_status[b] != DCZ_OK) return;  // workgroup-uniform
    const uint32_t maxlen_flag = d_maxlen[b];
    const uint32_t maxlen = maxlen_flag & 0x7Fu;
    const bool wide = maxlen > 26;
    if (maxlen_flag & 0x80u) return;  // 256 symbols of 8 bits: the payload is the input, k3_copy_identity wrote it

    // codebook -> LDS
    uint32_t* lut = lds;
    unsigned long long* lut64 = reinterpret_cast<unsigned long long*>(lds);
    if (!wide) {
        for (int i = tid; i < 256 * K3_COPIES; i += K3_WAVES * 64) {
            const int s = i >> K3_CSHIFT;
            const uint32_t cd = d_code[(uint64_t)b * 256u + s], ln = d_len[(uint64_t)b * 256u + s];
            lut[i] = (maxlen <= 16) ? ((cd << 16) | ln) : ((cd << 6) | ln);
        }
    } else {
        for (int s = tid; s < 256; s += K3_WAVES * 64)
            lut64[s] = ((unsigned long long)d_code[(uint64_t)b * 256u + s] << 8) |
                       (unsigned long long)d_len[(uint64_t)b * 256u + s];
    }
    uint32_t* ring = lds + 256 * K3_COPIES + w * RING_STRIDE;
    {
        uint4* r4 = reinterpret_cast<uint4*>(ring);
#pragma unroll
        for (int i = 0; i < RING_WORDS / 4 / 64; i++) r4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        if (lane == 0) r4[RING_WORDS / 4] = make_uint4(0, 0, 0, 0);  // slack
    }
    __syncthreads();

    const uint64_t bstart = (uint64_t)b * block_bytes;
    const uint64_t bend = (bstart + block_bytes < n) ? bstart + block_bytes : (uint64_t)n;
    const uint32_t blen = (uint32_t)(bend - bstart);
    const uint32_t nsb = (blen + SEG - 1) / SEG;
    const uint32_t j = grp * K3_WAVES + (uint32_t)w;
    if (j >= nsb) return;  // wave-uniform; no workgroup barrier below

    const uint32_t soff = j * SEG;
    const uint32_t slen = (blen - soff < SEG) ? blen - soff : SEG;
    const uint8_t* src = in + bstart + soff;
    const bool last_seg = (j + 1 == nsb);

    // output coordinates: virtual bit v = 8*(address & 15) + bit offset in block; origin = 16-byte chunk holding v0
    uint8_t* const blk_out = out + d_comp_off[b];
    const uint32_t a = (uint32_t)((uintptr_t)blk_out & 15u);
    const unsigned long long v0 = 8ull * a + d_seg_bitoff[(uint64_t)b * spb + j];
    const unsigned long long origin = v0 & ~127ull;
    EncState st;
    st.ring = ring;
    st.gbase = (blk_out - a) + (origin >> 3);
    st.rpos = (uint32_t)(v0 - origin);
    st.rflush = (st.rpos + 7u) >> 3;  // first byte whose first bit is ours
    const uint32_t col = (uint32_t)lane & (uint32_t)(K3_COPIES - 1);
    const bool fast = (((uintptr_t)src) & 15u) == 0u;
    // If the first owned byte is exactly the start of ring chunk 1, chunk 0 only ever holds the few leading
    // bits that belong to the previous segment's last byte: it is never flushed, so it must be cleared by
    // hand after the first OR phase or its bits would come back when the ring wraps.
    const bool clear_chunk0 = (st.rflush == 16u);

    if (!wide) {
        const uint32_t nchunks = (slen + 1023u) >> 10;
        const uint32_t step_bits = 1024u * maxlen + 512u;  // (+ the partly flushed chunk in front and the slack dwords)
        const uint32_t flush_at = DCZ_K3_LAZY_FLUSH && step_bits < RING_WORDS * 32u ? RING_WORDS * 32u - step_bits : 0u;
        uint4 cur = make_uint4(0, 0, 0, 0);
        {
            const int nb = (int)slen - lane * 16;
            cur = load_lane16(src + lane * 16, nb, fast);
        }
        for (uint32_t c = 0; c < nchunks; c++) {
            uint4 nxt = make_uint4(0, 0, 0, 0);
            if (c + 1 < nchunks) {
                const uint32_t o = (c + 1) * 1024u + (uint32_t)lane * 16u;
                nxt = load_lane16(src + o, (int)slen - (int)o, fast);
            }
            const int nb = (int)slen - (int)(c * 1024u) - lane * 16;
            const bool full = slen - c * 1024u >= 1024u;  // wave-uniform
            if (maxlen <= 8) {
                if (full) encode_chunk_packed<8, true>(st, lut, col, cur, nb, lane);
                else encode_chunk_packed<8, false>(st, lut, col, cur, nb, lane);
            } else if (maxlen <= 16) {
                if (full) encode_chunk_packed<4, true>(st, lut, col, cur, nb, lane);
                else encode_chunk_packed<4, false>(st, lut, col, cur, nb, lane);
            } else {
                if (full) encode_chunk_packed<2, true>(st, lut, col, cur, nb, lane);
                else encode_chunk_packed<2, false>(st, lut, col, cur, nb, lane);
            }
            if (c == 0 && clear_chunk0) {
                wave_lds_fence();
                if (lane < 4) st.ring[lane] = 0u;
            }
            // The ring is flushed when the NEXT step could run into bytes that are still in it (a step adds at most
            // 1024 codewords of maxlen bits), not after every step: a flush is two fences, a read, a store and a zeroing
            // pass whose lanes are mostly idle when a step produced little (zero pages: 8 of 64).
            if ((int)(st.rpos - 8u * st.rflush) > (int)flush_at) ring_flush(st, (st.rpos >> 7) << 4, lane);  // wave-uniform
            cur = nxt;
        }
    } else {
        const uint32_t nchunks = (slen + 511u) >> 9;
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t o = c * 512u + (uint32_t)lane * 8u;
            const int nb = (int)slen - (int)o;
            uint32_t lo = 0, hi = 0;
            for (int i = 0; i < 8; i++)
                if (i < nb) {
                    const uint32_t by = src[o + i];
                    if (i < 4) lo |= by << (8 * i); else hi |= by << (8 * (i - 4));
                }
            encode_chunk_wide(st, lut64, lo, hi, nb);
            if (c == 0 && clear_chunk0) {
                wave_lds_fence();
                if (lane < 4) st.ring[lane] = 0u;
            }
            ring_flush(st, (st.rpos >> 7) << 4, lane);
        }
    }

    // Last byte of the segment: complete it with look-ahead symbols of the next segment (at most 7 bits
    // are needed and every codeword has at least 1 bit), or leave the zero padding at the end of the block.
    const uint32_t own_end = (st.rpos + 7u) >> 3;
    if (!last_seg && (st.rpos & 7u) != 0u) {
        unsigned long long cd = 0;
        uint32_t l = 0;
        if (lane < 7 && soff + slen + (uint32_t)lane < blen) {
            const uint32_t sym = src[slen + lane];
            cd = d_code[(uint64_t)b * 256u + sym];
            l = d_len[(uint64_t)b * 256u + sym];
        }
        const uint32_t inc = wave_inclusive_scan_u32(l);
        ring_or(st.ring, st.rpos + inc - l, cd, l);
    }
    ring_flush(st, own_end, lane);
}

// Blocks whose code is 256 symbols of 8 bits (K2 sets bit 7 of d_maxlen): codeword(s) = s (CanonicalHuffman.java:123-129
// assigns ascending codes in symbol order), so encodeChunk's output (CpuCompressionService.java:303-315) is the input
// itself -- the reference's high-entropy case (app/logs/datacomp.log:3254).  A flat grid of one workgroup per (block,
// 16 KiB tile) over all blocks copies them with non-temporal 16 B/lane accesses (see k4_fixed.hip for why flat); a
// workgroup of any other block leaves after one byte load, and k3_encode leaves at once for the blocks handled here.
#ifndef DCZ_CP_TILE
#define DCZ_CP_TILE 16384
#endif
constexpr uint32_t CP_TILE = DCZ_CP_TILE;
#ifndef DCZ_CP_T
#define DCZ_CP_T 256  // threads per tile, four 16-byte accesses per lane (see k4_fixed.hip for the shapes measured)
#endif
constexpr int CP_T = DCZ_CP_T;
constexpr int CP_UPT = (int)(CP_TILE / 16u / (uint32_t)CP_T);
static_assert(CP_UPT >= 1 && CP_UPT * CP_T * 16 == (int)CP_TILE, "tile = threads x units x 16 bytes");
// one work item: tile `tile` of block b0 + bq
__device__ __forceinline__ void copy_identity_item(uint32_t bq, uint32_t tile, const uint8_t* __restrict__ in, size_t n,
                                                   size_t block_bytes, uint32_t b0, const uint8_t* __restrict__ d_maxlen,
                                                   const unsigned long long* __restrict__ d_comp_off,
                                                   const int32_t* __restrict__ d_status, uint8_t* __restrict__ out,
                                                   bool in_place, unsigned long long in_offset) {
    const uint32_t b = b0 + bq;
    if ((d_maxlen[b] & 0x80u) == 0u || d_status[b] != DCZ_OK) return;  // workgroup-uniform
    const uint64_t bstart = (uint64_t)b * block_bytes;
    if (in_place && d_comp_off[b] == in_offset + bstart) return;  // K1 stored this block where it belongs (ShapeHint)
    const uint64_t bend = (bstart + block_bytes < n) ? bstart + block_bytes : (uint64_t)n;
    const uint64_t t0 = (uint64_t)tile * CP_TILE;
    if (bstart + t0 >= bend) return;
    const uint32_t nout = (bend - bstart - t0 < CP_TILE) ? (uint32_t)(bend - bstart - t0) : CP_TILE;
    const uint8_t* src = in + bstart + t0;
    uint8_t* dst = out + d_comp_off[b] + t0;
    const int tid = (int)threadIdx.x;
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0u && nout == CP_TILE) {
        const u32x4* s4 = reinterpret_cast<const u32x4*>(src) + tid;
        u32x4* d4 = reinterpret_cast<u32x4*>(dst) + tid;
        u32x4 v[CP_UPT];
#pragma unroll
        for (int k = 0; k < CP_UPT; k++) v[k] = __builtin_nontemporal_load(s4 + CP_T * k);
#pragma unroll
        for (int k = 0; k < CP_UPT; k++) __builtin_nontemporal_store(v[k], d4 + CP_T * k);
    } else if (((((uintptr_t)src) | ((uintptr_t)dst)) & 3u) == 0u) {
        const uint32_t nw = nout >> 2;
        for (uint32_t i = (uint32_t)tid; i < nw; i += (uint32_t)CP_T)
            reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (uint32_t i = (nw << 2) + (uint32_t)tid; i < nout; i += (uint32_t)CP_T) dst[i] = src[i];
    } else {
        for (uint32_t i = (uint32_t)tid; i < nout; i += (uint32_t)CP_T) dst[i] = src[i];
    }
}

// PERSIST = false: flat grid, one workgroup per work item (what copies fastest).  PERSIST = true: a small grid that walks
// over the work items -- for calls in which no block is expected to have the identity code (ShapeHint, dcz_internal.h).
template <bool PERSIST>
__global__ __launch_bounds__(CP_T) void k3_copy_identity(const uint8_t* __restrict__ in, size_t n, size_t block_bytes,
                                                         uint32_t tiles_per_block, uint32_t b0, uint32_t nblk,
                                                         const uint8_t* __restrict__ d_maxlen,
                                                         const unsigned long long* __restrict__ d_comp_off,
                                                         const int32_t* __restrict__ d_status, uint8_t* __restrict__ out,
                                                         bool in_place, unsigned long long in_offset) {
    if constexpr (!PERSIST) {
        const uint32_t bq = blockIdx.x / tiles_per_block;
        copy_identity_item(bq, blockIdx.x - bq * tiles_per_block, in, n, block_bytes, b0, d_maxlen, d_comp_off, d_status, out,
                           in_place, in_offset);
    } else {
        // CP_T blocks at a time: every thread looks at one flag, the flagged blocks are compacted in block order (every
        // workgroup builds the same list) and the (flagged block, tile) pairs of the range are dealt round-robin over the
        // whole grid -- a wrong expectation must not leave most of the chip idle (see k4_fixed<true>)
        __shared__ uint16_t flagged[CP_T];
        __shared__ uint32_t fcnt[CP_T / 64];
        const uint32_t tid = threadIdx.x, w = tid >> 6;
        uint32_t rot = 0;
        for (uint32_t c0 = 0; c0 < nblk; c0 += (uint32_t)CP_T) {
            const uint32_t bi = c0 + tid;
            const bool mine = bi < nblk && (d_maxlen[b0 + bi] & 0x80u) != 0u &&
                              !(in_place && d_comp_off[b0 + bi] == in_offset + (unsigned long long)(b0 + bi) * block_bytes);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if ((tid & 63u) == 0u) fcnt[w] = (uint32_t)__builtin_popcountll(m);
            __syncthreads();
            uint32_t base = 0, nflag = 0;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)(CP_T / 64); i++) {
                base += i < w ? fcnt[i] : 0u;
                nflag += fcnt[i];
            }
            if (nflag == 0u) {  // workgroup-uniform
                __syncthreads();
                continue;
            }
            if (mine) flagged[base + (uint32_t)__builtin_popcountll(m & ((1ull << (tid & 63u)) - 1ull))] = (uint16_t)tid;
            __syncthreads();
            const unsigned long long items = (unsigned long long)nflag * tiles_per_block;
            const uint32_t first = (blockIdx.x + gridDim.x - rot) % gridDim.x;
            for (unsigned long long it = first; it < items; it += gridDim.x) {
                const uint32_t f = (uint32_t)(it / tiles_per_block);
                copy_identity_item(c0 + flagged[f], (uint32_t)(it - (unsigned long long)f * tiles_per_block), in, n, block_bytes,
                                   b0, d_maxlen, d_comp_off, d_status, out, in_place, in_offset);
            }
            rot = (uint32_t)((rot + items) % gridDim.x);
            __syncthreads();  // (flagged / fcnt are rewritten by the next range)
        }
    }
}

void launch_encode(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t segs_per_block, uint32_t K,
                   const uint8_t* d_len, const uint32_t* d_code, const uint8_t* d_maxlen, const uint64_t* d_comp_off,
                   const uint64_t* d_seg_bitoff, const int32_t* d_status, uint8_t* d_out, hipStream_t s,
                   const ShapeHint& hint) {
    if (K == 0) return;
    if (!hint.likely || hint.in_place) {  // (in place: nothing is expected to be left for this kernel)
        const size_t eff = (K <= 1) ? (n ? n : 1) : block_bytes;
        const uint64_t tpb = (eff + CP_TILE - 1) / CP_TILE;
        if (tpb <= 0xFFFFFFFFull)
            hipLaunchKernelGGL(k3_copy_identity<true>, dim3(HINT_PERSIST_GRID), dim3(CP_T), 0, s, d_in, n, block_bytes,
                               (uint32_t)tpb, 0u, K, d_maxlen, reinterpret_cast<const unsigned long long*>(d_comp_off),
                               d_status, d_out, hint.in_place, hint.in_offset);
    } else {
        const size_t eff = (K <= 1) ? (n ? n : 1) : block_bytes;
        const uint64_t tpb = (eff + CP_TILE - 1) / CP_TILE;
        const uint64_t per = (0x40000000ull / tpb) ? (0x40000000ull / tpb) : 1;  // blocks per launch (grid < 2^31)
        for (uint64_t b0 = 0; b0 < K; b0 += per) {
            const uint64_t kb = (K - b0 < per) ? K - b0 : per;
            hipLaunchKernelGGL(k3_copy_identity<false>, dim3((uint32_t)(kb * tpb)), dim3(CP_T), 0, s, d_in, n, block_bytes,
                               (uint32_t)tpb, (uint32_t)b0, (uint32_t)kb, d_maxlen,
                               reinterpret_cast<const unsigned long long*>(d_comp_off), d_status, d_out, false, 0ull);
        }
    }
    const uint32_t gpb = (segs_per_block + K3_WAVES - 1) / K3_WAVES;
    hipLaunchKernelGGL(k3_encode, dim3(K * gpb), dim3(K3_WAVES * 64), 0, s, d_in, n, block_bytes, segs_per_block, gpb,
                       d_len, d_code, d_maxlen, reinterpret_cast<const unsigned long long*>(d_comp_off),
                       reinterpret_cast<const unsigned long long*>(d_seg_bitoff), d_status, d_out);
}

}  // namespace dcz
