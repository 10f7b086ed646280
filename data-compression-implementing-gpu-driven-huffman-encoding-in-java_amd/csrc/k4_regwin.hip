// k4_regwin.hip -- K4 for medium code lengths (3.6 .. 6.5 bits per symbol, e.g. text): register-window, multi-symbol
// table walk (gfx950).
//
// Same contract as k4_decode.hip (TableBasedHuffmanDecoder.decode, core/TableBasedHuffmanDecoder.java:103-152 with the
// fallback of core/CanonicalHuffman.java:161-229; zero bits past the payload :204-208; "decode error at position i"
// :109-111) and the same outer structure: one workgroup walks its block in windows of W subsequences x 32 bytes, phase A
// finds every subsequence's entry by self-synchronisation, a workgroup scan turns symbol counts into output offsets,
// phase B decodes into an LDS tile that is flushed with aligned 16-byte stores.  What differs is the inner loop, which
// rocprof showed to be LDS-bound in k4_decode's medium class (profiles/r02_base_text8g_sq.txt: LDS busy 66 % of the
// kernel, 58 % of that bank conflicts of the random-index table reads, two LDS round trips per symbol on the chain):
//   * the subsequence lives in REGISTERS (8 big-endian dwords + 2 look-ahead dwords from the right neighbour), and the
//     walk is unrolled over the dword index: while the position is inside dword k the next 32 stream bits are a 64-bit
//     shift of the static pair {R[k], R[k+1]} -- no LDS read for the window, one LDS round trip per step;
//   * phase A does not decode symbols, it JUMPS: jt[12-bit window] = (total bits, number) of the complete codewords
//     inside the window, ~9 bits and ~2 symbols per step on text, and only has to deliver each subsequence's exit and
//     symbol count; position and count live in one register (rel | n << 8) that a step advances with a single add;
//   * phase B is count-driven (decode exactly the symbols phase A counted from the same entry) with up to three
//     symbols per lookup: mt[12-bit window] = s0 | s1 << 8 | s2 << 16 | bits << 24 | count << 30, bytes stored straight
//     into the tile at the offsets of the scan (flushes end on subsequence boundaries, so a lookup is only ever cut short
//     where the block ends); position and tile index live in one register (rel | t << 6) advanced by (entry >> 24).
// A subsequence owns the codewords from its entry up to the first JUMP LANDING at or past its end (not the first codeword
// boundary): entries and exits are still a fixed point of "my entry = my left neighbour's exit", the first one is exact,
// and jump chains merge like codeword parses do, so the rounds behave as before (2.0 per window on text); a window that is
// not synchronised after DCZ_K4_EXACT_AFTER rounds hands its block to the exact-entry launch of k4_decode.hip.
// Codewords longer than the table (or none: an invalid pattern of an incomplete code) take the canonical
// first-code/count search, which returns what the reference's table-then-HashMap path returns.
#include <cstdlib>
#include <utility>

#include "dcz_internal.h"

namespace dcz {

#ifndef DCZ_RW_TBJ
#define DCZ_RW_TBJ 12  // index bits of the jump table (u16 entries: bits | count << 8)
#endif
#ifndef DCZ_RW_TBM
#define DCZ_RW_TBM 11  // index bits of the output table (u32 entries)
#endif
#ifndef DCZ_RW_OC
#define DCZ_RW_OC 16384  // tile bytes per flush, many-blocks kernel: a whole window of text in one flush (with two,
                         // half of the waves idle in each: 20.4 -> 16.3 ms on 8 GiB of text)
#endif
#ifndef DCZ_RW_OCS
#define DCZ_RW_OCS 32768  // few-blocks kernel (1024 threads)
#endif
#ifndef DCZ_RW_UNROLL
#define DCZ_RW_UNROLL 3  // straight-line steps per dword before the loop (jump steps: ~3.4 per dword on text)
#endif
#ifndef DCZ_RW_ABL
#define DCZ_RW_ABL 0  // timing ablations (debug only, wrong results): 1 no jump-table read, 2 no tile stores, 4 no output-table read
#endif
#ifndef DCZ_RW_MINWAVES
#define DCZ_RW_MINWAVES 4
#endif
#ifndef DCZ_K4_EXACT_AFTER
#define DCZ_K4_EXACT_AFTER 12
#endif
#ifndef DCZ_K4_CLS2_A
#define DCZ_K4_CLS2_A 4
#endif
#ifndef DCZ_K4_CLS2_B
#define DCZ_K4_CLS2_B 9
#endif

constexpr int RW_SUB_BITS = 256;

template <int OC>
struct CAP_OK {
    static constexpr bool value = (OC + 512) < (1 << 24);
};

template <int W, int TBJ, int TBM, int OC>
struct RwLds {
    static_assert(W % 64 == 0 && TBM <= TBJ && TBJ <= 15 && (2 << TBJ) <= (4 << TBM), "table geometry");
    static_assert(CAP_OK<OC>::value, "tile index fits the packed phase-B state");
    static constexpr int CAP = OC + 512;                              // tile capacity in bytes
    __attribute__((aligned(16))) uint32_t mt[1 << TBM];               // (holds the single-symbol table during the build)
    __attribute__((aligned(16))) uint32_t tile[CAP / 4 + 8];
    uint16_t jt[1 << TBJ];  // bits | count << 8 of the complete codewords inside the window, 0 = none
    uint32_t head0[W + 1], head1[W + 1];                              // first two dwords of every stripe (+ the window's successor)
    uint16_t exits[W];
    uint16_t nbad[W];  // symbols a parse counted before it met a pattern without a codeword
    unsigned long long lim[40];  // lim[l] = (first[l] + cnt[l]) << (32 - l): exclusive left-aligned upper bound of length l
    uint32_t first[34], cnt[34], offs[34];
    uint32_t wsum[W / 64];
    uint32_t flag[3];
    uint8_t symtab[256], len8[256];
    uint32_t maxlen, err_idx, cend_vote;
    int bad_table;
};

// the next 32 stream bits at bit `pos` of the big-endian dword pair {hi, lo} (pos & 31 inside hi)
__device__ __forceinline__ uint32_t rw_bits(uint32_t hi, uint32_t lo, uint32_t pos) {
    return (uint32_t)((((unsigned long long)hi << 32) | lo) << (pos & 31u) >> 32);
}

// high dword of (pair << (sh & 63)): the next 32 stream bits at bit sh of the big-endian pair (sh < 32 for walking lanes)
__device__ __forceinline__ uint32_t rw_hi_shl(unsigned long long pair, uint32_t sh) {
    unsigned long long r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(pair));
    return (uint32_t)(r >> 32);
}
// lane in mask ? v : 0 with the wave mask in a scalar register pair
__device__ __forceinline__ uint32_t rw_select(uint32_t v, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

// lane in mask ? v : other
__device__ __forceinline__ uint32_t rw_select2(uint32_t v, uint32_t other, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(other), "v"(v), "s"(m));
    return r;
}

// wave masks straight from a vector compare (no bool -> 0/1 -> compare round trip)
#define RW_LT(a, b) __builtin_amdgcn_uicmp((uint32_t)(a), (uint32_t)(b), 36)  // unsigned <
#define RW_EQ(a, b) __builtin_amdgcn_uicmp((uint32_t)(a), (uint32_t)(b), 32)

// Codes longer than TB bits, and invalid patterns: the length of window w is the first l > TB with w < lim[l]
// (64-bit, left-aligned; canonical ranges are contiguous and ascending in the length).  (len << 8) | symbol, 0 = none.
template <int TB, class LdsT>
__device__ __attribute__((noinline)) uint32_t rw_slow(const LdsT& L, uint32_t win32) {
    const uint32_t maxlen = L.maxlen;
    const unsigned long long w = win32;
    for (uint32_t l = TB + 1; l <= maxlen; l += 4) {
        const unsigned long long a = L.lim[l], b = L.lim[l + 1], c = L.lim[l + 2], d = L.lim[l + 3];
        const uint32_t k = (w < a) ? 0u : (w < b) ? 1u : (w < c) ? 2u : (w < d) ? 3u : 4u;
        if (k < 4u) {
            const uint32_t ll = l + k;  // <= maxlen: lim[] is flat beyond maxlen
            return (ll << 8) | (uint32_t)L.symtab[L.offs[ll] + ((win32 >> (32u - ll)) - L.first[ll])];
        }
    }
    return 0;
}

template <int W, class LdsT>
__device__ __forceinline__ uint32_t rw_block_scan(uint32_t v, LdsT& L, uint32_t& total) {
    const uint32_t inc = wave_inclusive_scan_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) L.wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < W / 64; w++) {
        const uint32_t sv = L.wsum[w];
        if (w < (int)(threadIdx.x >> 6)) base += sv;
        tot += sv;
    }
    total = tot;
    return base + inc - v;
}

#if DCZ_K4_PROF
// debug build only: cycles per phase, summed over wave 0 of every workgroup (tools/k4prof.py)
__device__ unsigned long long rw_prof[12];  // [8] windows, [9] rounds, [10] flushes
#define RW_T(i)                                   \
    do {                                          \
        const unsigned long long t_ = clock64();  \
        pacc[i] += t_ - plast;                    \
        plast = t_;                               \
    } while (0)
#else
#define RW_T(i) do { } while (0)
#endif

template <int W, int TBJ, int TBM, int OC>
__global__ __launch_bounds__(W, W <= 256 ? DCZ_RW_MINWAVES : 1) void k4_regwin(
    const uint8_t* __restrict__ comp, const unsigned long long* __restrict__ d_comp_off,
    const uint32_t* __restrict__ d_comp_size, const uint32_t* __restrict__ d_orig_size, const uint8_t* __restrict__ d_len,
    size_t out_stride, uint8_t* __restrict__ out, int32_t* __restrict__ d_status, long long* __restrict__ d_errpos,
    uint8_t* __restrict__ d_cls) {
    using LdsT = RwLds<W, TBJ, TBM, OC>;
    __shared__ LdsT L;
    const uint32_t b = blockIdx.x;
    const int tid = (int)threadIdx.x;
#if DCZ_K4_PROF
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = clock64();
#endif
    if (d_cls[b] != 0) return;  // fixed-length, exact-entry or rejected block: another launch owns it (workgroup-uniform)

    // ---- block geometry and class (medium: 8 * A / B .. 6.5 bits per symbol, from the block's own sizes) ----
    const uint32_t orig = d_orig_size[b];
    const unsigned long long coff = d_comp_off[b];
    const uint32_t csize = d_comp_size[b];
    {
        const bool long_codes = (unsigned long long)csize * 16ull >= (unsigned long long)orig * 13ull;
        const bool medium = (unsigned long long)orig * (unsigned long long)DCZ_K4_CLS2_A <=
                            (unsigned long long)csize * (unsigned long long)DCZ_K4_CLS2_B;
        if (long_codes || !medium) return;  // workgroup-uniform; k4_decode's launches own those blocks
    }

    // ---- per-block tables (rebuildCodes: CpuCompressionService.java:582-586 -> CanonicalHuffman.java:99-132) ----
    if (tid < 34) L.cnt[tid] = 0;
    if (tid == 0) {
        L.bad_table = 0;
        L.err_idx = 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int sy = tid; sy < 256; sy += W) {
        const uint32_t l = d_len[(uint64_t)b * 256u + sy];
        L.len8[sy] = (uint8_t)l;
        if (l > 32) L.bad_table = 1;
        else if (l > 0) atomicAdd(&L.cnt[l], 1u);
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
            L.lim[l] = (unsigned long long)(c + L.cnt[l]) << (32 - l);
        }
        L.lim[0] = 0;
        for (int l = 33; l < 40; l++) L.lim[l] = L.lim[32];
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
    for (int sy = tid; sy < 256; sy += W) {
        const uint32_t l = L.len8[sy];
        if (l > 0) {
            uint32_t rank = 0;
            for (int t = 0; t < sy; t++) rank += (L.len8[t] == l) ? 1u : 0u;
            L.symtab[L.offs[l] + rank] = (uint8_t)sy;
        }
    }
    __syncthreads();
    // single-symbol table of TBJ bits (len << 8 | symbol, 0 = longer or none), kept in mt's memory while jt and mt are built
    uint16_t* const st = reinterpret_cast<uint16_t*>(L.mt);
    for (int idx = tid; idx < (1 << TBJ); idx += W) {
        uint32_t e = 0;
        for (uint32_t l = 1; l <= (uint32_t)TBJ; l++) {
            const uint32_t c = (uint32_t)idx >> (TBJ - l);
            const uint32_t f = L.first[l];
            if (c >= f && c - f < L.cnt[l]) {
                e = (l << 8) | L.symtab[L.offs[l] + (c - f)];
                break;
            }
        }
        st[idx] = (uint16_t)e;
    }
    __syncthreads();
    constexpr uint32_t JMASK = (1u << TBJ) - 1u;
    for (int idx = tid; idx < (1 << TBJ); idx += W) {  // jump table: all complete codewords inside the TBJ-bit window
        uint32_t pos = 0, cnt = 0;
        while (pos < (uint32_t)TBJ) {
            const uint32_t e = st[((uint32_t)idx << pos) & JMASK];
            const uint32_t len = e >> 8;
            if (e == 0 || len > (uint32_t)TBJ - pos) break;
            pos += len;
            cnt++;
        }
        L.jt[idx] = (uint16_t)(cnt ? ((cnt << 8) | pos) : 0u);
    }
    constexpr int MPT = (1 << TBM) / W > 0 ? (1 << TBM) / W : 1;
    static_assert((1 << TBM) % W == 0, "whole entries per thread");
    uint32_t me[MPT];
#pragma unroll
    for (int i = 0; i < MPT; i++) {  // output table: the first <= 3 complete codewords inside the TBM-bit window
        const uint32_t idx = (uint32_t)(tid + i * W);
        uint32_t pos = 0, cnt = 0, o = 0;
        while (cnt < 3u) {
            const uint32_t e = st[((idx << (TBJ - TBM)) << pos) & JMASK];
            const uint32_t len = e >> 8;
            if (e == 0 || len > (uint32_t)TBM - pos) break;
            o |= (e & 0xFFu) << (8 * cnt);
            pos += len;
            cnt++;
        }
        me[i] = cnt ? (o | (pos << 24) | (cnt << 30)) : 0u;
    }
    __syncthreads();  // every reader of st is done
#pragma unroll
    for (int i = 0; i < MPT; i++) L.mt[tid + i * W] = me[i];

    uint8_t* const oblk = out + (uint64_t)b * out_stride;
    const bool out_aligned = (((uintptr_t)oblk) & 15u) == 0u;
    // virtual byte 0 = 16-byte aligned address at or below the payload start
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)coff;
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew;
    const unsigned long long vhi = (unsigned long long)skew + csize;

    unsigned long long ventry = 8ull * skew;  // virtual bit of the next entry
    uint32_t produced = 0;                    // symbols decoded so far
    uint32_t gpos = 0;                        // block-relative output offset of tile byte 0 (multiple of 16)
    uint32_t ocarry = 0;                      // bytes at the front of the tile not yet stored (0..15)
    int status = DCZ_OK;
    long long errpos = 0;
    uint8_t* const ob = reinterpret_cast<uint8_t*>(L.tile);

    typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    const uint32_t jt_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.jt[0]));
    const uint32_t mt_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.mt[0]));
    uint4 pre[2];
    uint4 pre_m = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](unsigned long long wchunk0) {
#pragma unroll
        for (int c = 0; c < 2; c++) pre[c] = load_chunk16(vbase, (wchunk0 + (unsigned long long)(tid * 2 + c)) << 4, vlo, vhi);
        if (tid == W - 1) pre_m = load_chunk16(vbase, (wchunk0 + (unsigned long long)(W * 2)) << 4, vlo, vhi);
    };
    if (orig > 0) prefetch(ventry >> 7);
    __syncthreads();  // mt, jt complete

    RW_T(0);
    while (produced < orig) {
#if DCZ_K4_PROF
        pacc[8]++;
#endif
        const unsigned long long wchunk0 = ventry >> 7;
        const uint32_t g0 = (uint32_t)(ventry - (wchunk0 << 7));
        // the stripe in registers: R[0..7] own dwords (MSB first), R[8..9] the right neighbour's first two
        uint32_t R[10];
        R[0] = bswap32(pre[0].x);
        R[1] = bswap32(pre[0].y);
        R[2] = bswap32(pre[0].z);
        R[3] = bswap32(pre[0].w);
        R[4] = bswap32(pre[1].x);
        R[5] = bswap32(pre[1].y);
        R[6] = bswap32(pre[1].z);
        R[7] = bswap32(pre[1].w);
        L.head0[tid] = R[0];
        L.head1[tid] = R[1];
        if (tid == W - 1) {
            L.head0[W] = bswap32(pre_m.x);
            L.head1[W] = bswap32(pre_m.y);
        }
        __syncthreads();
        R[8] = L.head0[tid + 1];
        R[9] = L.head1[tid + 1];
        RW_T(1);

        // ---- phase A: self-synchronisation by jump chains ----
        // Subsequences that START past the payload hold nothing but zero padding (a periodic stream that need not
        // self-synchronise): they take no part; the symbols the reference would read from the padding are filled in
        // after the window (see "exhausted").
        const unsigned long long wbase_bits = wchunk0 << 7;
        const unsigned long long pay_end_bits = vhi << 3;
        const bool exhausted = wbase_bits + (unsigned long long)W * RW_SUB_BITS >= pay_end_bits;
        const bool beyond = exhausted && wbase_bits + (unsigned long long)tid * RW_SUB_BITS >= pay_end_bits;
        uint32_t g = (tid == 0) ? g0 : 0u;  // entry (bit offset inside the stripe)
        uint32_t x = 0, nsym = 0;           // exit (bits past the stripe's end), symbols owned
        bool bad = false, need = !beyond;
        uint32_t round = 0;
        while (true) {
            // Lanes that walk in this round as a scalar mask; state of a walking lane: st = rel | n << 8 with rel = bit offset
            // inside the current dword (>= 32: past it) and n = symbols counted.  A step adds the table entry
            // (bits | count << 8); after the lanes have left dword k, rel -= 32 for everybody.
            unsigned long long walking = __builtin_amdgcn_ballot_w64(need);
            const unsigned long long walking0 = walking;
            uint32_t stA = g;
            // (unrolled over the dword index through a fold expression: R[k] must be a compile-time register)
            auto walk = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const unsigned long long pair = ((unsigned long long)R[k] << 32) | R[k + 1];
                auto step = [&](unsigned long long am) __attribute__((always_inline)) {
                    const uint32_t w = rw_hi_shl(pair, stA);
#if DCZ_RW_ABL & 1
                    uint32_t e = 0x0209u + (w >> 31);  // ablation: no table read
#else
                    uint32_t e = *(lds_cu16*)(uintptr_t)(jt_addr + ((w >> (31 - TBJ)) & (uint32_t)(((1 << TBJ) - 1) << 1)));
#endif
                    asm("" : "+v"(e));  // a plain 32-bit value from here on
                    const unsigned long long esc = RW_EQ(e, 0u) & am;
                    if (__builtin_expect(esc != 0ull, 0)) {  // rare: a long codeword, or none
                        bool dd = false;
                        if (need && (stA & 0xE0u) == 0u && e == 0u) {
                            const uint32_t r = rw_slow<TBJ>(L, w);
                            e = (r >> 8) | 0x100u;  // one symbol of r >> 8 bits
                            if (r == 0u) {          // no codeword on this parse: it ends here
                                dd = true;
                                e = 0u;
                                L.nbad[tid] = (uint16_t)(stA >> 8);  // symbols before the undecodable pattern
                            }
                        }
                        walking &= ~__builtin_amdgcn_ballot_w64(dd);
                    }
                    stA += rw_select(e, am);
                };
                // a few straight-line steps (lanes that are past the dword are masked), then a loop for the rest
#pragma unroll
                for (int u = 0; u < DCZ_RW_UNROLL; u++) step(RW_EQ(stA & 0xE0u, 0u) & walking);
                while (true) {
                    const unsigned long long am = RW_EQ(stA & 0xE0u, 0u) & walking;  // rel < 32
                    if (am == 0ull) break;
                    step(am);
                }
                stA -= 32u;
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (walk(std::integral_constant<int, Ks>{}), ...);
            }(std::make_integer_sequence<int, 8>{});
            if (need) {
                bad = ((walking0 & ~walking) >> (tid & 63)) & 1ull;  // left the walk: no codeword on its parse
                x = bad ? 0u : (stA & 0xFFu);                       // (eight decrements of 32: rel - 256)
                nsym = bad ? (uint32_t)L.nbad[tid] : (stA >> 8);
            }
            RW_T(2);
#if DCZ_K4_PROF
            pacc[9]++;
#endif
            L.exits[tid] = (uint16_t)x;
            // One barrier per round; whether anybody had to walk in THIS round was recorded in flag[round % 3] during the
            // previous round's check (slot (round + 1) % 3 is cleared before the barrier, set after it, read after the next).
            if (tid == 0) L.flag[(round + 1u) % 3u] = 0;
            __syncthreads();
            RW_T(3);
            if (round > 0u && L.flag[round % 3u] == 0u) break;
            const uint32_t ng = (tid == 0) ? g0 : (uint32_t)L.exits[tid - 1];
            need = (ng != g) && !beyond;
            g = ng;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull && (tid & 63) == 0) L.flag[(round + 1u) % 3u] = 1;
            round++;
            if (round == (uint32_t)DCZ_K4_EXACT_AFTER) {  // workgroup-uniform: this block does not self-synchronise
                if (tid == 0) d_cls[b] = 1;                // the exact-entry launch (k4_decode.hip, MODE 1) decodes it
                return;
            }
        }

        // ---- offsets, errors ----
        uint32_t tw = 0;
        const uint32_t o = rw_block_scan<W>(nsym, L, tw);
        const uint32_t remaining = orig - produced;
        if (bad) atomicMin(&L.err_idx, o + nsym);
        const unsigned long long next_ventry =
            (wchunk0 << 7) + (unsigned long long)W * RW_SUB_BITS + (unsigned long long)L.exits[W - 1];
        __syncthreads();
        const uint32_t err_idx = L.err_idx;
        if (err_idx < remaining) {
            status = DCZ_E_BADSTREAM;
            errpos = (long long)produced + (long long)err_idx;
            break;
        }
        const uint32_t lim = (tw < remaining) ? tw : remaining;
        const bool more = produced + lim < orig;
        if (more) prefetch(next_ventry >> 7);  // lands in registers while phase B runs
        RW_T(4);

        // ---- phase B: count-driven decode into the tile, flushes end on subsequence boundaries ----
        // State of a lane: relB = bit offset relative to the current dword (< 32: inside it), tB = tile byte of its next
        // symbol; a step adds the entry's bits and count.  (Between the flushes of a window the position is kept as the
        // absolute stripe position bpos.)
        const uint32_t oe = (o + nsym < lim) ? o + nsym : (o < lim ? lim : o);
        uint32_t oi = o;    // window symbol index of this lane's next symbol
        uint32_t bpos = g;  // stripe bit position of that symbol
        for (uint32_t cbase = 0; cbase < lim;) {
            uint32_t cc = lim - cbase;
            const uint32_t room = (uint32_t)LdsT::CAP - ocarry;
            if (cc > room) {  // workgroup-uniform: the rest of the window does not fit one flush
                if (tid == 0) L.cend_vote = 0;
                __syncthreads();
                const uint32_t end0 = o + nsym;
                if (end0 > cbase + room / 2u && end0 <= cbase + room) atomicMax(&L.cend_vote, end0);
                __syncthreads();
                const uint32_t v = L.cend_vote;
                cc = v != 0u ? v - cbase : room;  // (a subsequence holds < 300 symbols: a boundary always exists)
            }
            const uint32_t cend = cbase + cc;
            const uint32_t tshift = ocarry - cbase;  // tile index = window symbol index + tshift
            const uint32_t ce = oe < cend ? oe : cend;
            const bool mine = oi < ce;        // this lane emits symbols in this flush
            const uint32_t te = ce + tshift;  // one past its last tile byte of this flush
            // fast steps while at least 3 symbols are owed (three unconditional byte stores: what a lookup with fewer
            // symbols leaves behind in the lane's own bytes is overwritten by its next steps)
            const uint32_t te3 = (mine && te >= 3u) ? te - 2u : 0u;  // tB < te3 <=> tB + 3 <= te
            const uint32_t tee = mine ? te : 0u;
            uint32_t relB = mine ? bpos : 0xFFFFu, tB = oi + tshift;
            auto emit = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const unsigned long long pair = ((unsigned long long)R[k] << 32) | R[k + 1];
                auto step = [&](unsigned long long am) __attribute__((always_inline)) {
                    const uint32_t w = rw_hi_shl(pair, relB);
#if DCZ_RW_ABL & 4
                    uint32_t e = 0x89414243u + (w >> 31);  // ablation: no table read (2 symbols, 9 bits)
#else
                    uint32_t e = *(lds_cu32*)(uintptr_t)(mt_addr + ((w >> (30 - TBM)) & (uint32_t)(((1 << TBM) - 1) << 2)));
#endif
                    if (__builtin_expect((RW_EQ(e, 0u) & am) != 0ull, 0)) {
                        if (relB < 32u && tB < te3 && e == 0u) {
                            const uint32_t r = rw_slow<TBM>(L, w);
                            e = (r & 0xFFu) | ((r >> 8) << 24) | (1u << 30);
                            if (r == 0u) e = 0xFFu << 24;  // (unreachable: phase A counted only decodable symbols)
                        }
                    }
#if !(DCZ_RW_ABL & 2)
                    {   // lanes outside the mask store to the slack bytes past the tile instead of being masked off
                        uint8_t* const tp = ob + rw_select2(tB, (uint32_t)LdsT::CAP + 16u, am);
                        tp[0] = (uint8_t)e;
                        tp[1] = (uint8_t)(e >> 8);
                        tp[2] = (uint8_t)(e >> 16);
                    }
#endif
                    e = rw_select(e, am);
                    relB += (e >> 24) & 63u;
                    tB += e >> 30;
                };
#pragma unroll
                for (int u = 0; u < DCZ_RW_UNROLL; u++) step(RW_LT(relB, 32u) & RW_LT(tB, te3));
                while (true) {
                    const unsigned long long am = RW_LT(relB, 32u) & RW_LT(tB, te3);
                    if (am == 0ull) break;
                    step(am);
                }
                // the last one or two symbols of the lane in this flush: stores limited to what is owed
                while (true) {
                    const bool a = relB < 32u && tB < tee;
                    if (__builtin_amdgcn_ballot_w64(a) == 0ull) break;
                    const uint32_t w = rw_hi_shl(pair, relB);
                    uint32_t e = *(lds_cu32*)(uintptr_t)(mt_addr + ((w >> (30 - TBM)) & (uint32_t)(((1 << TBM) - 1) << 2)));
                    if (__builtin_amdgcn_ballot_w64(a && e == 0u) != 0ull) {
                        if (a && e == 0u) {
                            const uint32_t r = rw_slow<TBM>(L, w);
                            e = (r & 0xFFu) | ((r >> 8) << 24) | (1u << 30);
                            if (r == 0u) e = 0xFFu << 24;
                        }
                    }
                    if (a) {
                        const uint32_t c = e >> 30;
                        const uint32_t take = (c < tee - tB) ? c : tee - tB;
                        ob[tB] = (uint8_t)e;
                        if (take > 1u) ob[tB + 1u] = (uint8_t)(e >> 8);
                        if (take > 2u) ob[tB + 2u] = (uint8_t)(e >> 16);
                        relB += (e >> 24) & 63u;
                        tB += take;
                    }
                }
                relB -= 32u;
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (emit(std::integral_constant<int, Ks>{}), ...);
            }(std::make_integer_sequence<int, 9>{});
            if (mine) {  // back to absolute coordinates for the next flush of this window
                oi = tB - tshift;
                bpos = relB + 32u * 9u;
            }
            RW_T(5);
#if DCZ_K4_PROF
            pacc[10]++;
#endif
            __syncthreads();
            RW_T(6);
            const uint32_t total = ocarry + cc;
            const bool last = !more && cend == lim;  // final flush of the block: store the ragged tail too
            const uint32_t full = last ? total : (total & ~15u);
            uint8_t* const dst = oblk + gpos;
            const uint32_t nunits = (full + 15u) >> 4;
            for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                const uint32_t lo = u << 4;
                const uint32_t* src = &L.tile[lo >> 2];
                if (out_aligned && lo + 16u <= full) {
                    *reinterpret_cast<uint4*>(dst + lo) = make_uint4(src[0], src[1], src[2], src[3]);
                } else {
                    for (uint32_t i = lo; i < lo + 16u && i < full; i++) dst[i] = ob[i];
                }
            }
            const uint32_t tail = total - full;  // < 16
            uint8_t tv = 0;
            if ((uint32_t)tid < tail) tv = ob[full + tid];
            __syncthreads();
            if ((uint32_t)tid < tail) ob[tid] = tv;
            gpos += full;
            ocarry = tail;
            cbase = cend;
            RW_T(7);
        }
        produced += lim;
        ventry = next_ventry;
        if (exhausted && produced < orig) {
            // The payload is used up but the chunk wants more symbols: the reference keeps reading zero bits
            // (TableBasedHuffmanDecoder.java:204-208), i.e. the all-zero codeword = first canonical symbol, forever.
            __syncthreads();
            if ((uint32_t)tid < ocarry) oblk[gpos + tid] = ob[tid];  // unflushed tail (gpos + ocarry == produced)
            if (L.maxlen == 0) {  // empty table: no codeword at all
                status = DCZ_E_BADSTREAM;
                errpos = (long long)produced;
            } else {
                const uint8_t z = L.symtab[0];
                for (uint32_t i = produced + (uint32_t)tid; i < orig; i += W) oblk[i] = z;
            }
            break;
        }
        __syncthreads();
    }

    if (tid == 0) {
        d_status[b] = status;
        if (d_errpos) d_errpos[b] = errpos;
#if DCZ_K4_PROF
        for (int i = 0; i < 12; i++) atomicAdd(&rw_prof[i], pacc[i]);
#endif
    }
}

void launch_decode_regwin(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                          const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                          int32_t* d_status, int64_t* d_errpos, const DecodeWs& ws, bool few_blocks, hipStream_t s) {
    if (K == 0) return;
    const unsigned long long* off = reinterpret_cast<const unsigned long long*>(d_comp_off);
    long long* ep = reinterpret_cast<long long*>(d_errpos);
    if (!few_blocks)
        hipLaunchKernelGGL((k4_regwin<256, DCZ_RW_TBJ, DCZ_RW_TBM, DCZ_RW_OC>), dim3(K), dim3(256), 0, s, d_comp, off,
                           d_comp_size, d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls);
    else
        hipLaunchKernelGGL((k4_regwin<1024, DCZ_RW_TBJ, DCZ_RW_TBM, DCZ_RW_OCS>), dim3(K), dim3(1024), 0, s, d_comp, off,
                           d_comp_size, d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls);
}

}  // namespace dcz

#if DCZ_K4_PROF
extern "C" void dcz_debug_rw_prof(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(dcz::rw_prof), sizeof(dcz::rw_prof));
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(dcz::rw_prof), z, sizeof(z));
    }
}
#endif
