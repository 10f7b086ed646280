// k4_decode.hip -- K4: parallel table-walk Huffman decode without side information (gfx950).
//
// Replaces TableBasedHuffmanDecoder (core/TableBasedHuffmanDecoder.java:36-152: 10-bit table, bit-serial
// peek(), HashMap fallback for long codes core/CanonicalHuffman.java:161-229) as called from
// CpuCompressionService.decodeChunkParallel (service/cpu/CpuCompressionService.java:511-556).  The
// reference's GPU decode kernels run ONE work-item per chunk and are dead code
// (service/gpu/GpuCompressionService.java:1340-1469); GpuCompressionService.decompress delegates to the CPU (:858).
//
// The frozen format stores no intra-block offsets, so parallelism inside a
// block comes from self-synchronisation: the workgroup walks its block in windows of W*NS subsequences
// of 32 bytes; every thread owns NS consecutive subsequences (its "stripe") and decodes them
// interleaved: NS independent dependency chains per lane hide the LDS latency.  Per window:
//   A. every subsequence is decoded from a guessed entry bit and reports where the codeword that
//      crosses its end finishes; guesses are replaced by the left neighbour's exit until nothing
//      changes (the first subsequence's entry is exact, so the fixed point is the true parse; Huffman
//      streams resynchronise within a few codewords, typically 2-3 rounds);
//   B. a workgroup scan of the symbol counts gives every subsequence its output offset; the threads
//      decode again, writing bytes into an LDS staging tile that is flushed with aligned 16-byte
//      stores (a partial 16-byte tail is carried into the next flush).
// LDS layout of the compressed window: one stripe per thread = its NS*8 MSB-first dwords + the first
// two dwords of the next stripe (so a decode never leaves its stripe) + one pad dword: the odd stripe
// stride makes the per-lane window fetch bank-conflict free.  Dwords are stored in DESCENDING address
// order and streams track a descending bit position npos = 8*top + 31 - pos, so the fetch is
//   addr = (npos >> 3) & ~3;  {lo,hi} = ds_read2_b32(addr);  bits = {hi,lo} >> ((npos & 31) + k)
// i.e. two VALU for the address and no register swap for the 64-bit shift.
// The next window's global loads are issued before phase B and land in registers while it runs.
// HBM traffic is the algorithmic C + N per block.
// Decode table: 2^11 entries (len<<8 | symbol) in LDS; longer codes take the canonical
// first-code/count search, which for a prefix code returns exactly what the reference's
// 10-bit-table-then-HashMap path returns.  Bits past the end of the payload read as zero
// (TableBasedHuffmanDecoder.java:204-208); a missing code is "Huffman decode error at position i" (:109-111).
#include <cstdlib>
#include <utility>

#include "dcz_internal.h"

namespace dcz {

constexpr int SUB_BYTES = 32;
constexpr int SUB_BITS = SUB_BYTES * 8;
constexpr int SUB_DW = SUB_BYTES / 4;
#ifndef DCZ_K4_TB
#define DCZ_K4_TB 11
#endif
#ifndef DCZ_K4_W
#define DCZ_K4_W 256
#endif
// DCZ_K4_TB: first-level table bits of the long-code and short-code instantiations; DCZ_K4_TBM: of the medium class,
// whose blocks have many codewords of 12-13 bits (text: 16.5 -> 14.6 ms with 13 bits; the 16 KiB table costs that
// instantiation no occupancy, it is VGPR-limited to 4 waves/SIMD anyway)
#ifndef DCZ_K4_TBM
#define DCZ_K4_TBM 13
#endif
// Tuning knobs (measured on MI355X, 8 GiB uniform random, 1 MiB blocks: NS=1/OC=8 KiB/W=256 is fastest -- more
// resident workgroups hide LDS latency better than in-thread ILP).
#ifndef DCZ_K4_OC
#define DCZ_K4_OC 8192   // output staging bytes per flush, many-blocks kernel
#endif
#ifndef DCZ_K4_NS
#define DCZ_K4_NS 1      // subsequences per thread, many-blocks kernel
#endif
#ifndef DCZ_K4_PRIV
#define DCZ_K4_PRIV 48   // symbols a subsequence parks in REGISTERS during phase A (PRIV/4 VGPRs), so that phase B
#endif                   // copies them into the tile instead of decoding again; 0 = off.  48 keeps the kernel at
                         // 96 VGPRs = 5 waves/SIMD without spills (64: 4 waves or spills; measured 6.95 vs 7.75 ms)
#ifndef DCZ_K4_PRIVM
#define DCZ_K4_PRIVM 96  // parking capacity of the instantiation for medium code lengths (3.6 .. 6.5 bits), 0 = none
#endif
#ifndef DCZ_K4S_PRIVM
#define DCZ_K4S_PRIVM 96
#endif
#ifndef DCZ_K4S_PRIV
#define DCZ_K4S_PRIV 64
#endif
#ifndef DCZ_K4M_OC
#define DCZ_K4M_OC 8192  // short-code instantiation, many-blocks kernel
#endif
#ifndef DCZ_K4_SPARSE
#define DCZ_K4_SPARSE 1  // short-code class: fill + single-byte stores for blocks dominated by a 1-bit symbol
#endif
#ifndef DCZ_K4L_OC
#define DCZ_K4L_OC 8192  // short-code instantiation (multi-symbol tables), many-blocks kernel
#endif
#ifndef DCZ_K4S_OC
#define DCZ_K4S_OC 32768 // few-blocks kernel (one 1024-thread workgroup per block owns the CU: use its LDS)
#endif
#ifndef DCZ_K4_MINWAVES
#define DCZ_K4_MINWAVES 5  // waves per SIMD the many-blocks kernel is compiled for (caps its VGPRs at 96)
#endif
#ifndef DCZ_K4_OUTLINE_SLOW
#define DCZ_K4_OUTLINE_SLOW 1  // parking loop calls the long-code search instead of inlining it in every step
#endif
#ifndef DCZ_K4_EXIT_EVERY
#define DCZ_K4_EXIT_EVERY 2  // measured over 3 runs each: 2 is ~1.5 % faster than 1; 4 and 8 are 35-45 % SLOWER
#endif
#ifndef DCZ_K4_GROUP
#define DCZ_K4_GROUP 0  // 1: a third copy of the unrolled steps for total tables with codewords <= 8 bits (one window fetch per
#endif                  // four symbols).  Its main customer -- 256 symbols of 8 bits -- is decoded by k4_fixed.hip now; without
                        // the copy the long-code instantiation spills 1 VGPR instead of 15 (scratch 16 B instead of 64 B per
                        // lane) at the same speed on every distribution measured (text, hi7, hi7_5, bin7_8, mid6).
#ifndef DCZ_K4_EXACT
#define DCZ_K4_EXACT 1        // exact-entry procedure for windows that do not self-synchronise
#endif
#ifndef DCZ_K4_EXACT_AFTER
#define DCZ_K4_EXACT_AFTER 12 // decode rounds a window may take before its block is handed to the exact-entry launch
                              // (measured: a block that needs ~7 rounds is still faster on the regular path)
#endif
#ifndef DCZ_K4_OCX
#define DCZ_K4_OCX 512   // tile bytes beyond OC: a window of 8-bit codes (W*32 symbols) plus a carried tail fits one flush
#endif
#ifndef DCZ_K4S_NS
#define DCZ_K4S_NS 1
#endif
#ifndef DCZ_K4_MEDIUM_DFA
#define DCZ_K4_MEDIUM_DFA 1  // medium class, one workgroup per block: k4_dfa.hip decodes the tables it can take
#endif
#ifndef DCZ_K4_TBS
#define DCZ_K4_TBS DCZ_K4_TB  // first-level table bits of the short-code (multi-symbol) instantiation
#endif
// class boundary medium / short codes: a block is of the medium class when orig * A <= csize * B (>= 8 * A / B bits per symbol)
#ifndef DCZ_K4_CLS2_A
#define DCZ_K4_CLS2_A 4
#endif
#ifndef DCZ_K4_CLS2_B
#define DCZ_K4_CLS2_B 9
#endif

template <int W, int NS, int OC, int PV, bool MULTI, int TBITS>
struct DecLds {
    static constexpr int TB = TBITS;  // index bits of the first-level decode table
    static_assert(W >= 64 && W % 64 == 0, "whole waves only");
    static constexpr int NSUB = W * NS;
    static constexpr int STRIPE = SUB_DW * NS;   // payload dwords per thread
    static constexpr int STRIDE = STRIPE + 3;    // + 2 look-ahead dwords + 1 pad (odd => conflict-free)
    __attribute__((aligned(16))) uint32_t cbuf[W * STRIDE + 4];
    // logical byte i of the output tile lives at i + 4 * (i >> 6): one pad dword per 64 bytes
    static constexpr int CAP = OC + DCZ_K4_OCX;  // tile capacity in bytes
    __attribute__((aligned(16))) uint32_t outbuf[(CAP + CAP / 16) / 4 + 32];
    // Phase A parks the first PRIV symbols of each subsequence in registers; after the fixed point the LAST decode
    // of a subsequence started at its true entry, so phase B ORs them into the (zeroed) tile as whole dwords
    // instead of decoding again.
    static constexpr int PRIV = PV;
    static constexpr bool ZT = PV > 0 || MULTI;  // the tile is kept zero outside the bytes already written
    static_assert(PRIV % 4 == 0, "whole registers");
    uint32_t cend_vote;
    uint32_t flag[3];
    uint16_t table[1 << TB];
    // short-code kernel (MULTI): per TB-bit window, the maximal run of complete codewords inside it
    //   mcount: (symbols << 4) | bits                      -> phase A skips several symbols per lookup
    //   mout:   s0 | s1 << 8 | s2 << 16 | bits << 24 | symbols << 28 (first <= 3 symbols) -> phase B
    uint16_t mcount[MULTI ? (1 << TB) : 1];
    uint32_t mout[MULTI ? (1 << TB) : 1];  // symbols > 3: a run of the first canonical symbol (bits = all of them)
    uint16_t exits[NSUB];
    unsigned long long lim[40];  // lim[l] = (first[l] + cnt[l]) << (32 - l): exclusive left-aligned upper bound of length l
    uint32_t first[34];
    uint32_t cnt[34];
    uint32_t offs[34];
    uint32_t wsum[W / 64];
    uint8_t symtab[256];
    uint8_t len8[256];
    uint32_t maxlen;
    uint32_t nomiss;  // complete code with maxlen <= TB: the 2^TB table has no escape entries
    uint32_t in_phase;  // one code length that divides SUB_BITS: entry offsets are known without decoding
    uint32_t dfa_takes;  // medium class: the nibble automaton (k4_dfa.hip) decodes this block
    uint32_t err_idx;
    int bad_table;
};

__device__ __forceinline__ uint32_t opad(uint32_t i) { return i + ((i >> 6) << 2); }

typedef __attribute__((address_space(3))) const uint32_t lds_cu32;

// 64-bit window {dword wi, dword wi+1} for descending position npos (see the layout note above).
__device__ __forceinline__ unsigned long long fetch64(uint32_t npos) {
    lds_cu32* p = (lds_cu32*)(uintptr_t)((npos >> 3) & ~3u);
    return (unsigned long long)p[0] | ((unsigned long long)p[1] << 32);
}
// next 32 stream bits at descending position npos: one v_alignbit over the dword pair that ends with the window's last bit
// (q = npos + 1: the pair is {dword holding the window's first bit or the one above it, dword holding its last bit}
// and the shift q & 31 is what v_alignbit takes from the low 5 bits of its operand)
__device__ __forceinline__ uint32_t window_q(uint32_t q) {
    lds_cu32* p = (lds_cu32*)(uintptr_t)((q >> 3) & ~3u);
    return __builtin_amdgcn_alignbit(p[1], p[0], q);
}
// w << ((e >> 8) & 0xFF) in one VALU op
__device__ __forceinline__ uint32_t shl_byte1(uint32_t w, uint32_t e) {
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
        : "=v"(r)
        : "v"(e), "v"(w));
    return r;
}
// a - ((e >> 8) & 0xFF) in one VALU op (SDWA byte select)
__device__ __forceinline__ uint32_t sub_byte1(uint32_t a, uint32_t e) {
    uint32_t r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
        : "=v"(r)
        : "v"(a), "v"(e));
    return r;
}
// lane in mask ? v : 0, and n + (lane in mask), with the wave mask in a scalar register pair
__device__ __forceinline__ uint32_t select_mask(uint32_t v, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t add_mask(uint32_t n, unsigned long long m) {
    uint32_t r;
    asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(r) : "v"(n), "s"(m) : "vcc");
    return r;
}
// byte offset into the u16 table of the TB-bit window at npos
template <int TB>
__device__ __forceinline__ uint32_t table_off(unsigned long long two, uint32_t npos) {
    return (uint32_t)(two >> ((npos & 31u) + (uint32_t)(32 - TB))) & (uint32_t)(((1 << TB) - 1) << 1);
}
// next 32 stream bits at npos (slow path only)
__device__ __forceinline__ uint32_t window32(unsigned long long two, uint32_t npos) {
    return (uint32_t)(two >> ((npos & 31u) + 1u));
}

// Codes longer than TB bits, and invalid patterns.  Canonical code ranges are contiguous and ascending in the
// length, and everything below the TB+1 range was already answered by the table, so the length of window w is
// the first l > TB with w < lim[l] (64-bit, left-aligned); four independent LDS reads per step.  No match = no
// codeword (the reference's "decode error at position i").
template <class LdsT>
__device__ __forceinline__ uint32_t slow_lookup(const LdsT& L, uint32_t win32) {
    constexpr int TB = LdsT::TB;
    const uint32_t maxlen = L.maxlen;
    const unsigned long long w = win32;
    for (uint32_t l = TB + 1; l <= maxlen; l += 4) {
        const unsigned long long a = L.lim[l], b = L.lim[l + 1], c = L.lim[l + 2], d = L.lim[l + 3];
        const uint32_t k = (w < a) ? 0u : (w < b) ? 1u : (w < c) ? 2u : (w < d) ? 3u : 4u;
        if (k < 4u) {
            const uint32_t ll = l + k;  // <= maxlen: lim[] is flat beyond maxlen, so a later length never wins
            return (ll << 8) | (uint32_t)L.symtab[L.offs[ll] + ((win32 >> (32u - ll)) - L.first[ll])];
        }
    }
    return 0;
}

// The same search as an out-of-line function on LDS addresses: the hand-unrolled parking loop would otherwise carry
// one inlined copy per step and outgrow the instruction cache.
typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
template <int TB>
__device__ __attribute__((noinline)) uint32_t slow_lookup_outlined(uint32_t lim_a, uint32_t first_a, uint32_t offs_a,
                                                                   uint32_t symtab_a, uint32_t maxlen, uint32_t win32) {
    lds_cu64* lim = (lds_cu64*)(uintptr_t)lim_a;
    lds_cu32* first = (lds_cu32*)(uintptr_t)first_a;
    lds_cu32* offs = (lds_cu32*)(uintptr_t)offs_a;
    lds_cu8* symtab = (lds_cu8*)(uintptr_t)symtab_a;
    const unsigned long long w = win32;
    for (uint32_t l = TB + 1; l <= maxlen; l += 4) {
        const unsigned long long a = lim[l], b = lim[l + 1], c = lim[l + 2], d = lim[l + 3];
        const uint32_t k = (w < a) ? 0u : (w < b) ? 1u : (w < c) ? 2u : (w < d) ? 3u : 4u;
        if (k < 4u) {
            const uint32_t ll = l + k;
            return (ll << 8) | (uint32_t)symtab[offs[ll] + ((win32 >> (32u - ll)) - first[ll])];
        }
    }
    return 0;
}

template <int W, class LdsT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, LdsT& L, uint32_t& total) {
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
__device__ unsigned long long k4_prof[12];  // [8] windows, [9] self-sync rounds (wave 0 of each workgroup)
#define PROF_T(i)                                         \
    do {                                                  \
        const unsigned long long t_ = clock64();          \
        pacc[i] += t_ - plast;                            \
        plast = t_;                                       \
    } while (0)
#else
#define PROF_T(i) do { } while (0)
#endif

// ---- exact entries for windows that do not self-synchronise ------------------------------------------------------
// Streams of (nearly) equal-length codewords whose length does not divide the subsequence keep a wrong phase for ever, so
// the fixed point of phase A advances one subsequence per round.  After DCZ_K4_EXACT_AFTER rounds a window switches to
// this procedure (and the rest of its block starts with it): every thread computes, for EVERY possible entry offset e of
// its subsequence (e < maxlen), the offset F[e] at which the parse from e leaves the subsequence -- descending in e, so
// that F[e] = F[e + len(e)] is a table read whenever the first codeword stays below maxlen, and only one parse per
// distinct track is decoded to the end -- then ONE lane walks the window: entry(q+1) = F_q[entry(q)].  The true entries
// go to L.exits[], the caller decodes once more from them.  F lives in the (unused in phase A) output tile, 32 bytes per
// subsequence from byte 64 on (the first 16 bytes of the tile hold the carried tail), and is zeroed again afterwards.
// All arguments that are addresses are LDS byte addresses.  Must be called by every thread of the workgroup.
template <int TB, int W>
__device__ __attribute__((noinline)) void k4_exact_entries(uint32_t ftab_a, uint32_t exits_a, uint32_t tbl_a, uint32_t lim_a,
                                                           uint32_t first_a, uint32_t offs_a, uint32_t symtab_a,
                                                           uint32_t maxlen, uint32_t nbase, uint32_t g0, uint32_t tid,
                                                           bool beyond) {
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    constexpr uint32_t QO = 32u - (uint32_t)TB;
    lds_u8* const frow = (lds_u8*)(uintptr_t)(ftab_a + 32u * tid);
    const uint32_t emax = maxlen < 32u ? (maxlen ? maxlen : 1u) : 32u;
    const uint32_t ql = nbase - (uint32_t)SUB_BITS + QO;  // limit of the subsequence in the q20 form
    // parse from stripe bit `start` (the first codeword is decoded by the caller when stop_below > 0): returns the exit
    auto run = [&](uint32_t start, uint32_t stop_below, bool active, uint32_t& landed) -> uint32_t {
        uint32_t q = active ? nbase - start + QO : ql;  // inactive lanes are past their limit from the beginning
        bool first = true;
        landed = 0xFFFFFFFFu;
        while (true) {
            unsigned long long am = __builtin_amdgcn_ballot_w64(q > ql);
            if (am == 0ull) break;
            uint32_t e = *(lds_u16*)(uintptr_t)(tbl_a + (window_q(q) & (uint32_t)(((1 << TB) - 1) << 1)));
            asm("" : "+v"(e));
            if ((__builtin_amdgcn_ballot_w64(e == 0u) & am) != 0ull) {
                bool dead = false;
                if (q > ql && e == 0u) {
                    e = slow_lookup_outlined<TB>(lim_a, first_a, offs_a, symtab_a, maxlen, window_q(q - (QO - 1u)));
                    if (e == 0u) {  // no codeword on this parse: it ends here (any exit will do, the true chain never
                        q = ql;     // follows it unless the stream is damaged, which the final decode reports)
                        dead = true;
                    }
                }
                am &= ~__builtin_amdgcn_ballot_w64(dead);
            }
            e = select_mask(e, am);
            q = sub_byte1(q, e);
            if (first) {  // position after the first codeword; a lane whose parse lands below stop_below is done
                first = false;
                const uint32_t pos = nbase + QO - q;
                if (active && pos < stop_below) {
                    landed = pos;
                    q = ql;
                }
            }
        }
        return (nbase + QO - q) - (uint32_t)SUB_BITS;
    };
    for (uint32_t e = emax; e-- > 0u;) {  // descending: F[e + len] is known when it is needed
        uint32_t landed;
        const uint32_t x = run(e, emax, !beyond, landed);
        if (!beyond) frow[e] = (uint8_t)((landed != 0xFFFFFFFFu) ? (uint32_t)frow[landed] : (x & 31u));
    }
    uint32_t x0 = 0;
    {   // the window's first subsequence enters at g0 (up to 127 bits in): its exit is decoded directly
        uint32_t landed;
        x0 = run(g0, 0u, tid == 0u && !beyond, landed) & 31u;
    }
    __syncthreads();
#if DCZ_K4_PROF
    const unsigned long long pc0 = clock64();
#endif
    // entry(q + 1) = F_q[entry(q)] over the window's W subsequences.  One lane walking all of them was 29 % of K4's time on
    // such streams (336 cycles per dependent step, tools/k4prof_dist.py), so every wave first composes its own 64
    // subsequences for EVERY possible entry (lane = entry offset at the wave's first subsequence; the composed row
    // replaces F_q in place: a wave's LDS operations execute in order, the reads of a step precede its writes), then
    // the W/64 wave entries are chained, then every subsequence looks its exit up.
    {
        lds_u8* const ft = (lds_u8*)(uintptr_t)ftab_a;
        const uint32_t w0 = tid & ~63u, lane = tid & 63u;
        if (lane < 32u) {
            uint32_t cur = lane;
            for (uint32_t k = (w0 == 0u) ? 1u : 0u; k < 64u; k++) {  // (subsequence 0's exit is x0, decoded above)
                const uint32_t q = w0 + k;
                cur = (uint32_t)ft[32u * q + cur] & 31u;
                ft[32u * q + lane] = (uint8_t)cur;
            }
        }
        __syncthreads();
        // entry of every wave's first subsequence (of subsequence 1 for wave 0): x0, then the composed last rows
        __attribute__((address_space(3))) uint16_t* const ex = (__attribute__((address_space(3))) uint16_t*)(uintptr_t)exits_a;
        if (tid == 0u) ex[0] = (uint16_t)x0;  // (only thread 0's x0 is real)
        __syncthreads();
        uint32_t ent = (uint32_t)ex[0];
        for (uint32_t w = 64u; w <= w0; w += 64u) ent = (uint32_t)ft[32u * (w - 1u) + ent] & 31u;
        if (tid != 0u) ex[tid] = (uint16_t)((uint32_t)ft[32u * tid + ent] & 31u);
#if DCZ_K4_PROF
        if (tid == 0u) atomicAdd(&k4_prof[11], clock64() - pc0);  // the chain over the window's subsequences
#endif
    }
    __syncthreads();
    {
        __attribute__((address_space(3))) uint32_t* const z = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(ftab_a + 32u * tid);
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = 0u;
    }
    __syncthreads();
}


// CMASK: the block classes this instantiation decodes (bit 2: >= 6.5 bits per symbol on average, bit 1: at most 72
// symbols per 32-byte subsequence, bit 0: shorter codes); every launch covers all blocks and each workgroup leaves at
// once unless its block is of a class it owns.
// MODE 0: the regular decoder (leaves at once if its block is flagged in d_slow).
// MODE 2: the probe, launched first: table build and phase A of the block's FIRST window only; a block whose window is not
//         synchronised after DCZ_K4_EXACT_AFTER rounds is flagged in d_slow.
// MODE 1: the exact-entry decoder, launched last: decodes the flagged blocks with k4_exact_entries in every window.
// MODE 3: the regular decoder run once per REGION of a block that k4_split.hip has cut up (sdp: the region table).
// Keeping the probe's give-up test and the exact-entry call out of the regular kernels keeps their register allocation
// what it was (with either inside, the uniform case lost 9-14 %).
template <int W, int NS, int OC, int PV, bool MULTI, int CMASK, int TBITS, int MODE>
__global__ __launch_bounds__(W, MODE == 3 ? 4 : (W <= 256 && !MULTI) ? (PV <= 48 ? DCZ_K4_MINWAVES : 4) : 1) void k4_decode(const uint8_t* __restrict__ comp, size_t comp_bytes,
                                               const unsigned long long* __restrict__ d_comp_off,
                                               const uint32_t* __restrict__ d_comp_size,
                                               const uint32_t* __restrict__ d_orig_size,
                                               const uint8_t* __restrict__ d_len, size_t out_stride,
                                               uint8_t* __restrict__ out, int32_t* __restrict__ d_status,
                                               long long* __restrict__ d_errpos, uint8_t* __restrict__ d_slow,
                                               const SplitDesc* __restrict__ sdp) {
    using LdsT = DecLds<W, NS, OC, PV, MULTI, TBITS>;
    __shared__ LdsT L;
    constexpr int TB = TBITS;
    constexpr int NCH = 2 * NS;  // 16-byte chunks per thread
    uint32_t b = blockIdx.x;
    const int tid = (int)threadIdx.x;
    constexpr bool XM = MODE == 1;
    // class byte of the block (k4_classify, then the probe / k4_split_scan): 0 = table walk, one workgroup per block,
    // 1 = exact-entry launch, 2 = table walk, one workgroup per REGION (k4_split.hip), 0x10 | L = fixed-length code
    // (k4_fixed), 0xFF = footer fields outside the buffers (nobody touches the block)
    constexpr bool split = MODE == 3;  // MODE 3 = MODE 0 with one workgroup per (block, region): own instantiations, so
                                       // that the per-block kernels keep their register allocation
    uint32_t reg = 0;
    if constexpr (split) {
        if (!split_region_of(sdp, blockIdx.x, b, reg)) return;
        if (d_slow[b] != 2 || reg >= sdp->nreg[b]) return;  // workgroup-uniform
    } else if constexpr (MODE == 1) {
        if (d_slow[b] != 1) return;  // workgroup-uniform: only the blocks the probe flagged
    } else {
        if (d_slow[b] != 0) return;  // another launch owns this block
    }

    // ---- block geometry ----
    const uint32_t orig_blk = d_orig_size[b];
    const unsigned long long coff = d_comp_off[b];
    const uint32_t csize = d_comp_size[b];
    // symbols this workgroup produces: the whole chunk, or what k4_split_scan gave its region
    const uint32_t orig = split ? sdp->count[(uint64_t)b * sdp->rmax + reg] : orig_blk;
    if (split && orig == 0u) return;
    {
        const int cls = ((unsigned long long)csize * 16ull >= (unsigned long long)orig_blk * 13ull) ? 4
                        : ((unsigned long long)orig_blk * (unsigned long long)DCZ_K4_CLS2_A <= (unsigned long long)csize * (unsigned long long)DCZ_K4_CLS2_B) ? 2 : 1;
        if ((cls & CMASK) == 0) return;  // workgroup-uniform; another launch owns this block
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
#if DCZ_K4_MEDIUM_DFA
        if constexpr (CMASK == 2 && MODE != 1) {  // (the probe too: k4_dfa gives a block up by itself)
            // blocks the nibble automaton takes (k4_dfa.hip applies the same test to the same table): no 1-bit codeword
            // and at most 255 internal nodes in the code tree; this kernel keeps the rest of the medium class
            uint32_t ni = 0, states = 0;
            for (int l = 31; l >= 0; l--) {
                ni = (L.cnt[l + 1] + ni + 1u) / 2u;
                states += ni;
            }
            L.dfa_takes = (L.cnt[1] == 0u && states <= 255u && mx > 0u && kraft <= (1ull << 32)) ? 1u : 0u;
        }
#endif
#if DCZ_K4_SPARSE_DFA
        if constexpr (CMASK == 1 && MODE == 0) {
            // sparse blocks the automaton takes (k4_dfa.hip, SPARSE instantiation, same test on the same numbers): one
            // 1-bit codeword, < 1.3 bits per symbol, at most 255 internal nodes
            uint32_t ni = 0, states = 0;
            for (int l = 31; l >= 0; l--) {
                ni = (L.cnt[l + 1] + ni + 1u) / 2u;
                states += ni;
            }
            L.dfa_takes = (L.cnt[1] == 1u && states <= 255u && mx >= 2u && kraft <= (1ull << 32) &&
                           (unsigned long long)csize * 80ull < (unsigned long long)orig_blk * 13ull) ? 1u : 0u;
        }
#endif
        L.nomiss = (mx >= 1u && mx <= (uint32_t)TB && kraft == (1ull << 32)) ? 1u : 0u;
        if (kraft > (1ull << 32)) L.bad_table = 1;  // not a prefix code
        // every codeword has the same length and that length divides the subsequence: always in phase
        L.in_phase = (mx >= 1u && L.cnt[mx] > 0u && kraft == ((unsigned long long)L.cnt[mx] << (32u - mx)) &&
                      ((uint32_t)SUB_BITS % mx) == 0u) ? 1u : 0u;
    }
    __syncthreads();
    if (L.bad_table) {
        if (tid == 0) {
            d_status[b] = DCZ_E_BADTABLE;
            if (d_errpos) d_errpos[b] = 0;
        }
        return;
    }
#if DCZ_K4_MEDIUM_DFA
    if constexpr (CMASK == 2 && MODE != 1) {  // (the probe too: k4_dfa gives a block up by itself)
        if (L.dfa_takes) return;  // workgroup-uniform
    }
#endif
#if DCZ_K4_SPARSE_DFA
    if constexpr (CMASK == 1 && MODE == 0) {
        if (L.dfa_takes) return;  // workgroup-uniform
    }
#endif
    if constexpr (MODE == 2) {
        if (L.in_phase) return;  // the probe has nothing to find out (e.g. 256 symbols of 8 bits)
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

    if constexpr (LdsT::ZT) {  // the tile is OR-ed into / runs of zero bytes are skipped: start from zero
        for (int i = tid; i < (int)(sizeof(L.outbuf) / 4); i += W) L.outbuf[i] = 0;
    }
    if constexpr (MULTI) {
        __syncthreads();  // L.table complete
        for (int idx = tid; idx < (1 << TB); idx += W) {
            uint32_t pos = 0, cnt = 0, out = 0, bits3 = 0, cnt3 = 0;
            bool allz = true;  // every codeword of the window is the first canonical symbol (the shortest code)
            const uint32_t z = L.symtab[0];
            while (pos < (uint32_t)TB) {
                const uint32_t e = L.table[((uint32_t)idx << pos) & ((1u << TB) - 1u)];
                const uint32_t len = e >> 8;
                if (e == 0 || len > (uint32_t)TB - pos) break;  // escape, or the codeword leaves the window
                allz = allz && (e & 0xFFu) == z;
                if (cnt < 3) {
                    out |= (e & 0xFFu) << (8 * cnt);
                    bits3 = pos + len;
                    cnt3 = cnt + 1;
                }
                pos += len;
                cnt++;
            }
            L.mcount[idx] = (uint16_t)(cnt ? ((cnt << 4) | pos) : 0u);
            // a window that holds more than 3 codewords, all of them the shortest code: one RUN entry (count > 3)
            if (cnt > 3u && allz) L.mout[idx] = out | (pos << 24) | (cnt << 28);
            else L.mout[idx] = cnt3 ? (out | (bits3 << 24) | (cnt3 << 28)) : 0u;
        }
    }
    const uint32_t zsym = L.maxlen ? L.symtab[0] : 0u;           // first canonical symbol and its length: the run
    const uint32_t zlen = L.maxlen ? L.len8[zsym] : 1u;          // entries of mout
    // block-uniform: a 1-bit symbol and < 1.3 bits per symbol on average (from the block's own sizes)
    const bool sparse = MULTI && DCZ_K4_SPARSE && zlen == 1u && L.maxlen > 1u &&
                        (unsigned long long)csize * 80ull < (unsigned long long)orig_blk * 13ull;
    // output of this workgroup: the chunk's slot, or the region's offset inside it.  A region starts at any byte: the
    // tile is then laid over the 16-byte unit that holds its first byte, whose first hskip bytes belong to the
    // region before and are never stored from here.
    uint8_t* const oblk = out + (uint64_t)b * out_stride + (split ? sdp->off[(uint64_t)b * sdp->rmax + reg] : 0u);
    uint32_t hskip = (split && !sparse) ? (uint32_t)(((uintptr_t)oblk) & 15u) : 0u;
    uint8_t* const obase = oblk - hskip;  // address of tile byte 0 of the first flush
    const bool out_aligned = (((uintptr_t)obase) & 15u) == 0u;
    // virtual byte 0 = 16-byte aligned address at or below the payload start
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)coff;
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew;                              // first valid virtual byte
    const unsigned long long vhi = (unsigned long long)skew + csize;  // one past the last valid virtual byte
    (void)comp_bytes;  // checked against coff + csize by k4_classify

    // Park symbols only when a 32-byte subsequence is expected to hold at most 3/4 of the register capacity
    // (expected symbols = 32 * orig / csize from the block's own sizes); otherwise too many subsequences
    // overflow and are decoded twice anyway.
    const bool park = LdsT::PRIV > 0 &&
                      (unsigned long long)orig_blk * 128ull <= (unsigned long long)csize * 3ull * (unsigned long long)LdsT::PRIV;
    bool slow_block = XM;                     // exact-entry instantiation: every window starts from exact entries
    // virtual bit of the next codeword boundary: the payload start, or the region's proven entry
    unsigned long long ventry = split ? 8ull * reg * sdp->region_bytes + sdp->entry[(uint64_t)b * sdp->rmax + reg] : 8ull * skew;
    uint32_t produced = 0;                    // symbols decoded so far
    uint32_t gpos = 0;                        // offset of tile byte 0 from obase (multiple of 16)
    uint32_t ocarry = hskip;                  // bytes at the front of the tile not to be stored from it now (0..15)
    int status = DCZ_OK;
    long long errpos = 0;

    uint32_t* const cb = &L.cbuf[(uint32_t)tid * (uint32_t)LdsT::STRIDE];  // this thread's stripe (descending)
    uint8_t* const ob = reinterpret_cast<uint8_t*>(L.outbuf);
    constexpr int NR = LdsT::PRIV > 0 ? LdsT::PRIV / 4 : 1;
    uint32_t R[NS][NR];  // parked symbols of each stream, 4 per register, first symbol in the low byte
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int j = 0; j < NR; j++) R[s][j] = 0;
    // descending-position origin: logical dword j of the stripe lives at cb[STRIPE + 1 - j]
    const uint32_t top_addr =
        (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(cb + LdsT::STRIPE));
    const uint32_t nbase = 8u * top_addr + 31u;  // npos = nbase - pos
    const uint32_t tbl_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.table[0]));
    const uint32_t lim_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned long long*)(&L.lim[0]));
    const uint32_t first_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.first[0]));
    const uint32_t offs_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.offs[0]));
    const uint32_t symtab_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)(&L.symtab[0]));
    const uint32_t mcount_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.mcount[0]));
    const uint32_t mout_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.mout[0]));

    // prefetch registers for the window at wchunk0 (+ the look-ahead chunk, last thread only)
    uint4 pre[NCH];
    uint4 pre_m = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](unsigned long long wchunk0) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
            pre[c] = load_chunk16(vbase, (wchunk0 + (unsigned long long)(tid * NCH + c)) << 4, vlo, vhi);
        if (tid == W - 1) pre_m = load_chunk16(vbase, (wchunk0 + (unsigned long long)(W * NCH)) << 4, vlo, vhi);
    };
    if (orig > 0) prefetch(ventry >> 7);

#if DCZ_K4_PROF
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = clock64();
#endif
    while (produced < orig) {
        const unsigned long long wchunk0 = ventry >> 7;
        const uint32_t g0 = (uint32_t)(ventry - (wchunk0 << 7));

        // registers -> LDS stripes (descending dword order); the first two dwords are duplicated at the low
        // end of the previous stripe as its look-ahead
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            pre[c] = make_uint4(bswap32(pre[c].x), bswap32(pre[c].y), bswap32(pre[c].z), bswap32(pre[c].w));
            cb[LdsT::STRIPE + 1 - (4 * c + 0)] = pre[c].x;
            cb[LdsT::STRIPE + 1 - (4 * c + 1)] = pre[c].y;
            cb[LdsT::STRIPE + 1 - (4 * c + 2)] = pre[c].z;
            cb[LdsT::STRIPE + 1 - (4 * c + 3)] = pre[c].w;
        }
        if (tid > 0) {
            cb[1 - LdsT::STRIDE] = pre[0].x;  // previous stripe, logical dword STRIPE
            cb[0 - LdsT::STRIDE] = pre[0].y;  // previous stripe, logical dword STRIPE + 1
        }
        if (tid == W - 1) {
            cb[1] = bswap32(pre_m.x);
            cb[0] = bswap32(pre_m.y);
        }
        __syncthreads();
        PROF_T(1);

        // ---- phase A: self-synchronisation (positions are stripe-local bit offsets) ----
        uint32_t g[NS], x[NS], nsym[NS];
        bool bad[NS], need[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            g[s] = 0;
            x[s] = 0;
            nsym[s] = 0;
            bad[s] = false;
            need[s] = true;
        }
        if (tid == 0) g[0] = g0;
        const uint32_t q0 = (uint32_t)tid * NS;  // first subsequence of this thread
        // Subsequences that START past the payload hold nothing but zero padding: a periodic stream of the shortest
        // code that never self-synchronises when its length does not divide 256 (measured: W sync rounds in the last
        // window of every text block).  They take no part in phase A; the symbols the reference would read from
        // that padding are filled in after the window (see "exhausted").
        const unsigned long long wbase_bits = wchunk0 << 7;
        const unsigned long long pay_end_bits = vhi << 3;
        const bool exhausted = wbase_bits + (unsigned long long)LdsT::NSUB * SUB_BITS >= pay_end_bits;
        bool beyond[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) beyond[s] = false;
        if (exhausted) {  // workgroup-uniform: only the last window(s) of a block
#pragma unroll
            for (int s = 0; s < NS; s++) {
                beyond[s] = wbase_bits + (unsigned long long)(q0 + s) * SUB_BITS >= pay_end_bits;
                if (beyond[s]) need[s] = false;
            }
        }
        uint32_t round = 0;
#if DCZ_K4_PROF
        pacc[8]++;
#endif
        bool exact_done = false;
        auto take_exact_entries = [&]() __attribute__((always_inline)) {
            if constexpr (XM) {
                static_assert(!MULTI && NS == 1, "exact-entry instantiation");
                static_assert(sizeof(L.outbuf) >= 64 + 32 * (size_t)W, "entry-to-exit tables fit the output tile");
                const uint32_t ftab_a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.outbuf[16]));
                const uint32_t exits_a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.exits[0]));
                k4_exact_entries<TB, W>(ftab_a, exits_a, tbl_addr, lim_addr, first_addr, offs_addr, symtab_addr, L.maxlen,
                                        nbase, g0, (uint32_t)tid, beyond[0]);
                const uint32_t ng = (tid == 0) ? g0 : (uint32_t)L.exits[tid - 1];
                // (need[0] still set = the lane's registers belong to an older entry than g[0])
                need[0] = (need[0] || ng != g[0] || round == 0u) && !beyond[0];
                g[0] = ng;
                exact_done = true;
                slow_block = true;
            }
        };
#if DCZ_K4_PROF
        const unsigned long long px0 = clock64();
#endif
        if (slow_block) take_exact_entries();  // workgroup-uniform: an earlier window of this block did not converge
#if DCZ_K4_PROF
        pacc[10] += clock64() - px0;  // (inside "A decode" of tools/k4prof.py: the exact-entry procedure)
        plast = clock64();
#endif
        while (true) {
            // Integer-only inner loop.  Per stream: np = descending bit position, nl = position of its limit
            // (stream active <=> np > nl; nl = ~0 parks it), cnt = symbols decoded.  All NS window fetches are
            // issued back to back, then all table reads, then predicated updates; errors and codes longer than TB
            // bits are handled inside a wave-uniform, rarely taken branch.
            uint32_t np[NS], nl[NS];
            bool any = false;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t start = (uint32_t)s * SUB_BITS + g[s];
                np[s] = nbase - (need[s] ? start : 0u);
                nl[s] = need[s] ? nbase - (uint32_t)(s + 1) * SUB_BITS : 0xFFFFFFFFu;
                if (need[s]) {
                    nsym[s] = 0;
                    bad[s] = false;
                }
                any |= np[s] > nl[s];
            }
            if constexpr (MULTI) {
                // short codes: while at least TB bits remain before the limit, one lookup in mcount consumes every
                // complete codeword of the window (all of them start before the limit); the last < TB bits and
                // escapes go one symbol at a time
                static_assert(!MULTI || NS == 1, "short-code kernel: one subsequence per thread");
                while (any) {
                    const unsigned long long two = fetch64(np[0]);
                    const uint32_t off = table_off<TB>(two, np[0]);
                    const bool a = np[0] > nl[0];
                    const bool far = a && (np[0] - nl[0] >= (uint32_t)TB);
                    uint32_t e = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)((far ? mcount_addr : tbl_addr) + off);
                    uint32_t bits = far ? (e & 15u) : (e >> 8);
                    uint32_t cnt = far ? (e >> 4) : 1u;
                    if (__builtin_amdgcn_ballot_w64(a && e == 0) != 0ull) {
                        if (a && e == 0) {
                            e = slow_lookup(L, window32(two, np[0]));
                            bits = e >> 8;
                            cnt = 1;
                            if (e == 0) {
                                bad[0] = true;
                                nl[0] = 0xFFFFFFFFu;
                            }
                        }
                    }
                    const bool a2 = np[0] > nl[0];
                    np[0] -= a2 ? bits : 0u;
                    nsym[0] += a2 ? cnt : 0u;
                    any = np[0] > nl[0];
                }
            }
            if constexpr (LdsT::PRIV > 0) {
                if (park) {  // workgroup-uniform
                    // Hand-unrolled: symbol k of every stream lands in a compile-time register; the fold leaves
                    // (wave-uniformly) as soon as no lane has an active stream.  Per-lane predication: a stream that
                    // does not decode (any more) inserts nothing (identity selector when it keeps its registers from
                    // an earlier round, a zero byte past its end otherwise) and moves by zero bits.
                    // q20 = descending position + 21: window_q(q20) is the 32 bits that END 12 bits past the
                    // position, so the TB-bit table index sits at bits [TB:1] of the aligned pair and one v_and
                    // yields the byte offset of the u16 entry.
                    constexpr uint32_t QO = 32u - (uint32_t)TB;  // q20 = descending position + QO (21 for the 11-bit table)
                    uint32_t q20[NS], ql20[NS], selv[NS][4];
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        if (need[s]) {
#pragma unroll
                            for (int j = 0; j < NR; j++) R[s][j] = 0;
                        }
                        q20[s] = np[s] + QO;
                        ql20[s] = (nl[s] == 0xFFFFFFFFu) ? 0xFFFFFFFFu : nl[s] + QO;
                        selv[s][0] = need[s] ? 0x03020104u : 0x03020100u;
                        selv[s][1] = need[s] ? 0x03020400u : 0x03020100u;
                        selv[s][2] = need[s] ? 0x03040100u : 0x03020100u;
                        selv[s][3] = need[s] ? 0x04020100u : 0x03020100u;
                    }
                    // Lane activity is kept as wave masks in scalar registers (am[s]); the per-lane selects and the
                    // symbol count read them directly (v_cndmask / v_addc with a scalar mask operand).
                    // nomiss: the table answers every TB-bit pattern (complete code, no codeword longer than TB bits), so
                    // the escape test and its branch are compiled out of the steps
                    auto step = [&](auto kc, auto nm) __attribute__((always_inline)) -> bool {
                        constexpr int k = decltype(kc)::value;
                        constexpr bool NOMISS = decltype(nm)::value;
                        unsigned long long am[NS], any_m = 0;
#pragma unroll
                        for (int s = 0; s < NS; s++) {
                            am[s] = __builtin_amdgcn_ballot_w64(q20[s] > ql20[s]);
                            any_m |= am[s];
                        }
                        // the early exit is only an optimisation (finished lanes are predicated off): test it every
                        // DCZ_K4_EXIT_EVERY steps
                        if constexpr ((k % DCZ_K4_EXIT_EVERY) == 0) {
                            if (any_m == 0ull) return false;
                        }
                        uint32_t e[NS];
                        unsigned long long miss_m = 0;
#pragma unroll
                        for (int s = 0; s < NS; s++) {
                            e[s] = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr + (window_q(q20[s]) & (uint32_t)(((1 << TB) - 1) << 1)));
                            asm("" : "+v"(e[s]));  // a plain 32-bit value from here on (no 16-bit compare + re-extension)
                            if constexpr (!NOMISS) miss_m |= __builtin_amdgcn_ballot_w64(e[s] == 0u) & am[s];
                        }
                        if (!NOMISS && miss_m != 0ull) {  // rare: long codeword or no codeword
#pragma unroll
                            for (int s = 0; s < NS; s++) {
                                bool dead = false;
                                if (q20[s] > ql20[s] && e[s] == 0u) {
#if DCZ_K4_OUTLINE_SLOW
                                    e[s] = slow_lookup_outlined<TB>(lim_addr, first_addr, offs_addr, symtab_addr, L.maxlen,
                                                                    window_q(q20[s] - (QO - 1u)));
#else
                                    e[s] = slow_lookup(L, window_q(q20[s] - (QO - 1u)));
#endif
                                    if (e[s] == 0u) {
                                        bad[s] = true;
                                        ql20[s] = 0xFFFFFFFFu;
                                        dead = true;
                                    }
                                }
                                am[s] &= ~__builtin_amdgcn_ballot_w64(dead);
                            }
                        }
#pragma unroll
                        for (int s = 0; s < NS; s++) {
                            e[s] = select_mask(e[s], am[s]);
                            R[s][k >> 2] = __builtin_amdgcn_perm(e[s], R[s][k >> 2], selv[s][k & 3]);
                            q20[s] = sub_byte1(q20[s], e[s]);
                            nsym[s] = add_mask(nsym[s], am[s]);
                        }
                        return true;
                    };
                    // (only the instantiations for long codes get the second copy of the unrolled steps: blocks of the
                    // medium class practically always have codewords longer than TB bits)
#if DCZ_K4_GROUP
                    // Total table and no codeword longer than 8 bits: one 32-bit window holds four whole codewords, so
                    // the window is fetched once per four symbols and shifted in a register in between -- three of
                    // four steps have one LDS round trip (the table) instead of two on their dependency chain.
                    if (NS == 1 && LdsT::PRIV <= 64 && L.nomiss && L.maxlen <= 8u) {  // block-uniform
                        auto quad = [&](auto kc) __attribute__((always_inline)) -> bool {
                            constexpr int k = 4 * decltype(kc)::value;
                            if (__builtin_amdgcn_ballot_w64(q20[0] > ql20[0]) == 0ull) return false;
                            uint32_t w = window_q(q20[0] - (QO - 1u));  // the 32 bits at the position
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const unsigned long long am = __builtin_amdgcn_ballot_w64(q20[0] > ql20[0]);
                                uint32_t e = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr + ((w >> (31 - TB)) & (uint32_t)(((1 << TB) - 1) << 1)));
                                asm("" : "+v"(e));
                                e = select_mask(e, am);
                                R[0][(k + j) >> 2] = __builtin_amdgcn_perm(e, R[0][(k + j) >> 2], selv[0][(k + j) & 3]);
                                q20[0] = sub_byte1(q20[0], e);
                                nsym[0] = add_mask(nsym[0], am);
                                if (j < 3) w = shl_byte1(w, e);
                            }
                            return true;
                        };
                        [&]<int... Is>(std::integer_sequence<int, Is...>) {
                            (void)(quad(std::integral_constant<int, Is>{}) && ...);
                        }(std::make_integer_sequence<int, LdsT::PRIV / 4>{});
                    } else
#endif
                    if (LdsT::PRIV <= 64 && L.nomiss) {  // block-uniform
                        [&]<int... Is>(std::integer_sequence<int, Is...>) {
                            (void)(step(std::integral_constant<int, Is>{}, std::true_type{}) && ...);
                        }(std::make_integer_sequence<int, LdsT::PRIV>{});
                    } else {
                        [&]<int... Is>(std::integer_sequence<int, Is...>) {
                            (void)(step(std::integral_constant<int, Is>{}, std::false_type{}) && ...);
                        }(std::make_integer_sequence<int, LdsT::PRIV>{});
                    }
                    any = false;
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        np[s] = q20[s] - QO;
                        nl[s] = (ql20[s] == 0xFFFFFFFFu) ? 0xFFFFFFFFu : ql20[s] - QO;
                        any |= np[s] > nl[s];
                    }
                }
            }
            while (!MULTI && any) {
                unsigned long long two[NS];
                uint32_t e[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) two[s] = fetch64(np[s]);
#pragma unroll
                for (int s = 0; s < NS; s++)
                    e[s] = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr +
                                                                                         table_off<TB>(two[s], np[s]));
                bool miss = false;
#pragma unroll
                for (int s = 0; s < NS; s++) miss |= (e[s] == 0 && np[s] > nl[s]);
                if (__builtin_amdgcn_ballot_w64(miss) != 0ull) {
#pragma unroll
                    for (int s = 0; s < NS; s++)
                        if (e[s] == 0 && np[s] > nl[s]) {
                            e[s] = slow_lookup(L, window32(two[s], np[s]));
                            if (e[s] == 0) {  // no codeword matches: stop this stream
                                bad[s] = true;
                                nl[s] = 0xFFFFFFFFu;
                            }
                        }
                }
                any = false;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const bool a = np[s] > nl[s];
                    np[s] -= a ? (e[s] >> 8) : 0u;
                    nsym[s] += a ? 1u : 0u;
                    any |= np[s] > nl[s];
                }
            }
            PROF_T(0);
#pragma unroll
            for (int s = 0; s < NS; s++)  // final pos >= limit (or the start itself when it lies past the limit)
                if (need[s]) x[s] = bad[s] ? 0u : (nbase - np[s]) - (uint32_t)(s + 1) * SUB_BITS;
#pragma unroll
            for (int s = 0; s < NS; s++) L.exits[q0 + s] = (uint16_t)x[s];
            // One barrier per round.  Whether anybody had to decode in THIS round was recorded in flag[round % 3]
            // during the previous round's check; it becomes visible with this round's barrier.  If nobody had to,
            // every exit is unchanged and the entries are final.  (Slot (round + 1) % 3 is cleared before the
            // barrier, set after it and read after the next one.)
            if (tid == 0) L.flag[(round + 1u) % 3u] = 0;
            __syncthreads();
            PROF_T(2);
            if (exact_done) break;  // the entries were exact: this decode was the final one
            if (round > 0u && L.flag[round % 3u] == 0u) break;
            bool anyneed = false;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t q = q0 + s;
                const uint32_t ng = (q == 0) ? g0 : (uint32_t)L.exits[q - 1];
                need[s] = (ng != g[s]) && !beyond[s];
                g[s] = ng;
                anyneed |= need[s];
            }
            if (__builtin_amdgcn_ballot_w64(anyneed) != 0ull && (tid & 63) == 0) L.flag[(round + 1u) % 3u] = 1;
            round++;
#if DCZ_K4_PROF
            pacc[9]++;
#endif
            if constexpr (MODE == 2) {
                if (round == (uint32_t)DCZ_K4_EXACT_AFTER) {  // workgroup-uniform: this block does not self-synchronise
                    if (tid == 0) d_slow[b] = 1;
                    return;
                }
            }
        }

        if constexpr (MODE == 2) return;  // the probe only wanted to know whether the first window synchronises

        // ---- offsets, errors ----
        uint32_t tsum = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) tsum += nsym[s];
        uint32_t tw = 0;
        const uint32_t o = block_exclusive_scan<W>(tsum, L, tw);
        const uint32_t remaining = orig - produced;
        {
            uint32_t oo = o;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (bad[s]) atomicMin(&L.err_idx, oo + nsym[s]);
                oo += nsym[s];
            }
        }
        const unsigned long long next_ventry =
            (wchunk0 << 7) + (unsigned long long)LdsT::NSUB * SUB_BITS + (unsigned long long)L.exits[LdsT::NSUB - 1];
        __syncthreads();
        PROF_T(3);
        const uint32_t err_idx = L.err_idx;
        if (err_idx < remaining) {
            status = DCZ_E_BADSTREAM;
            errpos = (long long)produced + (long long)err_idx;
            break;
        }
        const uint32_t lim = (tw < remaining) ? tw : remaining;
        const bool more = produced + lim < orig;
        if (more) prefetch(next_ventry >> 7);  // lands in registers while phase B runs
        PROF_T(4);

        // ---- phase B: decode into the staging tile, flush aligned 16-byte units ----
        // per stream: np = descending bit position, oi = window symbol index of its next symbol, oe = one past its
        // last symbol inside the block
        uint32_t np[NS], oi[NS], oe[NS], os[NS];  // os = window symbol index of the stream's first symbol
        {
            uint32_t oo = o;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t end = oo + nsym[s];
                oi[s] = oo;
                os[s] = oo;
                oe[s] = end < lim ? end : lim;
                if (oe[s] < oi[s]) oe[s] = oi[s];
                np[s] = nbase - ((nsym[s] > 0) ? (uint32_t)s * SUB_BITS + g[s] : 0u);
                oo = end;
            }
        }
        bool direct = false;
        if constexpr (MULTI) {
            // Blocks that are almost entirely the 1-bit symbol (< 1.3 bits per symbol on average): the window's output
            // range is filled with that symbol by wide stores, then every thread walks its subsequence and stores only
            // the OTHER symbols, one byte each, straight to global memory.  No staging tile, so all subsequences of the
            // window are expanded at once (through the tile only ~OC/237 threads would work at a time).
            direct = sparse;
            if (sparse) {
                uint8_t* const dst = oblk + produced;  // lim bytes
                {
                    uint32_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
                    if (head > lim) head = lim;
                    if ((uint32_t)tid < head) dst[tid] = (uint8_t)zsym;
                    const uint32_t body = (lim - head) >> 4;
                    const uint32_t z4 = zsym * 0x01010101u;
                    uint4* const d4 = reinterpret_cast<uint4*>(dst + head);
                    for (uint32_t u = (uint32_t)tid; u < body; u += W) d4[u] = make_uint4(z4, z4, z4, z4);
                    const uint32_t t0 = head + (body << 4);
                    if ((uint32_t)tid < lim - t0) dst[t0 + tid] = (uint8_t)zsym;
                }
                // Workgroup-scope release + barrier: the fill of every wave is ordered before the single-byte stores any
                // other wave of this workgroup issues afterwards to the same lines (one CU, one L1, same-address order).
                // A device-scope fence here costs 8x the whole kernel (measured: 4.2 -> 32 ms).
                __threadfence_block();
                __syncthreads();
                bool any = oi[0] < oe[0];
                while (any) {
                    const unsigned long long two = fetch64(np[0]);
                    const uint32_t off = table_off<TB>(two, np[0]);
                    const bool a = oi[0] < oe[0];
                    const bool big = a && (oe[0] - oi[0] >= 3u);
                    uint32_t e;
                    if (big) e = *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)(mout_addr + 2u * off);
                    else e = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr + off);
                    uint32_t bits = big ? ((e >> 24) & 15u) : (e >> 8);
                    uint32_t cnt = big ? (e >> 28) : 1u;
                    if (__builtin_amdgcn_ballot_w64(a && e == 0) != 0ull) {
                        if (a && e == 0) {
                            e = slow_lookup(L, window32(two, np[0]));
                            bits = e >> 8;
                            cnt = 1;
                        }
                    }
                    if (cnt > 3u) {  // a run of the fill symbol: nothing to store
                        const uint32_t room = oe[0] - oi[0];
                        if (cnt > room) {
                            cnt = room;
                            bits = room * zlen;
                        }
                    } else if (a) {
                        const uint32_t s0 = e & 0xFFu, s1 = (e >> 8) & 0xFFu, s2 = (e >> 16) & 0xFFu;
                        if (s0 != zsym) dst[oi[0]] = (uint8_t)s0;
                        if (cnt > 1u && s1 != zsym) dst[oi[0] + 1u] = (uint8_t)s1;
                        if (cnt > 2u && s2 != zsym) dst[oi[0] + 2u] = (uint8_t)s2;
                    }
                    np[0] -= a ? bits : 0u;
                    oi[0] += a ? cnt : 0u;
                    any = oi[0] < oe[0];
                }
                gpos += lim;  // nothing is carried: gpos + ocarry == produced stays true (ocarry is 0 in this mode)
            }
        }
        for (uint32_t cbase = 0; !direct && cbase < lim;) {
            uint32_t cc = lim - cbase;
            const uint32_t room = (uint32_t)LdsT::CAP - ocarry;
            if (cc > room) {  // workgroup-uniform: the rest of the window does not fit one flush
                cc = room;
                if constexpr (LdsT::PRIV > 0) {
                    if (park) {  // end the flush on a subsequence boundary: no parked run straddles it
                        if (tid == 0) L.cend_vote = 0;
                        __syncthreads();
#pragma unroll
                        for (int s = 0; s < NS; s++) {
                            const uint32_t end0 = os[s] + nsym[s];
                            if (end0 > cbase + room / 2u && end0 <= cbase + room) atomicMax(&L.cend_vote, end0);
                        }
                        __syncthreads();
                        const uint32_t v = L.cend_vote;
                        if (v != 0u) cc = v - cbase;
                    }
                }
            }
            const uint32_t cend = cbase + cc;
            const uint32_t tshift = ocarry - cbase;  // tile index = window symbol index + tshift
            uint32_t ce[NS];                         // this chunk's end for each stream
            bool any = false;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                ce[s] = oe[s] < cend ? oe[s] : cend;
                if constexpr (LdsT::PRIV > 0) {
                    static_assert(LdsT::PRIV <= 112, "a parked run crosses at most two tile pads");
                    // a wholly parked subsequence that lies inside this flush: OR its registers into the tile,
                    // shifted to the byte phase of its first symbol (bytes outside the run are zero)
                    if (park && nsym[s] <= (uint32_t)LdsT::PRIV && oi[s] == os[s] && oi[s] < oe[s] && oe[s] <= cend) {
                        const uint32_t d = oi[s] + tshift;  // logical tile byte of the first symbol
                        const uint32_t sh = d & 3u, dw = d >> 2;
                        uint32_t* const tp = &L.outbuf[dw + (dw >> 4)];
                        const uint32_t kc = 16u - (dw & 15u);  // first register index past the next pad dword
                        const uint32_t sel = 0x07060504u - sh * 0x01010101u;
                        uint32_t prev = 0;
#pragma unroll
                        for (int k = 0; k <= NR; k++) {
                            const uint32_t cur = (k < NR) ? R[s][k < NR ? k : 0] : 0u;
                            const uint32_t v = __builtin_amdgcn_perm(cur, prev, sel);
                            prev = cur;
                            uint32_t pads = ((uint32_t)k >= kc) ? 1u : 0u;
                            if (k > 16) pads += ((uint32_t)k >= kc + 16u) ? 1u : 0u;
                            atomicOr(tp + k + pads, v);  // OR-ing zero is a no-op: no guard
                        }
                        oi[s] = oe[s];
                    }
                }
                any |= oi[s] < ce[s];
            }
            if constexpr (MULTI) {
                // short codes: up to 3 symbols per lookup while at least 3 symbols of this chunk are still owed
                while (any) {
                    const unsigned long long two = fetch64(np[0]);
                    const uint32_t off = table_off<TB>(two, np[0]);  // byte offset of the u16 entry = 2 * index
                    const bool a = oi[0] < ce[0];
                    const bool big = a && (ce[0] - oi[0] >= 3u);
                    uint32_t e;
                    if (big) e = *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)(mout_addr + 2u * off);
                    else e = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr + off);
                    uint32_t bits = big ? ((e >> 24) & 15u) : (e >> 8);
                    uint32_t cnt = big ? (e >> 28) : 1u;
                    if (__builtin_amdgcn_ballot_w64(a && e == 0) != 0ull) {
                        if (a && e == 0) {
                            e = slow_lookup(L, window32(two, np[0]));
                            bits = e >> 8;
                            cnt = 1;
                        }
                    }
                    const uint32_t t = oi[0] + tshift;
                    if (cnt > 3u) {  // a run of the shortest code: clip it to what this flush still owes
                        const uint32_t room = ce[0] - oi[0];
                        if (cnt > room) {
                            cnt = room;
                            bits = room * zlen;
                        }
                        if (zsym != 0u) {  // block-uniform; a run of 0x00 is already in the zeroed tile
                            for (uint32_t j = 3; j < cnt; j++) ob[opad(t + j)] = (uint8_t)zsym;
                        }
                    }
                    if (a) ob[opad(t)] = (uint8_t)e;
                    if (a && cnt > 1u) ob[opad(t + 1u)] = (uint8_t)(e >> 8);
                    if (a && cnt > 2u) ob[opad(t + 2u)] = (uint8_t)(e >> 16);
                    np[0] -= a ? bits : 0u;
                    oi[0] += a ? cnt : 0u;
                    any = oi[0] < ce[0];
                }
            }
            while (!MULTI && any) {
                unsigned long long two[NS];
                uint32_t e[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) two[s] = fetch64(np[s]);
#pragma unroll
                for (int s = 0; s < NS; s++)
                    e[s] = *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(tbl_addr +
                                                                                         table_off<TB>(two[s], np[s]));
                bool miss = false;
#pragma unroll
                for (int s = 0; s < NS; s++) miss |= (e[s] == 0 && oi[s] < ce[s]);
                if (__builtin_amdgcn_ballot_w64(miss) != 0ull) {
#pragma unroll
                    for (int s = 0; s < NS; s++)
                        if (e[s] == 0 && oi[s] < ce[s]) e[s] = slow_lookup(L, window32(two[s], np[s]));
                }
                any = false;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const bool a = oi[s] < ce[s];
                    if (a) ob[opad(oi[s] + tshift)] = (uint8_t)e[s];
                    np[s] -= a ? (e[s] >> 8) : 0u;
                    oi[s] += a ? 1u : 0u;
                    any |= oi[s] < ce[s];
                }
            }
            __syncthreads();
            PROF_T(5);
            const uint32_t total = ocarry + cc;
            const bool last = !more && cend == lim;  // final flush of the block: store the ragged tail too
            const uint32_t full = last ? total : (total & ~15u);
            uint8_t* const dst = obase + gpos;
            const uint32_t nunits = (full + 15u) >> 4;
            for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                const uint32_t lo = u << 4;
                uint32_t* src = &L.outbuf[(lo >> 2) + (lo >> 6)];  // a unit never straddles a pad
                if (out_aligned && lo + 16u <= full && lo >= hskip) {
                    *reinterpret_cast<uint4*>(dst + lo) = make_uint4(src[0], src[1], src[2], src[3]);
                } else {
                    for (uint32_t i = lo > hskip ? lo : hskip; i < lo + 16u && i < full; i++) dst[i] = ob[opad(i)];
                }
                if constexpr (LdsT::ZT) src[0] = src[1] = src[2] = src[3] = 0;
            }
            const uint32_t tail = total - full;  // < 16
            uint8_t tv = 0;
            if ((uint32_t)tid < tail) {
                tv = ob[opad(full + tid)];
                if constexpr (LdsT::ZT) {
                    if (full > 0u) ob[opad(full + tid)] = 0;
                }
            }
            __syncthreads();
            PROF_T(6);
            if ((uint32_t)tid < tail) ob[opad((uint32_t)tid)] = tv;
            gpos += full;
            ocarry = tail;
            if (full > 0u) hskip = 0;  // (the unit shared with the region before has been written)
            cbase = cend;
        }
        produced += lim;
        ventry = next_ventry;
        if (exhausted && produced < orig) {
            // The payload is used up but the chunk wants more symbols: the reference keeps reading zero bits
            // (TableBasedHuffmanDecoder.java:204-208), i.e. the all-zero codeword = first canonical symbol, forever.
            if ((uint32_t)tid < ocarry && (uint32_t)tid >= hskip) obase[gpos + tid] = ob[opad((uint32_t)tid)];  // unflushed tail
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
        PROF_T(7);
    }

    if (tid == 0 && (!split || status != DCZ_OK)) {  // (a split block got its status from k4_split_scan)
        d_status[b] = status;
        if (d_errpos) d_errpos[b] = errpos + (split ? (long long)sdp->off[(uint64_t)b * sdp->rmax + reg] : 0ll);
    }
    if (tid == 0) {
#if DCZ_K4_PROF
        for (int i = 0; i < 12; i++) atomicAdd(&k4_prof[i], pacc[i]);
#endif
    }
}

void launch_decode(const uint8_t* d_comp, size_t comp_bytes, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                   const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                   int32_t* d_status, int64_t* d_errpos, const DecodeWs& ws, hipStream_t s) {
    if (K == 0) return;
    const unsigned long long* off = reinterpret_cast<const unsigned long long*>(d_comp_off);
    long long* ep = reinterpret_cast<long long*>(d_errpos);
    uint8_t* const d_slow = ws.cls;
    // class byte of every block (bounds of the footer fields, fixed-length codes), then the analytic decoder for the latter
    launch_classify(d_len, d_comp_off, d_comp_size, d_orig_size, comp_bytes, out_stride, K, ws, d_status, d_errpos, s);
    launch_decode_fixed(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, K, out_stride, d_out, ws, s);
    // Below this many blocks one 1024-thread workgroup per block keeps more waves resident than 256-thread ones.
    static const uint32_t few_below = [] {
        const char* e = getenv("DCZ_K4_FEW_BLOCKS_BELOW");  // tuning knob
        return e ? (uint32_t)atoi(e) : 1024u;
    }();
#define DCZ_K4_LAUNCH_G(GRID, SDP, WW, NSS, OCC, PVV, MM, CC, TT, XX)                                                     \
    hipLaunchKernelGGL((k4_decode<WW, NSS, OCC, PVV, MM, CC, TT, XX>), dim3(GRID), dim3(WW), 0, s, d_comp, comp_bytes, off, \
                       d_comp_size, d_orig_size, d_len, out_stride, d_out, d_status, ep, d_slow, SDP)
#define DCZ_K4_LAUNCH(WW, NSS, OCC, PVV, MM, CC, TT, XX) \
    DCZ_K4_LAUNCH_G(K, (const SplitDesc*)nullptr, WW, NSS, OCC, PVV, MM, CC, TT, XX)
    // Few large blocks (the reference's 16-32 MiB chunks, a single dcz_decode_block call): cut every block into regions,
    // prove the region entries (k4_split.hip) and run the table-walk kernels once per region.  Blocks that cannot be
    // proven keep class 0 and are decoded by the per-block launches below.
    static const uint32_t split_below = [] {
        const char* e = getenv("DCZ_K4_SPLIT_BELOW");  // tuning knob; 0 = never split
        return e ? (uint32_t)atoi(e) : (uint32_t)SPLIT_MAX_BLOCKS;
    }();
    if (K < split_below && K <= SPLIT_MAX_BLOCKS && comp_bytes / K >= 4u * 65536u) {
        SplitDesc sd;
        // regions per block the grid provides for: the smallest region size (64 KiB) applied to the whole payload
        // buffer, capped by the table space; the region size itself is chosen on the device (k4_split_setup)
        unsigned long long rm = comp_bytes / 65536ull + 2;
        if (rm > SPLIT_ENTRIES / K) rm = SPLIT_ENTRIES / K;
        sd.rmax = (uint32_t)rm;
        sd.region_bytes = 65536;
        {
            sd.entry = ws.split;
            sd.count = ws.split + SPLIT_ENTRIES;
            sd.exit = ws.split + 2 * SPLIT_ENTRIES;
            sd.off = ws.split + 3 * SPLIT_ENTRIES;
            sd.nreg = ws.split + 4 * SPLIT_ENTRIES;
            sd.rbase = ws.split + 4 * SPLIT_ENTRIES + SPLIT_MAX_BLOCKS;
            sd.nblk = K;
            launch_split_count(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, K, d_slow, d_status, d_errpos, sd,
                               ws.sdesc, s);
            const uint32_t grid = SPLIT_GRID;
            DCZ_K4_LAUNCH_G(grid, ws.sdesc, DCZ_K4_W, DCZ_K4_NS, DCZ_K4_OC, DCZ_K4_PRIV, false, 4, DCZ_K4_TB, 3);
            DCZ_K4_LAUNCH_G(grid, ws.sdesc, DCZ_K4_W, 1, DCZ_K4M_OC, DCZ_K4_PRIVM, false, 2, DCZ_K4_TBM, 3);
#if DCZ_K4_MEDIUM_DFA
            launch_decode_dfa(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, K, out_stride, d_out, d_status, d_errpos, ws,
                              grid, s);
#endif
            DCZ_K4_LAUNCH_G(grid, ws.sdesc, DCZ_K4_W, 1, DCZ_K4L_OC, 0, true, 1, DCZ_K4_TBS, 3);
        }
    }
    if (K >= few_below) {
        static_assert(DCZ_K4_W <= 512, "many-blocks kernel");
#if DCZ_K4_EXACT
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4_OC, DCZ_K4_PRIV, false, 4, DCZ_K4_TB, 2);
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4M_OC, DCZ_K4_PRIVM, false, 2, DCZ_K4_TBM, 2);
#endif
        DCZ_K4_LAUNCH(DCZ_K4_W, DCZ_K4_NS, DCZ_K4_OC, DCZ_K4_PRIV, false, 4, DCZ_K4_TB, 0);
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4M_OC, DCZ_K4_PRIVM, false, 2, DCZ_K4_TBM, 0);
#if DCZ_K4_MEDIUM_DFA
        launch_decode_dfa(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, K, out_stride, d_out, d_status, d_errpos, ws, 0, s);
#endif
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4L_OC, 0, true, 1, DCZ_K4_TBS, 0);
#if DCZ_K4_EXACT
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4_OC, DCZ_K4_PRIV, false, 4, DCZ_K4_TB, 1);
        DCZ_K4_LAUNCH(DCZ_K4_W, 1, DCZ_K4M_OC, DCZ_K4_PRIVM, false, 2, DCZ_K4_TBM, 1);
#endif
    } else {
#if DCZ_K4_EXACT
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, DCZ_K4S_PRIV, false, 4, DCZ_K4_TB, 2);
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, DCZ_K4S_PRIVM, false, 2, DCZ_K4_TBM, 2);
#endif
        DCZ_K4_LAUNCH(1024, DCZ_K4S_NS, DCZ_K4S_OC, DCZ_K4S_PRIV, false, 4, DCZ_K4_TB, 0);
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, DCZ_K4S_PRIVM, false, 2, DCZ_K4_TBM, 0);
#if DCZ_K4_MEDIUM_DFA
        launch_decode_dfa(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, K, out_stride, d_out, d_status, d_errpos, ws, 0, s);
#endif
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, 0, true, 1, DCZ_K4_TBS, 0);
#if DCZ_K4_EXACT
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, DCZ_K4S_PRIV, false, 4, DCZ_K4_TB, 1);
        DCZ_K4_LAUNCH(1024, 1, DCZ_K4S_OC, DCZ_K4S_PRIVM, false, 2, DCZ_K4_TBM, 1);
#endif
    }
#undef DCZ_K4_LAUNCH
#undef DCZ_K4_LAUNCH_G
}

}  // namespace dcz

#if DCZ_K4_PROF
extern "C" void dcz_debug_k4_prof(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(dcz::k4_prof), sizeof(dcz::k4_prof));
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(dcz::k4_prof), z, sizeof(z));
    }
}
#endif
