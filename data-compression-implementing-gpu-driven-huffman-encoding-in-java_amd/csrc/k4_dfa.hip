// k4_dfa.hip -- K4 for medium code lengths (3.6 .. 6.5 bits per symbol, e.g. text): nibble automaton (gfx950).
//
// Same contract as k4_decode.hip (TableBasedHuffmanDecoder.decode, core/TableBasedHuffmanDecoder.java:103-152 with the
// fallback of core/CanonicalHuffman.java:161-229; zero bits past the payload :204-208; "decode error at position i"
// :109-111) and the same outer structure (windows of W subsequences x 32 bytes, self-synchronisation, workgroup scan,
// LDS tile flushed with aligned 16-byte stores).  What the counters of k4_decode's medium class say
// (profiles/r02_base_text8g_sq.txt): the kernel is INSTRUCTION-ISSUE bound -- per 64 symbols 33 vector + 14 scalar + 6 LDS
// instructions, the vector pipe of a SIMD ~60 % busy with four waves -- and a wave runs as long as its slowest lane
// (70 steps for 52 symbols on average).  This kernel does the same work with fewer, branch-free instructions:
//   * the decoder is a finite automaton over NIBBLES: state = internal node of the code tree (<= 255 states, root = 0),
//     T[state][nibble] = next state << 6 | symbols completed inside the nibble (0..2) | their bytes << 16.  Every lane
//     takes exactly 64 steps per subsequence, whatever the code lengths: no divergence, no ballots, no branches, no
//     escape path for long codewords (a 21-bit codeword is six ordinary steps), straight-line code;
//   * the subsequence lives in registers (8 dwords; the nibble of step j is a compile-time bit field) and needs no
//     look-ahead: a codeword that crosses a subsequence boundary is carried by the STATE, so a subsequence's entry and exit
//     are automaton states instead of bit offsets and the fixed point "my entry = my left neighbour's exit" is iterated
//     on states (first one exact; converges like codeword synchronisation does: 2 walks per window on text);
//   * round 0 is an exit-only walk over the subsequence's last 48 nibbles (1 vector + 1 LDS instruction per step), round 1
//     walks everything from the neighbour's exit and counts the symbols that COMPLETE inside the subsequence (3 + 1),
//     the output pass (phase B) is the same walk collecting the symbols in a register and storing whole dwords into the
//     tile at the offsets of the scan (9 + 1.25).
// Instantiations: MODE 0 one workgroup per block (256 threads, or 1024 for few blocks), MODE 1 / MODE 2 the decoding and
// the counting pass of the split decoder (one workgroup per region, k4_split.hip), SPARSE the variant for blocks dominated
// by a 1-bit symbol (up to four symbols per nibble, no tile).
// Not handled here (the block is left to k4_decode.hip's kernels, decided from the length table alone): tables with a
// 1-bit codeword that are not sparse (a nibble could complete three or four symbols), tables whose code tree has more
// than 255 internal nodes (foreign incomplete tables, or 256 symbols in a deep tree).  A window that is not synchronised
// after DCZ_K4_EXACT_AFTER rounds hands its block to the exact-entry launch, as everywhere.
#include <cstdlib>
#include <utility>

#include "dcz_internal.h"

namespace dcz {

#ifndef DCZ_DFA_W
#define DCZ_DFA_W 256
#endif
#ifndef DCZ_DFA_OC
#define DCZ_DFA_OC 16384  // tile bytes per flush: a whole window of text (13.3 KiB on average) in one flush
#endif
#ifndef DCZ_DFA_HOIST
#define DCZ_DFA_HOIST 1  // 1: the 64 nibble offsets of a subsequence are computed once per window and kept in registers
#endif                   // (1 vector instruction per walk step, ~120 VGPRs); 0: recomputed in every walk (3 per step, ~50 VGPRs)
// The nibbles of subsequence dwords >= HOIST_DW are extracted where they are used, not once per window: the instantiations
// that would spill otherwise (regions, 1024 threads) give up the last dword(s) -- a spill reload is a load, and on gfx950 a
// load waits for every older store of the wave.
// The nibble offsets of a subsequence dword are loop-invariant, so the compiler computes them once per window and keeps
// them in registers (DCZ_DFA_HOIST).  Where that must not happen -- dwords >= HOIST_DW, and every dword of the walks a
// recording instantiation only falls back to -- a step takes its dword from DFA_RJ: a COPY the compiler cannot see through,
// made when the walk reaches the dword.  The subsequence registers R themselves are never touched: an earlier form marked
// R[k] itself as modified ("+v"(R[k])), which made R loop-carried through every walk of the round loop, and two builds of
// this kernel then re-recorded the last symbols of a subsequence from a different dword 7 in repair rounds (same source,
// different register shuffles; found by a fuzz case, tools/dbg_stress.py).
#define DFA_RJ_DECL uint32_t rfresh_ = 0
#define DFA_RJ(rj, ALWAYS)                                                     \
    if constexpr ((j & 7) == 0 && ((ALWAYS) || (j >> 3) >= HOIST_DW)) {        \
        rfresh_ = R[j >> 3];                                                   \
        asm volatile("" : "+v"(rfresh_));                                      \
    }                                                                          \
    const uint32_t rj = ((ALWAYS) || (j >> 3) >= HOIST_DW) ? rfresh_ : R[j >> 3]
#ifndef DCZ_DFA_HOIST_DW_SPLIT
#define DCZ_DFA_HOIST_DW_SPLIT 3
#endif
#ifndef DCZ_DFA_HOIST_DW_REC
#define DCZ_DFA_HOIST_DW_REC 0  // (8 spills three registers at four waves per SIMD; 7: 127 registers, text 8 GiB 6.81 ms; 5: 6.62;
                                 //  4, 2, 0: 6.48 with 104-116 registers -- the kernel waits for LDS, it is not short of issue slots)
#endif
#ifndef DCZ_DFA_HOIST_DW_COUNT
#define DCZ_DFA_HOIST_DW_COUNT 8  // the counting pass of the split decoder (no recording: its registers are free; 0 or 4: 32 MiB
                                  // chunks K4 1.60 -> 1.58 ms, within the noise)
#endif
#ifndef DCZ_DFA_HOIST_DW_1024
#define DCZ_DFA_HOIST_DW_1024 5
#endif
#ifndef DCZ_DFA_X_FROM
#define DCZ_DFA_X_FROM 16  // first nibble of the exit-only walk (0: the whole subsequence)
#endif
#ifndef DCZ_DFA_KATTR
#define DCZ_DFA_KATTR  // debugging: an attribute for every k4_dfa instantiation, e.g. __attribute__((amdgpu_num_vgpr(120)))
#endif
#ifndef DCZ_DFA_ALLWALK
#define DCZ_DFA_ALLWALK 0  // debugging: every wave walks in every round, whether one of its lanes needs it or not
#endif
#ifndef DCZ_DFA_X6_BIT0
#define DCZ_DFA_X6_BIT0 64  // first bit of the six-bit exit-only walk (256 - this must be a multiple of 6): text 8 GiB K4 from bit 40
                            // 6.91 ms, 64: 6.89, 88: 7.01, 112: 7.39
#endif
#ifndef DCZ_DFA_X_FROM_SPARSE
#define DCZ_DFA_X_FROM_SPARSE 48  // first nibble of the exit-only walk of the SPARSE instantiations
#endif
#ifndef DCZ_DFA_SPARSE_OC
#define DCZ_DFA_SPARSE_OC 16384  // SPARSE: bytes of output composed per chunk (a window is ~60 KiB of output): what four workgroups
#endif                           // per CU leave after the table (16 KiB) and the lists (6.8 KiB)
#ifndef DCZ_DFA_ABL
#define DCZ_DFA_ABL 0  // timing ablations (WRONG output; tools/run_variants.sh --no-verify): 2 = no phase B stores,
#endif                 // 4 = no exit-only round, 8 = no phase B walk at all, 16 = one round only
#ifndef DCZ_DFA_DBG
#define DCZ_DFA_DBG 0
#endif
#ifndef DCZ_DFA_X6
#define DCZ_DFA_X6 1  // exit-only walk in 6-bit steps for code trees of <= 127 internal nodes (0: nibbles always)
#endif
#ifndef DCZ_DFA_RECORD
#define DCZ_DFA_RECORD 1  // 1: two walks per subsequence (exit-only, then ONE walk that counts AND records the symbols in a
#endif                    // private LDS slot; the slots are closed up in place); 0: three walks (exit-only, count, output)
#ifndef DCZ_DFA_MINWAVES
#define DCZ_DFA_MINWAVES 4  // waves per SIMD the 256-thread instantiations are compiled for
#endif
#define DFA_OC_DECODE (DCZ_DFA_RECORD ? -1 : DCZ_DFA_OC)  // tile parameter of the decoding instantiations (-1: slot area)
#ifndef DCZ_K4_EXACT_AFTER
#define DCZ_K4_EXACT_AFTER 12
#endif
#ifndef DCZ_K4_CLS2_A
#define DCZ_K4_CLS2_A 4
#endif
#ifndef DCZ_K4_CLS2_B
#define DCZ_K4_CLS2_B 9
#endif

constexpr uint32_t DFA_ERR = 255;  // sticky state: the stream left the code tree

// Recording walk (DCZ_DFA_RECORD).  Three walks per subsequence (exit-only, count, output) are two when the walk that counts
// also RECORDS the symbols: every lane has a private slot in LDS, and after the scan the slots are closed up in place (the
// slot area is the tile).  A subsequence of 32 bytes completes at most 128 symbols (a nibble completes two).  The 1024-thread
// instantiation (one workgroup per CU) affords slots of 128 + 8 bytes; the 256-thread one must fit four workgroups per CU
// (three measured 32 % SLOWER than the three-walk kernel, four 16 % faster: text 8 GiB 8.18 -> 6.88 ms), which leaves slots
// of 80 + 8 bytes: blocks that average more than DCZ_DFA_REC_AVG symbols per subsequence keep the three walks, and a window
// in which some subsequence still completes more than 80 (detected from the counts; the stores of such a lane run into its
// neighbours' slots, never out of the area) is redone by the output walk.  Slot strides of 34 and 22 dwords keep the 8-byte
// reads of 32 consecutive lanes on 32 different bank pairs.  The first 16 bytes of the area hold the bytes carried over
// from the window before.
#ifndef DCZ_DFA_SLOT_SMALL
#define DCZ_DFA_SLOT_SMALL 88
#endif
#ifndef DCZ_DFA_REC_AVG
#define DCZ_DFA_REC_AVG 60
#endif
constexpr int dfa_slot(int W) { return W > 512 ? 136 : DCZ_DFA_SLOT_SMALL; }
// SPARSE: the walk that counts records the symbols other than z as (position inside the subsequence, byte) pairs of 16 bits in
// a per-lane list of DFA_PCAP (12) entries (+ 1: the entry under construction); the window's output is then composed in LDS,
// chunk by chunk, and written ONCE (PL > 0: the tile is a chunk of OC bytes).
#ifndef DCZ_DFA_PCAP
#define DCZ_DFA_PCAP 12  // (16 with a 13 KiB chunk tile: 2.85 ms on config 5's slice; 12 with 16 KiB: 2.70; 10 with 18 KiB does not
                         //  fit four workgroups per CU any more: 3.44)
#endif
constexpr int DFA_PCAP = DCZ_DFA_PCAP;
template <int W, int OC, int PL = 0>
struct DfaLds {
    // tile capacity of the output walk (its last lane may run 300 bytes past the end of a flush); OC < 0: the slot area
    // of the recording walk is the tile
    static constexpr int CAP = (OC < 0) ? (16 + dfa_slot(W) * W - 512) : (OC + 512);
    static constexpr int TILE_BYTES = PL ? (OC + 64) : (CAP + 640);
    __attribute__((aligned(16))) uint32_t T[256 * 16];          // [state][nibble]
    union {
        __attribute__((aligned(16))) uint32_t tile[TILE_BYTES / 4];
        struct {  // scratch of the table build: dead once T is complete, and the tile is not touched before that
            uint32_t node_p[256];  // prefix of internal node `id` (the l bits that lead to it)
            uint32_t first[34], cnt[34], offs[34], nint[34], base[34];
            uint8_t node_l[256];   // its depth
            uint8_t symtab[256], len8[256];
        };
    };
    // (a lane that meets more than PL symbols other than z writes on into its neighbours' lists -- the window is then
    // redone -- and at most 128 entries far: the slack keeps the last lanes inside)
    uint16_t plist[PL ? W * (PL + 1) + 128 : 2];
    uint8_t exits[W];
    uint32_t wsum[W / 64];
    uint32_t flag[3];
    uint32_t maxlen, nstates, err_idx, cend_vote, zsym;
    int bad_table;
};

template <int W, class LdsT>
__device__ __forceinline__ uint32_t dfa_block_scan(uint32_t v, LdsT& L, uint32_t& total) {
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

// 16 payload bytes at payload-relative byte `off` (any alignment) from the aligned chunks of the virtual buffer
template <int Q>
__device__ __forceinline__ uint4 dfa_shift(const uint4& a, const uint4& b, uint32_t r) {
    const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return make_uint4(__builtin_amdgcn_alignbyte(d[Q + 1], d[Q], r), __builtin_amdgcn_alignbyte(d[Q + 2], d[Q + 1], r),
                      __builtin_amdgcn_alignbyte(d[Q + 3], d[Q + 2], r), __builtin_amdgcn_alignbyte(d[Q + 4], d[Q + 3], r));
}

#if DCZ_DFA_DBG
__device__ unsigned long long dfa_dbg[8];  // debugging (-DDCZ_DFA_DBG=1, tools/dbg_split.py): the walk stamps count and
                                            // byte checksum into its slot, the compaction compares what it is about to copy
#endif
#if DCZ_K4_PROF
__device__ unsigned long long dfa_prof[12];  // [8] windows, [9] rounds, [10] flushes
#define DFA_T(i)                                  \
    do {                                          \
        const unsigned long long t_ = clock64();  \
        pacc[i] += t_ - plast;                    \
        plast = t_;                               \
    } while (0)
#else
#define DFA_T(i) do { } while (0)
#endif

// SPLIT: one workgroup per REGION of a block that k4_split.hip has cut up (sdp: the region table); otherwise per block.
//
// SPARSE: the automaton for blocks of the SHORT-code class that are almost entirely one symbol z with a 1-bit codeword
// (< 1.3 bits per symbol; e.g. zero pages with 1 % noise: 237 symbols per 32-byte subsequence).  A nibble then completes
// up to FOUR symbols, of which at most two are not z (a codeword other than z has >= 2 bits; two of them fit a nibble
// only as [rest of a codeword, 1 bit][2 bits][z] or [..][z][2 bits] or [2 bits][2 bits] or [.., 2][2]: the first symbol
// of the nibble is then one of the two):
//   T[state][nibble] = count (0..4) | others (0..2) << 3 | next << 6 | pos << 14 | sym0 << 16 | sym1 << 24
//   (one other symbol: sym0 at index pos of the nibble's symbols; two: sym0 at index 0 and sym1 at index pos).
// Walks are the same (entry/exit states, fixed point, scan of the counts); the output pass has no tile: the window's
// output range is filled with z by 16-byte stores and the walk stores the other symbols as single bytes straight to
// global memory (the scheme of k4_decode.hip's sparse path, whose table walk this replaces: 4.25 -> see DESIGN.md).
// MODE 0: one workgroup per block.  MODE 1 (SPLIT): one workgroup per REGION of a block that has been cut up, entry state,
// output offset and symbol count from the region table.  MODE 2 (COUNT): the counting pass that fills that table for the
// blocks this automaton takes (k4_split.hip's jump walk does it for the others): regions are S payload bytes, the
// workgroup of region r > 0 first walks the window in front of its region from the guess "codeword boundary" -- the
// state it arrives in at the region's first byte is the region's entry --, then walks its region window by window
// (rounds of phase A only) and reports (entry state, symbols completed inside the region, exit state).
// k4_split_scan proves the chain exit(r-1) == entry(r): equal states at the same byte are the same parse from there on.
template <int W, int OC, int MODE, bool SPARSE>
__global__ __launch_bounds__(W, W > 512 ? 1 : DCZ_DFA_MINWAVES) DCZ_DFA_KATTR void k4_dfa(
    const uint8_t* __restrict__ comp, const unsigned long long* __restrict__ d_comp_off,
    const uint32_t* __restrict__ d_comp_size, const uint32_t* __restrict__ d_orig_size, const uint8_t* __restrict__ d_len,
    size_t out_stride, uint8_t* __restrict__ out, int32_t* __restrict__ d_status, long long* __restrict__ d_errpos,
    uint8_t* __restrict__ d_cls, const SplitDesc* __restrict__ sdp) {
    using LdsT = DfaLds<W, OC, SPARSE ? DFA_PCAP : 0>;
    constexpr bool SPLIT = MODE == 1, COUNT = MODE == 2;
    constexpr bool RECORD = OC < 0;  // the walk that counts also records the symbols (DfaLds: the slot area is the tile)
    static_assert(!RECORD || (!SPARSE && !COUNT), "only the decoding passes of the dense automaton record");
    constexpr int HOIST_DW = !DCZ_DFA_HOIST ? 0 : SPLIT ? DCZ_DFA_HOIST_DW_SPLIT : W > 512 ? DCZ_DFA_HOIST_DW_1024 : COUNT ? DCZ_DFA_HOIST_DW_COUNT : DCZ_DFA_HOIST_DW_REC;
    __shared__ LdsT L;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    uint32_t b = blockIdx.x, reg = 0;
    const int tid = (int)threadIdx.x;
#if DCZ_K4_PROF
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long plast = clock64();
#endif
    if constexpr (SPLIT) {
        if (!split_region_of(sdp, blockIdx.x, b, reg)) return;
        if (d_cls[b] != 2 || reg >= sdp->nreg[b]) return;  // workgroup-uniform
    } else if constexpr (COUNT) {
        if (!split_region_of(sdp, blockIdx.x, b, reg)) return;
        if (d_cls[b] != 0) return;  // fixed-length or rejected block
    } else {
        if (d_cls[b] != 0) return;  // fixed-length, exact-entry, split or rejected block (workgroup-uniform)
    }
    const uint32_t orig_blk = d_orig_size[b];
    const unsigned long long coff = d_comp_off[b];
    const uint32_t csize = d_comp_size[b];
    if constexpr (COUNT) {  // the regions k4_split_scan expects of this block (k4_split_count's rule)
        const unsigned long long S = sdp->region_bytes;
        const unsigned long long nreg = ((((uintptr_t)comp + (uintptr_t)coff) & 15u) + csize + S - 1) / S;
        if (nreg < 2 || reg >= nreg || nreg > sdp->rmax) return;
    }
    // symbols this workgroup produces: the whole chunk, or what k4_split_scan gave its region
    const uint32_t orig = SPLIT ? sdp->count[(uint64_t)b * sdp->rmax + reg] : COUNT ? 0xFFFFFFFFu : orig_blk;
    if (SPLIT && orig == 0u) return;
    {
        const bool long_codes = (unsigned long long)csize * 16ull >= (unsigned long long)orig_blk * 13ull;
        const bool medium = (unsigned long long)orig_blk * (unsigned long long)DCZ_K4_CLS2_A <=
                            (unsigned long long)csize * (unsigned long long)DCZ_K4_CLS2_B;
        if constexpr (SPARSE) {
            static_assert(!SPLIT, "regions of sparse blocks are decoded by k4_decode.hip");
            const bool sparse = (unsigned long long)csize * 80ull < (unsigned long long)orig_blk * 13ull;
            if (long_codes || medium || !sparse) return;
        } else {
            if (long_codes || !medium) return;  // workgroup-uniform; other launches own those blocks
        }
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
        }
        L.maxlen = mx;
        if (kraft > (1ull << 32)) L.bad_table = 1;  // not a prefix code
        // internal nodes per depth: the prefixes of the longer codewords follow the leaves of a depth contiguously
        // (canonical code), so nint[l] = ceil((cnt[l+1] + nint[l+1]) / 2); state id = base[l] + index among them
        L.nint[32] = 0;
        L.nint[33] = 0;
        for (int l = 31; l >= 0; l--) L.nint[l] = (L.cnt[l + 1] + L.nint[l + 1] + 1u) / 2u;
        uint32_t acc = 0;
        for (int l = 0; l < 34; l++) {
            L.base[l] = acc;
            acc += L.nint[l];
        }
        L.nstates = acc;
    }
    __syncthreads();
    if (L.bad_table) {
        if (!COUNT && tid == 0) {  // (COUNT: k4_split_count marks the regions of such a block unusable)
            d_status[b] = DCZ_E_BADTABLE;
            if (d_errpos) d_errpos[b] = 0;
        }
        return;
    }
    // tables this automaton does not take (k4_decode.hip's medium-class kernel applies the same test and decodes them)
    // (a region adds up to two rows for its entry, see below)
    if constexpr (SPARSE) {  // (k4_decode.hip's short-code kernel applies the same test and leaves the block alone)
        if (L.cnt[1] != 1u || L.nstates > 255u || L.maxlen < 2u) return;
    } else {
        if (L.cnt[1] != 0u || L.nstates > 255u || L.maxlen == 0u) return;
    }
    // Code trees of at most 127 internal nodes (text: 96) leave the upper half of T unused: it holds a second table for the
    // exit-only walk, X6[state][6 bits] = next state (one byte), and that walk takes 32 steps of 6 bits instead of 48
    // nibbles.  The error state of such a block is 127 instead of 255 (block-uniform).
    constexpr bool X6_OK = DCZ_DFA_X6 && !SPARSE && !COUNT;  // (the counting pass keeps the nibbles: its registers are full;
    const bool small = X6_OK && L.nstates <= 127u;           //  state numbers are the same either way)
    const uint32_t errst = small ? 127u : DFA_ERR;
    const uint32_t trows = small ? 128u : 256u;
    for (int sy = tid; sy < 256; sy += W) {
        const uint32_t l = L.len8[sy];
        if (l > 0) {
            uint32_t rank = 0;
            for (int t = 0; t < sy; t++) rank += (L.len8[t] == l) ? 1u : 0u;
            L.symtab[L.offs[l] + rank] = (uint8_t)sy;
        }
    }
    for (uint32_t id = (uint32_t)tid; id < L.nstates; id += W) {  // (depth, prefix) of every internal node
        uint32_t l = 0;
        while (l < 33u && id >= L.base[l + 1]) l++;
        L.node_l[id] = (uint8_t)l;
        L.node_p[id] = L.first[l] + L.cnt[l] + (id - L.base[l]);
    }
    __syncthreads();
    if (tid == 0) L.zsym = L.symtab[0];  // the first canonical symbol (symtab shares its memory with the tile)
    for (uint32_t idx = (uint32_t)tid; idx < trows * 16u; idx += W) {
        const uint32_t st = idx >> 4, nib = idx & 15u;
        uint32_t e = errst << 6;  // unused rows and the error state: stay in the error state
        if (st < L.nstates) {
            uint32_t l = L.node_l[st], p = L.node_p[st], c = 0, syms = 0;
            uint32_t nz = 0, pos = 0;  // SPARSE: symbols other than z and the index of the last one
            const uint32_t z = L.symtab[0];
            bool err = false;
            for (int i = 3; i >= 0 && !err; i--) {
                p = 2u * p + ((nib >> i) & 1u);
                l++;
                if (l > 32u) {
                    err = true;
                    break;
                }
                const uint32_t rel = p - L.first[l];
                if (rel < L.cnt[l]) {  // a leaf: the codeword is complete
                    const uint32_t sy = L.symtab[L.offs[l] + rel];
                    if constexpr (SPARSE) {
                        if (sy != z) {
                            syms |= sy << (8 * nz);
                            pos = c;
                            nz++;
                        }
                    } else {
                        syms |= sy << (8 * c);
                    }
                    c++;
                    l = 0;
                    p = 0;
                } else if (rel - L.cnt[l] >= L.nint[l]) {
                    err = true;  // no codeword has this prefix (incomplete code)
                }
            }
            // (a nibble that leaves the code tree still counts the symbols it completed before: the reference's error
            // position is the number of symbols decoded so far)
            const uint32_t next = err ? errst : L.base[l] + (p - L.first[l] - L.cnt[l]);
            if constexpr (SPARSE) e = c | (nz << 3) | (next << 6) | (pos << 14) | (syms << 16);
            else e = (next << 6) | c | (c << 3) | (syms << 16);
        }
        L.T[idx] = e;
    }
    if (small) {  // X6: the state six bits on (symbols do not matter to the exit-only walk)
        uint8_t* const x6 = reinterpret_cast<uint8_t*>(&L.T[128 * 16]);
        for (uint32_t idx = (uint32_t)tid; idx < 128u * 64u; idx += W) {
            const uint32_t st = idx >> 6, v = idx & 63u;
            uint32_t next = errst;
            if (st < L.nstates) {
                uint32_t l = L.node_l[st], p = L.node_p[st];
                bool err = false;
                for (int i = 5; i >= 0 && !err; i--) {
                    p = 2u * p + ((v >> i) & 1u);
                    l++;
                    if (l > 32u) {
                        err = true;
                        break;
                    }
                    const uint32_t rel = p - L.first[l];
                    if (rel < L.cnt[l]) {  // a leaf: the codeword is complete
                        l = 0;
                        p = 0;
                    } else if (rel - L.cnt[l] >= L.nint[l]) {
                        err = true;  // no codeword has this prefix (incomplete code)
                    }
                }
                if (!err) next = L.base[l] + (p - L.first[l] - L.cnt[l]);
            }
            x6[idx] = (uint8_t)next;
        }
    }
    const uint32_t t_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.T[0]));
    const uint32_t tile_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(&L.tile[0]));

    // output of this workgroup: the chunk's slot, or the region's offset inside it.  A region starts at any byte: the
    // tile is laid over the 16-byte unit that holds its first byte, whose first hskip bytes belong to the region before.
    uint8_t* const oblk = out + (uint64_t)b * out_stride + (SPLIT ? sdp->off[(uint64_t)b * sdp->rmax + reg] : 0u);
    uint32_t hskip = SPLIT ? (uint32_t)(((uintptr_t)oblk) & 15u) : 0u;
    uint8_t* const obase = oblk - hskip;
    const bool out_aligned = (((uintptr_t)obase) & 15u) == 0u;
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)coff;
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew, vhi = (unsigned long long)skew + csize;

    uint32_t produced = 0, gpos = 0, ocarry = hskip;
    uint32_t entry0 = 0;  // state at the first nibble of the window (the block starts at the root)
    unsigned long long wbyte = 0;  // payload byte of the window's first subsequence
    uint32_t nwin = 0xFFFFFFFFu;  // COUNT: windows still to walk, the one in front of the region included
    bool preroll = false;
    if constexpr (SPLIT) {  // the region's proven entry: the automaton's state at its first payload byte
        wbyte = (unsigned long long)reg * sdp->region_bytes;
        entry0 = sdp->entry[(uint64_t)b * sdp->rmax + reg];
    }
    if constexpr (COUNT) {
        const unsigned long long S = sdp->region_bytes;
        nwin = (uint32_t)(S / ((unsigned long long)W * 32ull));  // (S is a multiple of 8 KiB)
        wbyte = (unsigned long long)reg * S;
        if (reg > 0) {
            wbyte -= (unsigned long long)W * 32ull;
            nwin++;
            preroll = true;
        }
    }
    int status = DCZ_OK;
    long long errpos = 0;
    constexpr uint32_t AMASK = SPARSE ? 0x3FC0u : 0xFFC0u;  // state field of an entry, as the byte offset of its row
    uint8_t* const ob = reinterpret_cast<uint8_t*>(L.tile);

    // this lane's 32 payload bytes of the window at `wb`: three aligned chunks, shifted by the payload's skew
    uint4 pre[3];
    auto prefetch = [&](unsigned long long wb) {
        const unsigned long long v0 = (wb + 32ull * (unsigned long long)tid + skew) & ~15ull;
#pragma unroll
        for (int c = 0; c < 3; c++) pre[c] = load_chunk16(vbase, v0 + 16ull * c, vlo, vhi);
    };
    // byte offset of a lane's first payload byte inside its first aligned chunk: the same for every lane and window
    // (lanes are 32 bytes apart, windows W * 32)
    const uint32_t sk2 = (skew + (uint32_t)(wbyte & 15ull)) & 15u;
    const uint32_t sq = sk2 >> 2, sr = sk2 & 3u;
    if (orig > 0 && wbyte < csize) prefetch(wbyte);
    const uint32_t slot_addr = tile_addr + 16u + (uint32_t)dfa_slot(W) * (uint32_t)tid;  // RECORD: this lane's slot
    __syncthreads();  // T complete; the scratch of the table build, which shares the tile's memory, is dead
    // RECORD: does this block record?  (block-uniform; slots of 128 bytes take anything)
    const bool rec_block = RECORD && (dfa_slot(W) - 8 >= 128 || (unsigned long long)orig_blk * 32ull <=
                                                               (unsigned long long)csize * (unsigned long long)DCZ_DFA_REC_AVG);
    auto zero_tile = [&](uint32_t from16) {  // the output walk ORs into a zeroed tile
        for (uint32_t i = from16 + (uint32_t)tid; i < (uint32_t)(sizeof(L.tile) / 16u); i += W)
            reinterpret_cast<uint4*>(L.tile)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    };
    if (!rec_block) zero_tile(0u);  // (SPARSE: the chunk tile holds byte ^ z over a zero background)
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const uint32_t pl_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.plist[0])) +
                             2u * (uint32_t)(DFA_PCAP + 1) * (uint32_t)tid;  // SPARSE: this lane's patch list
    DFA_T(0);

    uint32_t rentry = 0, wexit = 0;  // COUNT: the region's entry state; every mode: state after the window's last subsequence
    unsigned long long rcount = 0;   // COUNT: symbols completed inside the region
    bool unusable = false;           // COUNT: the region cannot be proven (left the code tree, no fixed point)
    while (COUNT ? nwin != 0u : produced < orig) {
#if DCZ_K4_PROF
        pacc[8]++;
#endif
        uint32_t R[8];
        {
            uint4 lo, hi;
            switch (sq) {  // block-uniform
                case 0: lo = dfa_shift<0>(pre[0], pre[1], sr); hi = dfa_shift<0>(pre[1], pre[2], sr); break;
                case 1: lo = dfa_shift<1>(pre[0], pre[1], sr); hi = dfa_shift<1>(pre[1], pre[2], sr); break;
                case 2: lo = dfa_shift<2>(pre[0], pre[1], sr); hi = dfa_shift<2>(pre[1], pre[2], sr); break;
                default: lo = dfa_shift<3>(pre[0], pre[1], sr); hi = dfa_shift<3>(pre[1], pre[2], sr); break;
            }
            R[0] = bswap32(lo.x);
            R[1] = bswap32(lo.y);
            R[2] = bswap32(lo.z);
            R[3] = bswap32(lo.w);
            R[4] = bswap32(hi.x);
            R[5] = bswap32(hi.y);
            R[6] = bswap32(hi.z);
            R[7] = bswap32(hi.w);
        }
        DFA_T(1);
        // Subsequences that START past the payload hold nothing but zero padding: they take no part; the symbols the
        // reference would read from the padding are filled in after the window (see "exhausted").
        const bool exhausted = wbyte + (unsigned long long)W * 32ull >= csize;
        const bool beyond = wbyte + 32ull * (unsigned long long)tid >= csize;
        const bool more_payload = !exhausted;
        if (more_payload) prefetch(wbyte + (unsigned long long)W * 32ull);  // the window grid is fixed: load ahead

        // ---- phase A: fixed point of "my entry state = my left neighbour's exit state" ----
        // Round 0 walks every subsequence from the guess "at a codeword boundary" and only wants its exit state; round 1
        // walks every subsequence again from its neighbour's exit and counts the symbols that complete inside it (a wave
        // walks as long as any of its lanes does, so lanes whose guess was right cost nothing extra and get their count
        // here too); later rounds (rare) re-walk the lanes whose entry still changed.
        uint32_t g = (tid == 0) ? entry0 : 0u;  // entry state
        uint32_t x = 0, nsym = 0, pc = 0;  // pc (SPARSE): symbols other than z this lane has recorded
        bool pover = false;  // some walk of this lane ran over its list (SPARSE) or slot (RECORD) -- in ANY round: a walk from a wrong
                             // entry state that runs over scribbles on the lists / slots of lanes in the NEXT wave, which may not
                             // walk again when this lane does (its exit can be the same after all)
        bool need = !beyond;
        uint32_t round = (DCZ_DFA_ABL & 4) ? 1u : 0u;
        while (true) {
            if (DCZ_DFA_ALLWALK || __builtin_amdgcn_ballot_w64(need) != 0ull) {  // wave-uniform: somebody in this wave walks
                uint32_t e = g << 6, n = 0;
                if (round == 0u && small) {  // workgroup-uniform: 32 steps of six bits over the last 192 bits
                    constexpr int X6_BIT0 = DCZ_DFA_X6_BIT0, X6_STEPS = (256 - X6_BIT0) / 6;
                    static_assert(X6_BIT0 + 6 * X6_STEPS == 256, "the six-bit walk ends on the subsequence's last bit");
                    uint32_t rq[8], st6 = 0;  // (copies the compiler cannot see through: nothing of this walk is kept in
#pragma unroll                                //  registers across the rounds, unlike the nibble offsets)
                    for (int k = X6_BIT0 >> 5; k < 8; k++) {
                        rq[k] = R[k];
                        asm volatile("" : "+v"(rq[k]));
                    }
                    auto stepX6 = [&](auto ic) __attribute__((always_inline)) {
                        constexpr int pbit = X6_BIT0 + 6 * decltype(ic)::value, k = pbit >> 5, off = pbit & 31;
                        uint32_t v;
                        if constexpr (off <= 26) v = (rq[k] >> (26 - off)) & 63u;
                        else v = __builtin_amdgcn_alignbit(rq[k], rq[k < 7 ? k + 1 : 7], 58 - off) & 63u;
                        st6 = (uint32_t)*(const lds_u8*)(uintptr_t)(t_addr + 8192u + ((st6 << 6) | v));
                    };
                    [&]<int... Is>(std::integer_sequence<int, Is...>) {
                        (stepX6(std::integral_constant<int, Is>{}), ...);
                    }(std::make_integer_sequence<int, X6_STEPS>{});
                    e = st6 << 6;
                } else if (round == 0u) {  // workgroup-uniform
                    // (SPARSE starts later: in a run of the 1-bit symbol every position is a codeword boundary, a wrong guess is
                    //  put right by the first such run -- from nibble 16: 2.75 ms on config 5's slice, 40: 2.68, 48: 2.63,
                    //  56: 2.62; 1 GiB at 3 % noise: 0.66 / 0.63 / 0.61 / 0.65 ms: tools/sparse_noise.py)
                    constexpr int XF = SPARSE ? DCZ_DFA_X_FROM_SPARSE : DCZ_DFA_X_FROM;
                    if (XF > 0) e = 0;  // (mid-subsequence every lane guesses "codeword boundary")
                    DFA_RJ_DECL;
                    auto stepX = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int sh = 26 - 4 * (j & 7);  // nibble j of the dword, as a byte offset of a u32 entry
                        DFA_RJ(rj, false);
                        const uint32_t nib4 = sh >= 0 ? ((rj >> (sh >= 0 ? sh : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                        e = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib4));
                    };
                    // The guess only has to be right often: the walk covers the subsequence's last 64 - DCZ_DFA_X_FROM
                    // nibbles (text synchronises within 5 nibbles on average, 0.03 % of the subsequences need more
                    // than 48); a wrong exit is found and repaired by the rounds that follow.  Text 8 GiB: 8.37 ms from nibble 0, 8.02 from 16 or 24, 8.34 from 32 (more repair rounds), 8.98 from 48.
                    [&]<int... Js>(std::integer_sequence<int, Js...>) {
                        (stepX(std::integral_constant<int, XF + Js>{}), ...);
                    }(std::make_integer_sequence<int, 64 - XF>{});
                } else if constexpr (SPARSE) {
                    // One walk counts and records: (index inside the subsequence, byte) of every symbol other than z goes
                    // into the lane's list.  The entry is stored every step and the list pointer moves on only when the
                    // nibble had such a symbol; two in one nibble (1 step in 10^4 on 1 % noise) take the branch.
                    uint32_t pa = pl_addr;
                    DFA_RJ_DECL;
                    auto stepP = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int sh = 26 - 4 * (j & 7);
                        DFA_RJ(rj, false);
                        const uint32_t nib4 = sh >= 0 ? ((rj >> (sh >= 0 ? sh : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                        e = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib4));
                        const uint32_t oth = e & 0x18u;  // 8 * symbols other than z
                        const uint32_t tp = n + ((e >> 14) & 3u);
                        *(lds_u16*)(uintptr_t)pa = (uint16_t)(tp | ((e >> 8) & 0xFF00u));
                        if (__builtin_amdgcn_ballot_w64(oth == 0x10u) != 0ull) {
                            if (oth == 0x10u) {  // sym0 is the nibble's first symbol, sym1 the one at index pos
                                *(lds_u16*)(uintptr_t)pa = (uint16_t)(n | ((e >> 8) & 0xFF00u));
                                *(lds_u16*)(uintptr_t)(pa + 2u) = (uint16_t)(tp | ((e >> 16) & 0xFF00u));
                            }
                        }
                        pa += oth >> 2;
                        n += e & 7u;
                    };
                    [&]<int... Js>(std::integer_sequence<int, Js...>) {
                        (stepP(std::integral_constant<int, Js>{}), ...);
                    }(std::make_integer_sequence<int, 64>{});
                    if (need) pc = (pa - pl_addr) >> 1;
                } else if (RECORD && rec_block) {
                    // One walk counts AND records: the symbols of two nibbles (up to four bytes) are shifted into a
                    // register at the lane's fill level and the dword under construction is stored into the lane's
                    // private slot every time (a plain store: nobody else writes there); when it is full the spill-over
                    // becomes the next dword.  The count is the fill level at the end.
                    uint32_t ab = slot_addr, k8 = 0, alo = 0;
#if DCZ_DFA_DBG
                    uint32_t chk = 0;  // sum of the recorded symbol bytes
#endif
                    DFA_RJ_DECL;
                    auto stepR = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = 2 * decltype(jc)::value;
                        constexpr int sh0 = 26 - 4 * (j & 7), sh1 = 26 - 4 * ((j + 1) & 7);
                        DFA_RJ(rj, false);
                        const uint32_t nib0 = (rj >> sh0) & 0x3Cu;
                        const uint32_t nib1 = sh1 >= 0 ? ((rj >> (sh1 >= 0 ? sh1 : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                        const uint32_t e0 = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib0));
                        e = *(lds_cu32*)(uintptr_t)(t_addr + ((e0 & AMASK) | nib1));
                        const uint32_t c0 = e0 & 0x18u;  // 8 * symbols of the first nibble (bytes past the count are zero)
                        const uint32_t w = ((e >> 16) << c0) | (e0 >> 16);
                        const unsigned long long v = (unsigned long long)w << k8;
#if DCZ_DFA_DBG
                        chk = __builtin_amdgcn_sad_u8(w, 0u, chk);
#endif
                        alo |= (uint32_t)v;
                        *(lds_u32*)(uintptr_t)ab = alo;
                        k8 += c0 + (e & 0x18u);
                        const bool full = k8 >= 32u;
                        alo = full ? (uint32_t)(v >> 32) : alo;
                        ab += full ? 4u : 0u;
                        k8 &= 31u;
                    };
                    [&]<int... Js>(std::integer_sequence<int, Js...>) {
                        (stepR(std::integral_constant<int, Js>{}), ...);
                    }(std::make_integer_sequence<int, 32>{});
                    *(lds_u32*)(uintptr_t)ab = alo;  // (the dword under construction; at most slot byte 128..131)
                    n = (ab - slot_addr) + (k8 >> 3);
#if DCZ_DFA_DBG
                    *(lds_u32*)(uintptr_t)(slot_addr + (uint32_t)dfa_slot(W) - 4u) = (n & 0xFFu) | (chk << 8);
#endif
                } else {
                    DFA_RJ_DECL;
                    auto stepA = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int sh = 26 - 4 * (j & 7);
                        DFA_RJ(rj, RECORD || SPARSE);
                        const uint32_t nib4 = sh >= 0 ? ((rj >> (sh >= 0 ? sh : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                        e = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib4));
                        n += e & (SPARSE ? 7u : 3u);
                    };
                    [&]<int... Js>(std::integer_sequence<int, Js...>) {
                        (stepA(std::integral_constant<int, Js>{}), ...);
                    }(std::make_integer_sequence<int, 64>{});
                }
                if (need) {
                    x = (e >> 6) & 0xFFu;
                    nsym = n;
                    if constexpr (SPARSE) pover |= pc > (uint32_t)DFA_PCAP;
                    else if constexpr (RECORD) pover |= n > (uint32_t)dfa_slot(W) - 8u;
                }
            }
            DFA_T(2);
#if DCZ_K4_PROF
            pacc[9]++;
#endif
            L.exits[tid] = (uint8_t)x;
            if (tid == 0) L.flag[(round + 1u) % 3u] = 0;
            __syncthreads();
            DFA_T(3);
            if (round > 1u && L.flag[round % 3u] == 0u) break;
            if (DCZ_DFA_ABL & 16) break;
            const uint32_t ng = (tid == 0) ? entry0 : (uint32_t)L.exits[tid - 1];
            need = ((ng != g) || round == 0u) && !beyond;  // after the exit-only round everybody walks once more
            g = ng;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull && (tid & 63) == 0) L.flag[(round + 1u) % 3u] = 1;
            round++;
            if constexpr (MODE == 0 && !SPARSE) {  // (a proven region and a sparse block simply keep iterating: at most W rounds)
                if (round == (uint32_t)DCZ_K4_EXACT_AFTER) {  // workgroup-uniform: this block does not self-synchronise
                    if (tid == 0) d_cls[b] = 1;                // the exact-entry launch (k4_decode.hip, MODE 1) decodes it
                    return;
                }
            }
            if constexpr (COUNT) {
                if (round == (uint32_t)DCZ_K4_EXACT_AFTER) {  // no fixed point: the block stays with the per-block kernels
                    if (tid == 0) sdp->exit[(uint64_t)b * sdp->rmax + reg] = 0xFFFFFFFFu;
                    return;
                }
            }
        }

        // ---- offsets, errors ----
        const bool bad = !beyond && x == errst;
        uint32_t tw = 0;
        // (RECORD with small slots: a subsequence that ran over its slot rides along in bit 20 of the scanned value -- the
        // counts of a window add up to < 2^16 -- so the window learns about it without another barrier)
        constexpr uint32_t SLOT_SYMS = RECORD ? (uint32_t)dfa_slot(W) - 8u : 0u;
        constexpr bool SLOT_CHECK = (RECORD && SLOT_SYMS < 128u) || SPARSE;
        const bool ran_over = !beyond && pover;
        uint32_t o = dfa_block_scan<W>(nsym + ((SLOT_CHECK && ran_over) ? (1u << 20) : 0u), L, tw);
        const bool slot_ran_over = SLOT_CHECK && (tw >> 20) != 0u;
        if constexpr (SLOT_CHECK) {
            o &= 0xFFFFFu;
            tw &= 0xFFFFFu;
        }
        const uint32_t remaining = orig - produced;
        if (bad) atomicMin(&L.err_idx, o + nsym);  // (the error state completes no symbol: nsym = symbols before it)
        // state after the last subsequence that starts inside the payload (the window's entry if none does)
        {
            uint32_t nreal = csize > wbyte ? (uint32_t)((csize - wbyte + 31ull) >> 5) : 0u;
            if (nreal > (uint32_t)W) nreal = (uint32_t)W;
            wexit = nreal ? (uint32_t)L.exits[nreal - 1u] : entry0;
        }
        const uint32_t next_entry = wexit;
        __syncthreads();
        if constexpr (COUNT) {
            // nothing is decoded: add up, move on.  A subsequence of the region that left the code tree makes the region
            // unusable (the per-block kernels then report the error position); in the window in front of the region only
            // the state it ends in matters.
            if (preroll) {
                rentry = wexit;
                preroll = false;
            } else {
                rcount += tw;
                if (L.err_idx != 0xFFFFFFFFu) unusable = true;
            }
            __syncthreads();
            if (tid == 0) L.err_idx = 0xFFFFFFFFu;
            entry0 = wexit;
            wbyte += (unsigned long long)W * 32ull;
            nwin--;
            if (exhausted) break;
            __syncthreads();
            continue;
        }
        const uint32_t err_idx = L.err_idx;
        if (err_idx < remaining) {
            status = DCZ_E_BADSTREAM;
            errpos = (long long)produced + (long long)err_idx;
            break;
        }
        const uint32_t lim = (tw < remaining) ? tw : remaining;
        const bool more = produced + lim < orig;
        DFA_T(4);

        if constexpr (SPARSE) {
            const uint32_t z = L.zsym, z4 = z * 0x01010101u;
            if (!slot_ran_over) {
                // ---- phase B, sparse: the window's output is composed in LDS, chunk by chunk, and written once ----
                // Every byte of the window is z unless a list says otherwise.  The tile holds byte ^ z over a zero
                // background: the lanes whose symbols fall into the chunk drop their entries in, the flush turns units of 16
                // bytes into output (tile ^ zzzz, aligned 16-byte stores) and zeroes them again.  As in the dense kernels the
                // first ocarry bytes of the tile are the ones carried over (the output position is not a multiple of 16).
                const uint32_t pcn = (!beyond && nsym > 0u) ? pc : 0u;
                for (uint32_t cbase = 0; cbase < lim;) {
                    const uint32_t room = (uint32_t)OC - ocarry;
                    const uint32_t cc = (lim - cbase < room) ? lim - cbase : room;
                    const uint32_t cend = cbase + cc;
                    {
                        const uint32_t wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
                        const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(o + nsym), 63);
                        if (whi > cbase && wlo < cend) {  // wave-uniform: this wave's symbols meet the chunk
                            for (uint32_t i = 0; __builtin_amdgcn_ballot_w64(i < pcn) != 0ull; i++) {
                                if (i < pcn) {
                                    const uint32_t e16 = *(lds_u16*)(uintptr_t)(pl_addr + 2u * i);
                                    const uint32_t pos = o + (e16 & 0xFFu);
                                    if (pos >= cbase && pos < cend)
                                        *(lds_u8*)(uintptr_t)(tile_addr + ocarry + (pos - cbase)) = (uint8_t)((e16 >> 8) ^ z);
                                }
                            }
                        }
                    }
                    __syncthreads();
                    const uint32_t total = ocarry + cc;
                    const bool last = !more && cend == lim;  // final flush of the block: store the ragged tail too
                    const uint32_t full = last ? total : (total & ~15u);
                    uint8_t* const dst = obase + gpos;
                    const uint32_t nunits = (full + 15u) >> 4;
                    for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                        const uint32_t lo = u << 4;
                        uint4* const src = reinterpret_cast<uint4*>(&L.tile[lo >> 2]);
                        const uint4 v = *src;
                        if (out_aligned && lo + 16u <= full && lo >= hskip) {
                            *reinterpret_cast<uint4*>(dst + lo) = make_uint4(v.x ^ z4, v.y ^ z4, v.z ^ z4, v.w ^ z4);
                        } else {
                            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                            for (uint32_t i = lo > hskip ? lo : hskip; i < lo + 16u && i < full; i++)
                                dst[i] = (uint8_t)((wv[(i - lo) >> 2] >> (8u * (i & 3u))) ^ z);
                        }
                        *src = make_uint4(0u, 0u, 0u, 0u);
                    }
                    const uint32_t tail = total - full;  // < 16: bytes that wait for the next chunk (still byte ^ z)
                    uint8_t tv = 0;
                    if ((uint32_t)tid < tail) tv = ob[full + tid];
                    if (full > 0u && tid < 64) {  // (tail < 16: the first wave alone; LDS keeps one wave's operations in order)
                        wave_lds_fence();
                        if (tid < 4) L.tile[(full >> 2) + (uint32_t)tid] = 0u;
                    }
                    __syncthreads();
                    // the carry lands in front of everything the next chunk's entries can touch (they start at byte `tail`):
                    // no barrier between this and the next chunk's entries
                    if (full > 0u && (uint32_t)tid < tail) ob[tid] = tv;
                    gpos += full;
                    ocarry = tail;
                    if (full > 0u) hskip = 0;
                    cbase = cend;
                }
            } else {
                // ---- a list ran over (more than DFA_PCAP symbols other than z in one subsequence): this window goes to
                // global memory the way every window used to: fill with z, then a third walk stores the other symbols ----
                if ((uint32_t)tid >= hskip && (uint32_t)tid < ocarry) obase[gpos + tid] = (uint8_t)(ob[tid] ^ z);  // bytes carried over
                __syncthreads();
                if (tid < 4) L.tile[tid] = 0u;
                uint8_t* const dst = oblk + produced;  // lim bytes
                {
                    uint32_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
                    if (head > lim) head = lim;
                    if ((uint32_t)tid < head) dst[tid] = (uint8_t)z;
                    const uint32_t body = (lim - head) >> 4;
                    uint4* const d4 = reinterpret_cast<uint4*>(dst + head);
                    for (uint32_t u = (uint32_t)tid; u < body; u += W) d4[u] = make_uint4(z4, z4, z4, z4);
                    const uint32_t t0 = head + (body << 4);
                    if ((uint32_t)tid < lim - t0) dst[t0 + tid] = (uint8_t)z;
                }
                // Workgroup-scope release + barrier: the fill of every wave is ordered before the single-byte stores any other
                // wave of this workgroup issues afterwards to the same lines (one CU, one L1, same-address order; a device-
                // scope fence here costs 8x the whole kernel, see k4_decode.hip).
                __threadfence_block();
                __syncthreads();
                const bool mine = nsym > 0u && o < lim;
                if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
                    uint32_t e = g << 6;
                    uint32_t t = o;
                    const uint32_t cmask = mine ? 7u : 0u, zmask = mine ? 0x18u : 0u;  // switched-off lanes store nothing
                    DFA_RJ_DECL;
                    auto stepS = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int sh = 26 - 4 * (j & 7);
                        DFA_RJ(rj, RECORD || SPARSE);
                        const uint32_t nib4 = sh >= 0 ? ((rj >> (sh >= 0 ? sh : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                        e = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib4));
                        const uint32_t nz = e & zmask;
                        if (nz != 0u) {
                            const uint32_t tp = t + ((e >> 14) & 3u);
                            if (nz == 0x08u) {
                                if (tp < lim) dst[tp] = (uint8_t)(e >> 16);
                            } else {
                                if (t < lim) dst[t] = (uint8_t)(e >> 16);
                                if (tp < lim) dst[tp] = (uint8_t)(e >> 24);
                            }
                        }
                        t += e & cmask;
                    };
                    [&]<int... Js>(std::integer_sequence<int, Js...>) {
                        (stepS(std::integral_constant<int, Js>{}), ...);
                    }(std::make_integer_sequence<int, 64>{});
                }
                // the next window starts in the middle of a 16-byte unit whose first bytes are in memory already
                const uint32_t np = produced + lim;
                gpos = np & ~15u;
                ocarry = np & 15u;
                hskip = ocarry;
                __syncthreads();
            }
            (void)more;
        } else {
        bool rec_win = rec_block;
        if constexpr (SLOT_CHECK) {
            if (rec_block && slot_ran_over) {  // workgroup-uniform: a slot ran over
                rec_win = false;
                zero_tile(1u);  // (the first 16 bytes hold the bytes carried over)
                if ((uint32_t)tid >= ocarry && tid < 16) ob[tid] = 0;
                __syncthreads();
            }
        }
        if (RECORD && rec_win) {
            // ---- phase B, recorded: close the slots up, in place, then flush ----
            // Symbol i of the window belongs at byte ocarry + i of the area (the first ocarry bytes are carried over from
            // the window before).  Lane l's destination ends below slot l + 1 and may lie anywhere in the slots of the
            // lanes before it: every lane reads its whole slot into registers, barrier, then writes.  Whole dwords are
            // written by one lane only; the <= 3 bytes in front of a lane's first whole dword and behind its last one
            // share their dwords with the neighbours and go out as single bytes.
            const uint32_t take = (o < lim) ? ((nsym < lim - o) ? nsym : lim - o) : 0u;
            const uint32_t dpos = ocarry + o;
            uint32_t hb = (4u - (dpos & 3u)) & 3u;
            if (hb > take) hb = take;
            const uint32_t nd = (take - hb) >> 2, tb = (take - hb) & 3u, shb = hb << 3;
            uint32_t w[33];
            [&]<int... Qs>(std::integer_sequence<int, Qs...>) __attribute__((always_inline)) {
                ([&]() __attribute__((always_inline)) {
                    constexpr int q = Qs;
                    w[2 * q] = 0;
                    w[2 * q + 1] = 0;
                    if (__builtin_amdgcn_ballot_w64(take > 8u * (uint32_t)q) != 0ull) {  // wave-uniform
                        const unsigned long long v2 =
                            *(__attribute__((address_space(3))) const unsigned long long*)(uintptr_t)(slot_addr + 8u * (uint32_t)q);
                        w[2 * q] = (uint32_t)v2;
                        w[2 * q + 1] = (uint32_t)(v2 >> 32);
                    }
                }(), ...);
            }(std::make_integer_sequence<int, 16>{});
            w[32] = 0;
#if DCZ_DFA_DBG
            {   // the bytes the compaction is about to copy against what the walk that wrote the slot said it recorded
                const uint32_t stamp = *(lds_cu32*)(uintptr_t)(slot_addr + (uint32_t)dfa_slot(W) - 4u);
                if (!beyond && nsym > 0u && take == nsym) {
                    uint32_t sum = 0;
#pragma unroll
                    for (int k = 0; k < 32; k++) {
                        const uint32_t have = take > 4u * (uint32_t)k ? take - 4u * (uint32_t)k : 0u;  // bytes of dword k that count
                        const uint32_t m = have >= 4u ? 0xFFFFFFFFu : ((1u << (8u * have)) - 1u);
                        sum = __builtin_amdgcn_sad_u8(w[k] & m, 0u, sum);
                    }
                    atomicAdd(&dfa_dbg[0], 1ull);
                    if ((stamp & 0xFFu) != (nsym & 0xFFu)) atomicAdd(&dfa_dbg[1], 1ull);
                    if ((stamp >> 8) != (sum & 0xFFFFFFu)) {
                        atomicAdd(&dfa_dbg[3], 1ull);
                        dfa_dbg[2] = ((unsigned long long)stamp << 32) | (sum << 8) | (nsym & 0xFFu);
                    }
                }
            }
#endif
            uint32_t tailv;
            {
                const uint32_t t0 = *(lds_cu32*)(uintptr_t)(slot_addr + 4u * nd);
                const uint32_t t1 = *(lds_cu32*)(uintptr_t)(slot_addr + 4u * nd + 4u);
                tailv = __builtin_amdgcn_alignbit(t1, t0, shb);
            }
            __syncthreads();
            {
                typedef __attribute__((address_space(3))) uint8_t lds_u8;
                const uint32_t dst = tile_addr + dpos, d4 = dst + hb;
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if ((uint32_t)i < hb) *(lds_u8*)(uintptr_t)(dst + (uint32_t)i) = (uint8_t)(w[0] >> (8 * i));
                [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
                    ([&]() __attribute__((always_inline)) {
                        constexpr int k = Ks;
                        if ((uint32_t)k < nd)
                            *(lds_u32*)(uintptr_t)(d4 + 4u * (uint32_t)k) = __builtin_amdgcn_alignbit(w[k + 1], w[k], shb);
                    }(), ...);
                }(std::make_integer_sequence<int, 32>{});
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if ((uint32_t)i < tb) *(lds_u8*)(uintptr_t)(d4 + 4u * nd + (uint32_t)i) = (uint8_t)(tailv >> (8 * i));
            }
            DFA_T(5);
            __syncthreads();
            DFA_T(6);
            const uint32_t total = ocarry + lim;
            const uint32_t full = more ? (total & ~15u) : total;  // final flush of the block: store the ragged tail too
            uint8_t* const dst = obase + gpos;
            const uint32_t nunits = (full + 15u) >> 4;
            for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                const uint32_t lo = u << 4;
                const uint32_t* src = &L.tile[lo >> 2];
                if (out_aligned && lo + 16u <= full && lo >= hskip) {
                    *reinterpret_cast<uint4*>(dst + lo) = make_uint4(src[0], src[1], src[2], src[3]);
                } else {
                    for (uint32_t i = lo > hskip ? lo : hskip; i < lo + 16u && i < full; i++) dst[i] = ob[i];
                }
            }
            const uint32_t tail = total - full;  // < 16
            uint8_t tv = 0;
            if ((uint32_t)tid < tail) tv = ob[full + tid];
            __syncthreads();
            if ((uint32_t)tid < tail) ob[tid] = tv;  // the ragged tail moves to the front of the area
            gpos += full;
            ocarry = tail;
            if (full > 0u) hskip = 0;  // (the unit shared with the region before has been written)
            DFA_T(7);
        } else {
        // ---- phase B: the same walk from the final entry state, two byte stores per step ----
        for (uint32_t cbase = 0; cbase < lim;) {
            uint32_t cc = lim - cbase;
            const uint32_t room = (uint32_t)LdsT::CAP - ocarry;
            if (cc > room) {  // workgroup-uniform: the rest of the window does not fit one flush: end it on a subsequence
                if (tid == 0) L.cend_vote = 0;
                __syncthreads();
                const uint32_t end0 = o + nsym;
                if (end0 > cbase + room / 2u && end0 <= cbase + room) atomicMax(&L.cend_vote, end0);
                __syncthreads();
                const uint32_t v = L.cend_vote;
                cc = v != 0u ? v - cbase : room;  // (a subsequence completes < 130 symbols: a boundary always exists)
            }
            const uint32_t cend = cbase + cc;
            const uint32_t tshift = ocarry - cbase;  // tile index = window symbol index + tshift
            // a lane takes part in this flush when its first symbol lies inside it (flushes end on subsequence
            // boundaries; at the end of the block the last lane may run past `lim` into the tile's slack)
            const bool mine = nsym > 0u && o >= cbase && o < cend;
            // Symbols are collected in a 64-bit register and leave as whole dwords: a lane stores the tile dword that
            // holds its k-th .. (k+3)-th byte when it has them (checked every second step: at most 3 + 4 bytes wait).
            // The dword a lane shares with its left neighbour gets the lane's plain store with the neighbour's bytes as
            // zeros; what is left in the register after the walk (the lane's bytes of the dword it shares with its right
            // neighbour) is ORed in after a barrier, when all plain stores are done.  Dwords that only receive ORs were
            // zeroed by the flush that read them last.
            uint32_t ab = 0, k8 = 0, alo = 0;  // ab: LDS byte address of the dword being collected
            if (!(DCZ_DFA_ABL & 8) && __builtin_amdgcn_ballot_w64(mine) != 0ull) {
                uint32_t e = g << 6;
                const uint32_t p0 = mine ? o + tshift : 0u;
                ab = tile_addr + (p0 & ~3u);
                k8 = (p0 & 3u) << 3;
                uint32_t ahi = 0;
                // (the first lane of a flush continues the bytes the flush before left in the tile)
                if (mine && o == cbase) alo = *(lds_u32*)(uintptr_t)ab & ((1u << k8) - 1u);
                const uint32_t tmask = mine ? 0x18u : 0u;  // switched-off lanes collect nothing
                DFA_RJ_DECL;
                auto stepB = [&](auto jc) __attribute__((always_inline)) {  // two nibbles: up to four symbols
                    constexpr int j = 2 * decltype(jc)::value;
                    constexpr int sh0 = 26 - 4 * (j & 7), sh1 = 26 - 4 * ((j + 1) & 7);
                    DFA_RJ(rj, RECORD || SPARSE);
                    const uint32_t nib0 = (rj >> sh0) & 0x3Cu;
                    const uint32_t nib1 = sh1 >= 0 ? ((rj >> (sh1 >= 0 ? sh1 : 0)) & 0x3Cu) : ((rj << 2) & 0x3Cu);
                    const uint32_t e0 = *(lds_cu32*)(uintptr_t)(t_addr + ((e & AMASK) | nib0));
                    e = *(lds_cu32*)(uintptr_t)(t_addr + ((e0 & AMASK) | nib1));
                    const uint32_t c0 = e0 & tmask;  // 8 * symbols of the first nibble (bytes past the count are zero)
                    const uint32_t w = ((e >> 16) << c0) | (e0 >> 16);
                    const unsigned long long v = (unsigned long long)w << k8;
                    alo |= (uint32_t)v;
                    ahi |= (uint32_t)(v >> 32);
                    k8 += c0 + (e & tmask);
                    if (k8 >= 32u) {
#if !(DCZ_DFA_ABL & 2)
                        *(lds_u32*)(uintptr_t)ab = alo;
#endif
                        alo = ahi;
                        ahi = 0;
                        k8 -= 32u;
                        ab += 4u;
                    }
                };
                [&]<int... Js>(std::integer_sequence<int, Js...>) {
                    (stepB(std::integral_constant<int, Js>{}), ...);
                }(std::make_integer_sequence<int, 32>{});
            }
            __syncthreads();
            if (k8 != 0u)
                __hip_atomic_fetch_or((lds_u32*)(uintptr_t)ab, alo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            DFA_T(5);
#if DCZ_K4_PROF
            pacc[10]++;
#endif
            __syncthreads();
            DFA_T(6);
            const uint32_t total = ocarry + cc;
            const bool last = !more && cend == lim;  // final flush of the block: store the ragged tail too
            const uint32_t full = last ? total : (total & ~15u);
            uint8_t* const dst = obase + gpos;
            const uint32_t nunits = (full + 15u) >> 4;
            for (uint32_t u = (uint32_t)tid; u < nunits; u += W) {
                const uint32_t lo = u << 4;
                const uint32_t* src = &L.tile[lo >> 2];
                if (out_aligned && lo + 16u <= full && lo >= hskip) {
                    *reinterpret_cast<uint4*>(dst + lo) = make_uint4(src[0], src[1], src[2], src[3]);
                } else {
                    for (uint32_t i = lo > hskip ? lo : hskip; i < lo + 16u && i < full; i++) dst[i] = ob[i];
                }
                *reinterpret_cast<uint4*>(&L.tile[lo >> 2]) = make_uint4(0u, 0u, 0u, 0u);
            }
            const uint32_t tail = total - full;  // < 16
            uint8_t tv = 0;
            if ((uint32_t)tid < tail) tv = ob[full + tid];
            __syncthreads();
            if (full > 0u) {  // the ragged tail moves to the front of the (zeroed) tile
                if (tid < 4) L.tile[(full >> 2) + (uint32_t)tid] = 0u;
                if ((uint32_t)tid < tail) ob[tid] = tv;
            }
            gpos += full;
            ocarry = tail;
            if (full > 0u) hskip = 0;  // (the unit shared with the region before has been written)
            cbase = cend;
            DFA_T(7);
            if (cbase < lim) __syncthreads();  // (the next flush reads the bytes carried over)
        }
        }
        }
        produced += lim;
        entry0 = next_entry;
        wbyte += (unsigned long long)W * 32ull;
        if (exhausted && produced < orig) {
            // The payload is used up but the chunk wants more symbols: the reference keeps reading zero bits
            // (TableBasedHuffmanDecoder.java:204-208).  A codeword that the end of the payload cut (damaged streams only:
            // the encoder pads to a byte) is completed by those zeros -- the walk below, from the state after the last
            // subsequence that starts inside the payload; everything after it is the all-zero codeword = first canonical
            // symbol, forever.  Zeros that leave the code tree are the reference's "decode error at position produced".
            __syncthreads();
            if ((uint32_t)tid < ocarry && (uint32_t)tid >= hskip) obase[gpos + tid] = (uint8_t)(ob[tid] ^ (SPARSE ? L.zsym : 0u));  // unflushed tail
            const uint8_t z = L.zsym;
            uint32_t fs = z;
            if (tid == 0) {
                uint32_t st = wexit;
                while (st != 0u) {
                    const uint32_t e = L.T[st * 16u];
                    if ((e & (SPARSE ? 7u : 3u)) != 0u) {  // the cut codeword is complete: it is the nibble's first symbol
                        fs = (e >> 16) & 0xFFu;            // (SPARSE: a codeword that was under way is never z)
                        break;
                    }
                    st = (e >> 6) & 0xFFu;
                    if (st == errst) {
                        status = DCZ_E_BADSTREAM;
                        errpos = (long long)produced;
                        break;
                    }
                }
            }
            for (uint32_t i = produced + (uint32_t)tid; i < orig; i += W) oblk[i] = (i == produced) ? (uint8_t)fs : z;
            break;
        }
        __syncthreads();
    }

    if constexpr (COUNT) {
        if (tid == 0) {
            const uint64_t i = (uint64_t)b * sdp->rmax + reg;
            const bool bad = unusable || wexit == errst || rentry == errst;
            sdp->entry[i] = rentry;
            sdp->count[i] = (uint32_t)(rcount > 0xFFFFFFFFull ? 0xFFFFFFFFull : rcount);
            sdp->exit[i] = bad ? 0xFFFFFFFFu : wexit;
        }
        return;
    }
    if (tid == 0 && (!SPLIT || status != DCZ_OK)) {  // (a split block got its status from k4_split_scan)
        d_status[b] = status;
        if (d_errpos) d_errpos[b] = errpos + (SPLIT ? (long long)sdp->off[(uint64_t)b * sdp->rmax + reg] : 0ll);
    }
#if DCZ_K4_PROF
    if (tid == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&dfa_prof[i], pacc[i]);
#endif
}

// counting pass of the split decoder for the blocks the automaton takes (between k4_split_setup and k4_split_scan)
void launch_count_dfa(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                      const uint32_t* d_orig_size, const uint8_t* d_len, uint8_t* d_cls, const SplitDesc* d_sd, hipStream_t s) {
    hipLaunchKernelGGL((k4_dfa<DCZ_DFA_W, 0, 2, false>), dim3(SPLIT_GRID), dim3(DCZ_DFA_W), 0, s, d_comp,
                       reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_orig_size, d_len, (size_t)0,
                       (uint8_t*)nullptr, (int32_t*)nullptr, (long long*)nullptr, d_cls, d_sd);
}

void launch_decode_dfa(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                       const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                       int32_t* d_status, int64_t* d_errpos, const DecodeWs& ws, uint32_t split_grid, hipStream_t s) {
    if (K == 0) return;
    const unsigned long long* off = reinterpret_cast<const unsigned long long*>(d_comp_off);
    long long* ep = reinterpret_cast<long long*>(d_errpos);
    if (split_grid) {  // one workgroup per (block, region)
        hipLaunchKernelGGL((k4_dfa<DCZ_DFA_W, DFA_OC_DECODE, 1, false>), dim3(split_grid), dim3(DCZ_DFA_W), 0, s, d_comp, off,
                           d_comp_size, d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls, ws.sdesc);
        return;
    }
    static const uint32_t few_below = [] {
        const char* e = getenv("DCZ_DFA_FEW_BLOCKS_BELOW");  // tuning knob
        return e ? (uint32_t)atoi(e) : 768u;
    }();
    if (K >= few_below) {  // 4 workgroups of 4 waves per CU
        hipLaunchKernelGGL((k4_dfa<DCZ_DFA_W, DFA_OC_DECODE, 0, false>), dim3(K), dim3(DCZ_DFA_W), 0, s, d_comp, off,
                           d_comp_size, d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls, (const SplitDesc*)nullptr);
#if DCZ_K4_SPARSE_DFA
        hipLaunchKernelGGL((k4_dfa<DCZ_DFA_W, DCZ_DFA_SPARSE_OC, 0, true>), dim3(K), dim3(DCZ_DFA_W), 0, s, d_comp, off, d_comp_size,
                           d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls, (const SplitDesc*)nullptr);
#endif
    } else {  // few blocks: one 16-wave workgroup per block owns its CU (a window is 32 KiB of payload)
        hipLaunchKernelGGL((k4_dfa<1024, DCZ_DFA_RECORD ? -1 : 4 * DCZ_DFA_OC, 0, false>), dim3(K), dim3(1024), 0, s, d_comp, off, d_comp_size,
                           d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls, (const SplitDesc*)nullptr);
#if DCZ_K4_SPARSE_DFA
        hipLaunchKernelGGL((k4_dfa<1024, 4 * DCZ_DFA_SPARSE_OC, 0, true>), dim3(K), dim3(1024), 0, s, d_comp, off, d_comp_size,
                           d_orig_size, d_len, out_stride, d_out, d_status, ep, ws.cls, (const SplitDesc*)nullptr);
#endif
    }
}

}  // namespace dcz

#if DCZ_DFA_DBG
extern "C" void dcz_debug_dfa_dbg(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(dcz::dfa_dbg), sizeof(dcz::dfa_dbg));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(dcz::dfa_dbg), z, sizeof(z));
    }
}
#endif
#if DCZ_K4_PROF
extern "C" void dcz_debug_dfa_prof(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(dcz::dfa_prof), sizeof(dcz::dfa_prof));
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(dcz::dfa_prof), z, sizeof(z));
    }
}
#endif
