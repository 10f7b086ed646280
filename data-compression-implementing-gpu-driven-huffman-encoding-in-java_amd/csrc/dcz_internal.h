// dcz_internal.h -- shared constants, device helpers and kernel launchers of libdczhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dcz.h"

namespace dcz {

constexpr uint32_t SEG = DCZ_SEGMENT_BYTES;  // bytes of one block handled by one wave in K1 and K3
constexpr int WAVE = 64;                     // gfx950 wavefront

// ---- host-side launchers (each defined next to its kernel) ---------------------------------
// K1: per-segment 256-bin histograms.  seg_hist is [nseg][256] u16 (a segment holds <= 32768 bytes).
void launch_histogram(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t segs_per_block, uint64_t nseg,
                      uint16_t* seg_hist, hipStream_t s, uint8_t* copy_out = nullptr);
// Histogram of a byte window into 256 x i64 (single-block API; sums the segment rows on the device).
void launch_sum_hist(const uint16_t* seg_hist, uint64_t nseg, int64_t* d_hist, hipStream_t s);

// Launch-shape hint.  Two kernels serve blocks that most inputs do not have (k4_fixed: fixed-length complete codes,
// k3_copy_identity: 256 symbols of 8 bits) with a flat grid of one workgroup per tile of EVERY block, because that is what
// copies fastest; on inputs without such blocks those grids cost 0.06-0.11 ms per 8 GiB for nothing.  The classification
// kernels (k4_classify, k2_codebuild) publish what they met in host-mapped words: the sequence number of the last call
// that had such a block, and the sequence number of the last call that was classified at all.  The host reads the words
// without synchronising and compares them WITH EACH OTHER, never with the number of calls it has issued: the words only
// move when a call completes, so the decision is "what the last completed calls looked like" however many calls are
// queued behind them (no news = same as before).  Both shapes are correct for any input and the persistent shape still
// spreads its work over the whole grid; the hint only chooses the cheaper one.
struct ShapeHint {
    uint32_t* dev = nullptr;   // device address of the "had such a block" word (nullptr: nothing is recorded)
    uint32_t* done = nullptr;  // decoder: device address of the "classified" word.  The encoder needs none: every block
                               // of k2_codebuild stores into dev[0] (identity code) or dev[1] (any other code)
    uint32_t epoch = 0;        // this call's sequence number (> 0)
    bool likely = true;        // flat grid
    // Encoder only.  `in_place`: K1 copied the input to the output at the same offsets (launch_histogram's copy_out)
    // because the last completed calls had nothing but identity blocks; k3_copy_identity then leaves alone every identity
    // block whose payload offset equals its input offset.
    bool in_place = false;
    unsigned long long in_offset = 0;  // byte offset, inside the call's input, of the block range a launch covers
};
constexpr uint32_t HINT_PERSIST_GRID = 1024;

// K2: per-block code build + per-segment bit offsets.
void launch_codebuild(const uint16_t* seg_hist, const int64_t* d_hist_in, size_t n, size_t block_bytes,
                      uint32_t segs_per_block, uint32_t K, uint8_t* d_len, uint32_t* d_code, uint8_t* d_maxlen,
                      uint32_t* d_comp_size, uint64_t* d_seg_bitoff, int32_t* d_status, hipStream_t s,
                      const ShapeHint& hint = ShapeHint());
// Canonical codes from stored lengths (CH.generateCanonicalCodesFromLengths), one block per workgroup.
void launch_codes_from_lengths(const int32_t* d_len32, uint32_t* d_code, int32_t* d_status, hipStream_t s);
// Exclusive scan of comp_size -> comp_off, total, capacity check.
// d_carry_in (nullable): payload bytes that precede this block range; d_total receives carry + sum.
void launch_offsets(const uint32_t* d_comp_size, uint32_t K, uint64_t* d_comp_off, uint64_t* d_total,
                    const uint64_t* d_carry_in, size_t out_cap, int32_t* d_status, hipStream_t s);

// K3: encode.
void launch_encode(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t segs_per_block, uint32_t K,
                   const uint8_t* d_len, const uint32_t* d_code, const uint8_t* d_maxlen, const uint64_t* d_comp_off,
                   const uint64_t* d_seg_bitoff, const int32_t* d_status, uint8_t* d_out, hipStream_t s,
                   const ShapeHint& hint = ShapeHint());

// K4: decode.  Region table of the split decoder (k4_split.hip): a block cut into regions of region_bytes of payload,
// one workgroup per region; the arrays are [block][rmax].
#ifndef DCZ_K4_MEDIUM_DFA
#define DCZ_K4_MEDIUM_DFA 1  // medium class: k4_dfa.hip decodes (and, for split blocks, counts) the tables it can take
#endif
#ifndef DCZ_K4_CLS2_A
#define DCZ_K4_CLS2_A 4  // class boundary medium / short codes: medium when orig * A <= csize * B (>= 8 * A / B bits/symbol)
#endif
#ifndef DCZ_K4_CLS2_B
#define DCZ_K4_CLS2_B 9
#endif
#ifndef DCZ_K4_SPARSE_DFA
#define DCZ_K4_SPARSE_DFA 1  // sparse short-code blocks (one 1-bit symbol, < 1.3 bits/symbol): k4_dfa's SPARSE instantiation
#endif

struct SplitDesc {
    uint32_t* entry;   // bits from the region's first payload bit (virtual) to its first codeword
    uint32_t* count;   // symbols the region produces
    uint32_t* exit;    // bits from the region's end to the first codeword past it, 0xFFFFFFFF = not usable
    uint32_t* off;     // output offset of the region inside its chunk
    uint32_t* nreg;    // [block] regions in use, 0 = the block is not split
    uint32_t* rbase;   // [block + 1] first workgroup of the block's regions in the region grids (k4_split_setup)
    uint32_t rmax;
    uint32_t nblk;     // blocks of the call
    unsigned long long region_bytes;
};
// Region grids are one-dimensional and dense: workgroup i serves region i - rbase[b] of the block b with
// rbase[b] <= i < rbase[b + 1] (a [block][rmax] grid launched 30x more workgroups than there are regions: 0.09 ms per
// launch for 32 chunks of 32 MiB).  SPLIT_GRID workgroups cover any call: sum ceil((csize + 15) / S) <= REGIONS + 2 * blocks.
__device__ __forceinline__ bool split_region_of(const SplitDesc* sdp, uint32_t i, uint32_t& b, uint32_t& reg) {
    const uint32_t n = sdp->nblk;
    const uint32_t* rb = sdp->rbase;
    if (i >= rb[n]) return false;
    uint32_t lo = 0, hi = n;  // rb[lo] <= i < rb[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rb[mid] <= i) lo = mid;
        else hi = mid;
    }
    b = lo;
    reg = i - rb[lo];
    return true;
}
constexpr size_t SPLIT_MAX_BLOCKS = 128;      // blocks per call below which a call may be split
constexpr size_t SPLIT_REGIONS = 2048;        // regions aimed at per call
constexpr size_t SPLIT_ENTRIES = SPLIT_MAX_BLOCKS * (SPLIT_REGIONS + 2);
constexpr uint32_t SPLIT_GRID = (uint32_t)(SPLIT_REGIONS + 2 * SPLIT_MAX_BLOCKS);
// Workspace of one decode call (device memory, see decode_ws_bytes).
struct DecodeWs {
    uint8_t* cls;        // K x u8: class byte of every block (k4_classify, then the probe launch / k4_split_scan)
    SplitDesc* sdesc;    // device copy of the region table descriptor of this call
    uint32_t* split;     // 4 x SPLIT_ENTRIES + SPLIT_MAX_BLOCKS + (SPLIT_MAX_BLOCKS + 1) u32
    ShapeHint fixed;     // launch shape of k4_fixed (set by the caller of launch_decode)
};
inline size_t decode_ws_bytes(size_t K) {
    return ((K + 255) & ~(size_t)255) + 256 + (4 * SPLIT_ENTRIES + 2 * SPLIT_MAX_BLOCKS + 4) * 4;
}
inline DecodeWs decode_ws_at(void* base, size_t K) {
    uint8_t* p = static_cast<uint8_t*>(base);
    const size_t kb = (K + 255) & ~(size_t)255;
    return DecodeWs{p, reinterpret_cast<SplitDesc*>(p + kb), reinterpret_cast<uint32_t*>(p + kb + 256), ShapeHint()};
}
void launch_decode(const uint8_t* d_comp, size_t comp_bytes, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                   const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                   int32_t* d_status, int64_t* d_errpos, const DecodeWs& ws, hipStream_t s);
// k4_fixed.hip: class byte per block + bounds check of the footer fields; analytic decode of fixed-length complete codes
void launch_classify(const uint8_t* d_len, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                     const uint32_t* d_orig_size, size_t comp_bytes, size_t out_stride, uint32_t K, const DecodeWs& ws,
                     int32_t* d_status, int64_t* d_errpos, hipStream_t s);
void launch_count_dfa(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                      const uint32_t* d_orig_size, const uint8_t* d_len, uint8_t* d_cls, const SplitDesc* d_sd, hipStream_t s);
void launch_decode_fixed(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                         const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                         const DecodeWs& ws, hipStream_t s);

// k4_split.hip: regions of few large blocks: counting pass + proof/scan (fills the tables of sd, sets class byte 2)
void launch_split_count(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                        const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, uint8_t* d_cls, int32_t* d_status,
                        int64_t* d_errpos, const SplitDesc& sd, SplitDesc* d_sd, hipStream_t s);
// k4_dfa.hip: nibble automaton for the medium class (one workgroup per block)
void launch_decode_dfa(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                       const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                       int32_t* d_status, int64_t* d_errpos, const DecodeWs& ws, uint32_t split_grid, hipStream_t s);

// K5: SHA-256 of every block -> digests[K][32].
void launch_sha256(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t K, uint8_t* d_digests, hipStream_t s);

// Generators.
void launch_fill_java_random(uint8_t* d, size_t n, int64_t seed, uint64_t start, hipStream_t s);
void launch_fill_text(uint8_t* d, size_t n, uint64_t seed, uint64_t start, hipStream_t s);
void launch_fill_lowentropy(uint8_t* d, size_t n, uint64_t seed, uint64_t start, hipStream_t s);

#if defined(__HIPCC__)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // native vector: what the non-temporal builtins take

// ---- device helpers --------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Wave-wide inclusive scan of a u32 with DPP (row_shr 1/2/4/8, row_bcast15, row_bcast31): no LDS traffic.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan_u32(v), 63);
}

__device__ __forceinline__ uint64_t wave_reduce_add_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// Load one 16-byte chunk of the payload (virtual byte vb), zero outside [vlo, vhi).
__device__ __forceinline__ uint4 load_chunk16(const uint8_t* vbase, unsigned long long vb, unsigned long long vlo,
                                            unsigned long long vhi) {
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
    return v;  // little-endian as loaded: the byte swap to MSB-first dwords happens when the registers are staged,
               // so that nothing waits for the load where it is issued
}

// Wave-scope ordering of LDS traffic between lanes of ONE wave (LDS executes a wave's operations in
// issue order; this only stops the compiler from reordering across it).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif  // __HIPCC__

}  // namespace dcz
