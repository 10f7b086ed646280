// k4_split.hip -- K4 for FEW LARGE blocks: one block decoded by many workgroups (gfx950).
//
// The reference decodes a chunk with one thread (CpuCompressionService.decodeChunkParallel,
// service/cpu/CpuCompressionService.java:511-556, runs TableBasedHuffmanDecoder.decode per chunk) and its chunk sizes are
// 16 MiB (app/src/main/resources/application.conf:10) and 32 MiB (cli/DataCompCLI.java:35): a file of a few such chunks, or
// one dcz_decode_block call, would keep 1..K of the 256 CUs busy with one workgroup per block.  The format stores no
// intra-block offsets, so the split is found the way windows are synchronised inside a workgroup, one level up:
//   pass 1 (k4_split_count): the payload of a block is cut into REGIONS of S bytes (S a multiple of the 8 KiB window, at
//     least 64 KiB; the size is chosen on the device from the payload bytes the call really has).  The workgroup of region r
//     first walks the window in front of its region from a guessed entry (a jump walk: jt[12-bit
//     window] = bits and number of the complete codewords inside it, one codeword at a time in the last 12 bits of a
//     subsequence); Huffman streams re-synchronise within a few codewords, so the first codeword boundary at or past the
//     region start that this walk finds is the region's entry with overwhelming probability.  It then walks its region
//     window by window (phase A only, nothing is decoded) and reports (entry, symbols owned, exit = first codeword
//     boundary at or past the region's end).
//   k4_split_scan: entries are PROVEN, not trusted: region 0's entry is exact, and exit(r-1) == entry(r) for every r makes
//     every entry exact by induction.  A block that fails the check (a stream that does not self-synchronise, a damaged
//     stream, a pattern without a codeword) keeps class 0 and is decoded by the one-workgroup-per-block kernels as
//     before -- error positions and the exact-entry path included.  For a proven block an exclusive scan of the symbol
//     counts gives every region its output offset (clipped to the chunk's originalSize; the last region also produces the
//     symbols the reference reads from the zero padding, TableBasedHuffmanDecoder.java:204-208), class byte 2.
//   pass 2: the table-walk kernels of k4_decode.hip run once per region with (entry bit, output offset, symbol count).
// Fixed-length blocks never come here (k4_fixed.hip is already one workgroup per 16 KiB of output).
#include <cstdlib>
#include <utility>

#include "dcz_internal.h"

namespace dcz {

#ifndef DCZ_SP_TBJ
#define DCZ_SP_TBJ 12
#endif
#ifndef DCZ_SP_ROUNDS
#define DCZ_SP_ROUNDS 12  // rounds a window may take before the block is left to the sequential kernels
#endif

constexpr int SP_W = 256;
constexpr int SP_SUB_BITS = 256;
constexpr uint32_t SP_WIN_BITS = SP_W * SP_SUB_BITS;  // 65536: one window = 8 KiB of payload

template <int TBJ>
struct SpLds {
    uint16_t st[1 << TBJ];   // single-symbol table len << 8 | symbol (only while jt is built)
    uint16_t jt[1 << TBJ];   // bits | count << 8 of the complete codewords inside the window, 0 = none
    uint32_t head0[SP_W + 1];
    uint16_t exits[SP_W];
    unsigned long long lim[40];
    uint32_t first[34], cnt[34], offs[34];
    uint32_t flag[3];
    uint32_t wsum[SP_W / 64];
    uint8_t len8[256];
    uint32_t maxlen, anybad;
    int bad_table;
};

__device__ __forceinline__ uint32_t sp_hi_shl(unsigned long long pair, uint32_t sh) {
    unsigned long long r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(pair));
    return (uint32_t)(r >> 32);
}
__device__ __forceinline__ uint32_t sp_select(uint32_t v, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}
#define SP_EQ(a, b) __builtin_amdgcn_uicmp((uint32_t)(a), (uint32_t)(b), 32)

// length of the first codeword of window w among lengths > TB (canonical search over the left-aligned limits), 0 = none
template <int TB, class LdsT>
__device__ __attribute__((noinline)) uint32_t sp_slow_len(const LdsT& L, uint32_t win32) {
    const uint32_t maxlen = L.maxlen;
    const unsigned long long w = win32;
    for (uint32_t l = TB + 1; l <= maxlen; l++)
        if (w < L.lim[l]) return l;
    return 0;
}

__global__ __launch_bounds__(SP_W) void k4_split_count(const uint8_t* __restrict__ comp,
                                                       const unsigned long long* __restrict__ d_comp_off,
                                                       const uint32_t* __restrict__ d_comp_size,
                                                       const uint32_t* __restrict__ d_orig_size,
                                                       const uint8_t* __restrict__ d_len, const uint8_t* __restrict__ d_cls,
                                                       const SplitDesc* __restrict__ sdp) {
    constexpr int TBJ = DCZ_SP_TBJ;
    constexpr int W = SP_W;
    __shared__ SpLds<TBJ> L;
    uint32_t b, r;
    if (!split_region_of(sdp, blockIdx.x, b, r)) return;  // workgroup-uniform
    const int tid = (int)threadIdx.x;
    if (d_cls[b] != 0) return;  // fixed-length or rejected block (workgroup-uniform)
    const SplitDesc sd = *sdp;  // (region size chosen on the device from the payload bytes this call really has)
    const uint32_t csize = d_comp_size[b];
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)d_comp_off[b];
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew, vhi = (unsigned long long)skew + csize;
    const unsigned long long S = sd.region_bytes;
    const unsigned long long nreg = (vhi + S - 1) / S;
    if (nreg < 2 || r >= nreg || nreg > sd.rmax) return;  // a block of one region is not split

    // ---- tables (as in k4_decode.hip) ----
    if (tid < 34) L.cnt[tid] = 0;
    if (tid == 0) {
        L.bad_table = 0;
        L.anybad = 0;
    }
    __syncthreads();
    {
        const uint32_t l = d_len[(uint64_t)b * 256u + tid];
        L.len8[tid] = (uint8_t)l;
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
        if (kraft > (1ull << 32) || mx == 0) L.bad_table = 1;
#if DCZ_K4_MEDIUM_DFA
        // blocks the nibble automaton takes are counted by its own counting pass (k4_dfa.hip, MODE 2: same test)
        {
            const uint32_t orig_blk = d_orig_size[b];
            const bool long_codes = (unsigned long long)csize * 16ull >= (unsigned long long)orig_blk * 13ull;
            const bool medium = (unsigned long long)orig_blk * (unsigned long long)DCZ_K4_CLS2_A <=
                                (unsigned long long)csize * (unsigned long long)DCZ_K4_CLS2_B;
            uint32_t ni = 0, states = 0;
            for (int l = 31; l >= 0; l--) {
                ni = (L.cnt[l + 1] + ni + 1u) / 2u;
                states += ni;
            }
            L.anybad = (!L.bad_table && !long_codes && medium && L.cnt[1] == 0u && states <= 255u) ? 2u : 0u;
        }
#endif
    }
    __syncthreads();
    if (L.anybad == 2u) return;  // workgroup-uniform
    uint32_t* const o_entry = sd.entry + (uint64_t)b * sd.rmax + r;
    uint32_t* const o_count = sd.count + (uint64_t)b * sd.rmax + r;
    uint32_t* const o_exit = sd.exit + (uint64_t)b * sd.rmax + r;
    if (L.bad_table) {  // the sequential kernels report DCZ_E_BADTABLE / the empty-table error
        if (tid == 0) *o_exit = 0xFFFFFFFFu;
        return;
    }
    for (int idx = tid; idx < (1 << TBJ); idx += W) {  // only lengths matter here
        uint32_t e = 0;
        for (uint32_t l = 1; l <= (uint32_t)TBJ; l++) {
            const uint32_t c = (uint32_t)idx >> (TBJ - l);
            const uint32_t f = L.first[l];
            if (c >= f && c - f < L.cnt[l]) {
                e = l | 0x100u;  // one symbol of l bits, in jt's format
                break;
            }
        }
        L.st[idx] = (uint16_t)e;
    }
    __syncthreads();
    constexpr uint32_t JMASK = (1u << TBJ) - 1u;
    for (int idx = tid; idx < (1 << TBJ); idx += W) {
        uint32_t pos = 0, cnt = 0;
        while (pos < (uint32_t)TBJ) {
            const uint32_t len = L.st[((uint32_t)idx << pos) & JMASK] & 0xFFu;
            if (len == 0 || len > (uint32_t)TBJ - pos) break;
            pos += len;
            cnt++;
        }
        L.jt[idx] = (uint16_t)(cnt ? ((cnt << 8) | pos) : 0u);
    }
    typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
    const uint32_t jt_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.jt[0]));
    const uint32_t st_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)(&L.st[0]));
    __syncthreads();

    // ---- the windows of this region: fixed 8 KiB grid in virtual coordinates (region starts are window aligned) ----
    const unsigned long long pay_end_bits = vhi << 3;
    const unsigned long long Bv0 = 8ull * r * S, Bv1 = Bv0 + 8ull * S;
    const unsigned long long end_bits = Bv1 < pay_end_bits ? Bv1 : pay_end_bits;
    unsigned long long wbase = (r == 0) ? 0ull : Bv0 - SP_WIN_BITS;  // r > 0: the window in front of the region first
    uint32_t g0 = (r == 0) ? 8u * skew : 0u;                        // lane 0's entry: exact for region 0, a guess otherwise
    bool preroll = r != 0;
    uint32_t entry_x = 8u * skew, exit_x = 0;
    uint32_t mycount = 0;
    bool mybad = false, gaveup = false;

    uint4 pre[2];
    uint4 pre_m = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](unsigned long long wb) {
        const unsigned long long c0 = wb >> 7;
#pragma unroll
        for (int c = 0; c < 2; c++) pre[c] = load_chunk16(vbase, (c0 + (unsigned long long)(tid * 2 + c)) << 4, vlo, vhi);
        if (tid == W - 1) pre_m = load_chunk16(vbase, (c0 + (unsigned long long)(W * 2)) << 4, vlo, vhi);
    };
    prefetch(wbase);
    while (true) {
        uint32_t R[9];
        R[0] = bswap32(pre[0].x);
        R[1] = bswap32(pre[0].y);
        R[2] = bswap32(pre[0].z);
        R[3] = bswap32(pre[0].w);
        R[4] = bswap32(pre[1].x);
        R[5] = bswap32(pre[1].y);
        R[6] = bswap32(pre[1].z);
        R[7] = bswap32(pre[1].w);
        L.head0[tid] = R[0];
        if (tid == W - 1) L.head0[W] = bswap32(pre_m.x);
        __syncthreads();
        R[8] = L.head0[tid + 1];
        // (the next window of the grid is known: its loads fly while this one is walked)
        const unsigned long long wend = preroll ? Bv0 : end_bits;  // subsequences that start at or past it take no part
        const bool lastwin = wbase + SP_WIN_BITS >= wend;
        if (!lastwin || preroll) prefetch(wbase + SP_WIN_BITS);

        const bool beyond = wbase + (unsigned long long)tid * SP_SUB_BITS >= wend;
        uint32_t g = (tid == 0) ? g0 : 0u;
        uint32_t x = 0, nsym = 0;
        bool bad = false, need = !beyond;
        uint32_t round = 0;
        while (true) {
            unsigned long long walking = __builtin_amdgcn_ballot_w64(need);
            const unsigned long long walking0 = walking;
            uint32_t stA = g;  // rel | n << 8
            auto walk = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const unsigned long long pair = ((unsigned long long)R[k] << 32) | R[k + 1];
                auto step = [&](unsigned long long am) __attribute__((always_inline)) {
                    const uint32_t w = sp_hi_shl(pair, stA);
                    // A jump may only take codewords that START inside the subsequence: in its last TBJ bits the walk goes
                    // one codeword at a time (st), so that the exit is the first codeword boundary at or past the end
                    // whatever the entry was -- chains of different phase inside a run of short codewords would otherwise
                    // never land on the same bits (measured: 2 of 64 low-entropy blocks took > 12 rounds somewhere).
                    uint32_t base = jt_addr;
                    if constexpr (k == 7) base = ((stA & 0xFFu) > 32u - (uint32_t)TBJ) ? st_addr : jt_addr;
                    uint32_t e = *(lds_cu16*)(uintptr_t)(base + ((w >> (31 - TBJ)) & (uint32_t)(((1 << TBJ) - 1) << 1)));
                    asm("" : "+v"(e));
                    const unsigned long long esc = SP_EQ(e, 0u) & am;
                    if (__builtin_expect(esc != 0ull, 0)) {  // rare: a long codeword, or none
                        bool dd = false;
                        if (need && (stA & 0xE0u) == 0u && e == 0u) {
                            const uint32_t len = sp_slow_len<TBJ>(L, w);
                            e = len | 0x100u;
                            if (len == 0u) {  // no codeword on this parse
                                dd = true;
                                e = 0u;
                            }
                        }
                        walking &= ~__builtin_amdgcn_ballot_w64(dd);
                    }
                    stA += sp_select(e, am);
                };
#pragma unroll
                for (int u = 0; u < 3; u++) step(SP_EQ(stA & 0xE0u, 0u) & walking);
                while (true) {
                    const unsigned long long am = SP_EQ(stA & 0xE0u, 0u) & walking;
                    if (am == 0ull) break;
                    step(am);
                }
                stA -= 32u;
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (walk(std::integral_constant<int, Ks>{}), ...);
            }(std::make_integer_sequence<int, 8>{});
            if (need) {
                bad = ((walking0 & ~walking) >> (tid & 63)) & 1ull;
                x = bad ? 0u : (stA & 0xFFu);
                nsym = stA >> 8;
            }
            L.exits[tid] = (uint16_t)x;
            if (tid == 0) L.flag[(round + 1u) % 3u] = 0;
            __syncthreads();
            if (round > 0u && L.flag[round % 3u] == 0u) break;
            const uint32_t ng = (tid == 0) ? g0 : (uint32_t)L.exits[tid - 1];
            need = (ng != g) && !beyond;
            g = ng;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull && (tid & 63) == 0) L.flag[(round + 1u) % 3u] = 1;
            round++;
            if (round == (uint32_t)DCZ_SP_ROUNDS) {  // workgroup-uniform: not synchronised
                gaveup = true;
                break;
            }
        }
        if (gaveup) break;
        // index of the last subsequence that took part (wend - wbase is a multiple of 256 unless the payload ends here)
        const unsigned long long span = wend - wbase;
        const uint32_t tlast = span >= SP_WIN_BITS ? (uint32_t)W - 1u : (uint32_t)((span + SP_SUB_BITS - 1) / SP_SUB_BITS) - 1u;
        const uint32_t xl = L.exits[tlast];
        if (preroll) {
            entry_x = xl;  // first landing at or past the region start
            preroll = false;
            if (Bv0 >= end_bits) break;  // (cannot happen: r < nreg)
        } else {
            if (!beyond) {
                mycount += nsym;
                mybad |= bad;
            }
            if (lastwin) {
                exit_x = xl;
                break;
            }
        }
        g0 = xl;
        wbase += SP_WIN_BITS;
        __syncthreads();  // exits / head0 are rewritten by the next window
    }

    // ---- report ----
    const uint32_t wsumv = wave_reduce_add_u32(mycount);
    if ((tid & 63) == 0) L.wsum[tid >> 6] = wsumv;
    if (__builtin_amdgcn_ballot_w64(mybad) != 0ull && (tid & 63) == 0) L.anybad = 1;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < W / 64; w++) tot += L.wsum[w];
        *o_entry = entry_x;
        *o_count = tot;
        *o_exit = (gaveup || L.anybad) ? 0xFFFFFFFFu : exit_x;
    }
}

// One workgroup per block: prove the chain of entries, scan the counts, publish the class.
__global__ __launch_bounds__(256) void k4_split_scan(const unsigned long long* __restrict__ d_comp_off,
                                                     const uint32_t* __restrict__ d_comp_size,
                                                     const uint32_t* __restrict__ d_orig_size, const uint8_t* __restrict__ comp,
                                                     uint8_t* __restrict__ d_cls, int32_t* __restrict__ d_status,
                                                     long long* __restrict__ d_errpos, const SplitDesc* __restrict__ sdp) {
    __shared__ uint32_t fail;
    __shared__ unsigned long long carry;
    __shared__ unsigned long long wsum[4];
    const uint32_t b = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const SplitDesc sd = *sdp;
    if (tid == 0) sd.nreg[b] = 0;
    if (d_cls[b] != 0) return;
    const uint32_t csize = d_comp_size[b], orig = d_orig_size[b];
    const uint32_t skew = (uint32_t)(((uintptr_t)comp + (uintptr_t)d_comp_off[b]) & 15u);
    const unsigned long long S = sd.region_bytes;
    const unsigned long long nreg = ((unsigned long long)skew + csize + S - 1) / S;
    if (nreg < 2 || nreg > sd.rmax || orig == 0) return;
    uint32_t* const entry = sd.entry + (uint64_t)b * sd.rmax;
    uint32_t* const count = sd.count + (uint64_t)b * sd.rmax;
    uint32_t* const exitx = sd.exit + (uint64_t)b * sd.rmax;
    uint32_t* const off = sd.off + (uint64_t)b * sd.rmax;
    if (tid == 0) {
        fail = 0;
        carry = 0;
    }
    __syncthreads();
    for (uint32_t r = (uint32_t)tid; r < nreg; r += 256u) {
        bool f = exitx[r] == 0xFFFFFFFFu;
        if (r > 0 && exitx[r - 1] != entry[r]) f = true;  // the guessed entry is not where the proven parse arrives
        if (f) fail = 1;
    }
    __syncthreads();
    if (fail) return;  // class stays 0: the sequential kernels decode the block (and report what is wrong with it)
    for (uint32_t r0 = 0; r0 < nreg; r0 += 256u) {
        const uint32_t r = r0 + (uint32_t)tid;
        const unsigned long long v = (r < nreg) ? count[r] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if ((tid & 63) >= o) inc += t;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        unsigned long long wb = 0, tot = 0;
        for (int w = 0; w < 4; w++) {
            if (w < (tid >> 6)) wb += wsum[w];
            tot += wsum[w];
        }
        const unsigned long long o = carry + wb + inc - v;  // symbols before region r
        if (r < nreg) {
            uint32_t c = 0;
            if (o < orig) {
                const unsigned long long room = orig - o;
                // the last region produces everything that is left (the reference reads zero bits past the payload)
                c = (r + 1 == nreg) ? (uint32_t)room : (uint32_t)(v < room ? v : room);
            }
            off[r] = (uint32_t)(o < orig ? o : orig);
            count[r] = c;
        }
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        sd.nreg[b] = (uint32_t)nreg;
        d_cls[b] = 2;  // decoded per region; the per-block launches skip it
        d_status[b] = DCZ_OK;
        if (d_errpos) d_errpos[b] = 0;
    }
}

// Region size for this call: the payload bytes of the blocks that can be split / SPLIT_REGIONS, at least 64 KiB, whole
// windows; written with the table pointers to the device copy of the descriptor every consumer reads.
__global__ void k4_split_setup(SplitDesc* dst, SplitDesc v, const uint8_t* __restrict__ comp,
                               const unsigned long long* __restrict__ d_comp_off, const uint32_t* __restrict__ d_comp_size,
                               const uint8_t* __restrict__ d_cls, uint32_t K) {
    unsigned long long total = 0;
    for (uint32_t b = 0; b < K; b++)
        if (d_cls[b] == 0) total += d_comp_size[b];
    unsigned long long S = (total + SPLIT_REGIONS - 1) / SPLIT_REGIONS;
    if (S < 65536ull) S = 65536ull;
    v.region_bytes = S = (S + 8191ull) & ~8191ull;
    // workgroups of the region grids: the regions of every block that can be split (what k4_split_count computes)
    uint32_t acc = 0;
    for (uint32_t b = 0; b < K; b++) {
        v.rbase[b] = acc;
        if (d_cls[b] != 0) continue;
        const unsigned long long skew = ((uintptr_t)comp + (uintptr_t)d_comp_off[b]) & 15u;
        const unsigned long long nreg = (skew + d_comp_size[b] + S - 1) / S;
        if (nreg >= 2 && nreg <= v.rmax) acc += (uint32_t)nreg;
    }
    v.rbase[K] = acc;
    v.nblk = K;
    *dst = v;
}

void launch_split_count(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                        const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, uint8_t* d_cls, int32_t* d_status,
                        int64_t* d_errpos, const SplitDesc& sd, SplitDesc* d_sd, hipStream_t s) {
    hipLaunchKernelGGL(k4_split_setup, dim3(1), dim3(1), 0, s, d_sd, sd, d_comp,
                       reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_cls, K);
    hipLaunchKernelGGL(k4_split_count, dim3(SPLIT_GRID), dim3(SP_W), 0, s, d_comp,
                       reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_orig_size, d_len, d_cls, d_sd);
#if DCZ_K4_MEDIUM_DFA
    launch_count_dfa(d_comp, d_comp_off, d_comp_size, d_orig_size, d_len, d_cls, d_sd, s);
#endif
    hipLaunchKernelGGL(k4_split_scan, dim3(K), dim3(256), 0, s, reinterpret_cast<const unsigned long long*>(d_comp_off),
                       d_comp_size, d_orig_size, d_comp, d_cls, d_status, reinterpret_cast<long long*>(d_errpos), d_sd);
}

}  // namespace dcz
