// k4_fixed.hip -- K4, block classification and the analytic decoder for fixed-length complete codes (gfx950).
//
// Replaces TableBasedHuffmanDecoder.decode (core/TableBasedHuffmanDecoder.java:103-134) for blocks whose stored table
// gives every present symbol the same length L and has exactly 2^L symbols (L <= 8).  That is what the reference's
// encoder produces for high-entropy input -- its own headline case (app/logs/datacomp.log:3254, ratio 99.98 %): 256
// symbols of 8 bits, codeword(s) = s, payload == input -- and for any source of 2^L equiprobable symbols.  In such a
// stream the i-th codeword starts at bit i*L: there is nothing to synchronise and no dependent chain.  Symbol i is
// symtab[bits(i*L, L)] with symtab = the present symbols in ascending order (canonical code, CanonicalHuffman.java:123-129),
// which is exactly what the reference's 10-bit table returns for a complete code of length <= 10
// (TableBasedHuffmanDecoder.java:78-88); bits past the payload read as zero (:204-208), so a short payload decodes to
// symtab[0] like the reference; a complete code has no invalid pattern, so there is no error path.
//
// k4_classify runs first on every decode call (one thread per block): validates the untrusted footer fields against the
// buffers (DCZ_E_INVALID, no kernel touches such a block) and flags fixed-length blocks (class byte 0x10 | L); the
// table-walk kernels of k4_decode.hip skip every block whose class byte is not 0.
// k4_fixed is a FLAT grid of one workgroup per (block, 8 KiB output tile) over all blocks; a workgroup whose block is not
// of the class leaves after one byte load.  Measured on MI355X (tools/micro/copybench.hip, 8 GiB): a flat grid of small
// tiles copies at 6.1-6.6 TB/s with non-temporal accesses, persistent grid-stride loops reach 4.5-5.5 TB/s (inside one
// wave a load's data waits for every older store: vmcnt counts them in order), and 512 K workgroups that leave at once
// cost 0.11 ms -- so the flat grid wins whether or not the class is present.  A block of any size is spread over the
// whole chip, the single 16-32 MiB chunk of the reference's own API included.  L == 8 is a copy (16 B/lane, HBM-bound:
// C + N bytes); L < 8 stages the tile's payload in LDS and extracts 16 fields per lane.
#include "dcz_internal.h"

namespace dcz {

// Work item shape.  One 16-byte access per lane copies fastest (tools/micro/copybench2.hip, 8 GiB, flat grids, non-temporal:
// 256 threads x 4 per lane 6.06 TB/s, 1024 x 1 6.39, 512 x 1 6.53, 256 x 1 with 4 KiB tiles 6.60), but takes four times the
// waves, and the waves of a launch that finds no fixed-length block cost what they cost (0.1-0.15 ms per 4 GiB on text
// and low-entropy input) -- which the launch-shape hint now avoids: such calls get the small persistent grid.  In the
// library, three runs each: 512 threads x 1 with 8 KiB tiles K4 2.91-2.92 ms per 8 GiB, 1024 x 1 with 16 KiB 2.94-2.96,
// 256 x 4 with 16 KiB 3.02-3.03.
#ifndef DCZ_FX_TILE
#define DCZ_FX_TILE 8192
#endif
constexpr uint32_t FX_TILE = DCZ_FX_TILE;  // output bytes per work item
#ifndef DCZ_FX_T
#define DCZ_FX_T 512
#endif
constexpr int FX_T = DCZ_FX_T;
constexpr int FX_UPT = (int)(FX_TILE / 16u / (uint32_t)FX_T);  // 16-byte units per thread
static_assert(FX_T >= 256 && FX_UPT * FX_T * 16 == (int)FX_TILE, "tile = threads x units x 16 bytes");

__global__ __launch_bounds__(256) void k4_classify(const uint8_t* __restrict__ d_len,
                                                   const unsigned long long* __restrict__ d_comp_off,
                                                   const uint32_t* __restrict__ d_comp_size,
                                                   const uint32_t* __restrict__ d_orig_size, unsigned long long comp_bytes,
                                                   unsigned long long out_stride, uint32_t K, uint8_t* __restrict__ cls,
                                                   int32_t* __restrict__ d_status, long long* __restrict__ d_errpos,
                                                   uint32_t* __restrict__ hint, uint32_t* __restrict__ hint_done,
                                                   uint32_t epoch) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= K) return;
    if (b == 0u && hint_done) *hint_done = epoch;  // (ShapeHint: this call has been classified)
    const uint4* row = reinterpret_cast<const uint4*>(d_len + (uint64_t)b * 256u);
    uint32_t n = 0, mn = 255u, mx = 0;
#pragma unroll 4
    for (int q = 0; q < 16; q++) {
        const uint4 v = row[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t l = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            n += l ? 1u : 0u;
            mn = (l && l < mn) ? l : mn;
            mx = l > mx ? l : mx;
        }
    }
    const unsigned long long coff = d_comp_off[b];
    const unsigned long long csize = d_comp_size[b];
    const unsigned long long orig = d_orig_size[b];
    uint8_t c = 0;
    if (coff > comp_bytes || csize > comp_bytes - coff || orig > out_stride) {
        c = 0xFFu;  // footer fields point outside the buffers: nothing reads or writes for this block
        d_status[b] = DCZ_E_INVALID;
        if (d_errpos) d_errpos[b] = 0;
    } else if (orig > 0 && n >= 2u && mn == mx && mx <= 8u && n == (1u << mx)) {
        c = (uint8_t)(0x10u | mx);
        d_status[b] = DCZ_OK;
        if (d_errpos) d_errpos[b] = 0;
    }
    cls[b] = c;
    if ((c & 0xF0u) == 0x10u && hint) *hint = epoch;  // (ShapeHint: k4_fixed has work in calls like this one)
}

// 16 bytes at virtual byte offset vb0 (any alignment) of the payload whose 16-byte aligned base is vbase; bytes outside
// [vlo, vhi) read as zero.  q/r: dword and byte part of vb0 & 15 (q is wave-uniform here: the skew of the block).
template <int Q>
__device__ __forceinline__ uint4 shift_units(const uint4& a, const uint4& b, uint32_t r) {
    const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return make_uint4(__builtin_amdgcn_alignbyte(d[Q + 1], d[Q], r), __builtin_amdgcn_alignbyte(d[Q + 2], d[Q + 1], r),
                      __builtin_amdgcn_alignbyte(d[Q + 3], d[Q + 2], r), __builtin_amdgcn_alignbyte(d[Q + 4], d[Q + 3], r));
}

__device__ __forceinline__ void store_unit(uint8_t* dst, const uint4& v, uint32_t nvalid, bool aligned) {
    if (aligned && nvalid >= 16u) {
        *reinterpret_cast<uint4*>(dst) = v;
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((uint32_t)i < nvalid) dst[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    }
}

// 16 symbols of L bits from the big-endian dwords bw[0..3] (field j at bit j*L), mapped through the LDS symbol table.
template <int L>
__device__ __forceinline__ uint4 extract16(const uint32_t (&bw)[5], const uint8_t* symtab) {
    uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        constexpr uint32_t mask = (1u << L) - 1u;
        const int bit = j * L, idx = bit >> 5, s = bit & 31;
        uint32_t v;
        if (s + L <= 32) v = (bw[idx] >> (32 - s - L)) & mask;
        else v = (uint32_t)((((unsigned long long)bw[idx] << 32) | bw[idx + 1]) >> (64 - s - L)) & mask;
        o[j >> 2] |= (uint32_t)symtab[v] << (8 * (j & 3));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

struct FxLds {
    __attribute__((aligned(16))) uint32_t stage[(FX_TILE / 8 * 7 + 64) / 4];  // payload of one tile, L <= 7
    uint8_t symtab[256];
    uint32_t wcnt[FX_T / 64];
};

// one work item: tile `tile` of block b
__device__ __forceinline__ void fixed_item(FxLds& S, uint32_t b, uint32_t tile, const uint8_t* __restrict__ comp,
                                           const unsigned long long* __restrict__ d_comp_off,
                                           const uint32_t* __restrict__ d_comp_size,
                                           const uint32_t* __restrict__ d_orig_size, const uint8_t* __restrict__ d_len,
                                           unsigned long long out_stride, uint8_t* __restrict__ out,
                                           const uint8_t* __restrict__ cls) {
    uint32_t* const stage = S.stage;
    uint8_t* const symtab = S.symtab;
    uint32_t* const wcnt = S.wcnt;
    const uint32_t c = cls[b];
    if ((c & 0xF0u) != 0x10u) return;  // workgroup-uniform: not a fixed-length block
    const uint32_t L = c & 15u;
    const int tid = (int)threadIdx.x;
    const uint32_t orig = d_orig_size[b];
    const unsigned long long t0 = (unsigned long long)tile * FX_TILE;
    if (t0 >= orig) return;
    const uint32_t nout = (orig - t0 < FX_TILE) ? (uint32_t)(orig - t0) : FX_TILE;
    const uintptr_t pay = (uintptr_t)comp + (uintptr_t)d_comp_off[b];
    const uint32_t skew = (uint32_t)(pay & 15u);
    const uint8_t* const vbase = reinterpret_cast<const uint8_t*>(pay - skew);
    const unsigned long long vlo = skew, vhi = (unsigned long long)skew + d_comp_size[b];
    uint8_t* const dst0 = out + (unsigned long long)b * out_stride + t0;
    const bool dst_al = (((uintptr_t)dst0) & 15u) == 0u;
    if (L == 8u) {
        // codeword(s) = s (256 symbols of 8 bits, ascending): the payload is the output
        if (skew == 0u && dst_al && nout == FX_TILE && t0 + FX_TILE <= vhi) {  // interior tile: a plain streaming copy
            const u32x4* s4 = reinterpret_cast<const u32x4*>(vbase + t0) + tid;
            u32x4* d4 = reinterpret_cast<u32x4*>(dst0) + tid;
            u32x4 v[FX_UPT];
#pragma unroll
            for (int k = 0; k < FX_UPT; k++) v[k] = __builtin_nontemporal_load(s4 + FX_T * k);
#pragma unroll
            for (int k = 0; k < FX_UPT; k++) __builtin_nontemporal_store(v[k], d4 + FX_T * k);
            return;
        }
        uint4 v[FX_UPT];
        const uint32_t q = skew >> 2, r = skew & 3u;
#pragma unroll
        for (int k = 0; k < FX_UPT; k++) {
            const uint32_t o = ((uint32_t)tid + (uint32_t)FX_T * k) * 16u;
            v[k] = make_uint4(0, 0, 0, 0);
            if (o < nout) {
                const uint4 a = load_chunk16(vbase, t0 + o, vlo, vhi);  // aligned chunk holding the unit's first byte
                if (skew == 0u) {
                    v[k] = a;
                } else {
                    const uint4 e = load_chunk16(vbase, t0 + o + 16u, vlo, vhi);
                    v[k] = (q == 0u) ? shift_units<0>(a, e, r) : (q == 1u) ? shift_units<1>(a, e, r)
                           : (q == 2u) ? shift_units<2>(a, e, r) : shift_units<3>(a, e, r);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < FX_UPT; k++) {
            const uint32_t o = ((uint32_t)tid + (uint32_t)FX_T * k) * 16u;
            if (o < nout) store_unit(dst0 + o, v[k], nout - o, dst_al);
        }
        return;
    }
    // ---- L < 8: the present symbols in ascending order ----
    {
        const bool present = tid < 256 && d_len[(uint64_t)b * 256u + (tid & 255)] != 0;  // (thread = symbol)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(present);
        if (tid < 256 && (tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t rank = (uint32_t)__builtin_popcountll(m & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < ((tid & 255) >> 6); w++) rank += wcnt[w];
        if (present) symtab[rank] = (uint8_t)tid;
    }
    // payload bytes of this tile: virtual [skew + t0 * L / 8, + nout * L / 8 rounded up); t0 * L / 8 is a multiple of 16
    const unsigned long long vin = (t0 >> 3) * L;      // relative to the payload start (multiple of 2048 * L)
    const uint32_t nin = (nout * L + 7u) >> 3;         // payload bytes holding the tile's symbols
    const uint32_t nchunks = (skew + nin + 15u) >> 4;  // aligned chunks from virtual byte vin
    for (uint32_t ch = (uint32_t)tid; ch < nchunks; ch += FX_T)
        reinterpret_cast<uint4*>(stage)[ch] = load_chunk16(vbase, vin + 16ull * ch, vlo, vhi);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < FX_UPT; k++) {
        const uint32_t u = (uint32_t)tid + (uint32_t)FX_T * k;  // unit of 16 symbols inside the tile
        const uint32_t o = u * 16u;
        if (o >= nout) break;
        const uint32_t off = skew + 2u * L * u;  // byte of the unit's first bit inside the stage
        const uint32_t r = off & 3u;
        const uint32_t* p = stage + (off >> 2);
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
        uint32_t bw[5];
        bw[0] = bswap32(__builtin_amdgcn_alignbyte(d1, d0, r));
        bw[1] = bswap32(__builtin_amdgcn_alignbyte(d2, d1, r));
        bw[2] = bswap32(__builtin_amdgcn_alignbyte(d3, d2, r));
        bw[3] = bswap32(__builtin_amdgcn_alignbyte(d4, d3, r));
        bw[4] = 0;
        uint4 v;
        switch (L) {  // workgroup-uniform
            case 1: v = extract16<1>(bw, symtab); break;
            case 2: v = extract16<2>(bw, symtab); break;
            case 3: v = extract16<3>(bw, symtab); break;
            case 4: v = extract16<4>(bw, symtab); break;
            case 5: v = extract16<5>(bw, symtab); break;
            case 6: v = extract16<6>(bw, symtab); break;
            default: v = extract16<7>(bw, symtab); break;
        }
        store_unit(dst0 + o, v, nout - o, dst_al);
    }
}

// PERSIST = false: flat grid, one workgroup per work item (what copies fastest).  PERSIST = true: a small grid that finds
// the fixed-length blocks itself -- for calls in which none is expected (ShapeHint, dcz_internal.h).  When the
// expectation was wrong the persistent shape must still use the whole chip: FX_T blocks at a time, every thread looks at
// one class byte, the flagged blocks are compacted (in block order, so every workgroup builds the same list) and the
// (flagged block, tile) pairs of the range are dealt round-robin over the grid; `rot` carries the deal across ranges so
// that remainders do not pile up on the first workgroups.
template <bool PERSIST>
__global__ __launch_bounds__(FX_T) void k4_fixed(const uint8_t* __restrict__ comp,
                                                 const unsigned long long* __restrict__ d_comp_off,
                                                 const uint32_t* __restrict__ d_comp_size,
                                                 const uint32_t* __restrict__ d_orig_size,
                                                 const uint8_t* __restrict__ d_len, unsigned long long out_stride,
                                                 uint8_t* __restrict__ out, const uint8_t* __restrict__ cls,
                                                 uint32_t tiles_per_block, uint32_t wg0, uint32_t nblk) {
    __shared__ FxLds S;
    if constexpr (!PERSIST) {
        const uint32_t wi = wg0 + blockIdx.x;
        const uint32_t b = wi / tiles_per_block;
        fixed_item(S, b, wi - b * tiles_per_block, comp, d_comp_off, d_comp_size, d_orig_size, d_len, out_stride, out, cls);
    } else {
        __shared__ uint16_t flagged[FX_T];
        __shared__ uint32_t fcnt[FX_T / 64];
        const uint32_t tid = threadIdx.x, w = tid >> 6;
        uint32_t rot = 0;  // items dealt so far, mod gridDim.x
        for (uint32_t c0 = 0; c0 < nblk; c0 += (uint32_t)FX_T) {
            const uint32_t bi = c0 + tid;
            const bool mine = bi < nblk && (cls[bi] & 0xF0u) == 0x10u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if ((tid & 63u) == 0u) fcnt[w] = (uint32_t)__builtin_popcountll(m);
            __syncthreads();
            uint32_t base = 0, nflag = 0;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)(FX_T / 64); i++) {
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
                fixed_item(S, c0 + flagged[f], (uint32_t)(it - (unsigned long long)f * tiles_per_block), comp, d_comp_off,
                           d_comp_size, d_orig_size, d_len, out_stride, out, cls);
                __syncthreads();  // (the staging area and the symbol table are reused)
            }
            rot = (uint32_t)((rot + items) % gridDim.x);
            __syncthreads();  // (flagged / fcnt are rewritten by the next range)
        }
    }
}

void launch_classify(const uint8_t* d_len, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                     const uint32_t* d_orig_size, size_t comp_bytes, size_t out_stride, uint32_t K, const DecodeWs& ws,
                     int32_t* d_status, int64_t* d_errpos, hipStream_t s) {
    hipLaunchKernelGGL(k4_classify, dim3((K + 255) / 256), dim3(256), 0, s, d_len,
                       reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_orig_size,
                       (unsigned long long)comp_bytes, (unsigned long long)out_stride, K, ws.cls, d_status,
                       reinterpret_cast<long long*>(d_errpos), ws.fixed.dev, ws.fixed.done, ws.fixed.epoch);
}

void launch_decode_fixed(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_size,
                         const uint32_t* d_orig_size, const uint8_t* d_len, uint32_t K, size_t out_stride, uint8_t* d_out,
                         const DecodeWs& ws, hipStream_t s) {
    const unsigned long long tpb = ((unsigned long long)out_stride + FX_TILE - 1) / FX_TILE;
    if (tpb == 0 || tpb > 0x7FFFFFFFull) return;
    if (!ws.fixed.likely) {  // no fixed-length block in the last calls: a small grid that finds out for itself
        hipLaunchKernelGGL(k4_fixed<true>, dim3(HINT_PERSIST_GRID), dim3(FX_T), 0, s, d_comp,
                           reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_orig_size, d_len,
                           (unsigned long long)out_stride, d_out, ws.cls, (uint32_t)tpb, 0u, K);
        return;
    }
    const unsigned long long items = (unsigned long long)K * tpb;  // < 2^62; launched in slices of < 2^31 workgroups
    const unsigned long long slice = (0x40000000ull / tpb) * tpb ? (0x40000000ull / tpb) * tpb : tpb;
    for (unsigned long long w0 = 0; w0 < items; w0 += slice) {
        const unsigned long long cnt = (items - w0 < slice) ? items - w0 : slice;
        if (w0 > 0xFFFFFFFFull) break;  // (K < 2^31 blocks of >= 1 tile: unreachable for buffers that exist)
        hipLaunchKernelGGL(k4_fixed<false>, dim3((uint32_t)cnt), dim3(FX_T), 0, s, d_comp,
                           reinterpret_cast<const unsigned long long*>(d_comp_off), d_comp_size, d_orig_size, d_len,
                           (unsigned long long)out_stride, d_out, ws.cls, (uint32_t)tpb, (uint32_t)w0, K);
    }
}

}  // namespace dcz
