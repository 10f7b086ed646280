// k1_histogram.hip -- K1: 256-bin byte histogram per 32 KiB segment (gfx950).
//
// Replaces FrequencyService.computeHistogram (service/FrequencyService.java:16): the CPU version
// service/cpu/CpuFrequencyService.java:37-46 and the TornadoVM tile kernel service/gpu/TornadoKernels.java:89-100
// (one work-item per 64 KiB tile; 256 work-items for a 16 MiB chunk).
//
// Design (HBM-bound, integer): one WAVE owns one segment and streams it with 16 B/lane coalesced loads.
// Counting uses LDS atomics into a wave-private histogram replicated K1_COPIES times across the banks:
//   dword[(bin & 127) * COPIES + (lane % COPIES)], low half = bin < 128, high half = bin >= 128.
// With 32 replicas every ds_add_u32 is conflict-free for ANY data; with 16 (the default) lanes l and l+16 share
// a replica, so the worst case -- every byte equal, BASELINE config 5 -- is a 2-way same-address serialisation,
// while the halved LDS footprint doubles the resident waves.  Measured on MI355X (8 GiB): 32 replicas 2.12 ms
// random / 2.16 ms 99%-zeros; 16 replicas 1.55 / 1.80 ms; 8 replicas 1.64 / 3.40 ms.
// A (lane group, bin) count is at most a few thousand per segment, a bin total at most 32768: u16 halves never
// carry into each other, and the replicas can be summed as packed dwords.
// No global atomics, no inter-workgroup traffic: K2 sums the segment rows of its block.
#include "dcz_internal.h"

namespace dcz {

#ifndef DCZ_K1_COPIES
#define DCZ_K1_COPIES 16
#endif
#ifndef DCZ_K1_NT
#define DCZ_K1_NT 1  // read-once stream: non-temporal loads.  With the next step's loads issued ahead: 0.79 -> 0.72 ms per
                     // 4 GiB (6.0 TB/s); without that, and for 4 instead of 6 vector instructions per byte, no difference
#endif
#ifndef DCZ_K1_COPY_NT
#define DCZ_K1_COPY_NT 1
#endif
#ifndef DCZ_K1_COPY_LATE
#define DCZ_K1_COPY_LATE 1
#endif
#ifndef DCZ_K1_WAVES
#define DCZ_K1_WAVES 4
#endif
#ifndef DCZ_K1_RUNS
#define DCZ_K1_RUNS 1  // steps whose 16-byte units are mostly runs of one byte add 16 at a time (hist_add_vec_runs)
#endif
constexpr int K1_WAVES = DCZ_K1_WAVES;    // waves (= segments) per workgroup
constexpr int K1_COPIES = DCZ_K1_COPIES;  // histogram replicas per wave: 32 = one per bank (conflict-free for any data)
constexpr int K1_CSHIFT = (K1_COPIES == 32) ? 5 : (K1_COPIES == 16) ? 4 : 3;
constexpr int K1_DW = 128 * K1_COPIES;    // dwords per wave-private histogram

__device__ __forceinline__ void hist_add(uint32_t* h, uint32_t col, uint32_t byte) {
    const uint32_t idx = ((byte & 127u) << K1_CSHIFT) + col;
    const uint32_t val = (byte >> 7) * 0xFFFFu + 1u;  // 1 or 0x10000
    __hip_atomic_fetch_add(&h[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Four bytes of a dword, 4 vector instructions + 1 LDS atomic each (the kernel is vector-issue bound: 6 per byte measured
// 0.80 ms per 4 GiB, 4 per byte see DESIGN.md): bit field extract of the low 7 bits, shift-add onto the lane's column
// address; bit field extract of bit 7, multiply-add to 1 or 0x10000.
template <int K>
__device__ __forceinline__ void hist_add_byte(uint32_t hcol_addr, uint32_t d) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    // (inline assembly: the optimiser rewrites the C forms of these into shift + and + add and and + compare + select)
    uint32_t t, addr, b, val;
    asm("v_bfe_u32 %0, %1, %2, 7" : "=v"(t) : "v"(d), "n"(8 * K));
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(addr) : "v"(t), "n"(K1_CSHIFT + 2), "v"(hcol_addr));
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(b) : "v"(d), "n"(8 * K + 7));
    asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(val) : "v"(b), "s"(0xFFFFu));
    __hip_atomic_fetch_add((lds_u32*)(uintptr_t)addr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ void hist_add_dword(uint32_t* h, uint32_t col, uint32_t d) {
    const uint32_t hcol_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)(h + col));
    hist_add_byte<0>(hcol_addr, d);
    hist_add_byte<1>(hcol_addr, d);
    hist_add_byte<2>(hcol_addr, d);
    hist_add_byte<3>(hcol_addr, d);
}

__device__ __forceinline__ void hist_add_vec(uint32_t* h, uint32_t col, const uint4& v) {
    hist_add_dword(h, col, v.x);
    hist_add_dword(h, col, v.y);
    hist_add_dword(h, col, v.z);
    hist_add_dword(h, col, v.w);
}

// Runs (zero pages, BASELINE config 5): a 16-byte unit whose bytes are all equal is ONE add of 16 instead of 16 adds of 1
// that all go to the same address -- the lanes of a replica serialise on it (0.88 against 0.72 ms per 4 GiB on zeros + 1 %
// noise, where 85 % of the units are runs).  The other units of the step take the 16 adds as always.
__device__ __forceinline__ bool unit_is_run(const uint4& v) {
    return v.x == v.y && v.x == v.z && v.x == v.w && v.x == __builtin_amdgcn_alignbit(v.x, v.x, 8);
}
__device__ __forceinline__ void hist_add_vec_runs(uint32_t* h, uint32_t col, const uint4& v) {
    if (unit_is_run(v)) {
        const uint32_t byte = v.x & 0xFFu;
        const uint32_t idx = ((byte & 127u) << K1_CSHIFT) + col;
        const uint32_t val = ((byte >> 7) * 0xFFFFu + 1u) << 4;  // 16 or 16 << 16
        __hip_atomic_fetch_add(&h[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
        hist_add_vec(h, col, v);
    }
}

// COPY: the segment is also stored, byte for byte, at the same offset of copy_out.  That is the whole encoder output when
// every block of the call turns out to have the identity code (256 symbols of 8 bits: the reference's high-entropy case,
// payload = input, block b at offset b * block_bytes); the caller asks for it when the calls before were like that
// (ShapeHint, dcz_internal.h) and K3 then finds its blocks already in place -- one read and one write of the input
// instead of two reads and one write.  If a block is not of that kind after all, K3 writes it as always.
template <bool COPY>
__global__ __launch_bounds__(K1_WAVES * 64) void k1_histogram(const uint8_t* __restrict__ in, size_t n,
                                                               size_t block_bytes, uint32_t spb, uint64_t nseg,
                                                               uint16_t* __restrict__ seg_hist,
                                                               uint8_t* __restrict__ copy_out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[K1_WAVES][K1_DW];
    const int w = (int)(threadIdx.x >> 6);
    const int lane = lane_id();
    const uint64_t seg = (uint64_t)blockIdx.x * K1_WAVES + (uint64_t)w;
    if (seg >= nseg) return;  // wave-uniform; this kernel has no workgroup barrier
    uint32_t* h = lds[w];
    const uint32_t col = (uint32_t)lane & (uint32_t)(K1_COPIES - 1);

    {
        uint4* h4 = reinterpret_cast<uint4*>(h);
        const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < K1_DW / 4 / 64; i++) h4[i * 64 + lane] = z;
    }
    wave_lds_fence();

    const uint64_t b = seg / spb, j = seg % spb;
    const uint64_t bstart = b * (uint64_t)block_bytes;
    const uint64_t bend = (bstart + block_bytes < n) ? bstart + block_bytes : (uint64_t)n;
    const uint64_t s0 = bstart + j * (uint64_t)SEG;
    uint32_t len = 0;
    if (s0 < bend) len = (uint32_t)((bend - s0 < SEG) ? bend - s0 : SEG);
    const uint8_t* p = in + s0;

    // unaligned head (only when block_bytes or the base pointer is not a multiple of 16)
    uint32_t nhead = (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u);
    if (nhead > len) nhead = len;
    if ((uint32_t)lane < nhead) {
        const uint8_t v = p[lane];
        hist_add(h, col, v);
        if constexpr (COPY) copy_out[s0 + lane] = v;
    }
    const uint4* pv = reinterpret_cast<const uint4*>(p + nhead);
    // (COPY: the caller guarantees copy_out - in is a multiple of 16, so these units are aligned in both buffers)
    u32x4* const cv = COPY ? reinterpret_cast<u32x4*>(copy_out + s0 + nhead) : nullptr;
    auto store4 = [&](uint32_t base, const uint4& d0, const uint4& d1, const uint4& d2, const uint4& d3) {
        if constexpr (COPY) {
            const uint32_t i0 = base + (uint32_t)lane;
            const u32x4 a = {d0.x, d0.y, d0.z, d0.w}, b2 = {d1.x, d1.y, d1.z, d1.w};
            const u32x4 c = {d2.x, d2.y, d2.z, d2.w}, e = {d3.x, d3.y, d3.z, d3.w};
#if DCZ_K1_COPY_NT
            __builtin_nontemporal_store(a, cv + i0);
            __builtin_nontemporal_store(b2, cv + i0 + 64);
            __builtin_nontemporal_store(c, cv + i0 + 128);
            __builtin_nontemporal_store(e, cv + i0 + 192);
#else
            cv[i0] = a;
            cv[i0 + 64] = b2;
            cv[i0 + 128] = c;
            cv[i0 + 192] = e;
#endif
        }
    };
    const uint32_t nvec = (len - nhead) >> 4;

    // body: 4 x 16 B per lane per step, the next step's loads issued before this step's atomics (a wave alternates between
    // waiting for HBM and feeding the LDS otherwise, and only five waves per SIMD are there to cover for it)
    auto load4 = [&](uint32_t base, uint4& d0, uint4& d1, uint4& d2, uint4& d3) {
        const uint32_t i0 = base + (uint32_t)lane;
#if DCZ_K1_NT
        const u32x4* pn = reinterpret_cast<const u32x4*>(pv);
        const u32x4 n0 = __builtin_nontemporal_load(pn + i0), n1 = __builtin_nontemporal_load(pn + i0 + 64);
        const u32x4 n2 = __builtin_nontemporal_load(pn + i0 + 128), n3 = __builtin_nontemporal_load(pn + i0 + 192);
        d0 = make_uint4(n0.x, n0.y, n0.z, n0.w);
        d1 = make_uint4(n1.x, n1.y, n1.z, n1.w);
        d2 = make_uint4(n2.x, n2.y, n2.z, n2.w);
        d3 = make_uint4(n3.x, n3.y, n3.z, n3.w);
#else
        d0 = pv[i0];
        d1 = pv[i0 + 64];
        d2 = pv[i0 + 128];
        d3 = pv[i0 + 192];
#endif
    };
    const uint32_t nfull = nvec & ~255u;  // whole steps
    uint4 c0, c1, c2, c3;
    if (nfull) load4(0, c0, c1, c2, c3);
    for (uint32_t base = 0; base < nfull; base += 256) {
        uint4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
        if (base + 256 < nfull) load4(base + 256, n0, n1, n2, n3);
#if !DCZ_K1_COPY_LATE
        store4(base, c0, c1, c2, c3);
#endif
        // (wave-uniform: a step whose first units are mostly runs takes the path that looks for them)
        if (DCZ_K1_RUNS && __builtin_popcountll(__builtin_amdgcn_ballot_w64(unit_is_run(c0))) >= 32) {
            hist_add_vec_runs(h, col, c0);
            hist_add_vec_runs(h, col, c1);
            hist_add_vec_runs(h, col, c2);
            hist_add_vec_runs(h, col, c3);
        } else {
            hist_add_vec(h, col, c0);
            hist_add_vec(h, col, c1);
            hist_add_vec(h, col, c2);
            hist_add_vec(h, col, c3);
        }
#if DCZ_K1_COPY_LATE
        store4(base, c0, c1, c2, c3);
#endif
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
    for (uint32_t i = nfull + (uint32_t)lane; i < nvec; i += 64) {  // the ragged last step
        const uint4 v = pv[i];
        hist_add_vec(h, col, v);
        if constexpr (COPY) {
            const u32x4 a = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(a, cv + i);
        }
    }
    // tail (< 16 bytes)
    {
        const uint32_t done = nhead + (nvec << 4);
        const uint32_t ntail = len - done;
        if ((uint32_t)lane < ntail) {
            const uint8_t v = p[done + lane];
            hist_add(h, col, v);
            if constexpr (COPY) copy_out[s0 + done + lane] = v;
        }
    }
    wave_lds_fence();

    // Sum the replicas. Lane l owns rows l and l+64 (bins l, l+128 and l+64, l+192).  The 16-byte chunk order is
    // rotated by lane so that the lanes of a ds_read_b128 group spread over the banks.
    uint32_t acc0 = 0, acc1 = 0;
    const uint4* r0 = reinterpret_cast<const uint4*>(h + (uint32_t)lane * (uint32_t)K1_COPIES);
    const uint4* r1 = reinterpret_cast<const uint4*>(h + ((uint32_t)lane + 64u) * (uint32_t)K1_COPIES);
    constexpr int NQ = K1_COPIES / 4;
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        const int q = (t + (lane >> 1)) & (NQ - 1);
        const uint4 a = r0[q];
        const uint4 c = r1[q];
        acc0 += a.x + a.y + a.z + a.w;
        acc1 += c.x + c.y + c.z + c.w;
    }
    uint16_t* row = seg_hist + seg * 256u;
    row[lane] = (uint16_t)(acc0 & 0xFFFFu);
    row[lane + 128] = (uint16_t)(acc0 >> 16);
    row[lane + 64] = (uint16_t)(acc1 & 0xFFFFu);
    row[lane + 192] = (uint16_t)(acc1 >> 16);
}

// Single-window histogram for the host API: sum segment rows into 256 x i64 (d_hist pre-zeroed).
__global__ __launch_bounds__(256) void k1_sum_rows(const uint16_t* __restrict__ seg_hist, uint64_t nseg,
                                                   unsigned long long* __restrict__ d_hist) {
    const uint64_t per = (nseg + gridDim.x - 1) / gridDim.x;
    const uint64_t a = (uint64_t)blockIdx.x * per;
    uint64_t e = a + per;
    if (e > nseg) e = nseg;
    unsigned long long acc = 0;
    for (uint64_t s = a; s < e; s++) acc += seg_hist[s * 256u + threadIdx.x];
    if (acc) atomicAdd(&d_hist[threadIdx.x], acc);
}

void launch_histogram(const uint8_t* d_in, size_t n, size_t block_bytes, uint32_t segs_per_block, uint64_t nseg,
                      uint16_t* seg_hist, hipStream_t s, uint8_t* copy_out) {
    if (nseg == 0) return;
    const uint32_t grid = (uint32_t)((nseg + K1_WAVES - 1) / K1_WAVES);
    if (copy_out)
        hipLaunchKernelGGL(k1_histogram<true>, dim3(grid), dim3(K1_WAVES * 64), 0, s, d_in, n, block_bytes, segs_per_block,
                           nseg, seg_hist, copy_out);
    else
        hipLaunchKernelGGL(k1_histogram<false>, dim3(grid), dim3(K1_WAVES * 64), 0, s, d_in, n, block_bytes, segs_per_block,
                           nseg, seg_hist, (uint8_t*)nullptr);
}

void launch_sum_hist(const uint16_t* seg_hist, uint64_t nseg, int64_t* d_hist, hipStream_t s) {
    (void)hipMemsetAsync(d_hist, 0, 256 * sizeof(int64_t), s);
    if (nseg == 0) return;
    uint32_t grid = (uint32_t)((nseg + 63) / 64);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k1_sum_rows, dim3(grid), dim3(256), 0, s, seg_hist, nseg,
                       reinterpret_cast<unsigned long long*>(d_hist));
}

}  // namespace dcz
