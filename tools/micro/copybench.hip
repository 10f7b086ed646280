// copybench.hip -- what a plain device-to-device copy reaches on this GPU for several launch shapes (the ceiling of
// k4_fixed L=8 and of K3's identity path).  hipcc --offload-arch=gfx950 -O3 -o copybench copybench.hip && ./copybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t per = 256 * (size_t)U;
    const size_t ntiles = nvec / per;
    for (size_t t = blockIdx.x; t < ntiles; t += PERSIST ? gridDim.x : ntiles) {
        const size_t base = t * per + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&src[base + 256 * k]) : src[base + 256 * k];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) __builtin_nontemporal_store(v[k], &dst[base + 256 * k]);
            else dst[base + 256 * k] = v[k];
        }
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void readk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t per = 256 * (size_t)U;
    const size_t ntiles = nvec / per;
    u32x4 acc = (u32x4)(0u);
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t base = t * per + threadIdx.x;
#pragma unroll
        for (int k = 0; k < U; k++) {
            const u32x4 v = NT ? __builtin_nontemporal_load(&src[base + 256 * k]) : src[base + 256 * k];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[threadIdx.x] = acc;
}

// what an all-blocks flat grid costs when no block is of the class: every workgroup reads one class byte and leaves
__global__ __launch_bounds__(256) void emptyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t tiles_per_block) {
    const uint8_t* cls = reinterpret_cast<const uint8_t*>(src);
    if (cls[blockIdx.x / tiles_per_block] != 0x55) return;
    dst[threadIdx.x] = src[threadIdx.x];
}

// persistent, software pipelined: the loads of tile i+1 are issued before the stores of tile i
template <int U, bool NT>
__global__ __launch_bounds__(256) void copyp(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t per = 256 * (size_t)U;
    const size_t ntiles = nvec / per;
    size_t t = blockIdx.x;
    if (t >= ntiles) return;
    u32x4 a[U], b[U];
#pragma unroll
    for (int k = 0; k < U; k++) a[k] = NT ? __builtin_nontemporal_load(&src[t * per + threadIdx.x + 256 * k]) : src[t * per + threadIdx.x + 256 * k];
    while (true) {
        const size_t tn = t + gridDim.x;
        const bool more = tn < ntiles;
        if (more) {
#pragma unroll
            for (int k = 0; k < U; k++) b[k] = NT ? __builtin_nontemporal_load(&src[tn * per + threadIdx.x + 256 * k]) : src[tn * per + threadIdx.x + 256 * k];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) __builtin_nontemporal_store(a[k], &dst[t * per + threadIdx.x + 256 * k]);
            else dst[t * per + threadIdx.x + 256 * k] = a[k];
        }
        if (!more) break;
#pragma unroll
        for (int k = 0; k < U; k++) a[k] = b[k];
        t = tn;
    }
}

#define RUN(name, kern, grid, bytes_moved)                                                   \
    do {                                                                                     \
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, s, d, nvec); \
        hipEventRecord(e0);                                                                  \
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, s, d, nvec); \
        hipEventRecord(e1);                                                                  \
        hipEventSynchronize(e1);                                                             \
        float ms;                                                                            \
        hipEventElapsedTime(&ms, e0, e1);                                                    \
        printf("%-34s grid %8zu  %7.3f ms  %7.1f GB/s\n", name, (size_t)(grid), ms / 5, (bytes_moved) / (ms / 5 * 1e-3) / 1e9); \
    } while (0)

int main() {
    const size_t n = (size_t)8 << 30;
    u32x4 *s, *d;
    hipMalloc(&s, n);
    hipMalloc(&d, n);
    hipMemset(s, 1, n);
    hipMemset(d, 2, n);
    const size_t nvec = n / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    {
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.3f ms  %7.1f GB/s\n", "hipMemcpyAsync D2D", ms / 5, 2.0 * n / (ms / 5 * 1e-3) / 1e9);
    }
    RUN("copy U1 flat", (copyk<1, false, false>), nvec / 256, 2.0 * n);
    RUN("copy U4 flat", (copyk<4, false, false>), nvec / 1024, 2.0 * n);
    RUN("copy U8 flat", (copyk<8, false, false>), nvec / 2048, 2.0 * n);
    RUN("copy U4 flat nt", (copyk<4, true, false>), nvec / 1024, 2.0 * n);
    RUN("copy U8 flat nt", (copyk<8, true, false>), nvec / 2048, 2.0 * n);
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "copy U4 persist g%d", g);
        RUN(nm, (copyk<4, false, true>), g, 2.0 * n);
        snprintf(nm, 64, "copy U8 persist g%d", g);
        RUN(nm, (copyk<8, false, true>), g, 2.0 * n);
        snprintf(nm, 64, "copy U4 persist nt g%d", g);
        RUN(nm, (copyk<4, true, true>), g, 2.0 * n);
        snprintf(nm, 64, "copy U8 persist nt g%d", g);
        RUN(nm, (copyk<8, true, true>), g, 2.0 * n);
    }
    RUN("empty flat 4KiB tiles (2M wgs)", emptyk, nvec / 256, 0.0);
    {
        size_t nvec_save = nvec;
        const size_t tpb = 256;  // 1 MiB blocks of 4 KiB tiles
        auto nvec = tpb;         // third kernel argument of emptyk
        RUN("empty flat 2M wgs", emptyk, nvec_save / 256, 0.0);
        RUN("empty flat 512K wgs", emptyk, nvec_save / 1024, 0.0);
        RUN("empty flat 128K wgs", emptyk, nvec_save / 4096, 0.0);
    }
    for (int g : {2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "copy U16 persist g%d", g);
        RUN(nm, (copyk<16, false, true>), g, 2.0 * n);
        snprintf(nm, 64, "copy U16 persist nt g%d", g);
        RUN(nm, (copyk<16, true, true>), g, 2.0 * n);
        snprintf(nm, 64, "copy U4 pipelined g%d", g);
        RUN(nm, (copyp<4, false>), g, 2.0 * n);
        snprintf(nm, 64, "copy U8 pipelined g%d", g);
        RUN(nm, (copyp<8, false>), g, 2.0 * n);
        snprintf(nm, 64, "copy U8 pipelined nt g%d", g);
        RUN(nm, (copyp<8, true>), g, 2.0 * n);
    }
    RUN("copy U1 flat nt", (copyk<1, true, false>), nvec / 256, 2.0 * n);
    RUN("copy U2 flat", (copyk<2, false, false>), nvec / 512, 2.0 * n);
    RUN("copy U2 flat nt", (copyk<2, true, false>), nvec / 512, 2.0 * n);
    for (int g : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "read U4 g%d", g);
        RUN(nm, (readk<4, false>), g, 1.0 * n);
        snprintf(nm, 64, "read U8 g%d", g);
        RUN(nm, (readk<8, false>), g, 1.0 * n);
        snprintf(nm, 64, "read U8 nt g%d", g);
        RUN(nm, (readk<8, true>), g, 1.0 * n);
    }
    return 0;
}
