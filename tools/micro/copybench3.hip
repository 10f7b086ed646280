// copybench3.hip -- does the relative placement of source and destination matter for a flat copy?  One allocation, the
// destination at 8 GiB + delta from the source, the shape k4_fixed / k3_copy_identity use (256 threads x 4 x 16 B, nt).
// hipcc --offload-arch=gfx950 -O3 -o copybench3 copybench3.hip && ./copybench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load(&src[base + 256 * k]);
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(v[k], &dst[base + 256 * k]);
}
int main() {
    const size_t n = (size_t)8 << 30;
    char* base;
    if (hipMalloc(&base, 2 * n + (80u << 20)) != hipSuccess) return 1;
    (void)hipMemset(base, 1, 2 * n + (80u << 20));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    size_t deltas[128];
    int nd = 0;
    for (size_t d = 0; d < (64u << 10); d += 4096) deltas[nd++] = d;          // 4 KiB steps
    for (size_t d = (64u << 10); d <= (2u << 20); d += (64u << 10)) deltas[nd++] = d;  // 64 KiB steps
    for (size_t d = (4u << 20); d <= (64u << 20); d += (4u << 20)) deltas[nd++] = d;   // 4 MiB steps
    for (int rep = 0; rep < 1; rep++)
        for (int di = 0; di < nd; di++) {
            const size_t d = deltas[di];
            const u32x4* s = reinterpret_cast<const u32x4*>(base);
            u32x4* t = reinterpret_cast<u32x4*>(base + n + d);
            const size_t grid = n / 16 / 1024;
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, 0, s, t);
            (void)hipEventRecord(e0);
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copyk, dim3(grid), dim3(256), 0, 0, s, t);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("delta %10zu  %7.3f ms  %7.1f GB/s\n", d, ms / 5, 2.0 * n / (ms / 5 * 1e-3) / 1e9);
        }
    return 0;
}
