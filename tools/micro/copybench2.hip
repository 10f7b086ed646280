// copybench2.hip -- follow-up to copybench.hip: one 16-byte access per lane (the fastest flat shape there), with workgroup
// sizes from 256 to 1024 threads and with/without non-temporal accesses; a 16 KiB tile is then one 1024-thread workgroup.
// hipcc --offload-arch=gfx950 -O3 -o copybench2 copybench2.hip && ./copybench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int T, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * (T * U) + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = NTL ? __builtin_nontemporal_load(&src[base + T * k]) : src[base + T * k];
#pragma unroll
    for (int k = 0; k < U; k++) {
        if (NTS) __builtin_nontemporal_store(v[k], &dst[base + T * k]);
        else dst[base + T * k] = v[k];
    }
}

template <int T, int U, bool NTL, bool NTS>
static void run(const char* name, const u32x4* s, u32x4* d, size_t nvec) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t grid = nvec / (T * U);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((copyk<T, U, NTL, NTS>), dim3(grid), dim3(T), 0, 0, s, d, nvec);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((copyk<T, U, NTL, NTS>), dim3(grid), dim3(T), 0, 0, s, d, nvec);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-36s grid %8zu  %7.3f ms  %7.1f GB/s\n", name, grid, ms / 5, 2.0 * nvec * 16 / (ms / 5 * 1e-3) / 1e9);
}

int main() {
    const size_t n = (size_t)8 << 30;
    u32x4 *s, *d;
    hipMalloc(&s, n);
    hipMalloc(&d, n);
    hipMemset(s, 1, n);
    hipMemset(d, 2, n);
    const size_t nvec = n / 16;
    run<256, 1, false, false>("T256 U1", s, d, nvec);
    run<256, 1, true, true>("T256 U1 nt", s, d, nvec);
    run<256, 1, true, false>("T256 U1 nt-load", s, d, nvec);
    run<256, 1, false, true>("T256 U1 nt-store", s, d, nvec);
    run<256, 2, false, false>("T256 U2", s, d, nvec);
    run<256, 2, true, true>("T256 U2 nt", s, d, nvec);
    run<256, 4, true, true>("T256 U4 nt (k4_fixed today)", s, d, nvec);
    run<512, 1, false, false>("T512 U1", s, d, nvec);
    run<512, 1, true, true>("T512 U1 nt", s, d, nvec);
    run<1024, 1, false, false>("T1024 U1", s, d, nvec);
    run<1024, 1, true, true>("T1024 U1 nt", s, d, nvec);
    run<1024, 1, true, false>("T1024 U1 nt-load", s, d, nvec);
    run<512, 2, true, true>("T512 U2 nt", s, d, nvec);
    run<64, 1, false, false>("T64 U1", s, d, nvec);
    run<128, 1, false, false>("T128 U1", s, d, nvec);
    return 0;
}
