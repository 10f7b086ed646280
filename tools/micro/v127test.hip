// Does a value held in v127 (the 128th register: four waves of 128 registers fill a SIMD's register file) read back right
// in a 1024-thread workgroup under LDS traffic?  Mimics the four instructions of k4_dfa's recording step in which the
// hoist-7 build keeps the count of the first nibble in v127.  Prints the number of mismatches (0 expected).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024, 1) void vtest(const uint32_t* in, unsigned long long* bad, int iters, int reg, int active_waves) {
    __shared__ uint32_t T[4096 + 34 * 1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 1024) T[i] = in[i];
    __syncthreads();
    uint32_t e = in[tid & 4095], acc = 0, slot = 4096 + 34 * tid;
    unsigned long long nbad = 0;
    if ((tid >> 6) >= active_waves) iters = 0;  // the other waves leave: an LDS read then returns within a few cycles
    for (int it = 0; it < iters; it++) {
        const uint32_t e0 = T[(e >> 6) & 4095];
        const uint32_t e1 = T[(e0 >> 6) & 4095];
        uint32_t w, want;
        want = ((e1 >> 16) << (e0 & 24u)) | (e0 >> 16);
        if (reg == 127)
            asm volatile("v_and_b32 v127, 24, %1\n v_lshrrev_b32 %0, 16, %1\n s_nop 0\n v_lshrrev_b32 v126, 16, %2\n v_lshl_or_b32 %0, v126, v127, %0"
                         : "=&v"(w) : "v"(e0), "v"(e1) : "v126", "v127");
        else
            asm volatile("v_and_b32 v119, 24, %1\n v_lshrrev_b32 %0, 16, %1\n s_nop 0\n v_lshrrev_b32 v118, 16, %2\n v_lshl_or_b32 %0, v118, v119, %0"
                         : "=&v"(w) : "v"(e0), "v"(e1) : "v118", "v119", "v127");
        nbad += (w != want);
        acc |= w;
        T[slot + (it & 31)] = acc;
        e = e1 + it;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc == 0x12345678u) bad[1] = 1;
}
int main() {
    uint32_t* h = (uint32_t*)malloc(4096 * 4);
    uint32_t s = 12345;
    for (int i = 0; i < 4096; i++) { s = s * 1664525u + 1013904223u; h[i] = s; }
    uint32_t* d; unsigned long long* b;
    hipMalloc(&d, 4096 * 4); hipMalloc(&b, 16); hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int aw : {16, 4, 2, 1})
        for (int reg : {127, 119}) {
            hipMemset(b, 0, 16);
            hipLaunchKernelGGL(vtest, dim3(256), dim3(1024), 0, 0, d, b, 400000, reg, aw);
            unsigned long long r[2];
            hipMemcpy(r, b, 16, hipMemcpyDeviceToHost);
            printf("%2d waves of 16 active, count held in v%d: %llu mismatches in %llu steps (%s)\n", aw, reg, r[0], 256ull * 64 * aw * 400000,
                   hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
