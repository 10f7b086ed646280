// The recording step of k4_dfa's hoist-7 build (the reproducer), instruction for instruction with its
// registers (v120-v127, the pair v[42:43]), in a 1024-thread workgroup of 128 registers per wave: every lane records 8 dwords
// of "payload" through a random nibble automaton in LDS, twice, and the slots are compared with the same recurrence in C++.
// Prints mismatching lanes per physical wave (0 expected everywhere).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define STEP(S0, S1)                                                                     \
    "v_lshrrev_b32 v42, " #S0 ", v126\n v_and_b32 v121, 0xffc0, v125\n v_and_or_b32 v42, v42, 60, v121\n ds_read_b32 v42, v42\n" \
    S1 "\n s_waitcnt lgkmcnt(0)\n v_and_b32 v121, 0xffc0, v42\n v_and_or_b32 v120, v120, 60, v121\n ds_read_b32 v125, v120\n"    \
    "v_and_b32 v127, 24, v42\n v_lshrrev_b32 v42, 16, v42\n s_waitcnt lgkmcnt(0)\n v_lshrrev_b32 v120, 16, v125\n"                \
    "v_lshl_or_b32 v42, v120, v127, v42\n v_lshlrev_b64 v[120:121], v123, v[42:43]\n v_or_b32 v42, v124, v120\n"                 \
    "v_and_b32 v120, 24, v125\n v_add3_u32 v120, v120, v127, v123\n v_cmp_lt_u32 vcc, 31, v120\n ds_write_b32 v122, v42\n s_nop 0\n" \
    "v_cndmask_b32 v124, v42, v121, vcc\n v_cndmask_b32_e64 v123, 0, 4, vcc\n v_add_u32 v122, v123, v122\n v_and_b32 v123, 24, v120\n"
__global__ __launch_bounds__(1024, 1) void rec(const uint32_t* tab, const uint32_t* pay, uint32_t* out, int passes) {
    __shared__ uint32_t L[4096 + 36 * 1024];  // T (256 states x 16 nibbles), then a slot of 36 dwords per lane
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 1024) L[i] = tab[i];
    for (int i = 0; i < 36; i++) L[4096 + 36 * tid + i] = 0;
    __syncthreads();
    for (int p = 0; p < passes; p++) {
        uint32_t e = 0, alo = 0, k8 = 0, ab = (4096 + 36 * tid) * 4;
        for (int d = 0; d < 8; d++) {
            const uint32_t r = pay[(blockIdx.x * 1024 + tid) * 8 + d];
            asm volatile("v_mov_b32 v43, 0\n v_mov_b32 v126, %4\n v_mov_b32 v125, %0\n v_mov_b32 v124, %1\n v_mov_b32 v123, %2\n v_mov_b32 v122, %3\n"
                         STEP(26, "v_lshrrev_b32 v120, 22, v126") STEP(18, "v_lshrrev_b32 v120, 14, v126")
                         STEP(10, "v_lshrrev_b32 v120, 6, v126") STEP(2, "v_lshlrev_b32 v120, 2, v126")
                         "v_mov_b32 %0, v125\n v_mov_b32 %1, v124\n v_mov_b32 %2, v123\n v_mov_b32 %3, v122\n"
                         : "+v"(e), "+v"(alo), "+v"(k8), "+v"(ab) : "v"(r)
                         : "v42", "v43", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "vcc", "memory");
        }
        *(volatile uint32_t*)((char*)L + ab) = alo;
        __syncthreads();
    }
    for (int i = 0; i < 36; i++) out[(blockIdx.x * 1024 + tid) * 36 + i] = L[4096 + 36 * tid + i];
}
int main() {
    const int NB = 256, NL = NB * 1024;
    uint32_t* tab = (uint32_t*)malloc(4096 * 4); uint32_t* pay = (uint32_t*)malloc((size_t)NL * 8 * 4);
    uint32_t s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int i = 0; i < 4096; i++) { uint32_t c = rnd() % 3, nx = rnd() & 255, sy = rnd() & 0xFFFF; if (c < 2) sy &= 0xFF; if (c == 0) sy = 0;
        tab[i] = (nx << 6) | c | (c << 3) | (sy << 16); }
    for (size_t i = 0; i < (size_t)NL * 8; i++) pay[i] = (rnd() << 8) ^ rnd();
    uint32_t *dt, *dp, *dout; hipMalloc(&dt, 4096 * 4); hipMalloc(&dp, (size_t)NL * 32); hipMalloc(&dout, (size_t)NL * 36 * 4);
    hipMemcpy(dt, tab, 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(dp, pay, (size_t)NL * 32, hipMemcpyHostToDevice);
    uint32_t* out = (uint32_t*)malloc((size_t)NL * 36 * 4);
    for (int passes : {1, 2, 3}) {
        long long bad[16] = {0}, tot = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(rec, dim3(NB), dim3(1024), 0, 0, dt, dp, dout, passes);
            hipMemcpy(out, dout, (size_t)NL * 36 * 4, hipMemcpyDeviceToHost);
            for (int l = 0; l < NL; l++) {
                uint32_t ref[36] = {0}, e = 0, alo = 0, k8 = 0, ab = 0;
                for (int d = 0; d < 8; d++) {
                    const uint32_t r = pay[(size_t)l * 8 + d];
                    for (int j = 0; j < 8; j += 2) {
                        const uint32_t n0 = (r >> (28 - 4 * j)) & 15, n1 = (r >> (24 - 4 * j)) & 15;
                        const uint32_t e0 = tab[((e >> 6) & 255) * 16 + n0];
                        e = tab[((e0 >> 6) & 255) * 16 + n1];
                        const uint32_t c0 = e0 & 24;
                        const unsigned long long v = (unsigned long long)(((e >> 16) << c0) | (e0 >> 16)) << k8;
                        alo |= (uint32_t)v; ref[ab] = alo; k8 += c0 + (e & 24);
                        if (k8 >= 32) { alo = (uint32_t)(v >> 32); ab++; } k8 &= 31;
                    }
                }
                ref[ab] = alo;
                bool ok = true;
                for (int i = 0; i <= (int)ab && ok; i++) ok = out[(size_t)l * 36 + i] == ref[i];
                if (!ok) { bad[(l & 1023) >> 6]++; tot++; }
            }
        }
        printf("%d pass(es): %lld bad lanes of %lld; per physical wave:", passes, tot, 20ll * NL);
        for (int w = 0; w < 16; w++) printf(" %lld", bad[w]);
        printf("\n");
    }
    return 0;
}
