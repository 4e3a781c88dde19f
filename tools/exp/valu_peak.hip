// valu_peak.hip -- what the chip sustains in simple vector instructions per second, by waves per SIMD (diagnostic for the
// "issue peak" the VALU-bound kernels are priced against).  Whole chip: 256 CUs x (waves per SIMD) x 4 SIMDs.
//   hipcc --offload-arch=gfx950 -O2 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) a[q] = seed + q + threadIdx.x;
    uint32_t x = seed * 3 + threadIdx.x, y = seed * 5 + blockIdx.x, z = seed ^ 0x55555555u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (KIND == 0) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[q]) : "v"(x));
                if (KIND == 1) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(a[q]) : "v"(x), "v"(y));
                if (KIND == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[q]) : "v"(x));
                if (KIND == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[q]) : "v"(x));
                if (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[q]) : "v"(x), "v"(z));
                if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[q]) : "v"(x));
                if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[q]) : "v"(x), "v"(z));
                if (KIND == 7) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(a[q]) : "v"(x));
                if (KIND == 8) asm volatile("v_alignbyte_b32 %0, %1, %0, %2" : "+v"(a[q]) : "v"(x), "v"(z));
                if (KIND == 9) asm volatile("v_add_f64 %0, %1, %0" : "+v"(*(double *)&a[q & ~1]) : "v"(*(double *)&a[10]));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < 12; ++q) s ^= a[q];
    if (s == 0x12345678u) out[0] = s;
}

template <int KIND> void run(const char *what) {
    uint32_t *d; (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int iters = 4000, grid = 256 * wps;                  // workgroups of 256 threads: one wave per SIMD each
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 10, 1u);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, iters, 1u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)grid * 4 * iters * 48.0;       // wave instructions
        printf("%-18s %d wave(s) per SIMD: %8.1f G wave-instructions/s (%5.2f cycles per instruction and SIMD at 2.4 GHz)\n", what, wps, instr / ms / 1e6,
               2.4e9 * 1024.0 * (ms / 1e3) / instr);
        fflush(stdout);
    }
    (void)hipFree(d);
}

int main() {
    run<0>("v_bcnt_u32_b32"); run<1>("v_bitop3_b32"); run<2>("v_and_b32"); run<3>("v_add_u32"); run<4>("v_fma_f32");
    run<5>("v_mul_lo_u32"); run<6>("v_perm_b32"); run<7>("v_lshl_or_b32"); run<8>("v_alignbyte_b32"); run<9>("v_add_f64");
    return 0;
}
