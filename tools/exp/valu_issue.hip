// valu_issue.hip -- what one wave per SIMD pays per instruction for the triple scan's two instructions, by the registers of their
// operands (diagnostic: is it the register banks that hold k_epi_triples1 at 43 % of the issue peak?).
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP9(X) X X X X X X X X X
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40"

template <int MODE> __global__ __launch_bounds__(256) void k(long long *out, int iters) {
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)        // bitop3, three operands in one bank (4, 8, 12), nine destinations
            asm volatile(REP9("v_bitop3_b32 v20, v4, v8, v12 bitop3:0x80\n v_bitop3_b32 v21, v4, v8, v12 bitop3:0x80\n v_bitop3_b32 v22, v4, v8, v12 bitop3:0x80\n") ::: CLOB);
        else if (MODE == 1)   // bitop3, operands in three banks (4, 9, 14)
            asm volatile(REP9("v_bitop3_b32 v20, v4, v9, v14 bitop3:0x80\n v_bitop3_b32 v21, v4, v9, v14 bitop3:0x80\n v_bitop3_b32 v22, v4, v9, v14 bitop3:0x80\n") ::: CLOB);
        else if (MODE == 2)   // bitop3, two operands in one bank (4, 8, 13)
            asm volatile(REP9("v_bitop3_b32 v20, v4, v8, v13 bitop3:0x80\n v_bitop3_b32 v21, v4, v8, v13 bitop3:0x80\n v_bitop3_b32 v22, v4, v8, v13 bitop3:0x80\n") ::: CLOB);
        else if (MODE == 3)   // bcnt accumulate, independent chains
            asm volatile(REP9("v_bcnt_u32_b32 v20, v4, v20\n v_bcnt_u32_b32 v21, v5, v21\n v_bcnt_u32_b32 v22, v6, v22\n") ::: CLOB);
        else if (MODE == 4)   // v_and_b32 (two operands)
            asm volatile(REP9("v_and_b32 v20, v4, v8\n v_and_b32 v21, v4, v9\n v_and_b32 v22, v4, v10\n") ::: CLOB);
        else if (MODE == 5)   // the scan's own mix, operands as the compiler placed them: 9 bitop3 (one bank) then 9 bcnt
            asm volatile(REP9("v_bitop3_b32 v20, v4, v8, v12 bitop3:0x80\n v_bitop3_b32 v21, v4, v16, v12 bitop3:0x80\n v_bitop3_b32 v22, v4, v24, v12 bitop3:0x80\n")
                         REP9("v_bcnt_u32_b32 v30, v20, v30\n v_bcnt_u32_b32 v31, v21, v31\n v_bcnt_u32_b32 v32, v22, v32\n") ::: CLOB);
        else if (MODE == 6)   // the same mix, operands in three banks
            asm volatile(REP9("v_bitop3_b32 v20, v4, v9, v14 bitop3:0x80\n v_bitop3_b32 v21, v4, v17, v14 bitop3:0x80\n v_bitop3_b32 v22, v4, v25, v14 bitop3:0x80\n")
                         REP9("v_bcnt_u32_b32 v30, v20, v30\n v_bcnt_u32_b32 v31, v21, v31\n v_bcnt_u32_b32 v32, v22, v32\n") ::: CLOB);
        else if (MODE == 7)   // v_and_b32 with an SGPR operand + bcnt (the pair's product kept scalar)
            asm volatile(REP9("v_and_b32 v20, s20, v12\n v_and_b32 v21, s21, v12\n v_and_b32 v22, s22, v12\n")
                         REP9("v_bcnt_u32_b32 v30, v20, v30\n v_bcnt_u32_b32 v31, v21, v31\n v_bcnt_u32_b32 v32, v22, v32\n") ::: CLOB, "s20", "s21", "s22");
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
}

template <int MODE> void run(const char *what, int n_instr) {
    long long *d; hipMalloc(&d, 16);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, d, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, d, iters);
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-70s %6.2f shader cycles per instruction (clock64), %7.3f ns wall per instruction\n", what, (double)h[0] / ((double)iters * n_instr),
           (double)h[1] * 10.0 / ((double)iters * n_instr));      // wall_clock64: 100 MHz
    hipFree(d);
}

int main() {
    run<0>("bitop3, three operands in one bank", 27);
    run<1>("bitop3, operands in three banks", 27);
    run<2>("bitop3, two operands in one bank", 27);
    run<3>("bcnt accumulate", 27);
    run<4>("v_and_b32, two operands", 27);
    run<5>("scan mix 27 bitop3 (one bank) + 27 bcnt", 54);
    run<6>("scan mix 27 bitop3 (three banks) + 27 bcnt", 54);
    run<7>("27 v_and_b32 (SGPR, VGPR) + 27 bcnt", 54);
    return 0;
}
