// Cycles per MFMA, issued back to back on independent accumulators by one wave per SIMD (diagnostic):
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_rate.hip -o tools/exp/mfma_rate && gpurun -- ./tools/exp/mfma_rate
// (the binary is not kept).  Measured: f32_16x16x32_bf16 31 ticks, i32_16x16x64_i8 45, scale_f32_16x16x128_f8f6f4 (FP4) 33
// per instruction back to back; vector fillers of the same wave add to that rather than hide under it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
template <int KIND, int FILL>
__global__ void __launch_bounds__(256, 1) k(int iters, unsigned long long *out, int seed) {
    v4i a4 = {seed, seed * 3, seed * 5, seed * 7}, b4 = {seed * 11, seed * 13, seed * 17, seed * 19};
    v8i a8 = {seed, seed * 3, seed * 5, seed * 7, 0, 0, 0, 0}, b8 = {seed * 11, seed * 13, seed * 17, seed * 19, 0, 0, 0, 0};
    v8s as = {1, 2, 3, 4, 5, 6, 7, 8}, bs = {1, 2, 3, 4, 5, 6, 7, 8};
    v4i ci[9]; v4f cf[9];
    for (int c = 0; c < 9; c++) { ci[c] = v4i{0, 0, 0, 0}; cf[c] = v4f{0, 0, 0, 0}; }
    int filler = seed, fl[8] = {seed, seed + 1, seed + 2, seed + 3, seed + 4, seed + 5, seed + 6, seed + 7};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int c = 0; c < 9; c++) {
            if constexpr (KIND == 0) ci[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, ci[c], 0, 0, 0);
            if constexpr (KIND == 1) cf[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, cf[c], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if constexpr (KIND == 2) cf[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as, bs, cf[c], 0, 0, 0);
            #pragma unroll
            for (int f = 0; f < FILL; f++) fl[f] = fl[f] & (0x11111111 + it);     /* independent of one another */
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int f = 0; f < 8; f++) filler += fl[f];
    int s = filler; float sf = 0;
    for (int c = 0; c < 9; c++) { s += ci[c].x + ci[c].y + ci[c].z + ci[c].w; sf += cf[c].x + cf[c].y + cf[c].z + cf[c].w; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s + (unsigned long long)sf; }
}
template <int KIND, int FILL>
static void run(const char *name, unsigned long long *d) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(256), 0, 0, iters, d, 3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(256), 0, 0, iters, d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("{\"mfma\": \"%s\", \"fillers_per_mfma\": %d, \"ns_per_mfma\": %.2f, \"memtime_ticks_per_mfma\": %.2f}\n", name, FILL, ms * 1e6 / (iters * 9.0), (double)h[0] / (iters * 9.0));
}
int main() {
    unsigned long long *d; hipMalloc(&d, 64);
    run<2, 0>("f32_16x16x32_bf16", d); run<0, 0>("i32_16x16x64_i8", d); run<1, 0>("scale_f32_16x16x128_f8f6f4 (fp4)", d);
    run<2, 3>("f32_16x16x32_bf16", d); run<0, 3>("i32_16x16x64_i8", d); run<1, 3>("scale_f32_16x16x128_f8f6f4 (fp4)", d);
    run<1, 6>("scale_f32_16x16x128_f8f6f4 (fp4)", d); run<1, 8>("scale_f32_16x16x128_f8f6f4 (fp4)", d);
    return 0;
}
