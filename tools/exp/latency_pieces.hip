// experiment: where do the microseconds of a tiny synchronous kernel call go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <algorithm>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + 1e-3 * t.tv_nsec; }
__global__ void k_null(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_flag(volatile unsigned *flag, unsigned gen, unsigned *counter, unsigned n) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned prev = atomicAdd(counter, 1u);
        if (prev == n - 1) { *counter = 0; __threadfence_system(); __hip_atomic_store((unsigned *)flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
__global__ void k_read(const uint4 *src, size_t n16, unsigned *sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 q = src[i]; acc.x |= q.x; acc.y |= q.y; acc.z |= q.z; acc.w |= q.w;
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345u) *sink = 1;
}
template <typename F> static double med(int reps, F f) {
    std::vector<double> t;
    for (int i = -5; i < reps; i++) { double a = now(); f(); double b = now(); if (i >= 0) t.push_back(b - a); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned *flag; hipHostMalloc(&flag, 4096, hipHostMallocDefault); *flag = 0;
    unsigned *dflag; hipHostGetDevicePointer((void **)&dflag, flag, 0);
    unsigned *counter; hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
    unsigned *sink; hipMalloc(&sink, 4);
    printf("null kernel + hipStreamSynchronize: %.1f us\n", med(300, [&] { hipLaunchKernelGGL(k_null, dim3(200), dim3(256), 0, st, nullptr); hipStreamSynchronize(st); }));
    unsigned gen = 0;
    printf("flag kernel + host spin          : %.1f us\n", med(300, [&] { gen++; hipLaunchKernelGGL(k_flag, dim3(200), dim3(256), 0, st, dflag, gen, counter, 200u);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != gen) {} }));
    hipStreamSynchronize(st);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    printf("null kernel + event query spin    : %.1f us\n", med(300, [&] { hipLaunchKernelGGL(k_null, dim3(200), dim3(256), 0, st, nullptr); hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} }));
    printf("null kernel + hipStreamQuery spin : %.1f us\n", med(300, [&] { hipLaunchKernelGGL(k_null, dim3(200), dim3(256), 0, st, nullptr); while (hipStreamQuery(st) == hipErrorNotReady) {} }));
    hipPointerAttribute_t a;
    printf("hipPointerGetAttributes (pinned)  : %.2f us\n", med(1000, [&] { hipPointerGetAttributes(&a, flag); }));
    void *pg = malloc(4096);
    printf("hipPointerGetAttributes (pageable): %.2f us\n", med(1000, [&] { if (hipPointerGetAttributes(&a, pg) != hipSuccess) (void)hipGetLastError(); }));
    int dev; printf("hipGetDevice                      : %.2f us\n", med(1000, [&] { hipGetDevice(&dev); }));
    for (size_t mb : {2, 20}) {
        uint4 *h; hipHostMalloc(&h, mb << 20, hipHostMallocDefault);
        for (size_t i = 0; i < (mb << 20) / 16; i++) h[i] = make_uint4(1, 2, 3, 4);
        uint4 *d; hipMalloc(&d, mb << 20);
        for (int blocks : {200, 400, 800, 1600}) {
            double t = med(100, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st, h, (mb << 20) / 16, sink); hipStreamSynchronize(st); });
            printf("zero-copy read %zu MB, %4d blocks + sync: %.1f us = %.1f GB/s\n", mb, blocks, t, (mb << 20) / t / 1e3);
        }
        double t = med(100, [&] { hipMemcpyAsync(d, h, mb << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); });
        printf("hipMemcpyAsync H2D %zu MB + sync         : %.1f us = %.1f GB/s\n", mb, t, (mb << 20) / t / 1e3);
        hipHostFree(h); hipFree(d);
    }
    return 0;
}
