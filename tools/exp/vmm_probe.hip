// Probe of HIP's virtual memory management on the GPU box: reserve a large address range, back it piece by piece, time the
// calls, run a kernel and copies across piece boundaries.   hipcc --offload-arch=gfx950 -O2 tools/exp/vmm_probe.hip -o /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(uint32_t *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u);
}
__global__ void sum(const uint32_t *p, size_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i] ^ (uint32_t)(i * 2654435761u);
    if (s) atomicAdd(out, s);
}
int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atol(argv[1]) : 50) << 30, piece = (size_t)(argc > 2 ? atol(argv[2]) : 1024) << 20;
    CK(hipSetDevice(0));
    int vmm = 0;
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    double t0 = now();
    void *base = nullptr;
    CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
    printf("reserve %zu GB: %.3f ms\n", total >> 30, (now() - t0) * 1e3);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const int np = 8;
    for (int k = 0; k < np; k++) {
        hipMemGenericAllocationHandle_t h;
        double a = now(); CK(hipMemCreate(&h, piece, &prop, 0));
        double b = now(); CK(hipMemMap((char *)base + k * piece, piece, 0, h, 0));
        double c = now(); CK(hipMemSetAccess((char *)base + k * piece, piece, &acc, 1));
        double d = now();
        printf("piece %d (%zu MB): create %.3f ms, map %.3f ms, access %.3f ms\n", k, piece >> 20, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3);
        hs.push_back(h);
    }
    const size_t n = np * piece / 4;
    unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    t0 = now(); fill<<<4096, 256>>>((uint32_t *)base, n); CK(hipDeviceSynchronize());
    printf("fill %zu GB across the pieces: %.3f ms\n", (np * piece) >> 30, (now() - t0) * 1e3);
    t0 = now(); sum<<<4096, 256>>>((const uint32_t *)base, n, bad); CK(hipDeviceSynchronize());
    unsigned long long hb = 1; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("check: %.3f ms, mismatch sum %llu\n", (now() - t0) * 1e3, hb);
    // a copy to the host that spans a piece boundary, and one up
    char *host; CK(hipHostMalloc(&host, 64 << 20));
    t0 = now(); CK(hipMemcpy(host, (char *)base + piece - (32 << 20), 64 << 20, hipMemcpyDeviceToHost));
    printf("D2H 64 MB across a boundary: %.3f ms, word ok %d\n", (now() - t0) * 1e3, ((uint32_t *)host)[5] == (uint32_t)(((piece - (32 << 20)) / 4 + 5) * 2654435761u));
    t0 = now(); CK(hipMemcpy((char *)base + 3 * piece - (32 << 20), host, 64 << 20, hipMemcpyHostToDevice));
    printf("H2D 64 MB across a boundary: %.3f ms\n", (now() - t0) * 1e3);
    // mapping more while a kernel runs on the pieces already there
    fill<<<4096, 256>>>((uint32_t *)base, n);
    {
        hipMemGenericAllocationHandle_t h;
        double a = now(); CK(hipMemCreate(&h, piece, &prop, 0)); CK(hipMemMap((char *)base + np * piece, piece, 0, h, 0)); CK(hipMemSetAccess((char *)base + np * piece, piece, &acc, 1));
        printf("create+map+access beside a running kernel: %.3f ms\n", (now() - a) * 1e3);
        hs.push_back(h);
    }
    CK(hipDeviceSynchronize());
    t0 = now();
    for (size_t k = 0; k < hs.size(); k++) { CK(hipMemUnmap((char *)base + k * piece, piece)); CK(hipMemRelease(hs[k])); }
    CK(hipMemAddressFree(base, total));
    printf("unmap + release + free: %.3f ms\n", (now() - t0) * 1e3);
    // plain hipMalloc of the same amount for comparison
    void *q; t0 = now(); CK(hipMalloc(&q, (np + 1) * piece)); printf("hipMalloc %zu GB: %.3f ms\n", ((np + 1) * piece) >> 30, (now() - t0) * 1e3);
    t0 = now(); CK(hipFree(q)); printf("hipFree: %.3f ms\n", (now() - t0) * 1e3);
    return 0;
}
