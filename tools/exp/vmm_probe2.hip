// hipMemSetAccess refusals on some hosts: which way of backing a reservation piece by piece do they accept?  API calls only
// (nothing is written to a range whose set-access failed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const size_t M = (size_t)1 << 20;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static const char *E(hipError_t e) { (void)hipGetLastError(); return e == hipSuccess ? "ok" : hipGetErrorString(e); }
// pieces of `piece` MB one after the other; mode 0: set-access per piece, 1: set-access over everything mapped so far,
// 2: map all pieces first, one set-access at the end
static void run(const char *name, size_t align, size_t piece, int n, int mode) {
    void *base = nullptr;
    hipError_t e = hipMemAddressReserve(&base, (size_t)4 << 30, align, nullptr, 0);
    printf("%s: reserve(align %zu MB) %s base %p\n", name, align / M, E(e), base);
    if (e != hipSuccess) return;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    int mapped = 0;
    for (int k = 0; k < n; k++) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, piece * M, &prop, 0);
        if (e != hipSuccess) { printf("  piece %d create %s\n", k, E(e)); break; }
        hs.push_back(h);
        e = hipMemMap((char *)base + k * piece * M, piece * M, 0, h, 0);
        if (e != hipSuccess) { printf("  piece %d map %s\n", k, E(e)); break; }
        mapped++;
        if (mode == 0) e = hipMemSetAccess((char *)base + k * piece * M, piece * M, &acc, 1);
        else if (mode == 1) e = hipMemSetAccess(base, (k + 1) * piece * M, &acc, 1);
        else e = hipSuccess;
        printf("  piece %d at +%zu MB: set-access %s\n", k, k * piece, mode == 2 ? "(later)" : E(e));
        if (e != hipSuccess) break;
    }
    if (mode == 2) { e = hipMemSetAccess(base, mapped * piece * M, &acc, 1); printf("  one set-access over %zu MB: %s\n", mapped * piece, E(e)); }
    for (int k = 0; k < mapped; k++) (void)hipMemUnmap((char *)base + k * piece * M, piece * M);
    for (auto h : hs) (void)hipMemRelease(h);
    (void)hipMemAddressFree(base, (size_t)4 << 30);
    (void)hipGetLastError();
}
int main() {
    (void)hipSetDevice(0);
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    run("A per piece, 128 MB pieces, align 2 MB", 2 * M, 128, 6, 0);
    run("B cumulative set-access", 2 * M, 128, 6, 1);
    run("C map all, then one set-access", 2 * M, 128, 6, 2);
    run("D per piece, align 1 GB", 1024 * M, 128, 6, 0);
    run("E per piece, 1 GB pieces, align 1 GB", 1024 * M, 1024, 3, 0);
    run("F per piece, 64 MB pieces", 2 * M, 64, 8, 0);
    run("G per piece, align 0", 0, 128, 6, 0);
    return 0;
}
