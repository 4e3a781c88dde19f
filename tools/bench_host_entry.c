/* Latency of the synchronous per-batch entry points of include/hpgv.h (host genotype batch in, host results out) --
 * what an assoc_test / tdt_test adapter pays per call (assoc_runner.c:192-195, tdt_runner.c:184-185).  Diagnostic: this
 * PCIe-inclusive rate is never bench.py's `value`.
 *
 *   gcc -O2 -std=gnu99 -Iinclude tools/bench_host_entry.c -Lhpg-variant_amd/lib -lhpgv -Wl,-rpath,$PWD/hpg-variant_amd/lib -lm -o /tmp/bench_host_entry
 *   /tmp/bench_host_entry [samples]
 *
 * For every batch size the call is timed with the genotypes in page-locked memory from hpgv_host_alloc (what the host
 * adapters stage into: the fused kernel then reads them in place) and in ordinary pageable memory (malloc), with the
 * fused kernel (default) and with the option batch_fused = 0 (copy + layout + scan + statistics kernels). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hpgv.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int cmp(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hpgv_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000;
    hpgv_ctx *ctx = NULL;
    if (hpgv_create(0, &ctx)) { fprintf(stderr, "hpgv_create: %s\n", hpgv_last_error(NULL)); return 1; }
    uint8_t *cond = malloc((size_t)N);
    for (int j = 0; j < N; j++) cond[j] = (uint8_t)(j & 1);
    CHK(hpgv_set_cohort(ctx, cond, N));
    /* trios (3k, 3k+1, 3k+2) for the tdt call */
    const int nt = N / 3;
    int32_t *fc = malloc(sizeof(int32_t) * (size_t)(nt + 1)), *mc = malloc(sizeof(int32_t) * (size_t)(nt + 1)),
            *co = malloc(sizeof(int32_t) * (size_t)(nt + 2)), *cc = malloc(sizeof(int32_t) * (size_t)(nt + 1));
    uint8_t *cs = malloc((size_t)nt + 1);
    for (int k = 0; k < nt; k++) { fc[k] = 3 * k; mc[k] = 3 * k + 1; cc[k] = 3 * k + 2; co[k] = k; cs[k] = (uint8_t)(k & 1); }
    co[nt] = nt;
    CHK(hpgv_set_families(ctx, N, nt, fc, mc, co, cc, cs));
    CHK(hpgv_set_stats_cohort(ctx, N));
    const int batches[] = {200, 2000, 20000};
    static const uint8_t codes[4] = {0x00, 0x01, 0x11, 0xFF};
    for (unsigned bi = 0; bi < sizeof batches / sizeof batches[0]; bi++) {
        const int B = batches[bi];
        const size_t bytes = (size_t)B * (size_t)N;
        uint8_t *pinned = NULL, *pageable = malloc(bytes + (size_t)B);
        CHK(hpgv_host_alloc(ctx, bytes + (size_t)B, (void **)&pinned));
        unsigned long long st = 88172645463325252ULL;
        for (size_t i = 0; i < bytes; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; pageable[i] = codes[(st >> 20) & 3]; }
        memset(pageable + bytes, 0, (size_t)B);
        memcpy(pinned, pageable, bytes + (size_t)B);
        int32_t *ints = malloc(sizeof(int32_t) * 8 * (size_t)B);
        double *dbl = malloc(sizeof(double) * 3 * (size_t)B);
        for (int fused = 2; fused >= 0; fused--) {                   /* 2: the fused kernel after a copy by the copy engine (option batch_copy) */
            CHK(hpgv_set_option(ctx, "batch_fused", fused ? 1 : 0));
            CHK(hpgv_set_option(ctx, "batch_copy", fused == 2));
            for (int where = 0; where < 2; where++) {
                const uint8_t *gt = where == 0 ? pinned : pageable;
                for (int tool = 0; tool < 3; tool++) {
                    const int reps = B <= 200 ? 400 : (B <= 2000 ? 60 : 8);
                    double *t = malloc(sizeof(double) * (size_t)reps);
                    for (int r = -3; r < reps; r++) {
                        const double t0 = now();
                        if (tool == 0) CHK(hpgv_assoc(ctx, HPGV_TASK_CHISQ, gt, (size_t)N, B, gt + bytes, ints, ints + B, ints + 2 * B, ints + 3 * B, dbl, dbl + B, dbl + 2 * B));
                        if (tool == 1) CHK(hpgv_tdt(ctx, gt, (size_t)N, B, gt + bytes, ints, ints + B, dbl, dbl + B, dbl + 2 * B));
                        if (tool == 2) CHK(hpgv_stats(ctx, gt, (size_t)N, B, ints, dbl, dbl + B));
                        if (r >= 0) t[r] = now() - t0;
                    }
                    qsort(t, (size_t)reps, sizeof(double), cmp);
                    double mean = 0; for (int r = 0; r < reps; r++) mean += t[r]; mean /= reps;
                    printf("{\"call\": \"%s\", \"batch_variants\": %d, \"samples\": %d, \"genotypes_in\": \"%s\", \"fused_kernel\": %d, "
                           "\"us_per_call_median\": %.1f, \"us_per_call_mean\": %.1f, \"us_per_call_min\": %.1f, \"variants_per_s\": %.0f, \"host_to_device_GBps\": %.2f}\n",
                           tool == 0 ? "hpgv_assoc(chisq)" : tool == 1 ? "hpgv_tdt" : "hpgv_stats", B, N, where == 0 ? "page-locked" : "pageable", fused,
                           t[reps / 2] * 1e6, mean * 1e6, t[0] * 1e6, B / t[reps / 2], bytes / t[reps / 2] / 1e9);
                    free(t);
                }
            }
        }
        CHK(hpgv_set_option(ctx, "batch_fused", 1));
        CHK(hpgv_set_option(ctx, "batch_copy", 0));
        free(ints); free(dbl); free(pageable);
        CHK(hpgv_host_free(ctx, pinned));
    }
    hpgv_destroy(ctx);
    return 0;
}
