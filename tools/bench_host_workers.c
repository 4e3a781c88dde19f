/* The per-batch call from concurrent workers: the reference's runner calls assoc_test from num_threads OpenMP workers at once
 * (assoc_runner.c:106-207; hpg-variant.conf: num-threads = 4), each on a batch of its own.  T threads call hpgv_assoc on
 * batches of B variants x N samples in page-locked memory for a fixed time; aggregate calls/s and host-to-device GB/s against
 * the bus (57 GB/s).  On one context and on a group of two contexts on device 0.  Diagnostic (PCIe-inclusive: never bench.py's value).
 *
 *   bench_host_workers [samples] [batch_variants] [seconds]
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hpgv.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { hpgv_ctx *ctx; int N, B; uint8_t *gt; double until; long calls; double busy; int rc; } worker_t;

static void *work(void *v) {
    worker_t *w = (worker_t *)v;
    int32_t *ints = malloc(sizeof(int32_t) * 4 * (size_t)w->B);
    double *dbl = malloc(sizeof(double) * 3 * (size_t)w->B);
    const size_t bytes = (size_t)w->B * (size_t)w->N;
    while (now() < w->until) {
        const double t0 = now();
        const int rc = hpgv_assoc(w->ctx, HPGV_TASK_CHISQ, w->gt, (size_t)w->N, w->B, w->gt + bytes, ints, ints + w->B, ints + 2 * w->B, ints + 3 * w->B,
                                  dbl, dbl + w->B, dbl + 2 * w->B);
        if (rc) { w->rc = rc; break; }
        w->busy += now() - t0;
        w->calls++;
    }
    free(ints); free(dbl);
    return NULL;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000, B = argc > 2 ? atoi(argv[2]) : 200;
    const double secs = argc > 3 ? atof(argv[3]) : 0.5;
    for (int group = 0; group < 2; group++) {
        hpgv_ctx *ctx = NULL;
        const int ids[2] = {0, 0};
        if (group ? hpgv_create_multi(ids, 2, &ctx) : hpgv_create(0, &ctx)) { fprintf(stderr, "create: %s\n", hpgv_last_error(NULL)); return 1; }
        uint8_t *cond = malloc((size_t)N);
        for (int j = 0; j < N; j++) cond[j] = (uint8_t)(j & 1);
        if (hpgv_set_cohort(ctx, cond, N)) { fprintf(stderr, "%s\n", hpgv_last_error(ctx)); return 1; }
        static const uint8_t codes[4] = {0x00, 0x01, 0x11, 0xFF};
        const size_t bytes = (size_t)B * (size_t)N;
        for (int T = 1; T <= 8; T *= 2) {
            worker_t w[8];
            pthread_t th[8];
            memset(w, 0, sizeof w);
            for (int k = 0; k < T; k++) {
                w[k].ctx = ctx; w[k].N = N; w[k].B = B;
                if (hpgv_host_alloc(ctx, bytes + (size_t)B, (void **)&w[k].gt)) { fprintf(stderr, "%s\n", hpgv_last_error(ctx)); return 1; }
                unsigned long long st = 88172645463325252ULL + (unsigned)k;
                for (size_t i = 0; i < bytes; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; w[k].gt[i] = codes[(st >> 20) & 3]; }
                memset(w[k].gt + bytes, 0, (size_t)B);
            }
            for (int warm = 0; warm < 2; warm++) {                   /* the first round creates the slots and their streams */
                const double t0 = now();
                for (int k = 0; k < T; k++) { w[k].until = t0 + (warm ? secs : 0.05); w[k].calls = 0; w[k].busy = 0; pthread_create(&th[k], NULL, work, &w[k]); }
                for (int k = 0; k < T; k++) pthread_join(th[k], NULL);
                if (!warm) continue;
                const double dt = now() - t0;
                long calls = 0; double busy = 0;
                for (int k = 0; k < T; k++) { calls += w[k].calls; busy += w[k].busy; if (w[k].rc) { fprintf(stderr, "hpgv_assoc -> %d: %s\n", w[k].rc, hpgv_last_error(ctx)); return 1; } }
                printf("{\"call\": \"hpgv_assoc(chisq)\", \"context\": \"%s\", \"workers\": %d, \"batch_variants\": %d, \"samples\": %d, \"calls_per_s\": %.0f, "
                       "\"us_per_call_in_a_worker\": %.1f, \"variants_per_s\": %.0f, \"host_to_device_GBps\": %.2f}\n",
                       group ? "group [0,0]" : "one", T, B, N, calls / dt, busy / calls * 1e6, calls * (double)B / dt, calls * (double)bytes / dt / 1e9);
            }
            for (int k = 0; k < T; k++) hpgv_host_free(ctx, w[k].gt);
        }
        free(cond);
        hpgv_destroy(ctx);
    }
    return 0;
}
