/* The literal drop-in, end to end: assoc_test / tdt_test / get_variants_stats (include/hpgv_host.h) on batches of
 * vcf_record_t built the way hpg-libs' reader leaves them -- one record per variant, ONE heap string per sample
 * (assoc.c:52 reads record->samples item by item) -- called from the reference's worker shape: `workers` OpenMP threads,
 * each on batches of its own, one thread draining the shared result list like the writer section does
 * (assoc_runner.c:106-207, 301-322; hpg-variant.conf: batch-lines = 200, num-threads = 4).
 *
 * Per configuration one JSON line: microseconds per call, split by the adapter's own clocks (hpgv_host_adapter_times)
 * into staging (sample strings -> one byte per genotype), engine (cohort check + the hpgv_* call: transfer, kernel,
 * results back) and records (result structs, their strings, list inserts); variants/s of the team; staging ns/genotype.
 * "lone caller" = the same call from a thread that is NOT inside a parallel region: its staging is dealt to
 * HPGV_STAGE_THREADS workers (hpgv_host_stage_records).  `--stage-only` times the staging function alone and needs no
 * GPU.  Diagnostic (host strings + PCIe inclusive): never bench.py's value.
 *
 *   bench_adapter [--call assoc|fisher|tdt|stats] [--samples N] [--batch B] [--seconds S] [--workers 1,2,4,8]
 *                 [--batches K] [--stage-only]
 */
#define _GNU_SOURCE
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hpgv_host.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

enum { CALL_ASSOC, CALL_FISHER, CALL_TDT, CALL_STATS };

/* one batch: B records x N samples; every sample string is its own strdup, as the reader's array_list of samples */
static vcf_record_t **make_batch(int B, int N, unsigned long long seed) {
    static const char *const calls[8] = {"0/0", "0/0", "0/0", "0/1", "0/1", "1/1", "1/0", "./."};
    vcf_record_t **recs = (vcf_record_t **)malloc(sizeof *recs * (size_t)B);
    unsigned long long st = 88172645463325252ULL ^ (seed * 0x9E3779B97F4A7C15ULL);
    for (int i = 0; i < B; i++) {
        vcf_record_t *r = vcf_record_new();
        char id[32];
        const int idn = snprintf(id, sizeof id, "rs%llu", (unsigned long long)(seed * 1000003ULL + (unsigned)i));
        set_vcf_record_chromosome(strdup("1"), 1, r);
        set_vcf_record_position(1000 + i, r);
        set_vcf_record_id(strdup(id), idn, r);
        set_vcf_record_reference(strdup("A"), 1, r);
        set_vcf_record_alternate(strdup("G"), 1, r);
        set_vcf_record_format(strdup("GT"), 2, r);
        for (int j = 0; j < N; j++) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            array_list_insert(strdup(calls[(st >> 20) & 7]), r->samples);
        }
        recs[i] = r;
    }
    return recs;
}

typedef struct {
    int call, N, B;
    individual_t **samples;            /* assoc */
    double *logfact;
    family_t **families; int n_families; sample_ids_t *ids;      /* tdt */
    list_t *out;
    file_stats_t *fstats;
} job_t;

static int one_call(const job_t *J, vcf_record_t **batch) {
    switch (J->call) {
    case CALL_ASSOC: assoc_test(CHI_SQUARE, batch, J->B, J->samples, J->N, NULL, J->out); return 0;
    case CALL_FISHER: assoc_test(FISHER, batch, J->B, J->samples, J->N, J->logfact, J->out); return 0;
    case CALL_TDT: return tdt_test(batch, J->B, J->families, J->n_families, J->ids, J->out);
    default: return get_variants_stats(batch, J->B, NULL, NULL, 0, J->out, J->fstats);
    }
}

static void free_result(int call, void *p) {
    if (call == CALL_ASSOC) assoc_basic_result_free((assoc_basic_result_t *)p);
    else if (call == CALL_FISHER) assoc_fisher_result_free((assoc_fisher_result_t *)p);
    else if (call == CALL_TDT) tdt_result_free((tdt_result_t *)p);
    else variant_stats_free((variant_stats_t *)p);
}

typedef struct { list_t *l; int call; long *n; } drain_t;
static void *drain_main(void *v) {                 /* the writer section's loop (assoc_runner.c:306-321), minus the fprintf */
    drain_t *d = (drain_t *)v;
    list_item_t *it;
    while ((it = list_remove_item(d->l))) { free_result(d->call, it->data_p); list_item_free(it); (*d->n)++; }
    return NULL;
}

int main(int argc, char **argv) {
    int call = CALL_ASSOC, N = 10000, B = 200, K = 2, stage_only = 0;
    double secs = 1.0;
    const char *workers = "1,2,4,8", *call_name = "assoc";
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "--call") && a + 1 < argc) call_name = argv[++a];
        else if (!strcmp(argv[a], "--samples") && a + 1 < argc) N = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--batch") && a + 1 < argc) B = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--seconds") && a + 1 < argc) secs = atof(argv[++a]);
        else if (!strcmp(argv[a], "--workers") && a + 1 < argc) workers = argv[++a];
        else if (!strcmp(argv[a], "--batches") && a + 1 < argc) K = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--stage-only")) stage_only = 1;
        else { fprintf(stderr, "usage: bench_adapter [--call assoc|fisher|tdt|stats] [--samples N] [--batch B] [--seconds S] [--workers 1,2,4,8] [--batches K] [--stage-only]\n"); return 2; }
    }
    call = !strcmp(call_name, "fisher") ? CALL_FISHER : !strcmp(call_name, "tdt") ? CALL_TDT : !strcmp(call_name, "stats") ? CALL_STATS : CALL_ASSOC;
    if (call == CALL_TDT) N = N / 3 * 3;
    int wl[16], nw = 0, wmax = 1;
    for (const char *p = workers; *p && nw < 16;) { char *e; long v = strtol(p, &e, 10); if (e == p || v < 1) break; wl[nw++] = (int)v; if (v > wmax) wmax = (int)v; p = *e ? e + 1 : e; }
    omp_set_nested(1);
    omp_set_dynamic(0);

    /* ---- the staging function alone (no device): single thread, strings not in cache (K batches in rotation) ---- */
    {
        vcf_record_t ***bt = (vcf_record_t ***)malloc(sizeof *bt * (size_t)(K > 4 ? K : 4));
        const int kb = K > 4 ? K : 4;
        for (int k = 0; k < kb; k++) bt[k] = make_batch(B, N, 7000 + (unsigned)k);
        uint8_t *gt = (uint8_t *)malloc((size_t)B * (size_t)N + (size_t)B);
        for (int team = 1; team <= 8; team *= 8) {
            setenv("HPGV_STAGE_THREADS", team == 1 ? "1" : "8", 1);
            hpgv_host_set_stage_threads(team);
            for (int k = 0; k < kb; k++) hpgv_host_stage_records(bt[k], B, N, 1, gt, gt + (size_t)B * N);
            const double t0 = now();
            long calls = 0;
            while (now() - t0 < secs * 0.5) { hpgv_host_stage_records(bt[calls % kb], B, N, 1, gt, gt + (size_t)B * N); calls++; }
            const double dt = now() - t0;
            printf("{\"what\": \"hpgv_host_stage_records alone\", \"staging_team\": %d, \"batch_variants\": %d, \"samples\": %d, \"us_per_batch\": %.1f, "
                   "\"ns_per_genotype\": %.3f, \"strings\": \"one strdup per sample, %d batches in rotation\"}\n",
                   team, B, N, dt / calls * 1e6, dt / calls / ((double)B * N) * 1e9, kb);
            fflush(stdout);
        }
        hpgv_host_set_stage_threads(0);
        free(gt);
        for (int k = 0; k < kb; k++) { for (int i = 0; i < B; i++) { array_list_free(bt[k][i]->samples, free); bt[k][i]->samples = NULL; } free(bt[k]); }
        free(bt);
    }
    if (stage_only) return 0;

    /* ---- cohort ---- */
    job_t J; memset(&J, 0, sizeof J);
    J.call = call; J.N = N; J.B = B;
    char (*names)[16] = (char (*)[16])malloc((size_t)N * 16);
    J.samples = (individual_t **)malloc(sizeof(void *) * (size_t)N);
    J.ids = sample_ids_new((size_t)N);
    if (call == CALL_TDT) {
        J.n_families = N / 3;
        J.families = (family_t **)malloc(sizeof(void *) * (size_t)J.n_families);
        for (int f = 0; f < J.n_families; f++) {
            char fid[16]; snprintf(fid, sizeof fid, "F%d", f);
            J.families[f] = family_new(strdup(fid));
            for (int m = 0; m < 3; m++) snprintf(names[3 * f + m], 16, "S%d", 3 * f + m);
            individual_t *fa = individual_new(names[3 * f], -1, MALE, UNAFFECTED, NULL, NULL, J.families[f]);
            individual_t *mo = individual_new(names[3 * f + 1], -1, FEMALE, UNAFFECTED, NULL, NULL, J.families[f]);
            individual_t *ch = individual_new(names[3 * f + 2], -1, (f & 1) ? MALE : FEMALE, AFFECTED, fa, mo, J.families[f]);
            family_set_parent(fa, J.families[f]); family_set_parent(mo, J.families[f]); family_add_child(ch, J.families[f]);
            J.samples[3 * f] = fa; J.samples[3 * f + 1] = mo; J.samples[3 * f + 2] = ch;
            for (int m = 0; m < 3; m++) sample_ids_put(J.ids, names[3 * f + m], 3 * f + m);
        }
    } else {
        for (int j = 0; j < N; j++) {
            snprintf(names[j], 16, "S%d", j);
            J.samples[j] = individual_new(names[j], -1, MALE, (j & 1) ? AFFECTED : UNAFFECTED, NULL, NULL, NULL);
            sample_ids_put(J.ids, names[j], j);
        }
    }
    if (call == CALL_FISHER) J.logfact = init_logarithm_array(N * 10);          /* assoc_runner.c:164-166 */
    J.fstats = file_stats_new();

    if (hpgv_host_init(0)) { fprintf(stderr, "hpgv_host_init: %s\n", hpgv_host_last_error()); return 1; }
    vcf_record_t ***batches = (vcf_record_t ***)malloc(sizeof *batches * (size_t)(wmax * K));
    for (int k = 0; k < wmax * K; k++) batches[k] = make_batch(B, N, (unsigned)k + 1);
    hpgv_host_adapter_profile(1);

    for (int wi = -1; wi < nw; wi++) {
        const int lone = wi < 0, T = lone ? 1 : wl[wi];
        for (int warm = 0; warm < 2; warm++) {
            list_t out;
            list_init("output", T, (size_t)T * (size_t)B * 4, &out);
            J.out = &out;
            long calls = 0, drained = 0;
            int failed = 0;
            hpgv_host_adapter_times(NULL, NULL, 1);
            const double t0 = now(), until = t0 + (warm ? secs : 0.2);
            if (lone) {
                /* not inside a parallel region: the drain runs on a plain thread of its own */
                pthread_t th;
                drain_t da = { &out, call, &drained };
                pthread_create(&th, NULL, drain_main, &da);
                while (now() < until && !failed) { failed = one_call(&J, batches[calls % K]); calls++; }
                list_decr_writers(&out);
                pthread_join(th, NULL);
            } else {
                #pragma omp parallel sections num_threads(2)
                {
                    #pragma omp section
                    {
                        #pragma omp parallel num_threads(T) reduction(+ : calls) reduction(| : failed)
                        {
                            const int w = omp_get_thread_num();
                            long c = 0;
                            while (now() < until && !failed) { failed |= one_call(&J, batches[w * K + (int)(c % K)]); c++; }
                            calls += c;
                            list_decr_writers(&out);
                        }
                    }
                    #pragma omp section
                    {
                        list_item_t *it;
                        while ((it = list_remove_item(&out))) { free_result(call, it->data_p); list_item_free(it); drained++; }
                    }
                }
            }
            const double dt = now() - t0;
            list_free_deep(&out, NULL);
            if (failed) { fprintf(stderr, "%s failed: %s\n", call_name, hpgv_host_last_error()); return 1; }
            if (!warm) continue;
            double s4[4]; long pc = 0;
            hpgv_host_adapter_times(s4, &pc, 1);
            if (pc < 1) pc = 1;
            const double us = 1e6 / (double)pc;
            printf("{\"call\": \"%s\", \"caller\": \"%s\", \"workers\": %d, \"batch_variants\": %d, \"samples\": %d, "
                   "\"us_per_call\": %.1f, \"us_staging\": %.1f, \"us_engine\": %.1f, \"us_records\": %.1f, "
                   "\"staging_ns_per_genotype\": %.3f, \"non_staging_share\": %.3f, \"calls_per_s\": %.1f, \"variants_per_s\": %.0f, \"records_drained\": %ld}\n",
                   call == CALL_ASSOC ? "assoc_test(CHI_SQUARE)" : call == CALL_FISHER ? "assoc_test(FISHER)" : call == CALL_TDT ? "tdt_test" : "get_variants_stats",
                   lone ? "alone (staging dealt to the idle cores)" : "runner's worker team (assoc_runner.c:106-207)", T, B, N,
                   s4[3] * us, s4[0] * us, s4[1] * us, s4[2] * us, s4[0] / (double)pc / ((double)B * N) * 1e9,
                   (s4[1] + s4[2]) / (s4[3] > 0 ? s4[3] : 1), calls / dt, calls * (double)B / dt, drained);
            fflush(stdout);
        }
    }
    hpgv_host_shutdown();
    return 0;
}
