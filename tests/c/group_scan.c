/* group_scan.c -- a C host drives the variant-sharded resident scan of a group context (hpgv_group_*), the way a C
 * runner would (the reference's runners are C: assoc_runner.c:106-207, tdt_runner.c:150-200, stats_runner.c:176-215).
 *
 *   group_scan <n_members> <n_variants> <n_samples> [self]
 *
 * A group of n_members contexts, ALL on device 0 (what a one-GPU box can do; on a node with several GPUs the ids would
 * differ and the hand-over would run over xGMI).  Every member's shard of a synthetic cohort is generated on its device
 * (hpgv_synth_dev with the shard's first variant id), the group scan runs twice (two generations of scratch), and all
 * results on member 0 must be BIT-IDENTICAL to one ordinary context scanning all variants.  "self": member 0 hands its own
 * results over through the communicator too (ncclSend / ncclRecv to itself), so a one-member group exercises RCCL.
 * The communicator is really created (ncclCommInitAll): hpgv_group_comm_ranks must say 1 here. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpgv.h"

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        int rc_ = (call);                                                                                 \
        if (rc_ != HPGV_OK) {                                                                             \
            fprintf(stderr, "%s -> %d: %s / %s\n", #call, rc_, hpgv_last_error(g), hpgv_last_error(NULL)); \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

static hpgv_ctx *g, *one;

static int same(const char *what, const void *d_a, hpgv_ctx *ca, const void *d_b, hpgv_ctx *cb, size_t bytes) {
    void *a = malloc(bytes ? bytes : 1), *b = malloc(bytes ? bytes : 1);
    int bad = 1;
    if (a && b && hpgv_memcpy_d2h(ca, a, d_a, bytes, NULL) == HPGV_OK && hpgv_memcpy_d2h(cb, b, d_b, bytes, NULL) == HPGV_OK)
        bad = memcmp(a, b, bytes) != 0;
    if (bad) fprintf(stderr, "MISMATCH: %s (%zu bytes)\n", what, bytes);
    free(a);
    free(b);
    return bad;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: group_scan n_members n_variants n_samples [self]\n"); return 2; }
    const int G = atoi(argv[1]);
    const int64_t V = atoll(argv[2]);
    const int N = atoi(argv[3]);
    const int self = argc > 4 && !strcmp(argv[4], "self");
    if (G < 1 || G > 16 || V < 0 || N < 3) return 2;
    int ids[16] = {0};
    CHECK(hpgv_create_multi(ids, G, &g));
    CHECK(hpgv_create(0, &one));
    if (self) CHECK(hpgv_set_option(g, "group_self_exchange", 1));

    /* cohort: odd samples are cases; trio k = columns (3k, 3k+1, 3k+2) */
    uint8_t *cond = malloc((size_t)N), *sex = malloc((size_t)N);
    const int T = N / 3;
    int32_t *fa = malloc(sizeof(int32_t) * (size_t)T), *mo = malloc(sizeof(int32_t) * (size_t)T), *ch = malloc(sizeof(int32_t) * (size_t)T),
            *off = malloc(sizeof(int32_t) * (size_t)(T + 1));
    for (int j = 0; j < N; ++j) cond[j] = (uint8_t)(j & 1);
    for (int k = 0; k < T; ++k) { fa[k] = 3 * k; mo[k] = 3 * k + 1; ch[k] = 3 * k + 2; off[k] = k; sex[k] = (uint8_t)(k & 1); }
    off[T] = T;
    double *lf = malloc(sizeof(double) * (size_t)(N * 10));
    lf[0] = 0.0;
    for (int i = 1; i < N * 10; ++i) lf[i] = lf[i - 1] + __builtin_log((double)i);
    hpgv_ctx *both[2] = {g, one};
    for (int c = 0; c < 2; ++c) {
        CHECK(hpgv_set_cohort(both[c], cond, N));
        CHECK(hpgv_set_families(both[c], 3 * T, T, fa, mo, off, ch, sex));
        CHECK(hpgv_set_stats_cohort(both[c], N));
        CHECK(hpgv_set_logfact(both[c], lf, (size_t)N * 10));
    }
    CHECK(hpgv_group_comm_init(g));
    const int ranks = hpgv_group_comm_ranks(g);
    printf("members=%d rccl_ranks=%d self_exchange=%d\n", hpgv_group_size(g), ranks, self);
    if (ranks != 1) { fprintf(stderr, "expected a 1-rank communicator over device 0\n"); return 1; }

    int bad = 0;
    for (int which = 0; which < 3; ++which) {              /* HPGV_LAYOUT_ASSOC, _TDT, _STATS */
        size_t pitch = 0;
        if (which == 0) CHECK(hpgv_assoc_layout(one, NULL, NULL, &pitch));
        if (which == 1) CHECK(hpgv_tdt_layout(one, NULL, NULL, &pitch));
        if (which == 2) CHECK(hpgv_stats_layout(one, &pitch));
        /* the shards, each on its member's device, and the whole cohort for the single context */
        const uint8_t *d_gt[16] = {0};
        void *d_all = NULL;
        CHECK(hpgv_dev_alloc(one, (size_t)V * pitch + 16, &d_all));
        CHECK(hpgv_synth_dev(one, which, 0, (int)V, d_all, NULL));
        int64_t covered = 0;
        for (int k = 0; k < G; ++k) {
            int64_t lo, hi;
            CHECK(hpgv_group_shard(g, V, k, &lo, &hi));
            if (lo != covered) { fprintf(stderr, "shards are not contiguous\n"); return 1; }
            covered = hi;
            hpgv_ctx *m = hpgv_group_member(g, k);
            void *p = NULL;
            CHECK(hpgv_dev_alloc(m, (size_t)(hi - lo) * pitch + 16, &p));
            CHECK(hpgv_synth_dev(m, which, (uint64_t)lo, (int)(hi - lo), p, NULL));
            CHECK(hpgv_stream_sync(m, NULL));
            d_gt[k] = p;
        }
        if (covered != V) { fprintf(stderr, "shards do not cover the cohort\n"); return 1; }
        CHECK(hpgv_stream_sync(one, NULL));
        /* result arrays: two sets for the group (the second call's scans overlap the first call's hand-over), one for the reference */
        const size_t ints = which == 0 ? 16 : which == 1 ? 8 : 32;
        void *r_int = NULL, *r_f[3] = {0}, *g_int[2] = {0}, *g_f[2][3] = {{0}};
        void *r_miss = NULL, *g_miss[2] = {0};
        hpgv_ctx *m0 = hpgv_group_member(g, 0);
        CHECK(hpgv_dev_alloc(one, (size_t)V * ints + 16, &r_int));
        for (int i = 0; i < 3; ++i) CHECK(hpgv_dev_alloc(one, (size_t)V * 8 + 16, &r_f[i]));
        CHECK(hpgv_dev_alloc(one, (size_t)N * 4 + 16, &r_miss));
        for (int s = 0; s < 2; ++s) {
            CHECK(hpgv_dev_alloc(m0, (size_t)V * ints + 16, &g_int[s]));
            for (int i = 0; i < 3; ++i) CHECK(hpgv_dev_alloc(m0, (size_t)V * 8 + 16, &g_f[s][i]));
            CHECK(hpgv_dev_alloc(m0, (size_t)N * 4 + 16, &g_miss[s]));
        }
        for (int task = 1; task <= (which == 0 ? 2 : 1); ++task) {
            /* reference: one context, every variant */
            if (which == 0) {
                CHECK(hpgv_assoc_scan_dev(one, d_all, (int)V, NULL, r_int, NULL));
                if (task == HPGV_TASK_CHISQ) CHECK(hpgv_assoc_chisq_dev(one, r_int, (int)V, r_f[0], r_f[1], r_f[2], NULL));
                else CHECK(hpgv_assoc_fisher_dev(one, r_int, (int)V, r_f[0], r_f[2], NULL));
            } else if (which == 1) {
                CHECK(hpgv_tdt_scan_dev(one, d_all, (int)V, NULL, r_int, NULL));
                CHECK(hpgv_tdt_stats_dev(one, r_int, (int)V, r_f[0], r_f[1], r_f[2], NULL));
            } else {
                CHECK(hpgv_stats_scan_dev(one, d_all, (int)V, r_int, NULL));
                CHECK(hpgv_stats_hwe_dev(one, r_int, (int)V, r_f[1], r_f[2], NULL));
                static int32_t zero[1 << 20];
                CHECK(hpgv_memcpy_h2d(one, r_miss, zero, (size_t)N * 4, NULL));
                for (int64_t v0 = 0; v0 < V; v0 += 1000000)
                    CHECK(hpgv_sample_missing_dev(one, (const uint8_t *)d_all + (size_t)v0 * pitch, (int)(V - v0 < 1000000 ? V - v0 : 1000000), r_miss, NULL));
            }
            CHECK(hpgv_stream_sync(one, NULL));
            /* the group: three calls back to back, alternating result sets; the last two are compared */
            for (int call = 0; call < 3; ++call) {
                const int s = call & 1;
                if (which == 0)
                    CHECK(hpgv_group_assoc(g, task, d_gt, NULL, V, g_int[s], g_f[s][0], task == HPGV_TASK_CHISQ ? g_f[s][1] : NULL, g_f[s][2]));
                else if (which == 1)
                    CHECK(hpgv_group_tdt(g, d_gt, NULL, V, g_int[s], g_f[s][0], g_f[s][1], g_f[s][2]));
                else
                    CHECK(hpgv_group_stats(g, d_gt, V, g_int[s], g_f[s][1], g_f[s][2], g_miss[s]));
            }
            CHECK(hpgv_group_sync(g));
            for (int s = 0; s < 2; ++s) {
                bad |= same("integer tallies", g_int[s], m0, r_int, one, (size_t)V * ints);
                for (int i = 0; i < 3; ++i) {
                    if ((which == 0 && task == HPGV_TASK_FISHER && i == 1) || (which == 2 && i == 0)) continue;
                    bad |= same(i == 0 ? "odds" : i == 1 ? "chi-square" : "p", g_f[s][i], m0, r_f[i], one, (size_t)V * 8);
                }
                if (which == 2) bad |= same("per-sample missing counters", g_miss[s], m0, r_miss, one, (size_t)N * 4);
            }
            printf("layout %d task %d: %s\n", which, task, bad ? "DIFFERENT" : "bit-identical to the single-context scan");
        }
        for (int k = 0; k < G; ++k) CHECK(hpgv_dev_free(hpgv_group_member(g, k), (void *)d_gt[k]));
        CHECK(hpgv_dev_free(one, d_all));
        CHECK(hpgv_dev_free(one, r_int));
        CHECK(hpgv_dev_free(one, r_miss));
        for (int i = 0; i < 3; ++i) CHECK(hpgv_dev_free(one, r_f[i]));
        for (int s = 0; s < 2; ++s) {
            CHECK(hpgv_dev_free(m0, g_int[s]));
            CHECK(hpgv_dev_free(m0, g_miss[s]));
            for (int i = 0; i < 3; ++i) CHECK(hpgv_dev_free(m0, g_f[s][i]));
        }
    }
    hpgv_destroy(g);
    hpgv_destroy(one);
    free(cond); free(sex); free(fa); free(mo); free(ch); free(off); free(lf);
    if (bad) return 1;
    printf("group scan ok\n");
    return 0;
}
