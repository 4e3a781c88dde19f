/* group_epi.c -- a C host deals the epistasis scan to the members of a group context (hpgv_group_epi_rank), the way the
 * reference deals block coordinates to its workers (singlenode/epistasis_runner.c:114-145).
 *
 *   group_epi <n_members> <n_variants> <n_affected> <n_unaffected> <num_folds> <order> [self]
 *
 * A group of n_members contexts, ALL on device 0 (what a one-GPU box can do).  A seeded vcf2epi dataset and fold assignment
 * go to the group (every member receives them); the group's ranking must be BIT-IDENTICAL to the ranking one ordinary
 * context makes of the whole dataset (hpgv_epi_rank_pairs / hpgv_epi_rank_triples / hpgv_epi_rank_order), for both
 * evaluation subsets.  "self": member 0's list goes through ncclSend / ncclRecv to itself, so a one-member group moves its
 * records through RCCL.  The shares are printed: they tile [0, V) and, for pairs, start on multiples of 64. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpgv.h"

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        int rc_ = (call);                                                                                 \
        if (rc_ != HPGV_OK) {                                                                             \
            fprintf(stderr, "%s -> %d: %s / %s / %s\n", #call, rc_, hpgv_last_error(g), hpgv_last_error(one), hpgv_last_error(NULL)); \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

static hpgv_ctx *g, *one;

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: group_epi n_members n_variants n_affected n_unaffected num_folds order [self]\n"); return 2; }
    const int G = atoi(argv[1]), V = atoi(argv[2]), nA = atoi(argv[3]), nU = atoi(argv[4]), K = atoi(argv[5]), order = atoi(argv[6]);
    const int self = argc > 7 && !strcmp(argv[7], "self");
    if (G < 1 || G > 16 || V < 1 || nA < 1 || nU < 1 || K < 1 || K > 16) return 2;
    int ids[16] = {0};
    CHECK(hpgv_create_multi(ids, G, &g));
    CHECK(hpgv_create(0, &one));
    if (self) CHECK(hpgv_set_option(g, "group_self_exchange", 1));

    const int n = nA + nU, N = 24;
    uint8_t *data = malloc((size_t)V * (size_t)n);
    int32_t *fold = malloc(sizeof(int32_t) * (size_t)n);
    unsigned long long st = 0x9E3779B97F4A7C15ULL ^ (unsigned long long)(V * 131 + n);
    for (size_t i = 0; i < (size_t)V * (size_t)n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const unsigned r = (unsigned)(st >> 33) % 100u;
        data[i] = r < 50 ? 0 : r < 83 ? 1 : r < 98 ? 2 : 255;         /* 2 % missing calls */
    }
    for (int s = 0; s < n; ++s) fold[s] = (s < nA ? s : s - nA) % K;   /* both classes dealt round-robin to the folds */
    CHECK(hpgv_epi_set_dataset(g, data, V, nA, nU));
    CHECK(hpgv_epi_set_folds(g, fold, K));
    CHECK(hpgv_epi_set_dataset(one, data, V, nA, nU));
    CHECK(hpgv_epi_set_folds(one, fold, K));

    int prev = 0;
    printf("shares:");
    for (int k = 0; k < G; ++k) {
        int lo, hi;
        CHECK(hpgv_group_epi_share(g, order, k, &lo, &hi));
        printf(" [%d, %d)", lo, hi);
        if (lo != prev || hi < lo || (order == 2 && lo % 64 && lo != V)) { fprintf(stderr, "bad share of member %d: [%d, %d)\n", k, lo, hi); return 1; }
        prev = hi;
    }
    printf("\n");
    if (prev != V) { fprintf(stderr, "the shares do not cover the first SNPs\n"); return 1; }

    const size_t KN = (size_t)K * (size_t)N;
    int32_t *gc = calloc(KN * (size_t)order, sizeof(int32_t)), *oc = calloc(KN * (size_t)order, sizeof(int32_t));
    int32_t *ci = calloc(KN, sizeof(int32_t)), *cj = calloc(KN, sizeof(int32_t)), *ck = calloc(KN, sizeof(int32_t));
    double *ga = calloc(KN, sizeof(double)), *oa = calloc(KN, sizeof(double));
    uint32_t *gm = calloc(KN * 8, sizeof(uint32_t)), *om = calloc(KN * 8, sizeof(uint32_t));
    int32_t gn[16], on[16];
    int bad = 0;
    for (int subset = 0; subset < 2; ++subset) {
        memset(gm, 0, KN * 8 * sizeof(uint32_t)); memset(om, 0, KN * 8 * sizeof(uint32_t));
        CHECK(hpgv_group_epi_rank(g, order, subset, N, gc, ga, gm, gn, NULL));
        if (order == 2) CHECK(hpgv_epi_rank_pairs(one, subset, N, ci, cj, oa, om, on, NULL));
        else if (order == 3) CHECK(hpgv_epi_rank_triples(one, subset, N, ci, cj, ck, oa, om, on, NULL));
        else CHECK(hpgv_epi_rank_order(one, order, subset, N, oc, oa, om, on, NULL));
        for (int f = 0; f < K; ++f) {
            if (gn[f] != on[f]) { fprintf(stderr, "MISMATCH: fold %d ranks %d models, one context %d\n", f, gn[f], on[f]); bad = 1; continue; }
            for (int e = 0; e < gn[f]; ++e) {
                const size_t o = (size_t)f * (size_t)N + (size_t)e;
                int32_t want[5] = {-1, -1, -1, -1, -1};
                uint32_t wm[8] = {0};
                if (order <= 3) { want[0] = ci[o]; want[1] = cj[o]; if (order == 3) want[2] = ck[o]; wm[0] = om[o]; }      /* one mask word per model */
                else { for (int s = 0; s < order; ++s) want[s] = oc[o * (size_t)order + (size_t)s]; memcpy(wm, om + o * 8, sizeof wm); }
                if (memcmp(&ga[o], &oa[o], sizeof(double)) || memcmp(gm + o * 8, wm, sizeof wm)) { fprintf(stderr, "MISMATCH: subset %d fold %d rank %d (accuracy / risky cells)\n", subset, f, e); bad = 1; }
                for (int s = 0; s < order; ++s)
                    if (gc[o * (size_t)order + (size_t)s] != want[s]) { fprintf(stderr, "MISMATCH: subset %d fold %d rank %d SNP %d\n", subset, f, e, s); bad = 1; }
            }
        }
        printf("subset %d: ranking bit-identical (%d folds, top model of fold 0: accuracy %.6f)\n", subset, K, gn[0] ? ga[0] : -1.0);
    }
    printf("rccl_ranks=%d\n", hpgv_group_comm_ranks(g));
    if (!bad) printf("group epistasis ok: %d member(s), %d SNPs, order %d\n", G, V, order);
    hpgv_destroy(g);
    hpgv_destroy(one);
    return bad;
}
