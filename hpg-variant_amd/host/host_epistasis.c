/* host_epistasis.c -- epistasis (MDR): k-fold masks (cross_validation.c), the run over a vcf2epi dataset, the report (epistasis_report.c:30-81).
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* epistasis (MDR): k-fold masks, the run over a vcf2epi dataset, the report  */
/* ------------------------------------------------------------------------ */

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

/* cross_validation.c:16-100: cases and controls are shuffled separately and dealt to the folds one case and
 * one control per fold and round, fold 0 first; sizes[3 * f + {0, 1, 2}] = samples, cases, controls of fold f;
 * every fold lists its samples in increasing order (cases have the indices below num_samples_affected).
 * The shuffle is hpg-libs' array_shuffle_int there; here a Fisher-Yates pass on rand(). */
int **get_k_folds(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k, unsigned int **sizes) {
    const unsigned int n = num_samples_affected + num_samples_unaffected;
    if (k == 0) return NULL;
    int *samples = (int *)malloc(sizeof(int) * (n + 1));
    int **folds = (int **)calloc(k, sizeof(int *));
    unsigned int *fold_sizes = (unsigned int *)calloc(3 * (size_t)k, sizeof(unsigned int));
    if (!samples || !folds || !fold_sizes) { free(samples); free(folds); free(fold_sizes); return NULL; }
    for (unsigned int i = 0; i < n; i++) samples[i] = (int)i;
    for (unsigned int part = 0; part < 2; part++) {
        int *a = samples + (part ? num_samples_affected : 0);
        const unsigned int m = part ? num_samples_unaffected : num_samples_affected;
        for (unsigned int i = m; i > 1; i--) { unsigned int j = (unsigned int)rand() % i; int t = a[i - 1]; a[i - 1] = a[j]; a[j] = t; }
    }
    for (unsigned int f = 0; f < k; f++) {                            /* round-robin: fold f gets every k-th case and control */
        fold_sizes[3 * f + 1] = num_samples_affected / k + (f < num_samples_affected % k);
        fold_sizes[3 * f + 2] = num_samples_unaffected / k + (f < num_samples_unaffected % k);
        fold_sizes[3 * f] = fold_sizes[3 * f + 1] + fold_sizes[3 * f + 2];
        folds[f] = (int *)malloc(sizeof(int) * (fold_sizes[3 * f] + 1));
        unsigned int o = 0;
        for (unsigned int i = f; i < num_samples_affected; i += k) folds[f][o++] = samples[i];
        for (unsigned int i = f; i < num_samples_unaffected; i += k) folds[f][o++] = samples[num_samples_affected + i];
        qsort(folds[f], fold_sizes[3 * f], sizeof(int), cmp_int);
    }
    free(samples);
    *sizes = fold_sizes;
    return folds;
}

/* cross_validation.c:247-281: k x num_samples_with_padding bytes, 1 = the sample is in the TRAINING part of the
 * fold, 0 = in its testing part or padding; cases and controls each padded to a multiple of 16 */
uint8_t *get_k_folds_masks(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k,
                           int **folds, unsigned int *sizes) {
    const size_t pad_a = ((size_t)num_samples_affected + 15) / 16 * 16, pad_u = ((size_t)num_samples_unaffected + 15) / 16 * 16;
    const size_t padded = pad_a + pad_u;
    uint8_t *masks = (uint8_t *)malloc(padded * k + 16);
    if (!masks) return NULL;
    memset(masks, 1, padded * k);
    for (unsigned int f = 0; f < k; f++) {
        uint8_t *m = masks + (size_t)f * padded;
        for (unsigned int j = 0; j < sizes[3 * f + 1]; j++) m[folds[f][j]] = 0;
        for (unsigned int j = sizes[3 * f + 1]; j < sizes[3 * f]; j++) m[(size_t)folds[f][j] + (pad_a - num_samples_affected)] = 0;
        memset(m + num_samples_affected, 0, pad_a - num_samples_affected);
        memset(m + pad_a + num_samples_unaffected, 0, pad_u - num_samples_unaffected);
    }
    return masks;
}

enum { EPI_ORDER_MAX = 5, EPI_MASK_WORDS = 8 };            /* 3^5 = 243 cells in 8 x 32 bits (hpgv.h "ANY order") */
typedef struct { int c[EPI_ORDER_MAX]; int count; uint32_t risky[EPI_MASK_WORDS]; double accuracy; } epi_model_t;     /* unused SNP slots: -1 */

static int cmp_model_comb(const void *a, const void *b) {            /* compare_risky, epistasis.c:162-175 */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    for (int s = 0; s < EPI_ORDER_MAX; s++) if (x->c[s] != y->c[s]) return x->c[s] < y->c[s] ? -1 : 1;
    return 0;
}
static int cmp_model_cva(const void *a, const void *b) {             /* CV-a: accuracy, then the combination */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    if (x->accuracy != y->accuracy) return x->accuracy > y->accuracy ? -1 : 1;
    return cmp_model_comb(a, b);
}
static int cmp_model_cvc(const void *a, const void *b) {             /* CV-c: folds that ranked the model, then accuracy */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    return cmp_model_cva(a, b);
}

int hpgv_run_epistasis(const char *dataset_path, int num_folds, int num_cv_repetitions, int max_ranking_size,
                       int eval_subset, int eval_mode, const char *out_prefix) {
    return hpgv_run_epistasis_order(dataset_path, 2, num_folds, num_cv_repetitions, max_ranking_size, eval_subset, eval_mode, out_prefix);
}

int hpgv_run_epistasis_order(const char *dataset_path, int order, int num_folds, int num_cv_repetitions, int max_ranking_size,
                             int eval_subset, int eval_mode, const char *out_prefix) {
    int rc = ensure_engine();
    if (rc) return rc;
    /* the reference's --order is any integer (main_epistasis.c:128,142); the engine takes 2 to 5 (243 cells) */
    if (order < 2 || order > EPI_ORDER_MAX) { snprintf(g_err, sizeof g_err, "combinations of %d SNPs are not supported (2 to %d)", order, EPI_ORDER_MAX); return HPGV_ERR_UNSUPPORTED; }
    if (num_folds < 1 || num_cv_repetitions < 1 || max_ranking_size < 1 || !dataset_path || !out_prefix) {
        snprintf(g_err, sizeof g_err, "bad epistasis arguments");
        return HPGV_ERR_INVALID;
    }
    FILE *f = fopen(dataset_path, "rb");
    uint32_t head[3];
    if (!f || fread(head, sizeof(uint32_t), 3, f) != 3) { if (f) fclose(f); snprintf(g_err, sizeof g_err, "cannot read the dataset %s", dataset_path); return HPGV_ERR_INVALID; }
    const size_t V = head[0], nA = head[1], nU = head[2], bytes = V * (nA + nU);
    uint8_t *data = (uint8_t *)malloc(bytes + 1);
    if (!data || fread(data, 1, bytes, f) != bytes) { fclose(f); free(data); snprintf(g_err, sizeof g_err, "the dataset %s is shorter than its header says", dataset_path); return HPGV_ERR_INVALID; }
    fclose(f);
    rc = hpgv_epi_set_dataset(g_ctx, data, (int)V, (int)nA, (int)nU);
    free(data);
    if (rc) return host_fail("hpgv_epi_set_dataset", rc);
    const size_t N = (size_t)max_ranking_size, K = (size_t)num_folds, O = (size_t)order;
    int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * K * N), *cj = (int32_t *)malloc(sizeof(int32_t) * K * N), *cnt = (int32_t *)malloc(sizeof(int32_t) * K);
    int32_t *ck = (int32_t *)malloc(sizeof(int32_t) * K * N);
    int32_t *cn = (int32_t *)malloc(sizeof(int32_t) * K * N * O);                    /* orders 4, 5: the combinations themselves */
    uint32_t *risky = (uint32_t *)calloc(K * N * EPI_MASK_WORDS, sizeof(uint32_t));
    double *acc = (double *)malloc(sizeof(double) * K * N);
    epi_model_t *all = (epi_model_t *)malloc(sizeof(epi_model_t) * (K * N + 1));
    char *path = (char *)malloc(strlen(out_prefix) + 32);
    if (!ci || !cj || !ck || !cn || !cnt || !risky || !acc || !all || !path) rc = HPGV_ERR_NOMEM;
    for (int r = 0; r < num_cv_repetitions && !rc; r++) {
        unsigned int *sizes = NULL;
        int **folds = get_k_folds((unsigned)nA, (unsigned)nU, (unsigned)num_folds, &sizes);
        uint8_t *masks = folds ? get_k_folds_masks((unsigned)nA, (unsigned)nU, (unsigned)num_folds, folds, sizes) : NULL;
        if (!masks) rc = HPGV_ERR_NOMEM;
        if (!rc && (rc = hpgv_epi_set_fold_masks(g_ctx, masks, num_folds))) host_fail("hpgv_epi_set_fold_masks", rc);
        /* orders 2 and 3: the tile scans (one mask word per model); any other order: the listed-combination kernel */
        if (!rc && order == 2 && (rc = hpgv_epi_rank_pairs(g_ctx, eval_subset, max_ranking_size, ci, cj, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_pairs", rc);
        if (!rc && order == 3 && (rc = hpgv_epi_rank_triples(g_ctx, eval_subset, max_ranking_size, ci, cj, ck, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_triples", rc);
        if (!rc && order > 3 && (rc = hpgv_epi_rank_order(g_ctx, order, eval_subset, max_ranking_size, cn, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_order", rc);
        if (folds) { for (int k = 0; k < num_folds; k++) free(folds[k]); free(folds); }
        free(sizes); free(masks);
        if (rc) break;
        size_t n = 0;
        for (size_t k = 0; k < K; k++)
            for (int e = 0; e < cnt[k]; e++) {
                const size_t o = k * N + (size_t)e;
                epi_model_t m;
                memset(&m, 0, sizeof m);
                for (int s = 0; s < EPI_ORDER_MAX; s++) m.c[s] = -1;
                if (order <= 3) { m.c[0] = ci[o]; m.c[1] = cj[o]; if (order == 3) m.c[2] = ck[o]; m.risky[0] = risky[o]; }
                else { for (size_t s = 0; s < O; s++) m.c[s] = cn[o * O + s]; memcpy(m.risky, risky + o * EPI_MASK_WORDS, sizeof m.risky); }
                m.count = 1; m.accuracy = acc[o];
                all[n++] = m;
            }
        qsort(all, n, sizeof *all, cmp_model_comb);                  /* stable enough: equal combinations differ only by fold */
        size_t m = 0;
        for (size_t e = 0; e < n; e++) {
            if (m > 0 && !cmp_model_comb(&all[m - 1], &all[e])) { all[m - 1].accuracy += all[e].accuracy; all[m - 1].count += 1; }
            else all[m++] = all[e];
        }
        for (size_t e = 0; e < m; e++) all[e].accuracy /= num_folds;
        qsort(all, m, sizeof *all, eval_mode == 1 ? cmp_model_cva : cmp_model_cvc);
        sprintf(path, "%s.cv%d.epi", out_prefix, r + 1);
        FILE *fd = fopen(path, "w");
        if (!fd) { snprintf(g_err, sizeof g_err, "cannot create %s", path); rc = HPGV_ERR_INVALID; break; }
        fprintf(fd, "#CROSS VALIDATION %d\n#COMBINATIONS OF: %d SNPs\n", r + 1, order);
        fprintf(fd, eval_mode == 1 ? "#EVALUATION MODE: Cross-validation accuracy\n" : "#EVALUATION MODE: Cross-validation consistency\n");
        fprintf(fd, eval_subset == HPGV_EPI_TRAINING ? "#EVALUATION PARTITION: Training\n" : "#EVALUATION PARTITION: Testing\n");
        fprintf(fd, "#POSITION\tSNPs\tGENOTYPES\tCV-C\tCV-A\n");
        int cells = 1;
        for (int s = 0; s < order; s++) cells *= 3;
        for (size_t e = 0; e < m && e < N; e++) {
            /* epistasis_report.c:62-77: "( i, j, ... )"; a risky cell "(g0-g1, g2, ..., gn), " -- '-' after the first genotype,
             * ", " after the ones between, the last one closes */
            fprintf(fd, "%d\t(", (int)e + 1);
            for (int s = 0; s < order - 1; s++) fprintf(fd, " %d,", all[e].c[s]);
            fprintf(fd, " %d )\t", all[e].c[order - 1]);
            for (int c = 0; c < cells; c++) {
                if (!(all[e].risky[c >> 5] >> (c & 31) & 1u)) continue;
                int g[EPI_ORDER_MAX], q = c;
                for (int s = order - 1; s >= 0; s--) { g[s] = q % 3; q /= 3; }         /* the last SNP varies fastest */
                fprintf(fd, "(%d-", g[0]);
                for (int s = 1; s < order - 1; s++) fprintf(fd, "%d, ", g[s]);
                fprintf(fd, "%d), ", g[order - 1]);
            }
            fprintf(fd, "%d\t%.3f\n", all[e].count, all[e].accuracy);
        }
        fclose(fd);
    }
    free(ci); free(cj); free(ck); free(cn); free(cnt); free(risky); free(acc); free(all); free(path);
    if (rc == HPGV_ERR_NOMEM) snprintf(g_err, sizeof g_err, "out of memory");
    return rc;
}
