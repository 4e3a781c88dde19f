/*
 * hpgv_epi_oracle.c -- CPU ORACLE of the epistasis (MDR) counting path (test
 * infrastructure, never shipped, never on the product path).  Plain scalar C
 * restatement of the reference's SSE code; every function cites the lines it
 * follows.  Pinned by the reference's own unit tests (tests/golden/
 * reference_kats.json, section "epistasis": test/test_epistasis_model.c:116-520,
 * test/test_mdr.c:33-65).
 *
 * Data model (src/gwas/epistasis/dataset.c:63-76, cross_validation.c:247-281):
 * the vcf2epi dataset holds one row per SNP, num_affected + num_unaffected
 * bytes, cases first, codes 0 / 1 / 2 and 255 for a missing call.  The
 * reference pads both groups to multiples of 16 for its SSE loads; the
 * padding carries no information and does not exist here.
 */
#define _GNU_SOURCE
#include "hpgv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int ipow3(int order) { int n = 1; for (int i = 0; i < order; i++) n *= 3; return n; }

/* genotype of SNP j in cell c of an order-`order` table: the last SNP varies fastest
 * (get_genotype_combinations / get_next_genotype_combination, dataset.c:170-200) */
static int cell_genotype(int order, int c, int j) {
    for (int k = order - 1; k > j; k--) c /= 3;
    return c % 3;
}

/* combination_counts, model.c:76-124: for every cell, the number of affected and of
 * unaffected samples whose genotypes at the `order` SNPs are the cell's (the reference
 * ANDs 0xFF byte masks and divides the popcount by 8). */
void orc_epi_counts(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                    int32_t *counts_aff, int32_t *counts_unaff) {
    const int cells = ipow3(order);
    for (int c = 0; c < cells; c++) {
        int aff = 0, unaff = 0;
        for (int s = 0; s < n_affected + n_unaffected; s++) {
            int match = 1;
            for (int j = 0; j < order && match; j++) match = rows[j][s] == cell_genotype(order, c, j);
            if (match) { if (s < n_affected) aff++; else unaff++; }
        }
        counts_aff[c] = aff; counts_unaff[c] = unaff;
    }
}

/* combination_counts_all_folds, model.c:126-206: the same counts restricted to the samples
 * whose fold mask byte is 1 (the training part of fold f), laid out [fold][cell] as for one
 * combination in a row (model.c:166-168).  fold_masks: num_folds x (n_affected + n_unaffected). */
void orc_epi_counts_all_folds(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                              const uint8_t *fold_masks, int num_folds,
                              int32_t *counts_aff, int32_t *counts_unaff) {
    const int cells = ipow3(order), n = n_affected + n_unaffected;
    for (int f = 0; f < num_folds; f++)
        for (int c = 0; c < cells; c++) {
            int aff = 0, unaff = 0;
            for (int s = 0; s < n; s++) {
                if (!(fold_masks[(size_t)f * n + s] & 1)) continue;
                int match = 1;
                for (int j = 0; j < order && match; j++) match = rows[j][s] == cell_genotype(order, c, j);
                if (match) { if (s < n_affected) aff++; else unaff++; }
            }
            counts_aff[f * cells + c] = aff; counts_unaff[f * cells + c] = unaff;
        }
}

/* mdr_high_risk_combinations, mdr.c:23-42 (double arithmetic; used by the reference's tests) */
int orc_mdr_high_risk(unsigned count_affected, unsigned count_unaffected,
                      unsigned samples_affected, unsigned samples_unaffected) {
    if (count_affected == 0 && count_unaffected == 0) return 0;
    int total_in_cell = (int)(count_affected + count_unaffected);
    double affected_unaffected_ratio = (double)samples_affected / samples_unaffected;
    double proportional_unaffected = count_unaffected * affected_unaffected_ratio;
    double reduction_ratio = total_in_cell / (proportional_unaffected + count_affected);
    double normalized_unaffected = proportional_unaffected * reduction_ratio;
    double normalized_affected = total_in_cell - normalized_unaffected;
    return normalized_affected >= normalized_unaffected;
}

/* mdr_high_risk_combinations2, mdr.c:45-76: what the runner uses (epistasis.c:30-34).  SINGLE
 * precision, operation by operation as the SSE code; an empty cell gives 0/0 = NaN and the
 * comparison is false.  volatile keeps gcc from contracting or widening the float operations. */
int orc_mdr_high_risk2(int count_affected, int count_unaffected, unsigned num_affected, unsigned num_unaffected) {
    volatile float ratio = (float)num_affected / num_unaffected;
    volatile float ca = (float)count_affected, cu = (float)count_unaffected;
    volatile float total = ca + cu;
    volatile float prop = cu * ratio;
    volatile float denom = prop + ca;
    volatile float red = total / denom;
    volatile float norm_unaff = prop * red;
    volatile float norm_aff = total - norm_unaff;
    return norm_aff >= norm_unaff;
}

/* confusion_matrix, model.c:337-456.  risky: n_risky cells, `order` genotypes each
 * (risky_combination.genotypes).  fold_mask: one byte per sample, 1 = training part.
 * subset 1 = TRAINING, 0 = TESTING (enum evaluation_subset, model.h:73).  matrix = {TP, FN, FP, TN}.
 * With no risky cell the reference reads an empty array (undefined); here nobody is predicted. */
void orc_epi_confusion(int order, const uint8_t *risky, int n_risky, const uint8_t *const *rows,
                       int n_affected, int n_unaffected, const uint8_t *fold_mask, int subset,
                       const int32_t size_aff_unaff[2], uint32_t matrix[4]) {
    int pop_aff = 0, pop_unaff = 0;
    for (int s = 0; s < n_affected + n_unaffected; s++) {
        int predicted = 0;
        for (int i = 0; i < n_risky && !predicted; i++) {
            int match = 1;
            for (int j = 0; j < order && match; j++) match = rows[j][s] == risky[i * order + j];
            predicted = match;
        }
        const int in_training = fold_mask[s] & 1;
        if (predicted && (subset == 1 ? in_training : !in_training)) { if (s < n_affected) pop_aff++; else pop_unaff++; }
    }
    matrix[0] = (uint32_t)pop_aff;                                   /* TP */
    matrix[2] = (uint32_t)pop_unaff;                                 /* FP */
    matrix[1] = (uint32_t)(size_aff_unaff[0] - pop_aff);             /* FN */
    matrix[3] = (uint32_t)(size_aff_unaff[1] - pop_unaff);           /* TN */
}

/* evaluate_model, model.c:458-476: 0 CA, 1 BA, 2 wBA (not implemented there), 3 GAMMA, 4 TAU_B */
double orc_epi_evaluate(const uint32_t m[4], int function) {
    double TP = m[0], FN = m[1], FP = m[2], TN = m[3];
    switch (function) {
        case 0: return (TP + TN) / (TP + FN + TN + FP);
        case 3: return (TP * TN - FP * FN) / (TP * TN + FP * FN);
        case 4: return (TP * TN - FP * FN) / sqrt((TP + FN) * (TN + FP) * (TP + FP) * (TN + FN));
        default: return ((TP / (TP + FN)) + (TN / (TN + FP))) / 2;
    }
}

/* process_set_of_combinations for ONE combination, epistasis.c:14-95: training counts per fold
 * (combination_counts_all_folds), high-risk cells (choose_high_risk_combinations2 with
 * mdr_high_risk_combinations2 on the WHOLE cohort's num_affected / num_unaffected,
 * epistasis.c:30-34), confusion matrix on the chosen subset with the fold's training / testing
 * sizes (test_model, model.c:320-335) and balanced accuracy.  Outputs per fold: accuracy,
 * bit c of risky_mask = cell c is high risk, and the confusion matrix. */
/* risky_mask: mask_words 32-bit words per fold (bit c % 32 of word c / 32 = cell c is high risk): 1 word holds the 9 or 27
 * cells of orders 2 and 3, 8 words the 243 of order 5 */
void orc_epi_model_wide(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                        const uint8_t *fold_masks, int num_folds, int subset, int mask_words,
                        double *accuracy, uint32_t *risky_mask, uint32_t *matrices /* num_folds x 4, may be NULL */) {
    const int cells = ipow3(order), n = n_affected + n_unaffected;
    int32_t *ca = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells * num_folds);
    int32_t *cu = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells * num_folds);
    uint8_t *risky = (uint8_t *)malloc((size_t)cells * order);
    orc_epi_counts_all_folds(order, rows, n_affected, n_unaffected, fold_masks, num_folds, ca, cu);
    for (int f = 0; f < num_folds; f++) {
        const uint8_t *mask = fold_masks + (size_t)f * n;
        int n_risky = 0;
        for (int w = 0; w < mask_words; w++) risky_mask[(size_t)f * mask_words + w] = 0;
        for (int c = 0; c < cells; c++)
            if (orc_mdr_high_risk2(ca[f * cells + c], cu[f * cells + c], (unsigned)n_affected, (unsigned)n_unaffected)) {
                for (int j = 0; j < order; j++) risky[n_risky * order + j] = (uint8_t)cell_genotype(order, c, j);
                n_risky++;
                if (c / 32 < mask_words) risky_mask[(size_t)f * mask_words + c / 32] |= 1u << (c % 32);
            }
        /* training_sizes / testing_sizes of the fold (epistasis_runner.c:96-101) */
        int32_t train[2] = {0, 0}, size[2];
        for (int s = 0; s < n; s++) if (mask[s] & 1) train[s < n_affected ? 0 : 1]++;
        size[0] = subset == 1 ? train[0] : n_affected - train[0];
        size[1] = subset == 1 ? train[1] : n_unaffected - train[1];
        uint32_t m[4];
        orc_epi_confusion(order, risky, n_risky, rows, n_affected, n_unaffected, mask, subset, size, m);
        accuracy[f] = orc_epi_evaluate(m, 1);
        if (matrices) memcpy(matrices + 4 * f, m, sizeof m);
    }
    free(ca); free(cu); free(risky);
}

void orc_epi_model(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                   const uint8_t *fold_masks, int num_folds, int subset,
                   double *accuracy, uint32_t *risky_mask, uint32_t *matrices /* num_folds x 4, may be NULL */) {
    orc_epi_model_wide(order, rows, n_affected, n_unaffected, fold_masks, num_folds, subset, 1, accuracy, risky_mask, matrices);
}

/* every pair i < j of the dataset (the union of the runner's blocks, dataset.c:94-168):
 * accuracy[f * n_pairs + p], risky_mask[f * n_pairs + p] with p = the pair's rank in
 * lexicographic order ((0,1), (0,2), ..., (1,2), ...) */
void orc_epi_scan_pairs(const uint8_t *dataset, int n_variants, int n_affected, int n_unaffected,
                        const uint8_t *fold_masks, int num_folds, int subset,
                        double *accuracy, uint32_t *risky_mask) {
    const size_t n = (size_t)(n_affected + n_unaffected);
    const size_t n_pairs = (size_t)n_variants * (size_t)(n_variants - 1) / 2;
    #pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n_variants; i++) {
        double acc[64]; uint32_t rm[64];
        size_t p = (size_t)i * (size_t)(2 * n_variants - i - 1) / 2;
        for (int j = i + 1; j < n_variants; j++, p++) {
            const uint8_t *rows[2] = { dataset + (size_t)i * n, dataset + (size_t)j * n };
            orc_epi_model(2, rows, n_affected, n_unaffected, fold_masks, num_folds, subset, acc, rm, NULL);
            for (int f = 0; f < num_folds; f++) { accuracy[(size_t)f * n_pairs + p] = acc[f]; risky_mask[(size_t)f * n_pairs + p] = rm[f]; }
        }
    }
}
