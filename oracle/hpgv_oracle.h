/*
 * hpgv_oracle.h -- CPU ORACLE for the per-variant statistics hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a from-scratch plain-C restatement of
 * the reference algorithm (opencb/hpg-variant) used as the parity checker.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it.  The product (hpg-variant_amd/) never does.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - TDT transmission tallies : PINNED by the reference's 9 unit KATs
 *                                (test/test_tdt_runner.c:93-433)
 *   - check_mendel             : PINNED by the reference's 41 assertions
 *                                (test/test_checks_family.c:16-111), biallelic
 *   - assoc allele counting    : restated line by line from
 *                                src/gwas/assoc/assoc.c:87-134 (no reference test
 *                                exists; "parity unpinned" by reference KATs,
 *                                pinned by hand-derived tables in tests/)
 *   - chi-square / OR / p      : restated from assoc_basic_test.c:23-41,58-61;
 *                                cross-checked against scipy goldens
 *   - Fisher, get_alleles, HWE, variant stats: bodies live in the un-vendored
 *                                hpg-libs => PARITY UNPINNED; definitions
 *                                documented here, cross-checked vs exact
 *                                rational arithmetic / scipy goldens.
 *
 *   - epistasis / MDR counting : PINNED by the reference's unit tests
 *                                (test/test_epistasis_model.c:116-520: pair and triple
 *                                counts, counts per fold, confusion matrices, evaluation
 *                                formulas; test/test_mdr.c:33-65: high-risk cells);
 *                                restated from src/gwas/epistasis/model.c, mdr.c
 *                                (hpgv_epi_oracle.c)
 *
 * The reference itself cannot be compiled here (needs hpg-libs, GSL,
 * argtable2, libconfig; none present), so there is no oracle/_ref build.
 */
#ifndef HPGV_ORACLE_H
#define HPGV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enums mirror the (absent) hpg-libs ones only by meaning */
enum { ORC_SEX_MALE = 0, ORC_SEX_FEMALE = 1, ORC_SEX_UNKNOWN = 2 };
enum { ORC_COND_UNAFFECTED = 0, ORC_COND_AFFECTED = 1, ORC_COND_OTHER = 2 };

/* get_alleles() return codes [recollection of hpg-libs vcf_util.h; unpinned] */
enum {
    ORC_ALLELES_OK = 0,
    ORC_FIRST_ALLELE_MISSING = 1,
    ORC_SECOND_ALLELE_MISSING = 2,
    ORC_ALL_ALLELES_MISSING = 3,
    ORC_HAPLOID = 4
};

enum { ORC_TASK_CHISQ = 1, ORC_TASK_FISHER = 2 };   /* assoc.h:55 */

/* ---- text level (what assoc.c:45-56 / tdt.c:46-48,97-108 do per genotype) */
int  orc_get_field_position_in_format(const char *field, const char *format);
int  orc_get_alleles(const char *sample, int gt_position, int *a1, int *a2);

/* ---- packed genotype code (design of this repo, include/hpgv.h) */
uint8_t orc_encode_alleles(int status, int a1, int a2, int strict);
uint8_t orc_encode_sample(const char *sample, int gt_position, int strict);
/* returns ORC_ALLELES_OK and sets a1/a2, or non-zero when the byte is not a
 * complete diploid genotype */
int  orc_decode(uint8_t code, int *a1, int *a2);

/* ---- assoc (src/gwas/assoc/assoc.c:87-134, assoc_basic_test.c) */
void   orc_assoc_count_individual(int condition, int chrom_is_x, int a1, int a2,
                                  int *A1, int *A2, int *U1, int *U2);
double orc_assoc_basic_test(int a, int b, int c, int d);
double orc_assoc_odds_ratio(int A1, int A2, int U1, int U2);
double orc_chisq_p_value(double chisq);           /* 1 - gsl_cdf_chisq_P(x, 1) */
int    orc_chrom_is_x(const char *chrom, int len); /* assoc.c:94 */

/* Fisher (assoc_fisher_test.c:24-26 -> hpg-libs fisher_test TWO_SIDED) */
void   orc_init_logarithm_array(int n, double *table); /* table[i] = ln(i!) */
double orc_fisher_two_sided(int a, int b, int c, int d, const double *logfact);
/* terms of that sum a tail-cut evaluation needs (bench.py's algorithmic work count of the Fisher p-pass) */
long orc_fisher_terms_needed(int a, int b, int c, int d, const double *lf, double rel_cut);

/* batch on a packed variant-major matrix in VCF column order */
void orc_assoc_packed(const uint8_t *gt, size_t pitch, int n_variants, int n_samples,
                      const uint8_t *condition, const uint8_t *chrom_is_x /* may be NULL */,
                      int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2);
void orc_assoc_stats(int task, int n_variants, const int32_t *A1, const int32_t *A2,
                     const int32_t *U1, const int32_t *U2, const double *logfact,
                     double *odds, double *chisq /* NULL for Fisher */, double *p);

/* text-faithful batch: per genotype strdup + get_alleles + free, as assoc.c:50-57 */
void orc_assoc_text(const char *const *samples /* n_variants*n_samples strings */,
                    int n_variants, int n_samples, const char *const *formats,
                    const uint8_t *condition, const uint8_t *chrom_is_x,
                    int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2);

/* ---- TDT (src/gwas/tdt/tdt.c:56-260, 279-292) */
int  orc_check_mendel(const char *chrom, int f1, int f2, int m1, int m2,
                      int c1, int c2, int child_sex);
/* families in CSR form: family f has father_col[f], mother_col[f] (<0 => absent)
 * and qualifying children child_off[f]..child_off[f+1] (already filtered to
 * child->father && child->mother && AFFECTED && in sample_ids, in the order the
 * caller iterates family->members). */
void orc_tdt_packed(const uint8_t *gt, size_t pitch, int n_variants,
                    const uint8_t *chrom_is_x /* may be NULL */,
                    int n_families, const int32_t *father_col, const int32_t *mother_col,
                    const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
                    int32_t *t1, int32_t *t2);
/* Mendelian errors of trios (father_col, mother_col, child_col, child_sex)[n_trios] through
 * orc_check_mendel, counted per variant (errors[v]) and per trio (trio_errors[t], accumulated);
 * a trio with any genotype not ALLELES_OK is not checked */
void orc_mendel_counts(const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *chrom_is_x,
                       int n_trios, const int32_t *father_col, const int32_t *mother_col,
                       const int32_t *child_col, const uint8_t *child_sex,
                       int32_t *errors, int32_t *trio_errors);
void orc_tdt_stats(int n_variants, const int32_t *t1, const int32_t *t2,
                   double *odds, double *chisq, double *p);

/* ---- variant stats / HWE (hpg-libs get_variants_stats; PARITY UNPINNED) */
#define ORC_MAX_ALLELES 15
typedef struct {
    int32_t num_alleles;                    /* 1 + number of ALT */
    int32_t alleles_count[ORC_MAX_ALLELES];
    int32_t genotypes_count[ORC_MAX_ALLELES * ORC_MAX_ALLELES]; /* [a1*num_alleles+a2] */
    int32_t missing_alleles;
    int32_t missing_genotypes;
    double  maf;  int32_t maf_allele;
    double  mgf;  int32_t mgf_genotype;
    /* Hardy-Weinberg on the first two alleles */
    int32_t hw_n_AA, hw_n_Aa, hw_n_aa;
    double  hw_chi2, hw_p;
} orc_variant_stats_t;
void orc_variant_stats(const uint8_t *row, int n_samples, int num_alleles,
                       orc_variant_stats_t *out);
void orc_hwe(int n_AA, int n_Aa, int n_aa, double *chi2, double *p);
/* hpg-libs get_sample_stats (call site stats_runner.c:197-198): per sample, the number of
 * variants in which the genotype has a missing allele; accumulated into missing[] */
void orc_sample_missing(const uint8_t *gt, size_t pitch, int n_variants, int n_samples, int32_t *missing);

/* ---- text staging: VCF data lines -> HPGV8, the way the reference's parser + assoc.c:45-56
 *      would see them (TAB split, GT located in FORMAT, get_alleles per sample).  Returns the
 *      number of lines; rows beyond max_lines are not written.  status as include/hpgv.h. */
int orc_tokenize(const char *text, size_t bytes, int n_samples, int strict, int max_lines,
                 uint8_t *gt, size_t pitch, uint8_t *is_x, int32_t *status);

/* ---- synthetic cohort (bit-reproducible on device) */
#define ORC_SYNTH_SEED 0x4850475631ULL
uint64_t orc_splitmix64(uint64_t x);
void     orc_synth_thresholds(uint64_t variant, uint32_t thr[3]);
uint8_t  orc_synth_genotype(uint64_t variant, uint64_t sample, const uint32_t thr[3]);
void     orc_synth_matrix(uint64_t v0, int n_variants, int n_samples, size_t pitch, uint8_t *gt);

/* ---- CPU baseline driver (OpenMP, 200-variant batches like hpg-variant.conf:33) */
double orc_baseline_assoc_packed(const uint8_t *gt, size_t pitch, int n_variants, int n_samples,
                                 const uint8_t *condition, int n_threads, int *threads_used);
double orc_baseline_assoc_text(uint64_t v0, int n_variants, int n_samples,
                               const uint8_t *condition, int n_threads, int *threads_used);

/* the same driver for either statistic of assoc.c:59-76 (task ORC_TASK_CHISQ / ORC_TASK_FISHER with its table),
 * for tdt.c:41-271 and for the stats tool's per-variant pass; all text-faithful (strdup + get_alleles + free) */
double orc_baseline_assoc_text_task(uint64_t v0, int n_variants, int n_samples, const uint8_t *condition,
                                    int task, const double *logfact, int n_threads, int *threads_used);
void   orc_tdt_text(const char *const *samples, int n_variants, int n_samples, const char *const *formats,
                    const uint8_t *chrom_is_x, int n_families, const int32_t *father_col, const int32_t *mother_col,
                    const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
                    int32_t *t1, int32_t *t2);
double orc_baseline_tdt_text(uint64_t v0, int n_variants, int n_samples, int n_families,
                             const int32_t *father_col, const int32_t *mother_col, const int32_t *child_off,
                             const int32_t *child_col, const uint8_t *child_sex, int n_threads, int *threads_used);
double orc_baseline_stats_text(uint64_t v0, int n_variants, int n_samples, int n_threads, int *threads_used);

void orc_set_threads(int n);           /* team size of the oracle's OpenMP loops */

/* ---- epistasis / MDR (hpgv_epi_oracle.c; src/gwas/epistasis/model.c, mdr.c, epistasis.c) ---- */
void orc_epi_counts(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                    int32_t *counts_aff, int32_t *counts_unaff);                         /* model.c:76-124 */
void orc_epi_counts_all_folds(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                              const uint8_t *fold_masks, int num_folds,
                              int32_t *counts_aff, int32_t *counts_unaff);               /* model.c:126-206 */
int  orc_mdr_high_risk(unsigned count_affected, unsigned count_unaffected,
                       unsigned samples_affected, unsigned samples_unaffected);          /* mdr.c:23-42 */
int  orc_mdr_high_risk2(int count_affected, int count_unaffected,
                        unsigned num_affected, unsigned num_unaffected);                 /* mdr.c:45-76 */
void orc_epi_confusion(int order, const uint8_t *risky, int n_risky, const uint8_t *const *rows,
                       int n_affected, int n_unaffected, const uint8_t *fold_mask, int subset,
                       const int32_t size_aff_unaff[2], uint32_t matrix[4]);             /* model.c:337-456 */
double orc_epi_evaluate(const uint32_t matrix[4], int function);                         /* model.c:458-476 */
void orc_epi_model(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                   const uint8_t *fold_masks, int num_folds, int subset,
                   double *accuracy, uint32_t *risky_mask, uint32_t *matrices);          /* epistasis.c:14-95 */
void orc_epi_model_wide(int order, const uint8_t *const *rows, int n_affected, int n_unaffected,
                        const uint8_t *fold_masks, int num_folds, int subset, int mask_words,
                        double *accuracy, uint32_t *risky_mask, uint32_t *matrices);
void orc_epi_scan_pairs(const uint8_t *dataset, int n_variants, int n_affected, int n_unaffected,
                        const uint8_t *fold_masks, int num_folds, int subset,
                        double *accuracy, uint32_t *risky_mask);

#ifdef __cplusplus
}
#endif
#endif
