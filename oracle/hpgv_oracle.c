/*
 * hpgv_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on
 * the product path).  Plain-C restatement of the reference algorithm of
 * opencb/hpg-variant's per-variant statistics path.  Every function cites the
 * reference file:line it follows.  See hpgv_oracle.h for the pinning status.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#define _GNU_SOURCE
#include "hpgv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Text level
 * ---------------------------------------------------------------------- */

/* hpg-libs get_field_position_in_format(), called at assoc.c:46, tdt.c:47:
 * index of `field` among the ':'-separated FORMAT keys, -1 when absent. */
int orc_get_field_position_in_format(const char *field, const char *format) {
    size_t flen = strlen(field);
    int pos = 0;
    const char *p = format;
    while (*p) {
        const char *e = strchr(p, ':');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == flen && !strncmp(p, field, flen)) return pos;
        if (!e) break;
        p = e + 1;
        pos++;
    }
    return -1;
}

/* hpg-libs get_alleles(), called at assoc.c:53, tdt.c:103-104,154.
 * [recollection; PARITY UNPINNED beyond "a/b" and "./." which the TDT KATs and
 * test_epistasis_dataset.c:106-154 exercise.]  Definition used here:
 *   - take the gt_position-th ':'-separated field of the sample string;
 *   - alleles are separated by '/' or '|';
 *   - an allele that is "." (or empty) is missing; otherwise atoi();
 *   - return 0 OK, 1 first missing, 2 second missing, 3 both missing,
 *     4 haploid (single allele, no separator). */
int orc_get_alleles(const char *sample, int gt_position, int *a1, int *a2) {
    const char *p = sample;
    for (int i = 0; i < gt_position; i++) {
        const char *e = strchr(p, ':');
        if (!e) { *a1 = -1; *a2 = -1; return ORC_ALL_ALLELES_MISSING; }
        p = e + 1;
    }
    const char *end = p;
    while (*end && *end != ':') end++;
    const char *sep = p;
    while (sep < end && *sep != '/' && *sep != '|') sep++;

    int ret = ORC_ALLELES_OK;
    /* first allele: [p, sep) */
    if (sep == p || (sep - p == 1 && *p == '.')) {
        *a1 = -1; ret += ORC_FIRST_ALLELE_MISSING;
    } else {
        *a1 = atoi(p);
    }
    if (sep == end) {                     /* no separator: haploid call */
        *a2 = -1;
        return (ret == ORC_ALLELES_OK) ? ORC_HAPLOID : ORC_ALL_ALLELES_MISSING;
    }
    const char *q = sep + 1;
    if (q == end || (end - q == 1 && *q == '.')) {
        *a2 = -1; ret += ORC_SECOND_ALLELE_MISSING;
    } else {
        *a2 = atoi(q);
    }
    return ret;
}

/* ------------------------------------------------------------------------
 * Packed code (this repo's design; include/hpgv.h): byte = a1<<4 | a2,
 * allele indices clamped to 14, nibble 0xF = missing allele, 0xFF = missing
 * genotype.  strict=1 (assoc / tdt staging): anything get_alleles() does not
 * report as ALLELES_OK becomes 0xFF, because assoc.c:53 and tdt.c:103-108,154
 * drop such genotypes entirely.
 * ---------------------------------------------------------------------- */
uint8_t orc_encode_alleles(int status, int a1, int a2, int strict) {
    if (strict && status != ORC_ALLELES_OK) return 0xFF;
    int n1 = (status == ORC_FIRST_ALLELE_MISSING || status == ORC_ALL_ALLELES_MISSING || a1 < 0) ? 0xF
             : (a1 > 14 ? 14 : a1);
    int n2 = (status == ORC_SECOND_ALLELE_MISSING || status == ORC_ALL_ALLELES_MISSING ||
              status == ORC_HAPLOID || a2 < 0) ? 0xF : (a2 > 14 ? 14 : a2);
    /* the reference compares the raw allele ints (tdt.c:113,185-187): two different alleles that both clamp to 14 are
     * stored as 13/14 so that a1 != a2 survives the packed code */
    if (n1 == 14 && n2 == 14 && a1 != a2) n1 = 13;
    return (uint8_t)((n1 << 4) | n2);
}

uint8_t orc_encode_sample(const char *sample, int gt_position, int strict) {
    int a1, a2;
    int st = orc_get_alleles(sample, gt_position, &a1, &a2);
    return orc_encode_alleles(st, a1, a2, strict);
}

int orc_decode(uint8_t code, int *a1, int *a2) {
    int n1 = code >> 4, n2 = code & 0xF;
    *a1 = (n1 == 0xF) ? -1 : n1;
    *a2 = (n2 == 0xF) ? -1 : n2;
    if (n1 == 0xF && n2 == 0xF) return ORC_ALL_ALLELES_MISSING;
    if (n1 == 0xF) return ORC_FIRST_ALLELE_MISSING;
    if (n2 == 0xF) return ORC_SECOND_ALLELE_MISSING;
    return ORC_ALLELES_OK;
}

/* ------------------------------------------------------------------------
 * assoc
 * ---------------------------------------------------------------------- */

/* assoc.c:94  `!strncmp("X", record->chromosome, record->chromosome_len)` */
int orc_chrom_is_x(const char *chrom, int len) {
    return !strncmp("X", chrom, (size_t)len);
}

/* assoc.c:87-134, statement by statement. */
void orc_assoc_count_individual(int condition, int chrom_is_x, int allele1, int allele2,
                                int *affected1, int *affected2, int *unaffected1, int *unaffected2) {
    int A1 = 0, A2 = 0, U1 = 0, U2 = 0;
    if (chrom_is_x) {                                   /* assoc.c:94-107 */
        if (condition == ORC_COND_AFFECTED) {
            if (!allele1 && !allele2) A1++;
            else if (allele1 && allele2) A2++;
        } else if (condition == ORC_COND_UNAFFECTED) {
            if (!allele1 && !allele2) U1++;
            else if (allele1 && allele2) U2++;
        }
    } else {                                            /* assoc.c:108-125 */
        if (condition == ORC_COND_AFFECTED) {
            if (!allele1 && !allele2) A1 += 2;
            else if (allele1 && allele2) A2 += 2;
            else if (allele1 != allele2) { A1++; A2++; }
        } else if (condition == ORC_COND_UNAFFECTED) {
            if (!allele1 && !allele2) U1 += 2;
            else if (allele1 && allele2) U2 += 2;
            else if (allele1 != allele2) { U1++; U2++; }
        }
    }
    *affected1 += A1; *affected2 += A2; *unaffected1 += U1; *unaffected2 += U2;
}

/* assoc_basic_test.c:23-41; called as (A1, U1, A2, U2) at assoc.c:61.
 * Same operation order; build with -ffp-contract=off. */
double orc_assoc_basic_test(int a, int b, int c, int d) {
    double total_alleles = a + c + b + d;
    double total_affected = a + c;
    double total_unaffected = b + d;
    double total_allele1 = a + b;
    double total_allele2 = c + d;

    double e_a1 = (total_affected * total_allele1) / total_alleles;
    double e_a2 = (total_affected * total_allele2) / total_alleles;
    double e_u1 = (total_unaffected * total_allele1) / total_alleles;
    double e_u2 = (total_unaffected * total_allele2) / total_alleles;

    return ((a - e_a1) * (a - e_a1)) / e_a1 +
           ((c - e_a2) * (c - e_a2)) / e_a2 +
           ((b - e_u1) * (b - e_u1)) / e_u1 +
           ((d - e_u2) * (d - e_u2)) / e_u2;
}

/* assoc_basic_test.c:58-59 */
double orc_assoc_odds_ratio(int A1, int A2, int U1, int U2) {
    return (A2 == 0 || U1 == 0) ? NAN : ((double)A1 / A2) * ((double)U2 / U1);
}

/* assoc_basic_test.c:61, tdt.c:292:  p = 1 - gsl_cdf_chisq_P(x, 1).
 * GSL (absent) evaluates gsl_cdf_chisq_P(x,nu) = gsl_cdf_gamma_P(x, nu/2, 2):
 *   x <= 0        -> P = 0
 *   y = x/2 > a   -> P = 1 - Q(a, y)          (a = 1/2: Q = erfc(sqrt y))
 *   otherwise     -> P = P(a, y)              (a = 1/2: P = erf(sqrt y))
 * Restated with the closed forms for a = 1/2, keeping the 1-(1-Q) structure so
 * tiny p-values quantise as the reference's do.  NaN in -> NaN out (GSL's
 * behaviour on NaN is an error-handler call; unpinned). */
double orc_chisq_p_value(double x) {
    if (isnan(x)) return NAN;
    if (x <= 0.0) return 1.0;
    double y = x / 2.0;
    double P;
    if (y > 0.5) P = 1.0 - erfc(sqrt(y));
    else         P = erf(sqrt(y));
    return 1.0 - P;
}

/* hpg-libs init_logarithm_array(), called at assoc_runner.c:164-166 with
 * num_samples*10 entries [recollection: cumulative sum of log(i)].  table[i] = ln(i!) */
void orc_init_logarithm_array(int n, double *table) {
    if (n <= 0) return;
    table[0] = 0.0;
    for (int i = 1; i < n; i++) table[i] = table[i - 1] + log((double)i);
}

/* hpg-libs fisher_test(a,b,c,d,TWO_SIDED,logfact) via assoc_fisher_test.c:24-26,
 * called as (A1, A2, U1, U2) at assoc.c:70.  Body not in tree: PARITY UNPINNED.
 * Definition: table [[a,b],[c,d]]; sum of the hypergeometric probabilities of
 * all tables with the observed margins whose probability is <= p_observed *
 * (1 + 1e-7) (the relative tolerance R's fisher.test uses), each probability
 * evaluated through the log-factorial table; clamped to 1. */
double orc_fisher_two_sided(int a, int b, int c, int d, const double *lf) {
    int r1 = a + b, r2 = c + d, c1 = a + c, n = r1 + r2;
    int lo = c1 - r2; if (lo < 0) lo = 0;
    int hi = r1 < c1 ? r1 : c1;
    double konst = lf[r1] + lf[r2] + lf[c1] + lf[n - c1] - lf[n];
    double p_obs = exp(konst - lf[a] - lf[r1 - a] - lf[c1 - a] - lf[r2 - c1 + a]);
    double thr = p_obs * (1.0 + 1e-7);
    double sum = 0.0;
    for (int x = lo; x <= hi; x++) {
        double p = exp(konst - lf[x] - lf[r1 - x] - lf[c1 - x] - lf[r2 - c1 + x]);
        if (p <= thr) sum += p;
    }
    return sum > 1.0 ? 1.0 : sum;
}

/* How many hypergeometric terms a two-sided Fisher p-value NEEDS when a tail may stop at rel_cut x its own largest term
 * (the engine's cut, 1e-22 by default: below one ulp of the sum): the tables x with P(x) <= P_obs (1 + 1e-7) whose P(x) is
 * above rel_cut x the largest included term on their side of the mode.  This is bench.py's algorithmic work count for
 * the Fisher p-pass (terms x the cheapest instruction sequence per term); rel_cut = 0 counts every included table. */
long orc_fisher_terms_needed(int a, int b, int c, int d, const double *lf, double rel_cut) {
    int r1 = a + b, r2 = c + d, c1 = a + c, n = r1 + r2;
    int lo = c1 - r2; if (lo < 0) lo = 0;
    int hi = r1 < c1 ? r1 : c1;
    double konst = lf[r1] + lf[r2] + lf[c1] + lf[n - c1] - lf[n];
    double thr = (konst - lf[a] - lf[r1 - a] - lf[c1 - a] - lf[r2 - c1 + a]) + 9.9999995000000333e-08;   /* ln(P_obs (1 + 1e-7)) */
    long mode = (long)(((double)(r1 + 1) * (double)(c1 + 1)) / (double)(n + 2));
    if (mode < lo) mode = lo;
    if (mode > hi) mode = hi;
    double ln_cut = rel_cut > 0.0 ? log(rel_cut) : -INFINITY;
    long terms = 0;
    double top = -INFINITY;                               /* left of (and at) the mode: P grows with x, the largest included term is the last */
    for (long x = mode; x >= lo; x--) {
        double e = konst - lf[x] - lf[r1 - x] - lf[c1 - x] - lf[r2 - c1 + x];
        if (e > thr) continue;
        if (top == -INFINITY) top = e;
        if (e < top + ln_cut) break;
        terms++;
    }
    top = -INFINITY;
    for (long x = mode + 1; x <= hi; x++) {
        double e = konst - lf[x] - lf[r1 - x] - lf[c1 - x] - lf[r2 - c1 + x];
        if (e > thr) continue;
        if (top == -INFINITY) top = e;
        if (e < top + ln_cut) break;
        terms++;
    }
    return terms;
}

void orc_assoc_packed(const uint8_t *gt, size_t pitch, int n_variants, int n_samples,
                      const uint8_t *condition, const uint8_t *chrom_is_x,
                      int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2) {
    /* variants are independent (assoc.c:42-43 resets the counters per variant); OpenMP only
     * spreads them over cores when called from serial code (the baseline driver is already parallel) */
    #pragma omp parallel for schedule(static) if (n_variants > 4096 && !omp_in_parallel())
    for (int i = 0; i < n_variants; i++) {               /* assoc.c:38 */
        const uint8_t *row = gt + (size_t)i * pitch;
        int a1c = 0, a2c = 0, u1c = 0, u2c = 0;           /* assoc.c:42-43 */
        int is_x = chrom_is_x ? chrom_is_x[i] : 0;
        for (int j = 0; j < n_samples; j++) {             /* assoc.c:50 */
            int al1, al2;
            if (orc_decode(row[j], &al1, &al2) == ORC_ALLELES_OK)     /* assoc.c:53 */
                orc_assoc_count_individual(condition[j], is_x, al1, al2, &a1c, &a2c, &u1c, &u2c);
        }
        A1[i] = a1c; A2[i] = a2c; U1[i] = u1c; U2[i] = u2c;
    }
}

void orc_assoc_stats(int task, int n_variants, const int32_t *A1, const int32_t *A2,
                     const int32_t *U1, const int32_t *U2, const double *logfact,
                     double *odds, double *chisq, double *p) {
    for (int i = 0; i < n_variants; i++) {
        odds[i] = orc_assoc_odds_ratio(A1[i], A2[i], U1[i], U2[i]);
        if (task == ORC_TASK_CHISQ) {                     /* assoc.c:60-66 */
            double x = orc_assoc_basic_test(A1[i], U1[i], A2[i], U2[i]);
            chisq[i] = x;
            p[i] = orc_chisq_p_value(x);
        } else {                                          /* assoc.c:69-75 */
            p[i] = orc_fisher_two_sided(A1[i], A2[i], U1[i], U2[i], logfact);
        }
    }
}

void orc_assoc_text(const char *const *samples, int n_variants, int n_samples,
                    const char *const *formats, const uint8_t *condition,
                    const uint8_t *chrom_is_x,
                    int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2) {
    for (int i = 0; i < n_variants; i++) {
        int a1c = 0, a2c = 0, u1c = 0, u2c = 0;
        char *format = strdup(formats[i]);                /* assoc.c:45-47 */
        int gt_position = orc_get_field_position_in_format("GT", format);
        free(format);
        int is_x = chrom_is_x ? chrom_is_x[i] : 0;
        for (int j = 0; j < n_samples; j++) {
            char *sample_data = strdup(samples[(size_t)i * n_samples + j]);   /* assoc.c:52 */
            int al1, al2;
            if (orc_get_alleles(sample_data, gt_position, &al1, &al2) == ORC_ALLELES_OK)
                orc_assoc_count_individual(condition[j], is_x, al1, al2, &a1c, &a2c, &u1c, &u2c);
            free(sample_data);                            /* assoc.c:56 */
        }
        A1[i] = a1c; A2[i] = a2c; U1[i] = u1c; U2[i] = u2c;
    }
}

/* ------------------------------------------------------------------------
 * TDT
 * ---------------------------------------------------------------------- */

/* hpg-libs check_mendel(), called at tdt.c:161.  Body not in tree; restated
 * from the 41 assertions of test/test_checks_family.c:16-111 (which pin it for
 * alleles in {0,1}); for other allele values genotypes are classed by
 * zero-ness (both zero = hom-ref, both non-zero = hom-alt, else het), the same
 * classing tdt.c:175-213 uses.  Codes:
 *  1 00x00->het   2 11x11->het
 *  3 kid 00, mother 11, father not 11    4 kid 00, father 11, mother not 11
 *  5 kid 00, both 11
 *  6 kid 11, father 00, mother not 00    7 kid 11, mother 00, father not 00
 *  8 kid 11, both 00
 *  9 chr "X", male kid 11, mother 00    10 chr "X", male kid 00, mother 11
 * On chr "X" a male kid is checked against the mother only (codes 9/10, else 0). */
int orc_check_mendel(const char *chrom, int f1, int f2, int m1, int m2,
                     int c1, int c2, int child_sex) {
    int f_ref = !f1 && !f2, f_alt = f1 && f2;
    int m_ref = !m1 && !m2, m_alt = m1 && m2;
    int c_ref = !c1 && !c2, c_alt = c1 && c2;

    if (!strcmp(chrom, "X") && child_sex == ORC_SEX_MALE) {
        if (c_alt && m_ref) return 9;
        if (c_ref && m_alt) return 10;
        return 0;
    }
    if (!c_ref && !c_alt) {                 /* het kid */
        if (f_ref && m_ref) return 1;
        if (f_alt && m_alt) return 2;
        return 0;
    }
    if (c_ref) {
        if (f_alt && m_alt) return 5;
        if (m_alt) return 3;
        if (f_alt) return 4;
        return 0;
    }
    /* c_alt */
    if (f_ref && m_ref) return 8;
    if (f_ref) return 6;
    if (m_ref) return 7;
    return 0;
}

void orc_tdt_packed(const uint8_t *gt, size_t pitch, int n_variants,
                    const uint8_t *chrom_is_x,
                    int n_families, const int32_t *father_col, const int32_t *mother_col,
                    const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
                    int32_t *t1_out, int32_t *t2_out) {
    #pragma omp parallel for schedule(static) if (n_variants > 1024 && !omp_in_parallel())
    for (int v = 0; v < n_variants; v++) {                          /* tdt.c:41 */
        const uint8_t *row = gt + (size_t)v * pitch;
        const char *chrom = (chrom_is_x && chrom_is_x[v]) ? "X" : "1";
        int t1 = 0, t2 = 0;                                          /* tdt.c:51-52 */
        for (int f = 0; f < n_families; f++) {                       /* tdt.c:56 */
            if (father_col[f] < 0 || mother_col[f] < 0) continue;    /* tdt.c:77-95 */
            int fa1, fa2, ma1, ma2;
            if (orc_decode(row[father_col[f]], &fa1, &fa2) != ORC_ALLELES_OK ||
                orc_decode(row[mother_col[f]], &ma1, &ma2) != ORC_ALLELES_OK)
                continue;                                            /* tdt.c:103-108 */
            if (fa1 == fa2 && ma1 == ma2) continue;                  /* tdt.c:113-117 */
            if ((fa1 && !fa2) || (ma1 && !ma2)) continue;            /* tdt.c:119-123 */

            int trA = 0, unA = 0, trB = 0, unB = 0;                  /* tdt.c:128-132: family scope */
            for (int k = child_off[f]; k < child_off[f + 1]; k++) {  /* tdt.c:135 */
                int ca1, ca2;
                if (orc_decode(row[child_col[k]], &ca1, &ca2)) continue;          /* tdt.c:154 */
                if (orc_check_mendel(chrom, fa1, fa2, ma1, ma2, ca1, ca2, child_sex[k]))
                    continue;                                        /* tdt.c:161-166 */
                if (!ca1 && !ca2) {                                  /* tdt.c:175-181 */
                    if (((!fa1) && fa2) && ((!ma1) && ma2)) { trA = 1; unA = 2; trB = 1; unB = 2; }
                    else { trA = 1; unA = 2; }
                } else if ((!ca1) && ca2) {                          /* tdt.c:182-202 */
                    if (fa1 != fa2) {
                        if (ma1 != ma2) { trA = 1; trB = 2; unA = 2; unB = 1; }
                        else if (!ma1) { trA = 2; unA = 1; }
                        else { trA = 1; unA = 2; }
                    } else if (!fa1) { trA = 2; unA = 1; }
                    else { trA = 1; unA = 2; }
                } else {                                             /* tdt.c:203-213 */
                    if (((!fa1) && fa2) && ((!ma1) && ma2)) { trA = 2; unA = 1; trB = 2; unB = 1; }
                    else { trA = 2; unA = 1; }
                }
                if (trA == 1) t1++; else if (trA == 2) t2++;         /* tdt.c:235-239 */
                if (trB == 1) t1++; else if (trB == 2) t2++;
            }
            (void)unA; (void)unB;
        }
        t1_out[v] = t1; t2_out[v] = t2;
    }
}

void orc_mendel_counts(const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *chrom_is_x,
                       int n_trios, const int32_t *father_col, const int32_t *mother_col,
                       const int32_t *child_col, const uint8_t *child_sex,
                       int32_t *errors, int32_t *trio_errors) {
    for (int v = 0; v < n_variants; v++) {
        const uint8_t *row = gt + (size_t)v * pitch;
        const char *chrom = (chrom_is_x && chrom_is_x[v]) ? "X" : "1";
        int n = 0;
        for (int t = 0; t < n_trios; t++) {
            int f1, f2, m1, m2, c1, c2;
            if (orc_decode(row[father_col[t]], &f1, &f2) || orc_decode(row[mother_col[t]], &m1, &m2) ||
                orc_decode(row[child_col[t]], &c1, &c2)) continue;
            if (orc_check_mendel(chrom, f1, f2, m1, m2, c1, c2, child_sex[t])) { n++; if (trio_errors) trio_errors[t]++; }
        }
        if (errors) errors[v] = n;
    }
}

/* tdt.c:255-260 (chi-square, int arithmetic before the cast) and 279-292 */
void orc_tdt_stats(int n_variants, const int32_t *t1, const int32_t *t2,
                   double *odds, double *chisq, double *p) {
    for (int i = 0; i < n_variants; i++) {
        double x = -1;
        if (t1[i] + t2[i] > 0)
            x = ((double)((t1[i] - t2[i]) * (t1[i] - t2[i]))) / (t1[i] + t2[i]);
        double dt1 = t1[i], dt2 = t2[i];
        odds[i] = (dt2 == 0.0) ? NAN : (dt1 / dt2);
        chisq[i] = x;
        p[i] = orc_chisq_p_value(x);
    }
}

/* ------------------------------------------------------------------------
 * Variant stats / HWE  (hpg-libs; PARITY UNPINNED -- definitions of this repo)
 * ---------------------------------------------------------------------- */
void orc_hwe(int n_AA, int n_Aa, int n_aa, double *chi2, double *p) {
    int n = n_AA + n_Aa + n_aa;
    if (n == 0) { *chi2 = NAN; *p = NAN; return; }
    double pf = (2.0 * n_AA + n_Aa) / (2.0 * n);
    double qf = 1.0 - pf;
    double e_AA = pf * pf * n, e_Aa = 2.0 * pf * qf * n, e_aa = qf * qf * n;
    double x = 0.0;
    if (e_AA > 0.0) x += ((n_AA - e_AA) * (n_AA - e_AA)) / e_AA;
    if (e_Aa > 0.0) x += ((n_Aa - e_Aa) * (n_Aa - e_Aa)) / e_Aa;
    if (e_aa > 0.0) x += ((n_aa - e_aa) * (n_aa - e_aa)) / e_aa;
    *chi2 = x;
    *p = orc_chisq_p_value(x);
}

void orc_variant_stats(const uint8_t *row, int n_samples, int num_alleles,
                       orc_variant_stats_t *out) {
    memset(out, 0, sizeof *out);
    out->num_alleles = num_alleles;
    for (int j = 0; j < n_samples; j++) {
        int n1 = row[j] >> 4, n2 = row[j] & 0xF;
        if (n1 == 0xF) out->missing_alleles++; else if (n1 < num_alleles) out->alleles_count[n1]++;
        if (n2 == 0xF) out->missing_alleles++; else if (n2 < num_alleles) out->alleles_count[n2]++;
        if (n1 == 0xF || n2 == 0xF) out->missing_genotypes++;
        else if (n1 < num_alleles && n2 < num_alleles) out->genotypes_count[n1 * num_alleles + n2]++;
    }
    int total_alleles = 0, total_gt = 0;
    for (int i = 0; i < num_alleles; i++) total_alleles += out->alleles_count[i];
    for (int i = 0; i < num_alleles * num_alleles; i++) total_gt += out->genotypes_count[i];
    out->maf = 1.0; out->maf_allele = 0;
    for (int i = 0; i < num_alleles; i++) {
        double f = total_alleles ? (double)out->alleles_count[i] / total_alleles : 0.0;
        if (f < out->maf) { out->maf = f; out->maf_allele = i; }
    }
    out->mgf = 1.0; out->mgf_genotype = 0;
    for (int i = 0; i < num_alleles; i++) for (int k = i; k < num_alleles; k++) {
        int c = out->genotypes_count[i * num_alleles + k];
        if (k != i) c += out->genotypes_count[k * num_alleles + i];
        double f = total_gt ? (double)c / total_gt : 0.0;
        if (f < out->mgf) { out->mgf = f; out->mgf_genotype = i * num_alleles + k; }
    }
    if (num_alleles >= 2) {
        out->hw_n_AA = out->genotypes_count[0];
        out->hw_n_Aa = out->genotypes_count[1] + out->genotypes_count[num_alleles];
        out->hw_n_aa = out->genotypes_count[num_alleles + 1];
    } else {
        out->hw_n_AA = out->genotypes_count[0];
    }
    orc_hwe(out->hw_n_AA, out->hw_n_Aa, out->hw_n_aa, &out->hw_chi2, &out->hw_p);
}

void orc_sample_missing(const uint8_t *gt, size_t pitch, int n_variants, int n_samples, int32_t *missing) {
    for (int i = 0; i < n_variants; i++) {
        const uint8_t *row = gt + (size_t)i * pitch;
        for (int j = 0; j < n_samples; j++)
            if ((row[j] >> 4) == 0xF || (row[j] & 0xF) == 0xF) missing[j]++;
    }
}

/* ------------------------------------------------------------------------
 * Text staging oracle
 * ---------------------------------------------------------------------- */
int orc_tokenize(const char *text, size_t bytes, int n_samples, int strict, int max_lines,
                 uint8_t *gt, size_t pitch, uint8_t *is_x, int32_t *status) {
    int line = 0;
    size_t ls = 0;
    while (ls < bytes) {
        const char *nl = (const char *)memchr(text + ls, '\n', bytes - ls);
        size_t le = nl ? (size_t)(nl - text) : bytes;
        if (line < max_lines) {
            /* split the line on TABs */
            char *buf = (char *)malloc(le - ls + 1);
            memcpy(buf, text + ls, le - ls);
            buf[le - ls] = 0;
            int nf = 1;
            for (size_t i = 0; i < le - ls; i++) if (buf[i] == '\t') nf++;
            char **f = (char **)malloc(sizeof(char *) * (size_t)nf);
            int k = 0;
            f[k++] = buf;
            for (size_t i = 0; i < le - ls; i++) if (buf[i] == '\t') { buf[i] = 0; f[k++] = buf + i + 1; }
            uint8_t *row = gt + (size_t)line * pitch;
            int st = 0;
            if (is_x) is_x[line] = (uint8_t)orc_chrom_is_x(f[0], (int)strlen(f[0]));
            if (nf < 10) { st = 1; for (int j = 0; j < n_samples; j++) row[j] = 0xFF; }
            else {
                int gtpos = orc_get_field_position_in_format("GT", f[8]);
                if (gtpos < 0) { st = 2; for (int j = 0; j < n_samples; j++) row[j] = 0xFF; }
                else {
                    int have = nf - 9;
                    for (int j = 0; j < n_samples; j++)
                        row[j] = j < have ? orc_encode_sample(f[9 + j], gtpos, strict) : 0xFF;
                    if (have < n_samples) st = 3;
                }
            }
            if (status) status[line] = st;
            free(f); free(buf);
        }
        line++;
        ls = le + 1;
    }
    return line;
}

/* ------------------------------------------------------------------------
 * Synthetic cohort.  Integer-only per genotype so the device
 * generator reproduces it bit for bit.
 * ---------------------------------------------------------------------- */
uint64_t orc_splitmix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

void orc_synth_thresholds(uint64_t v, uint32_t thr[3]) {
    uint64_t h = orc_splitmix64(ORC_SYNTH_SEED ^ 0xA11E1EULL ^ v);
    double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    double q = 0.05 + 0.45 * u;
    const uint32_t t_miss = 167772u;                  /* floor(0.01 * 2^24) */
    double rest = (double)(16777216u - t_miss);
    double omq = 1.0 - q;
    double p00 = omq * omq;
    double p01 = 2.0 * q * omq;
    uint32_t t00 = t_miss + (uint32_t)floor(p00 * rest);
    uint32_t t01 = t00 + (uint32_t)floor(p01 * rest);
    thr[0] = t_miss; thr[1] = t00; thr[2] = t01;
}

uint8_t orc_synth_genotype(uint64_t v, uint64_t s, const uint32_t thr[3]) {
    uint32_t r = (uint32_t)(orc_splitmix64(ORC_SYNTH_SEED + v * 0x9E3779B97F4A7C15ULL + s) >> 40);
    if (r < thr[0]) return 0xFF;
    if (r < thr[1]) return 0x00;
    if (r < thr[2]) return 0x01;
    return 0x11;
}

void orc_synth_matrix(uint64_t v0, int n_variants, int n_samples, size_t pitch, uint8_t *gt) {
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < n_variants; i++) {
        uint32_t thr[3];
        orc_synth_thresholds(v0 + (uint64_t)i, thr);
        uint8_t *row = gt + (size_t)i * pitch;
        for (int s = 0; s < n_samples; s++) row[s] = orc_synth_genotype(v0 + (uint64_t)i, (uint64_t)s, thr);
        for (size_t s = (size_t)n_samples; s < pitch; s++) row[s] = 0xFF;
    }
}

/* ------------------------------------------------------------------------
 * CPU baseline drivers: the reference's worker structure -- N OpenMP workers
 * pulling batches of 200 variants (etc/hpg-variant/hpg-variant.conf:29-34;
 * assoc_runner.c:106-207), each batch run through the per-batch kernel
 * followed by the chi-square statistics.  Returns elapsed seconds.
 * ---------------------------------------------------------------------- */
static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

double orc_baseline_assoc_packed(const uint8_t *gt, size_t pitch, int n_variants, int n_samples,
                                 const uint8_t *condition, int n_threads, int *threads_used) {
    const int B = 200;
    int n_batches = (n_variants + B - 1) / B;
    int used = 1;
    volatile double sink = 0.0;
    double t0 = now_s();
    #pragma omp parallel num_threads(n_threads)
    {
        #pragma omp single
        {
#ifdef _OPENMP
            used = omp_get_num_threads();
#endif
        }
        int32_t A1[200], A2[200], U1[200], U2[200];
        double od[200], ch[200], pv[200];
        double local = 0.0;
        #pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < n_batches; b++) {
            int start = b * B, n = (start + B <= n_variants) ? B : n_variants - start;
            orc_assoc_packed(gt + (size_t)start * pitch, pitch, n, n_samples, condition, NULL, A1, A2, U1, U2);
            orc_assoc_stats(ORC_TASK_CHISQ, n, A1, A2, U1, U2, NULL, od, ch, pv);
            for (int i = 0; i < n; i++) local += pv[i];
        }
        #pragma omp atomic
        sink += local;
    }
    double t1 = now_s();
    if (threads_used) *threads_used = used;
    (void)sink;
    return t1 - t0;
}

/* Text-faithful: every batch is first rendered to "a/b" sample strings (not
 * timed), then scanned exactly as assoc.c:45-57 does (strdup + parse + free
 * per genotype).  Only the scan + statistics are timed.  task selects the
 * statistic of assoc.c:59-76 (chi-square, or Fisher on the caller's table). */
static void render_batch(uint64_t v0, int start, int n, int n_samples, char *pool, const char **samples,
                         const char **formats) {
    for (int i = 0; i < n; i++) {
        uint32_t thr[3];
        uint64_t v = v0 + (uint64_t)(start + i);
        orc_synth_thresholds(v, thr);
        formats[i] = "GT";
        for (int s = 0; s < n_samples; s++) {
            uint8_t g = orc_synth_genotype(v, (uint64_t)s, thr);
            char *dst = pool + ((size_t)i * n_samples + s) * 4;
            if (g == 0xFF) { dst[0] = '.'; dst[2] = '.'; }
            else { dst[0] = (char)('0' + (g >> 4)); dst[2] = (char)('0' + (g & 0xF)); }
            dst[1] = '/'; dst[3] = 0;
            samples[(size_t)i * n_samples + s] = dst;
        }
    }
}

double orc_baseline_assoc_text_task(uint64_t v0, int n_variants, int n_samples, const uint8_t *condition,
                                    int task, const double *logfact, int n_threads, int *threads_used) {
    const int B = 200;
    int n_batches = (n_variants + B - 1) / B;
    int used = 1;
    double total = 0.0;
    volatile double sink = 0.0;
    #pragma omp parallel num_threads(n_threads)
    {
        #pragma omp single
        {
#ifdef _OPENMP
            used = omp_get_num_threads();
#endif
        }
        char *pool = (char *)malloc((size_t)B * n_samples * 4);
        const char **samples = (const char **)malloc((size_t)B * n_samples * sizeof(char *));
        const char *formats[200];
        int32_t A1[200], A2[200], U1[200], U2[200];
        double od[200], ch[200], pv[200];
        double mine = 0.0, local = 0.0;
        #pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < n_batches; b++) {
            int start = b * B, n = (start + B <= n_variants) ? B : n_variants - start;
            render_batch(v0, start, n, n_samples, pool, samples, formats);
            double t0 = now_s();
            orc_assoc_text(samples, n, n_samples, formats, condition, NULL, A1, A2, U1, U2);
            orc_assoc_stats(task, n, A1, A2, U1, U2, logfact, od, ch, pv);
            mine += now_s() - t0;
            for (int i = 0; i < n; i++) local += pv[i];
        }
        free(pool); free((void *)samples);
        #pragma omp critical
        { if (mine > total) total = mine; sink += local; }
    }
    if (threads_used) *threads_used = used;
    (void)sink;
    return total;   /* max over workers of their scan time = parallel wall time of the scan */
}

double orc_baseline_assoc_text(uint64_t v0, int n_variants, int n_samples,
                               const uint8_t *condition, int n_threads, int *threads_used) {
    return orc_baseline_assoc_text_task(v0, n_variants, n_samples, condition, ORC_TASK_CHISQ, NULL, n_threads, threads_used);
}

/* tdt.c:41-271 on sample strings: per family strdup + get_alleles of both parents (tdt.c:97-108), per counted
 * child strdup + get_alleles (tdt.c:150-157) and a strndup of the chromosome for check_mendel (tdt.c:160).  The
 * sample_ids hash lookups of tdt.c:83-95,146-148 are NOT repeated per variant here (columns come resolved):
 * the port is on the fast side of the reference. */
void orc_tdt_text(const char *const *samples, int n_variants, int n_samples, const char *const *formats,
                  const uint8_t *chrom_is_x, int n_families, const int32_t *father_col, const int32_t *mother_col,
                  const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
                  int32_t *t1_out, int32_t *t2_out) {
    for (int v = 0; v < n_variants; v++) {
        const char *const *row = samples + (size_t)v * n_samples;
        char *format = strdup(formats[v]);                               /* tdt.c:46-48 */
        int gt_position = orc_get_field_position_in_format("GT", format);
        free(format);
        const char *chrom = (chrom_is_x && chrom_is_x[v]) ? "X" : "1";
        int t1 = 0, t2 = 0;
        for (int f = 0; f < n_families; f++) {
            if (father_col[f] < 0 || mother_col[f] < 0) continue;
            char *fs = strdup(row[father_col[f]]), *ms = strdup(row[mother_col[f]]);   /* tdt.c:97-98 */
            int fa1, fa2, ma1, ma2;
            if (orc_get_alleles(fs, gt_position, &fa1, &fa2) != ORC_ALLELES_OK ||
                orc_get_alleles(ms, gt_position, &ma1, &ma2) != ORC_ALLELES_OK) { free(fs); free(ms); continue; }
            if ((fa1 == fa2 && ma1 == ma2) || (fa1 && !fa2) || (ma1 && !ma2)) { free(fs); free(ms); continue; }
            int trA = 0, trB = 0;
            for (int k = child_off[f]; k < child_off[f + 1]; k++) {
                char *cs = strdup(row[child_col[k]]);                    /* tdt.c:150 */
                int ca1, ca2;
                if (orc_get_alleles(cs, gt_position, &ca1, &ca2)) { free(cs); continue; }
                char *aux = strdup(chrom);                               /* tdt.c:160 */
                int bad = orc_check_mendel(aux, fa1, fa2, ma1, ma2, ca1, ca2, child_sex[k]);
                free(aux);
                if (bad) { free(cs); continue; }
                if (!ca1 && !ca2) {
                    if (((!fa1) && fa2) && ((!ma1) && ma2)) { trA = 1; trB = 1; } else { trA = 1; }
                } else if ((!ca1) && ca2) {
                    if (fa1 != fa2) {
                        if (ma1 != ma2) { trA = 1; trB = 2; }
                        else if (!ma1) { trA = 2; }
                        else { trA = 1; }
                    } else if (!fa1) { trA = 2; }
                    else { trA = 1; }
                } else {
                    if (((!fa1) && fa2) && ((!ma1) && ma2)) { trA = 2; trB = 2; } else { trA = 2; }
                }
                if (trA == 1) t1++; else if (trA == 2) t2++;
                if (trB == 1) t1++; else if (trB == 2) t2++;
                free(cs);
            }
            free(fs); free(ms);
        }
        t1_out[v] = t1; t2_out[v] = t2;
    }
}

double orc_baseline_tdt_text(uint64_t v0, int n_variants, int n_samples, int n_families,
                             const int32_t *father_col, const int32_t *mother_col, const int32_t *child_off,
                             const int32_t *child_col, const uint8_t *child_sex, int n_threads, int *threads_used) {
    const int B = 200;
    int n_batches = (n_variants + B - 1) / B;
    int used = 1;
    double total = 0.0;
    volatile double sink = 0.0;
    #pragma omp parallel num_threads(n_threads)
    {
        #pragma omp single
        {
#ifdef _OPENMP
            used = omp_get_num_threads();
#endif
        }
        char *pool = (char *)malloc((size_t)B * n_samples * 4);
        const char **samples = (const char **)malloc((size_t)B * n_samples * sizeof(char *));
        const char *formats[200];
        int32_t t1[200], t2[200];
        double od[200], ch[200], pv[200];
        double mine = 0.0, local = 0.0;
        #pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < n_batches; b++) {
            int start = b * B, n = (start + B <= n_variants) ? B : n_variants - start;
            render_batch(v0, start, n, n_samples, pool, samples, formats);
            double t0 = now_s();
            orc_tdt_text(samples, n, n_samples, formats, NULL, n_families, father_col, mother_col, child_off, child_col,
                         child_sex, t1, t2);
            orc_tdt_stats(n, t1, t2, od, ch, pv);
            mine += now_s() - t0;
            for (int i = 0; i < n; i++) local += pv[i];
        }
        free(pool); free((void *)samples);
        #pragma omp critical
        { if (mine > total) total = mine; sink += local; }
    }
    if (threads_used) *threads_used = used;
    (void)sink;
    return total;
}

/* the stats tool's per-variant pass (get_variants_stats, call site stats_runner.c:194-195; body in hpg-libs, this
 * repo's definition orc_variant_stats) on sample strings: per genotype a strdup + get_alleles + free as the assoc
 * loop does, then the counters and the Hardy-Weinberg test. */
double orc_baseline_stats_text(uint64_t v0, int n_variants, int n_samples, int n_threads, int *threads_used) {
    const int B = 200;
    int n_batches = (n_variants + B - 1) / B;
    int used = 1;
    double total = 0.0;
    volatile double sink = 0.0;
    #pragma omp parallel num_threads(n_threads)
    {
        #pragma omp single
        {
#ifdef _OPENMP
            used = omp_get_num_threads();
#endif
        }
        char *pool = (char *)malloc((size_t)B * n_samples * 4);
        const char **samples = (const char **)malloc((size_t)B * n_samples * sizeof(char *));
        uint8_t *codes = (uint8_t *)malloc((size_t)n_samples);
        const char *formats[200];
        double mine = 0.0, local = 0.0;
        #pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < n_batches; b++) {
            int start = b * B, n = (start + B <= n_variants) ? B : n_variants - start;
            render_batch(v0, start, n, n_samples, pool, samples, formats);
            double t0 = now_s();
            for (int i = 0; i < n; i++) {
                char *format = strdup(formats[i]);
                int gt_position = orc_get_field_position_in_format("GT", format);
                free(format);
                for (int s = 0; s < n_samples; s++) {
                    char *sd = strdup(samples[(size_t)i * n_samples + s]);
                    int a1 = -1, a2 = -1;
                    int st = orc_get_alleles(sd, gt_position, &a1, &a2);
                    codes[s] = orc_encode_alleles(st, a1, a2, 0);
                    free(sd);
                }
                orc_variant_stats_t vs;
                orc_variant_stats(codes, n_samples, 2, &vs);
                local += vs.hw_p;
            }
            mine += now_s() - t0;
        }
        free(pool); free((void *)samples); free(codes);
        #pragma omp critical
        { if (mine > total) total = mine; sink += local; }
    }
    if (threads_used) *threads_used = used;
    (void)sink;
    return total;
}

/* OpenMP team size of the oracle's parallel loops (a cgroup CPU quota can be far below the visible CPUs) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

