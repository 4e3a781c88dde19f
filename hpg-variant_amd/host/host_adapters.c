/* host_adapters.c -- the reference's per-batch functions re-hosted on the engine: assoc_test (assoc.h:137), tdt_test (tdt.h:119), get_variants_stats / get_sample_stats (stats_runner.c:194-198).
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* where a per-batch adapter call spends its time (hpgv_host.h "hpgv_host_adapter_times"): nanoseconds, added atomically */
typedef struct { int on; uint64_t t[4]; } adapter_clock_t;
static int g_adapter_profile = 0;
static uint64_t g_adapter_ns[4];
static long g_adapter_calls;
void hpgv_host_adapter_profile(int on) { __atomic_store_n(&g_adapter_profile, on, __ATOMIC_RELAXED); }
void hpgv_host_adapter_times(double *seconds4, long *calls, int reset) {
    for (int k = 0; k < 4; k++) {
        if (seconds4) seconds4[k] = 1e-9 * (double)__atomic_load_n(&g_adapter_ns[k], __ATOMIC_RELAXED);
        if (reset) __atomic_store_n(&g_adapter_ns[k], 0, __ATOMIC_RELAXED);
    }
    if (calls) *calls = __atomic_load_n(&g_adapter_calls, __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(&g_adapter_calls, 0, __ATOMIC_RELAXED);
}
static inline uint64_t mono_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
static inline void adapter_clock_start(adapter_clock_t *c) { memset(c, 0, sizeof *c); c->on = __atomic_load_n(&g_adapter_profile, __ATOMIC_RELAXED); if (c->on) c->t[0] = mono_ns(); }
static inline void adapter_clock_mark(adapter_clock_t *c, int k) { if (c->on) c->t[k] = mono_ns(); }     /* k = 1 staged, 2 engine done */
static inline void adapter_clock_stop(adapter_clock_t *c) {
    if (!c->on) return;
    c->t[3] = mono_ns();
    __atomic_fetch_add(&g_adapter_ns[0], c->t[1] - c->t[0], __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_adapter_ns[1], c->t[2] - c->t[1], __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_adapter_ns[2], c->t[3] - c->t[2], __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_adapter_ns[3], c->t[3] - c->t[0], __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_adapter_calls, 1, __ATOMIC_RELAXED);
}

/* ------------------------------------------------------------------------ */
/* assoc_test                                                                 */
/* ------------------------------------------------------------------------ */

static int assoc_prepare(enum ASSOC_task task, individual_t **samples, int num_samples, const void *opt_input) {
    uint64_t h = 1469598103934665603ULL;
    for (int j = 0; j < num_samples; j++) {
        /* assert(individual) of assoc.c:92: a VCF sample absent from the PED is fatal there */
        if (!samples[j]) { snprintf(g_err, sizeof g_err, "sample %d has no individual (assoc.c:92)", j); return HPGV_ERR_INVALID; }
        h = (h ^ (uint64_t)samples[j]->condition) * 1099511628211ULL;
    }
    /* called and left with the READ lock held.  Installing needs the write lock; rwlocks do not upgrade, so the read lock is
     * dropped, the write lock taken, and after coming back to the read lock the keys are checked AGAIN: a thread with another
     * cohort may have installed its own in the gap.  The loop ends with the wanted layout installed under our read lock. */
    for (;;) {
        int same_cohort = g_assoc_key.set && g_assoc_key.num_samples == num_samples && g_assoc_key.cond_hash == h && g_assoc_key.cond;
        /* the 64-bit hash only says "probably": the vector decides (a collision would scan with another cohort's layout) */
        for (int j = 0; same_cohort && j < num_samples; j++) {
            const enum Condition c = samples[j]->condition;
            same_cohort = g_assoc_key.cond[j] == ((c == AFFECTED) ? HPGV_COND_AFFECTED : (c == UNAFFECTED) ? HPGV_COND_UNAFFECTED : HPGV_COND_OTHER);
        }
        const int need_cohort = !same_cohort;
        const int need_lf = (task == FISHER) && !(g_lf_key.table == opt_input && g_lf_key.n == num_samples * 10);
        if (!need_cohort && !need_lf) return HPGV_OK;
        pthread_rwlock_unlock(&g_cohort_lock);
        pthread_rwlock_wrlock(&g_cohort_lock);
        int rc = HPGV_OK;
        if (need_cohort) {                                            /* (re-checked from scratch by the next turn of the loop, under the read lock) */
            uint8_t *cond = (uint8_t *)malloc((size_t)(num_samples > 0 ? num_samples : 1));
            if (!cond) { snprintf(g_err, sizeof g_err, "out of memory"); rc = HPGV_ERR_NOMEM; }
            else {
                for (int j = 0; j < num_samples; j++) {
                    enum Condition c = samples[j]->condition;                 /* assoc.c:95,101 */
                    cond[j] = (c == AFFECTED) ? HPGV_COND_AFFECTED : (c == UNAFFECTED) ? HPGV_COND_UNAFFECTED : HPGV_COND_OTHER;
                }
                rc = hpgv_set_cohort(g_ctx, cond, num_samples);
                if (rc == HPGV_OK) {
                    g_assoc_key.samples = samples; g_assoc_key.num_samples = num_samples;
                    g_assoc_key.cond_hash = h; g_assoc_key.set = 1;
                    free(g_assoc_key.cond); g_assoc_key.cond = cond;
                } else { free(cond); host_fail("hpgv_set_cohort", rc); }
            }
        }
        if (rc == HPGV_OK && task == FISHER && !(g_lf_key.table == opt_input && g_lf_key.n == num_samples * 10)) {
            if (!opt_input) { snprintf(g_err, sizeof g_err, "FISHER needs the log-factorial table (opt_input)"); rc = HPGV_ERR_INVALID; }
            else {
                /* assoc_runner.c:164-166: the table has num_samples * 10 entries */
                rc = hpgv_set_logfact(g_ctx, (const double *)opt_input, (size_t)num_samples * 10);
                if (rc == HPGV_OK) { g_lf_key.table = opt_input; g_lf_key.n = num_samples * 10; }
                else host_fail("hpgv_set_logfact", rc);
            }
        }
        pthread_rwlock_unlock(&g_cohort_lock);
        pthread_rwlock_rdlock(&g_cohort_lock);
        if (rc != HPGV_OK) return rc;
    }
}

static int assoc_test_impl(enum ASSOC_task test_type, vcf_record_t **variants, int num_variants,
                           individual_t **samples, int num_samples, const void *opt_input,
                           list_t *output_list) {
    if (test_type != CHI_SQUARE && test_type != FISHER) return HPGV_OK;      /* assoc.c:60,69: NONE emits nothing */
    if (num_variants <= 0) return HPGV_OK;
    int rc = ensure_engine();
    if (rc) return rc;
    int tid = thread_id();                                                   /* assoc.c:25 */
    size_t n = (size_t)num_variants, ns = (size_t)(num_samples > 0 ? num_samples : 0);
    size_t pitch = ns ? ns : 1;
    int gt_slot;
    uint8_t *gt = stage_get(n * pitch + n, &gt_slot);
    int32_t *cnt = (int32_t *)malloc(n * 4 * sizeof(int32_t));
    double *st = (double *)malloc(n * 3 * sizeof(double));
    if (!gt || !cnt || !st) { if (gt) stage_put(gt, gt_slot); free(cnt); free(st); snprintf(g_err, sizeof g_err, "out of memory"); return HPGV_ERR_NOMEM; }
    uint8_t *is_x = gt + n * pitch;
    adapter_clock_t clk;
    adapter_clock_start(&clk);
    hpgv_host_stage_records(variants, num_variants, num_samples, 1, gt, is_x);   /* assoc.c:45-57 */
    adapter_clock_mark(&clk, 1);

    pthread_rwlock_rdlock(&g_cohort_lock);
    rc = assoc_prepare(test_type, samples, num_samples, opt_input);
    if (rc == HPGV_OK) {
        rc = hpgv_assoc(g_ctx, (int)test_type, gt, pitch, num_variants, is_x, cnt, cnt + n, cnt + 2 * n, cnt + 3 * n,
                        st, test_type == CHI_SQUARE ? st + n : NULL, st + 2 * n);
        if (rc != HPGV_OK) host_fail("hpgv_assoc", rc);
    }
    pthread_rwlock_unlock(&g_cohort_lock);
    adapter_clock_mark(&clk, 2);

    if (rc == HPGV_OK) {
        item_chain_t chain = { NULL, NULL, 0 };
        for (size_t i = 0; i < n; i++) {
            vcf_record_t *record = variants[i];
            void *result;
            if (test_type == CHI_SQUARE) {                                   /* assoc.c:61-66 */
                assoc_basic_result_t *r = (assoc_basic_result_t *)malloc(sizeof *r);
                r->chromosome = dupn(record->chromosome, record->chromosome_len);
                r->position = record->position;
                r->id = dupn(record->id, record->id_len);
                r->reference = dupn(record->reference, record->reference_len);
                r->alternate = dupn(record->alternate, record->alternate_len);
                r->affected1 = cnt[i]; r->affected2 = cnt[n + i];
                r->unaffected1 = cnt[2 * n + i]; r->unaffected2 = cnt[3 * n + i];
                r->odds_ratio = st[i]; r->chi_square = st[n + i]; r->p_value = st[2 * n + i];
                result = r;
            } else {                                                         /* assoc.c:70-75 */
                assoc_fisher_result_t *r = (assoc_fisher_result_t *)malloc(sizeof *r);
                r->chromosome = dupn(record->chromosome, record->chromosome_len);
                r->position = record->position;
                r->id = dupn(record->id, record->id_len);
                r->reference = dupn(record->reference, record->reference_len);
                r->alternate = dupn(record->alternate, record->alternate_len);
                r->affected1 = cnt[i]; r->affected2 = cnt[n + i];
                r->unaffected1 = cnt[2 * n + i]; r->unaffected2 = cnt[3 * n + i];
                r->odds_ratio = st[i]; r->p_value = st[2 * n + i];
                result = r;
            }
            chain_add(&chain, list_item_new(tid, 0, result));                /* assoc.c:67-68,76-77 */
        }
        list_insert_chain(chain.first, chain.last, chain.n, output_list);
    }
    stage_put(gt, gt_slot); free(cnt); free(st);
    adapter_clock_stop(&clk);
    return rc;
}

void assoc_test(enum ASSOC_task test_type, vcf_record_t **variants, int num_variants,
                individual_t **samples, int num_samples, const void *opt_input, list_t *output_list) {
    int rc = assoc_test_impl(test_type, variants, num_variants, samples, num_samples, opt_input, output_list);
    if (rc != HPGV_OK) {
        /* the reference function is void; its failures are LOG_FATAL (process exit) */
        fprintf(stderr, "FATAL: assoc_test: %s\n", g_err);
        exit(1);
    }
}

/* ------------------------------------------------------------------------ */
/* tdt_test                                                                   */
/* ------------------------------------------------------------------------ */

/* Builds the CSR pedigree description exactly the way tdt.c walks the families
 * (father/mother choice tdt.c:62-73, sample lookup :83-95, counted children
 * :135-148) and installs it when it differs from what the engine holds. */
static int tdt_prepare(family_t **families, int num_families, sample_ids_t *sample_ids, int num_columns) {
    size_t total_children = 0;
    for (int f = 0; f < num_families; f++) total_children += families[f]->members->size;
    size_t nf = (size_t)(num_families > 0 ? num_families : 1);
    int32_t *fcol = (int32_t *)malloc(nf * sizeof(int32_t)), *mcol = (int32_t *)malloc(nf * sizeof(int32_t));
    int32_t *coff = (int32_t *)malloc((nf + 1) * sizeof(int32_t));
    int32_t *ccol = (int32_t *)malloc((total_children + 1) * sizeof(int32_t));
    uint8_t *csex = (uint8_t *)malloc(total_children + 1);
    int rc = HPGV_OK;
    if (!fcol || !mcol || !coff || !ccol || !csex) {
        snprintf(g_err, sizeof g_err, "out of memory");
        rc = HPGV_ERR_NOMEM;
    } else {
        int nchild = 0;
        uint64_t h = 1469598103934665603ULL;
        coff[0] = 0;
        for (int f = 0; f < num_families; f++) {
            family_t *family = families[f];
            individual_t *father = NULL, *mother = NULL;
            for (size_t i = 0; i < family->founders->size; i++) {          /* tdt.c:62-73 */
                if (father && mother) break;
                individual_t *indiv = (individual_t *)family->founders->items[i];
                if (indiv->sex == MALE) father = indiv;
                else if (indiv->sex == FEMALE) mother = indiv;
            }
            fcol[f] = mcol[f] = -1;
            if (father && mother) {                                        /* tdt.c:77-95 */
                int fp = sample_ids_get(sample_ids, father->id), mp = sample_ids_get(sample_ids, mother->id);
                if (fp >= 0 && mp >= 0 && fp < num_columns && mp < num_columns) { fcol[f] = fp; mcol[f] = mp; }
            }
            if (fcol[f] >= 0) {
                for (size_t i = 0; i < family->members->size; i++) {       /* tdt.c:135-148 */
                    individual_t *child = (individual_t *)family->members->items[i];
                    if (!child->father || !child->mother) continue;
                    if (child->condition != AFFECTED) continue;
                    int cp = sample_ids_get(sample_ids, child->id);
                    if (cp < 0 || cp >= num_columns) continue;
                    ccol[nchild] = cp;
                    csex[nchild] = (child->sex == MALE) ? HPGV_SEX_MALE : (child->sex == FEMALE) ? HPGV_SEX_FEMALE : HPGV_SEX_UNKNOWN;
                    h = (h ^ (uint64_t)(uint32_t)cp ^ ((uint64_t)csex[nchild] << 40)) * 1099511628211ULL;
                    nchild++;
                }
            }
            coff[f + 1] = nchild;
            h = (h ^ (uint64_t)(uint32_t)fcol[f]) * 1099511628211ULL;
            h = (h ^ (uint64_t)(uint32_t)mcol[f]) * 1099511628211ULL;
            h = (h ^ (uint64_t)(uint32_t)nchild) * 1099511628211ULL;
        }
        /* installed under the write lock, re-checked after coming back to the read lock (see assoc_prepare).  The 64-bit hash
         * only says "probably": the arrays decide (a collision would count with another pedigree's columns). */
        const size_t nfz = (size_t)(num_families > 0 ? num_families : 0), ncz = (size_t)nchild;
#define TDT_SAME() (g_tdt_key.set && g_tdt_key.num_families == num_families && g_tdt_key.num_columns == num_columns &&            \
                    g_tdt_key.hash == h && g_tdt_key.n_children == ncz && g_tdt_key.csr &&                                          \
                    !memcmp(g_tdt_key.csr, fcol, nfz * 4) && !memcmp(g_tdt_key.csr + nfz, mcol, nfz * 4) &&                        \
                    !memcmp(g_tdt_key.csr + 2 * nfz, coff, (nfz + 1) * 4) && !memcmp(g_tdt_key.csr + 3 * nfz + 1, ccol, ncz * 4) && \
                    !memcmp(g_tdt_key.csex, csex, ncz))
        while (rc == HPGV_OK && !TDT_SAME()) {
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_wrlock(&g_cohort_lock);
            if (!TDT_SAME()) {
                int32_t *keep = (int32_t *)malloc((3 * nfz + 1 + ncz + 1) * sizeof(int32_t));
                uint8_t *keep_sex = (uint8_t *)malloc(ncz + 1);
                if (!keep || !keep_sex) { snprintf(g_err, sizeof g_err, "out of memory"); rc = HPGV_ERR_NOMEM; }
                else {
                    rc = hpgv_set_families(g_ctx, num_columns, num_families, fcol, mcol, coff, ccol, csex);
                    if (rc != HPGV_OK) host_fail("hpgv_set_families", rc);
                }
                if (rc == HPGV_OK) {
                    memcpy(keep, fcol, nfz * 4); memcpy(keep + nfz, mcol, nfz * 4);
                    memcpy(keep + 2 * nfz, coff, (nfz + 1) * 4); memcpy(keep + 3 * nfz + 1, ccol, ncz * 4);
                    memcpy(keep_sex, csex, ncz);
                    free(g_tdt_key.csr); free(g_tdt_key.csex);
                    g_tdt_key.csr = keep; g_tdt_key.csex = keep_sex; g_tdt_key.n_children = ncz;
                    g_tdt_key.num_families = num_families; g_tdt_key.num_columns = num_columns;
                    g_tdt_key.hash = h; g_tdt_key.set = 1;
                } else { free(keep); free(keep_sex); }
            }
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_rdlock(&g_cohort_lock);
        }
#undef TDT_SAME
    }
    free(fcol); free(mcol); free(coff); free(ccol); free(csex);
    return rc;
}

int tdt_test(vcf_record_t **variants, int num_variants, family_t **families, int num_families,
             sample_ids_t *sample_ids, list_t *output_list) {
    if (num_variants <= 0) return 0;
    int rc = ensure_engine();
    if (rc) return rc;
    int tid = thread_id();                                                     /* tdt.c:27 */
    int num_columns = (int)variants[0]->samples->size;
    size_t n = (size_t)num_variants, pitch = (size_t)(num_columns > 0 ? num_columns : 1);
    int gt_slot;
    uint8_t *gt = stage_get(n * pitch + n, &gt_slot);
    int32_t *tu = (int32_t *)malloc(n * 2 * sizeof(int32_t));
    double *st = (double *)malloc(n * 3 * sizeof(double));
    if (!gt || !tu || !st) { if (gt) stage_put(gt, gt_slot); free(tu); free(st); snprintf(g_err, sizeof g_err, "out of memory"); return HPGV_ERR_NOMEM; }
    uint8_t *is_x = gt + n * pitch;
    adapter_clock_t clk;
    adapter_clock_start(&clk);
    hpgv_host_stage_records(variants, num_variants, num_columns, 1, gt, is_x);
    adapter_clock_mark(&clk, 1);

    pthread_rwlock_rdlock(&g_cohort_lock);
    rc = tdt_prepare(families, num_families, sample_ids, num_columns);
    if (rc == HPGV_OK) {
        rc = hpgv_tdt(g_ctx, gt, pitch, num_variants, is_x, tu, tu + n, st, st + n, st + 2 * n);
        if (rc != HPGV_OK) host_fail("hpgv_tdt", rc);
    }
    pthread_rwlock_unlock(&g_cohort_lock);
    adapter_clock_mark(&clk, 2);

    if (rc == HPGV_OK) {
        item_chain_t chain = { NULL, NULL, 0 };
        for (size_t i = 0; i < n; i++) {                                       /* tdt.c:262-268 */
            vcf_record_t *record = variants[i];
            tdt_result_t *r = (tdt_result_t *)malloc(sizeof *r);
            r->chromosome = dupn(record->chromosome, record->chromosome_len);
            r->position = record->position;
            r->id = dupn(record->id, record->id_len);
            r->reference = dupn(record->reference, record->reference_len);
            r->alternate = dupn(record->alternate, record->alternate_len);
            r->t1 = tu[i]; r->t2 = tu[n + i];
            r->odds_ratio = st[i]; r->chi_square = st[n + i]; r->p_value = st[2 * n + i];
            chain_add(&chain, list_item_new(tid, 0, r));
        }
        list_insert_chain(chain.first, chain.last, chain.n, output_list);
    }
    stage_put(gt, gt_slot); free(tu); free(st);
    adapter_clock_stop(&clk);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* get_variants_stats                                                         */
/* ------------------------------------------------------------------------ */

static int stats_prepare(int num_samples) {
    int rc = HPGV_OK;
    while (rc == HPGV_OK && !(g_stats_key.set && g_stats_key.num_samples == num_samples)) {    /* re-checked under the read lock */
        pthread_rwlock_unlock(&g_cohort_lock);
        pthread_rwlock_wrlock(&g_cohort_lock);
        if (!(g_stats_key.set && g_stats_key.num_samples == num_samples)) {
            rc = hpgv_set_stats_cohort(g_ctx, num_samples);
            if (rc == HPGV_OK) { g_stats_key.set = 1; g_stats_key.num_samples = num_samples; }
            else host_fail("hpgv_set_stats_cohort", rc);
        }
        pthread_rwlock_unlock(&g_cohort_lock);
        pthread_rwlock_rdlock(&g_cohort_lock);
    }
    return rc;
}

/* 1 + number of comma-separated ALT alleles ("." = none) */
static int count_alleles(const char *alt, int len) {
    if (len <= 0 || (len == 1 && alt[0] == '.')) return 1;
    int n = 2;
    for (int i = 0; i < len; i++) if (alt[i] == ',') n++;
    return n;
}

int get_variants_stats(vcf_record_t **variants, int num_variants, individual_t **individuals,
                       sample_ids_t *sample_ids, int num_variables, list_t *output_list,
                       file_stats_t *file_stats) {
    (void)sample_ids;
    if (num_variants <= 0) return 0;
    int rc = ensure_engine();
    if (rc) return rc;
    int tid = thread_id();
    int num_samples = (int)variants[0]->samples->size;
    size_t n = (size_t)num_variants, pitch = (size_t)(num_samples > 0 ? num_samples : 1);
    int gt_slot;
    uint8_t *gt = stage_get(n * pitch, &gt_slot);
    int32_t *c8 = (int32_t *)malloc(n * 8 * sizeof(int32_t));
    double *hw = (double *)malloc(n * 2 * sizeof(double));
    int32_t *midx = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *mtab = (int32_t *)malloc(n * 256 * sizeof(int32_t));
    if (!gt || !c8 || !hw || !midx || !mtab) {
        if (gt) stage_put(gt, gt_slot);
        free(c8); free(hw); free(midx); free(mtab);
        snprintf(g_err, sizeof g_err, "out of memory");
        return HPGV_ERR_NOMEM;
    }
    adapter_clock_t clk;
    adapter_clock_start(&clk);
    hpgv_host_stage_records(variants, num_variants, num_samples, 0, gt, NULL);
    adapter_clock_mark(&clk, 1);
    int n_multi = num_variants;

    /* per-phenotype counters (one report per phenotype, stats_runner.c:300-303,319-323): the group of VCF
     * column j is the id of its individual's PED variable, 0 .. num_variables-1 */
    const int ng = (individuals && num_variables > 0 && num_samples > 0) ? num_variables : 0;
    int32_t *group = NULL, *gc8 = NULL;
    double *ghw = NULL;
    uint64_t gh = 1469598103934665603ULL;
    if (ng) {
        group = (int32_t *)malloc((size_t)num_samples * sizeof(int32_t));
        gc8 = (int32_t *)malloc((size_t)ng * n * 8 * sizeof(int32_t));
        ghw = (double *)malloc((size_t)ng * n * 2 * sizeof(double));
        if (!group || !gc8 || !ghw) {
            free(group); free(gc8); free(ghw); stage_put(gt, gt_slot); free(c8); free(hw); free(midx); free(mtab);
            snprintf(g_err, sizeof g_err, "out of memory");
            return HPGV_ERR_NOMEM;
        }
        for (int j = 0; j < num_samples; j++) {
            const individual_t *ind = individuals[j];
            const int k = ind ? (int)ind->variable : -1;
            group[j] = (ind && ind->variable >= 0 && k < ng) ? k : -1;
            gh = (gh ^ (uint64_t)(uint32_t)group[j]) * 1099511628211ULL;
        }
    }

    pthread_rwlock_rdlock(&g_cohort_lock);
    rc = stats_prepare(num_samples);
    if (rc == HPGV_OK) {
        rc = hpgv_stats_ex(g_ctx, gt, pitch, num_variants, c8, hw, hw + n, NULL, midx, mtab, &n_multi);
        if (rc != HPGV_OK) host_fail("hpgv_stats_ex", rc);
    }
    if (rc == HPGV_OK && ng) {
        while (rc == HPGV_OK && !(g_group_key.set && g_group_key.num_samples == num_samples && g_group_key.n_groups == ng && g_group_key.hash == gh)) {
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_wrlock(&g_cohort_lock);
            if (!(g_group_key.set && g_group_key.num_samples == num_samples && g_group_key.n_groups == ng && g_group_key.hash == gh)) {
                rc = hpgv_set_stats_groups(g_ctx, group, num_samples, ng);
                if (rc == HPGV_OK) { g_group_key.set = 1; g_group_key.num_samples = num_samples; g_group_key.n_groups = ng; g_group_key.hash = gh; }
                else host_fail("hpgv_set_stats_groups", rc);
            }
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_rdlock(&g_cohort_lock);
        }
        if (rc == HPGV_OK) {
            rc = hpgv_stats_groups(g_ctx, gt, pitch, num_variants, gc8, ghw, ghw + (size_t)ng * n);
            if (rc != HPGV_OK) host_fail("hpgv_stats_groups", rc);
        }
    }
    pthread_rwlock_unlock(&g_cohort_lock);
    adapter_clock_mark(&clk, 2);

    if (rc == HPGV_OK) {
        item_chain_t chain = { NULL, NULL, 0 };
        int next_multi = 0;
        for (size_t i = 0; i < n; i++) {
            vcf_record_t *record = variants[i];
            const int32_t *c = c8 + 8 * i;
            variant_stats_t *s = (variant_stats_t *)calloc(1, sizeof *s);
            s->chromosome = dupn(record->chromosome, record->chromosome_len);
            s->position = record->position;
            s->ref_allele = dupn(record->reference, record->reference_len);
            s->alt_alleles = dupn(record->alternate, record->alternate_len);
            int na = count_alleles(record->alternate, record->alternate_len);
            const int32_t *tab = NULL;
            if (next_multi < n_multi && midx[next_multi] == (int32_t)i) {   /* multi-allelic: full 256-bin table */
                tab = mtab + (size_t)next_multi * 256;
                next_multi++;
                for (int code = 0; code < 256; code++) {                    /* alleles the calls actually use */
                    if (!tab[code]) continue;
                    int a1 = code >> 4, a2 = code & 0xF;
                    if (a1 != 0xF && a1 + 1 > na) na = a1 + 1;
                    if (a2 != 0xF && a2 + 1 > na) na = a2 + 1;
                }
            }
            if (na < 2) na = 2;
            s->num_alleles = na;
            s->alleles_count = (int *)calloc((size_t)na, sizeof(int));
            s->genotypes_count = (int *)calloc((size_t)na * na, sizeof(int));
            s->alleles_freq = (float *)calloc((size_t)na, sizeof(float));
            s->genotypes_freq = (float *)calloc((size_t)na * na, sizeof(float));
            s->missing_genotypes = c[4]; s->missing_alleles = c[5];
            if (tab) {
                for (int code = 0; code < 256; code++) {
                    int k = tab[code], a1 = code >> 4, a2 = code & 0xF;
                    if (!k) continue;
                    if (a1 != 0xF) s->alleles_count[a1] += k;
                    if (a2 != 0xF) s->alleles_count[a2] += k;
                    if (a1 != 0xF && a2 != 0xF) s->genotypes_count[a1 * na + a2] += k;
                }
            } else {
                s->alleles_count[0] = c[6]; s->alleles_count[1] = c[7];
                s->genotypes_count[0] = c[0]; s->genotypes_count[1] = c[1];
                s->genotypes_count[na] = c[2]; s->genotypes_count[na + 1] = c[3];
            }
            int ta = 0, tg = 0;
            for (int k = 0; k < na; k++) ta += s->alleles_count[k];
            for (int k = 0; k < na * na; k++) tg += s->genotypes_count[k];
            s->maf = 1.0f;
            for (int k = 0; k < na; k++) {
                s->alleles_freq[k] = ta ? (float)s->alleles_count[k] / ta : 0.0f;
                if (s->alleles_freq[k] < s->maf) s->maf = s->alleles_freq[k];
            }
            for (int k = 0; k < na * na; k++) s->genotypes_freq[k] = tg ? (float)s->genotypes_count[k] / tg : 0.0f;
            s->hw_chi2 = hw[i]; s->hw_p_value = hw[n + i];
            if (ng) {
                s->num_phenotypes = ng;
                s->phenotype_stats = (variant_phenotype_stats_t *)calloc((size_t)ng, sizeof *s->phenotype_stats);
                for (int k = 0; s->phenotype_stats && k < ng; k++) {
                    const int32_t *g = gc8 + ((size_t)k * n + i) * 8;
                    variant_phenotype_stats_t *ps = &s->phenotype_stats[k];
                    ps->genotypes_count[0] = g[0]; ps->genotypes_count[1] = g[1]; ps->genotypes_count[2] = g[2]; ps->genotypes_count[3] = g[3];
                    ps->missing_genotypes = g[4]; ps->missing_alleles = g[5];
                    ps->alleles_count[0] = g[6]; ps->alleles_count[1] = g[7];
                    const int t2 = g[6] + g[7];
                    ps->alleles_freq[0] = t2 ? (float)g[6] / t2 : 0.0f;
                    ps->alleles_freq[1] = t2 ? (float)g[7] / t2 : 0.0f;
                    ps->maf = ps->alleles_freq[0] < ps->alleles_freq[1] ? ps->alleles_freq[0] : ps->alleles_freq[1];
                    ps->hw_chi2 = ghw[(size_t)k * n + i]; ps->hw_p_value = ghw[((size_t)ng + k) * n + i];
                }
            }
            chain_add(&chain, list_item_new(tid, 0, s));
        }
        list_insert_chain(chain.first, chain.last, chain.n, output_list);
        if (file_stats) {
            pthread_mutex_lock(&file_stats->lock);
            file_stats->variants_count += num_variants;
            file_stats->samples_count = num_samples;
            file_stats->multiallelics_count += n_multi;
            file_stats->biallelics_count += num_variants - n_multi;
            pthread_mutex_unlock(&file_stats->lock);
        }
    }
    stage_put(gt, gt_slot); free(c8); free(hw); free(midx); free(mtab); free(group); free(gc8); free(ghw);
    adapter_clock_stop(&clk);
    return rc;
}

int get_sample_stats(vcf_record_t **variants, int num_variants, individual_t **individuals,
                     sample_ids_t *sample_ids, sample_stats_t **sample_stats, file_stats_t *file_stats) {
    (void)file_stats;
    if (num_variants <= 0) return 0;
    int rc = ensure_engine();
    if (rc) return rc;
    int num_samples = (int)variants[0]->samples->size;
    size_t n = (size_t)num_variants, pitch = (size_t)(num_samples > 0 ? num_samples : 1);
    uint8_t *gt = (uint8_t *)malloc(n * pitch + n);
    int32_t *c8 = (int32_t *)malloc(n * 8 * sizeof(int32_t));
    double *hw = (double *)malloc(n * 2 * sizeof(double));
    int32_t *miss = (int32_t *)calloc((size_t)(num_samples > 0 ? num_samples : 1), sizeof(int32_t));
    /* trios for the Mendelian check: every individual (VCF column j) whose father and mother are VCF columns too */
    int n_trios = 0;
    int32_t *tf = (int32_t *)malloc(pitch * sizeof(int32_t)), *tm = (int32_t *)malloc(pitch * sizeof(int32_t));
    int32_t *tc = (int32_t *)malloc(pitch * sizeof(int32_t)), *terr = (int32_t *)calloc(pitch, sizeof(int32_t));
    uint8_t *ts = (uint8_t *)malloc(pitch);
    if (!gt || !c8 || !hw || !miss || !tf || !tm || !tc || !ts || !terr) {
        free(gt); free(c8); free(hw); free(miss); free(tf); free(tm); free(tc); free(ts); free(terr);
        snprintf(g_err, sizeof g_err, "out of memory");
        return HPGV_ERR_NOMEM;
    }
    uint8_t *is_x = gt + n * pitch;
    uint64_t h = 1469598103934665603ULL;
    if (individuals && sample_ids) {
        for (int j = 0; j < num_samples; j++) {
            individual_t *ind = individuals[j];
            if (!ind || !ind->father || !ind->mother) continue;
            int fp = sample_ids_get(sample_ids, ind->father->id), mp = sample_ids_get(sample_ids, ind->mother->id);
            if (fp < 0 || mp < 0 || fp >= num_samples || mp >= num_samples) continue;
            tf[n_trios] = fp; tm[n_trios] = mp; tc[n_trios] = j;
            ts[n_trios] = (ind->sex == MALE) ? HPGV_SEX_MALE : (ind->sex == FEMALE) ? HPGV_SEX_FEMALE : HPGV_SEX_UNKNOWN;
            h = (h ^ (uint64_t)(uint32_t)fp ^ ((uint64_t)(uint32_t)mp << 20) ^ ((uint64_t)j << 40) ^ ((uint64_t)ts[n_trios] << 62)) * 1099511628211ULL;
            n_trios++;
        }
    }
    hpgv_host_stage_records(variants, num_variants, num_samples, 0, gt, is_x);
    pthread_rwlock_rdlock(&g_cohort_lock);
    rc = stats_prepare(num_samples);
    if (rc == HPGV_OK) {
        rc = hpgv_stats_ex(g_ctx, gt, pitch, num_variants, c8, hw, hw + n, miss, NULL, NULL, NULL);
        if (rc != HPGV_OK) host_fail("hpgv_stats_ex", rc);
    }
    if (rc == HPGV_OK && n_trios > 0) {
        while (rc == HPGV_OK && !(g_ped_key.set && g_ped_key.num_samples == num_samples && g_ped_key.n_trios == n_trios && g_ped_key.hash == h)) {
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_wrlock(&g_cohort_lock);
            if (!(g_ped_key.set && g_ped_key.num_samples == num_samples && g_ped_key.n_trios == n_trios && g_ped_key.hash == h)) {
                rc = hpgv_set_pedigree(g_ctx, num_samples, n_trios, tf, tm, tc, ts);
                if (rc == HPGV_OK) { g_ped_key.set = 1; g_ped_key.num_samples = num_samples; g_ped_key.n_trios = n_trios; g_ped_key.hash = h; }
                else host_fail("hpgv_set_pedigree", rc);
            }
            pthread_rwlock_unlock(&g_cohort_lock);
            pthread_rwlock_rdlock(&g_cohort_lock);
        }
        if (rc == HPGV_OK) {
            rc = hpgv_mendel(g_ctx, gt, pitch, num_variants, is_x, NULL, terr);
            if (rc != HPGV_OK) host_fail("hpgv_mendel", rc);
        }
    }
    pthread_rwlock_unlock(&g_cohort_lock);
    if (rc == HPGV_OK) {
        /* several workers may update the same sample_stats (stats_runner.c:189-198) */
        static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
        pthread_mutex_lock(&mu);
        for (int j = 0; j < num_samples; j++) sample_stats[j]->missing_genotypes += miss[j];
        for (int t = 0; t < n_trios; t++) sample_stats[tc[t]]->mendelian_errors += terr[t];
        pthread_mutex_unlock(&mu);
    }
    free(gt); free(c8); free(hw); free(miss); free(tf); free(tm); free(tc); free(ts); free(terr);
    return rc;
}
