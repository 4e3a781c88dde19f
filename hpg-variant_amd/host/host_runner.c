/* host_runner.c -- the file runners: reader -> engine threads -> writer (hpgv_run_assoc / tdt / aggregate / stats / vcf2epi).
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ---- the runners' pipeline: reader -> engine threads -> writer, batches in rotation ------------------ */
/* batches in rotation and engine threads: two engine threads per device (one batch's bus copies beside the other's
 * kernels) and three more batches than engines (reader ahead, writer behind); one device: 5 batches, 2 engines */
enum { B_FREE = 0, B_FILLED = 1, B_BUSY = 2, B_DONE = 3 };
typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv;
    run_batch_t bt[RUN_NB_MAX]; int state[RUN_NB_MAX]; long seq[RUN_NB_MAX];
    int nb, n_engines;
    long n_filled, n_taken, n_written;                  /* sequence numbers handed out so far per stage */
    int eof, rc, kind;
    size_t batch_bytes;
    line_reader_t *rd;
    double t_read, t_engine, t_write;
    char err[256];
} run_pipe_t;

static void pipe_fail(run_pipe_t *P, int rc, const char *msg) {      /* mu held */
    if (!P->rc) { P->rc = rc; snprintf(P->err, sizeof P->err, "%s", msg); }
    pthread_cond_broadcast(&P->cv);
}

static void *pipe_reader(void *v) {
    run_pipe_t *P = (run_pipe_t *)v;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        int k = -1;
        while (!P->rc) {
            for (int i = 0; i < P->nb && k < 0; i++) if (P->state[i] == B_FREE) k = i;
            if (k >= 0) break;
            pthread_cond_wait(&P->cv, &P->mu);
        }
        if (P->rc) { pthread_mutex_unlock(&P->mu); return NULL; }
        P->state[k] = B_BUSY;
        pthread_mutex_unlock(&P->mu);
        const double t0 = now_s();
        if (!P->bt[k].text) P->bt[k].text = text_buf_get(P->bt[k].text_cap);
        const size_t n = !P->bt[k].text ? (size_t)-1 : P->rd->devwin ? read_lines_dev(P->rd, P->bt[k].text, P->bt[k].text_cap, P->batch_bytes) : read_lines(P->rd, P->bt[k].text, P->batch_bytes);
        P->bt[k].dev_text = P->rd->last_dev; P->bt[k].dev_ctx = P->rd->devwin ? P->rd->last_ctx : NULL;
        P->bt[k].dev_base = P->rd->last_base; P->bt[k].dev_tiles = P->rd->last_dev ? P->rd->last_tiles : NULL; P->bt[k].dev_n_tiles = P->rd->last_n_tiles;
        const double dt = now_s() - t0;
        pthread_mutex_lock(&P->mu);
        P->t_read += dt;
        if (n == (size_t)-1) { P->state[k] = B_FREE; pipe_fail(P, HPGV_ERR_UNSUPPORTED, "read error, or a VCF line is longer than batch_bytes"); pthread_mutex_unlock(&P->mu); return NULL; }
        if (n == 0) { P->state[k] = B_FREE; P->eof = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); return NULL; }
        P->bt[k].bytes = n; P->seq[k] = P->n_filled++; P->state[k] = B_FILLED;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

static void *pipe_engine(void *v) {
    run_pipe_t *P = (run_pipe_t *)v;
    const int kind = P->kind;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        int k = -1;
        while (!P->rc) {
            for (int i = 0; i < P->nb && k < 0; i++) if (P->state[i] == B_FILLED && P->seq[i] == P->n_taken) k = i;
            if (k >= 0 || (P->eof && P->n_taken == P->n_filled)) break;
            pthread_cond_wait(&P->cv, &P->mu);
        }
        if (P->rc || k < 0) { pthread_mutex_unlock(&P->mu); return NULL; }
        P->state[k] = B_BUSY; P->n_taken++;
        pthread_mutex_unlock(&P->mu);
        const double t0 = now_s();
        run_batch_t *b = &P->bt[k];
        int rc = HPGV_OK;
        /* tokenize the device copy in place (on the device that holds it): no H2D of the text -- and, when the decoder left its tile
         * records, no counting sweep over it either */
        if (b->dev_text) (void)hpgv_text_alias_tiles(b->dev_ctx ? b->dev_ctx : g_ctx, b->text, b->dev_text, b->dev_base, b->dev_tiles, (uint64_t)b->dev_n_tiles);
        /* max_lines is sized for complete records; a batch of short (damaged) lines can hold more: the
         * engine reports the true count, the arrays grow and the batch is done again */
        for (int attempt = 0; attempt < 4; attempt++) {
            const int m = b->max_lines;
            if (kind == 5 || kind == 6) {
                memset(b->smiss, 0, sizeof(int32_t) * (size_t)b->n_smiss);
                memset(b->cerr, 0, sizeof(int32_t) * (size_t)b->n_cerr);
                /* the 256-bin tables of multi-allelic lines: room for 4 096 of them (4 MB), grown to what a batch really holds -- one
                 * per possible line is 1 KB x max_lines, hundreds of MB per batch for a narrow cohort in 256 MB windows */
                if (!b->mtab) { b->multi_cap = m < 4096 ? m : 4096; b->mtab = (int32_t *)malloc(sizeof(int32_t) * 256 * (size_t)b->multi_cap); if (!b->mtab) { rc = HPGV_ERR_NOMEM; break; } }
                b->n_multi = b->multi_cap;
                const int mend = kind == 6 && b->n_cerr > 0;
                const size_t gm = (size_t)m * (size_t)b->n_groups;
                rc = hpgv_stats_text_groups(g_ctx, b->text, b->bytes, m, &b->n_lines, b->line_off, b->field_off, b->status, b->c8, b->hw, b->hw + m,
                                            kind == 6 ? b->smiss : NULL, b->midx, b->mtab, &b->n_multi, mend ? b->merr : NULL, mend ? b->cerr : NULL,
                                            b->n_groups ? b->gc8 : NULL, b->n_groups ? b->ghw : NULL, b->n_groups ? b->ghw + gm : NULL);
            } else if (kind == 4)
                rc = hpgv_epi_dataset_text(g_ctx, b->text, b->bytes, m, &b->n_lines, b->line_off, b->field_off, b->status, b->rows);
            else if (kind == 3)
                rc = hpgv_tdt_text(g_ctx, b->text, b->bytes, m, &b->n_lines, b->line_off, b->field_off, b->status,
                                   b->ints, b->ints + m, b->dbl, b->dbl + m, b->dbl + 2 * m);
            else
                rc = hpgv_assoc_text(g_ctx, kind, b->text, b->bytes, m, &b->n_lines, b->line_off, b->field_off, b->status,
                                     b->ints, b->ints + m, b->ints + 2 * m, b->ints + 3 * m,
                                     b->dbl, kind == CHI_SQUARE ? b->dbl + m : NULL, b->dbl + 2 * m);
            if (!rc && b->n_lines <= b->max_lines && (kind == 5 || kind == 6) && b->n_multi > b->multi_cap) {      /* more multi-allelic lines than tables: again, with room */
                free(b->mtab);
                b->multi_cap = b->n_multi + b->n_multi / 8 + 16;
                if (b->multi_cap > b->max_lines) b->multi_cap = b->max_lines;
                b->mtab = (int32_t *)malloc(sizeof(int32_t) * 256 * (size_t)b->multi_cap);
                if (!b->mtab) { rc = HPGV_ERR_NOMEM; break; }
                continue;
            }
            if (rc || b->n_lines <= b->max_lines) break;
            free(b->mtab); b->mtab = NULL;
            if (run_batch_reserve(b, b->n_lines)) { rc = HPGV_ERR_NOMEM; break; }
        }
        if (b->dev_text) (void)hpgv_text_alias(b->dev_ctx ? b->dev_ctx : g_ctx, b->text, NULL);
        const double dt = now_s() - t0;
        pthread_mutex_lock(&P->mu);
        P->t_engine += dt;
        if (rc) {
            char msg[256];
            snprintf(msg, sizeof msg, "%s failed (%d): %s", kind >= 5 ? "hpgv_stats_text" : kind == 4 ? "hpgv_epi_dataset_text" : kind == 3 ? "hpgv_tdt_text" : "hpgv_assoc_text", rc,
                     rc == HPGV_ERR_NOMEM ? "out of memory" : hpgv_last_error(g_ctx));
            pipe_fail(P, rc, msg);
            pthread_mutex_unlock(&P->mu);
            return NULL;
        }
        P->state[k] = B_DONE;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

/* ---- hpg-var-vcf stats: what the run accumulates besides the per-variant lines (sample_stats_t, file_stats_t;
 *      the report writers live in hpg-libs, so the two files below are this project's rendering) ---- */
typedef struct {
    long *smiss, *serr;                                   /* per VCF column: missing genotypes, Mendelian errors as a child */
    long variants, biallelic, multiallelic, snps, indels, transitions, transversions, pass, with_quality;
    double quality_sum;
} run_stats_t;

static void run_stats_add(run_stats_t *R, const run_batch_t *b, int n_samples, const int32_t *trio_child) {
    for (int j = 0; j < n_samples; j++) R->smiss[j] += b->smiss[j];      /* counted over every line of the batch, as get_sample_stats does */
    for (int t = 0; t < b->n_cerr; t++) R->serr[trio_child[t]] += b->cerr[t];
    const int n = b->n_lines < b->max_lines ? b->n_lines : b->max_lines;
    for (int i = 0; i < n; i++) {
        if (!record_passes(b, i)) continue;
        const uint32_t *fo = b->field_off + 10 * (size_t)i;
        const char *l = b->text + b->line_off[i];
        const char *ref = l + fo[3], *alt = l + fo[4];
        const int lr = (int)(fo[4] - 1 - fo[3]), la = (int)(fo[5] - 1 - fo[4]);
        vcounts_t v;
        record_counts(b, i, alt, la, &v);
        R->variants++;
        if (v.na > 2) R->multiallelic++; else R->biallelic++;
        int snp = lr == 1, n_alt = 0;                      /* a SNP: REF and every ALT allele one base long */
        for (int k = 0; k <= la; k++)
            if (k == la || alt[k] == ',') { n_alt++; }
        for (int k = 0, start = 0; k <= la && snp; k++)
            if (k == la || alt[k] == ',') { if (k - start != 1 || alt[start] == '.') snp = 0; start = k + 1; }
        if (snp) {
            R->snps++;
            if (n_alt == 1) {
                const char a = (char)(ref[0] & ~0x20), c = (char)(alt[0] & ~0x20);
                const int purine_a = a == 'A' || a == 'G', purine_c = c == 'A' || c == 'G';
                if (purine_a == purine_c) R->transitions++; else R->transversions++;
            }
        } else if (!(la == 1 && alt[0] == '.')) R->indels++;
        if (fo[6] != 0xFFFFFFFFu) {
            const char *q = l + fo[5];
            if (*q != '.' && *q != '\t') { R->quality_sum += strtod(q, NULL); R->with_quality++; }
            const char *f = l + fo[6];
            const int lf = fo[7] != 0xFFFFFFFFu ? (int)(fo[7] - 1 - fo[6]) : 0;
            if (lf == 4 && !strncmp(f, "PASS", 4)) R->pass++;
        }
    }
}

static int run_stats_write(const run_stats_t *R, const char *prefix, char **names, int n_samples, long written) {
    char *path = (char *)malloc(strlen(prefix) + 32);
    if (!path) return HPGV_ERR_NOMEM;
    sprintf(path, "%s.stats-samples", prefix);
    FILE *f = fopen(path, "w");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot create %s", path); free(path); return HPGV_ERR_INVALID; }
    fprintf(f, "#SAMPLE\tMISS_GT\tMEND_ER\n");
    for (int j = 0; j < n_samples; j++) fprintf(f, "%s\t%ld\t%ld\n", names[j], R->smiss[j], R->serr[j]);
    fclose(f);
    sprintf(path, "%s.stats-summary", prefix);
    f = fopen(path, "w");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot create %s", path); free(path); return HPGV_ERR_INVALID; }
    fprintf(f, "Number of variants = %ld\nNumber of samples = %d\nNumber of biallelic variants = %ld\nNumber of multiallelic variants = %ld\n\n",
            written, n_samples, R->biallelic, R->multiallelic);
    fprintf(f, "Number of SNP = %ld\nNumber of indels = %ld\n\n", R->snps, R->indels);
    fprintf(f, "Number of transitions = %ld\nNumber of transversions = %ld\nTi/TV ratio = %.4f\n\n", R->transitions, R->transversions,
            R->transversions ? (double)R->transitions / (double)R->transversions : 0.0);
    fprintf(f, "Percentage of PASS = %.2f%%\nAverage quality = %.2f\n", R->variants ? 100.0 * (double)R->pass / (double)R->variants : 0.0,
            R->with_quality ? R->quality_sum / (double)R->with_quality : 0.0);
    fclose(f);
    free(path);
    return HPGV_OK;
}

/* the per-phenotype lines of a batch: the counters of the first two alleles within the group (variant_stats_t per
 * phenotype, stats_runner.c:319-323) */
static void write_group_lines(FILE **gfd, const run_batch_t *b) {
    const int n = b->n_lines < b->max_lines ? b->n_lines : b->max_lines, m = b->max_lines;
    for (int g = 0; g < b->n_groups; g++)
        for (int i = 0; i < n; i++) {
            if (!record_passes(b, i)) continue;
            const uint32_t *fo = b->field_off + 10 * (size_t)i;
            const char *l = b->text + b->line_off[i];
            const int32_t *c = b->gc8 + ((size_t)g * m + (size_t)i) * 8;
            const int ta = c[6] + c[7];
            const float f0 = ta ? (float)c[6] / ta : 0.0f, f1 = ta ? (float)c[7] / ta : 0.0f;
            fprintf(gfd[g], "%.*s\t%ld\t%.*s\t%.*s\t%d,%d\t%.4f,%.4f\t0/0:%d,0/1:%d,1/1:%d,./.:%d\t%d\t%d\t%.4f\t%.6g\t%.6g\n",
                    (int)(fo[1] - 1 - fo[0]), l + fo[0], atol(l + fo[1]), (int)(fo[4] - 1 - fo[3]), l + fo[3], (int)(fo[5] - 1 - fo[4]), l + fo[4],
                    c[6], c[7], f0, f1, c[0], c[1] + c[2], c[3], c[4], c[5], c[4], f0 < f1 ? f0 : f1,
                    b->ghw[(size_t)g * m + (size_t)i], b->ghw[((size_t)b->n_groups + (size_t)g) * m + (size_t)i]);
        }
}

static int run_file(const char *vcf_path, const char *ped_path, const char *out_path, int kind, size_t batch_bytes,
                    long *n_variants_out) {
    host_env_read();                                               /* the environment: once per run (hpgv_host.h "Environment") */
    const double t_enter = now_s();
    g_write_split[0] = g_write_split[1] = 0;
    g_input_err[0] = 0;
    int rc = ensure_engine();
    if (rc) return rc;
    if (batch_bytes < (1u << 16)) batch_bytes = 1u << 16;
    const int io_threads = default_io_threads();
    ped_table_t ped;
    memset(&ped, 0, sizeof ped);
    if (!ped_path && kind < 5) { snprintf(g_err, sizeof g_err, "this runner needs a PED file (ped_path is NULL)"); return HPGV_ERR_INVALID; }
    if (ped_path) { if ((rc = ped_table_read(ped_path, &ped))) return rc; }      /* aggregate / stats run without a PED too */
    line_reader_t rd;
    memset(&rd, 0, sizeof rd);
    if (source_open(&rd.src, vcf_path)) {
        ped_table_free(&ped);
        snprintf(g_err, sizeof g_err, "cannot open VCF file %s", vcf_path);
        return HPGV_ERR_INVALID;
    }
    char *hdr = NULL;
    char **names = NULL;
    size_t chrom_off = 0;
    const double t_opened = now_s();
    const int n_samples = vcf_header_read(&rd, &hdr, &names, &chrom_off);
    const double t_header = now_s();
    if (n_samples < 0) { source_close(&rd.src); free(rd.carry); free(hdr); ped_table_free(&ped); snprintf(g_err, sizeof g_err, "%s%sno #CHROM header line in %s", g_input_err, g_input_err[0] ? "; " : "", vcf_path); return HPGV_ERR_INVALID; }

    /* cohort: PED rows looked up by sample name (associate_samples_and_positions + sort_individuals) */
    sample_ids_t *ids = sample_ids_new((size_t)n_samples);
    for (int j = 0; j < n_samples; j++) sample_ids_put(ids, names[j], j);
    uint32_t epi_aff = 0, epi_unaff = 0;
    int n_trios = 0, n_groups = 0;
    char **group_names = NULL;                           /* stats: the phenotype values, pointing into the PED text */
    int32_t *trio_child = NULL;                          /* stats: VCF column of every trio's child */
    pthread_rwlock_wrlock(&g_cohort_lock);
    if (kind >= 5) {
        /* get_variants_stats / get_sample_stats over all columns; with a PED, the trios whose three members are VCF
         * columns give the Mendelian errors (stats_runner.c:165-170,194-198) */
        rc = hpgv_set_stats_cohort(g_ctx, n_samples);
        g_stats_key.set = 0;
        if (rc) host_fail("hpgv_set_stats_cohort", rc);
        if (!rc && kind == 6 && ped.n > 0) {
            /* phenotype groups (stats_runner.c:47-50,165-170): the distinct values of the PED's PHENO column, numbered in
             * order of first appearance; a VCF column without a PED row belongs to no group */
            int32_t *group = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_samples + 1));
            group_names = (char **)malloc(sizeof(char *) * (size_t)(ped.n + 1));
            for (int j = 0; j < n_samples; j++) group[j] = -1;
            for (int i = 0; i < ped.n; i++) {
                int gidx = -1;
                for (int k = 0; k < n_groups; k++) if (!strcmp(group_names[k], ped.phe[i])) { gidx = k; break; }
                if (gidx < 0 && n_groups < 4096) { gidx = n_groups; group_names[n_groups++] = ped.phe[i]; }
                const int j = sample_ids_get(ids, ped.iid[i]);
                if (j >= 0) group[j] = gidx;
            }
            if (n_groups > 0) {
                rc = hpgv_set_stats_groups(g_ctx, group, n_samples, n_groups);
                g_group_key.set = 0;
                if (rc) host_fail("hpgv_set_stats_groups", rc);
            }
            free(group);
        }
        if (!rc && kind == 6 && ped.n > 0) {
            int32_t *tf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1)), *tm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
            trio_child = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
            uint8_t *ts = (uint8_t *)malloc((size_t)ped.n + 1);
            for (int i = 0; i < ped.n; i++) {
                if (!strcmp(ped.pat[i], "0") || !strcmp(ped.mat[i], "0")) continue;
                const int cp = sample_ids_get(ids, ped.iid[i]), fp = sample_ids_get(ids, ped.pat[i]), mp = sample_ids_get(ids, ped.mat[i]);
                if (cp < 0 || fp < 0 || mp < 0) continue;
                tf[n_trios] = fp; tm[n_trios] = mp; trio_child[n_trios] = cp; ts[n_trios] = (uint8_t)ped.sex[i]; n_trios++;
            }
            if (n_trios > 0) {
                rc = hpgv_set_pedigree(g_ctx, n_samples, n_trios, tf, tm, trio_child, ts);
                g_ped_key.set = 0;
                if (rc) host_fail("hpgv_set_pedigree", rc);
            }
            free(tf); free(tm); free(ts);
        }
    } else if (kind == 3) {
        /* families in order of first appearance; father / mother = founders by sex (tdt.c:62-73);
         * counted children = rows with both parents named, affected, present in the VCF (tdt.c:139-148) */
        int32_t *fcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1)), *mcol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
        int32_t *coff = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 2)), *ccol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
        uint8_t *csex = (uint8_t *)malloc((size_t)ped.n + 1);
        char *done = (char *)calloc((size_t)ped.n + 1, 1);
        int nf = 0, nc = 0;
        coff[0] = 0;
        for (int i = 0; i < ped.n; i++) {
            if (done[i]) continue;
            int father = -1, mother = -1;
            for (int k = i; k < ped.n; k++) {
                if (strcmp(ped.fid[k], ped.fid[i])) continue;
                done[k] = 1;
                if (!strcmp(ped.pat[k], "0") && !strcmp(ped.mat[k], "0") && !(father >= 0 && mother >= 0)) {
                    if (ped.sex[k] == HPGV_SEX_MALE) father = k; else if (ped.sex[k] == HPGV_SEX_FEMALE) mother = k;
                }
            }
            int fp = father >= 0 ? sample_ids_get(ids, ped.iid[father]) : -1, mp = mother >= 0 ? sample_ids_get(ids, ped.iid[mother]) : -1;
            fcol[nf] = (fp >= 0 && mp >= 0) ? fp : -1;
            mcol[nf] = (fp >= 0 && mp >= 0) ? mp : -1;
            if (fcol[nf] >= 0)
                for (int k = i; k < ped.n; k++) {
                    if (strcmp(ped.fid[k], ped.fid[i])) continue;
                    if (!strcmp(ped.pat[k], "0") || !strcmp(ped.mat[k], "0")) continue;       /* child->father && child->mother */
                    if (ped.pheno[k] != HPGV_COND_AFFECTED) continue;
                    int cp = sample_ids_get(ids, ped.iid[k]);
                    if (cp < 0) continue;
                    ccol[nc] = cp; csex[nc] = (uint8_t)ped.sex[k]; nc++;
                }
            coff[++nf] = nc;
        }
        rc = hpgv_set_families(g_ctx, n_samples, nf, fcol, mcol, coff, ccol, csex);
        g_tdt_key.set = 0;
        if (rc) host_fail("hpgv_set_families", rc);
        free(fcol); free(mcol); free(coff); free(ccol); free(csex); free(done);
    } else {
        uint8_t *cond = (uint8_t *)malloc((size_t)n_samples + 1);
        for (int j = 0; j < n_samples; j++) cond[j] = HPGV_COND_OTHER;
        int matched = 0;
        for (int i = 0; i < ped.n; i++) { int j = sample_ids_get(ids, ped.iid[i]); if (j >= 0) { cond[j] = (uint8_t)ped.pheno[i]; matched++; } }
        if (matched == 0 && n_samples > 0) {             /* assert(individual) of assoc.c:92: a VCF whose samples the PED does not know */
            snprintf(g_err, sizeof g_err, "no sample of %s is a row of %s", vcf_path, ped_path);
            rc = HPGV_ERR_INVALID;
        }
        if (kind == 4) {                                 /* get_individual_phenotypes, dataset_creator.c:279-300: affected, or not */
            for (int j = 0; j < n_samples; j++) {
                if (cond[j] != HPGV_COND_AFFECTED) cond[j] = HPGV_COND_UNAFFECTED;
                if (cond[j] == HPGV_COND_AFFECTED) epi_aff++; else epi_unaff++;
            }
        }
        if (!rc && (rc = hpgv_set_cohort(g_ctx, cond, n_samples))) host_fail("hpgv_set_cohort", rc);
        g_assoc_key.set = 0;
        free(cond);
        if (!rc && kind == FISHER) {
            double *lf = init_logarithm_array(n_samples * 10 > 16 ? n_samples * 10 : 16);     /* assoc_runner.c:164-166 */
            rc = hpgv_set_logfact(g_ctx, lf, (size_t)(n_samples * 10 > 16 ? n_samples * 10 : 16));
            g_lf_key.table = NULL;
            if (rc) host_fail("hpgv_set_logfact", rc);
            free(lf);
        }
    }
    /* device-side record filters: the count filters scan the stats layout of all columns, the Mendelian filter the
     * trios of the PED whose three members are VCF columns (every child with both parents, whatever its phenotype) */
    const int dev_filters = g_filters.min_maf >= 0.0 || g_filters.max_missing >= 0.0 || g_filters.max_mendel_errors >= 0;
    if (!rc && (g_filters.min_maf >= 0.0 || g_filters.max_missing >= 0.0)) {
        rc = hpgv_set_stats_cohort(g_ctx, n_samples);
        g_stats_key.set = 0;
        if (rc) host_fail("hpgv_set_stats_cohort", rc);
    }
    if (!rc && g_filters.max_mendel_errors >= 0) {
        int32_t *tf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1)), *tm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
        int32_t *tc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ped.n + 1));
        uint8_t *ts = (uint8_t *)malloc((size_t)ped.n + 1);
        int nt = 0;
        for (int i = 0; i < ped.n; i++) {
            if (!strcmp(ped.pat[i], "0") || !strcmp(ped.mat[i], "0")) continue;
            const int cp = sample_ids_get(ids, ped.iid[i]), fp = sample_ids_get(ids, ped.pat[i]), mp = sample_ids_get(ids, ped.mat[i]);
            if (cp < 0 || fp < 0 || mp < 0) continue;
            tf[nt] = fp; tm[nt] = mp; tc[nt] = cp; ts[nt] = (uint8_t)ped.sex[i]; nt++;
        }
        rc = hpgv_set_pedigree(g_ctx, n_samples, nt, tf, tm, tc, ts);
        g_ped_key.set = 0;
        if (rc) host_fail("hpgv_set_pedigree", rc);
        free(tf); free(tm); free(tc); free(ts);
    }
    if (!rc) (void)hpgv_set_text_filters(g_ctx, g_filters.min_maf, g_filters.max_missing, (long)g_filters.max_mendel_errors);
    /* the run keeps the cohort lock (exclusive) until its pipeline is done: the engine threads scan with the layouts installed
     * above, and an adapter or another runner with a different cohort waits instead of swapping them mid-file */
    sample_ids_free(ids);

    char *path6 = NULL;
    if (kind == 6) {
        path6 = (char *)malloc(strlen(out_path) + 32);
        if (path6) sprintf(path6, "%s.stats-variants", out_path); else rc = rc ? rc : HPGV_ERR_NOMEM;
    }
    FILE *out = rc ? NULL : fopen(kind == 6 ? path6 : out_path, "wb");
    if (!rc && !out) { snprintf(g_err, sizeof g_err, "cannot create %s", kind == 6 ? path6 : out_path); rc = HPGV_ERR_INVALID; }
    FILE **gfd = NULL;
    if (!rc && kind == 6 && n_groups > 0) {               /* one file per phenotype (stats_runner.c:267-297) */
        gfd = (FILE **)calloc((size_t)n_groups, sizeof(FILE *));
        char *gp = (char *)malloc(strlen(out_path) + 300);
        for (int k = 0; gfd && gp && k < n_groups && !rc; k++) {
            snprintf(gp, strlen(out_path) + 300, "%s.phenotype-%.200s.stats-variants", out_path, group_names[k]);
            if (!(gfd[k] = fopen(gp, "w"))) { snprintf(g_err, sizeof g_err, "cannot create %s", gp); rc = HPGV_ERR_INVALID; }
            else fprintf(gfd[k], "#CHROM\tPOS\tREF\tALT\tALLELES_COUNT\tALLELES_FREQ\tGENOTYPES_COUNT\tMISS_AL\tMISS_GT\tMAF\tHWE_CHI2\tHWE_P\n");
        }
        free(gp);
        if (!gfd) rc = HPGV_ERR_NOMEM;
    }
    run_stats_t *RS = NULL;
    if (!rc && kind == 6) {
        RS = (run_stats_t *)calloc(1, sizeof *RS);
        if (RS) { RS->smiss = (long *)calloc((size_t)n_samples + 1, sizeof(long)); RS->serr = (long *)calloc((size_t)n_samples + 1, sizeof(long)); }
        if (!RS || !RS->smiss || !RS->serr) rc = HPGV_ERR_NOMEM;
    }
    if (out) setvbuf(out, NULL, _IOFBF, 1u << 20);
    long written = 0;
    double t_sort = 0;
    order_track_t ord;
    memset(&ord, 0, sizeof ord);
    const double t_start = now_s();
    cpu_set_t saved_cpus;
    const int numa_bound = numa_bind_to_device(&saved_cpus);        /* before the buffers are allocated and the threads start */
    run_pipe_t *P = (run_pipe_t *)calloc(1, sizeof *P);
    out_buf_t *fmt = (out_buf_t *)calloc(RUN_FMT_BUFS, sizeof *fmt);
    int have = 0;
    if (!P || !fmt) rc = rc ? rc : HPGV_ERR_NOMEM;
    if (P) {
        int devs = hpgv_group_size(g_ctx);
        P->n_engines = 2 * (devs < 1 ? 1 : devs);
        /* windows of a text decoded on member 0's device stay there; a batch is then a chain of short kernels and two
         * small copies back, which four in flight overlap better than two (8 GB of text: 0.111 -> 0.100 s) */
        if (rd.src.d_text && !g_env.no_device_windows) P->n_engines = rd.src.mp && 2 * rd.src.mp->n > 4 ? 2 * rd.src.mp->n : 4;
        if (g_env.engine_threads > 0) P->n_engines = (int)g_env.engine_threads;      /* diagnosis: engine threads (batches in flight on the devices) */
        if (P->n_engines > RUN_ENGINES_MAX) P->n_engines = RUN_ENGINES_MAX;
        P->nb = P->n_engines + 3;
    }
    /* windows of a text that is on the device are not copied anywhere, so they need not be as small as the caller's batches:
     * about 64 of them per file, 256 MB at most, amortise what a batch costs whatever its size (three waits for the
     * device and 50 us of short kernels beside 100 us per 64 MB of tokenizing and scanning) */
    if (P && rd.src.d_text && !g_env.no_device_windows && !g_env.no_large_windows) {
        size_t w = rd.src.text_est / 64;
        if (w > ((size_t)256 << 20)) w = (size_t)256 << 20;
        if (w > batch_bytes) batch_bytes = w;
    }
    for (; !rc && have < P->nb; have++) rc = run_batch_alloc(&P->bt[have], batch_bytes, n_samples, kind == 4 ? n_samples : 0, kind >= 5, n_trios, n_groups);
    if (rc == HPGV_ERR_NOMEM) snprintf(g_err, sizeof g_err, "out of memory for the batch buffers");
    if (!rc) {
        if (kind == 5) {
            /* write_vcf_header_nosamples after add_aggregator_header (aggregate_runner.c:171-173,226-245): the file's
             * meta lines, the INFO entries of the added fields (texts: etc/hpg-variant/vcf-info-fields.conf), the
             * delimiter line without FORMAT and samples */
            if (chrom_off && fwrite(hdr, 1, chrom_off, out) != chrom_off) rc = HPGV_ERR_INVALID;
            const char *pre = g_aggregate_overwrite ? "" : "HPG_", *by = g_aggregate_overwrite ? "" : "Calculated by HPG Variant: ";
            fprintf(out, "##INFO=<ID=%sAC,Number=.,Type=Integer,Description=\"%sAllele count in genotypes, for each ALT allele, in the same order as listed\">\n", pre, by);
            fprintf(out, "##INFO=<ID=%sAF,Number=.,Type=Float,Description=\"%sAllele Frequency, for each ALT allele, in the same order as listed\">\n", pre, by);
            fprintf(out, "##INFO=<ID=%sAN,Number=1,Type=Integer,Description=\"%sTotal number of alleles in called genotypes\">\n", pre, by);
            fprintf(out, "##INFO=<ID=HPG_GTC,Number=.,Type=String,Description=\"Calculated by HPG Variant: Genotype counts, in pairs genotype:count\">\n");
            fprintf(out, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
        } else if (kind == 6) {
            fprintf(out, "#CHROM\tPOS\tREF\tALT\tNUM_ALLELES\tALLELES_COUNT\tALLELES_FREQ\tGENOTYPES_COUNT\tMISS_AL\tMISS_GT\tMAF\tMEND_ER\tHWE_CHI2\tHWE_P\n");
        } else if (kind == 4) {                          /* room for the number of variants, then the class sizes (dataset_creator.c:186-193) */
            const uint32_t head[3] = {0, epi_aff, epi_unaff};
            if (fwrite(head, sizeof(uint32_t), 3, out) != 3) rc = HPGV_ERR_INVALID;
        } else if (kind == 3) { tdt_write_output_header(out); order_track_keep(&ord, "#CHR\tPOS\tID\tA1\tA2\tT\tU\tOR\tCHISQ\tP-VALUE", 38); }
        else {
            assoc_write_output_header((enum ASSOC_task)kind, out);
            const char *h = kind == 1 ? "#CHR\tPOS\tID\tA1\tC_A1\tC_U1\tF_A1\tF_U1\tA2\tC_A2\tC_U2\tF_A2\tF_U2\tOR\tCHISQ\tP-VALUE"
                                      : "#CHR\tPOS\tID\tA1\tC_A1\tC_U1\tF_A1\tF_U1\tA2\tC_A2\tC_U2\tF_A2\tF_U2\tOR\tP-VALUE";
            order_track_keep(&ord, h, strlen(h));
        }
        /* one reader thread (with its team of pread / inflate threads), two engine threads per device (each call
         * is H2D, tokenize, scan, statistics, D2H on its own stream, so two in flight overlap the copies of one
         * batch with the kernels of the other) and this thread as the writer (with its team of formatters);
         * batches are written in file order */
        pthread_mutex_init(&P->mu, NULL);
        pthread_cond_init(&P->cv, NULL);
        P->kind = kind; P->batch_bytes = batch_bytes; P->rd = &rd;
        io_pool_t rpool, wpool;
        pool_init(&rpool, io_threads);
        pool_init(&wpool, io_threads);
        rd.src.pool = &rpool;
        if (rd.src.d_text && !g_env.no_device_windows) {      /* bgzip decoded on the device: windows of the device text from the first data line on */
            rd.src.dev_pos -= rd.carry_len; rd.carry_len = 0; rd.devwin = 1;
        }
        const int n_fmt = io_threads < RUN_FMT_BUFS / 2 ? io_threads : RUN_FMT_BUFS / 2;      /* two sets of buffers: one is written while the other is filled */
        file_writer_t fw;
        memset(&fw, 0, sizeof fw);
        /* (the vcf2epi rows are written out of the batch itself, and the stats tool's group files by this thread) */
        const int use_fw = kind != 4 && !g_env.no_writer_thread && file_writer_start(&fw, out);
        int fmt_set = 0;
        pthread_t th[1 + RUN_ENGINES_MAX];
        int n_th = 0;
        if (pthread_create(&th[n_th], NULL, pipe_reader, P) == 0) n_th++;
        for (int e = 0; e < P->n_engines; e++) if (pthread_create(&th[n_th], NULL, pipe_engine, P) == 0) n_th++;
        pthread_mutex_lock(&P->mu);
        if (n_th < 2) pipe_fail(P, HPGV_ERR_NOMEM, "cannot start the pipeline threads");
        for (;;) {
            int k = -1;
            while (!P->rc) {
                for (int i = 0; i < P->nb && k < 0; i++) if (P->state[i] == B_DONE && P->seq[i] == P->n_written) k = i;
                if (k >= 0 || (P->eof && P->n_written == P->n_filled)) break;
                pthread_cond_wait(&P->cv, &P->mu);
            }
            if (P->rc || k < 0) break;
            P->state[k] = B_BUSY;
            pthread_mutex_unlock(&P->mu);
            const double t0 = now_s();
            const run_batch_t *b = &P->bt[k];
            const int bad = write_batch(out, kind, b, fmt + (fmt_set ? RUN_FMT_BUFS / 2 : 0), n_fmt, &wpool, &ord, use_fw ? &fw : NULL);
            fmt_set ^= use_fw;
            for (int i = 0; i < b->n_lines; i++) if (record_passes(b, i)) written++;
            if (kind == 6 && !bad) run_stats_add(RS, b, n_samples, trio_child);
            if (kind == 6 && !bad && gfd) write_group_lines(gfd, b);
            const double dt = now_s() - t0;
            pthread_mutex_lock(&P->mu);
            P->t_write += dt;
            if (bad) { pipe_fail(P, HPGV_ERR_INVALID, "cannot write the result file"); break; }
            P->state[k] = B_FREE; P->n_written++;
            pthread_cond_broadcast(&P->cv);
        }
        pthread_mutex_unlock(&P->mu);
        if (use_fw && file_writer_stop(&fw)) { pthread_mutex_lock(&P->mu); pipe_fail(P, HPGV_ERR_INVALID, "cannot write the result file"); pthread_mutex_unlock(&P->mu); }
        for (int i = 0; i < n_th; i++) pthread_join(th[i], NULL);
        rd.src.pool = NULL;
        pool_destroy(&rpool); pool_destroy(&wpool);
        if (P->rc) { rc = P->rc; snprintf(g_err, sizeof g_err, "%s%s%s", g_input_err, g_input_err[0] ? "; " : "", P->err); }
        g_run_times[0] = P->t_read; g_run_times[1] = P->t_engine; g_run_times[2] = P->t_write; g_run_times[5] = (double)P->n_filled;
        pthread_mutex_destroy(&P->mu); pthread_cond_destroy(&P->cv);
    }
    if (out && kind == 4 && !rc) {                       /* finally the real number of variants (dataset_creator.c:208-212) */
        const uint32_t nv = (uint32_t)written;
        if (fseek(out, 0, SEEK_SET) != 0 || fwrite(&nv, sizeof nv, 1, out) != 1) { snprintf(g_err, sizeof g_err, "cannot write %s", out_path); rc = HPGV_ERR_INVALID; }
    }
    if (out && fclose(out) != 0 && !rc) { snprintf(g_err, sizeof g_err, "cannot write %s", out_path); rc = HPGV_ERR_INVALID; }
    {
        const double t0 = now_s();
        /* (in order as written: nothing to do; HPGV_ALWAYS_SORT=1 reads the file back and checks all the same) */
        if (!rc && kind < 4 && (ord.disorder || !ord.have || g_env.always_sort) && hpgv_host_sort_output_file(out_path))      /* assoc_runner.c:255-261: only a warning there */
            fprintf(stderr, "WARN: results could not be sorted by chromosome and position\n");
        else if (!rc && kind < 4 && g_env.run_trace && !(ord.disorder || !ord.have || g_env.always_sort))
            fprintf(stderr, "hpgv run: the result file is in order as written\n");
        t_sort = now_s() - t0;
        free(ord.last);
    }
    if (!rc && kind == 6) rc = run_stats_write(RS, out_path, names, n_samples, written);
    if (RS) { free(RS->smiss); free(RS->serr); free(RS); }
    for (int k = 0; gfd && k < n_groups; k++) if (gfd[k]) fclose(gfd[k]);
    free(gfd); free(group_names);
    free(path6); free(trio_child);
    for (int k = 0; P && k < have; k++) run_batch_free(&P->bt[k]);
    for (int k = 0; fmt && k < RUN_FMT_BUFS; k++) free(fmt[k].p);
    free(fmt);
    const int n_engines_used = P ? P->n_engines : 0;
    free(P);
    if (dev_filters) (void)hpgv_set_text_filters(g_ctx, -1.0, -1.0, -1);
    numa_unbind(&saved_cpus, numa_bound);
    g_run_times[3] = t_sort; g_run_times[4] = now_s() - t_start;
    if (g_env.run_trace)
        fprintf(stderr, "hpgv run: %ld records, %.0f batches, %d io threads: read %.3f s, engine %.3f s (%d threads), write %.3f s (stages overlap), sort %.3f s, total %.3f s\n",
                written, g_run_times[5], io_threads, g_run_times[0], g_run_times[1], n_engines_used, g_run_times[2], t_sort, g_run_times[4]);
    const double t_done = now_s();
    source_close(&rd.src); free(rd.carry); free(rd.tailbuf); free(hdr); free(names); ped_table_free(&ped);
    if (n_variants_out) *n_variants_out = written;
    pthread_rwlock_unlock(&g_cohort_lock);
    if (g_env.run_trace)
        fprintf(stderr, "hpgv run: before the pipeline: PED and open %.4f s, VCF header %.4f s, cohort and buffers %.4f s; after it: %.4f s; of the write stage: formatting %.4f s, writing %.4f s\n",
                t_opened - t_enter, t_header - t_opened, t_start - t_header, now_s() - t_done, g_write_split[0], g_write_split[1]);
    return rc;
}

/* the runners' reader on its own: copies `in_path` (plain, gzip or BGZF) to `out_path` in whole-line batches
 * of at most batch_bytes; what the runners feed to the engine, batch by batch */
int hpgv_host_copy_lines(const char *in_path, const char *out_path, size_t batch_bytes, int skip_vcf_header, long *n_batches) {
    host_env_read();
    line_reader_t rd;
    memset(&rd, 0, sizeof rd);
    if (batch_bytes < (1u << 16)) batch_bytes = 1u << 16;
    if (source_open(&rd.src, in_path)) { snprintf(g_err, sizeof g_err, "cannot open %s", in_path); return HPGV_ERR_INVALID; }
    FILE *out = fopen(out_path, "wb");
    char *buf = (char *)malloc(batch_bytes);
    int rc = (out && buf) ? HPGV_OK : HPGV_ERR_INVALID;
    long nb = 0;
    io_pool_t pool;
    pool_init(&pool, default_io_threads());
    rd.src.pool = &pool;
    if (!rc && skip_vcf_header) {
        char *hdr = NULL, **names = NULL;
        if (vcf_header_read(&rd, &hdr, &names, NULL) < 0) { snprintf(g_err, sizeof g_err, "no #CHROM header line in %s", in_path); rc = HPGV_ERR_INVALID; }
        free(hdr); free(names);
    }
    while (!rc) {
        size_t n = read_lines(&rd, buf, batch_bytes);
        if (n == 0) break;
        if (n == (size_t)-1) { snprintf(g_err, sizeof g_err, "read error, or a line longer than batch_bytes, in %s", in_path); rc = HPGV_ERR_UNSUPPORTED; break; }
        if (!rd.eof && buf[n - 1] != '\n') { rc = HPGV_ERR_UNSUPPORTED; break; }
        if (fwrite(buf, 1, n, out) != n) { rc = HPGV_ERR_INVALID; break; }
        nb++;
    }
    if (out) fclose(out);
    pool_destroy(&pool);
    free(buf); free(rd.carry); source_close(&rd.src);
    if (n_batches) *n_batches = nb;
    return rc;
}

void hpgv_run_set_filters(const hpgv_run_filters_t *filters) {
    const hpgv_run_filters_t off = { -1.0, -1.0, -1, -1, -1.0 };
    g_filters = filters ? *filters : off;
}

void hpgv_host_last_run_times(double *seconds6) { memcpy(seconds6, g_run_times, sizeof g_run_times); }

int hpgv_run_assoc(const char *vcf_path, const char *ped_path, const char *out_path, enum ASSOC_task task,
                   size_t batch_bytes, long *n_variants_out) {
    if (task != CHI_SQUARE && task != FISHER) { snprintf(g_err, sizeof g_err, "task must be CHI_SQUARE or FISHER"); return HPGV_ERR_INVALID; }
    return run_file(vcf_path, ped_path, out_path, (int)task, batch_bytes, n_variants_out);
}

int hpgv_run_tdt(const char *vcf_path, const char *ped_path, const char *out_path, size_t batch_bytes, long *n_variants_out) {
    return run_file(vcf_path, ped_path, out_path, 3, batch_bytes, n_variants_out);
}

/* run_aggregate (src/vcf-tools/aggregate/aggregate_runner.c:23-222) */
int hpgv_run_aggregate(const char *vcf_path, const char *out_path, int overwrite, size_t batch_bytes, long *n_variants_out) {
    g_aggregate_overwrite = overwrite ? 1 : 0;
    return run_file(vcf_path, NULL, out_path, 5, batch_bytes, n_variants_out);
}

/* run_stats (src/vcf-tools/stats/stats_runner.c:23-420) without the per-phenotype files and the database */
int hpgv_run_stats(const char *vcf_path, const char *ped_path, const char *out_prefix, size_t batch_bytes, long *n_variants_out) {
    /* the per-sample counters are sums over every data line of a batch, so the record filters are not applied here */
    const hpgv_run_filters_t saved = g_filters;
    hpgv_run_set_filters(NULL);
    const int rc = run_file(vcf_path, ped_path, out_prefix, 6, batch_bytes, n_variants_out);
    g_filters = saved;
    return rc;
}

int hpgv_run_vcf2epi(const char *vcf_path, const char *ped_path, const char *out_path, size_t batch_bytes, long *n_variants_out) {
    return run_file(vcf_path, ped_path, out_path, 4, batch_bytes, n_variants_out);
}
