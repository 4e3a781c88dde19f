/*
 * host_driver.c -- test driver for the host mirror (include/hpgv_host.h).
 *
 *   host_driver kat
 *       the reference's 9 tdt_test unit cases, built the way
 *       test/test_tdt_runner.c:93-433 builds them (individual_new, family_*,
 *       vcf_record_new, sample strings), run through tdt_test on the GPU.
 *   host_driver assoc <batch.txt> <ped.txt> <out-prefix> <threads> <batch-lines>
 *       runner-shaped loop (assoc_runner.c:106-207): OpenMP workers call
 *       assoc_test on batches; results are written with the reference's
 *       writers to <out-prefix>.chisq and <out-prefix>.fisher.
 *   host_driver tdt <batch.txt> <ped.txt> <out-prefix> <threads> <batch-lines>
 *   host_driver stats <batch.txt> <out-file>
 *
 * batch.txt:  "N V" / N sample names / V lines "chrom pos id ref alt format s1..sN"
 * ped.txt:    PLINK-style "fid iid pat mat sex(1=M,2=F) pheno(1=unaffected,2=affected)"
 */
#define _GNU_SOURCE
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpgv_host.h"

static int failures = 0;
#define CHECK(cond, msg) do { if (!(cond)) { printf("FAIL %s\n", msg); failures++; } else printf("PASS %s\n", msg); } while (0)

/* ---- KATs ----------------------------------------------------------------- */

typedef struct { const char *f, *m, *c; int child_affected; } trio_t;

static void run_kat(const char *name, const trio_t *trios, int n_trios, int exp_t1, int exp_t2) {
    vcf_record_t *record = vcf_record_new();
    set_vcf_record_chromosome("1", 1, record);
    set_vcf_record_position(111111, record);
    set_vcf_record_id("rs1", 3, record);
    set_vcf_record_reference("C", 1, record);
    set_vcf_record_alternate("T", 1, record);
    set_vcf_record_format("GT", 2, record);
    family_t **families = (family_t **)calloc((size_t)n_trios, sizeof(family_t *));
    sample_ids_t *ids = sample_ids_new(16);
    char names[16][3][16];
    individual_t *people[16][3];
    for (int t = 0; t < n_trios; t++) {
        snprintf(names[t][0], 16, "FAT%d", t); snprintf(names[t][1], 16, "MOT%d", t); snprintf(names[t][2], 16, "CHILD%d", t);
        families[t] = family_new("TESTFAM");
        individual_t *father = individual_new(names[t][0], 2.0, MALE, AFFECTED, NULL, NULL, families[t]);
        individual_t *mother = individual_new(names[t][1], 2.0, FEMALE, AFFECTED, NULL, NULL, families[t]);
        individual_t *child = individual_new(names[t][2], 2.0, MALE, trios[t].child_affected ? AFFECTED : UNAFFECTED,
                                             father, mother, families[t]);
        family_set_parent(father, families[t]);
        family_set_parent(mother, families[t]);
        family_add_child(child, families[t]);
        people[t][0] = father; people[t][1] = mother; people[t][2] = child;
        array_list_insert((void *)trios[t].f, record->samples);
        array_list_insert((void *)trios[t].m, record->samples);
        array_list_insert((void *)trios[t].c, record->samples);
        sample_ids_put(ids, names[t][0], 3 * t);
        sample_ids_put(ids, names[t][1], 3 * t + 1);
        sample_ids_put(ids, names[t][2], 3 * t + 2);
    }
    list_t out;
    list_init("output", 1, 1, &out);
    int rc = tdt_test(&record, 1, families, n_trios, ids, &out);
    char msg[128];
    snprintf(msg, sizeof msg, "%s: tdt_test returned 0", name);
    CHECK(rc == 0, msg);
    snprintf(msg, sizeof msg, "%s: one result inserted", name);
    CHECK(out.length == 1, msg);
    if (out.length == 1) {
        tdt_result_t *r = (tdt_result_t *)out.first_p->data_p;
        snprintf(msg, sizeof msg, "%s: t1 == %d (got %d)", name, exp_t1, r->t1);
        CHECK(r->t1 == exp_t1, msg);
        snprintf(msg, sizeof msg, "%s: t2 == %d (got %d)", name, exp_t2, r->t2);
        CHECK(r->t2 == exp_t2, msg);
    }
    list_decr_writers(&out);
    list_free_deep(&out, (void (*)(void *))tdt_result_free);
    for (int t = 0; t < n_trios; t++) {
        for (int k = 0; k < 3; k++) individual_free(people[t][k]);
        family_free(families[t]);
    }
    free(families);
    sample_ids_free(ids);
    vcf_record_free(record);
    hpgv_host_shutdown();      /* every KAT starts from a fresh engine, like a fresh test fixture */
}

static int cmd_kat(void) {
    const trio_t k1[] = {{"0/1", "0/1", "0/0", 0}};
    run_kat("family_unaffected_child", k1, 1, 0, 0);
    const trio_t k2[] = {{"0/1", "0/1", "0/0", 1}};
    run_kat("family_01_01_00", k2, 1, 2, 0);
    const trio_t k3[] = {{"0/1", "0/0", "0/0", 1}};
    run_kat("family_01_00_00", k3, 1, 1, 0);
    const trio_t k4[] = {{"0/1", "0/1", "0/1", 1}};
    run_kat("family_01_01_01", k4, 1, 1, 1);
    const trio_t k5[] = {{"0/1", "0/0", "0/1", 1}};
    run_kat("family_01_00_01", k5, 1, 0, 1);
    const trio_t k6[] = {{"0/1", "1/1", "0/1", 1}};
    run_kat("family_01_11_01", k6, 1, 1, 0);
    const trio_t k7[] = {{"0/0", "0/1", "0/1", 1}};
    run_kat("family_00_01_01", k7, 1, 0, 1);
    const trio_t k8[] = {{"1/1", "0/1", "0/1", 1}};
    run_kat("family_11_01_01", k8, 1, 1, 0);
    const trio_t k9[] = {{"0/1", "0/1", "0/0", 1}, {"0/1", "0/0", "0/0", 1}};
    run_kat("combined_families", k9, 2, 3, 0);
    printf("%s\n", failures ? "KAT FAILED" : "KAT OK");
    return failures ? 1 : 0;
}

/* ---- file inputs ------------------------------------------------------------ */

typedef struct {
    int n_samples, n_variants;
    char **sample_names;
    vcf_record_t **records;
    char *blob;
} batch_file_t;

static char *next_tok(char **p) {
    char *s = *p;
    while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') s++;
    if (!*s) return NULL;
    char *e = s;
    while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++;
    if (*e) { *e = 0; e++; }
    *p = e;
    return s;
}

static int load_batch(const char *path, batch_file_t *b) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    b->blob = (char *)malloc((size_t)sz + 1);
    if (fread(b->blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 1; }
    b->blob[sz] = 0;
    fclose(f);
    char *p = b->blob;
    b->n_samples = atoi(next_tok(&p));
    b->n_variants = atoi(next_tok(&p));
    b->sample_names = (char **)malloc(sizeof(char *) * (size_t)(b->n_samples + 1));
    for (int j = 0; j < b->n_samples; j++) b->sample_names[j] = next_tok(&p);
    b->records = (vcf_record_t **)malloc(sizeof(vcf_record_t *) * (size_t)(b->n_variants + 1));
    for (int i = 0; i < b->n_variants; i++) {
        vcf_record_t *r = vcf_record_new();
        char *chrom = next_tok(&p), *pos = next_tok(&p), *id = next_tok(&p), *ref = next_tok(&p), *alt = next_tok(&p), *fmt = next_tok(&p);
        set_vcf_record_chromosome(chrom, (int)strlen(chrom), r);
        set_vcf_record_position(atol(pos), r);
        set_vcf_record_id(id, (int)strlen(id), r);
        set_vcf_record_reference(ref, (int)strlen(ref), r);
        set_vcf_record_alternate(alt, (int)strlen(alt), r);
        set_vcf_record_format(fmt, (int)strlen(fmt), r);
        for (int j = 0; j < b->n_samples; j++) array_list_insert(next_tok(&p), r->samples);
        b->records[i] = r;
    }
    return 0;
}

typedef struct {
    int n;
    individual_t **people;
    char **fid, **pat, **mat;
    family_t **families;
    int n_families;
    char *blob;
} ped_t;

static int load_ped(const char *path, ped_t *ped) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    ped->blob = (char *)malloc((size_t)sz + 1);
    if (fread(ped->blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 1; }
    ped->blob[sz] = 0;
    fclose(f);
    int cap = 1024;
    ped->people = (individual_t **)malloc(sizeof(void *) * (size_t)cap);
    ped->fid = (char **)malloc(sizeof(void *) * (size_t)cap);
    ped->pat = (char **)malloc(sizeof(void *) * (size_t)cap);
    ped->mat = (char **)malloc(sizeof(void *) * (size_t)cap);
    ped->n = 0;
    char *p = ped->blob, *fid;
    while ((fid = next_tok(&p))) {
        char *iid = next_tok(&p), *pat = next_tok(&p), *mat = next_tok(&p), *sex = next_tok(&p), *phe = next_tok(&p);
        if (!phe) break;
        if (ped->n == cap) {
            cap *= 2;
            ped->people = (individual_t **)realloc(ped->people, sizeof(void *) * (size_t)cap);
            ped->fid = (char **)realloc(ped->fid, sizeof(void *) * (size_t)cap);
            ped->pat = (char **)realloc(ped->pat, sizeof(void *) * (size_t)cap);
            ped->mat = (char **)realloc(ped->mat, sizeof(void *) * (size_t)cap);
        }
        enum Sex s = !strcmp(sex, "1") ? MALE : !strcmp(sex, "2") ? FEMALE : UNKNOWN_SEX;
        /* stats_runner.c:50,86-87: unaffected label "1", affected "2" */
        enum Condition c = !strcmp(phe, "2") ? AFFECTED : !strcmp(phe, "1") ? UNAFFECTED : MISSING_CONDITION;
        /* variable = id of the PED variable's value, in order of first appearance (get_phenotypes / set_variable_field
         * "PHENO", stats_runner.c:47-50): here the phenotype column itself, "1" -> 0, "2" -> 1, anything else -> 2 */
        float var = !strcmp(phe, "1") ? 0.0f : !strcmp(phe, "2") ? 1.0f : 2.0f;
        ped->people[ped->n] = individual_new(iid, var, s, c, NULL, NULL, NULL);
        ped->fid[ped->n] = fid; ped->pat[ped->n] = pat; ped->mat[ped->n] = mat;
        ped->n++;
    }
    /* families: grouped by fid in order of first appearance (ped_flatten_families) */
    ped->families = (family_t **)malloc(sizeof(void *) * (size_t)(ped->n + 1));
    ped->n_families = 0;
    for (int i = 0; i < ped->n; i++) {
        family_t *fam = NULL;
        for (int k = 0; k < ped->n_families; k++)
            if (!strcmp(ped->families[k]->id, ped->fid[i])) { fam = ped->families[k]; break; }
        if (!fam) { fam = family_new(ped->fid[i]); ped->families[ped->n_families++] = fam; }
        ped->people[i]->family = fam;
    }
    for (int i = 0; i < ped->n; i++) {
        individual_t *ind = ped->people[i];
        for (int k = 0; k < ped->n; k++) {
            if (strcmp(ped->fid[k], ped->fid[i])) continue;
            if (!strcmp(ped->people[k]->id, ped->pat[i])) ind->father = ped->people[k];
            if (!strcmp(ped->people[k]->id, ped->mat[i])) ind->mother = ped->people[k];
        }
        if (!strcmp(ped->pat[i], "0") && !strcmp(ped->mat[i], "0")) family_set_parent(ind, ind->family);
        else family_add_child(ind, ind->family);
    }
    return 0;
}

static individual_t *find_person(const ped_t *ped, const char *id) {
    for (int i = 0; i < ped->n; i++) if (!strcmp(ped->people[i]->id, id)) return ped->people[i];
    return NULL;
}

/* ---- assoc: runner-shaped -------------------------------------------------------- */

static int cmd_assoc(const char *batch_path, const char *ped_path, const char *prefix, int threads, int lines) {
    batch_file_t b; ped_t ped;
    if (load_batch(batch_path, &b) || load_ped(ped_path, &ped)) return 2;
    /* sort_individuals (assoc_runner.c:140): PED individuals in VCF column order */
    individual_t **individuals = (individual_t **)malloc(sizeof(void *) * (size_t)(b.n_samples + 1));
    for (int j = 0; j < b.n_samples; j++) {
        individuals[j] = find_person(&ped, b.sample_names[j]);
        if (!individuals[j]) { fprintf(stderr, "sample %s not in PED\n", b.sample_names[j]); return 2; }
    }
    double *logfact = init_logarithm_array(b.n_samples * 10);          /* assoc_runner.c:164-166 */
    enum ASSOC_task tasks[2] = {CHI_SQUARE, FISHER};
    const char *ext[2] = {"chisq", "fisher"};
    for (int t = 0; t < 2; t++) {
        list_t out;
        list_init("output", threads, 0, &out);                       /* assoc_runner.c:25 */
        int n_batches = (b.n_variants + lines - 1) / lines;
        #pragma omp parallel num_threads(threads)
        {
            #pragma omp for schedule(dynamic, 1)
            for (int k = 0; k < n_batches; k++) {
                int start = k * lines, n = (start + lines <= b.n_variants) ? lines : b.n_variants - start;
                assoc_test(tasks[t], b.records + start, n, individuals, b.n_samples, logfact, &out);
            }
            list_decr_writers(&out);                                   /* assoc_runner.c:226-228 */
        }
        char path[512];
        snprintf(path, sizeof path, "%s.%s", prefix, ext[t]);
        FILE *fd = fopen(path, "w");
        assoc_write_output_header(tasks[t], fd);
        assoc_write_output_body(tasks[t], &out, fd);
        fclose(fd);
        if (hpgv_host_sort_output_file(path)) fprintf(stderr, "results could not be sorted\n");   /* assoc_runner.c:255-261 */
        list_free_deep(&out, NULL);
    }
    printf("ASSOC OK\n");
    hpgv_host_shutdown();
    return 0;
}

static int cmd_tdt(const char *batch_path, const char *ped_path, const char *prefix, int threads, int lines) {
    batch_file_t b; ped_t ped;
    if (load_batch(batch_path, &b) || load_ped(ped_path, &ped)) return 2;
    sample_ids_t *ids = sample_ids_new((size_t)b.n_samples);            /* associate_samples_and_positions */
    for (int j = 0; j < b.n_samples; j++) sample_ids_put(ids, b.sample_names[j], j);
    list_t out;
    list_init("output", threads, 0, &out);
    int n_batches = (b.n_variants + lines - 1) / lines, bad = 0;
    #pragma omp parallel num_threads(threads)
    {
        #pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < n_batches; k++) {
            int start = k * lines, n = (start + lines <= b.n_variants) ? lines : b.n_variants - start;
            if (tdt_test(b.records + start, n, ped.families, ped.n_families, ids, &out) != 0) {
                #pragma omp atomic
                bad++;
            }
        }
        list_decr_writers(&out);
    }
    if (bad) { fprintf(stderr, "tdt_test failed: %s\n", hpgv_host_last_error()); return 1; }
    char path[512];
    snprintf(path, sizeof path, "%s.tdt", prefix);
    FILE *fd = fopen(path, "w");
    tdt_write_output_header(fd);
    tdt_write_output_body(&out, fd);
    fclose(fd);
    if (hpgv_host_sort_output_file(path)) fprintf(stderr, "results could not be sorted\n");       /* tdt_runner.c:255-261 */
    list_free_deep(&out, NULL);
    printf("TDT OK\n");
    hpgv_host_shutdown();
    return 0;
}

static int cmd_stats(const char *batch_path, const char *out_path, const char *ped_path) {
    batch_file_t b;
    if (load_batch(batch_path, &b)) return 2;
    individual_t **individuals = NULL;
    sample_ids_t *ids = NULL;
    ped_t ped;
    if (ped_path) {                                       /* stats_runner.c:165-170 */
        if (load_ped(ped_path, &ped)) return 2;
        individuals = (individual_t **)calloc((size_t)b.n_samples + 1, sizeof(void *));
        ids = sample_ids_new((size_t)b.n_samples);
        for (int j = 0; j < b.n_samples; j++) { individuals[j] = find_person(&ped, b.sample_names[j]); sample_ids_put(ids, b.sample_names[j], j); }
    }
    list_t out;
    list_init("output", 1, 0, &out);
    file_stats_t *fs = file_stats_new();
    sample_stats_t **ss = (sample_stats_t **)malloc(sizeof(void *) * (size_t)(b.n_samples + 1));
    for (int j = 0; j < b.n_samples; j++) ss[j] = sample_stats_new(b.sample_names[j]);
    /* stats_runner.c:184-199: the batch is cut into chunks, both functions run on every chunk */
    int chunk = 64, rc = 0;
    for (int start = 0; start < b.n_variants && !rc; start += chunk) {
        int n = start + chunk <= b.n_variants ? chunk : b.n_variants - start;
        rc = get_variants_stats(b.records + start, n, individuals, ids, individuals ? 3 : 0, &out, fs);
        rc |= get_sample_stats(b.records + start, n, individuals, ids, ss, fs);
    }
    if (rc) { fprintf(stderr, "stats failed: %s\n", hpgv_host_last_error()); return 1; }
    list_decr_writers(&out);
    FILE *fd = fopen(out_path, "w");
    list_item_t *it;
    while ((it = list_remove_item(&out))) {
        variant_stats_t *s = (variant_stats_t *)it->data_p;
        fprintf(fd, "V\t%s\t%lu\t%d\t%d\t%d\t%.17g\t%.17g", s->chromosome, s->position, s->num_alleles,
                s->missing_alleles, s->missing_genotypes, s->hw_chi2, s->hw_p_value);
        for (int k = 0; k < s->num_alleles; k++) fprintf(fd, "\t%d", s->alleles_count[k]);
        for (int k = 0; k < s->num_alleles * s->num_alleles; k++) fprintf(fd, "\t%d", s->genotypes_count[k]);
        fprintf(fd, "\n");
        for (int k = 0; k < s->num_phenotypes; k++) {
            const variant_phenotype_stats_t *ps = &s->phenotype_stats[k];
            fprintf(fd, "P\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.17g\t%.17g\t%.9g\n", k, ps->genotypes_count[0], ps->genotypes_count[1],
                    ps->genotypes_count[2], ps->genotypes_count[3], ps->missing_genotypes, ps->missing_alleles,
                    ps->alleles_count[0], ps->alleles_count[1], ps->hw_chi2, ps->hw_p_value, (double)ps->maf);
        }
        variant_stats_free(s);
        list_item_free(it);
    }
    for (int j = 0; j < b.n_samples; j++) {
        fprintf(fd, "S\t%s\t%d\t%d\n", ss[j]->name, ss[j]->missing_genotypes, ss[j]->mendelian_errors);
        sample_stats_free(ss[j]);
    }
    fclose(fd);
    printf("STATS OK variants=%d multiallelic=%d\n", fs->variants_count, fs->multiallelics_count);
    file_stats_free(fs);
    hpgv_host_shutdown();
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "kat")) return cmd_kat();
    if (argc >= 7 && !strcmp(argv[1], "assoc")) return cmd_assoc(argv[2], argv[3], argv[4], atoi(argv[5]), atoi(argv[6]));
    if (argc >= 7 && !strcmp(argv[1], "tdt")) return cmd_tdt(argv[2], argv[3], argv[4], atoi(argv[5]), atoi(argv[6]));
    if (argc >= 4 && !strcmp(argv[1], "stats")) return cmd_stats(argv[2], argv[3], argc >= 5 ? argv[4] : NULL);
    fprintf(stderr, "usage: host_driver kat | assoc|tdt <batch> <ped> <prefix> <threads> <lines> | stats <batch> <out>\n");
    return 2;
}
