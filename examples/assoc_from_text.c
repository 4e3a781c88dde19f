/*
 * assoc_from_text.c -- the whole chi-square association path through the C ABI, from the
 * text of a VCF batch to the reference's TSV (format of assoc_runner.c:295,314-318):
 *
 *   cc -Iinclude examples/assoc_from_text.c -Lhpg-variant_amd/lib -lhpgv -Wl,-rpath,$PWD/hpg-variant_amd/lib -lm
 *   ./a.out data.vcf affected_columns.txt > out.chisq
 *
 * data.vcf: VCF with the #CHROM header line; affected_columns.txt: one 0/1/2 per sample
 * column (0 unaffected, 1 affected, 2 neither), whitespace separated.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpgv.h"

static char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)sz + 1);
    if (fread(b, 1, (size_t)sz, f) != (size_t)sz) { perror(path); exit(2); }
    b[sz] = 0;
    fclose(f);
    *n = (size_t)sz;
    return b;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != HPGV_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hpgv_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s data.vcf conditions.txt\n", argv[0]); return 2; }
    size_t n = 0, nc = 0;
    char *vcf = slurp(argv[1], &n), *cond_txt = slurp(argv[2], &nc);
    /* skip the header: data starts after the line that begins with "#CHROM" */
    char *data = vcf;
    int n_samples = 0;
    while (*data == '#') {
        char *eol = strchr(data, '\n');
        if (!eol) { fprintf(stderr, "no data lines\n"); return 2; }
        if (!strncmp(data, "#CHROM", 6)) {
            int tabs = 0;
            for (char *p = data; p < eol; p++) if (*p == '\t') tabs++;
            n_samples = tabs - 8;
        }
        data = eol + 1;
    }
    if (n_samples <= 0) { fprintf(stderr, "no sample columns\n"); return 2; }
    uint8_t *cond = (uint8_t *)malloc((size_t)n_samples);
    char *p = cond_txt;
    for (int j = 0; j < n_samples; j++) cond[j] = (uint8_t)strtol(p, &p, 10);

    hpgv_ctx *ctx = NULL;
    int rc = hpgv_create(0, &ctx);
    if (rc != HPGV_OK) { fprintf(stderr, "hpgv_create -> %d: %s\n", rc, hpgv_last_error(NULL)); return 1; }
    CHECK(hpgv_set_cohort(ctx, cond, n_samples));

    size_t bytes = n - (size_t)(data - vcf);
    int max_lines = 1;
    for (size_t i = 0; i < bytes; i++) if (data[i] == '\n') max_lines++;
    int32_t *A1 = malloc(sizeof(int32_t) * 4 * (size_t)max_lines), *A2 = A1 + max_lines, *U1 = A2 + max_lines, *U2 = U1 + max_lines;
    double *odds = malloc(sizeof(double) * 3 * (size_t)max_lines), *chisq = odds + max_lines, *pv = chisq + max_lines;
    uint64_t *line_off = malloc(sizeof(uint64_t) * ((size_t)max_lines + 1));
    uint32_t *field_off = malloc(sizeof(uint32_t) * 10 * (size_t)max_lines);
    int n_lines = 0;
    CHECK(hpgv_assoc_text(ctx, HPGV_TASK_CHISQ, data, bytes, max_lines, &n_lines, line_off, field_off, NULL,
                          A1, A2, U1, U2, odds, chisq, pv));

    printf("#CHR\tPOS\tID\tA1\tC_A1\tC_U1\tF_A1\tF_U1\tA2\tC_A2\tC_U2\tF_A2\tF_U2\tOR\tCHISQ\tP-VALUE\n");
    for (int i = 0; i < n_lines; i++) {
        const char *l = data + line_off[i];
        const uint32_t *fo = field_off + 10 * (size_t)i;
        if (fo[5] == 0xFFFFFFFFu) continue;                        /* not a full record */
        int na = A1[i] + A2[i], nu = U1[i] + U2[i];
        printf("%.*s\t%.*s\t%.*s\t%.*s\t%d\t%d\t%6f\t%6f\t%.*s\t%d\t%d\t%6f\t%6f\t%6f\t%6f\t%6f\n",
               (int)(fo[1] - 1 - fo[0]), l + fo[0], (int)(fo[2] - 1 - fo[1]), l + fo[1], (int)(fo[3] - 1 - fo[2]), l + fo[2],
               (int)(fo[4] - 1 - fo[3]), l + fo[3], A1[i], U1[i], na ? (double)A1[i] / na : 0.0, nu ? (double)U1[i] / nu : 0.0,
               (int)(fo[5] - 1 - fo[4]), l + fo[4], A2[i], U2[i], na ? (double)A2[i] / na : 0.0, nu ? (double)U2[i] / nu : 0.0,
               odds[i], chisq[i], pv[i]);
    }
    hpgv_destroy(ctx);
    return 0;
}
