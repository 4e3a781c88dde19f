/*
 * host_cpu_check.c -- CPU-only checks of the host mirror's own logic (no engine
 * call): containers, the thread-safe result list, and GT-text staging.  Built
 * with -fsanitize=address,undefined by tests/test_host_logic_cpu.py.
 *
 *   host_cpu_check containers         self-checking, prints PASS/FAIL lines
 *   host_cpu_check copy <in> <out> <batch_bytes> [vcf]   the runners' line reader (plain / gzip / BGZF input)
 *   host_cpu_check stage <file>       file: "N V" then V lines "chrom format s1..sN";
 *                                     prints per line: is_x then the N code bytes (hex), strict and lax
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hpgv_host.h"

static int failures = 0;
#define CHECK(c, m) do { if (!(c)) { printf("FAIL %s\n", m); failures++; } else printf("PASS %s\n", m); } while (0)

static void *producer(void *arg) {
    list_t *l = (list_t *)arg;
    for (int i = 0; i < 5000; i++) {
        int *v = (int *)malloc(sizeof(int));
        *v = i;
        list_insert_item(list_item_new(i & 3, 0, v), l);
    }
    list_decr_writers(l);
    return NULL;
}

static int cmd_containers(void) {
    /* array_list grows past its initial capacity */
    array_list_t *al = array_list_new(2);
    for (long i = 0; i < 1000; i++) array_list_insert((void *)(i + 1), al);
    int ok = al->size == 1000;
    for (long i = 0; i < 1000; i++) ok &= ((long)array_list_get((size_t)i, al) == i + 1);
    CHECK(ok && array_list_get(1000, al) == NULL, "array_list: 1000 inserts, bounds-checked get");
    array_list_free(al, NULL);

    /* sample_ids: many names, rehashing, absent keys */
    sample_ids_t *ids = sample_ids_new(4);
    char (*names)[16] = malloc(20000 * 16);
    for (int i = 0; i < 20000; i++) { snprintf(names[i], 16, "NA%05d", i); sample_ids_put(ids, names[i], i); }
    ok = ids->size == 20000;
    for (int i = 0; i < 20000; i += 7) ok &= sample_ids_get(ids, names[i]) == i;
    ok &= sample_ids_get(ids, "absent") == -1;
    sample_ids_put(ids, names[5], 99);
    ok &= sample_ids_get(ids, names[5]) == 99 && ids->size == 20000;
    CHECK(ok, "sample_ids: 20000 names, lookups, overwrite, absent key");
    sample_ids_free(ids);
    free(names);

    /* list_t: 4 producers, 1 consumer, consumer ends when the writers are gone (assoc_runner.c:306) */
    list_t l;
    list_init("output", 4, 0, &l);
    pthread_t th[4];
    for (int t = 0; t < 4; t++) pthread_create(&th[t], NULL, producer, &l);
    long got = 0, sum = 0;
    list_item_t *it;
    while ((it = list_remove_item(&l))) { got++; sum += *(int *)it->data_p; free(it->data_p); list_item_free(it); }
    for (int t = 0; t < 4; t++) pthread_join(th[t], NULL);
    CHECK(got == 20000 && sum == 4L * (4999L * 5000L / 2), "list_t: 4 writers x 5000 items drained, then NULL");
    list_free_deep(&l, NULL);

    /* families / individuals */
    family_t *f = family_new("F1");
    individual_t *fa = individual_new("fa", -1, MALE, UNAFFECTED, NULL, NULL, f);
    individual_t *mo = individual_new("mo", -1, FEMALE, UNAFFECTED, NULL, NULL, f);
    individual_t *ch = individual_new("ch", -1, MALE, AFFECTED, fa, mo, f);
    family_set_parent(fa, f); family_set_parent(mo, f); family_add_child(ch, f);
    CHECK(f->founders->size == 2 && f->members->size == 1 && ch->father == fa && ch->mother == mo, "family construction");
    individual_free(fa); individual_free(mo); individual_free(ch); family_free(f);

    double *lf = init_logarithm_array(100);
    CHECK(lf && lf[0] == 0.0 && lf[1] == 0.0 && lf[5] > 4.787 && lf[5] < 4.788, "init_logarithm_array: ln(5!) = 4.7875");
    free(lf);
    printf("%s\n", failures ? "CONTAINERS FAILED" : "CONTAINERS OK");
    return failures ? 1 : 0;
}

static char *next_tok(char **p) {
    char *s = *p;
    while (*s == ' ' || *s == '\n' || *s == '\r') s++;
    if (!*s) return NULL;
    char *e = s;
    while (*e && *e != ' ' && *e != '\n' && *e != '\r') e++;
    if (*e) { *e = 0; e++; }
    *p = e;
    return s;
}

static int cmd_stage(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *blob = (char *)malloc((size_t)sz + 1);
    if (fread(blob, 1, (size_t)sz, f) != (size_t)sz) return 2;
    blob[sz] = 0;
    fclose(f);
    char *p = blob;
    int n = atoi(next_tok(&p)), v = atoi(next_tok(&p));
    vcf_record_t **recs = (vcf_record_t **)malloc(sizeof(void *) * (size_t)(v + 1));
    for (int i = 0; i < v; i++) {
        recs[i] = vcf_record_new();
        char *chrom = next_tok(&p), *fmt = next_tok(&p);
        set_vcf_record_chromosome(chrom, (int)strlen(chrom), recs[i]);
        set_vcf_record_format(fmt, (int)strlen(fmt), recs[i]);
        for (int j = 0; j < n; j++) {
            char *s = next_tok(&p);
            if (!strcmp(s, "<empty>")) s[0] = 0;
            array_list_insert(s, recs[i]->samples);
        }
    }
    uint8_t *gt = (uint8_t *)malloc((size_t)v * (size_t)(n > 0 ? n : 1)), *isx = (uint8_t *)malloc((size_t)v + 1);
    for (int strict = 1; strict >= 0; strict--) {
        hpgv_host_stage_records(recs, v, n, strict, gt, isx);
        for (int i = 0; i < v; i++) {
            printf("%d %d", strict, isx[i]);
            for (int j = 0; j < n; j++) printf(" %02x", gt[(size_t)i * n + j]);
            printf("\n");
        }
    }
    for (int i = 0; i < v; i++) vcf_record_free(recs[i]);
    free(recs); free(gt); free(isx); free(blob);
    return 0;
}


/* the runners' DEFLATE decoder against zlib: valid streams of several kinds must decode to the original, corrupted
 * ones must be refused or decoded without touching memory outside the buffers (this binary runs under ASan) */
#include <zlib.h>
static int deflate_raw(const unsigned char *src, size_t n, int level, int strategy, unsigned char *dst, size_t cap, size_t *out_n) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) return 1;
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n; zs.next_out = dst; zs.avail_out = (uInt)cap;
    int rc = deflate(&zs, Z_FINISH);
    *out_n = zs.total_out;
    deflateEnd(&zs);
    return rc != Z_STREAM_END;
}
static int cmd_inflate(void) {
    enum { MAXN = 1 << 20 };
    unsigned char *src = malloc(MAXN), *comp = malloc(MAXN * 2 + 1024), *out = malloc(MAXN + 16);
    unsigned seed = 12345u;
    int n_ok = 0, n_refused = 0, n_corrupt = 0;
    const size_t sizes[] = {0, 1, 2, 7, 100, 4096, 65280, 65281, 300000, MAXN};
    for (int kind = 0; kind < 5; kind++)
        for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; si++) {
            const size_t n = sizes[si];
            for (size_t i = 0; i < n; i++) {
                seed = seed * 1664525u + 1013904223u;
                const unsigned r = seed >> 16;
                src[i] = kind == 0 ? (unsigned char)"0/0\t0/1\t1/1\t./.\t0|1\n"[(i * 7 + (r % 3)) % 20]      /* genotype text */
                       : kind == 1 ? (unsigned char)r                                                         /* incompressible: stored blocks */
                       : kind == 2 ? 0                                                                        /* one long run: distance 1 */
                       : kind == 3 ? (unsigned char)("ACGT\t.\tPASS\n"[r % 12])                               /* small alphabet */
                       : (unsigned char)((i % 251) ^ (r % 5 == 0 ? r : 0));                                   /* long matches with noise */
            }
            for (int level = 1; level <= 9; level += 4)
                for (int strategy = 0; strategy <= 4; strategy++) {     /* default, filtered, huffman only, rle, fixed */
                    size_t cn = 0;
                    if (deflate_raw(src, n, level, strategy, comp, MAXN * 2 + 1024, &cn)) { printf("deflate failed\n"); return 1; }
                    memset(out, 0xEE, n + 16);
                    const int rc = hpgv_host_inflate_raw(comp, cn, out, n);
                    if (rc != 0) { printf("valid stream refused: kind %d size %zu level %d strategy %d\n", kind, n, level, strategy); return 1; }
                    if (memcmp(out, src, n) != 0) { printf("wrong bytes: kind %d size %zu level %d strategy %d\n", kind, n, level, strategy); return 1; }
                    for (int k = 0; k < 16; k++) if (out[n + k] != 0xEE) { printf("wrote past the end\n"); return 1; }
                    n_ok++;
                    /* wrong output size */
                    if (n > 0 && hpgv_host_inflate_raw(comp, cn, out, n - 1) == 0) { printf("short output accepted\n"); return 1; }
                    if (hpgv_host_inflate_raw(comp, cn, out, n + 1) == 0) { printf("long output accepted\n"); return 1; }
                    /* corrupted and truncated copies: any answer, no stray access */
                    if (cn > 4 && n <= 65281) {
                        unsigned char *bad = malloc(cn);
                        for (int t = 0; t < 24; t++) {
                            memcpy(bad, comp, cn);
                            seed = seed * 1664525u + 1013904223u;
                            bad[(seed >> 8) % cn] ^= (unsigned char)(1u << ((seed >> 4) & 7));
                            if (hpgv_host_inflate_raw(bad, t & 1 ? cn : cn - 1 - (seed >> 20) % (cn - 1), out, n) != 0) n_refused++;
                            n_corrupt++;
                        }
                        free(bad);
                    }
                }
        }
    printf("inflate ok: %d streams, %d of %d damaged ones refused\n", n_ok, n_refused, n_corrupt);
    free(src); free(comp); free(out);
    return 0;
}

/* sample strings that end exactly at the end of a mapped page whose successor is not accessible: the staging's four-byte
 * loads must not reach over the edge */
#include <sys/mman.h>
static int cmd_pageedge(void) {
    const size_t pg = 4096;
    char *m = (char *)mmap(NULL, 2 * pg, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return 2;
    if (mprotect(m + pg, pg, PROT_NONE)) return 2;
    static const char *const tails[4] = {"0/1", "1/", "1", ""};          /* 4, 3, 2, 1 bytes with their NUL */
    static const uint8_t expect_strict[4] = {0x01, 0xFF, 0xFF, 0xFF}, expect_loose[4] = {0x01, 0x1F, 0x1F, 0xFF};
    int bad = 0;
    for (int t = 0; t < 4; t++) {
        const size_t len = strlen(tails[t]) + 1;
        memset(m, 'Z', pg);
        char *edge = m + pg - len;
        memcpy(edge, tails[t], len);
        vcf_record_t *rec = vcf_record_new();
        set_vcf_record_chromosome("1", 1, rec);
        set_vcf_record_format("GT", 2, rec);
        for (int j = 0; j < 19; j++) array_list_insert(j % 3 == 1 ? (void *)edge : (void *)"1/1", rec->samples);
        uint8_t gt[19], isx[1];
        for (int strict = 1; strict >= 0; strict--) {
            hpgv_host_stage_records(&rec, 1, 19, strict, gt, isx);
            for (int j = 0; j < 19; j++) {
                const uint8_t want = j % 3 == 1 ? (strict ? expect_strict[t] : expect_loose[t]) : 0x11;
                if (gt[j] != want) { printf("tail '%s' strict %d sample %d: %02x, expected %02x\n", tails[t], strict, j, gt[j], want); bad++; }
            }
        }
        vcf_record_free(rec);
    }
    munmap(m, 2 * pg);
    printf("%s\n", bad ? "PAGEEDGE FAILED" : "PAGEEDGE OK");
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "containers")) return cmd_containers();
    if (argc >= 2 && !strcmp(argv[1], "pageedge")) return cmd_pageedge();
    if (argc >= 2 && !strcmp(argv[1], "inflate")) return cmd_inflate();
    if (argc >= 3 && !strcmp(argv[1], "stage")) return cmd_stage(argv[2]);
    if (argc >= 3 && !strcmp(argv[1], "sort")) return hpgv_host_sort_output_file(argv[2]);
    if (argc >= 5 && !strcmp(argv[1], "copy")) {
        long nb = 0;
        int rc = hpgv_host_copy_lines(argv[2], argv[3], (size_t)atol(argv[4]), argc >= 6 && !strcmp(argv[5], "vcf"), &nb);
        printf("%ld\n", nb);
        return rc;
    }
    return 2;
}
