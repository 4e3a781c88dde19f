/* host_output.c -- the reference's result writers (assoc_runner.c:292-342, tdt_runner.c:286-304) and `sort -k1,1h -k2,2n` in process.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* writers: the reference's exact formats                                     */
/* ------------------------------------------------------------------------ */

void assoc_write_output_header(enum ASSOC_task task, FILE *fd) {
    if (task == CHI_SQUARE)
        fprintf(fd, "#CHR\tPOS\tID\tA1\tC_A1\tC_U1\tF_A1\tF_U1\tA2\tC_A2\tC_U2\tF_A2\tF_U2\tOR\tCHISQ\tP-VALUE\n");
    else if (task == FISHER)
        fprintf(fd, "#CHR\tPOS\tID\tA1\tC_A1\tC_U1\tF_A1\tF_U1\tA2\tC_A2\tC_U2\tF_A2\tF_U2\tOR\tP-VALUE\n");
}

void assoc_write_output_body(enum ASSOC_task task, list_t *output_list, FILE *fd) {
    list_item_t *item;
    while ((item = list_remove_item(output_list))) {
        if (task == CHI_SQUARE) {
            assoc_basic_result_t *r = (assoc_basic_result_t *)item->data_p;
            int na = r->affected1 + r->affected2, nu = r->unaffected1 + r->unaffected2;   /* assoc_runner.c:309-312 */
            double fa1 = na > 0 ? (double)r->affected1 / na : 0.0f, fu1 = nu > 0 ? (double)r->unaffected1 / nu : 0.0f;
            double fa2 = na > 0 ? (double)r->affected2 / na : 0.0f, fu2 = nu > 0 ? (double)r->unaffected2 / nu : 0.0f;
            fprintf(fd, "%s\t%ld\t%s\t%s\t%d\t%d\t%6f\t%6f\t%s\t%d\t%d\t%6f\t%6f\t%6f\t%6f\t%6f\n",
                    r->chromosome, (long)r->position, r->id, r->reference, r->affected1, r->unaffected1, fa1, fu1,
                    r->alternate, r->affected2, r->unaffected2, fa2, fu2, r->odds_ratio, r->chi_square, r->p_value);
            assoc_basic_result_free(r);
        } else {
            assoc_fisher_result_t *r = (assoc_fisher_result_t *)item->data_p;
            int na = r->affected1 + r->affected2, nu = r->unaffected1 + r->unaffected2;
            double fa1 = na > 0 ? (double)r->affected1 / na : 0.0f, fu1 = nu > 0 ? (double)r->unaffected1 / nu : 0.0f;
            double fa2 = na > 0 ? (double)r->affected2 / na : 0.0f, fu2 = nu > 0 ? (double)r->unaffected2 / nu : 0.0f;
            fprintf(fd, "%s\t%ld\t%s\t%s\t%d\t%d\t%6f\t%6f\t%s\t%d\t%d\t%6f\t%6f\t%6f\t%6f\n",
                    r->chromosome, (long)r->position, r->id, r->reference, r->affected1, r->unaffected1, fa1, fu1,
                    r->alternate, r->affected2, r->unaffected2, fa2, fu2, r->odds_ratio, r->p_value);
            assoc_fisher_result_free(r);
        }
        list_item_free(item);
    }
}

void tdt_write_output_header(FILE *fd) { fprintf(fd, "#CHR\tPOS\tID\tA1\tA2\tT\tU\tOR\tCHISQ\tP-VALUE\n"); }

void tdt_write_output_body(list_t *output_list, FILE *fd) {
    list_item_t *item;
    while ((item = list_remove_item(output_list))) {
        tdt_result_t *r = (tdt_result_t *)item->data_p;
        fprintf(fd, "%s\t%ld\t%s\t%s\t%s\t%d\t%d\t%6f\t%6f\t%6f\n", r->chromosome, (long)r->position, r->id,
                r->reference, r->alternate, r->t1, r->t2, r->odds_ratio, r->chi_square, r->p_value);
        tdt_result_free(r);
        list_item_free(item);
    }
}

/* ------------------------------------------------------------------------ */
/* result ordering: `sort -k1,1h -k2,2n` in process                          */
/* ------------------------------------------------------------------------ */


static int is_blank(char c) { return c == ' ' || c == '\t'; }

/* GNU sort numeric token at p: optional blanks, '-', digits, optional fraction.  Text that is
 * not a number counts as 0.  Returns the position after the token in *end. */
static long double parse_number(const char *p, int *neg, const char **end, int *digits) {
    while (is_blank(*p)) p++;
    *neg = 0;
    if (*p == '-') { *neg = 1; p++; }
    long double v = 0.0L;
    *digits = (*p >= '0' && *p <= '9');
    while (*p >= '0' && *p <= '9') { v = v * 10.0L + (*p - '0'); p++; }
    if (*p == '.') {
        long double scale = 0.1L;
        p++;
        while (*p >= '0' && *p <= '9') { v += scale * (*p - '0'); scale *= 0.1L; p++; }
    }
    if (v == 0.0L) *neg = 0;
    *end = p;
    return v;
}

void make_key(const char *line, sort_key_t *k) {
    k->line = line;
    const char *e;
    int digits;
    k->v1 = parse_number(line, &k->neg1, &e, &digits);         /* -k1,1h */
    static const char units[] = "KMGTPEZY";
    k->order1 = 0;
    if (digits) {                                              /* a unit suffix only counts after digits */
        if (*e == 'k') k->order1 = 1;
        else { const char *u = *e ? strchr(units, *e) : NULL; if (u) k->order1 = (int)(u - units) + 1; }
    }
    /* field 2 = after the first field (non-blanks) of the line; sort's fields carry their leading blanks */
    const char *p = line;
    while (is_blank(*p)) p++;
    while (*p && !is_blank(*p)) p++;
    k->v2 = parse_number(p, &k->neg2, &e, &digits);            /* -k2,2n */
}

static int cmp_signed(int na, long double a, int nb, long double b) {
    if (na != nb) return na ? -1 : 1;
    if (a == b) return 0;
    int r = a < b ? -1 : 1;
    return na ? -r : r;
}

int cmp_keys(const void *pa, const void *pb) {
    const sort_key_t *a = (const sort_key_t *)pa, *b = (const sort_key_t *)pb;
    /* human numeric: sign, then unit magnitude, then value */
    if (a->neg1 != b->neg1) return a->neg1 ? -1 : 1;
    if (a->order1 != b->order1) { int r = a->order1 < b->order1 ? -1 : 1; return a->neg1 ? -r : r; }
    int r = cmp_signed(a->neg1, a->v1, b->neg1, b->v1);
    if (r) return r;
    r = cmp_signed(a->neg2, a->v2, b->neg2, b->v2);
    if (r) return r;
    return strcmp(a->line, b->line);                            /* last resort: whole line, bytewise */
}

typedef struct {
    sort_key_t *keys, *src, *dst; size_t k; int parts, width;
    int unsorted[64];
} sort_job_t;

static void sort_task_keys(void *v, int t) {           /* keys of one range, and whether the range (and its seam) is in order */
    sort_job_t *j = (sort_job_t *)v;
    const size_t lo = j->k * (size_t)t / (size_t)j->parts, hi = j->k * (size_t)(t + 1) / (size_t)j->parts;
    for (size_t i = lo; i < hi; i++) make_key(j->keys[i].line, &j->keys[i]);
    int bad = 0;
    for (size_t i = lo + 1; i < hi && !bad; i++) bad = cmp_keys(&j->keys[i - 1], &j->keys[i]) > 0;
    if (!bad && hi < j->k && hi > lo) {                 /* seam with the next range: its first key is made here too */
        sort_key_t nxt;
        make_key(j->keys[hi].line, &nxt);
        bad = cmp_keys(&j->keys[hi - 1], &nxt) > 0;
    }
    j->unsorted[t] = bad;
}
static void sort_task_runs(void *v, int t) {
    sort_job_t *j = (sort_job_t *)v;
    const size_t lo = j->k * (size_t)t / (size_t)j->parts, hi = j->k * (size_t)(t + 1) / (size_t)j->parts;
    qsort(j->keys + lo, hi - lo, sizeof *j->keys, cmp_keys);
}
static void sort_task_merge(void *v, int task) {
    sort_job_t *j = (sort_job_t *)v;
    const int t = task * 2 * j->width, parts = j->parts, width = j->width;
    const sort_key_t *src = j->src;
    sort_key_t *dst = j->dst;
    const size_t lo = j->k * (size_t)t / (size_t)parts;
    const size_t mid = j->k * (size_t)(t + width < parts ? t + width : parts) / (size_t)parts;
    const size_t hi = j->k * (size_t)(t + 2 * width < parts ? t + 2 * width : parts) / (size_t)parts;
    size_t a = lo, b = mid, o = lo;
    while (a < mid && b < hi) dst[o++] = cmp_keys(&src[b], &src[a]) < 0 ? src[b++] : src[a++];
    while (a < mid) dst[o++] = src[a++];
    while (b < hi) dst[o++] = src[b++];
}

int hpgv_host_sort_output_file(const char *path) {
    if (!g_ctx) host_env_read();                                   /* called on its own (inside a run the run has read it) */
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *blob = (char *)malloc((size_t)sz + 2);
    if (!blob || fread(blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(blob); return 1; }
    fclose(f);
    blob[sz] = 0;
    size_t n = 0;
    for (const char *q = blob, *end = blob + sz; q < end;) {          /* memchr runs at memory speed; a byte loop took 0.1 s per 100 MB */
        const char *e = (const char *)memchr(q, '\n', (size_t)(end - q));
        if (!e) break;
        n++; q = e + 1;
    }
    if (sz > 0 && blob[sz - 1] != '\n') n++;
    sort_key_t *keys = (sort_key_t *)malloc((n + 1) * sizeof *keys);
    if (!keys) { free(blob); return 1; }
    size_t k = 0;
    char *p = blob;
    while (k < n) {                                    /* line starts (memchr runs at memory speed) */
        keys[k++].line = p;
        char *e = (char *)memchr(p, '\n', (size_t)(blob + sz - p));
        if (!e) break;
        *e = 0;
        p = e + 1;
    }
    sort_job_t job;
    memset(&job, 0, sizeof job);
    job.keys = keys; job.k = k;
    io_pool_t pool;
    const int parts = k > 65536 ? default_io_threads() : 1;
    pool_init(&pool, parts);
    job.parts = parts;
    pool_run(&pool, sort_task_keys, &job, parts);
    int sorted = 1;
    for (int t = 0; t < parts; t++) sorted = sorted && !job.unsorted[t];
    if (sorted) { pool_destroy(&pool); free(keys); free(blob); return 0; }   /* a position-sorted VCF gives a sorted result file: leave it */
    /* sorted runs by the team, then pairwise merges */
    sort_key_t *tmpk = parts > 1 ? (sort_key_t *)malloc((k + 1) * sizeof *keys) : NULL;
    if (!tmpk && parts > 1) { job.parts = 1; }
    pool_run(&pool, sort_task_runs, &job, job.parts);
    job.src = keys; job.dst = tmpk;
    for (job.width = 1; job.width < job.parts; job.width *= 2) {
        pool_run(&pool, sort_task_merge, &job, (job.parts + 2 * job.width - 1) / (2 * job.width));
        sort_key_t *sw = job.src; job.src = job.dst; job.dst = sw;
    }
    if (job.src != keys) memcpy(keys, job.src, k * sizeof *keys);
    free(tmpk);
    pool_destroy(&pool);
    size_t len = strlen(path);
    char *tmp = (char *)malloc(len + 5);
    if (!tmp) { free(keys); free(blob); return 1; }
    memcpy(tmp, path, len);
    memcpy(tmp + len, ".tmp", 5);
    FILE *o = fopen(tmp, "wb");
    int rc = o ? 0 : 1;
    if (o) {
        setvbuf(o, NULL, _IOFBF, 1u << 20);
        for (size_t i = 0; i < k; i++) { fputs(keys[i].line, o); fputc('\n', o); }
        if (fclose(o) != 0 || rename(tmp, path) != 0) rc = 1;
    }
    free(tmp); free(keys); free(blob);
    return rc;
}
