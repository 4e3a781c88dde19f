/* host_format.c -- a batch's result lines: formatting, record filters, the writer thread.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

static int run_batch_stats_arrays(run_batch_t *b) {
    if (!b->stats) return HPGV_OK;
    free(b->c8); free(b->hw); free(b->merr); free(b->midx);
    b->c8 = (int32_t *)malloc(sizeof(int32_t) * 8 * (size_t)b->max_lines);
    b->hw = (double *)malloc(sizeof(double) * 2 * (size_t)b->max_lines);
    b->merr = (int32_t *)calloc((size_t)b->max_lines, sizeof(int32_t));
    b->midx = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->max_lines);
    if (b->n_groups > 0) {
        free(b->gc8); free(b->ghw);
        b->gc8 = (int32_t *)malloc(sizeof(int32_t) * 8 * (size_t)b->max_lines * (size_t)b->n_groups);
        b->ghw = (double *)malloc(sizeof(double) * 2 * (size_t)b->max_lines * (size_t)b->n_groups);
        if (!b->gc8 || !b->ghw) return HPGV_ERR_NOMEM;
    }
    if (!b->smiss) b->smiss = (int32_t *)calloc((size_t)b->n_smiss + 1, sizeof(int32_t));
    if (!b->cerr) b->cerr = (int32_t *)calloc((size_t)b->n_cerr + 1, sizeof(int32_t));
    return (b->c8 && b->hw && b->merr && b->midx && b->smiss && b->cerr) ? HPGV_OK : HPGV_ERR_NOMEM;
}

int run_batch_alloc(run_batch_t *b, size_t cap_bytes, int n_samples, int row_width, int stats, int n_trios, int n_groups) {
    memset(b, 0, sizeof *b);
    b->row_width = row_width;
    b->stats = stats; b->n_smiss = n_samples; b->n_cerr = n_trios; b->n_groups = n_groups;
    size_t min_line = (size_t)(2 * (n_samples > 0 ? n_samples : 1) + 18);
    b->max_lines = (int)(cap_bytes / min_line) + 2;
    b->text = NULL; b->text_cap = cap_bytes + 1;         /* taken by the reader at the batch's first use: the pipeline starts meanwhile */
    b->line_off = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)b->max_lines + 1));
    b->field_off = (uint32_t *)malloc(sizeof(uint32_t) * 10 * (size_t)b->max_lines);
    b->status = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->max_lines);
    b->ints = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)b->max_lines);
    b->dbl = (double *)malloc(sizeof(double) * 3 * (size_t)b->max_lines);
    if (b->row_width > 0) {
        b->rows_cap = (size_t)b->max_lines * (size_t)b->row_width;
        if (!(b->rows = (uint8_t *)malloc(b->rows_cap + 1))) return HPGV_ERR_NOMEM;
    }
    if (run_batch_stats_arrays(b)) return HPGV_ERR_NOMEM;
    return (b->line_off && b->field_off && b->status && b->ints && b->dbl) ? HPGV_OK : HPGV_ERR_NOMEM;
}
/* makes room for `lines` records (short or truncated lines can exceed the estimate) */
int run_batch_reserve(run_batch_t *b, int lines) {
    if (lines <= b->max_lines) return HPGV_OK;
    free(b->line_off); free(b->field_off); free(b->status); free(b->ints); free(b->dbl);
    b->max_lines = lines + lines / 8 + 2;
    b->line_off = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)b->max_lines + 1));
    b->field_off = (uint32_t *)malloc(sizeof(uint32_t) * 10 * (size_t)b->max_lines);
    b->status = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->max_lines);
    b->ints = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)b->max_lines);
    b->dbl = (double *)malloc(sizeof(double) * 3 * (size_t)b->max_lines);
    if (b->row_width > 0) {
        free(b->rows);
        b->rows_cap = (size_t)b->max_lines * (size_t)b->row_width;
        if (!(b->rows = (uint8_t *)malloc(b->rows_cap + 1))) return HPGV_ERR_NOMEM;
    }
    if (run_batch_stats_arrays(b)) return HPGV_ERR_NOMEM;
    return (b->line_off && b->field_off && b->status && b->ints && b->dbl) ? HPGV_OK : HPGV_ERR_NOMEM;
}
void run_batch_free(run_batch_t *b) {
    text_buf_put(b->text, b->text_cap);
    b->text = NULL;
    free(b->line_off); free(b->field_off); free(b->status); free(b->ints); free(b->dbl); free(b->rows);
    free(b->c8); free(b->hw); free(b->merr); free(b->midx); free(b->mtab); free(b->smiss); free(b->cerr); free(b->gc8); free(b->ghw);
}

/* ---- number formatting of the result lines: the characters printf would give, without printf ------------- */
static char *put_u64(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
static char *put_i64(char *p, long v) {
    if (v < 0) { *p++ = '-'; return put_u64(p, (uint64_t)(-(v + 1)) + 1u); }
    return put_u64(p, (uint64_t)v);
}
/* exactly the characters of printf("%6f", x) (the format of assoc_runner.c:314-318 / tdt_runner.c:297-299): the value
 * x = m * 2^e is scaled by 10^6 in 128-bit integer arithmetic and rounded half-to-even on the EXACT binary value, which
 * is what glibc does.  Non-finite values and |x| >= 9e12 go through sprintf (dst needs 320 characters in all). */
int hpgv_host_format_f6(double x, char *dst) {
    char *p = dst;
    uint64_t bits;
    memcpy(&bits, &x, sizeof bits);
    const int bexp = (int)((bits >> 52) & 0x7FF);
    const uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
    if (bexp == 0x7FF || !(fabs(x) < 9e12)) return sprintf(dst, "%6f", x);
    if (bits >> 63) *p++ = '-';
    const uint64_t m = bexp ? (frac | (1ull << 52)) : frac;
    const int e = (bexp ? bexp : 1) - 1075;
    const unsigned __int128 P = (unsigned __int128)m * 1000000u;     /* below 2^73 */
    uint64_t N;
    if (e >= 0) N = (uint64_t)(P << e);                              /* |x| < 9e12: fits */
    else {
        const int sh = -e;
        if (sh >= 75) N = 0;                                         /* P < 2^73 <= half an ulp of the last digit */
        else {
            const unsigned __int128 q = P >> sh, rem = P & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
            N = (uint64_t)q;
            if (rem > half || (rem == half && (N & 1u))) N++;
        }
    }
    p = put_u64(p, N / 1000000u);
    *p++ = '.';
    uint32_t f = (uint32_t)(N % 1000000u);
    for (int k = 5; k >= 0; k--) { p[k] = (char)('0' + f % 10); f /= 10; }
    p += 6;
    *p = 0;
    return (int)(p - dst);
}
#define PUT_F6(p, x) ((p) + hpgv_host_format_f6((x), (p)))
#define PUT_STR(p, s, n) (memcpy((p), (s), (size_t)(n)), (p) + (n))

/* one output line per record, the reference's formats (assoc_runner.c:314-318,332-336; tdt_runner.c:297-299).
 * Returns the number of characters (as snprintf: what the whole line needs). */
static int format_aggregate(char *dst, size_t room, const run_batch_t *b, int i);
static int format_stats_variant(char *dst, size_t room, const run_batch_t *b, int i);

static int format_record(char *dst, size_t room, int kind /* CHI_SQUARE, FISHER, 3 = tdt, 5 = aggregate, 6 = stats */, const run_batch_t *b, int i) {
    if (kind == 5) return format_aggregate(dst, room, b, i);
    if (kind == 6) return format_stats_variant(dst, room, b, i);
    const int m = b->max_lines;
    const uint32_t *fo = b->field_off + 10 * (size_t)i;
    const char *l = b->text + b->line_off[i];
    const int lc = (int)(fo[1] - 1 - fo[0]), li = (int)(fo[3] - 1 - fo[2]);
    const int lr = (int)(fo[4] - 1 - fo[3]), la = (int)(fo[5] - 1 - fo[4]);
    /* the line is put together by hand (the formats, for reference:
     *   tdt    "%s\t%ld\t%s\t%s\t%s\t%d\t%d\t%6f\t%6f\t%6f\n"
     *   chisq  "%s\t%ld\t%s\t%s\t%d\t%d\t%6f\t%6f\t%s\t%d\t%d\t%6f\t%6f\t%6f\t%6f\t%6f\n"     fisher: without the chi-square column);
     * printf's number parsing and its generic %f were a third of the writer's time */
    const size_t worst = (size_t)(lc + li + lr + la) + 24 + 4 * 12 + 7 * 320 + 20;      /* %f of the largest double: 316 characters */
    if (room <= worst) return (int)worst;                            /* the caller grows the buffer and comes again */
    char *p = dst;
    p = PUT_STR(p, l + fo[0], lc); *p++ = '\t';
    p = put_i64(p, atol(l + fo[1])); *p++ = '\t';
    p = PUT_STR(p, l + fo[2], li); *p++ = '\t';
    p = PUT_STR(p, l + fo[3], lr); *p++ = '\t';
    if (kind == 3) {
        const int t1 = b->ints[i], t2 = b->ints[m + i];
        p = PUT_STR(p, l + fo[4], la); *p++ = '\t';
        p = put_i64(p, t1); *p++ = '\t'; p = put_i64(p, t2); *p++ = '\t';
        p = PUT_F6(p, b->dbl[i]); *p++ = '\t'; p = PUT_F6(p, b->dbl[m + i]); *p++ = '\t'; p = PUT_F6(p, b->dbl[2 * m + i]);
        *p++ = '\n'; *p = 0;
        return (int)(p - dst);
    }
    const int A1 = b->ints[i], A2 = b->ints[m + i], U1 = b->ints[2 * m + i], U2 = b->ints[3 * m + i];
    const int na = A1 + A2, nu = U1 + U2;
    const double fa1 = na > 0 ? (double)A1 / na : 0.0, fu1 = nu > 0 ? (double)U1 / nu : 0.0;
    const double fa2 = na > 0 ? (double)A2 / na : 0.0, fu2 = nu > 0 ? (double)U2 / nu : 0.0;
    p = put_i64(p, A1); *p++ = '\t'; p = put_i64(p, U1); *p++ = '\t';
    p = PUT_F6(p, fa1); *p++ = '\t'; p = PUT_F6(p, fu1); *p++ = '\t';
    p = PUT_STR(p, l + fo[4], la); *p++ = '\t';
    p = put_i64(p, A2); *p++ = '\t'; p = put_i64(p, U2); *p++ = '\t';
    p = PUT_F6(p, fa2); *p++ = '\t'; p = PUT_F6(p, fu2); *p++ = '\t';
    p = PUT_F6(p, b->dbl[i]); *p++ = '\t';                            /* odds ratio */
    if (kind == CHI_SQUARE) { p = PUT_F6(p, b->dbl[m + i]); *p++ = '\t'; }
    p = PUT_F6(p, b->dbl[2 * m + i]);
    *p++ = '\n'; *p = 0;
    return (int)(p - dst);
}

/* does line i of the batch give an output record?  Not when it has fewer than CHROM..ALT, when a device-side filter
 * rejected it (--maf / --missing / --mendel), or when it fails --alleles (1 + number of ALT alleles, "." = none) or
 * --quality (QUAL >= minimum; a missing QUAL fails) */
int record_passes(const run_batch_t *b, int i) {
    const uint32_t *fo = b->field_off + 10 * (size_t)i;
    if (fo[5] == 0xFFFFFFFFu) return 0;
    if (b->status[i] & HPGV_LINE_FILTERED) return 0;
    const char *l = b->text + b->line_off[i];
    if (g_filters.num_alleles >= 0) {
        const char *alt = l + fo[4];
        const int la = (int)(fo[5] - 1 - fo[4]);
        int n = (la <= 0 || (la == 1 && alt[0] == '.')) ? 1 : 2;
        for (int k = 0; k < la; k++) if (alt[k] == ',') n++;
        if (n != g_filters.num_alleles) return 0;
    }
    if (g_filters.min_quality >= 0.0) {
        if (fo[6] == 0xFFFFFFFFu) return 0;
        const char *q = l + fo[5];
        if (*q == '.' || *q == '\t' || strtod(q, NULL) < g_filters.min_quality) return 0;
    }
    return 1;
}

/* ---- aggregate / stats: the counters of one record as variant_stats_t holds them (get_variants_stats above) ---- */

void record_counts(const run_batch_t *b, int i, const char *alt, int la, vcounts_t *v) {
    const int32_t *c = b->c8 + 8 * (size_t)i;
    memset(v, 0, sizeof *v);
    int na = (la <= 0 || (la == 1 && alt[0] == '.')) ? 1 : 2;
    for (int k = 0; k < la; k++) if (alt[k] == ',') na++;
    const int32_t *tab = NULL;
    if (b->n_multi > 0) {                                   /* multi-allelic: its 256-bin table (binary search: midx ascends) */
        int lo = 0, hi = b->n_multi < b->multi_cap ? b->n_multi : b->multi_cap;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (b->midx[mid] < i) lo = mid + 1; else hi = mid; }
        if (lo < (b->n_multi < b->multi_cap ? b->n_multi : b->multi_cap) && b->midx[lo] == i) tab = b->mtab + (size_t)lo * 256;
    }
    if (tab)
        for (int code = 0; code < 256; code++) {
            if (!tab[code]) continue;
            const int a1 = code >> 4, a2 = code & 0xF;
            if (a1 != 0xF && a1 + 1 > na) na = a1 + 1;
            if (a2 != 0xF && a2 + 1 > na) na = a2 + 1;
        }
    if (na < 2) na = 2;
    if (na > 15) na = 15;
    v->na = na; v->miss_gt = c[4]; v->miss_al = c[5];
    if (tab) {
        for (int code = 0; code < 256; code++) {
            const int k = tab[code], a1 = code >> 4, a2 = code & 0xF;
            if (!k) continue;
            if (a1 != 0xF) v->ac[a1] += k;
            if (a2 != 0xF) v->ac[a2] += k;
            if (a1 != 0xF && a2 != 0xF) v->gc[a1 * na + a2] += k;
        }
    } else {
        v->ac[0] = c[6]; v->ac[1] = c[7];
        v->gc[0] = c[0]; v->gc[1] = c[1]; v->gc[na] = c[2]; v->gc[na + 1] = c[3];
    }
}

#define APPEND(...) do { int w_ = snprintf(dst + o, o < room ? room - o : 0, __VA_ARGS__); if (w_ < 0) return -1; o += (size_t)w_; } while (0)

/* HPG_GTC / the stats file's genotype column: "i/j:count," for i <= j (both orders of a heterozygote together), then
 * the missing genotypes -- report_variant_genotypes_stats, aggregate_runner.c:376-403 */
static int append_gtc(char *dst, size_t room, size_t o0, const vcounts_t *v) {
    size_t o = o0;
    for (int i = 0; i < v->na; i++)
        for (int j = i; j < v->na; j++)
            APPEND("%d/%d:%d,", i, j, i == j ? v->gc[i * v->na + j] : v->gc[i * v->na + j] + v->gc[j * v->na + i]);
    APPEND("./.:%d", v->miss_gt);
    return (int)(o - o0);
}

int g_aggregate_overwrite = 0;

/* one line of the aggregated VCF (aggregate_runner.c:176-199): the record without its samples, INFO = the original
 * fields (AC / AF / AN dropped when overwriting) followed by [HPG_]AC, [HPG_]AF, [HPG_]AN and HPG_GTC
 * (merge_info_and_stats, :262-365; the reference emits the fields in the order of a hash table) */
static int format_aggregate(char *dst, size_t room, const run_batch_t *b, int i) {
    const uint32_t *fo = b->field_off + 10 * (size_t)i;
    const char *l = b->text + b->line_off[i];
    const char *eol = b->text + b->line_off[i + 1];
    while (eol > l && (eol[-1] == '\n' || eol[-1] == '\r')) eol--;
    size_t o = 0;
    vcounts_t v;
    record_counts(b, i, l + fo[4], (int)(fo[5] - 1 - fo[4]), &v);
    /* CHROM .. ALT as they are; QUAL and FILTER when the line has them */
    APPEND("%.*s", (int)(fo[5] - 1 - fo[0]), l + fo[0]);
    const int has_q = fo[6] != 0xFFFFFFFFu || (fo[5] != 0xFFFFFFFFu && l + fo[5] < eol);
    const char *q = l + fo[5], *qe = fo[6] != 0xFFFFFFFFu ? l + fo[6] - 1 : eol;
    if (has_q && qe > q) APPEND("\t%.*s", (int)(qe - q), q); else APPEND("\t.");
    const char *f = fo[6] != 0xFFFFFFFFu ? l + fo[6] : NULL, *fe = fo[7] != 0xFFFFFFFFu ? l + fo[7] - 1 : eol;
    if (f && fe > f) APPEND("\t%.*s", (int)(fe - f), f); else APPEND("\t.");
    const char *in = fo[7] != 0xFFFFFFFFu ? l + fo[7] : NULL, *ine = fo[8] != 0xFFFFFFFFu ? l + fo[8] - 1 : eol;
    APPEND("\t");
    if (in && ine > in && !(ine - in == 1 && in[0] == '.')) {
        const char *p = in;
        while (p < ine) {                                    /* field by field */
            const char *e = (const char *)memchr(p, ';', (size_t)(ine - p));
            if (!e) e = ine;
            const int klen = (int)((const char *)memchr(p, '=', (size_t)(e - p)) ? (const char *)memchr(p, '=', (size_t)(e - p)) - p : e - p);
            const int drop = g_aggregate_overwrite && klen == 2 && (!strncmp(p, "AC", 2) || !strncmp(p, "AF", 2) || !strncmp(p, "AN", 2));
            if (!drop && e > p) APPEND("%.*s;", (int)(e - p), p);
            p = e + 1;
        }
    }
    const char *pre = g_aggregate_overwrite ? "" : "HPG_";
    int ta = 0;
    for (int k = 0; k < v.na; k++) ta += v.ac[k];
    APPEND("%sAC=", pre);
    for (int k = 1; k < v.na; k++) APPEND(k + 1 < v.na ? "%d," : "%d", v.ac[k]);
    APPEND(";%sAF=", pre);
    for (int k = 1; k < v.na; k++) APPEND(k + 1 < v.na ? "%.3f," : "%.3f", ta ? (float)v.ac[k] / ta : 0.0f);
    APPEND(";%sAN=%d;HPG_GTC=", pre, ta);
    const int w = append_gtc(dst, room, o, &v);
    if (w < 0) return -1;
    o += (size_t)w;
    APPEND("\n");
    return (int)o;
}

/* one line of <prefix>.stats-variants.  The reference's report_vcf_variant_stats lives in hpg-libs (not in the
 * tree): this is the project's own tab-separated rendering of the same variant_stats_t fields. */
static int format_stats_variant(char *dst, size_t room, const run_batch_t *b, int i) {
    const uint32_t *fo = b->field_off + 10 * (size_t)i;
    const char *l = b->text + b->line_off[i];
    size_t o = 0;
    vcounts_t v;
    record_counts(b, i, l + fo[4], (int)(fo[5] - 1 - fo[4]), &v);
    int ta = 0;
    for (int k = 0; k < v.na; k++) ta += v.ac[k];
    APPEND("%.*s\t%ld\t%.*s\t%.*s\t%d\t", (int)(fo[1] - 1 - fo[0]), l + fo[0], atol(l + fo[1]), (int)(fo[4] - 1 - fo[3]), l + fo[3],
           (int)(fo[5] - 1 - fo[4]), l + fo[4], v.na);
    for (int k = 0; k < v.na; k++) APPEND(k + 1 < v.na ? "%d," : "%d\t", v.ac[k]);
    float maf = 1.0f;
    for (int k = 0; k < v.na; k++) {
        const float fr = ta ? (float)v.ac[k] / ta : 0.0f;
        if (fr < maf) maf = fr;
        APPEND(k + 1 < v.na ? "%.4f," : "%.4f\t", fr);
    }
    const int w = append_gtc(dst, room, o, &v);
    if (w < 0) return -1;
    o += (size_t)w;
    APPEND("\t%d\t%d\t%.4f\t%d\t%.6g\t%.6g\n", v.miss_al, v.miss_gt, maf, b->merr[i], b->hw[i], b->hw[b->max_lines + i]);
    return (int)o;
}
#undef APPEND

/* formats the records of a batch by a thread team (one contiguous range of lines and one growing buffer per
 * task), then writes the buffers in line order */
typedef struct { const run_batch_t *b; out_buf_t *bufs; int kind, n, parts, bad; } fmt_job_t;

/* Is the result file in `sort -k1,1h -k2,2n` order as it is written?  (It is for a position-sorted VCF, and reading two
 * million lines back to find that out took 0.09 of a 0.21 s run.)  Neighbouring records with the same CHROM bytes and a
 * growing POS are in order by the definition of the two keys; every other pair -- a new chromosome, equal positions, the
 * seams between the formatting tasks and between batches -- goes through the sort's own comparison of the two lines. */
static int lines_in_order(const char *a, size_t alen, const char *b, size_t blen) {       /* lines without their newline */
    char ta[2048], tb[2048];
    if (alen >= sizeof ta || blen >= sizeof tb) return 0;           /* (not decided here: the file is read back and checked) */
    memcpy(ta, a, alen); ta[alen] = 0; memcpy(tb, b, blen); tb[blen] = 0;
    sort_key_t ka, kb;
    make_key(ta, &ka); make_key(tb, &kb);
    return cmp_keys(&ka, &kb) <= 0;
}
static void order_track_line(order_track_t *o, const char *line, size_t len) {              /* the next line of the file */
    if (o->have && !lines_in_order(o->last, strlen(o->last), line, len)) o->disorder = 1;
}
void order_track_keep(order_track_t *o, const char *line, size_t len) {              /* ... remembered as the last one */
    if (len + 1 > o->cap) { char *n = (char *)realloc(o->last, len + 64); if (!n) { o->disorder = 1; return; } o->last = n; o->cap = len + 64; }
    memcpy(o->last, line, len); o->last[len] = 0; o->have = 1;
}

static void fmt_task(void *v, int t) {
    fmt_job_t *j = (fmt_job_t *)v;
    const run_batch_t *b = j->b;
    out_buf_t *o = &j->bufs[t];
    o->len = 0;
    o->disorder = 0;
    const int lo = (int)((long)j->n * t / j->parts), hi = (int)((long)j->n * (t + 1) / j->parts);
    const char *pc = NULL; size_t pclen = 0, prev_off = 0, prev_len = 0; unsigned long ppos = 0; int have_prev = 0;
    for (int i = lo; i < hi; i++) {
        if (!record_passes(b, i)) continue;
        for (;;) {
            int need = o->cap > o->len ? format_record(o->p + o->len, o->cap - o->len, j->kind, b, i) : -2;
            if (need >= 0 && (size_t)need < o->cap - o->len) {
                if (j->kind < 4 && b->field_off) {                  /* results that are sorted afterwards: in order so far? */
                    const uint32_t *fo = b->field_off + 10 * (size_t)i;
                    const char *l = b->text + b->line_off[i];
                    const char *c = l + fo[0]; const size_t clen = fo[1] - 1 - fo[0];
                    unsigned long pos = 0; int digits = 0;
                    for (const char *q = l + fo[1]; *q >= '0' && *q <= '9' && digits < 18; q++, digits++) pos = pos * 10 + (unsigned long)(*q - '0');
                    if (have_prev && !(digits && clen == pclen && !memcmp(c, pc, clen) && pos > ppos)
                        && !lines_in_order(o->p + prev_off, prev_len, o->p + o->len, (size_t)need - 1)) o->disorder = 1;
                    pc = c; pclen = clen; ppos = digits ? pos : 0; have_prev = digits ? 1 : 2;
                    if (!digits) { pclen = (size_t)-1; }             /* (a POS that is no number: the next pair is compared as lines) */
                    prev_off = o->len; prev_len = (size_t)need - 1;
                }
                o->len += (size_t)need; break;
            }
            size_t nc = o->cap ? o->cap * 2 : (size_t)1 << 16;
            if (need > 0 && nc < o->len + (size_t)need + 1) nc = o->len + (size_t)need + 1;
            char *np = need == -1 ? NULL : (char *)realloc(o->p, nc);
            if (!np) { __atomic_store_n(&j->bad, 1, __ATOMIC_RELAXED); return; }
            o->p = np; o->cap = nc;
        }
    }
}

double g_write_split[2];                         /* of the last run's write stage: formatting, writing (HPGV_RUN_TRACE) */

/* the formatted lines of one batch are written by a thread of their own while the next batch is formatted (a run over 2M
 * short records spent 0.02 s of its 0.07 s write stage in fwrite): one set of buffers is being written, the other filled */
static void *file_writer_main(void *v) {
    file_writer_t *w = (file_writer_t *)v;
    pthread_mutex_lock(&w->mu);
    for (;;) {
        while (!w->parts && !w->stop) pthread_cond_wait(&w->cv, &w->mu);
        if (!w->parts) break;
        out_buf_t *bufs = w->bufs; const int parts = w->parts;
        pthread_mutex_unlock(&w->mu);
        int bad = 0;
        for (int t = 0; t < parts && !bad; t++)
            if (bufs[t].len && fwrite(bufs[t].p, 1, bufs[t].len, w->fd) != bufs[t].len) bad = 1;
        pthread_mutex_lock(&w->mu);
        if (bad) w->bad = 1;
        w->parts = 0; w->busy = 0;
        pthread_cond_broadcast(&w->cv);
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}
int file_writer_start(file_writer_t *w, FILE *fd) {
    memset(w, 0, sizeof *w);
    w->fd = fd;
    pthread_mutex_init(&w->mu, NULL); pthread_cond_init(&w->cv, NULL);
    w->started = pthread_create(&w->th, NULL, file_writer_main, w) == 0;
    if (!w->started) { pthread_mutex_destroy(&w->mu); pthread_cond_destroy(&w->cv); }
    return w->started;
}
/* waits until the set handed over before is on its way to the file; 1 = a write has failed */
static int file_writer_idle(file_writer_t *w) {
    pthread_mutex_lock(&w->mu);
    while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
    const int bad = w->bad;
    pthread_mutex_unlock(&w->mu);
    return bad;
}
static int file_writer_submit(file_writer_t *w, out_buf_t *bufs, int parts) {
    if (file_writer_idle(w)) return 1;
    pthread_mutex_lock(&w->mu);
    w->bufs = bufs; w->parts = parts; w->busy = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    return 0;
}
int file_writer_stop(file_writer_t *w) {          /* everything handed over is written when this returns */
    if (!w->started) return 0;
    const int bad = file_writer_idle(w);
    pthread_mutex_lock(&w->mu); w->stop = 1; pthread_cond_broadcast(&w->cv); pthread_mutex_unlock(&w->mu);
    pthread_join(w->th, NULL);
    pthread_mutex_destroy(&w->mu); pthread_cond_destroy(&w->cv);
    w->started = 0;
    return bad;
}

/* fw (may be NULL: written here) takes the formatted set; the caller alternates between two sets of n_bufs buffers */
int write_batch(FILE *fd, int kind, const run_batch_t *b, out_buf_t *bufs, int n_bufs, io_pool_t *pool, order_track_t *ord, file_writer_t *fw) {
    if (kind == 4) {                                     /* vcf2epi: the rows of the records, in line order (dataset_creator.c:196-199) */
        const int n = b->n_lines < b->max_lines ? b->n_lines : b->max_lines;
        const size_t w = (size_t)b->row_width;
        int i = 0;
        while (i < n) {                                  /* runs of consecutive records go out in one write */
            while (i < n && !record_passes(b, i)) i++;
            int e = i;
            while (e < n && record_passes(b, e)) e++;
            if (e > i && w && fwrite(b->rows + (size_t)i * w, w, (size_t)(e - i), fd) != (size_t)(e - i)) return 1;
            i = e;
        }
        return 0;
    }
    fmt_job_t j;
    j.b = b; j.bufs = bufs; j.kind = kind; j.bad = 0;
    j.n = b->n_lines < b->max_lines ? b->n_lines : b->max_lines;
    j.parts = j.n >= 2048 ? n_bufs : 1;
    const double t_f0 = now_s();
    pool_run(pool, fmt_task, &j, j.parts);
    g_write_split[0] += now_s() - t_f0;
    if (j.bad) return 1;
    const double t_w0 = now_s();
    for (int t = 0; t < j.parts; t++) {
        if (!bufs[t].len) continue;
        if (ord && kind < 4) {                                       /* the seam before this task's lines, and its own verdict */
            const char *first_end = (const char *)memchr(bufs[t].p, '\n', bufs[t].len);
            const char *lastl = (const char *)memrchr(bufs[t].p, '\n', bufs[t].len - 1);
            lastl = lastl ? lastl + 1 : bufs[t].p;
            if (first_end) order_track_line(ord, bufs[t].p, (size_t)(first_end - bufs[t].p));
            if (bufs[t].disorder) ord->disorder = 1;
            order_track_keep(ord, lastl, (size_t)(bufs[t].p + bufs[t].len - 1 - lastl));
        }
        if (!fw && fwrite(bufs[t].p, 1, bufs[t].len, fd) != bufs[t].len) return 1;
    }
    if (fw && file_writer_submit(fw, bufs, j.parts)) return 1;
    g_write_split[1] += now_s() - t_w0;
    return 0;
}
