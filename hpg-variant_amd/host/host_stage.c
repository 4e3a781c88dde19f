/* host_stage.c -- sample strings -> one byte per genotype: what assoc.c:45-57 / tdt.c:97-108 do per genotype with strdup + get_alleles.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* text -> HPGV8                                                              */
/* ------------------------------------------------------------------------ */

/* position of `field` among the ':'-separated FORMAT keys (assoc.c:46, tdt.c:47) */
int get_field_position_in_format(const char *field, char *format) {
    size_t flen = strlen(field);
    int pos = 0;
    const char *p = format;
    for (;;) {
        const char *e = strchr(p, ':');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == flen && !memcmp(p, field, flen)) return pos;
        if (!e) return -1;
        p = e + 1; pos++;
    }
}

/* the reference calls this on a strdup'ed sample (assoc.c:52-53); here the
 * string is only read.  Return codes: 0 ok, 1 first allele missing, 2 second,
 * 3 both, 4 haploid (DESIGN.md "Genotype text"). */
int get_alleles(char *sample, int genotype_position, int *allele1, int *allele2) {
    const char *p = sample;
    for (int i = 0; i < genotype_position; i++) {
        p = strchr(p, ':');
        if (!p) { *allele1 = *allele2 = -1; return 3; }
        p++;
    }
    const char *end = p;
    while (*end && *end != ':') end++;
    const char *sep = p;
    while (sep < end && *sep != '/' && *sep != '|') sep++;
    int ret = 0;
    if (sep == p || (sep - p == 1 && *p == '.')) { *allele1 = -1; ret += 1; }
    else *allele1 = atoi(p);
    if (sep == end) { *allele2 = -1; return ret == 0 ? 4 : 3; }
    const char *q = sep + 1;
    if (q == end || (end - q == 1 && *q == '.')) { *allele2 = -1; ret += 2; }
    else *allele2 = atoi(q);
    return ret;
}

/* the general case of one sample string: any FORMAT position, multi-digit alleles, haploid and half-missing calls */
static uint8_t encode_gt_general(const char *s, int gt_position, int strict) {
    int a1, a2;
    int st = get_alleles((char *)s, gt_position, &a1, &a2);
    if (strict && st != 0) return 0xFF;
    int n1 = (a1 < 0) ? 0xF : (a1 > 14 ? 14 : a1);
    int n2 = (a2 < 0) ? 0xF : (a2 > 14 ? 14 : a2);
    /* allele indices above 14 share the code 14; two DIFFERENT alleles must stay different ("15/16" is heterozygous at
     * tdt.c:113,185-187), so such a pair is stored as 13/14 */
    if (n1 == 14 && n2 == 14 && a1 != a2) n1 = 13;
    return (uint8_t)((n1 << 4) | n2);
}

/* The hot shape of a sample string is "a/b" or "a|b" with one-character alleles ('0'..'9' or '.'), closed by NUL or ':':
 * its first FOUR BYTES decide it, and there are only 11 x 11 x 2 x 2 = 484 such words.  They sit in a 1024-entry table
 * under a perfect hash (one multiply, one shift: GT_HASH_MAGIC was searched for these 484 keys; gt_table_build checks it),
 * each entry the word itself and its two codes -- strict (assoc.c:53, tdt.c:103-108: only ALLELES_OK counts, so a missing
 * allele makes the whole call missing) and loose (the stats tool keeps the called allele of "./1").  One probe per string:
 * read the word, hash, compare the entry's word, take the code; no branch on the data.
 *
 * The word is read as the ALIGNED dword at (s & ~3): an aligned dword never crosses a page, so reading past the NUL of a
 * shorter string cannot fault, and what lies there is ignored (a NUL in byte 0..2 is not a hot word).  Heap strings (hpg-libs
 * strdup's one per sample) start on 16-byte boundaries; a string that does not start on a dword boundary simply goes the
 * general way, which reads byte by byte -- as does every row under the CPU suite's instrumented (ASan) build.
 * Measured on the GPU box's host (EPYC 9575F), one thread, strings 32 bytes apart: 1.18 ns per genotype at 10 k samples
 * (the memory system's floor for walking the strings: 0.90); the nibble-table form it replaces 1.97
 * (tools/exp/stage_variants.c, profiles/r04_stage_variants.jsonl). */
#if defined(__SANITIZE_ADDRESS__)
#define GT_WORD_LOADS 0
#elif defined(__has_feature)
#if __has_feature(address_sanitizer)
#define GT_WORD_LOADS 0
#endif
#endif
#ifndef GT_WORD_LOADS
#define GT_WORD_LOADS 1
#endif
#define GT_HASH_MAGIC 0xe9c3deefu
static uint64_t g_gt_table[1024];                       /* bits 0-31 the word, 32-39 strict code, 40-47 loose code */
static int g_gt_table_ok = 0;                           /* 1 built, -1 the magic does not separate the keys (never: then every row goes the general way) */
static pthread_once_t g_gt_table_once = PTHREAD_ONCE_INIT;
static void gt_table_build(void) {
    static const char allele[] = "0123456789.";
    int ok = 1;
    for (int i = 0; i < 1024; i++) g_gt_table[i] = 0xFFFFFFFFull;
    for (int a = 0; a < 11; a++) for (int b = 0; b < 11; b++) for (int sep = 0; sep < 2; sep++) for (int end = 0; end < 2; end++) {
        const uint32_t w = (uint32_t)allele[a] | (uint32_t)(sep ? '|' : '/') << 8 | (uint32_t)allele[b] << 16 | (uint32_t)(end ? ':' : 0) << 24;
        const unsigned n0 = a < 10 ? (unsigned)a : 15u, n2 = b < 10 ? (unsigned)b : 15u, loose = n0 << 4 | n2;
        const unsigned strict = (a == 10 || b == 10) ? 0xFFu : loose;
        const uint32_t i = (w * GT_HASH_MAGIC) >> 22;
        if ((uint32_t)g_gt_table[i] != 0xFFFFFFFFu) ok = 0;
        g_gt_table[i] = (uint64_t)w | (uint64_t)strict << 32 | (uint64_t)loose << 40;
    }
    /* an empty slot holds a word that does not hash to it (and is no hot word: its byte 1 is 0xFF): it matches nothing */
    for (uint32_t i = 0; i < 1024; i++)
        if ((uint32_t)g_gt_table[i] == 0xFFFFFFFFu) { uint32_t k = 0xFFFFFFFFu; while (((k * GT_HASH_MAGIC) >> 22) == i) k -= 0x100u; g_gt_table[i] = k; }
    __atomic_store_n(&g_gt_table_ok, ok ? 1 : -1, __ATOMIC_RELEASE);
}
/* code of the string if it is a hot word (whatever otherwise); *bad becomes non-zero when it is not */
static inline unsigned encode_gt0_word(const char *s, int shift /* 32 strict, 40 loose */, uint32_t *bad) {
    uint32_t w;
    memcpy(&w, (const void *)((uintptr_t)s & ~(uintptr_t)3), 4);
    const uint64_t e = g_gt_table[(w * GT_HASH_MAGIC) >> 22];
    *bad |= ((uint32_t)e ^ w) | ((uint32_t)(uintptr_t)s & 3u);
    return (unsigned)(e >> shift) & 0xFFu;
}

/* a row whose GT is the FIRST field of FORMAT (the usual case).  Eight strings per step, each decided by one probe without
 * a branch, so that the loads of several strings are in flight at once: every string is a heap block of its own and the
 * loop runs near the rate those cache lines arrive.  A row that holds any string of another shape (multi-digit alleles,
 * haploid calls, an empty string) is staged again the general way. */
static void stage_row_gt0(char *const *samples, int num_samples, int strict, uint8_t *row) {
    uint32_t bad = 0;
    int j = 0;
#if GT_WORD_LOADS
    pthread_once(&g_gt_table_once, gt_table_build);
    if (__atomic_load_n(&g_gt_table_ok, __ATOMIC_ACQUIRE) == 1) {
        const int shift = strict ? 32 : 40;
        for (; j + 8 <= num_samples; j += 8) {
            uint64_t v = 0;
            _Pragma("GCC unroll 8")
            for (int k = 0; k < 8; k++) v |= (uint64_t)encode_gt0_word(samples[j + k], shift, &bad) << (8 * k);
            memcpy(row + j, &v, 8);
        }
        for (; j < num_samples; j++) row[j] = (uint8_t)encode_gt0_word(samples[j], shift, &bad);
    } else bad = 1;
#else
    bad = 1;
#endif
    if (bad) for (j = 0; j < num_samples; j++) row[j] = encode_gt_general(samples[j], 0, strict);
}

/* position of "GT" among the ':'-separated keys of a FORMAT given by pointer and length (no copy, any length: the
 * reference strndup's the whole field, assoc.c:45-47) */
static int gt_position_in_format(const char *format, int format_len) {
    int pos = 0;
    const char *p = format, *end = format + (format_len > 0 ? format_len : 0);
    while (p <= end) {
        const char *e = p;
        while (e < end && *e != ':' && *e) e++;
        if (e - p == 2 && p[0] == 'G' && p[1] == 'T') return pos;
        if (e >= end || !*e) return -1;
        p = e + 1; pos++;
    }
    return -1;
}

/* one record's row */
static void stage_one_record(const vcf_record_t *record, int num_samples, int strict, uint8_t *row, uint8_t *is_x) {
    const int gt_position = record->format ? gt_position_in_format(record->format, record->format_len) : -1;
    if (gt_position < 0 || (int)record->samples->size < num_samples) {
        memset(row, 0xFF, (size_t)num_samples);
    } else if (gt_position == 0) {
        stage_row_gt0((char *const *)record->samples->items, num_samples, strict, row);
    } else {
        char *const *samples = (char *const *)record->samples->items;
        for (int j = 0; j < num_samples; j++) row[j] = encode_gt_general(samples[j], gt_position, strict);
    }
    /* assoc.c:94: !strncmp("X", record->chromosome, record->chromosome_len) */
    if (is_x) *is_x = !strncmp("X", record->chromosome, (size_t)record->chromosome_len);
}

/* workers a lone caller's staging is split over (0 = not yet read from HPGV_STAGE_THREADS; 1 = never split) */
static int g_stage_threads = 0;
static int stage_threads(void) {
    int t = __atomic_load_n(&g_stage_threads, __ATOMIC_RELAXED);
    if (t > 0) return t;
    static int env_read = 0;                               /* an adapter may be the process's first call into the library */
    if (!__atomic_exchange_n(&env_read, 1, __ATOMIC_RELAXED) && !g_ctx) host_env_read();
    t = g_env.stage_threads > 0 ? (int)g_env.stage_threads : 8;
    const long cores = sysconf(_SC_NPROCESSORS_ONLN);
    if (cores > 0 && t > cores) t = (int)cores;
    if (t < 1) t = 1;
    __atomic_store_n(&g_stage_threads, t, __ATOMIC_RELAXED);
    return t;
}

void hpgv_host_set_stage_threads(int n) { __atomic_store_n(&g_stage_threads, n > 0 ? n : 0, __ATOMIC_RELAXED); }

/* a lone caller's staging team: created at first use, one job at a time (a second lone caller arriving meanwhile stages
 * its batch itself), released by hpgv_host_shutdown */
static io_pool_t g_stage_team;
static int g_stage_team_threads = 0;
static pthread_mutex_t g_stage_team_mu = PTHREAD_MUTEX_INITIALIZER;
typedef struct { vcf_record_t **variants; int num_variants, num_samples, strict, chunk; uint8_t *out, *is_x; } stage_job_t;
static void stage_task(void *v, int t) {
    const stage_job_t *J = (const stage_job_t *)v;
    const int lo = t * J->chunk, hi = lo + J->chunk < J->num_variants ? lo + J->chunk : J->num_variants;
    for (int i = lo; i < hi; i++)
        stage_one_record(J->variants[i], J->num_samples, J->strict, J->out + (size_t)i * (size_t)J->num_samples, J->is_x ? J->is_x + i : NULL);
}
void stage_team_release(void) {
    pthread_mutex_lock(&g_stage_team_mu);
    if (g_stage_team_threads) { pool_destroy(&g_stage_team); g_stage_team_threads = 0; }
    pthread_mutex_unlock(&g_stage_team_mu);
}

int hpgv_host_stage_records(vcf_record_t **variants, int num_variants, int num_samples, int strict,
                            uint8_t *out, uint8_t *is_x) {
    /* The reference's runner calls the per-batch functions from its own OpenMP workers, one batch each
     * (assoc_runner.c:106-207): there the workers ARE the parallelism and a batch is staged by its caller.  A caller that
     * is alone (not inside an active parallel region) has idle cores: its batch's records are dealt to a small team. */
    const size_t work = (size_t)(num_variants > 0 ? num_variants : 0) * (size_t)(num_samples > 0 ? num_samples : 0);
    int alone = 1;
#ifdef _OPENMP
    alone = !omp_in_parallel();
#endif
    const int team = (alone && work >= ((size_t)1 << 18) && num_variants >= 8) ? stage_threads() : 1;
    if (team > 1 && pthread_mutex_trylock(&g_stage_team_mu) == 0) {
        if (g_stage_team_threads != team) {
            if (g_stage_team_threads) pool_destroy(&g_stage_team);
            pool_init(&g_stage_team, team);
            pool_spread(&g_stage_team);
            g_stage_team_threads = team;
        }
        stage_job_t J = { variants, num_variants, num_samples, strict, 1, out, is_x };
        J.chunk = (num_variants + team * 4 - 1) / (team * 4);
        pool_run(&g_stage_team, stage_task, &J, (num_variants + J.chunk - 1) / J.chunk);
        pthread_mutex_unlock(&g_stage_team_mu);
        return 0;
    }
    for (int i = 0; i < num_variants; i++)
        stage_one_record(variants[i], num_samples, strict, out + (size_t)i * (size_t)num_samples, is_x ? is_x + i : NULL);
    return 0;
}
