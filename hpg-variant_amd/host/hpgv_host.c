/*
 * hpgv_host.c -- the reference's per-batch functions (assoc_test, tdt_test,
 * get_variants_stats) re-hosted on the MI355X engine.  See include/hpgv_host.h.
 *
 * Host work per call is only what the reference does around the arithmetic:
 * locate GT, turn sample strings into bytes, build result records.  Every
 * count and statistic comes from the HIP kernels through include/hpgv.h.
 */
#define _GNU_SOURCE
#include "hpgv_host.h"

#include <fcntl.h>
#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* small containers                                                           */
/* ------------------------------------------------------------------------ */

array_list_t *array_list_new(size_t initial_capacity) {
    array_list_t *l = (array_list_t *)calloc(1, sizeof *l);
    if (!l) return NULL;
    l->capacity = initial_capacity ? initial_capacity : 8;
    l->items = (void **)malloc(l->capacity * sizeof(void *));
    if (!l->items) { free(l); return NULL; }
    return l;
}

int array_list_insert(void *item, array_list_t *list) {
    if (list->size == list->capacity) {
        size_t cap = list->capacity * 2;
        void **p = (void **)realloc(list->items, cap * sizeof(void *));
        if (!p) return 0;
        list->items = p; list->capacity = cap;
    }
    list->items[list->size++] = item;
    return 1;
}

void *array_list_get(size_t index, const array_list_t *list) {
    return index < list->size ? list->items[index] : NULL;
}

void array_list_free(array_list_t *list, void (*item_free)(void *)) {
    if (!list) return;
    if (item_free) for (size_t i = 0; i < list->size; i++) item_free(list->items[i]);
    free(list->items);
    free(list);
}

vcf_record_t *vcf_record_new(void) {
    vcf_record_t *r = (vcf_record_t *)calloc(1, sizeof *r);
    if (r) r->samples = array_list_new(16);
    return r;
}
void vcf_record_free(vcf_record_t *r) { if (r) { array_list_free(r->samples, NULL); free(r); } }
void set_vcf_record_chromosome(char *s, int n, vcf_record_t *r) { r->chromosome = s; r->chromosome_len = n; }
void set_vcf_record_position(long p, vcf_record_t *r) { r->position = (unsigned long)p; }
void set_vcf_record_id(char *s, int n, vcf_record_t *r) { r->id = s; r->id_len = n; }
void set_vcf_record_reference(char *s, int n, vcf_record_t *r) { r->reference = s; r->reference_len = n; }
void set_vcf_record_alternate(char *s, int n, vcf_record_t *r) { r->alternate = s; r->alternate_len = n; }
void set_vcf_record_format(char *s, int n, vcf_record_t *r) { r->format = s; r->format_len = n; }

individual_t *individual_new(char *id, float variable, enum Sex sex, enum Condition condition,
                             individual_t *father, individual_t *mother, struct family *family) {
    individual_t *i = (individual_t *)calloc(1, sizeof *i);
    if (!i) return NULL;
    i->id = id; i->variable = variable; i->sex = sex; i->condition = condition;
    i->father = father; i->mother = mother; i->family = family;
    return i;
}
void individual_free(individual_t *i) { free(i); }

family_t *family_new(char *id) {
    family_t *f = (family_t *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->id = id;
    f->founders = array_list_new(2);
    f->members = array_list_new(4);
    return f;
}
int family_set_parent(individual_t *p, family_t *f) { return array_list_insert(p, f->founders) ? 0 : 1; }
int family_add_child(individual_t *c, family_t *f) { return array_list_insert(c, f->members) ? 0 : 1; }
void family_free(family_t *f) {
    if (!f) return;
    array_list_free(f->founders, NULL);
    array_list_free(f->members, NULL);
    free(f);
}

static size_t str_hash(const char *s) {
    size_t h = 1469598103934665603ULL;
    for (; *s; s++) { h ^= (unsigned char)*s; h *= 1099511628211ULL; }
    return h;
}

sample_ids_t *sample_ids_new(size_t expected) {
    sample_ids_t *t = (sample_ids_t *)calloc(1, sizeof *t);
    if (!t) return NULL;
    size_t nb = 16;
    while (nb < expected * 2 + 1) nb <<= 1;
    t->n_buckets = nb;
    t->keys = (const char **)calloc(nb, sizeof(char *));
    t->vals = (int *)calloc(nb, sizeof(int));
    if (!t->keys || !t->vals) { free(t->keys); free(t->vals); free(t); return NULL; }
    return t;
}

static int sample_ids_grow(sample_ids_t *t) {
    sample_ids_t *n = sample_ids_new(t->n_buckets);
    if (!n) return 0;
    for (size_t i = 0; i < t->n_buckets; i++)
        if (t->keys[i]) sample_ids_put(n, t->keys[i], t->vals[i]);
    free(t->keys); free(t->vals);
    *t = *n;
    free(n);
    return 1;
}

int sample_ids_put(sample_ids_t *t, const char *name, int position) {
    if ((t->size + 1) * 2 > t->n_buckets && !sample_ids_grow(t)) return 0;
    size_t m = t->n_buckets - 1, i = str_hash(name) & m;
    while (t->keys[i] && strcmp(t->keys[i], name)) i = (i + 1) & m;
    if (!t->keys[i]) { t->keys[i] = name; t->size++; }
    t->vals[i] = position;
    return 1;
}

int sample_ids_get(const sample_ids_t *t, const char *name) {
    size_t m = t->n_buckets - 1, i = str_hash(name) & m;
    while (t->keys[i]) {
        if (!strcmp(t->keys[i], name)) return t->vals[i];
        i = (i + 1) & m;
    }
    return -1;
}

void sample_ids_free(sample_ids_t *t) { if (t) { free(t->keys); free(t->vals); free(t); } }

void list_init(const char *name, int writers, size_t max_length, list_t *list) {
    memset(list, 0, sizeof *list);
    list->name = name ? strdup(name) : NULL;
    list->writers = writers;
    list->max_length = max_length;
    pthread_mutex_init(&list->lock, NULL);
    pthread_cond_init(&list->condition, NULL);
}

list_item_t *list_item_new(int id, int type, void *data_p) {
    list_item_t *it = (list_item_t *)calloc(1, sizeof *it);
    if (it) { it->id = id; it->type = type; it->data_p = data_p; }
    return it;
}
void list_item_free(list_item_t *item) { free(item); }

int list_insert_item(list_item_t *item, list_t *list) {
    if (!item || !list) return 0;
    pthread_mutex_lock(&list->lock);
    item->next_p = NULL;
    if (list->last_p) list->last_p->next_p = item; else list->first_p = item;
    list->last_p = item;
    list->length++;
    pthread_cond_broadcast(&list->condition);
    pthread_mutex_unlock(&list->lock);
    return 1;
}

/* n items chained through next_p, appended in one go: what n list_insert_item calls from one thread leave behind, with one
 * lock and one wake-up instead of n (a consumer asleep on the condition costs a system call per broadcast) */
static void list_insert_chain(list_item_t *first, list_item_t *last, size_t n, list_t *list) {
    if (!first || !list) return;
    pthread_mutex_lock(&list->lock);
    last->next_p = NULL;
    if (list->last_p) list->last_p->next_p = first; else list->first_p = first;
    list->last_p = last;
    list->length += n;
    pthread_cond_broadcast(&list->condition);
    pthread_mutex_unlock(&list->lock);
}
typedef struct { list_item_t *first, *last; size_t n; } item_chain_t;
static inline void chain_add(item_chain_t *c, list_item_t *it) {
    if (!it) return;
    it->next_p = NULL;
    if (c->last) c->last->next_p = it; else c->first = it;
    c->last = it; c->n++;
}

list_item_t *list_remove_item(list_t *list) {
    pthread_mutex_lock(&list->lock);
    while (!list->first_p && list->writers > 0) pthread_cond_wait(&list->condition, &list->lock);
    list_item_t *it = list->first_p;
    if (it) {
        list->first_p = it->next_p;
        if (!list->first_p) list->last_p = NULL;
        list->length--;
        it->next_p = NULL;
    }
    pthread_mutex_unlock(&list->lock);
    return it;
}

int list_decr_writers(list_t *list) {
    pthread_mutex_lock(&list->lock);
    if (list->writers > 0) list->writers--;
    pthread_cond_broadcast(&list->condition);
    int w = list->writers;
    pthread_mutex_unlock(&list->lock);
    return w;
}

void list_free_deep(list_t *list, void (*data_free)(void *)) {
    list_item_t *it = list->first_p;
    while (it) {
        list_item_t *n = it->next_p;
        if (data_free && it->data_p) data_free(it->data_p);
        free(it);
        it = n;
    }
    free(list->name);
    pthread_mutex_destroy(&list->lock);
    pthread_cond_destroy(&list->condition);
    memset(list, 0, sizeof *list);
}

void assoc_basic_result_free(assoc_basic_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void assoc_fisher_result_free(assoc_fisher_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void tdt_result_free(tdt_result_t *r) {
    if (!r) return;
    free(r->chromosome); free(r->id); free(r->reference); free(r->alternate); free(r);
}
void variant_stats_free(variant_stats_t *s) {
    if (!s) return;
    free(s->chromosome); free(s->ref_allele); free(s->alt_alleles);
    free(s->alleles_count); free(s->genotypes_count); free(s->alleles_freq); free(s->genotypes_freq);
    free(s->phenotype_stats);
    free(s);
}
sample_stats_t *sample_stats_new(char *name) {
    sample_stats_t *s = (sample_stats_t *)calloc(1, sizeof *s);
    if (s) s->name = name;
    return s;
}
void sample_stats_free(sample_stats_t *s) { free(s); }
file_stats_t *file_stats_new(void) {
    file_stats_t *f = (file_stats_t *)calloc(1, sizeof *f);
    if (f) pthread_mutex_init(&f->lock, NULL);
    return f;
}
void file_stats_free(file_stats_t *f) { if (f) { pthread_mutex_destroy(&f->lock); free(f); } }

/* ------------------------------------------------------------------------ */
/* a small team of threads that sleep between jobs (condition variables, no spinning: a caller inside a CPU-limited      */
/* container must not burn its quota waiting): the runners' reader / formatter teams and a lone caller's staging        */
/* ------------------------------------------------------------------------ */

typedef void (*pool_fn)(void *arg, int task);
typedef struct {
    pthread_t *th; int n_threads;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_done;
    pool_fn fn; void *arg; int n_tasks, next, running, gen, stop;
} io_pool_t;

static void pool_drain(io_pool_t *p) {                  /* called with mu held; returns with mu held */
    while (p->next < p->n_tasks) {
        const int t = p->next++;
        pthread_mutex_unlock(&p->mu);
        p->fn(p->arg, t);
        pthread_mutex_lock(&p->mu);
    }
}

static void *pool_worker(void *v) {
    io_pool_t *p = (io_pool_t *)v;
    int seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->stop && p->gen == seen) pthread_cond_wait(&p->cv_work, &p->mu);
        if (p->stop) break;
        seen = p->gen;
        p->running++;
        pool_drain(p);
        if (--p->running == 0) pthread_cond_broadcast(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

/* n_threads counts the caller, which works too: n_threads - 1 threads are created */
static void pool_init(io_pool_t *p, int n_threads) {
    memset(p, 0, sizeof *p);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv_work, NULL);
    pthread_cond_init(&p->cv_done, NULL);
    if (n_threads > 1) p->th = (pthread_t *)calloc((size_t)n_threads - 1, sizeof(pthread_t));
    for (int i = 0; p->th && i < n_threads - 1; i++) {
        if (pthread_create(&p->th[p->n_threads], NULL, pool_worker, p) == 0) p->n_threads++;
    }
}

/* the pool's threads spread over the CPUs the process may use, one each (the caller's own CPU last): a thread woken from a
 * condition variable starts on its waker's CPU, and some schedulers (small VMs) leave it queued there for a whole slice
 * instead of moving it to an idle core -- a 2 ms job cannot wait for that */
static void pool_spread(io_pool_t *p) {
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    const int self = sched_getcpu();
    int cpus[CPU_SETSIZE], n = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed) && c != self) cpus[n++] = c;
    if (self >= 0 && CPU_ISSET(self, &allowed)) cpus[n++] = self;
    if (n < 2) return;
    for (int i = 0; i < p->n_threads; i++) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[i % n], &one);
        (void)pthread_setaffinity_np(p->th[i], sizeof one, &one);
    }
}

static void pool_destroy(io_pool_t *p) {
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_threads; i++) pthread_join(p->th[i], NULL);
    free(p->th);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_work); pthread_cond_destroy(&p->cv_done);
    memset(p, 0, sizeof *p);
}

/* runs fn(arg, 0 .. n_tasks-1), each task once, on the pool's threads and the caller; one job at a time per
 * pool.  p == NULL, or a pool without threads, runs the tasks inline. */
static void pool_run(io_pool_t *p, pool_fn fn, void *arg, int n_tasks) {
    if (!p || p->n_threads == 0 || n_tasks <= 1) { for (int t = 0; t < n_tasks; t++) fn(arg, t); return; }
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->arg = arg; p->n_tasks = n_tasks; p->next = 0; p->gen++;
    pthread_cond_broadcast(&p->cv_work);
    p->running++;
    pool_drain(p);
    p->running--;
    while (p->running > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    p->n_tasks = 0;
    pthread_mutex_unlock(&p->mu);
}

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
    t = 8;
    const char *e = getenv("HPGV_STAGE_THREADS");
    if (e && *e) t = atoi(e);
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
static void stage_team_release(void) {
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

/* ------------------------------------------------------------------------ */
/* engine binding                                                             */
/* ------------------------------------------------------------------------ */

static hpgv_ctx *g_ctx = NULL;
/* The bgzip device path may run in PARTS, one per member of a group context (bgzf_parts_stage): the threads that stage,
 * decode and hand out a part work on that member's device.  They say so once -- ctx_use(member context, member index) --
 * and everything below them that allocates, copies or launches uses CTX, and the per-member caches (streams, the decoded
 * text) slot t_member.  Threads that never say so work on g_ctx, which for a group is its first member: the single-device
 * path is unchanged. */
enum { MEMBERS_MAX = 16 };
static __thread hpgv_ctx *t_ctx = NULL;
static __thread int t_member = 0;
#define CTX (t_ctx ? t_ctx : g_ctx)
typedef struct { hpgv_ctx *ctx; int member; } ctx_saved_t;
static ctx_saved_t ctx_use(hpgv_ctx *ctx, int member) { ctx_saved_t o = { t_ctx, t_member }; t_ctx = ctx; t_member = ctx ? member : 0; return o; }
static void ctx_back(ctx_saved_t o) { t_ctx = o.ctx; t_member = o.member; }
static int g_device = 0;
static pthread_mutex_t g_init_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_rwlock_t g_cohort_lock = PTHREAD_RWLOCK_INITIALIZER;
static char g_err[512];

/* what the current device-side cohort descriptions were built from */
static struct {
    const void *samples; int num_samples; uint64_t cond_hash; int set;
    uint8_t *cond;                                       /* the installed condition vector itself: a hash hit is confirmed against it */
} g_assoc_key;
static struct {
    int num_families; int num_columns; uint64_t hash; int set;
    int32_t *csr; uint8_t *csex; size_t n_children;       /* the installed description itself (father | mother | offsets | child columns): a hash hit is confirmed against it */
} g_tdt_key;
static struct { int num_samples; int set; } g_stats_key;
static struct { int num_samples; int n_trios; uint64_t hash; int set; } g_ped_key;
static struct { int num_samples; int n_groups; uint64_t hash; int set; } g_group_key;
static struct { const void *table; int n; } g_lf_key;

const char *hpgv_host_last_error(void) { return g_err; }

static int host_fail(const char *what, int rc) {
    snprintf(g_err, sizeof g_err, "%s: hpgv status %d: %s", what, rc,
             g_ctx ? hpgv_last_error(g_ctx) : hpgv_last_error(NULL));
    return rc;
}

/* the engine's devices: hpgv_host_init(d) binds device d; hpgv_host_init_devices a list (a group context: batches are
 * dealt to the devices, hpgv.h hpgv_create_multi); with neither, the first call of an adapter or runner reads the
 * environment variable HPGV_DEVICES -- "0,1,2,3" or "all" -- and falls back to device 0 */
int hpgv_host_init_devices(const int *device_ids, int n_devices) {
    pthread_mutex_lock(&g_init_mu);
    int rc = HPGV_OK;
    if (!g_ctx) {
        if (!device_ids || n_devices < 1) { snprintf(g_err, sizeof g_err, "hpgv_host_init_devices: no device ids"); rc = HPGV_ERR_INVALID; }
        else {
            g_device = device_ids[0];
            rc = n_devices == 1 ? hpgv_create(device_ids[0], &g_ctx) : hpgv_create_multi(device_ids, n_devices, &g_ctx);
            if (rc != HPGV_OK) host_fail(n_devices == 1 ? "hpgv_create" : "hpgv_create_multi", rc);
        }
    }
    pthread_mutex_unlock(&g_init_mu);
    return rc;
}

int hpgv_host_init(int device_id) { return hpgv_host_init_devices(&device_id, 1); }

int hpgv_host_device_count(void) { return g_ctx ? hpgv_group_size(g_ctx) : 0; }

static int init_from_environment(void) {
    const char *e = getenv("HPGV_DEVICES");
    int ids[64], n = 0;
    if (e && *e) {
        if (!strcmp(e, "all")) {
            const int have = hpgv_device_count();
            for (int i = 0; i < have && n < 64; i++) ids[n++] = i;
        } else {
            const char *p = e;
            while (*p && n < 64) {
                char *end;
                const long v = strtol(p, &end, 10);
                if (end == p || v < 0) { snprintf(g_err, sizeof g_err, "HPGV_DEVICES='%s' is not a list of device ids", e); return HPGV_ERR_INVALID; }
                ids[n++] = (int)v;
                p = (*end == ',') ? end + 1 : end;
                if (*end && *end != ',') { snprintf(g_err, sizeof g_err, "HPGV_DEVICES='%s' is not a list of device ids", e); return HPGV_ERR_INVALID; }
            }
        }
    }
    if (n == 0) { ids[0] = g_device; n = 1; }
    return hpgv_host_init_devices(ids, n);
}

/* page-locked text buffers of the file runners, kept between runs: page-locking 5 x 64 MB costs 56 ms and releasing it
 * another 45 ms -- a third of a run over an 8 GB file.  Released by hpgv_host_shutdown. */
enum { TEXT_CACHE_N = 24, TEXT_LIVE_N = 160 };      /* live: every batch buffer of a run (RUN_NB_MAX) + the cache + the uploaders' rings */
static struct { char *p; size_t cap; } g_text_cache[TEXT_CACHE_N];
static struct { char *p; size_t cap; } g_text_live[TEXT_LIVE_N];      /* buffers that are out, with what they really hold */
static pthread_mutex_t g_text_mu = PTHREAD_MUTEX_INITIALIZER;
static char *text_buf_get(size_t cap) {
    char *p = NULL;
    size_t real = cap;
    pthread_mutex_lock(&g_text_mu);
    int best = -1;                                        /* the smallest one that is large enough: the 256 MB windows of a bgzip run stay for the next such run */
    for (int i = 0; i < TEXT_CACHE_N; i++)
        if (g_text_cache[i].p && g_text_cache[i].cap >= cap && (best < 0 || g_text_cache[i].cap < g_text_cache[best].cap)) best = i;
    if (best >= 0) { p = g_text_cache[best].p; real = g_text_cache[best].cap; g_text_cache[best].p = NULL; g_text_cache[best].cap = 0; }
    pthread_mutex_unlock(&g_text_mu);
    if (!p && hpgv_host_alloc(g_ctx, cap, (void **)&p) != HPGV_OK) p = NULL;      /* pinned: full-rate H2D */
    if (p) {                                              /* a larger buffer that served a smaller request goes back as what it is */
        pthread_mutex_lock(&g_text_mu);
        for (int i = 0; i < TEXT_LIVE_N; i++)
            if (!g_text_live[i].p) { g_text_live[i].p = p; g_text_live[i].cap = real; break; }
        pthread_mutex_unlock(&g_text_mu);
    }
    return p;
}
static void text_buf_put(char *p, size_t cap) {
    if (!p) return;
    pthread_mutex_lock(&g_text_mu);
    for (int i = 0; i < TEXT_LIVE_N; i++)
        if (g_text_live[i].p == p) { if (g_text_live[i].cap > cap) cap = g_text_live[i].cap; g_text_live[i].p = NULL; break; }
    int kept = 0;
    for (int i = 0; i < TEXT_CACHE_N && !kept; i++)
        if (!g_text_cache[i].p) { g_text_cache[i].p = p; g_text_cache[i].cap = cap; kept = 1; }
    pthread_mutex_unlock(&g_text_mu);
    if (!kept) (void)hpgv_host_free(g_ctx, p);
}
/* page-locked staging buffers of the per-batch adapters (assoc_test, tdt_test, get_variants_stats, get_sample_stats):
 * the genotype bytes are staged straight into page-locked memory, which the engine's per-batch kernel reads in place
 * over the bus (no copy on the way).  One buffer per concurrent worker, kept between calls, released by
 * hpgv_host_shutdown; when the pool is full or page-locking fails an ordinary malloc serves (the engine then copies). */
enum { STAGE_POOL_N = 64 };
static struct { uint8_t *p; size_t cap; int busy; } g_stage_pool[STAGE_POOL_N];
static pthread_mutex_t g_stage_mu = PTHREAD_MUTEX_INITIALIZER;
static uint8_t *stage_get(size_t bytes, int *slot) {
    *slot = -1;
    if (bytes == 0) bytes = 16;
    pthread_mutex_lock(&g_stage_mu);
    int k = -1, empty = -1;
    for (int i = 0; i < STAGE_POOL_N; i++) {
        if (g_stage_pool[i].busy) continue;
        if (g_stage_pool[i].p && g_stage_pool[i].cap >= bytes && (k < 0 || g_stage_pool[i].cap < g_stage_pool[k].cap)) k = i;
        if (!g_stage_pool[i].p && empty < 0) empty = i;
    }
    if (k < 0 && empty < 0)                             /* every idle buffer is too small: let the smallest one go */
        for (int i = 0; i < STAGE_POOL_N; i++) if (!g_stage_pool[i].busy && (empty < 0 || g_stage_pool[i].cap < g_stage_pool[empty].cap)) empty = i;
    if (k < 0 && empty >= 0) { k = empty; g_stage_pool[k].busy = 1; } else if (k >= 0) g_stage_pool[k].busy = 1;
    pthread_mutex_unlock(&g_stage_mu);
    if (k < 0) return (uint8_t *)malloc(bytes);
    if (g_stage_pool[k].cap < bytes) {                  /* slot k is ours (busy): grow it outside the lock */
        if (g_stage_pool[k].p) (void)hpgv_host_free(g_ctx, g_stage_pool[k].p);
        g_stage_pool[k].p = NULL; g_stage_pool[k].cap = 0;
        void *q = NULL;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hpgv_host_alloc(g_ctx, want, &q) == HPGV_OK) { g_stage_pool[k].p = (uint8_t *)q; g_stage_pool[k].cap = want; }
        else {
            pthread_mutex_lock(&g_stage_mu); g_stage_pool[k].busy = 0; pthread_mutex_unlock(&g_stage_mu);
            return (uint8_t *)malloc(bytes);
        }
    }
    *slot = k;
    return g_stage_pool[k].p;
}
static void stage_put(uint8_t *p, int slot) {
    if (slot < 0) { free(p); return; }
    pthread_mutex_lock(&g_stage_mu);
    g_stage_pool[slot].busy = 0;
    pthread_mutex_unlock(&g_stage_mu);
}
static void stage_pool_release(void) {                  /* g_ctx still alive */
    pthread_mutex_lock(&g_stage_mu);
    for (int i = 0; i < STAGE_POOL_N; i++)
        if (g_stage_pool[i].p && !g_stage_pool[i].busy) { (void)hpgv_host_free(g_ctx, g_stage_pool[i].p); g_stage_pool[i].p = NULL; g_stage_pool[i].cap = 0; }
    pthread_mutex_unlock(&g_stage_mu);
}

/* device memory for a decoded file, kept between runs like the page-locked buffers (an 8 GB allocation and its release
 * cost 0.1 s); released by hpgv_host_shutdown.  Where the device allows it the buffer is a reserved address range that is
 * backed as far as needed (hpgv_dev_reserve): a bgzip file's text size is known only when its last block has been seen. */
enum { DEV_TEXT_FIXED = 0, DEV_TEXT_GROWS = 1 };
#define DEV_TEXT_RESERVE ((size_t)56 << 30)
static void *g_dev_text_m[MEMBERS_MAX]; static size_t g_dev_text_cap_m[MEMBERS_MAX]; static int g_dev_text_kind_m[MEMBERS_MAX];
#define g_dev_text g_dev_text_m[t_member]
#define g_dev_text_cap g_dev_text_cap_m[t_member]
#define g_dev_text_kind g_dev_text_kind_m[t_member]
static void dev_text_free(void *p, int kind) {
    if (!p) return;
    if (kind == DEV_TEXT_GROWS) (void)hpgv_dev_release(CTX, p); else (void)hpgv_dev_free(CTX, p);
}
/* a buffer with `bytes` usable bytes (*cap: how many it has); *kind = DEV_TEXT_GROWS when dev_text_grow can extend it */
static void *dev_text_get(size_t bytes, size_t *cap, int *kind) {
    void *p = NULL;
    pthread_mutex_lock(&g_text_mu);
    if (g_dev_text && (g_dev_text_kind == DEV_TEXT_GROWS || g_dev_text_cap >= bytes)) {
        p = g_dev_text; *cap = g_dev_text_cap; *kind = g_dev_text_kind;
        g_dev_text = NULL; g_dev_text_cap = 0;
    }
    pthread_mutex_unlock(&g_text_mu);
    if (p) {
        if (*cap >= bytes) return p;
        if (bytes <= DEV_TEXT_RESERVE && hpgv_dev_commit(CTX, p, bytes) == HPGV_OK) { *cap = bytes; return p; }
        dev_text_free(p, *kind);
        p = NULL;
    }
    if (bytes <= DEV_TEXT_RESERVE && !getenv("HPGV_NO_GROWING_TEXT") && hpgv_dev_reserve(CTX, DEV_TEXT_RESERVE, &p) == HPGV_OK) {
        if (hpgv_dev_commit(CTX, p, bytes) == HPGV_OK) { *cap = bytes; *kind = DEV_TEXT_GROWS; return p; }
        (void)hpgv_dev_release(CTX, p);
        p = NULL;
    }
    if (hpgv_dev_alloc(CTX, bytes, &p) != HPGV_OK) return NULL;
    *cap = bytes; *kind = DEV_TEXT_FIXED;
    return p;
}
static int dev_text_grow(void *p, size_t bytes, size_t *cap) {
    if (bytes <= *cap) return 1;
    if (bytes > DEV_TEXT_RESERVE) return 0;
    if (hpgv_dev_commit(CTX, p, bytes) != HPGV_OK) {
        size_t have = 0;                                 /* a failed growth may still have mapped some pieces: the cached size follows */
        if (hpgv_dev_committed(CTX, p, &have) == HPGV_OK && have > *cap) *cap = have < bytes ? have : bytes;
        return *cap >= bytes;
    }
    *cap = bytes;
    return 1;
}
static void dev_text_put(void *p, size_t bytes, int kind) {
    if (!p) return;
    void *old = NULL; int old_kind = 0;
    pthread_mutex_lock(&g_text_mu);
    if (!g_dev_text || g_dev_text_cap < bytes || (kind == DEV_TEXT_GROWS && g_dev_text_kind != DEV_TEXT_GROWS)) {
        old = g_dev_text; old_kind = g_dev_text_kind;
        g_dev_text = p; g_dev_text_cap = bytes; g_dev_text_kind = kind;
        p = NULL;
    }
    pthread_mutex_unlock(&g_text_mu);
    dev_text_free(old, old_kind);
    dev_text_free(p, kind);
}

/* streams of the bgzip device path, kept between runs: creating the seven a run uses took 18 ms of a 0.14 s run (they are
 * idle when they come back) */
enum { STREAM_CACHE_N = 16 };
static void *g_stream_cache_m[MEMBERS_MAX][2][STREAM_CACHE_N];        /* per member; [0] normal priority, [1] lowest */
#define g_stream_cache g_stream_cache_m[t_member]
static int stream_get(int low, void **out) {
    pthread_mutex_lock(&g_text_mu);
    for (int i = 0; i < STREAM_CACHE_N; i++)
        if (g_stream_cache[low][i]) { *out = g_stream_cache[low][i]; g_stream_cache[low][i] = NULL; pthread_mutex_unlock(&g_text_mu); return HPGV_OK; }
    pthread_mutex_unlock(&g_text_mu);
    return low ? hpgv_stream_create_low(CTX, out) : hpgv_stream_create(CTX, out);
}
static void stream_put(int low, void *st) {
    if (!st) return;
    (void)hpgv_stream_sync(CTX, st);
    pthread_mutex_lock(&g_text_mu);
    for (int i = 0; i < STREAM_CACHE_N; i++)
        if (!g_stream_cache[low][i]) { g_stream_cache[low][i] = st; st = NULL; break; }
    pthread_mutex_unlock(&g_text_mu);
    if (st) (void)hpgv_stream_destroy(CTX, st);
}

static void text_cache_release(void) {                  /* g_ctx still alive */
    const int G = hpgv_group_size(g_ctx);
    for (int m = 0; m < G && m < MEMBERS_MAX; m++) {    /* every member's streams and decoded-text buffer */
        const ctx_saved_t o = ctx_use(G > 1 ? hpgv_group_member(g_ctx, m) : NULL, m);
        for (int low = 0; low < 2; low++)
            for (int i = 0; i < STREAM_CACHE_N; i++) {
                pthread_mutex_lock(&g_text_mu);
                void *st = g_stream_cache[low][i]; g_stream_cache[low][i] = NULL;
                pthread_mutex_unlock(&g_text_mu);
                if (st) (void)hpgv_stream_destroy(CTX, st);
            }
        pthread_mutex_lock(&g_text_mu);
        void *dt = g_dev_text; const int dk = g_dev_text_kind;
        g_dev_text = NULL; g_dev_text_cap = 0;
        pthread_mutex_unlock(&g_text_mu);
        dev_text_free(dt, dk);
        ctx_back(o);
    }
    pthread_mutex_lock(&g_text_mu);
    for (int i = 0; i < TEXT_CACHE_N; i++)
        if (g_text_cache[i].p) { (void)hpgv_host_free(g_ctx, g_text_cache[i].p); g_text_cache[i].p = NULL; g_text_cache[i].cap = 0; }
    pthread_mutex_unlock(&g_text_mu);
}

void hpgv_host_shutdown(void) {
    pthread_mutex_lock(&g_init_mu);
    /* per-batch calls run under the cohort lock's read side: the write side waits for the ones in flight before the engine
     * goes.  (A worker between stage_get and its call keeps its page-locked buffer: that memory is the HIP runtime's, not the
     * context's, and the slot serves the next engine.) */
    pthread_rwlock_wrlock(&g_cohort_lock);
    if (g_ctx) { text_cache_release(); stage_pool_release(); hpgv_destroy(g_ctx); g_ctx = NULL; }
    stage_team_release();
    pthread_rwlock_unlock(&g_cohort_lock);
    free(g_assoc_key.cond);
    memset(&g_assoc_key, 0, sizeof g_assoc_key);
    free(g_tdt_key.csr); free(g_tdt_key.csex);
    memset(&g_tdt_key, 0, sizeof g_tdt_key);
    memset(&g_stats_key, 0, sizeof g_stats_key);
    memset(&g_ped_key, 0, sizeof g_ped_key);
    memset(&g_group_key, 0, sizeof g_group_key);
    memset(&g_lf_key, 0, sizeof g_lf_key);
    pthread_mutex_unlock(&g_init_mu);
}

static int ensure_engine(void) { return g_ctx ? HPGV_OK : init_from_environment(); }

static char *dupn(const char *s, int n) { return strndup(s ? s : "", (size_t)(n > 0 ? n : 0)); }

static int thread_id(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

/* where a per-batch adapter call spends its time (hpgv_host.h "hpgv_host_adapter_times"): nanoseconds, added atomically */
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
typedef struct { int on; uint64_t t[4]; } adapter_clock_t;
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

/* hpg-libs init_logarithm_array: table[i] = ln(i!) */
double *init_logarithm_array(int n) {
    if (n <= 0) return NULL;
    double *t = (double *)malloc((size_t)n * sizeof(double));
    if (!t) return NULL;
    t[0] = 0.0;
    for (int i = 1; i < n; i++) t[i] = t[i - 1] + log((double)i);
    return t;
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
/* epistasis (MDR): k-fold masks, the run over a vcf2epi dataset, the report  */
/* ------------------------------------------------------------------------ */

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

/* cross_validation.c:16-100: cases and controls are shuffled separately and dealt to the folds one case and
 * one control per fold and round, fold 0 first; sizes[3 * f + {0, 1, 2}] = samples, cases, controls of fold f;
 * every fold lists its samples in increasing order (cases have the indices below num_samples_affected).
 * The shuffle is hpg-libs' array_shuffle_int there; here a Fisher-Yates pass on rand(). */
int **get_k_folds(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k, unsigned int **sizes) {
    const unsigned int n = num_samples_affected + num_samples_unaffected;
    if (k == 0) return NULL;
    int *samples = (int *)malloc(sizeof(int) * (n + 1));
    int **folds = (int **)calloc(k, sizeof(int *));
    unsigned int *fold_sizes = (unsigned int *)calloc(3 * (size_t)k, sizeof(unsigned int));
    if (!samples || !folds || !fold_sizes) { free(samples); free(folds); free(fold_sizes); return NULL; }
    for (unsigned int i = 0; i < n; i++) samples[i] = (int)i;
    for (unsigned int part = 0; part < 2; part++) {
        int *a = samples + (part ? num_samples_affected : 0);
        const unsigned int m = part ? num_samples_unaffected : num_samples_affected;
        for (unsigned int i = m; i > 1; i--) { unsigned int j = (unsigned int)rand() % i; int t = a[i - 1]; a[i - 1] = a[j]; a[j] = t; }
    }
    for (unsigned int f = 0; f < k; f++) {                            /* round-robin: fold f gets every k-th case and control */
        fold_sizes[3 * f + 1] = num_samples_affected / k + (f < num_samples_affected % k);
        fold_sizes[3 * f + 2] = num_samples_unaffected / k + (f < num_samples_unaffected % k);
        fold_sizes[3 * f] = fold_sizes[3 * f + 1] + fold_sizes[3 * f + 2];
        folds[f] = (int *)malloc(sizeof(int) * (fold_sizes[3 * f] + 1));
        unsigned int o = 0;
        for (unsigned int i = f; i < num_samples_affected; i += k) folds[f][o++] = samples[i];
        for (unsigned int i = f; i < num_samples_unaffected; i += k) folds[f][o++] = samples[num_samples_affected + i];
        qsort(folds[f], fold_sizes[3 * f], sizeof(int), cmp_int);
    }
    free(samples);
    *sizes = fold_sizes;
    return folds;
}

/* cross_validation.c:247-281: k x num_samples_with_padding bytes, 1 = the sample is in the TRAINING part of the
 * fold, 0 = in its testing part or padding; cases and controls each padded to a multiple of 16 */
uint8_t *get_k_folds_masks(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k,
                           int **folds, unsigned int *sizes) {
    const size_t pad_a = ((size_t)num_samples_affected + 15) / 16 * 16, pad_u = ((size_t)num_samples_unaffected + 15) / 16 * 16;
    const size_t padded = pad_a + pad_u;
    uint8_t *masks = (uint8_t *)malloc(padded * k + 16);
    if (!masks) return NULL;
    memset(masks, 1, padded * k);
    for (unsigned int f = 0; f < k; f++) {
        uint8_t *m = masks + (size_t)f * padded;
        for (unsigned int j = 0; j < sizes[3 * f + 1]; j++) m[folds[f][j]] = 0;
        for (unsigned int j = sizes[3 * f + 1]; j < sizes[3 * f]; j++) m[(size_t)folds[f][j] + (pad_a - num_samples_affected)] = 0;
        memset(m + num_samples_affected, 0, pad_a - num_samples_affected);
        memset(m + pad_a + num_samples_unaffected, 0, pad_u - num_samples_unaffected);
    }
    return masks;
}

enum { EPI_ORDER_MAX = 5, EPI_MASK_WORDS = 8 };            /* 3^5 = 243 cells in 8 x 32 bits (hpgv.h "ANY order") */
typedef struct { int c[EPI_ORDER_MAX]; int count; uint32_t risky[EPI_MASK_WORDS]; double accuracy; } epi_model_t;     /* unused SNP slots: -1 */

static int cmp_model_comb(const void *a, const void *b) {            /* compare_risky, epistasis.c:162-175 */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    for (int s = 0; s < EPI_ORDER_MAX; s++) if (x->c[s] != y->c[s]) return x->c[s] < y->c[s] ? -1 : 1;
    return 0;
}
static int cmp_model_cva(const void *a, const void *b) {             /* CV-a: accuracy, then the combination */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    if (x->accuracy != y->accuracy) return x->accuracy > y->accuracy ? -1 : 1;
    return cmp_model_comb(a, b);
}
static int cmp_model_cvc(const void *a, const void *b) {             /* CV-c: folds that ranked the model, then accuracy */
    const epi_model_t *x = (const epi_model_t *)a, *y = (const epi_model_t *)b;
    if (x->count != y->count) return x->count > y->count ? -1 : 1;
    return cmp_model_cva(a, b);
}

int hpgv_run_epistasis(const char *dataset_path, int num_folds, int num_cv_repetitions, int max_ranking_size,
                       int eval_subset, int eval_mode, const char *out_prefix) {
    return hpgv_run_epistasis_order(dataset_path, 2, num_folds, num_cv_repetitions, max_ranking_size, eval_subset, eval_mode, out_prefix);
}

int hpgv_run_epistasis_order(const char *dataset_path, int order, int num_folds, int num_cv_repetitions, int max_ranking_size,
                             int eval_subset, int eval_mode, const char *out_prefix) {
    int rc = ensure_engine();
    if (rc) return rc;
    /* the reference's --order is any integer (main_epistasis.c:128,142); the engine takes 2 to 5 (243 cells) */
    if (order < 2 || order > EPI_ORDER_MAX) { snprintf(g_err, sizeof g_err, "combinations of %d SNPs are not supported (2 to %d)", order, EPI_ORDER_MAX); return HPGV_ERR_UNSUPPORTED; }
    if (num_folds < 1 || num_cv_repetitions < 1 || max_ranking_size < 1 || !dataset_path || !out_prefix) {
        snprintf(g_err, sizeof g_err, "bad epistasis arguments");
        return HPGV_ERR_INVALID;
    }
    FILE *f = fopen(dataset_path, "rb");
    uint32_t head[3];
    if (!f || fread(head, sizeof(uint32_t), 3, f) != 3) { if (f) fclose(f); snprintf(g_err, sizeof g_err, "cannot read the dataset %s", dataset_path); return HPGV_ERR_INVALID; }
    const size_t V = head[0], nA = head[1], nU = head[2], bytes = V * (nA + nU);
    uint8_t *data = (uint8_t *)malloc(bytes + 1);
    if (!data || fread(data, 1, bytes, f) != bytes) { fclose(f); free(data); snprintf(g_err, sizeof g_err, "the dataset %s is shorter than its header says", dataset_path); return HPGV_ERR_INVALID; }
    fclose(f);
    rc = hpgv_epi_set_dataset(g_ctx, data, (int)V, (int)nA, (int)nU);
    free(data);
    if (rc) return host_fail("hpgv_epi_set_dataset", rc);
    const size_t N = (size_t)max_ranking_size, K = (size_t)num_folds, O = (size_t)order;
    int32_t *ci = (int32_t *)malloc(sizeof(int32_t) * K * N), *cj = (int32_t *)malloc(sizeof(int32_t) * K * N), *cnt = (int32_t *)malloc(sizeof(int32_t) * K);
    int32_t *ck = (int32_t *)malloc(sizeof(int32_t) * K * N);
    int32_t *cn = (int32_t *)malloc(sizeof(int32_t) * K * N * O);                    /* orders 4, 5: the combinations themselves */
    uint32_t *risky = (uint32_t *)calloc(K * N * EPI_MASK_WORDS, sizeof(uint32_t));
    double *acc = (double *)malloc(sizeof(double) * K * N);
    epi_model_t *all = (epi_model_t *)malloc(sizeof(epi_model_t) * (K * N + 1));
    char *path = (char *)malloc(strlen(out_prefix) + 32);
    if (!ci || !cj || !ck || !cn || !cnt || !risky || !acc || !all || !path) rc = HPGV_ERR_NOMEM;
    for (int r = 0; r < num_cv_repetitions && !rc; r++) {
        unsigned int *sizes = NULL;
        int **folds = get_k_folds((unsigned)nA, (unsigned)nU, (unsigned)num_folds, &sizes);
        uint8_t *masks = folds ? get_k_folds_masks((unsigned)nA, (unsigned)nU, (unsigned)num_folds, folds, sizes) : NULL;
        if (!masks) rc = HPGV_ERR_NOMEM;
        if (!rc && (rc = hpgv_epi_set_fold_masks(g_ctx, masks, num_folds))) host_fail("hpgv_epi_set_fold_masks", rc);
        /* orders 2 and 3: the tile scans (one mask word per model); any other order: the listed-combination kernel */
        if (!rc && order == 2 && (rc = hpgv_epi_rank_pairs(g_ctx, eval_subset, max_ranking_size, ci, cj, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_pairs", rc);
        if (!rc && order == 3 && (rc = hpgv_epi_rank_triples(g_ctx, eval_subset, max_ranking_size, ci, cj, ck, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_triples", rc);
        if (!rc && order > 3 && (rc = hpgv_epi_rank_order(g_ctx, order, eval_subset, max_ranking_size, cn, acc, risky, cnt, NULL))) host_fail("hpgv_epi_rank_order", rc);
        if (folds) { for (int k = 0; k < num_folds; k++) free(folds[k]); free(folds); }
        free(sizes); free(masks);
        if (rc) break;
        size_t n = 0;
        for (size_t k = 0; k < K; k++)
            for (int e = 0; e < cnt[k]; e++) {
                const size_t o = k * N + (size_t)e;
                epi_model_t m;
                memset(&m, 0, sizeof m);
                for (int s = 0; s < EPI_ORDER_MAX; s++) m.c[s] = -1;
                if (order <= 3) { m.c[0] = ci[o]; m.c[1] = cj[o]; if (order == 3) m.c[2] = ck[o]; m.risky[0] = risky[o]; }
                else { for (size_t s = 0; s < O; s++) m.c[s] = cn[o * O + s]; memcpy(m.risky, risky + o * EPI_MASK_WORDS, sizeof m.risky); }
                m.count = 1; m.accuracy = acc[o];
                all[n++] = m;
            }
        qsort(all, n, sizeof *all, cmp_model_comb);                  /* stable enough: equal combinations differ only by fold */
        size_t m = 0;
        for (size_t e = 0; e < n; e++) {
            if (m > 0 && !cmp_model_comb(&all[m - 1], &all[e])) { all[m - 1].accuracy += all[e].accuracy; all[m - 1].count += 1; }
            else all[m++] = all[e];
        }
        for (size_t e = 0; e < m; e++) all[e].accuracy /= num_folds;
        qsort(all, m, sizeof *all, eval_mode == 1 ? cmp_model_cva : cmp_model_cvc);
        sprintf(path, "%s.cv%d.epi", out_prefix, r + 1);
        FILE *fd = fopen(path, "w");
        if (!fd) { snprintf(g_err, sizeof g_err, "cannot create %s", path); rc = HPGV_ERR_INVALID; break; }
        fprintf(fd, "#CROSS VALIDATION %d\n#COMBINATIONS OF: %d SNPs\n", r + 1, order);
        fprintf(fd, eval_mode == 1 ? "#EVALUATION MODE: Cross-validation accuracy\n" : "#EVALUATION MODE: Cross-validation consistency\n");
        fprintf(fd, eval_subset == HPGV_EPI_TRAINING ? "#EVALUATION PARTITION: Training\n" : "#EVALUATION PARTITION: Testing\n");
        fprintf(fd, "#POSITION\tSNPs\tGENOTYPES\tCV-C\tCV-A\n");
        int cells = 1;
        for (int s = 0; s < order; s++) cells *= 3;
        for (size_t e = 0; e < m && e < N; e++) {
            /* epistasis_report.c:62-77: "( i, j, ... )"; a risky cell "(g0-g1, g2, ..., gn), " -- '-' after the first genotype,
             * ", " after the ones between, the last one closes */
            fprintf(fd, "%d\t(", (int)e + 1);
            for (int s = 0; s < order - 1; s++) fprintf(fd, " %d,", all[e].c[s]);
            fprintf(fd, " %d )\t", all[e].c[order - 1]);
            for (int c = 0; c < cells; c++) {
                if (!(all[e].risky[c >> 5] >> (c & 31) & 1u)) continue;
                int g[EPI_ORDER_MAX], q = c;
                for (int s = order - 1; s >= 0; s--) { g[s] = q % 3; q /= 3; }         /* the last SNP varies fastest */
                fprintf(fd, "(%d-", g[0]);
                for (int s = 1; s < order - 1; s++) fprintf(fd, "%d, ", g[s]);
                fprintf(fd, "%d), ", g[order - 1]);
            }
            fprintf(fd, "%d\t%.3f\n", all[e].count, all[e].accuracy);
        }
        fclose(fd);
    }
    free(ci); free(cj); free(ck); free(cn); free(cnt); free(risky); free(acc); free(all); free(path);
    if (rc == HPGV_ERR_NOMEM) snprintf(g_err, sizeof g_err, "out of memory");
    return rc;
}

/* ------------------------------------------------------------------------ */
/* worker pools of the file-level code: persistent threads, tasks handed out  */
/* by a counter (a nested OpenMP team is created anew on every entry, which   */
/* costs milliseconds per batch on a many-core host)                           */
/* ------------------------------------------------------------------------ */

static int default_io_threads(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    FILE *q = fopen("/sys/fs/cgroup/cpu.max", "r");         /* a container may be allotted far fewer CPUs than it sees */
    if (q) {
        long quota = 0, period = 0;
        if (fscanf(q, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < n) n = quota / period > 0 ? quota / period : 1;
        fclose(q);
    }
    int t = n >= 32 ? 16 : n >= 16 ? (int)n : n >= 4 ? (int)(n / 2) : 1;
    const char *e = getenv("HPGV_IO_THREADS");
    if (e && atoi(e) > 0) t = atoi(e) < 64 ? atoi(e) : 64;
    return t;
}

/* ------------------------------------------------------------------------ */
/* result ordering: `sort -k1,1h -k2,2n` in process                          */
/* ------------------------------------------------------------------------ */

typedef struct { const char *line; int neg1, order1; long double v1; int neg2; long double v2; } sort_key_t;

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

static void make_key(const char *line, sort_key_t *k) {
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

static int cmp_keys(const void *pa, const void *pb) {
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

/* ---- NUMA: a run's threads and page-locked buffers go to the node the GPU hangs off ------------------------------
 * Measured on a two-socket MI355X host: 580 k variants/s with the process on the GPU's node, 334 k on the other one,
 * anything in between when left to the scheduler (the copy out of the page cache into the staging buffers and the
 * H2D DMA then cross the socket link).  The calling thread's affinity is narrowed to the node's CPUs for the run
 * (the threads it creates inherit it, its allocations are first touched there) and put back afterwards.
 * HPGV_NO_NUMA_BIND=1 switches this off. */
static int numa_bind_to_device(cpu_set_t *saved) {
    if (getenv("HPGV_NO_NUMA_BIND") || !g_ctx) return 0;
    int node = -1;
    if (hpgv_device_numa_node(g_ctx, &node) != HPGV_OK || node < 0) return 0;
    char path[96], list[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const int ok = fgets(list, sizeof list, f) != NULL;
    fclose(f);
    if (!ok || sched_getaffinity(0, sizeof *saved, saved) != 0) return 0;
    cpu_set_t want;
    CPU_ZERO(&want);
    int n_set = 0;
    for (char *p = list; *p;) {                          /* "0-63,128-191" */
        char *e;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET((int)c, saved)) { CPU_SET((int)c, &want); n_set++; }
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    if (n_set == 0) return 0;                            /* the caller is not allowed on that node: leave it alone */
    return sched_setaffinity(0, sizeof want, &want) == 0;
}
static void numa_unbind(const cpu_set_t *saved, int bound) { if (bound) (void)sched_setaffinity(0, sizeof *saved, saved); }

/* ------------------------------------------------------------------------ */
/* file-level runners: VCF + PED in, sorted TSV out                           */
/* ------------------------------------------------------------------------ */

typedef struct {
    int n;
    char **fid, **iid, **pat, **mat, **phe;             /* phe: the PHENO column as written (the stats tool's variable) */
    int *sex, *pheno;
    char *blob;
} ped_table_t;

static void ped_table_free(ped_table_t *p) {
    free(p->fid); free(p->iid); free(p->pat); free(p->mat); free(p->phe); free(p->sex); free(p->pheno); free(p->blob);
    memset(p, 0, sizeof *p);
}

static char *next_ws_token(char **p) {
    char *s = *p;
    while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') s++;
    if (!*s) return NULL;
    char *e = s;
    while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++;
    if (*e) { *e = 0; e++; }
    *p = e;
    return s;
}

static int ped_table_read(const char *path, ped_table_t *ped) {
    memset(ped, 0, sizeof *ped);
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot open PED file %s", path); return HPGV_ERR_INVALID; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    ped->blob = (char *)malloc((size_t)sz + 1);
    if (!ped->blob || fread(ped->blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); ped_table_free(ped); return HPGV_ERR_NOMEM; }
    fclose(f);
    ped->blob[sz] = 0;
    int cap = 0;
    for (long i = 0; i < sz; i++) if (ped->blob[i] == '\n') cap++;
    cap += 2;
    ped->fid = (char **)malloc(sizeof(char *) * (size_t)cap); ped->iid = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->pat = (char **)malloc(sizeof(char *) * (size_t)cap); ped->mat = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->phe = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->sex = (int *)malloc(sizeof(int) * (size_t)cap); ped->pheno = (int *)malloc(sizeof(int) * (size_t)cap);
    if (!ped->fid || !ped->iid || !ped->pat || !ped->mat || !ped->phe || !ped->sex || !ped->pheno) {
        ped_table_free(ped);
        snprintf(g_err, sizeof g_err, "out of memory reading the PED file");
        return HPGV_ERR_NOMEM;
    }
    char *line = ped->blob;
    while (line && *line) {
        char *eol = strchr(line, '\n');
        if (eol) *eol = 0;
        if (*line && *line != '#') {
            char *p = line;
            char *a = next_ws_token(&p), *b = next_ws_token(&p), *c = next_ws_token(&p), *d = next_ws_token(&p);
            char *e = next_ws_token(&p), *g = next_ws_token(&p);
            if (!(a && b && c && d && e && g)) {         /* a row without FID IID PAT MAT SEX PHENO is damaged input, not a sample to skip */
                snprintf(g_err, sizeof g_err, "PED row %d has fewer than 6 columns", ped->n + 1);
                ped_table_free(ped);
                return HPGV_ERR_INVALID;
            }
            if (ped->n < cap) {
                ped->fid[ped->n] = a; ped->iid[ped->n] = b; ped->pat[ped->n] = c; ped->mat[ped->n] = d; ped->phe[ped->n] = g;
                /* SEX and PHENO are read as numbers (the reference keeps the phenotype as a float, individual->variable):
                 * "2" and "2.0" are the same label; anything that is not 1 / 2 -- 0, -9, text -- is unknown / other */
                char *end;
                const double sx = strtod(e, &end);
                const int sx_ok = end != e && *end == 0;
                ped->sex[ped->n] = (sx_ok && sx == 1.0) ? HPGV_SEX_MALE : (sx_ok && sx == 2.0) ? HPGV_SEX_FEMALE : HPGV_SEX_UNKNOWN;
                const double ph = strtod(g, &end);
                const int ph_ok = end != g && *end == 0;
                ped->pheno[ped->n] = (ph_ok && ph == 2.0) ? HPGV_COND_AFFECTED : (ph_ok && ph == 1.0) ? HPGV_COND_UNAFFECTED : HPGV_COND_OTHER;
                ped->n++;
            }
        }
        line = eol ? eol + 1 : NULL;
    }
    return HPGV_OK;
}

/* ---- byte sources: plain file (pread by a small thread team), BGZF (blocks inflated in parallel),
 *      generic gzip (one zlib stream) -- shared_options.c:60-61 `--compression gzip|bgzip` ------------ */
enum { SRC_RAW = 0, SRC_BGZF = 1, SRC_GZIP = 2 };
static hpgv_run_filters_t g_filters = { -1.0, -1.0, -1, -1, -1.0 };     /* hpgv_run_set_filters; negative = off */
static double g_run_times[6];                           /* last run: read, engine, write, sort, total seconds, batches */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
typedef struct {
    int kind, fd;
    off_t pos, size;                                    /* RAW: next unread byte, file size */
    const unsigned char *map; size_t map_pos;           /* BGZF: the mapped compressed file, next block */
    unsigned char *pend; size_t pend_len, pend_pos;     /* BGZF: a block inflated aside because the caller's room was short */
    size_t *blk;                                        /* BGZF: per-call block table (offset, length, destination, size) */
    io_pool_t *pool;                                    /* team for the pread segments / the block inflation (may be NULL) */
    char *job_buf; size_t job_want; int job_bad;        /* the job the team is working on */
    gzFile gz;                                          /* GZIP */
    /* BGZF decoded on the GPU: the whole file's text in device memory, handed out window by window */
    void *d_comp, *d_tab, *d_text, *d_status, *rstream, *cstream; size_t dev_len, dev_pos; int gpu_tried;
    uint64_t *g_in_off, *g_out_off; uint32_t *g_in_len, *g_out_len; size_t g_nb, g_done;      /* the stager's block tables */
    size_t dev_ready;                                   /* text bytes decoded so far (under g_mu) */
    size_t d_text_cap; int d_text_kind;
    size_t text_est;                                     /* about how much text the file holds (known when the stage has chosen its path) */
    int dev_len_known;                                  /* 0 while the stager is still finding the file's blocks (under g_mu) */
    void *d_scan;                                       /* the streaming stager's tables, statuses and scan scratch */
    int c_low;                                          /* cstream (and the slots' streams) have the lowest priority */
    pthread_t g_thread; pthread_mutex_t g_mu; pthread_cond_t g_cv; int g_started, g_sync, g_err, g_finished;
    /* the uploader: the compressed file goes up from the moment it is opened, beside the walk of its block headers */
    pthread_t u_thread; int u_started, u_cancel, u_err; size_t up_done;      /* bytes [0, up_done) are on the device (under g_mu) */
    /* a PART of a bgzip file staged on one member of a group context (bgzf_parts_stage): the bytes [file_off, file_off + size)
     * of the file, `map` pointing at its first byte, everything else as for a whole file.  ctx == NULL: an ordinary source */
    hpgv_ctx *ctx; int member; off_t file_off; int is_part;
    size_t map_len; const unsigned char *map_base;      /* the mapping as it was made (a part's is its file's) */
    struct src_parts *mp;                               /* the file's parts, when it is staged in parts (this source is part 0) */
} source_t;
/* the parts of one bgzip file, in file order; the reader walks them and joins the line that straddles two of them */
typedef struct src_parts {
    int n, cur;
    source_t *p[MEMBERS_MAX];                           /* p[0] = the reader's own source */
    char *seam; size_t seam_len, seam_cap;              /* the line across the seam in front of part `cur`, waiting to be handed out */
    off_t whole_size;
} src_parts_t;
#define SRC_CTX(s) ctx_use((s)->ctx, (s)->member)

static int bgzf_block(const unsigned char *p, size_t avail, size_t *bsize, size_t *cdata_off, size_t *isize) {
    if (avail < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
    size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8), off = 12, end = 12 + xlen;
    if (end > avail) return 0;
    size_t bs = 0;
    while (off + 4 <= end) {
        size_t slen = (size_t)p[off + 2] | ((size_t)p[off + 3] << 8);
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2 && off + 6 <= end) bs = ((size_t)p[off + 4] | ((size_t)p[off + 5] << 8)) + 1;
        off += 4 + slen;
    }
    if (bs < end + 8 || bs > avail) return 0;
    *bsize = bs; *cdata_off = end;
    *isize = (size_t)p[bs - 4] | ((size_t)p[bs - 3] << 8) | ((size_t)p[bs - 2] << 16) | ((size_t)p[bs - 1] << 24);
    return 1;
}

static int source_open(source_t *s, const char *path) {
    memset(s, 0, sizeof *s);
    s->fd = open(path, O_RDONLY);
    struct stat st;
    if (s->fd < 0 || fstat(s->fd, &st) != 0) { if (s->fd >= 0) close(s->fd); return 1; }
    s->size = st.st_size;
    unsigned char head[18];
    ssize_t got = pread(s->fd, head, sizeof head, 0);
    if (got >= 2 && head[0] == 31 && head[1] == 139) {
        size_t bs, co, is;
        s->map = (const unsigned char *)mmap(NULL, (size_t)s->size, PROT_READ, MAP_PRIVATE, s->fd, 0);
        s->map_base = s->map; s->map_len = (size_t)s->size;
        if (s->map != MAP_FAILED && bgzf_block(s->map, (size_t)s->size, &bs, &co, &is)) { s->kind = SRC_BGZF; return 0; }
        if (s->map != MAP_FAILED) munmap((void *)s->map, (size_t)s->size);
        s->map = NULL; s->map_base = NULL; s->map_len = 0;
        s->gz = gzdopen(dup(s->fd), "rb");
        if (!s->gz) { close(s->fd); return 1; }
        gzbuffer(s->gz, 1u << 20);
        s->kind = SRC_GZIP;
    }
    return 0;
}

static void source_close(source_t *s) {
    if (s->mp) {                                                     /* the other parts of the file (this one is part 0) */
        for (int k = 1; k < s->mp->n; k++) if (s->mp->p[k]) { source_close(s->mp->p[k]); free(s->mp->p[k]); }
        s->size = s->mp->whole_size;
        free(s->mp->seam); free(s->mp); s->mp = NULL;
    }
    const ctx_saved_t saved = SRC_CTX(s);
    if (s->g_started) pthread_join(s->g_thread, NULL);               /* the stager reads the mapping: it goes first */
    if (s->u_started) {
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (s->kind == SRC_BGZF && s->map_base && !s->is_part) munmap((void *)s->map_base, s->map_len);
    free(s->pend); free(s->blk);
    if (s->g_sync) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); }
    free(s->g_in_off); free(s->g_out_off); free(s->g_in_len); free(s->g_out_len);
    if (g_ctx) {
        if (s->d_comp) (void)hpgv_dev_free(CTX, s->d_comp);
        if (s->d_tab) (void)hpgv_dev_free(CTX, s->d_tab);
        if (s->d_status) (void)hpgv_dev_free(CTX, s->d_status);
        if (s->d_text) dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind);
        if (s->d_scan) (void)hpgv_dev_free(CTX, s->d_scan);
        stream_put(0, s->rstream);
        stream_put(s->c_low, s->cstream);
    }
    if (s->kind == SRC_GZIP && s->gz) gzclose(s->gz);
    if (s->fd >= 0) close(s->fd);
    ctx_back(saved);
}

/* ---- raw DEFLATE (RFC 1951) decoder for BGZF blocks ------------------------------------------------------------
 * zlib's inflate does 0.7 GB/s per thread on VCF text and bounded the bgzip runs; this one keeps a 64-bit bit
 * buffer, decodes through two-level tables (10-bit first level for literals / lengths, 8-bit for distances) and copies
 * matches eight bytes at a time.  Input and output sizes are known up front (a BGZF block states both).  It returns
 * non-zero on ANY irregularity -- the caller then decodes the block again with zlib, so this is an accelerator, never
 * the arbiter of what a valid block is. */
typedef struct { uint32_t e[1024 + 592 + 8]; } fi_lit_t;        /* entry: bits 0-7 code length, 8-15 kind / extra bits, 16-31 value */
typedef struct { uint32_t e[256 + 400 + 8]; } fi_dist_t;
enum { FI_LIT = 0x00, FI_LEN = 0x40, FI_EOB = 0x80, FI_SUB = 0xC0, FI_BAD = 0xFF };

/* canonical Huffman code lengths -> decode table with `root` first-level bits.  Entries: len in bits 0-7 (for a
 * sub-table link: the first level's bits), kind | extra bits in 8-15, value in 16-31.  Returns 0 when the code is
 * complete (or the single-code case deflate allows for distances), non-zero otherwise. */
static int fi_build(uint32_t *tab, int tab_cap, int root, const uint8_t *lens, int n, int is_dist) {
    static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint8_t len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint8_t dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    int count[16] = {0}, maxlen = 0, used = 0;
    for (int i = 0; i < n; i++) { count[lens[i]]++; if (lens[i] > maxlen) maxlen = lens[i]; if (lens[i]) used++; }
    const int first = 1 << root;
    for (int i = 0; i < first; i++) tab[i] = FI_BAD << 8;
    if (used == 0) return is_dist ? 0 : 1;                          /* no distance codes at all: literals only */
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return 1; }
    if (left > 0 && !(used == 1 && maxlen == 1)) return 1;           /* incomplete (the one-code case is allowed) */
    int next_code[16], code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next_code[l] = code; }
    int sub_next = first;                                           /* where the next sub-table goes */
    /* sub-tables: one per distinct first-level prefix among the long codes, sized for the longest code below it.
     * Symbols are visited in (length, symbol) order, i.e. in canonical code order: prefixes arrive grouped. */
    int sub_prefix = -1, sub_bits = 0, sub_base = 0;
    int remaining[16];
    for (int l = 0; l <= 15; l++) remaining[l] = count[l];
    for (int l = 1; l <= 15; l++) {
        for (int sym = 0; sym < n; sym++) {
            if (lens[sym] != l) continue;
            const int c = next_code[l]++;
            uint32_t rev = 0;                                       /* deflate packs codes starting from the MSB: reverse */
            for (int b = 0; b < l; b++) rev |= (uint32_t)((c >> b) & 1) << (l - 1 - b);
            uint32_t kind, value;
            if (is_dist) { if (sym >= 30) { kind = FI_BAD; value = 0; } else { kind = FI_LEN | dist_extra[sym]; value = dist_base[sym]; } }
            else if (sym < 256) { kind = FI_LIT; value = (uint32_t)sym; }
            else if (sym == 256) { kind = FI_EOB; value = 0; }
            else if (sym > 285) { kind = FI_BAD; value = 0; }         /* 286, 287: in the fixed code, never valid in data */
            else { kind = FI_LEN | len_extra[sym - 257]; value = len_base[sym - 257]; }
            if (l <= root) {
                const uint32_t ent = (value << 16) | (kind << 8) | (uint32_t)l;
                for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) tab[i] = ent;
            } else {
                const int prefix = (int)(rev & (uint32_t)(first - 1));
                if (prefix != sub_prefix) {
                    /* bits below this prefix: as many as the codes still to come under it need (zlib's inflate_table rule;
                     * remaining[] counts the codes of each length not placed yet, this one included) */
                    int bits = l - root, room = 1 << bits;
                    while (bits + root < maxlen) {
                        room -= remaining[bits + root];
                        if (room <= 0) break;
                        bits++; room <<= 1;
                    }
                    if (sub_next + (1 << bits) > tab_cap) return 1;
                    sub_prefix = prefix; sub_bits = bits; sub_base = sub_next; sub_next += 1 << bits;
                    for (int i = 0; i < (1 << bits); i++) tab[sub_base + i] = FI_BAD << 8;
                    tab[prefix] = ((uint32_t)sub_base << 16) | ((uint32_t)(FI_SUB | sub_bits) << 8) | (uint32_t)root;
                }
                const uint32_t ent = (value << 16) | (kind << 8) | (uint32_t)(l - root);
                for (uint32_t i = rev >> root; i < (1u << sub_bits); i += 1u << (l - root)) tab[sub_base + i] = ent;
            }
            remaining[l]--;
        }
    }
    return 0;
}

#define FI_REFILL() do {                                                                             \
        if (in + 8 <= in_end) { uint64_t w; memcpy(&w, in, 8); bitbuf |= w << bitcnt; in += (63 - bitcnt) >> 3; bitcnt |= 56; } \
        else while (bitcnt <= 56 && in < in_end) { bitbuf |= (uint64_t)*in++ << bitcnt; bitcnt += 8; } \
    } while (0)
#define FI_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))

static int fast_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    const unsigned char *in_end = in + in_len;
    unsigned char *const out0 = out, *const out_end = out + out_len;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    fi_lit_t lit_tab; fi_dist_t dist_tab;                          /* 9 KB of stack */
    fi_lit_t *const LT = &lit_tab;
    fi_dist_t *const DT = &dist_tab;
    int rc = 1, last = 0;
    while (!last) {
        FI_REFILL();
        if (bitcnt < 3) goto done;
        last = (int)(bitbuf & 1); const int type = (int)((bitbuf >> 1) & 3); FI_TAKE(3);
        if (type == 0) {                                            /* stored */
            FI_TAKE(bitcnt & 7);
            FI_REFILL();
            if (bitcnt < 32) goto done;
            const unsigned len = (unsigned)(bitbuf & 0xFFFF), nlen = (unsigned)((bitbuf >> 16) & 0xFFFF);
            FI_TAKE(32);
            if ((len ^ nlen) != 0xFFFF) goto done;
            in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0;              /* whole bytes still in the buffer go back */
            if ((size_t)(in_end - in) < len || (size_t)(out_end - out) < len) goto done;
            memcpy(out, in, len); out += len; in += len;
            continue;
        }
        if (type == 3) goto done;
        uint8_t lens[320];
        int nlit, ndist;
        if (type == 1) {
            nlit = 288; ndist = 32;                                 /* the fixed distance code has 32 five-bit codes (30, 31 never valid) */
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            for (int i = 0; i < 32; i++) lens[288 + i] = 5;
        } else {
            FI_REFILL();
            if (bitcnt < 14) goto done;
            nlit = (int)(bitbuf & 31) + 257; ndist = (int)((bitbuf >> 5) & 31) + 1; const int ncode = (int)((bitbuf >> 10) & 15) + 4;
            FI_TAKE(14);
            if (nlit > 286 || ndist > 30) goto done;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) { FI_REFILL(); if (bitcnt < 3) goto done; cl[order[i]] = (uint8_t)(bitbuf & 7); FI_TAKE(3); }
            uint32_t ct[128 + 8];
            if (fi_build(ct, 128, 7, cl, 19, 0) ) {                  /* code-length code: at most 7 bits, one level; kinds unused */
                /* fi_build classifies symbols as literals here (all < 256): a code with a single symbol is rejected */
                goto done;
            }
            int i = 0;
            while (i < nlit + ndist) {
                FI_REFILL();
                const uint32_t ent = ct[bitbuf & 127];
                const int l = (int)(ent & 0xFF);
                if (((ent >> 8) & 0xFF) == FI_BAD || l > bitcnt) goto done;
                FI_TAKE(l);
                const int sym = (int)(ent >> 16);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0 || bitcnt < 2) goto done; val = lens[i - 1]; rep = 3 + (int)(bitbuf & 3); FI_TAKE(2); }
                else if (sym == 17) { if (bitcnt < 3) goto done; rep = 3 + (int)(bitbuf & 7); FI_TAKE(3); }
                else { if (bitcnt < 7) goto done; rep = 11 + (int)(bitbuf & 127); FI_TAKE(7); }
                if (i + rep > nlit + ndist) goto done;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) goto done;                           /* no end-of-block code */
            memmove(lens + 288, lens + nlit, (size_t)ndist);        /* distances at a fixed place */
            for (int k = nlit; k < 288; k++) lens[k] = 0;
        }
        if (fi_build(LT->e, 1024 + 592, 10, lens, type == 1 ? 288 : nlit, 0)) goto done;
        if (fi_build(DT->e, 256 + 400, 8, lens + 288, ndist, 1)) goto done;
        for (;;) {                                                  /* symbols of this block */
            FI_REFILL();
            uint32_t ent = LT->e[bitbuf & 1023];
            if ((ent >> 8 & 0xC0) == FI_SUB && (ent >> 8 & 0xFF) != FI_BAD) {
                const int sb = (int)(ent >> 8) & 0x3F;
                ent = LT->e[(ent >> 16) + ((bitbuf >> 10) & ((1u << sb) - 1))];
                FI_TAKE(10);
            }
            int kind = (int)(ent >> 8) & 0xFF, l = (int)(ent & 0xFF);
            if (kind == FI_BAD || l > bitcnt) goto done;
            FI_TAKE(l);
            if (kind == FI_LIT) {
                if (out >= out_end) goto done;
                *out++ = (unsigned char)(ent >> 16);
                /* up to two more first-level literals out of the same refill (56 bits hold three 15-bit codes) */
                ent = LT->e[bitbuf & 1023]; kind = (int)(ent >> 8) & 0xFF; l = (int)(ent & 0xFF);
                if (kind != FI_LIT || l > bitcnt || out >= out_end) continue;
                FI_TAKE(l); *out++ = (unsigned char)(ent >> 16);
                ent = LT->e[bitbuf & 1023]; kind = (int)(ent >> 8) & 0xFF; l = (int)(ent & 0xFF);
                if (kind != FI_LIT || l > bitcnt || out >= out_end) continue;
                FI_TAKE(l); *out++ = (unsigned char)(ent >> 16);
                continue;
            }
            if (kind == FI_EOB) break;
            const int xb = kind & 0x3F;                             /* a length: extra bits, then the distance */
            if (xb > bitcnt) goto done;
            const unsigned len = (unsigned)(ent >> 16) + (unsigned)(bitbuf & ((1u << xb) - 1));
            FI_TAKE(xb);
            FI_REFILL();
            uint32_t de = DT->e[bitbuf & 255];
            if ((de >> 8 & 0xC0) == FI_SUB && (de >> 8 & 0xFF) != FI_BAD) {
                const int sb = (int)(de >> 8) & 0x3F;
                de = DT->e[(de >> 16) + ((bitbuf >> 8) & ((1u << sb) - 1))];
                FI_TAKE(8);
            }
            const int dk = (int)(de >> 8) & 0xFF, dl = (int)(de & 0xFF);
            if (dk == FI_BAD || (dk & 0xC0) != FI_LEN || dl > bitcnt) goto done;
            FI_TAKE(dl);
            const int dxb = dk & 0x3F;
            if (dxb > bitcnt) goto done;
            const size_t dist = (size_t)(de >> 16) + (size_t)(bitbuf & ((1u << dxb) - 1));
            FI_TAKE(dxb);
            if (dist > (size_t)(out - out0) || (size_t)(out_end - out) < len) goto done;
            const unsigned char *src = out - dist;
            if ((size_t)(out_end - out) >= len + 8) {               /* eight bytes at a time (may write up to 7 past len) */
                unsigned char *d = out;
                const unsigned char *e = out + len;
                if (dist < 8) {                                      /* a short period ("0/0\t"...): lay down a multiple of it that */
                    static const uint8_t big_of[8] = {0, 8, 8, 9, 8, 10, 12, 14};      /* is >= 8 byte by byte, then copy from there */
                    const unsigned big = big_of[dist], pre = len < big ? len : big;
                    for (unsigned k = 0; k < pre; k++) d[k] = src[k];
                    d += pre; src = d - big;
                }
                while (d < e) { uint64_t w; memcpy(&w, src, 8); memcpy(d, &w, 8); src += 8; d += 8; }
            } else for (unsigned k = 0; k < len; k++) out[k] = src[k];
            out += len;
        }
    }
    rc = out == out_end ? 0 : 1;
done:
    return rc;
}
#undef FI_REFILL
#undef FI_TAKE

/* exported for the tests: raw DEFLATE `in` -> exactly out_len bytes; 0 = done, non-zero = not decodable by the fast path */
int hpgv_host_inflate_raw(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    return fast_inflate(in, in_len, out, out_len);
}

/* HPGV_BGZF_VERIFY (default 1): every BGZF block's text is checked against the CRC-32 of its trailer, as htslib's bgzf reader
 * and zlib's gzread do -- a damaged stream can still inflate to ISIZE bytes.  On the device path the check runs on the device
 * (hpgv_bgzf_verify_dev); a block it rejects comes here like any block the device decoder refused, and fails the run. */
static char g_input_err[192];
static int bgzf_verify_on(void) {
    const char *e = getenv("HPGV_BGZF_VERIFY");
    return !e || atoi(e) != 0;
}
/* in[clen .. clen + 4) is the block's CRC-32 (the BGZF trailer follows the payload) */
static int block_crc_bad(const unsigned char *in, size_t clen, const unsigned char *out, size_t isize) {
    const unsigned char *t = in + clen;
    const uint32_t stored = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)isize) == stored) return 0;
    /* written by a reader / stager thread while the run may be reporting what it saw fail downstream: kept apart, and put in
     * front when the run ends with an error (run_file) */
    snprintf(g_input_err, sizeof g_input_err, "bgzip input: a block's text does not have the CRC-32 its trailer gives (damaged file)");
    return 1;
}
static int inflate_block(const unsigned char *in, size_t clen, unsigned char *out, size_t isize) {
    if (!getenv("HPGV_ZLIB_INFLATE") && fast_inflate(in, clen, out, isize) == 0)       /* anything unusual: zlib decides */
        return bgzf_verify_on() ? block_crc_bad(in, clen, out, isize) : 0;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return 1;
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)clen;
    zs.next_out = (Bytef *)out; zs.avail_out = (uInt)isize;
    int bad = inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != isize;
    inflateEnd(&zs);
    if (!bad && bgzf_verify_on()) bad = block_crc_bad(in, clen, out, isize);
    return bad;
}

enum { PREAD_SEG = 4 << 20, INFLATE_GROUP = 8, MAXB = 1 << 16 };
static void source_task_pread(void *v, int k) {
    source_t *s = (source_t *)v;
    size_t off = (size_t)k * PREAD_SEG, len = off + PREAD_SEG <= s->job_want ? (size_t)PREAD_SEG : s->job_want - off;
    while (len > 0) {
        ssize_t got = pread(s->fd, s->job_buf + off, len, s->pos + (off_t)off);
        if (got <= 0) { __atomic_store_n(&s->job_bad, 1, __ATOMIC_RELAXED); return; }
        off += (size_t)got; len -= (size_t)got;
    }
}
static void source_task_inflate(void *v, int g) {
    source_t *s = (source_t *)v;
    const size_t *b_in = s->blk, *b_clen = s->blk + MAXB, *b_out = s->blk + 2 * MAXB, *b_isize = s->blk + 3 * MAXB;
    const int nb = (int)s->job_want;
    for (int k = g * INFLATE_GROUP; k < nb && k < (g + 1) * INFLATE_GROUP; k++) {
        if (b_isize[k] == 0) continue;                                /* e.g. the BGZF end-of-file marker */
        if (inflate_block(s->map + b_in[k], b_clen[k], (unsigned char *)s->job_buf + b_out[k], b_isize[k]))
            __atomic_store_n(&s->job_bad, 1, __ATOMIC_RELAXED);
    }
}

/* BGZF on the GPU: the file's blocks are decoded on the device (hpgv_inflate_blocks_dev: one lane per block, tens of
 * thousands of blocks per launch; 8 GB of VCF text in 0.1 s) and the text stays in device memory.  The reader copies it
 * out window by window for the result writers and the engine tokenizes the device copy in place (hpgv_text_alias) --
 * the compressed bytes are all that goes up the bus, read from the file with pread into a page-locked buffer (the
 * mapping is not touched: faulting a gigabyte in and unmapping it costs more than the decoding).  A stager thread
 * uploads and decodes the file in stretches: 4 096 blocks first, decoded alone, so that the header reader and the
 * pipeline start after the time one block takes; then 32 768 and more, several launches side by side on streams of
 * their own.  A block the device decoder refuses is decoded by the host and patched in.  No memory, a file of more
 * than 48 GB of text or fewer than 256 blocks leave the CPU path in charge.  HPGV_NO_GPU_INFLATE=1 switches it off. */
enum { GPU_STRETCH = 32768, GPU_AHEAD_BYTES = 768 << 20 };
/* The upload: out of pageable memory a copy runs at a fifth of the bus rate, so the file goes through a page-locked ring
 * (pread, not the mapping: faulting a gigabyte in and unmapping it again costs ~90 ms).  Readers take the file's 4 MB
 * segments in turn and fill the ring's slots; the uploader copies the slots up in file order and announces every segment
 * that has arrived (s->up_done, under g_mu).  Nobody waits at a barrier: a reader waits only for its slot to be free, the
 * copier only for the next segment to be filled.  (Halves of a buffer filled by a team, with a barrier per half, left
 * the bus at 18 - 28 GB/s beside the pipeline's own threads.) */
enum { UP_SEG_DEFAULT = 4 << 20, UP_SLOTS = 16, UP_READERS_MAX = 12, UP_INFLIGHT = 2 };
static size_t g_up_seg = UP_SEG_DEFAULT;                 /* bytes per segment (HPGV_UPLOAD_SEGMENT_MB: diagnosis) */
#define UP_SEG g_up_seg
static size_t pread_full(int fd, void *buf, size_t n, size_t pos);
typedef struct {
    source_t *s; char *pin; size_t n_seg;
    pthread_mutex_t mu; pthread_cond_t cv;
    size_t next;                                         /* the next segment a reader takes */
    size_t copied;                                       /* segments [0, copied) are on the device: slot i % UP_SLOTS is free for segment i < copied + UP_SLOTS */
    unsigned char filled[UP_SLOTS];
    int bad, stop;
} up_ring_t;
static void *up_reader(void *v) {
    up_ring_t *r = (up_ring_t *)v;
    for (;;) {
        pthread_mutex_lock(&r->mu);
        const size_t i = r->next < r->n_seg ? r->next++ : (size_t)-1;
        while (i != (size_t)-1 && !r->stop && !r->bad && i >= r->copied + UP_SLOTS) pthread_cond_wait(&r->cv, &r->mu);
        const int quit = i == (size_t)-1 || r->stop || r->bad;
        pthread_mutex_unlock(&r->mu);
        if (quit) return NULL;
        const size_t off = i * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)r->s->size ? (size_t)UP_SEG : (size_t)r->s->size - off;
        const int ok = pread_full(r->s->fd, r->pin + (i % UP_SLOTS) * (size_t)UP_SEG, len, off + (size_t)r->s->file_off) == len;
        pthread_mutex_lock(&r->mu);
        if (ok) r->filled[i % UP_SLOTS] = 1; else r->bad = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
    }
}
static void *bgzf_uploader(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    void *up = NULL;
    int ok = stream_get(0, &up) == HPGV_OK;
    const char *us = getenv("HPGV_UPLOAD_SEGMENT_MB");
    g_up_seg = us && atoi(us) >= 1 && atoi(us) <= 64 ? (size_t)atoi(us) << 20 : (size_t)UP_SEG_DEFAULT;
    const size_t pin_cap = (size_t)UP_SEG * UP_SLOTS;
    char *pin = text_buf_get(pin_cap + 1);                          /* from the runs' cache of page-locked buffers */
    up_ring_t r;
    memset(&r, 0, sizeof r);
    r.s = s; r.pin = pin; r.n_seg = ((size_t)s->size + UP_SEG - 1) / UP_SEG;
    pthread_mutex_init(&r.mu, NULL); pthread_cond_init(&r.cv, NULL);
    pthread_t th[UP_READERS_MAX];
    int n_th = 0;
    if (ok && pin) {
        int want = default_io_threads() * 3 / 4;
        if (s->is_part || s->mp) want = want / 2 > 2 ? want / 2 : 2;    /* several parts go up side by side: they share the host's threads */
        want = want < 1 ? 1 : want > UP_READERS_MAX ? UP_READERS_MAX : want;
        for (; n_th < want; n_th++) if (pthread_create(&th[n_th], NULL, up_reader, &r) != 0) break;
    }
    ok = ok && pin && n_th > 0;
    double t_wait = 0, t_copy = 0; const double t_begin = now_s();
    /* UP_INFLIGHT copies in flight: one is waited for while the others are queued or running */
    enum { UP_INFLIGHT_MAX = 8 };
    const char *uf = getenv("HPGV_UPLOAD_INFLIGHT");
    const int depth = uf && atoi(uf) >= 1 && atoi(uf) <= UP_INFLIGHT_MAX ? atoi(uf) : UP_INFLIGHT;
    void *stq[UP_INFLIGHT_MAX] = { up };
    for (int k = 1; ok && k < depth; k++) ok = stream_get(0, &stq[k]) == HPGV_OK;
    for (size_t i = 0; ok && i < r.n_seg + (size_t)depth - 1; i++) {
        double t0 = now_s();
        if (i < r.n_seg) {
            pthread_mutex_lock(&r.mu);
            while (!r.filled[i % UP_SLOTS] && !r.bad) pthread_cond_wait(&r.cv, &r.mu);
            ok = !r.bad;
            pthread_mutex_unlock(&r.mu);
            t_wait += now_s() - t0; t0 = now_s();
            if (!ok) break;
            const size_t off = i * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)s->size ? (size_t)UP_SEG : (size_t)s->size - off;
            ok = hpgv_memcpy_h2d_async(CTX, (char *)s->d_comp + off, pin + (i % UP_SLOTS) * (size_t)UP_SEG, len, stq[i % (size_t)depth]) == HPGV_OK;
            if (!ok) break;
        }
        if (i + 1 < (size_t)depth) continue;
        const size_t j = i + 1 - (size_t)depth, off = j * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)s->size ? (size_t)UP_SEG : (size_t)s->size - off;
        ok = hpgv_stream_sync(CTX, stq[j % (size_t)depth]) == HPGV_OK;
        t_copy += now_s() - t0;
        pthread_mutex_lock(&r.mu);
        r.filled[j % UP_SLOTS] = 0; r.copied = j + 1;
        pthread_cond_broadcast(&r.cv);
        pthread_mutex_unlock(&r.mu);
        if (!ok) break;
        if (j == 0 && getenv("HPGV_RUN_TRACE")) fprintf(stderr, "uploader: first segment up %.4f s after its start\n", now_s() - t_begin);
        pthread_mutex_lock(&s->g_mu);
        s->up_done = off + len;
        const int cancel = s->u_cancel;
        pthread_cond_broadcast(&s->g_cv);
        pthread_mutex_unlock(&s->g_mu);
        if (cancel) { ok = 0; break; }
    }
    for (int k = 1; k < depth; k++) if (stq[k]) { (void)hpgv_stream_sync(CTX, stq[k]); stream_put(0, stq[k]); }
    if (up) (void)hpgv_stream_sync(CTX, up);
    if (getenv("HPGV_RUN_TRACE"))
        fprintf(stderr, "uploader: %.1f MB in %.4f s: %.4f s waiting for the readers (%d), %.4f s in copies\n", s->size / 1e6, now_s() - t_begin, t_wait, n_th, t_copy);
    pthread_mutex_lock(&r.mu); r.stop = 1; pthread_cond_broadcast(&r.cv); pthread_mutex_unlock(&r.mu);
    for (int k = 0; k < n_th; k++) pthread_join(th[k], NULL);
    pthread_mutex_destroy(&r.mu); pthread_cond_destroy(&r.cv);
    if (pin) text_buf_put(pin, pin_cap + 1);
    stream_put(0, up);
    pthread_mutex_lock(&s->g_mu);
    if (!ok && !s->u_cancel) s->u_err = 1;
    if (!ok && s->up_done < (size_t)s->size) s->u_err = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    return NULL;
}
#undef UP_SEG
/* the stager waits until the file's first `hi` bytes are on the device; 0 when the uploader failed */
static int wait_uploaded(source_t *s, size_t hi) {
    pthread_mutex_lock(&s->g_mu);
    while (s->up_done < hi && !s->u_err) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const int ok = s->up_done >= hi;
    pthread_mutex_unlock(&s->g_mu);
    return ok;
}

static void *bgzf_gpu_stager(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    const size_t nb = s->g_nb;
    char *t = (char *)s->d_tab;
    int ok = 1;
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * 4 * GPU_STRETCH);
    unsigned char *tmp = (unsigned char *)malloc(65536);
    ok = ok && st && tmp;
    /* (the host's table, HPGV_BGZF_HOST_TABLE=1)  A short first stretch, which the header reader and the pipeline wait for
     * (a wave per block: 1.5 ms), then stretches that double up to 131 072 blocks, decoding side by side, up to GPU_INFLIGHT
     * launches on streams of their own.  Published in file order. */
    enum { GPU_INFLIGHT = 4, GPU_FIRST = 4096 };
    const int inflight = GPU_INFLIGHT;
    void *cs[GPU_INFLIGHT] = { s->cstream, NULL, NULL, NULL };
    for (int q = 1; ok && q < inflight; q++) ok = stream_get(0, &cs[q]) == HPGV_OK;
    size_t q_hi[GPU_INFLIGHT];                           /* the stretches in flight end at these blocks; the oldest starts at g_done */
    int qh = 0, qn = 0;
    const int dbg = getenv("HPGV_RUN_TRACE") != NULL; const double T0 = now_s();
    size_t up_hi = 0;                                    /* compressed bytes [0, up_hi) are on the device */
    for (size_t first = 0; ok && (first < nb || qn > 0);) {
        if (first < nb && qn < inflight) {               /* upload the next stretch while the earlier ones decode */
            /* a short first stretch, which the header reader waits for, then 32 768 blocks doubling up to four times that */
            const size_t stretch = first == 0 ? (nb <= 3 * (size_t)GPU_FIRST ? nb : GPU_FIRST)      /* a small file: one launch */
                                 : first < GPU_FIRST + (size_t)GPU_STRETCH ? GPU_STRETCH
                                 : first < GPU_FIRST + 3 * (size_t)GPU_STRETCH ? 2 * (size_t)GPU_STRETCH : 4 * (size_t)GPU_STRETCH;
            const size_t next = first + stretch < nb ? first + stretch : nb;
            const size_t lo = (size_t)s->g_in_off[first], hi = (size_t)s->g_in_off[next - 1] + s->g_in_len[next - 1] + 8;   /* + the last block's trailer: its CRC-32 is checked */
            (void)lo;
            if (hi > up_hi) {                             /* the uploader has been at it since the file was opened */
                ok = wait_uploaded(s, hi);
                up_hi = hi;
                if (dbg) fprintf(stderr, "stager: [%zu,%zu) is up, to byte %.1f MB, at %.4f\n", first, next, hi / 1e6, now_s() - T0);
            }
            const int q = (qh + qn) % inflight;
            ok = ok && hpgv_inflate_blocks_dev(CTX, (const uint8_t *)s->d_comp, (const uint64_t *)t + first, (const uint32_t *)(t + nb * 16) + first,
                                               (const uint64_t *)(t + nb * 8) + first, (const uint32_t *)(t + nb * 20) + first, (int)(next - first),
                                               (uint8_t *)s->d_text, (int32_t *)s->d_status + first, cs[q]) == HPGV_OK;
            if (bgzf_verify_on())                         /* the blocks' CRC-32, on the device behind the decoder */
                ok = ok && hpgv_bgzf_verify_dev(CTX, (const uint8_t *)s->d_comp, (const uint64_t *)t + first, (const uint32_t *)(t + nb * 16) + first,
                                                (const uint64_t *)(t + nb * 8) + first, (const uint32_t *)(t + nb * 20) + first, (int)(next - first),
                                                (const uint8_t *)s->d_text, (int32_t *)s->d_status + first, cs[q]) == HPGV_OK;
            q_hi[q] = next; qn++;
            if (ok && first == 0 && next < nb) {
                /* launches that run side by side finish together, so the first stretch decodes alone (the time of one
                 * block, ~40 ms) while the uploader carries on; the next stretches are launched as soon as it is done */
                first = next;
            } else {
                first = next;
                if (first < nb && qn < inflight) continue;
            }
        }
        if (ok && qn > 0) {                              /* the oldest stretch in flight: wait, check, publish */
            const size_t a = s->g_done, n = q_hi[qh] - a;
            ok = hpgv_memcpy_d2h(CTX, st, (char *)s->d_status + a * 4, n * 4, cs[qh]) == HPGV_OK;      /* synchronises that stream */
            const char *fe = getenv("HPGV_TEST_GPU_INFLATE_REFUSE_EVERY");     /* tests: exercise the host patch path */
            const size_t refuse_every = fe ? (size_t)atol(fe) : 0;
            for (size_t k = 0; ok && k < n; k++)
                if (st[k] || (refuse_every && (a + k) % refuse_every == 0)) {                             /* not taken by the device decoder: the host decodes it, the text is patched */
                    const size_t bb = a + k;
                    ok = !inflate_block(s->map + s->g_in_off[bb], s->g_in_len[bb], tmp, s->g_out_len[bb])
                      && hpgv_memcpy_h2d(CTX, (char *)s->d_text + s->g_out_off[bb], tmp, s->g_out_len[bb], cs[qh]) == HPGV_OK;
                }
            if (dbg) fprintf(stderr, "stager: decoded up to %zu at %.4f\n", q_hi[qh], now_s() - T0);
            if (ok) {
                pthread_mutex_lock(&s->g_mu);
                s->g_done = q_hi[qh];
                s->dev_ready = q_hi[qh] == nb ? s->dev_len : (size_t)s->g_out_off[q_hi[qh]];
                pthread_cond_broadcast(&s->g_cv);
                pthread_mutex_unlock(&s->g_mu);
            }
            qh = (qh + 1) % inflight; qn--;
        }
    }
    for (int q = 0; q < GPU_INFLIGHT; q++) if (cs[q]) (void)hpgv_stream_sync(CTX, cs[q]);      /* after a failure launches may still be running */
    for (int q = 1; q < GPU_INFLIGHT; q++) stream_put(0, cs[q]);
    pthread_mutex_lock(&s->g_mu);
    if (!ok) s->g_err = 1;
    s->g_finished = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    free(st); free(tmp);
    if (s->u_started) {                                  /* the compressed bytes are freed below: the uploader must be through */
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (dbg) fprintf(stderr, "stager: finished %.4f\n", now_s() - T0);
    /* only the text is needed from here on */
    if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    if (s->d_status) { (void)hpgv_dev_free(CTX, s->d_status); s->d_status = NULL; }
    if (dbg) fprintf(stderr, "stager: freed %.4f\n", now_s() - T0);
    return NULL;
}

/* The BGZF header walk is a chain of dependent cache misses (122 000 blocks: 25 ms), so a team walks the file in
 * segments: every segment but the first finds a place where three valid block headers follow one another, walks from
 * there to the first block at or past its end, and the pieces are accepted only if every walk ends exactly where
 * the next one began -- then their concatenation IS the chain from offset 0.  Anything else: the serial walk.  The
 * team reads the file with pread (some 60 bytes per block), so that no page of the mapping is touched. */
typedef struct {
    int fd; size_t size, seg; int nseg;
    size_t *start, *end, *cnt;                           /* per segment: first block, where the walk stopped, blocks */
    uint64_t **in_off; uint32_t **in_len, **out_len; int *bad;
} bgzf_walk_t;
static size_t pread_full(int fd, void *buf, size_t n, size_t pos) {
    size_t have = 0;
    while (have < n) {
        const ssize_t got = pread(fd, (char *)buf + have, n - have, (off_t)(pos + have));
        if (got <= 0) break;
        have += (size_t)got;
    }
    return have;
}
/* bgzf_block without the trailer: the first `have` bytes of a block that has `avail` bytes of file left */
static int bgzf_header(const unsigned char *p, size_t have, size_t avail, size_t *bsize, size_t *cdata_off) {
    if (have < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8), end = 12 + xlen;
    size_t off = 12, bs = 0;
    if (end > have) return 0;                            /* more extra fields than the walk reads: the serial walk takes the file */
    while (off + 4 <= end) {
        const size_t slen = (size_t)p[off + 2] | ((size_t)p[off + 3] << 8);
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2 && off + 6 <= end) bs = ((size_t)p[off + 4] | ((size_t)p[off + 5] << 8)) + 1;
        off += 4 + slen;
    }
    if (bs < end + 8 || bs > avail) return 0;
    *bsize = bs; *cdata_off = end;
    return 1;
}
static void bgzf_walk_task(void *v, int k) {
    enum { WINDOW = 4 * 65536 + 256, HEAD = 60 };
    bgzf_walk_t *w = (bgzf_walk_t *)v;
    const size_t lim = k + 1 == w->nseg ? w->size : (size_t)(k + 1) * w->seg;
    size_t pos = (size_t)k * w->seg, bs, co, is;
    w->bad[k] = 1; w->cnt[k] = 0;
    if (k > 0) {                                         /* the first place in the segment where three valid blocks follow one another */
        unsigned char *win = (unsigned char *)malloc(WINDOW);
        if (!win) return;
        const size_t base = pos, wn = pread_full(w->fd, win, WINDOW, base), slim = lim - base < wn ? lim - base : wn;
        int found = 0;
        size_t o = 0;
        while (o < slim) {
            const unsigned char *c = (const unsigned char *)memchr(win + o, 31, slim - o);
            if (!c) break;
            o = (size_t)(c - win);
            size_t q = o;
            int chain = 0;
            while (chain < 3 && q < wn && bgzf_block(win + q, wn - q, &bs, &co, &is) && is <= 65536) { q += bs; chain++; }
            if (chain == 3 || (chain > 0 && base + q == w->size)) { found = 1; break; }
            o++;
        }
        free(win);
        if (!found) {
            if (slim < lim - base) return;               /* a segment longer than the window with no block start in the window: not BGZF as we know it */
            w->start[k] = w->end[k] = lim; w->bad[k] = 0; return;      /* no block begins in this segment */
        }
        pos = base + o;
    }
    w->start[k] = pos;
    size_t cap = w->seg / 8192 + 1024, n = 0;
    uint64_t *a = (uint64_t *)malloc(cap * 8);
    uint32_t *b = (uint32_t *)malloc(cap * 4), *c = (uint32_t *)malloc(cap * 4);
    int ok = a && b && c;
    unsigned char hb[4 + HEAD];
    size_t hn = ok && pos < lim ? pread_full(w->fd, hb + 4, HEAD, pos) : 0;      /* hb + 4: this block's first bytes */
    while (ok && pos < lim) {
        if (!bgzf_header(hb + 4, hn, w->size - pos, &bs, &co)) { ok = 0; break; }
        /* one read gets this block's last four bytes (ISIZE) and the next block's first ones */
        const size_t got = pread_full(w->fd, hb, 4 + HEAD, pos + bs - 4);
        if (got < 4) { ok = 0; break; }
        is = (size_t)hb[0] | ((size_t)hb[1] << 8) | ((size_t)hb[2] << 16) | ((size_t)hb[3] << 24);
        hn = got - 4;
        if (is > 65536) { ok = 0; break; }
        if (n == cap) {
            cap *= 2;
            uint64_t *a2 = (uint64_t *)realloc(a, cap * 8);
            uint32_t *b2 = (uint32_t *)realloc(b, cap * 4), *c2 = (uint32_t *)realloc(c, cap * 4);
            if (a2) a = a2;
            if (b2) b = b2;
            if (c2) c = c2;
            if (!a2 || !b2 || !c2) { ok = 0; break; }
        }
        a[n] = pos + co; b[n] = (uint32_t)(bs - co - 8); c[n] = (uint32_t)is;
        n++; pos += bs;
    }
    w->in_off[k] = a; w->in_len[k] = b; w->out_len[k] = c; w->cnt[k] = n; w->end[k] = pos; w->bad[k] = !ok;
}
/* 0 = the tables are filled (malloc'ed, *nb_out blocks, *text_out bytes of text); non-zero = walk serially */
static int bgzf_walk_parallel(int fd, size_t size, uint64_t **in_off, uint64_t **out_off, uint32_t **in_len, uint32_t **out_len,
                              size_t *nb_out, size_t *text_out) {
    enum { NSEG = 64 };
    if (size < (size_t)NSEG * 4096) return 1;
    bgzf_walk_t w;
    size_t start[NSEG], end[NSEG], cnt[NSEG];
    uint64_t *a[NSEG]; uint32_t *b[NSEG], *c[NSEG]; int bad[NSEG];
    memset(a, 0, sizeof a); memset(b, 0, sizeof b); memset(c, 0, sizeof c);
    w.fd = fd; w.size = size; w.seg = size / NSEG; w.nseg = NSEG;
    w.start = start; w.end = end; w.cnt = cnt; w.in_off = a; w.in_len = b; w.out_len = c; w.bad = bad;
    io_pool_t tp;
    pool_init(&tp, default_io_threads());
    pool_run(&tp, bgzf_walk_task, &w, NSEG);
    pool_destroy(&tp);
    int ok = 1;
    size_t nb = 0, expect = 0;
    for (int k = 0; k < NSEG && ok; k++) {
        if (bad[k]) ok = 0;
        else if (cnt[k] == 0) { if (start[k] != end[k]) ok = 0; }          /* an empty segment: the chain passes over it */
        else { if (start[k] != expect) ok = 0; expect = end[k]; nb += cnt[k]; }
    }
    if (ok && expect != size) ok = 0;
    uint64_t *io = NULL, *oo = NULL; uint32_t *il = NULL, *ol = NULL;
    if (ok) {
        io = (uint64_t *)malloc((nb + 1) * 8); oo = (uint64_t *)malloc((nb + 1) * 8);
        il = (uint32_t *)malloc((nb + 1) * 4); ol = (uint32_t *)malloc((nb + 1) * 4);
        ok = io && oo && il && ol;
    }
    if (ok) {
        size_t n = 0, text = 0;
        for (int k = 0; k < NSEG; k++) {
            if (cnt[k] == 0) continue;
            memcpy(io + n, a[k], cnt[k] * 8); memcpy(il + n, b[k], cnt[k] * 4); memcpy(ol + n, c[k], cnt[k] * 4);
            n += cnt[k];
        }
        for (size_t i = 0; i < nb; i++) { oo[i] = text; text += ol[i]; }
        *in_off = io; *out_off = oo; *in_len = il; *out_len = ol; *nb_out = nb; *text_out = text;
    } else { free(io); free(oo); free(il); free(ol); }
    for (int k = 0; k < NSEG; k++) { free(a[k]); free(b[k]); free(c[k]); }
    return !ok;
}

/* ---- the streaming form of the device path: the block table comes from the device too ----------------------------------
 * Walking the 490 000 block headers of a 4.6 GB file on the host takes 0.08 - 0.26 s of dependent reads (beside the
 * uploader's own reads of the same file) before the first block can be decoded.  Here the stager asks the device for the
 * blocks in what has been uploaded so far (hpgv_bgzf_scan_dev finds the headers in the compressed bytes and checks that
 * they form a chain), decodes them, and goes on where the chain stands: the blocks of the first 8 MB are decoded a few
 * milliseconds after the file was opened, then stretches of 16 384 and 32 768 blocks as their bytes arrive.  The text's size is not known in advance, so the text lies in a range of device
 * addresses that is backed as the table grows (dev_text_grow); the reader learns the text's end when the stager has seen
 * the file's last block.  A stretch of the file whose headers are not the ones bgzip writes is walked on the host (through
 * the mapping).  HPGV_BGZF_HOST_TABLE=1 keeps the table on the host (the form above). */
enum { SCAN_ROWS_MAX = 131072, SCAN_SLOTS = 4 };
#define SCAN_RANGE_MAX ((size_t)2 << 30)
typedef struct {
    uint64_t *d_in_off, *d_out_off; uint32_t *d_in_len, *d_out_len; int32_t *d_status;       /* device rows of this stretch */
    uint64_t *h_in_off, *h_out_off; uint32_t *h_in_len, *h_out_len;                          /* and their host copy (for blocks the device refuses) */
    size_t n, text_end;
    void *stream;
} scan_slot_t;
typedef struct {
    scan_slot_t slot[SCAN_SLOTS];
    void *d_scratch; size_t scratch_bytes;
    size_t chain_pos, text_pos, blocks;                   /* the chain stands at this file offset; text bytes and blocks before it */
    size_t first_n;                                       /* rows already in slot 0 (found when the path was chosen) */
    size_t rows_cap;                                      /* tests (HPGV_TEST_SCAN_ROWS): no stretch longer than this */
    size_t host_rows;                                     /* blocks the next walk on the host may take (doubles while the device finds no chain) */
} scan_state_t;

/* the uploader's progress: waits until at least `want` bytes are up (or all of the file); returns how many are, 0 on failure */
static size_t wait_uploaded_some(source_t *s, size_t want) {
    if (want > (size_t)s->size) want = (size_t)s->size;
    pthread_mutex_lock(&s->g_mu);
    while (s->up_done < want && !s->u_err) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const size_t have = s->u_err ? 0 : s->up_done;
    pthread_mutex_unlock(&s->g_mu);
    return have;
}
/* rows of the blocks from file offset `pos` on, walked through the mapping (headers the device scan does not know);
 * returns the number of rows (0: not a block), *end = where the walk stands */
static size_t bgzf_host_rows(source_t *s, size_t pos, size_t hi, size_t max_rows, size_t text, scan_slot_t *q, size_t *end, size_t *text_end) {
    size_t n = 0;
    while (n < max_rows && pos < hi) {
        size_t bs, co, is;
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536 || pos + bs > hi) break;
        q->h_in_off[n] = pos + co; q->h_in_len[n] = (uint32_t)(bs - co - 8); q->h_out_off[n] = text; q->h_out_len[n] = (uint32_t)is;
        text += is; pos += bs; n++;
    }
    *end = pos; *text_end = text;
    return n;
}
static int scan_slot_rows_to_host(scan_slot_t *q) {
    return hpgv_memcpy_d2h(CTX, q->h_in_off, q->d_in_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_out_off, q->d_out_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_in_len, q->d_in_len, q->n * 4, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_out_len, q->d_out_len, q->n * 4, q->stream) == HPGV_OK;
}
static int scan_slot_rows_to_device(scan_slot_t *q) {
    return hpgv_memcpy_h2d(CTX, q->d_in_off, q->h_in_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_out_off, q->h_out_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_in_len, q->h_in_len, q->n * 4, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_out_len, q->h_out_len, q->n * 4, q->stream) == HPGV_OK;
}
/* the next stretch's rows into slot q: up to max_rows blocks from the chain's position among the bytes that are up.
 * 1 = q->n rows (0 rows: the file has ended), 0 = failure, 2 = (only with wait = 0) the bytes for that many blocks are
 * not up yet */
static int scan_next_rows(source_t *s, scan_state_t *S, scan_slot_t *q, size_t max_rows, int wait, int dbg, double T0) {
    q->n = 0; q->text_end = S->text_pos;
    const size_t avg = S->blocks ? S->chain_pos / S->blocks + 1 : 16384;
    size_t want = S->chain_pos + max_rows * avg;                      /* bytes that should hold that many blocks */
    /* the file's first stretch is whatever the first two segments hold (1 400 blocks of level 6): the header reader and
     * the pipeline wait for it, and a launch of 1 400 blocks takes as long as one of 4 096 (the time of one block) */
    if (S->blocks == 0 && want > ((size_t)8 << 20)) want = (size_t)8 << 20;
    for (;;) {
        if (S->chain_pos >= (size_t)s->size) return 1;
        if (!wait) {
            const size_t now_up = wait_uploaded_some(s, 0);
            if (now_up < (want < (size_t)s->size ? want : (size_t)s->size)) return 2;
        }
        const size_t have = wait_uploaded_some(s, want);
        if (!have) return 0;
        size_t hi = have;
        if (hi - S->chain_pos > SCAN_RANGE_MAX) hi = S->chain_pos + SCAN_RANGE_MAX;
        uint64_t res[4] = { 0, 0, 0, 0 };
        if (hpgv_bgzf_scan_dev(CTX, (const uint8_t *)s->d_comp, S->chain_pos, hi, S->text_pos, (int)max_rows, q->d_in_off, q->d_in_len,
                               q->d_out_off, q->d_out_len, S->d_scratch, S->scratch_bytes, res, q->stream) != HPGV_OK) return 0;
        if (res[0] > 0) {
            q->n = (size_t)res[0]; q->text_end = (size_t)res[2];
            if (!scan_slot_rows_to_host(q)) return 0;
            S->chain_pos = (size_t)res[1]; S->text_pos = q->text_end; S->blocks += q->n; S->host_rows = 64;
            if (dbg) fprintf(stderr, "stager: %zu blocks found up to byte %.1f MB (%.1f MB are up) at %.4f\n", q->n, S->chain_pos / 1e6, have / 1e6, now_s() - T0);
            return 1;
        }
        /* no block at the chain's position among the bytes that are up: it is not all there yet, or its header is not bgzip's */
        if (hi < (size_t)s->size && hi - S->chain_pos < ((size_t)1 << 17)) { want = hi + ((size_t)1 << 20); continue; }
        size_t end = 0, tend = 0;
        if (S->host_rows < 64) S->host_rows = 64;
        q->n = bgzf_host_rows(s, S->chain_pos, hi, S->host_rows < max_rows ? S->host_rows : max_rows, S->text_pos, q, &end, &tend);
        S->host_rows *= 2;                                            /* a few blocks, then the device again; more if it still finds none */
        if (q->n == 0) {
            if (hi < (size_t)s->size) { want = hi + ((size_t)1 << 20); continue; }      /* a block that ends beyond what is up */
            return 0;                                                 /* bytes that are no block */
        }
        if (dbg) fprintf(stderr, "stager: %zu blocks walked on the host from byte %zu at %.4f\n", q->n, S->chain_pos, now_s() - T0);
        q->text_end = tend;
        if (!scan_slot_rows_to_device(q)) return 0;
        S->chain_pos = end; S->text_pos = tend; S->blocks += q->n;
        return 1;
    }
}

/* the stretches go through a ring of SCAN_SLOTS slots: the stager finds and launches them as their bytes arrive, the
 * publisher waits for them in file order, patches what the device refused and hands the text to the reader -- neither
 * waits for the other's event */
typedef struct {
    source_t *s; scan_state_t *S;
    pthread_mutex_t mu; pthread_cond_t cv;
    size_t n_launched, n_published;                       /* stretches; slot k % SCAN_SLOTS holds stretch k */
    int launch_done, bad;
    double T0; int dbg;
} scan_ring_t;
static void *bgzf_publisher(void *v) {
    scan_ring_t *R = (scan_ring_t *)v;
    source_t *s = R->s;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    unsigned char *tmp = (unsigned char *)malloc(65536);
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * SCAN_ROWS_MAX);
    int ok = tmp && st;
    const char *fe = getenv("HPGV_TEST_GPU_INFLATE_REFUSE_EVERY");     /* tests: exercise the host patch path */
    const size_t refuse_every = fe ? (size_t)atol(fe) : 0;
    size_t done_blocks = 0;
    for (size_t k = 0; ok; k++) {
        pthread_mutex_lock(&R->mu);
        while (R->n_launched <= k && !R->launch_done && !R->bad) pthread_cond_wait(&R->cv, &R->mu);
        const int have = R->n_launched > k && !R->bad;
        pthread_mutex_unlock(&R->mu);
        if (!have) break;
        scan_slot_t *q = &R->S->slot[k % SCAN_SLOTS];
        ok = hpgv_memcpy_d2h(CTX, st, q->d_status, q->n * 4, q->stream) == HPGV_OK;           /* synchronises that stream */
        for (size_t i = 0; ok && i < q->n; i++)
            if (st[i] || (refuse_every && (done_blocks + i) % refuse_every == 0)) {             /* not taken by the device decoder: the host decodes it, the text is patched */
                ok = !inflate_block(s->map + q->h_in_off[i], q->h_in_len[i], tmp, q->h_out_len[i])
                  && hpgv_memcpy_h2d(CTX, (char *)s->d_text + q->h_out_off[i], tmp, q->h_out_len[i], q->stream) == HPGV_OK;
            }
        done_blocks += q->n;
        if (R->dbg) fprintf(stderr, "stager: decoded up to block %zu at %.4f\n", done_blocks, now_s() - R->T0);
        if (ok) {
            pthread_mutex_lock(&s->g_mu);
            s->g_done = done_blocks;
            s->dev_ready = q->text_end;
            pthread_cond_broadcast(&s->g_cv);
            pthread_mutex_unlock(&s->g_mu);
        }
        pthread_mutex_lock(&R->mu);
        R->n_published = k + 1;
        if (!ok) R->bad = 1;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->mu);
    }
    if (!ok) { pthread_mutex_lock(&R->mu); R->bad = 1; pthread_cond_broadcast(&R->cv); pthread_mutex_unlock(&R->mu); }
    free(tmp); free(st);
    return NULL;
}
static void *bgzf_gpu_stream_stager(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    scan_state_t *S = (scan_state_t *)s->blk;                        /* (handed over in the field the host path uses for its block list) */
    s->blk = NULL;
    scan_ring_t R;
    memset(&R, 0, sizeof R);
    R.s = s; R.S = S; R.dbg = getenv("HPGV_RUN_TRACE") != NULL; R.T0 = now_s();
    const int dbg = R.dbg; const double T0 = R.T0;
    pthread_mutex_init(&R.mu, NULL); pthread_cond_init(&R.cv, NULL);
    pthread_t pub;
    int ok = pthread_create(&pub, NULL, bgzf_publisher, &R) == 0;
    const int have_pub = ok;
    size_t launched = 0;
    for (size_t k = 0; ok; k++) {
        pthread_mutex_lock(&R.mu);                                   /* a free slot */
        while (k >= R.n_published + SCAN_SLOTS && !R.bad) pthread_cond_wait(&R.cv, &R.mu);
        ok = !R.bad;
        pthread_mutex_unlock(&R.mu);
        if (!ok) break;
        scan_slot_t *q = &S->slot[k % SCAN_SLOTS];
        /* a short first stretch, which the header reader and the pipeline wait for, then 16 384 and 32 768 blocks at a time:
         * a stretch is launched when its bytes are up, and the decoder (23 GB/s of bgzip's level-6 bytes) is not much slower than
         * the bus -- behind stretches that double (up to 131 072 blocks) the device waited for the next one's bytes */
        size_t rows = launched == 0 ? 4096 : launched < 4096 + 16384 ? 16384 : 32768;
        if (S->rows_cap && rows > S->rows_cap) rows = S->rows_cap;
        if (k == 0 && S->first_n) q->n = S->first_n;                 /* found when the path was chosen */
        else ok = scan_next_rows(s, S, q, rows, 1, dbg, T0) == 1;
        if (!ok || q->n == 0) break;                                 /* (no rows: the file has ended) */
        if (k == 1) {                                                /* launches side by side finish together: the first stretch, which the */
            pthread_mutex_lock(&R.mu);                               /* header reader waits for, decodes alone (1.5 ms instead of 6) */
            while (R.n_published < 1 && !R.bad) pthread_cond_wait(&R.cv, &R.mu);
            ok = !R.bad;
            pthread_mutex_unlock(&R.mu);
            if (!ok) break;
        }
        ok = dev_text_grow(s->d_text, q->text_end + 16 + (q->text_end >> 4), &s->d_text_cap)      /* some room ahead: growing waits for the kernels that run */
          || dev_text_grow(s->d_text, q->text_end + 16, &s->d_text_cap);
        ok = ok && hpgv_inflate_blocks_dev(CTX, (const uint8_t *)s->d_comp, q->d_in_off, q->d_in_len, q->d_out_off, q->d_out_len,
                                           (int)q->n, (uint8_t *)s->d_text, q->d_status, q->stream) == HPGV_OK;
        if (bgzf_verify_on())                                        /* the blocks' CRC-32, on the device behind the decoder */
            ok = ok && hpgv_bgzf_verify_dev(CTX, (const uint8_t *)s->d_comp, q->d_in_off, q->d_in_len, q->d_out_off, q->d_out_len,
                                            (int)q->n, (const uint8_t *)s->d_text, q->d_status, q->stream) == HPGV_OK;
        if (!ok) break;
        if (dbg && k == 0) fprintf(stderr, "stager: first stretch launched at %.4f\n", now_s() - T0);
        launched += q->n;
        pthread_mutex_lock(&R.mu);
        R.n_launched = k + 1;
        pthread_cond_broadcast(&R.cv);
        pthread_mutex_unlock(&R.mu);
        if (S->chain_pos >= (size_t)s->size) break;
    }
    pthread_mutex_lock(&R.mu);
    R.launch_done = 1;
    if (!ok) R.bad = 1;
    pthread_cond_broadcast(&R.cv);
    pthread_mutex_unlock(&R.mu);
    if (have_pub) pthread_join(pub, NULL);
    ok = ok && !R.bad;
    pthread_mutex_destroy(&R.mu); pthread_cond_destroy(&R.cv);
    for (int q = 0; q < SCAN_SLOTS; q++) if (S->slot[q].stream) (void)hpgv_stream_sync(CTX, S->slot[q].stream);      /* after a failure launches may still be running */
    pthread_mutex_lock(&s->g_mu);
    if (!ok) s->g_err = 1;
    else { s->dev_len = S->text_pos; s->dev_len_known = 1; s->g_nb = S->blocks; }
    s->g_finished = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    if (s->u_started) {                                  /* the compressed bytes are freed below: the uploader must be through */
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (dbg) fprintf(stderr, "stager: finished (%zu blocks, %.1f MB of text) at %.4f\n", S->blocks, S->text_pos / 1e6, now_s() - T0);
    for (int q = 1; q < SCAN_SLOTS; q++) stream_put(s->c_low, S->slot[q].stream);
    for (int q = 0; q < SCAN_SLOTS; q++) free(S->slot[q].h_in_off);
    free(S);
    if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }      /* only the text is needed from here on */
    if (s->d_scan) { (void)hpgv_dev_free(CTX, s->d_scan); s->d_scan = NULL; }
    return NULL;
}

/* 0 = the streaming stager has the file; 1 = not taken (the caller goes on with the host's table; nothing is left behind) */
static int bgzf_stream_stage(source_t *s) {
    const int dbg = getenv("HPGV_RUN_TRACE") != NULL; const double T0 = now_s();
    scan_state_t *S = (scan_state_t *)calloc(1, sizeof *S);
    if (!S) return 1;
    /* the decoder's streams have the lowest priority: the batches' kernels go first whenever a compute unit has room */
    const int low = !getenv("HPGV_NO_LOW_PRIORITY");
    s->c_low = low;
    int ok = stream_get(0, &s->rstream) == HPGV_OK && stream_get(low, &s->cstream) == HPGV_OK;
    const size_t slot_bytes = (size_t)SCAN_ROWS_MAX * 28;
    S->scratch_bytes = hpgv_bgzf_scan_scratch_bytes(SCAN_RANGE_MAX + 16, SCAN_ROWS_MAX);
    if (ok) ok = hpgv_dev_alloc(CTX, slot_bytes * SCAN_SLOTS + S->scratch_bytes + 256, &s->d_scan) == HPGV_OK;
    for (int k = 0; ok && k < SCAN_SLOTS; k++) {
        scan_slot_t *q = &S->slot[k];
        char *d = (char *)s->d_scan + (size_t)k * slot_bytes;
        q->d_in_off = (uint64_t *)d; q->d_out_off = (uint64_t *)(d + (size_t)SCAN_ROWS_MAX * 8);
        q->d_in_len = (uint32_t *)(d + (size_t)SCAN_ROWS_MAX * 16); q->d_out_len = (uint32_t *)(d + (size_t)SCAN_ROWS_MAX * 20);
        q->d_status = (int32_t *)(d + (size_t)SCAN_ROWS_MAX * 24);
        char *h = (char *)malloc((size_t)SCAN_ROWS_MAX * 24);
        ok = h != NULL;
        q->h_in_off = (uint64_t *)h; q->h_out_off = (uint64_t *)(h + (size_t)SCAN_ROWS_MAX * 8);
        q->h_in_len = (uint32_t *)(h + (size_t)SCAN_ROWS_MAX * 16); q->h_out_len = (uint32_t *)(h + (size_t)SCAN_ROWS_MAX * 20);
        if (k == 0) q->stream = s->cstream; else ok = ok && stream_get(low, &q->stream) == HPGV_OK;
    }
    if (ok) S->d_scratch = (char *)s->d_scan + slot_bytes * SCAN_SLOTS;
    if (dbg) fprintf(stderr, "stage: streams and tables at %.4f\n", now_s() - T0);
    /* the first blocks, from the file's first megabytes: is this a file the device can chain, and how much text is it? */
    const char *tr = getenv("HPGV_TEST_SCAN_ROWS");
    S->rows_cap = tr && atol(tr) > 0 ? (size_t)atol(tr) : 0;
    if (ok) ok = scan_next_rows(s, S, &S->slot[0], S->rows_cap && S->rows_cap < 4096 ? S->rows_cap : 4096, 1, dbg, T0) == 1 && S->slot[0].n > 0;
    size_t est = 0;
    if (ok) {
        /* committed at once, with room (growing later waits for the kernels that are running) */
        est = (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size * 1.10) + ((size_t)128 << 20);
        if (S->chain_pos >= (size_t)s->size) est = S->text_pos + 16;
        if (est > ((size_t)48 << 30)) ok = 0;                        /* as with the host's table: such a text stays on the host path, */
        if (S->chain_pos >= (size_t)s->size && S->blocks < 256 && !s->is_part && !s->mp) ok = 0;      /* and a small file is as quick there */
    }
    const char *tp = getenv("HPGV_TEST_TEXT_ESTIMATE_PERCENT");    /* tests: a text that outgrows what was committed for it */
    if (ok && tp && atoi(tp) > 0 && S->chain_pos < (size_t)s->size) {
        est = (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size) / 100 * (size_t)atoi(tp);
        if (est < S->text_pos + 16) est = S->text_pos + 16;
        pthread_mutex_lock(&g_text_mu);
        void *old = g_dev_text; const int old_kind = g_dev_text_kind;
        g_dev_text = NULL; g_dev_text_cap = 0;
        pthread_mutex_unlock(&g_text_mu);
        dev_text_free(old, old_kind);
    }
    if (ok) {
        s->text_est = S->chain_pos >= (size_t)s->size ? S->text_pos : (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size);
        s->d_text = dev_text_get(est, &s->d_text_cap, &s->d_text_kind);
        ok = s->d_text != NULL && ((s->d_text_kind == DEV_TEXT_GROWS && !getenv("HPGV_NO_GROWING_TEXT")) || S->chain_pos >= (size_t)s->size);
        if (!ok && s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
    }
    if (dbg) fprintf(stderr, "stage: first %zu blocks found, text estimate %.1f MB, %s at %.4f\n", S->slot[0].n, est / 1e6, ok ? "streaming" : "not taken", now_s() - T0);
    if (ok) {
        S->first_n = S->slot[0].n;
        s->dev_len = 0; s->dev_len_known = 0; s->dev_pos = 0; s->dev_ready = 0; s->g_done = 0; s->g_err = 0; s->g_finished = 0; s->g_nb = 0;
        s->blk = (size_t *)S;
        ok = pthread_create(&s->g_thread, NULL, bgzf_gpu_stream_stager, s) == 0;
        if (ok) s->g_started = 1; else s->blk = NULL;
    }
    if (!ok) {
        for (int k = 1; k < SCAN_SLOTS; k++) stream_put(low, S->slot[k].stream);
        for (int k = 0; k < SCAN_SLOTS; k++) free(S->slot[k].h_in_off);
        free(S);
        if (s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
        if (s->d_scan) { (void)hpgv_dev_free(CTX, s->d_scan); s->d_scan = NULL; }
        stream_put(0, s->rstream); s->rstream = NULL;
        stream_put(low, s->cstream); s->cstream = NULL; s->c_low = 0;
        return 1;
    }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

/* ---- a bgzip file on SEVERAL devices (a group context: HPGV_DEVICES, hpgv_host_init_devices; --num-threads / the devices
 * option of shared_options.c:60-61 has no such notion: the reference reads the file with one thread).  BGZF blocks are
 * independent, so the file is cut at block starts into one contiguous PART per member: every part goes up ITS device's own
 * link (the upload is what a run of the device path waits for), is decoded and kept there, and its windows are tokenized and
 * scanned there.  Each part is staged by the streaming stager exactly as a whole file is (a source_t whose map / size /
 * file_off describe the part).  The reader walks the parts in file order; a part's text ends inside a line as a rule: the
 * line's head (the end of part k) and tail (the start of part k + 1) come back to the host, and the joined line goes through
 * the ordinary host-text entry as a batch of one line between the two parts' windows (read_lines_dev). ---- */
static int bgzf_stream_stage(source_t *s);
static void *bgzf_uploader(void *v);
/* one part (or, when the parts are given up, nothing): 0 = the streaming stager has it */
static int part_stream_stage(source_t *s) {
    const ctx_saved_t saved = SRC_CTX(s);
    int taken = 0;
    s->gpu_tried = 1;
    pthread_mutex_init(&s->g_mu, NULL); pthread_cond_init(&s->g_cv, NULL);
    s->g_sync = 1; s->up_done = 0; s->u_cancel = 0; s->u_err = 0;
    if (hpgv_dev_alloc(CTX, (size_t)s->size + 16, &s->d_comp) == HPGV_OK) {
        if (pthread_create(&s->u_thread, NULL, bgzf_uploader, s) == 0) s->u_started = 1;
        else { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    }
    if (s->u_started && bgzf_stream_stage(s) == 0) taken = 1;
    if (!taken) {                                                    /* nothing is left behind */
        if (s->u_started) {
            pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
            pthread_join(s->u_thread, NULL); s->u_started = 0;
        }
        if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
        pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0;
    }
    ctx_back(saved);
    return taken ? 0 : 1;
}
/* the first block start at or after `from` from which four blocks chain (a header's magic inside compressed data does not) */
static size_t bgzf_find_block_start(const source_t *s, size_t from) {
    const size_t size = (size_t)s->size, lim = from + ((size_t)1 << 20) < size ? from + ((size_t)1 << 20) : size;
    for (size_t pos = from; pos + 28 <= lim; pos++) {
        if (s->map[pos] != 31 || s->map[pos + 1] != 139) continue;
        size_t q = pos;
        int n = 0;
        while (n < 4 && q < size) {
            size_t bs, co, is;
            if (!bgzf_block(s->map + q, size - q, &bs, &co, &is) || is > 65536) break;
            q += bs; n++;
        }
        if (n == 4 || (n > 0 && q == size)) return pos;
    }
    return 0;
}
/* 0 = the file is staged in parts (s is part 0); 1 = not taken, s is as it was */
static int bgzf_parts_stage(source_t *s) {
    const int G = g_ctx ? hpgv_group_size(g_ctx) : 1;
    if (G < 2 || getenv("HPGV_BGZF_ONE_DEVICE") || getenv("HPGV_NO_DEVICE_WINDOWS") || getenv("HPGV_BGZF_HOST_TABLE") ||
        getenv("HPGV_SERIAL_BGZF_WALK") || getenv("HPGV_NO_GROWING_TEXT")) return 1;
    const char *pm = getenv("HPGV_BGZF_PART_MIN_KB");               /* tests: parts of small files */
    const size_t part_min = pm && atol(pm) > 0 ? (size_t)atol(pm) << 10 : (size_t)64 << 20;
    int n = G < MEMBERS_MAX ? G : MEMBERS_MAX;
    if ((size_t)s->size / part_min < (size_t)n) n = (int)((size_t)s->size / part_min);
    if (n < 2) return 1;
    size_t b[MEMBERS_MAX + 1];
    b[0] = 0; b[n] = (size_t)s->size;
    for (int k = 1; k < n; k++) {
        b[k] = bgzf_find_block_start(s, (size_t)s->size / (size_t)n * (size_t)k);
        if (b[k] == 0 || b[k] <= b[k - 1]) return 1;
    }
    src_parts_t *mp = (src_parts_t *)calloc(1, sizeof *mp);
    if (!mp) return 1;
    mp->n = n; mp->p[0] = s; mp->whole_size = s->size;
    const int dbg = getenv("HPGV_RUN_TRACE") != NULL;
    int ok = 1;
    for (int k = 1; ok && k < n; k++) {                              /* the later parts first: if one of them is not taken, part 0 is still the whole file */
        source_t *p = (source_t *)calloc(1, sizeof *p);
        if (!p) { ok = 0; break; }
        p->kind = SRC_BGZF; p->fd = dup(s->fd); p->is_part = 1;
        p->map_base = s->map_base; p->map_len = s->map_len;
        p->map = s->map + b[k]; p->size = (off_t)(b[k + 1] - b[k]); p->file_off = (off_t)b[k];
        p->ctx = hpgv_group_member(g_ctx, k); p->member = k;
        mp->p[k] = p;
        ok = p->fd >= 0 && p->ctx && part_stream_stage(p) == 0;
        if (ok) p->map_pos = (size_t)p->size;
    }
    if (ok) {
        s->ctx = hpgv_group_member(g_ctx, 0); s->member = 0; s->mp = mp; s->size = (off_t)b[1];
        ok = part_stream_stage(s) == 0;
        if (!ok) { s->ctx = NULL; s->mp = NULL; s->size = mp->whole_size; s->gpu_tried = 0; }
    }
    if (!ok) {
        for (int k = 1; k < n; k++) if (mp->p[k]) { source_close(mp->p[k]); free(mp->p[k]); }
        free(mp);
        if (dbg) fprintf(stderr, "stage: the file is not taken in parts\n");
        return 1;
    }
    if (dbg) { fprintf(stderr, "stage: %d parts, one per device, cut at bytes", n); for (int k = 1; k < n; k++) fprintf(stderr, " %zu", b[k]); fprintf(stderr, "\n"); }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

static int bgzf_gpu_stage(source_t *s) {
    s->gpu_tried = 1;
    if (getenv("HPGV_NO_GPU_INFLATE") || !g_ctx || s->map_pos != 0) return 1;
    if (!s->is_part && !s->mp && (size_t)s->size >= ((size_t)64 << 10) && bgzf_parts_stage(s) == 0) return 0;
    const int dbg = getenv("HPGV_RUN_TRACE") != NULL; double T0 = now_s();
    size_t nb = 0, cap = 1 << 16, text = 0;
    uint64_t *in_off = NULL, *out_off = NULL;
    uint32_t *in_len = NULL, *out_len = NULL;
    int ok = 1;
    if ((size_t)s->size < ((size_t)64 << 10)) return 1;             /* a tiny file is as quick on the host (the block count decides below) */
    /* the compressed bytes start going up NOW, beside the walk of the block headers (0.08 s for the 490 000 blocks of a
     * 4.6 GB file): the decoder needs the table, the bus does not */
    pthread_mutex_init(&s->g_mu, NULL); pthread_cond_init(&s->g_cv, NULL);
    s->g_sync = 1; s->up_done = 0; s->u_cancel = 0; s->u_err = 0;
    if (hpgv_dev_alloc(CTX, (size_t)s->size + 16, &s->d_comp) == HPGV_OK) {
        if (dbg) fprintf(stderr, "stage: room for the compressed file at %.4f\n", now_s() - T0);
        if (pthread_create(&s->u_thread, NULL, bgzf_uploader, s) == 0) s->u_started = 1;
        else { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    }
    if (!s->u_started) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0; return 1; }
    if (!getenv("HPGV_BGZF_HOST_TABLE") && !getenv("HPGV_SERIAL_BGZF_WALK") && bgzf_stream_stage(s) == 0) return 0;
    if (getenv("HPGV_SERIAL_BGZF_WALK") || bgzf_walk_parallel(s->fd, (size_t)s->size, &in_off, &out_off, &in_len, &out_len, &nb, &text)) {
        nb = 0; text = 0;
        if (dbg) fprintf(stderr, "stage: serial walk\n");
        in_off = (uint64_t *)malloc(cap * 8); out_off = (uint64_t *)malloc(cap * 8);
        in_len = (uint32_t *)malloc(cap * 4); out_len = (uint32_t *)malloc(cap * 4);
        ok = in_off && out_off && in_len && out_len;
        size_t pos = 0;
        while (ok && pos < (size_t)s->size) {
            size_t bs, co, is;
            if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536) { ok = 0; break; }
            if (nb == cap) {
                cap *= 2;
                uint64_t *a = (uint64_t *)realloc(in_off, cap * 8), *b = (uint64_t *)realloc(out_off, cap * 8);
                uint32_t *c = (uint32_t *)realloc(in_len, cap * 4), *d = (uint32_t *)realloc(out_len, cap * 4);
                if (a) in_off = a;
                if (b) out_off = b;
                if (c) in_len = c;
                if (d) out_len = d;
                if (!a || !b || !c || !d) { ok = 0; break; }
            }
            in_off[nb] = pos + co; in_len[nb] = (uint32_t)(bs - co - 8); out_off[nb] = text; out_len[nb] = (uint32_t)is;
            text += is; pos += bs; nb++;
        }
    }
    if (dbg) fprintf(stderr, "stage: walk %.4f\n", now_s() - T0);
    if (ok && (nb < 256 || nb > 0x7FFFFFFFu || text > ((size_t)48 << 30))) ok = 0;     /* a small file is as quick on the host */
    if (ok) { s->c_low = 0; ok = stream_get(0, &s->rstream) == HPGV_OK && stream_get(0, &s->cstream) == HPGV_OK; }
    if (ok) ok = hpgv_dev_alloc(CTX, nb * 24 + 64, &s->d_tab) == HPGV_OK;
    if (ok) { s->text_est = text; s->d_text = dev_text_get(text + 16, &s->d_text_cap, &s->d_text_kind); ok = s->d_text != NULL; }
    if (ok) ok = hpgv_dev_alloc(CTX, nb * 4 + 16, &s->d_status) == HPGV_OK;
    if (dbg) fprintf(stderr, "stage: alloc %.4f\n", now_s() - T0);
    if (ok) {
        char *t = (char *)s->d_tab;
        ok = hpgv_memcpy_h2d(CTX, t, in_off, nb * 8, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 8, out_off, nb * 8, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 16, in_len, nb * 4, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 20, out_len, nb * 4, s->cstream) == HPGV_OK;
    }
    if (dbg) fprintf(stderr, "stage: tab %.4f\n", now_s() - T0);
    if (ok) {
        s->g_in_off = in_off; s->g_out_off = out_off; s->g_in_len = in_len; s->g_out_len = out_len; s->g_nb = nb;
        s->dev_len = text; s->dev_len_known = 1; s->dev_pos = 0; s->dev_ready = 0; s->g_done = 0; s->g_err = 0; s->g_finished = 0;
        ok = pthread_create(&s->g_thread, NULL, bgzf_gpu_stager, s) == 0;
        if (ok) s->g_started = 1;
    }
    if (!ok) {
        if (s->u_started) {                              /* not a file for the device path after all: call the uploader back */
            pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
            pthread_join(s->u_thread, NULL); s->u_started = 0;
        }
        free(in_off); free(out_off); free(in_len); free(out_len);
        s->g_in_off = s->g_out_off = NULL; s->g_in_len = s->g_out_len = NULL;
        if (s->g_sync) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0; }
            if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
        if (s->d_tab) { (void)hpgv_dev_free(CTX, s->d_tab); s->d_tab = NULL; }
        if (s->d_status) { (void)hpgv_dev_free(CTX, s->d_status); s->d_status = NULL; }
        if (s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
        stream_put(0, s->rstream); s->rstream = NULL;
        stream_put(0, s->cstream); s->cstream = NULL;
        return 1;
    }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

/* appends up to cap bytes of (decompressed) data to buf; 0 = end of data, (size_t)-1 = error.  RAW and GZIP
 * fill the room unless the data ends; BGZF hands out the whole blocks that fit (inflated in parallel, straight
 * into buf), or -- when not even the next block fits -- the part of it that does, so callers loop. */
static size_t source_read(source_t *s, char *buf, size_t cap) {
    if (s->kind == SRC_RAW) {
        if (s->pos >= s->size) return 0;
        size_t want = (size_t)(s->size - s->pos) < cap ? (size_t)(s->size - s->pos) : cap;
        s->job_buf = buf; s->job_want = want; s->job_bad = 0;
        pool_run(s->pool, source_task_pread, s, (int)((want + PREAD_SEG - 1) / PREAD_SEG));
        const int bad = s->job_bad;
        if (bad) return (size_t)-1;
        s->pos += (off_t)want;
        return want;
    }
    if (s->kind == SRC_GZIP) {
        size_t n = 0;
        while (n < cap) {
            unsigned chunk = (cap - n) > (1u << 30) ? (1u << 30) : (unsigned)(cap - n);
            int got = gzread(s->gz, buf + n, chunk);
            if (got < 0) return (size_t)-1;
            if (got == 0) break;
            n += (size_t)got;
        }
        return n;
    }
    /* BGZF */
    if (cap == 0) return 0;
    if (!s->gpu_tried) (void)bgzf_gpu_stage(s);
    if (s->d_text) {                                                 /* the text is on the device: copy the next stretch out */
        pthread_mutex_lock(&s->g_mu);                                /* until the stager has decoded that far (or knows where the text ends) */
        for (;;) {
            const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
            const size_t want = limit - s->dev_pos < cap ? limit : s->dev_pos + cap;
            if (s->g_err || s->dev_ready >= want || s->g_finished) break;
            pthread_cond_wait(&s->g_cv, &s->g_mu);
        }
        const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
        const size_t end = limit - s->dev_pos < cap ? limit : s->dev_pos + cap;
        const int bad = s->g_err || s->dev_ready < end;
        pthread_mutex_unlock(&s->g_mu);
        if (bad) return (size_t)-1;
        const size_t n = end - s->dev_pos;
        if (n == 0) return 0;
        if (hpgv_memcpy_d2h(CTX, buf, (const char *)s->d_text + s->dev_pos, n, s->rstream) != HPGV_OK) return (size_t)-1;
        s->dev_pos += n;
        return n;
    }
    if (s->pend_pos < s->pend_len) {                                  /* rest of the block set aside last time */
        size_t n = s->pend_len - s->pend_pos < cap ? s->pend_len - s->pend_pos : cap;
        memcpy(buf, s->pend + s->pend_pos, n);
        s->pend_pos += n;
        return n;
    }
    size_t total = 0;
    int nb = 0;
    if (!s->blk && !(s->blk = (size_t *)malloc(sizeof(size_t) * 4 * MAXB))) return (size_t)-1;
    size_t *const b_in = s->blk, *const b_clen = s->blk + MAXB, *const b_out = s->blk + 2 * MAXB, *const b_isize = s->blk + 3 * MAXB;
    size_t pos = s->map_pos;
    while (pos < (size_t)s->size && nb < MAXB) {
        size_t bs, co, is;
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536) return (size_t)-1;
        if (total + is > cap) break;
        b_in[nb] = pos + co; b_clen[nb] = bs - co - 8; b_out[nb] = total; b_isize[nb] = is;
        total += is; pos += bs; nb++;
    }
    if (total == 0) {
        if (pos >= (size_t)s->size) { s->map_pos = pos; return 0; }   /* only empty blocks (the EOF marker) were left */
        size_t bs, co, is;                                            /* the next block is larger than the room */
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is)) return (size_t)-1;
        if (!s->pend && !(s->pend = (unsigned char *)malloc(65536))) return (size_t)-1;
        if (inflate_block(s->map + pos + co, bs - co - 8, s->pend, is)) return (size_t)-1;
        s->map_pos = pos + bs;
        s->pend_len = is; s->pend_pos = cap;
        memcpy(buf, s->pend, cap);
        return cap;
    }
    s->job_buf = buf; s->job_want = (size_t)nb; s->job_bad = 0;
    pool_run(s->pool, source_task_inflate, s, (nb + INFLATE_GROUP - 1) / INFLATE_GROUP);
    const int bad = s->job_bad;
    if (bad) return (size_t)-1;
    s->map_pos = pos;
    return total;
}

/* batch reader on top of a source: whole lines, about batch_bytes per batch; the unfinished last line is
 * carried over to the next batch */
typedef struct {
    source_t src;
    char *carry; size_t carry_len, carry_cap;
    int eof;
    const char *last_dev;                               /* device copy of the batch read_lines just returned (BGZF on the GPU), or NULL */
    hpgv_ctx *last_ctx;                                 /* ... and the member context of that device when the file is staged in parts */
    int devwin;                                         /* batches are windows of the device text; nothing but their cut points is read back */
    char *tailbuf; size_t tailcap;
} line_reader_t;

/* bgzip decoded on the device: the next batch is a window of the device text that ends with a line.  Only the stretch
 * around the window's end comes back to the host, to find that line end; the engine fills the batch's host buffer with
 * the line heads (hpgv_text_alias).  Returns the window's bytes (0 at the end, (size_t)-1 on error), its device address
 * in r->last_dev and the context of its device in r->last_ctx.  `last`: the text's end is the file's end (the last line may
 * lack its newline); otherwise -- a part that another part follows -- every window ends with a newline and what is left
 * behind the text's last newline stays unread (the seam line's head). */
static size_t read_window(line_reader_t *r, source_t *s, size_t cap, int last) {
    const ctx_saved_t saved = SRC_CTX(s);
    const size_t start = s->dev_pos;
    size_t ret = (size_t)-1;
    r->last_dev = (const char *)s->d_text + start;
    r->last_ctx = s->ctx;
    pthread_mutex_lock(&s->g_mu);                                    /* until the stager has decoded that far (or knows where the text ends) */
    for (;;) {
        const size_t lim = s->dev_len_known ? s->dev_len : (size_t)-1;
        const size_t want = lim - start <= cap ? lim : start + cap;
        if (s->g_err || s->dev_ready >= want || s->g_finished) break;
        pthread_cond_wait(&s->g_cv, &s->g_mu);
    }
    const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
    size_t end = limit - start <= cap ? limit : start + cap;
    const int bad = s->g_err || s->dev_ready < end;
    pthread_mutex_unlock(&s->g_mu);
    if (bad) goto out;
    if (start >= limit) { ret = 0; goto out; }
    if (end < limit || !last) {                                      /* cut at the last newline before `end` */
        size_t look = 1u << 18;
        for (;;) {
            if (look > end - start) look = end - start;
            if (r->tailcap < look) { free(r->tailbuf); r->tailbuf = (char *)malloc(look); r->tailcap = r->tailbuf ? look : 0; }
            if (!r->tailbuf) goto out;
            if (hpgv_memcpy_d2h(CTX, r->tailbuf, (const char *)s->d_text + end - look, look, s->rstream) != HPGV_OK) goto out;
            const char *nl = (const char *)memrchr(r->tailbuf, '\n', look);
            if (nl) { end = end - look + (size_t)(nl - r->tailbuf) + 1; break; }
            if (look == end - start) {
                if (end == limit && !last) { ret = 0; goto out; }    /* the part's rest holds no line end: all of it is the seam line's head */
                goto out;                                            /* a line longer than the batch */
            }
            look *= 4;
        }
    }
    s->dev_pos = end;
    ret = end - start;
out:
    ctx_back(saved);
    return ret;
}
/* a part's text length, known when its stager has seen its last block; (size_t)-1 on failure */
static size_t part_text_len(source_t *s) {
    pthread_mutex_lock(&s->g_mu);
    while (!s->dev_len_known && !s->g_err && !s->g_finished) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const size_t n = (s->g_err || !s->dev_len_known) ? (size_t)-1 : s->dev_len;
    pthread_mutex_unlock(&s->g_mu);
    return n;
}
/* bytes [from, to) of a part's text to the host, once they are decoded */
static int part_text_fetch(source_t *s, size_t from, size_t to, char *dst) {
    if (to <= from) return 1;
    pthread_mutex_lock(&s->g_mu);
    while (s->dev_ready < to && !s->g_err && !s->g_finished) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const int ok = !s->g_err && s->dev_ready >= to;
    pthread_mutex_unlock(&s->g_mu);
    if (!ok) return 0;
    const ctx_saved_t saved = SRC_CTX(s);
    const int rc = hpgv_memcpy_d2h(CTX, dst, (const char *)s->d_text + from, to - from, s->rstream);
    ctx_back(saved);
    return rc == HPGV_OK;
}
static int seam_room(src_parts_t *mp, size_t more) {
    if (mp->seam_len + more <= mp->seam_cap) return 1;
    const size_t cap = (mp->seam_len + more) * 2 + 4096;
    char *q = (char *)realloc(mp->seam, cap);
    if (!q) return 0;
    mp->seam = q; mp->seam_cap = cap;
    return 1;
}
/* the next batch of a file on the device: a window of its text (r->last_dev set: the engine tokenizes it in place and writes
 * the lines' heads into the batch's host buffer), or -- a file staged in parts -- the line that straddles two parts, copied
 * into buf as ordinary host text (r->last_dev NULL). */
static size_t read_lines_dev(line_reader_t *r, char *buf, size_t bufcap, size_t cap) {
    src_parts_t *mp = r->src.mp;
    if (!mp) return read_window(r, &r->src, cap, 1);
    for (;;) {
        if (mp->seam_len) {                                          /* the seam line in front of part `cur` */
            if (mp->seam_len > bufcap) return (size_t)-1;
            memcpy(buf, mp->seam, mp->seam_len);
            const size_t n = mp->seam_len;
            mp->seam_len = 0;
            r->last_dev = NULL; r->last_ctx = NULL;
            return n;
        }
        source_t *s = mp->p[mp->cur];
        const int last = mp->cur == mp->n - 1;
        const size_t n = read_window(r, s, cap, last);
        if (n != 0 || last) return n;
        /* part `cur` is through: what lies behind its last newline is the head of a line that goes on in the next part(s) */
        const size_t len = part_text_len(s);
        if (len == (size_t)-1) return (size_t)-1;
        const size_t head = len - s->dev_pos;
        if (!seam_room(mp, head) || !part_text_fetch(s, s->dev_pos, len, mp->seam)) return (size_t)-1;
        mp->seam_len = head;
        s->dev_pos = len;
        while (++mp->cur < mp->n) {
            source_t *t = mp->p[mp->cur];
            if (head == 0 && mp->seam_len == 0) break;               /* the part ended with a newline: the next one starts a line */
            /* the tail: the next part's text up to its first newline, looked for in growing stretches */
            size_t have = 0, look = 1u << 16, nl_at = (size_t)-1;
            for (;;) {
                size_t tlen = (size_t)-1;
                pthread_mutex_lock(&t->g_mu);
                while (t->dev_ready < have + look && !t->dev_len_known && !t->g_err && !t->g_finished) pthread_cond_wait(&t->g_cv, &t->g_mu);
                if (t->dev_len_known) tlen = t->dev_len;
                const int bad = t->g_err || (!t->dev_len_known && t->dev_ready < have + look);
                pthread_mutex_unlock(&t->g_mu);
                if (bad) return (size_t)-1;
                size_t to = have + look;
                if (tlen != (size_t)-1 && to > tlen) to = tlen;
                if (!seam_room(mp, to - have) || !part_text_fetch(t, have, to, mp->seam + mp->seam_len)) return (size_t)-1;
                const char *nl = (const char *)memchr(mp->seam + mp->seam_len, '\n', to - have);
                if (nl) { nl_at = have + (size_t)(nl - (mp->seam + mp->seam_len)); mp->seam_len += nl_at - have + 1; break; }
                mp->seam_len += to - have;
                have = to;
                if (tlen != (size_t)-1 && have >= tlen) break;       /* the whole part is one piece of this line: on to the next part */
                look *= 4;
            }
            if (nl_at != (size_t)-1) { t->dev_pos = nl_at + 1; break; }
            t->dev_pos = have;
        }
        if (mp->cur >= mp->n) mp->cur = mp->n - 1;                   /* the line ran to the file's end: hand it out, then the last part reports the end */
    }
}

/* fills buf (capacity cap) with whole lines; returns the byte count, 0 at the end, (size_t)-1 when a
 * single line does not fit or the source fails */
static size_t read_lines(line_reader_t *r, char *buf, size_t cap) {
    size_t n = 0;
    /* the carry is the stretch of the source right before its read position: with the text on the device the batch is
     * the device bytes from (position - carry) on */
    if (!r->src.gpu_tried && r->src.kind == SRC_BGZF) (void)bgzf_gpu_stage(&r->src);
    r->last_dev = r->src.d_text ? (const char *)r->src.d_text + (r->src.dev_pos - r->carry_len) : NULL;
    if (r->carry_len) {                                 /* may exceed cap: what followed the header in its read buffer */
        n = r->carry_len < cap ? r->carry_len : cap;
        memcpy(buf, r->carry, n);
        r->carry_len -= n;
        if (r->carry_len) {
            memmove(r->carry, r->carry + n, r->carry_len);
            size_t end = n;
            while (end > 0 && buf[end - 1] != '\n') end--;
            if (end == 0) return (size_t)-1;
            const size_t tail = n - end;                /* put the cut line back in front of the rest */
            if (r->carry_len + tail > r->carry_cap) return (size_t)-1;     /* cannot happen: n bytes were just taken out */
            memmove(r->carry + tail, r->carry, r->carry_len);
            memcpy(r->carry, buf + end, tail);
            r->carry_len += tail;
            return end;
        }
    }
    while (!r->eof && n < cap) {
        size_t got = source_read(&r->src, buf + n, cap - n);
        if (got == (size_t)-1) return (size_t)-1;
        if (got == 0) { r->eof = 1; break; }
        n += got;
    }
    if (n == 0) return 0;
    if (r->eof) return n;                               /* last batch: may end without a newline */
    size_t end = n;
    while (end > 0 && buf[end - 1] != '\n') end--;
    if (end == 0) return (size_t)-1;
    size_t tail = n - end;
    if (tail > r->carry_cap) { free(r->carry); r->carry = (char *)malloc(tail); r->carry_cap = r->carry ? tail : 0; }
    if (tail && !r->carry) return (size_t)-1;
    memcpy(r->carry, buf + end, tail);
    r->carry_len = tail;
    return end;
}

/* VCF header: skips the '##' lines and takes the sample names from the '#CHROM' line; what follows that line
 * in the bytes already read becomes the reader's carry.  Returns the number of samples, -1 without a
 * '#CHROM' line; *hdr_out owns the text the names point into. */
static int vcf_header_read(line_reader_t *rd, char **hdr_out, char ***names_out, size_t *chrom_off) {
    char *hdr = NULL;
    int n_samples = -1;
    char **names = NULL;
    size_t cap = 4u << 20, have = 0;
    hdr = (char *)malloc(cap + 1);
    int done = 0;
    while (hdr && !done) {
        const int starved = cap == have;             /* header longer than the buffer: grow first */
        size_t got = starved ? 0 : source_read(&rd->src, hdr + have, cap - have);
        if (got == (size_t)-1) break;
        have += got;
        hdr[have] = 0;
        char *p = hdr, *chrom = NULL;
        size_t data_start = 0;
        while (*p == '#') {
            char *eol = (char *)memchr(p, '\n', have - (size_t)(p - hdr));
            if (!eol) { p = NULL; break; }
            if (!strncmp(p, "#CHROM", 6)) { chrom = p; *eol = 0; data_start = (size_t)(eol + 1 - hdr); break; }
            p = eol + 1;
        }
        if (chrom) {
            if (chrom_off) *chrom_off = (size_t)(chrom - hdr);
            size_t len = strlen(chrom);
            while (len > 0 && chrom[len - 1] == '\r') chrom[--len] = 0;
            int tabs = 0;
            for (size_t i = 0; i < len; i++) if (chrom[i] == '\t') tabs++;
            n_samples = tabs >= 9 ? tabs - 8 : 0;
            names = (char **)malloc(sizeof(char *) * (size_t)(n_samples + 1));
            int k = 0, col = 0;
            for (char *q = chrom; names && *q; q++)
                if (*q == '\t') { *q = 0; col++; if (col >= 9 && k < n_samples) names[k++] = q + 1; }
            size_t rest = have - data_start;
            if (rest) {
                rd->carry = (char *)malloc(rest);
                rd->carry_cap = rd->carry ? rest : 0;
                if (rd->carry) { memcpy(rd->carry, hdr + data_start, rest); rd->carry_len = rest; } else n_samples = -1;
            }
            if (!names) n_samples = -1;
            done = 1;
        } else if (p != NULL || (!starved && got == 0)) {
            break;                                   /* a data line came first, or the file ended */
        } else if (starved) {
            cap *= 2;
            char *nh = (char *)realloc(hdr, cap + 1);
            if (!nh) break;
            hdr = nh;
        }
    }
    *hdr_out = hdr; *names_out = names;
    return n_samples;
}

typedef struct {
    char *text; size_t text_cap;                         /* page-locked, taken from the cache when the batch is first filled */
    const char *dev_text;                                /* the same bytes on the device (BGZF decoded there), or NULL */
    hpgv_ctx *dev_ctx;                                   /* the member context of that device (a file staged in parts), or NULL */
    size_t bytes; int max_lines, n_lines;
    uint64_t *line_off; uint32_t *field_off; int32_t *status;
    int32_t *ints; double *dbl;                          /* 4 (assoc) or 2 (tdt) int arrays, 3 double arrays */
    uint8_t *rows; size_t rows_cap; int row_width;       /* vcf2epi: one dataset row per line */
    int stats;                                           /* aggregate / stats: the counters of hpgv_stats_text */
    int32_t *c8, *merr, *midx, *mtab, *smiss, *cerr; double *hw;
    int n_multi, multi_cap, n_smiss, n_cerr;
    int n_groups; int32_t *gc8; double *ghw;              /* stats: per-phenotype counters, [g * max_lines + v] */
} run_batch_t;

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

static int run_batch_alloc(run_batch_t *b, size_t cap_bytes, int n_samples, int row_width, int stats, int n_trios, int n_groups) {
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
static int run_batch_reserve(run_batch_t *b, int lines) {
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
static void run_batch_free(run_batch_t *b) {
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
static int record_passes(const run_batch_t *b, int i) {
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
typedef struct { int na, miss_al, miss_gt, ac[15], gc[225]; } vcounts_t;

static void record_counts(const run_batch_t *b, int i, const char *alt, int la, vcounts_t *v) {
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

static int g_aggregate_overwrite = 0;

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
typedef struct { char *p; size_t len, cap; int disorder; } out_buf_t;
typedef struct { const run_batch_t *b; out_buf_t *bufs; int kind, n, parts, bad; } fmt_job_t;

/* Is the result file in `sort -k1,1h -k2,2n` order as it is written?  (It is for a position-sorted VCF, and reading two
 * million lines back to find that out took 0.09 of a 0.21 s run.)  Neighbouring records with the same CHROM bytes and a
 * growing POS are in order by the definition of the two keys; every other pair -- a new chromosome, equal positions, the
 * seams between the formatting tasks and between batches -- goes through the sort's own comparison of the two lines. */
typedef struct { char *last; size_t cap; int have, disorder; } order_track_t;
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
static void order_track_keep(order_track_t *o, const char *line, size_t len) {              /* ... remembered as the last one */
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

static double g_write_split[2];                         /* of the last run's write stage: formatting, writing (HPGV_RUN_TRACE) */

/* the formatted lines of one batch are written by a thread of their own while the next batch is formatted (a run over 2M
 * short records spent 0.02 s of its 0.07 s write stage in fwrite): one set of buffers is being written, the other filled */
typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv; pthread_t th;
    FILE *fd; out_buf_t *bufs; int parts;              /* the set handed over (parts > 0), taken when the thread starts on it */
    int busy, stop, bad, started;
} file_writer_t;
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
static int file_writer_start(file_writer_t *w, FILE *fd) {
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
static int file_writer_stop(file_writer_t *w) {          /* everything handed over is written when this returns */
    if (!w->started) return 0;
    const int bad = file_writer_idle(w);
    pthread_mutex_lock(&w->mu); w->stop = 1; pthread_cond_broadcast(&w->cv); pthread_mutex_unlock(&w->mu);
    pthread_join(w->th, NULL);
    pthread_mutex_destroy(&w->mu); pthread_cond_destroy(&w->cv);
    w->started = 0;
    return bad;
}

/* fw (may be NULL: written here) takes the formatted set; the caller alternates between two sets of n_bufs buffers */
static int write_batch(FILE *fd, int kind, const run_batch_t *b, out_buf_t *bufs, int n_bufs, io_pool_t *pool, order_track_t *ord, file_writer_t *fw) {
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

/* ---- the runners' pipeline: reader -> engine threads -> writer, batches in rotation ------------------ */
/* batches in rotation and engine threads: two engine threads per device (one batch's bus copies beside the other's
 * kernels) and three more batches than engines (reader ahead, writer behind); one device: 5 batches, 2 engines */
enum { RUN_ENGINES_MAX = 16, RUN_NB_MAX = RUN_ENGINES_MAX + 3, RUN_FMT_BUFS = 64 };
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
        if (b->dev_text) (void)hpgv_text_alias(b->dev_ctx ? b->dev_ctx : g_ctx, b->text, b->dev_text);       /* tokenize the device copy in place (on the device that holds it): no H2D of the text */
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
        if (rd.src.d_text && !getenv("HPGV_NO_DEVICE_WINDOWS")) P->n_engines = rd.src.mp && 2 * rd.src.mp->n > 4 ? 2 * rd.src.mp->n : 4;
        const char *et = getenv("HPGV_ENGINE_THREADS");              /* diagnosis: engine threads (batches in flight on the devices) */
        if (et && atoi(et) > 0) P->n_engines = atoi(et);
        if (P->n_engines > RUN_ENGINES_MAX) P->n_engines = RUN_ENGINES_MAX;
        P->nb = P->n_engines + 3;
    }
    /* windows of a text that is on the device are not copied anywhere, so they need not be as small as the caller's batches:
     * about 64 of them per file, 256 MB at most, amortise what a batch costs whatever its size (three waits for the
     * device and 50 us of short kernels beside 100 us per 64 MB of tokenizing and scanning) */
    if (P && rd.src.d_text && !getenv("HPGV_NO_DEVICE_WINDOWS") && !getenv("HPGV_NO_LARGE_WINDOWS")) {
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
        if (rd.src.d_text && !getenv("HPGV_NO_DEVICE_WINDOWS")) {      /* bgzip decoded on the device: windows of the device text from the first data line on */
            rd.src.dev_pos -= rd.carry_len; rd.carry_len = 0; rd.devwin = 1;
        }
        const int n_fmt = io_threads < RUN_FMT_BUFS / 2 ? io_threads : RUN_FMT_BUFS / 2;      /* two sets of buffers: one is written while the other is filled */
        file_writer_t fw;
        memset(&fw, 0, sizeof fw);
        /* (the vcf2epi rows are written out of the batch itself, and the stats tool's group files by this thread) */
        const int use_fw = kind != 4 && !getenv("HPGV_NO_WRITER_THREAD") && file_writer_start(&fw, out);
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
        if (!rc && kind < 4 && (ord.disorder || !ord.have || getenv("HPGV_ALWAYS_SORT")) && hpgv_host_sort_output_file(out_path))      /* assoc_runner.c:255-261: only a warning there */
            fprintf(stderr, "WARN: results could not be sorted by chromosome and position\n");
        else if (!rc && kind < 4 && getenv("HPGV_RUN_TRACE") && !(ord.disorder || !ord.have || getenv("HPGV_ALWAYS_SORT")))
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
    if (getenv("HPGV_RUN_TRACE"))
        fprintf(stderr, "hpgv run: %ld records, %.0f batches, %d io threads: read %.3f s, engine %.3f s (%d threads), write %.3f s (stages overlap), sort %.3f s, total %.3f s\n",
                written, g_run_times[5], io_threads, g_run_times[0], g_run_times[1], n_engines_used, g_run_times[2], t_sort, g_run_times[4]);
    const double t_done = now_s();
    source_close(&rd.src); free(rd.carry); free(rd.tailbuf); free(hdr); free(names); ped_table_free(&ped);
    if (n_variants_out) *n_variants_out = written;
    pthread_rwlock_unlock(&g_cohort_lock);
    if (getenv("HPGV_RUN_TRACE"))
        fprintf(stderr, "hpgv run: before the pipeline: PED and open %.4f s, VCF header %.4f s, cohort and buffers %.4f s; after it: %.4f s; of the write stage: formatting %.4f s, writing %.4f s\n",
                t_opened - t_enter, t_header - t_opened, t_start - t_header, now_s() - t_done, g_write_split[0], g_write_split[1]);
    return rc;
}

/* the runners' reader on its own: copies `in_path` (plain, gzip or BGZF) to `out_path` in whole-line batches
 * of at most batch_bytes; what the runners feed to the engine, batch by batch */
int hpgv_host_copy_lines(const char *in_path, const char *out_path, size_t batch_bytes, int skip_vcf_header, long *n_batches) {
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
