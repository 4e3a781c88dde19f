/* host_engine.c -- the process-wide engine context behind the adapters and runners, and the buffers kept between calls.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* engine binding                                                             */
/* ------------------------------------------------------------------------ */

hpgv_ctx *g_ctx = NULL;
__thread hpgv_ctx *t_ctx = NULL;                        /* the member context a staging thread works on (hpgv_host_internal.h: CTX) */
__thread int t_member = 0;
/* what the current device-side cohort descriptions were built from */
struct host_assoc_key g_assoc_key;
struct host_tdt_key g_tdt_key;
struct host_stats_key g_stats_key;
struct host_ped_key g_ped_key;
struct host_group_key g_group_key;
struct host_lf_key g_lf_key;
static int g_device = 0;
static pthread_mutex_t g_init_mu = PTHREAD_MUTEX_INITIALIZER;
pthread_rwlock_t g_cohort_lock = PTHREAD_RWLOCK_INITIALIZER;
char g_err[512];


host_env_t g_env;
void host_env_read(void) {
    host_env_t e;
    memset(&e, 0, sizeof e);
#define ENV_NUM(name, field, lo, hi) do { const char *v_ = getenv(name); if (v_ && *v_) { long x_ = atol(v_); e.field = x_ < (lo) ? (lo) : x_ > (hi) ? (hi) : x_; } } while (0)
#define ENV_FLAG(name, field) do { e.field = getenv(name) != NULL; } while (0)       /* set at all (as before: "HPGV_X=0" also switches it on) */
    e.bgzf_verify = 1;
    e.devices_set = getenv("HPGV_DEVICES") != NULL;
    ENV_NUM("HPGV_IO_THREADS", io_threads, 0, 64);
    ENV_NUM("HPGV_STAGE_THREADS", stage_threads, 0, 1024);
    ENV_NUM("HPGV_ENGINE_THREADS", engine_threads, 0, 64);
    ENV_FLAG("HPGV_RUN_TRACE", run_trace);
    ENV_NUM("HPGV_BGZF_VERIFY", bgzf_verify, 0, 1);
    ENV_FLAG("HPGV_ZLIB_INFLATE", zlib_inflate);
    ENV_FLAG("HPGV_NO_GPU_INFLATE", no_gpu_inflate);
    ENV_FLAG("HPGV_NO_DEVICE_WINDOWS", no_device_windows);
    ENV_FLAG("HPGV_NO_LARGE_WINDOWS", no_large_windows);
    ENV_FLAG("HPGV_BGZF_ONE_DEVICE", bgzf_one_device);
    ENV_FLAG("HPGV_BGZF_HOST_TABLE", bgzf_host_table);
    ENV_FLAG("HPGV_SERIAL_BGZF_WALK", serial_bgzf_walk);
    ENV_FLAG("HPGV_NO_GROWING_TEXT", no_growing_text);
    ENV_FLAG("HPGV_NO_LOW_PRIORITY", no_low_priority);
    { const char *v_ = getenv("HPGV_DECODE_TILES"); e.no_decode_tiles = v_ && *v_ && atol(v_) == 0; }
    ENV_FLAG("HPGV_NO_NUMA_BIND", no_numa_bind);
    ENV_FLAG("HPGV_NO_WRITER_THREAD", no_writer_thread);
    ENV_FLAG("HPGV_ALWAYS_SORT", always_sort);
    ENV_NUM("HPGV_UPLOAD_SEGMENT_MB", upload_segment_mb, 0, 1 << 20);
    ENV_NUM("HPGV_UPLOAD_INFLIGHT", upload_inflight, 0, 1 << 20);
    ENV_NUM("HPGV_TEST_GPU_INFLATE_REFUSE_EVERY", test_refuse_every, 0, 1L << 40);
    ENV_NUM("HPGV_TEST_SCAN_ROWS", test_scan_rows, 0, 1L << 40);
    ENV_NUM("HPGV_TEST_TEXT_ESTIMATE_PERCENT", test_text_estimate_percent, 0, 1000);
    ENV_NUM("HPGV_BGZF_PART_MIN_KB", bgzf_part_min_kb, 0, 1L << 40);
#undef ENV_NUM
#undef ENV_FLAG
    g_env = e;
}

const char *hpgv_host_last_error(void) { return g_err; }

int host_fail(const char *what, int rc) {
    snprintf(g_err, sizeof g_err, "%s: hpgv status %d: %s", what, rc,
             g_ctx ? hpgv_last_error(g_ctx) : hpgv_last_error(NULL));
    return rc;
}

/* the engine's devices: hpgv_host_init(d) binds device d; hpgv_host_init_devices a list (a group context: batches are
 * dealt to the devices, hpgv.h hpgv_create_multi); with neither, the first call of an adapter or runner reads the
 * environment variable HPGV_DEVICES -- "0,1,2,3" or "all" -- and falls back to device 0 */
int hpgv_host_init_devices(const int *device_ids, int n_devices) {
    host_env_read();
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
    host_env_read();
    const char *e = getenv("HPGV_DEVICES");                 /* (the list itself: parsed here, once, when the engine is bound) */
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
char *text_buf_get(size_t cap) {
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
void text_buf_put(char *p, size_t cap) {
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
uint8_t *stage_get(size_t bytes, int *slot) {
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
void stage_put(uint8_t *p, int slot) {
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
#define DEV_TEXT_RESERVE ((size_t)56 << 30)
static void *g_dev_text_m[MEMBERS_MAX]; static size_t g_dev_text_cap_m[MEMBERS_MAX]; static int g_dev_text_kind_m[MEMBERS_MAX];
#define g_dev_text g_dev_text_m[t_member]
#define g_dev_text_cap g_dev_text_cap_m[t_member]
#define g_dev_text_kind g_dev_text_kind_m[t_member]
void dev_text_free(void *p, int kind) {
    if (!p) return;
    if (kind == DEV_TEXT_GROWS) (void)hpgv_dev_release(CTX, p); else (void)hpgv_dev_free(CTX, p);
}
/* a buffer with `bytes` usable bytes (*cap: how many it has); *kind = DEV_TEXT_GROWS when dev_text_grow can extend it */
void *dev_text_get(size_t bytes, size_t *cap, int *kind) {
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
    if (bytes <= DEV_TEXT_RESERVE && !g_env.no_growing_text && hpgv_dev_reserve(CTX, DEV_TEXT_RESERVE, &p) == HPGV_OK) {
        if (hpgv_dev_commit(CTX, p, bytes) == HPGV_OK) { *cap = bytes; *kind = DEV_TEXT_GROWS; return p; }
        (void)hpgv_dev_release(CTX, p);
        p = NULL;
    }
    if (hpgv_dev_alloc(CTX, bytes, &p) != HPGV_OK) return NULL;
    *cap = bytes; *kind = DEV_TEXT_FIXED;
    return p;
}
/* lets the buffer kept for this member go (a test makes the next text outgrow what would be committed for it) */
void dev_text_drop_cached(void) {
    pthread_mutex_lock(&g_text_mu);
    void *old = g_dev_text; const int old_kind = g_dev_text_kind;
    g_dev_text = NULL; g_dev_text_cap = 0;
    pthread_mutex_unlock(&g_text_mu);
    dev_text_free(old, old_kind);
}
int dev_text_grow(void *p, size_t bytes, size_t *cap) {
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
void dev_text_put(void *p, size_t bytes, int kind) {
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

/* the tile records that go with such a text (32 bytes per 2 KiB of it), kept between runs the same way */
static void *g_dev_tiles_m[MEMBERS_MAX]; static size_t g_dev_tiles_cap_m[MEMBERS_MAX];
void *dev_tiles_get(size_t bytes, size_t *cap) {
    void *p = NULL;
    pthread_mutex_lock(&g_text_mu);
    if (g_dev_tiles_m[t_member] && g_dev_tiles_cap_m[t_member] >= bytes) {
        p = g_dev_tiles_m[t_member]; *cap = g_dev_tiles_cap_m[t_member];
        g_dev_tiles_m[t_member] = NULL; g_dev_tiles_cap_m[t_member] = 0;
    }
    pthread_mutex_unlock(&g_text_mu);
    if (p) return p;
    if (hpgv_dev_alloc(CTX, bytes, &p) != HPGV_OK) return NULL;
    *cap = bytes;
    return p;
}
void dev_tiles_put(void *p, size_t cap) {
    if (!p) return;
    void *old = NULL;
    pthread_mutex_lock(&g_text_mu);
    if (g_dev_tiles_cap_m[t_member] < cap) { old = g_dev_tiles_m[t_member]; g_dev_tiles_m[t_member] = p; g_dev_tiles_cap_m[t_member] = cap; p = NULL; }
    pthread_mutex_unlock(&g_text_mu);
    if (old) (void)hpgv_dev_free(CTX, old);
    if (p) (void)hpgv_dev_free(CTX, p);
}

/* streams of the bgzip device path, kept between runs: creating the seven a run uses took 18 ms of a 0.14 s run (they are
 * idle when they come back) */
enum { STREAM_CACHE_N = 16 };
static void *g_stream_cache_m[MEMBERS_MAX][2][STREAM_CACHE_N];        /* per member; [0] normal priority, [1] lowest */
#define g_stream_cache g_stream_cache_m[t_member]
int stream_get(int low, void **out) {
    pthread_mutex_lock(&g_text_mu);
    for (int i = 0; i < STREAM_CACHE_N; i++)
        if (g_stream_cache[low][i]) { *out = g_stream_cache[low][i]; g_stream_cache[low][i] = NULL; pthread_mutex_unlock(&g_text_mu); return HPGV_OK; }
    pthread_mutex_unlock(&g_text_mu);
    return low ? hpgv_stream_create_low(CTX, out) : hpgv_stream_create(CTX, out);
}
void stream_put(int low, void *st) {
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
        pthread_mutex_lock(&g_text_mu);
        void *tl = g_dev_tiles_m[t_member];
        g_dev_tiles_m[t_member] = NULL; g_dev_tiles_cap_m[t_member] = 0;
        pthread_mutex_unlock(&g_text_mu);
        if (tl) (void)hpgv_dev_free(CTX, tl);
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

int ensure_engine(void) { return g_ctx ? HPGV_OK : init_from_environment(); }

char *dupn(const char *s, int n) { return strndup(s ? s : "", (size_t)(n > 0 ? n : 0)); }

int thread_id(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
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
