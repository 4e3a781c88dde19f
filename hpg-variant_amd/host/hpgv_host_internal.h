/*
 * hpgv_host_internal.h -- what the translation units of libhpgv_host.so share: the engine binding and its caches
 * (host_engine.c), the thread teams (host_pool.c), the byte sources and readers of the file runners (host_source.c,
 * host_inflate.c, host_bgzf.c, host_reader.c), the result formatting (host_format.c, host_output.c).  Nothing here is part
 * of the library's interface (include/hpgv_host.h is): every symbol below has hidden visibility.
 *
 *   host_containers.c  the hpg-libs stand-in containers and records (array_list, list_t, result structs)
 *   host_stage.c       sample strings -> one byte per genotype (hpgv_host_stage_records)
 *   host_engine.c      the process-wide engine context, page-locked / device buffer caches, shutdown
 *   host_adapters.c    assoc_test, tdt_test, get_variants_stats, get_sample_stats
 *   host_output.c      the reference's writers (assoc_runner.c:292-342, tdt_runner.c:286-304) and the in-process sort
 *   host_epistasis.c   k-folds, the epistasis run and its report
 *   host_pool.c        sleeping thread teams, NUMA placement
 *   host_source.c      PED table, byte sources (plain / gzip / bgzip)
 *   host_inflate.c     raw DEFLATE on the host
 *   host_bgzf.c        bgzip files staged and decoded on the device(s)
 *   host_reader.c      whole-line batches, the VCF header
 *   host_format.c      result lines of a batch, the writer thread
 *   host_runner.c      the file runners' pipeline (hpgv_run_*)
 */
#ifndef HPGV_HOST_INTERNAL_H
#define HPGV_HOST_INTERNAL_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
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

#pragma GCC visibility push(hidden)

extern hpgv_ctx *g_ctx;

/* The bgzip device path may run in PARTS, one per member of a group context (bgzf_parts_stage): the threads that stage,
 * decode and hand out a part work on that member's device.  They say so once -- ctx_use(member context, member index) --
 * and everything below them that allocates, copies or launches uses CTX, and the per-member caches (streams, the decoded
 * text) slot t_member.  Threads that never say so work on g_ctx, which for a group is its first member: the single-device
 * path is unchanged. */
enum { MEMBERS_MAX = 16 };
extern __thread hpgv_ctx *t_ctx;
extern __thread int t_member;
#define CTX (t_ctx ? t_ctx : g_ctx)
typedef struct { hpgv_ctx *ctx; int member; } ctx_saved_t;
static inline ctx_saved_t ctx_use(hpgv_ctx *ctx, int member) { ctx_saved_t o = { t_ctx, t_member }; t_ctx = ctx; t_member = ctx ? member : 0; return o; }
static inline void ctx_back(ctx_saved_t o) { t_ctx = o.ctx; t_member = o.member; }

typedef void (*pool_fn)(void *arg, int task);
typedef struct {
    pthread_t *th; int n_threads;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_done;
    pool_fn fn; void *arg; int n_tasks, next, running, gen, stop;
} io_pool_t;

enum { DEV_TEXT_FIXED = 0, DEV_TEXT_GROWS = 1 };

typedef struct { list_item_t *first, *last; size_t n; } item_chain_t;
static inline void chain_add(item_chain_t *c, list_item_t *it) {
    if (!it) return;
    it->next_p = NULL;
    if (c->last) c->last->next_p = it; else c->first = it;
    c->last = it; c->n++;
}

typedef struct {
    int n;
    char **fid, **iid, **pat, **mat, **phe;             /* phe: the PHENO column as written (the stats tool's variable) */
    int *sex, *pheno;
    char *blob;
} ped_table_t;

enum { SRC_RAW = 0, SRC_BGZF = 1, SRC_GZIP = 2 };

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
    void *d_tiles; size_t n_tiles, d_tiles_cap;                    /* the tokenizer's tile records of the decoded text, left by the CRC check (hpgv_bgzf_verify_tiles_dev) */
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

enum { PREAD_SEG = 4 << 20, INFLATE_GROUP = 8, MAXB = 1 << 16 };

typedef struct {
    source_t src;
    char *carry; size_t carry_len, carry_cap;
    int eof;
    const char *last_dev;                               /* device copy of the batch read_lines just returned (BGZF on the GPU), or NULL */
    hpgv_ctx *last_ctx;                                 /* ... and the member context of that device when the file is staged in parts */
    const char *last_base; const void *last_tiles; size_t last_n_tiles;      /* ... where that device text begins, and its tile records */
    int devwin;                                         /* batches are windows of the device text; nothing but their cut points is read back */
    char *tailbuf; size_t tailcap;
} line_reader_t;

typedef struct {
    char *text; size_t text_cap;                         /* page-locked, taken from the cache when the batch is first filled */
    const char *dev_text;                                /* the same bytes on the device (BGZF decoded there), or NULL */
    hpgv_ctx *dev_ctx;                                   /* the member context of that device (a file staged in parts), or NULL */
    const char *dev_base; const void *dev_tiles; size_t dev_n_tiles;      /* where the decoded text begins on the device, and its tile records (or NULL) */
    size_t bytes; int max_lines, n_lines;
    uint64_t *line_off; uint32_t *field_off; int32_t *status;
    int32_t *ints; double *dbl;                          /* 4 (assoc) or 2 (tdt) int arrays, 3 double arrays */
    uint8_t *rows; size_t rows_cap; int row_width;       /* vcf2epi: one dataset row per line */
    int stats;                                           /* aggregate / stats: the counters of hpgv_stats_text */
    int32_t *c8, *merr, *midx, *mtab, *smiss, *cerr; double *hw;
    int n_multi, multi_cap, n_smiss, n_cerr;
    int n_groups; int32_t *gc8; double *ghw;              /* stats: per-phenotype counters, [g * max_lines + v] */
} run_batch_t;

typedef struct { int na, miss_al, miss_gt, ac[15], gc[225]; } vcounts_t;

typedef struct { char *p; size_t len, cap; int disorder; } out_buf_t;

typedef struct { char *last; size_t cap; int have, disorder; } order_track_t;

typedef struct {
    pthread_mutex_t mu; pthread_cond_t cv; pthread_t th;
    FILE *fd; out_buf_t *bufs; int parts;              /* the set handed over (parts > 0), taken when the thread starts on it */
    int busy, stop, bad, started;
} file_writer_t;

typedef struct { const char *line; int neg1, order1; long double v1; int neg2; long double v2; } sort_key_t;

enum { RUN_ENGINES_MAX = 16, RUN_NB_MAX = RUN_ENGINES_MAX + 3, RUN_FMT_BUFS = 64 };

/* what the current device-side cohort descriptions were built from */
struct host_assoc_key {
    const void *samples; int num_samples; uint64_t cond_hash; int set;
    uint8_t *cond;                                       /* the installed condition vector itself: a hash hit is confirmed against it */
};
extern struct host_assoc_key g_assoc_key;
struct host_tdt_key {
    int num_families; int num_columns; uint64_t hash; int set;
    int32_t *csr; uint8_t *csex; size_t n_children;       /* the installed description itself (father | mother | offsets | child columns): a hash hit is confirmed against it */
};
extern struct host_tdt_key g_tdt_key;
struct host_stats_key { int num_samples; int set; };
extern struct host_stats_key g_stats_key;
struct host_ped_key { int num_samples; int n_trios; uint64_t hash; int set; };
extern struct host_ped_key g_ped_key;
struct host_group_key { int num_samples; int n_groups; uint64_t hash; int set; };
extern struct host_group_key g_group_key;
struct host_lf_key { const void *table; int n; };
extern struct host_lf_key g_lf_key;

/* The environment of libhpgv_host.so (include/hpgv_host.h "Environment"): read by host_env_read() when the engine is bound and
 * at the start of every file run -- never on a per-batch or per-block path -- into this one structure. */
typedef struct {
    long devices_set;                /* HPGV_DEVICES given (its text is parsed by the engine binding) */
    long io_threads;                 /* HPGV_IO_THREADS: reader / formatter team (0: by the cores) */
    long stage_threads;              /* HPGV_STAGE_THREADS: team of a lone adapter caller's staging (0: 8) */
    long engine_threads;             /* HPGV_ENGINE_THREADS: batches in flight on the devices (0: by the input) */
    long run_trace;                  /* HPGV_RUN_TRACE=1: stage times of a run on stderr */
    long bgzf_verify;                /* HPGV_BGZF_VERIFY=0: no CRC-32 check of decoded blocks (default 1) */
    long zlib_inflate;               /* HPGV_ZLIB_INFLATE=1: host blocks through zlib instead of the built-in decoder */
    long no_gpu_inflate;             /* HPGV_NO_GPU_INFLATE=1: bgzip decoded on the host */
    long no_device_windows;          /* HPGV_NO_DEVICE_WINDOWS=1: device-decoded text copied back whole */
    long no_large_windows;           /* HPGV_NO_LARGE_WINDOWS=1: windows of the caller's batch size */
    long bgzf_one_device;            /* HPGV_BGZF_ONE_DEVICE=1: a group decodes the file on member 0 only */
    long bgzf_host_table;            /* HPGV_BGZF_HOST_TABLE=1: the block table walked by the host's team */
    long serial_bgzf_walk;           /* HPGV_SERIAL_BGZF_WALK=1: ... by one thread */
    long no_growing_text;            /* HPGV_NO_GROWING_TEXT=1: a fixed device buffer for the decoded text */
    long no_low_priority;            /* HPGV_NO_LOW_PRIORITY=1: the decoder's streams at normal priority */
    long no_decode_tiles;            /* HPGV_DECODE_TILES=0: the CRC check leaves no tile records (windows tokenized with the counting sweep) */
    long no_numa_bind;               /* HPGV_NO_NUMA_BIND=1: a run's threads stay where the scheduler puts them */
    long no_writer_thread;           /* HPGV_NO_WRITER_THREAD=1: result lines written by the formatting thread */
    long always_sort;                /* HPGV_ALWAYS_SORT=1: the output is sorted even when it came out in order */
    long upload_segment_mb;          /* HPGV_UPLOAD_SEGMENT_MB (1..64, default 4), upload_inflight: HPGV_UPLOAD_INFLIGHT (1..8, default 2) */
    long upload_inflight;
    /* tests only */
    long test_refuse_every;          /* HPGV_TEST_GPU_INFLATE_REFUSE_EVERY: every n-th block handed back to the host decoder */
    long test_scan_rows;             /* HPGV_TEST_SCAN_ROWS: blocks per stretch of the streaming stager */
    long test_text_estimate_percent; /* HPGV_TEST_TEXT_ESTIMATE_PERCENT: the first commitment as a share of the estimate */
    long bgzf_part_min_kb;           /* HPGV_BGZF_PART_MIN_KB: smallest part of a file staged in parts (default 65536) */
} host_env_t;
extern host_env_t g_env;
void host_env_read(void);

/* ---- functions and data the units share ---- */
/* host_containers.c */
void list_insert_chain(list_item_t *first, list_item_t *last, size_t n, list_t *list);
/* host_pool.c */
void pool_init(io_pool_t *p, int n_threads);
void pool_spread(io_pool_t *p);
void pool_destroy(io_pool_t *p);
void pool_run(io_pool_t *p, pool_fn fn, void *arg, int n_tasks);
int default_io_threads(void);
int numa_bind_to_device(cpu_set_t *saved);
void numa_unbind(const cpu_set_t *saved, int bound);
/* host_stage.c */
void stage_team_release(void);
/* host_engine.c */
int host_fail(const char *what, int rc);
char *text_buf_get(size_t cap);
void text_buf_put(char *p, size_t cap);
uint8_t *stage_get(size_t bytes, int *slot);
void stage_put(uint8_t *p, int slot);
void dev_text_free(void *p, int kind);
void *dev_text_get(size_t bytes, size_t *cap, int *kind);
int dev_text_grow(void *p, size_t bytes, size_t *cap);
void dev_text_drop_cached(void);
void dev_text_put(void *p, size_t bytes, int kind);
void *dev_tiles_get(size_t bytes, size_t *cap);
void dev_tiles_put(void *p, size_t cap);
int stream_get(int low, void **out);
void stream_put(int low, void *st);
int ensure_engine(void);
char *dupn(const char *s, int n);
int thread_id(void);
/* host_output.c */
void make_key(const char *line, sort_key_t *k);
int cmp_keys(const void *pa, const void *pb);
/* host_source.c */
void ped_table_free(ped_table_t *p);
int ped_table_read(const char *path, ped_table_t *ped);
double now_s(void);
int bgzf_block(const unsigned char *p, size_t avail, size_t *bsize, size_t *cdata_off, size_t *isize);
int source_open(source_t *s, const char *path);
void source_close(source_t *s);
/* host_inflate.c */
int fast_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len);
/* host_bgzf.c */
int inflate_block(const unsigned char *in, size_t clen, unsigned char *out, size_t isize);
void source_task_pread(void *v, int k);
void source_task_inflate(void *v, int g);
int bgzf_gpu_stage(source_t *s);
/* host_reader.c */
size_t read_lines_dev(line_reader_t *r, char *buf, size_t bufcap, size_t cap);
size_t read_lines(line_reader_t *r, char *buf, size_t cap);
int vcf_header_read(line_reader_t *rd, char **hdr_out, char ***names_out, size_t *chrom_off);
/* host_format.c */
int run_batch_alloc(run_batch_t *b, size_t cap_bytes, int n_samples, int row_width, int stats, int n_trios, int n_groups);
int run_batch_reserve(run_batch_t *b, int lines);
void run_batch_free(run_batch_t *b);
int record_passes(const run_batch_t *b, int i);
void record_counts(const run_batch_t *b, int i, const char *alt, int la, vcounts_t *v);
void order_track_keep(order_track_t *o, const char *line, size_t len);
int file_writer_start(file_writer_t *w, FILE *fd);
int file_writer_stop(file_writer_t *w);
int write_batch(FILE *fd, int kind, const run_batch_t *b, out_buf_t *bufs, int n_bufs, io_pool_t *pool, order_track_t *ord, file_writer_t *fw);

extern pthread_rwlock_t g_cohort_lock;   /* host_engine.c */
extern char g_err[512];   /* host_engine.c */
extern hpgv_run_filters_t g_filters;   /* host_source.c */
extern double g_run_times[6];   /* host_source.c */
extern char g_input_err[192];   /* host_bgzf.c */
extern int g_aggregate_overwrite;   /* host_format.c */
extern double g_write_split[2];   /* host_format.c */

#pragma GCC visibility pop
#endif
