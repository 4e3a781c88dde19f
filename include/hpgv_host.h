/*
 * hpgv_host.h -- host-side mirror of the reference's per-batch runner API.
 *
 * hpg-variant's runners call three C functions once per parsed VCF batch
 * (SURVEY.md 8b).  This header gives those functions with the reference's own
 * names, argument order, ownership and error behaviour, re-hosted on the
 * MI355X engine (include/hpgv.h).  The hpg-libs types they take are not in the
 * reference tree (lib/ is an empty submodule), so the few fields the hot path
 * touches are defined here with the reference's field names:
 *
 *   assoc_test          src/gwas/assoc/assoc.h:137-138   (body assoc.c:23-84)
 *   tdt_test            src/gwas/tdt/tdt.h:119           (body tdt.c:23-276)
 *   get_variants_stats  call site src/vcf-tools/stats/stats_runner.c:194-195
 *   result records      assoc_basic_test.h:30-46, assoc_fisher_test.h:29-44, tdt.h:104-117
 *   writers             assoc_runner.c:292-342, tdt_runner.c:286-304
 *
 * What the adapters do per call: find GT in FORMAT, turn every sample string
 * into one HPGV8 byte (the reference's strdup + get_alleles per genotype,
 * assoc.c:45-56), hand the batch to the engine, and build one result record
 * per variant on the caller's list_t, stamped with the OpenMP thread id
 * (assoc.c:25,67-68).  No statistics are computed on the host.
 */
#ifndef HPGV_HOST_H
#define HPGV_HOST_H

#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "hpgv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hpg-libs stand-in types (only what the hot path reads) ---------------- */

typedef struct array_list {            /* containers/array_list.h: items, size */
    size_t capacity;
    size_t size;
    void **items;
} array_list_t;

array_list_t *array_list_new(size_t initial_capacity);
int   array_list_insert(void *item, array_list_t *list);
void *array_list_get(size_t index, const array_list_t *list);
void  array_list_free(array_list_t *list, void (*item_free)(void *));

typedef struct vcf_record {            /* bioformats/vcf/vcf_file_structure.h; fields as used in
                                          assoc.c:40-65, tdt.c:43-48,262-266 */
    char *chromosome;  int chromosome_len;
    unsigned long position;
    char *id;          int id_len;
    char *reference;   int reference_len;
    char *alternate;   int alternate_len;
    char *format;      int format_len;
    array_list_t *samples;             /* NUL-terminated per-sample strings "0/1:..." */
} vcf_record_t;

vcf_record_t *vcf_record_new(void);
void vcf_record_free(vcf_record_t *record);          /* frees the record and its samples list, not the strings */
void set_vcf_record_chromosome(char *chromosome, int length, vcf_record_t *record);
void set_vcf_record_position(long position, vcf_record_t *record);
void set_vcf_record_id(char *id, int length, vcf_record_t *record);
void set_vcf_record_reference(char *reference, int length, vcf_record_t *record);
void set_vcf_record_alternate(char *alternate, int length, vcf_record_t *record);
void set_vcf_record_format(char *format, int length, vcf_record_t *record);

enum Sex { MALE = HPGV_SEX_MALE, FEMALE = HPGV_SEX_FEMALE, UNKNOWN_SEX = HPGV_SEX_UNKNOWN };
enum Condition { UNAFFECTED = HPGV_COND_UNAFFECTED, AFFECTED = HPGV_COND_AFFECTED,
                 MISSING_CONDITION = HPGV_COND_OTHER };

struct family;
typedef struct individual {            /* bioformats/family/family.h; ctor shape from
                                          test/test_tdt_runner.c:95-97 */
    char *id;
    float variable;
    enum Sex sex;
    enum Condition condition;
    struct individual *father;
    struct individual *mother;
    struct family *family;
} individual_t;

individual_t *individual_new(char *id, float variable, enum Sex sex, enum Condition condition,
                             individual_t *father, individual_t *mother, struct family *family);
void individual_free(individual_t *individual);

/* family->founders / family->members are khashes in the reference (tdt.c:62-66,
 * 135-137); here they are ordered lists, iterated in insertion order. */
typedef struct family {
    char *id;
    array_list_t *founders;            /* individual_t*, parents without parents */
    array_list_t *members;             /* individual_t*, everybody else */
} family_t;

family_t *family_new(char *id);
int  family_set_parent(individual_t *parent, family_t *family);
int  family_add_child(individual_t *child, family_t *family);
void family_free(family_t *family);    /* frees the lists, not the individuals */

/* khash_t(ids): sample name -> VCF column (tdt.c:83-95); open addressing, strings borrowed */
typedef struct sample_ids {
    size_t n_buckets;
    size_t size;
    const char **keys;
    int *vals;
} sample_ids_t;

sample_ids_t *sample_ids_new(size_t expected);
int  sample_ids_put(sample_ids_t *ids, const char *name, int position);
int  sample_ids_get(const sample_ids_t *ids, const char *name);   /* -1 when absent */
void sample_ids_free(sample_ids_t *ids);

/* containers/list.h: thread-safe producer/consumer list (assoc_runner.c:25,306; assoc.c:67-68) */
typedef struct list_item {
    int id;
    int type;
    void *data_p;
    struct list_item *next_p;
} list_item_t;

typedef struct list {
    char *name;
    int writers;
    size_t max_length;
    size_t length;
    list_item_t *first_p, *last_p;
    pthread_mutex_t lock;
    pthread_cond_t condition;
} list_t;

void list_init(const char *name, int writers, size_t max_length, list_t *list);
list_item_t *list_item_new(int id, int type, void *data_p);
void list_item_free(list_item_t *item);
int  list_insert_item(list_item_t *item, list_t *list);
list_item_t *list_remove_item(list_t *list);   /* blocks while empty and writers > 0; NULL at the end */
int  list_decr_writers(list_t *list);
void list_free_deep(list_t *list, void (*data_free)(void *));

/* ---- result records: field for field the reference's ----------------------- */

enum ASSOC_task { NONE = HPGV_TASK_NONE, CHI_SQUARE = HPGV_TASK_CHISQ, FISHER = HPGV_TASK_FISHER };

typedef struct {                       /* assoc_basic_test.h:30-46 */
    char *chromosome; char *id; char *reference; char *alternate;
    unsigned long int position;
    int affected1, affected2, unaffected1, unaffected2;
    double odds_ratio, chi_square, p_value;
} assoc_basic_result_t;

typedef struct {                       /* assoc_fisher_test.h:29-44 */
    char *chromosome; char *id; char *reference; char *alternate;
    unsigned long int position;
    int affected1, affected2, unaffected1, unaffected2;
    double odds_ratio, p_value;
} assoc_fisher_result_t;

typedef struct {                       /* tdt.h:104-117 */
    char *chromosome; char *id; char *reference; char *alternate;
    unsigned long int position;
    int t1, t2;
    double odds_ratio, chi_square, p_value;
} tdt_result_t;

void assoc_basic_result_free(assoc_basic_result_t *result);
void assoc_fisher_result_free(assoc_fisher_result_t *result);
void tdt_result_free(tdt_result_t *result);

/* variant_stats_t: the fields visible in the reference tree
 * (aggregate_runner.c:187-191,288-312,379-400), plus the Hardy-Weinberg record the
 * stats tool reports.  Arrays are sized by num_alleles (1 + number of ALT alleles). */
typedef struct {                       /* the counters of one phenotype group (first two alleles) */
    int alleles_count[2];
    int genotypes_count[4];            /* 0/0, 0/1, 1/0, 1/1 */
    float alleles_freq[2];
    int missing_alleles, missing_genotypes;
    float maf;
    double hw_chi2, hw_p_value;
} variant_phenotype_stats_t;

typedef struct {
    char *chromosome; unsigned long position;
    char *ref_allele; char *alt_alleles;
    int num_alleles;
    int *alleles_count;                /* [num_alleles] */
    int *genotypes_count;              /* [a1 * num_alleles + a2] */
    float *alleles_freq;
    float *genotypes_freq;
    int missing_alleles, missing_genotypes;
    float maf;
    double hw_chi2, hw_p_value;        /* on the first two alleles */
    int num_phenotypes;                /* num_variables of the call (0 without a PED) */
    variant_phenotype_stats_t *phenotype_stats;   /* [num_phenotypes]: samples whose individual->variable is that id
                                          (report_vcf_variant_phenotype_stats(fd, n, batch, i), stats_runner.c:319-323) */
} variant_stats_t;

void variant_stats_free(variant_stats_t *stats);

typedef struct {                       /* sample_stats_t (stats_runner.c:158-160): per-sample counters */
    char *name;
    int missing_genotypes;
    int mendelian_errors;              /* variants at which the sample, as a child, contradicts its parents */
} sample_stats_t;

sample_stats_t *sample_stats_new(char *name);
void sample_stats_free(sample_stats_t *stats);

typedef struct {                       /* file_stats_t: summary counters (simple sums) */
    int variants_count, samples_count, biallelics_count, multiallelics_count;
    pthread_mutex_t lock;
} file_stats_t;

file_stats_t *file_stats_new(void);
void file_stats_free(file_stats_t *stats);

/* ---- engine binding ---------------------------------------------------------- */
/* Chooses the device the adapters run on (default 0).  Called lazily by the
 * adapters; returns HPGV_OK or an hpgv status.  There is no CPU fallback. */
int  hpgv_host_init(int device_id);
/* several devices (hpgv_create_multi of include/hpgv.h): the adapters' and runners' batches are dealt to them; the
 * runners start two engine threads per device.  Without an explicit init the first call reads the environment variable
 * HPGV_DEVICES ("0,1,2,3" or "all"; the same id may repeat) and falls back to device 0. */
int  hpgv_host_init_devices(const int *device_ids, int n_devices);
int  hpgv_host_device_count(void);                     /* devices behind the adapters (0 before the first call) */
void hpgv_host_shutdown(void);
const char *hpgv_host_last_error(void);

/* ---- the reference's per-batch functions ------------------------------------- */

/* assoc.h:137-138.  void like the reference: an engine failure is logged to
 * stderr and exits the process with 1, which is what LOG_FATAL does there. */
void assoc_test(enum ASSOC_task test_type, vcf_record_t **variants, int num_variants,
                individual_t **samples, int num_samples,
                const void *opt_input, list_t *output_list);

/* assoc_runner.c:164-166 builds opt_input with init_logarithm_array(num_samples * 10) */
double *init_logarithm_array(int n);

/* tdt.h:119.  0 = OK (tdt_runner.c:186-188 treats non-zero as fatal). */
int  tdt_test(vcf_record_t **variants, int num_variants, family_t **families, int num_families,
              sample_ids_t *sample_ids, list_t *output_list);

/* call shape of stats_runner.c:194-195; returns 0 on success.  With individuals != NULL and
 * num_variables > 0 every record also carries the counters of each phenotype group: VCF column j
 * belongs to group (int)individuals[j]->variable when that is in [0, num_variables). */
int  get_variants_stats(vcf_record_t **variants, int num_variants, individual_t **individuals,
                        sample_ids_t *sample_ids, int num_variables, list_t *output_list,
                        file_stats_t *file_stats);

/* call shape of stats_runner.c:197-198; accumulates into sample_stats[j] for VCF column j */
int  get_sample_stats(vcf_record_t **variants, int num_variants, individual_t **individuals,
                      sample_ids_t *sample_ids, sample_stats_t **sample_stats, file_stats_t *file_stats);

/* ---- writers with the reference's exact formats ------------------------------ */
void assoc_write_output_header(enum ASSOC_task task, FILE *fd);             /* assoc_runner.c:292-299 */
void assoc_write_output_body(enum ASSOC_task task, list_t *output_list, FILE *fd);   /* :301-342 */
void tdt_write_output_header(FILE *fd);                                     /* tdt_runner.c:286-289 */
void tdt_write_output_body(list_t *output_list, FILE *fd);                  /* tdt_runner.c:291-304 */

/* In-process replacement of the runners' `system("sort -k1,1h -k2,2n FILE > FILE.tmp && mv ...")`
 * (assoc_runner.c:255-258, tdt_runner.c:255-261): sorts the lines of `path` (header included, as
 * the reference does) by column 1 "human numeric", then column 2 numeric, then the whole line
 * bytewise -- the order GNU sort gives under LC_ALL=C.  Returns 0, or non-zero when the file
 * cannot be read / rewritten (the reference only warns in that case). */
int  hpgv_host_sort_output_file(const char *path);

/* The characters of printf("%6f", x) -- the number format of the result files (assoc_runner.c:314-318,
 * tdt_runner.c:297-299) -- written to dst (room for 320 characters: the largest double takes 316); returns their count.  Rounds half to
 * even on the exact binary value, as glibc does; used by the file runners' writers in place of printf. */
int  hpgv_host_format_f6(double x, char *dst);

/* The runners' own raw-DEFLATE decoder for BGZF blocks (`--compression bgzip`, shared_options.c:60-61): `in` -> exactly
 * out_len bytes.  0 = decoded; non-zero = something the fast path does not take (the runners then let zlib decode and
 * judge the block).  Exported for the tests. */
int  hpgv_host_inflate_raw(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len);

/* ---- file level: what run_association_test (assoc_runner.c:23-276) and run_tdt_test
 *      (tdt_runner.c:23-279) do, minus options/filters: read the PED, read the VCF (plain text,
 *      gzip or bgzip -- `--compression`, shared_options.c:60-61; detected from the file's magic, BGZF
 *      blocks are inflated in parallel) in text batches of about batch_bytes, hand every batch to the engine as TEXT (the GPU
 *      tokenizes it), write the reference's TSV and sort it in process.  (A bgzip file of 256 blocks or more is decoded on the
 *      device and its text stays there: batch_bytes is then a lower bound, the windows are about a 64th of the text, 256 MB
 *      at most.)  Reader, engine and
 *      writer overlap (one batch in flight each way).  PED columns: FID IID PAT MAT SEX PHENO
 *      with SEX 1 = male, 2 = female and PHENO 2 = affected, 1 = unaffected
 *      (stats_runner.c:50,86-87).  Returns 0 or an hpgv / errno-style non-zero code;
 *      *n_variants_out (may be NULL) receives the number of records written. */
int  hpgv_run_assoc(const char *vcf_path, const char *ped_path, const char *out_path,
                    enum ASSOC_task task, size_t batch_bytes, long *n_variants_out);
int  hpgv_run_tdt(const char *vcf_path, const char *ped_path, const char *out_path,
                  size_t batch_bytes, long *n_variants_out);
/* record filters of the file-level runners (filter_records on every batch, assoc_runner.c:191; options
 * shared_options.c:42-47,86-115).  A negative member switches that filter off; NULL switches all off.  The
 * count-derived ones run on the GPU from the same tokenized batch:
 *   min_maf            --maf      keep records whose minor-allele frequency (first two alleles) is >= min_maf
 *   max_missing        --missing  keep records whose rate of missing genotypes is <= max_missing
 *   max_mendel_errors  --mendel   keep records with at most this many Mendelian errors over the PED's trios
 *   num_alleles        --alleles  keep records with exactly this many alleles (1 + ALT alleles)
 *   min_quality        --quality  keep records with QUAL >= min_quality
 * The filter bodies live in hpg-libs (absent from the reference tree): these are the definitions used here.
 * The setting applies to the runs started afterwards on this process. */
typedef struct {
    double min_maf, max_missing;
    int max_mendel_errors, num_alleles;
    double min_quality;
} hpgv_run_filters_t;
void hpgv_run_set_filters(const hpgv_run_filters_t *filters);

/* create_dataset_from_vcf (src/vcf-tools/vcf2epi/dataset_creator.c:24-222) without its filters: the binary
 * dataset hpgv_run_epistasis reads -- uint32 num_variants, num_affected, num_unaffected, then per variant
 * one byte per sample, cases first (0 "0/0", 1 heterozygous, 2 homozygous non-reference, 255 missing);
 * every sample that is not AFFECTED counts as unaffected (dataset_creator.c:279-300) */
int  hpgv_run_vcf2epi(const char *vcf_path, const char *ped_path, const char *out_path,
                      size_t batch_bytes, long *n_variants_out);
/* run_aggregate (src/vcf-tools/aggregate/aggregate_runner.c:23-222): the VCF without its samples, every record's
 * INFO extended by HPG_AC, HPG_AF, HPG_AN (AC / AF / AN replacing the record's own with overwrite != 0) and HPG_GTC,
 * computed by get_variants_stats on the GPU.  Values as merge_info_and_stats (:262-365) and
 * report_variant_genotypes_stats (:376-403) print them; the order of the INFO fields is original fields first, then
 * AC, AF, AN, GTC (the reference walks a hash table); QUAL is copied as written. */
int  hpgv_run_aggregate(const char *vcf_path, const char *out_path, int overwrite, size_t batch_bytes,
                        long *n_variants_out);
/* run_stats (src/vcf-tools/stats/stats_runner.c:23-420): <out_prefix>.stats-variants / .stats-samples /
 * .stats-summary from one pass over the VCF; with a PED also the Mendelian errors and one
 * <out_prefix>.phenotype-<value>.stats-variants per value of its phenotype column (stats_runner.c:267-297); ped_path
 * may be NULL.  The report
 * writers of the reference live in hpg-libs; the three files are tab-separated renderings of the same fields
 * (header lines name the columns). */
int  hpgv_run_stats(const char *vcf_path, const char *ped_path, const char *out_prefix, size_t batch_bytes,
                    long *n_variants_out);
/* stage times of the last run on this process: {read, engine, write, sort, total} seconds and the number of
 * batches; read / engine / write overlap (three threads, three batch buffers in rotation).  The reader and
 * formatter teams use HPGV_IO_THREADS threads (environment; default half the cores, at most 16);
 * HPGV_RUN_TRACE=1 prints the same numbers to stderr. */
void hpgv_host_last_run_times(double *seconds6);

/* ---- epistasis (hpg-var-gwas epi): k-fold cross-validation helpers and the run over a vcf2epi dataset ------ */
/* cross_validation.c:16-100 and :247-281, same signatures; the shuffle uses rand() */
int **get_k_folds(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k,
                  unsigned int **sizes);
uint8_t *get_k_folds_masks(unsigned int num_samples_affected, unsigned int num_samples_unaffected, unsigned int k,
                           int **folds, unsigned int *sizes);
/* run_epistasis (singlenode/epistasis_runner.c:23-330) for pairs of SNPs: dataset file in, one report
 * <out_prefix>.cv<r>.epi per repetition out (epistasis_report.c:30-81).  eval_subset HPGV_EPI_TESTING /
 * HPGV_EPI_TRAINING; eval_mode 0 = CV-c (consistency), 1 = CV-a (accuracy). */
int  hpgv_run_epistasis(const char *dataset_path, int num_folds, int num_cv_repetitions, int max_ranking_size,
                        int eval_subset, int eval_mode, const char *out_prefix);
/* the same for combinations of `order` SNPs (the --order option, main_epistasis.c:128,142): 2 and 3 through the tile scans,
 * 4 and 5 through the listed-combination kernel (hpgv.h "ANY order"); report lines as epistasis_report.c:62-77 writes them
 * for any order: "( i, j, k, l )" and risky cells "(a-b, c, d), " */
int  hpgv_run_epistasis_order(const char *dataset_path, int order, int num_folds, int num_cv_repetitions,
                              int max_ranking_size, int eval_subset, int eval_mode, const char *out_prefix);

/* The runners' reader on its own: copies `in_path` (plain / gzip / BGZF) to `out_path` in the whole-line
 * batches (at most batch_bytes each) the runners hand to the engine, optionally after consuming the VCF
 * header as the runners do; *n_batches may be NULL. */
int  hpgv_host_copy_lines(const char *in_path, const char *out_path, size_t batch_bytes, int skip_vcf_header,
                          long *n_batches);

/* ---- staging (GT text -> HPGV8), exposed for tests ---------------------------- */
int  get_field_position_in_format(const char *field, char *format);
int  get_alleles(char *sample, int genotype_position, int *allele1, int *allele2);
/* out: num_variants x num_samples bytes, row-major; is_x: num_variants flags.  A caller that is alone (not inside an
 * OpenMP parallel region) has its batch's records dealt to HPGV_STAGE_THREADS workers (environment; default 8, at
 * most the cores; 1 = never); inside the runner's own worker team the caller stages its batch itself. */
int  hpgv_host_stage_records(vcf_record_t **variants, int num_variants, int num_samples,
                             int strict, uint8_t *out, uint8_t *is_x);

/* ---- where a per-batch adapter call spends its time (tools/bench_adapter.c) ---------------------------------
 * hpgv_host_adapter_profile(1) switches the clocks on (four clock_gettime per call); hpgv_host_adapter_times
 * returns what assoc_test / tdt_test / get_variants_stats calls have accumulated since the last reset, summed over
 * the calling threads: seconds4 = {staging (sample strings -> bytes), engine (cohort check + the hpgv_* call),
 * records (result structs, strings, list inserts), whole call}; *calls (may be NULL) their number. */
void hpgv_host_set_stage_threads(int n);        /* the team of a lone caller's staging; 0 = back to HPGV_STAGE_THREADS / default */
void hpgv_host_adapter_profile(int on);
void hpgv_host_adapter_times(double *seconds4, long *calls, int reset);

/* ---- Environment ----------------------------------------------------------------------------------------------
 * libhpgv_host.so reads the environment when the engine is bound (the first adapter / runner call, or hpgv_host_init*)
 * and at the start of every file run (hpgv_run_*, hpgv_host_copy_lines) -- never per batch, per block or per launch.
 *   HPGV_DEVICES=0,1,2,3 | all   devices of the engine when no hpgv_host_init* call named them (default: device 0)
 *   HPGV_IO_THREADS=n            reader / formatter team of a run (default: half the cores, at most 16)
 *   HPGV_STAGE_THREADS=n         team a lone adapter caller's staging is dealt to (default 8; 1 = never)
 *   HPGV_ENGINE_THREADS=n        batches in flight on the devices (default: by the input)
 *   HPGV_RUN_TRACE=1             stage times of a run on stderr
 *   HPGV_BGZF_VERIFY=0           no CRC-32 check of decoded BGZF blocks (default 1: checked, on the device where decoded there)
 *   HPGV_NO_GPU_INFLATE=1        bgzip decoded on the host          HPGV_ZLIB_INFLATE=1     ... through zlib
 *   HPGV_BGZF_ONE_DEVICE=1       a group decodes a bgzip file on its first device only (default: in parts, one per device)
 *   HPGV_NO_NUMA_BIND=1          a run's threads are not moved to the GPU's NUMA node
 *   HPGV_ALWAYS_SORT=1           the output file is sorted even when it came out in order
 *   HPGV_DECODE_TILES=0          the device's CRC check leaves no tokenizer tile records (windows of decoded text are then
 *                                tokenized with their counting sweep: the A/B of that short cut)
 * Diagnosis and tests (each switches one stage of the bgzip device path to its fall-back): HPGV_NO_DEVICE_WINDOWS,
 * HPGV_NO_LARGE_WINDOWS, HPGV_BGZF_HOST_TABLE, HPGV_SERIAL_BGZF_WALK, HPGV_NO_GROWING_TEXT, HPGV_NO_LOW_PRIORITY,
 * HPGV_NO_WRITER_THREAD, HPGV_UPLOAD_SEGMENT_MB, HPGV_UPLOAD_INFLIGHT, HPGV_TEST_GPU_INFLATE_REFUSE_EVERY,
 * HPGV_TEST_SCAN_ROWS, HPGV_TEST_TEXT_ESTIMATE_PERCENT, HPGV_BGZF_PART_MIN_KB (host/hpgv_host_internal.h: host_env_t).
 * The engine underneath has its own short table: include/hpgv.h "Environment". */

#ifdef __cplusplus
}
#endif
#endif
