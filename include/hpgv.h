/*
 * hpgv.h -- C ABI of the MI355X-native per-variant statistics engine.
 *
 * This is the drop-in boundary for opencb/hpg-variant's per-batch hot path.
 * Each entry point names the reference interface it stands behind (paths are
 * relative to the reference tree):
 *
 *   hpgv_assoc*      <- assoc_test()            src/gwas/assoc/assoc.h:137-138, assoc.c:23-84
 *                       assoc_count_individual  src/gwas/assoc/assoc.c:87-134
 *                       assoc_basic_test/result src/gwas/assoc/assoc_basic_test.c:23-64
 *                       assoc_fisher_test       src/gwas/assoc/assoc_fisher_test.c:24-26
 *   hpgv_set_logfact <- init_logarithm_array()  call site src/gwas/assoc/assoc_runner.c:164-166
 *   hpgv_tdt*        <- tdt_test()              src/gwas/tdt/tdt.h:119, tdt.c:23-276
 *                       tdt_result_new          src/gwas/tdt/tdt.c:279-295
 *   hpgv_stats*      <- get_variants_stats()    call site src/vcf-tools/stats/stats_runner.c:194-195
 *   hpgv_set_cohort  <- sort_individuals() + individual->condition
 *                                               src/gwas/assoc/assoc_runner.c:138-140, assoc.c:95,101
 *   hpgv_set_families<- ped_flatten_families() + family->founders/members walk
 *                                               src/gwas/tdt/tdt_runner.c:87, tdt.c:56-95,135-148
 *
 * Plain C: opaque context, plain pointers and sizes, int status codes, caller
 * allocated outputs.  No function aborts; hpgv_last_error() gives the text.
 * The library has NO CPU fallback: without a usable HIP device hpgv_create()
 * fails with HPGV_ERR_NO_DEVICE.
 *
 * Genotype code (one byte per call, "HPGV8"):
 *      byte = (allele1 << 4) | allele2      allele index 0..14 (larger ones are stored as 14; two DIFFERENT alleles
 *                                           above 13 as 13/14, so that "15/16" stays heterozygous: tdt.c:113,185-187),
 *      nibble 0xF = missing allele, 0xFF = missing / unusable genotype.
 *   Biallelic data uses 0x00 0x01 0x10 0x11 0xFF.  The ORDER of the alleles is
 *   kept because tdt.c:119,176-213 tests them in order.
 */
#ifndef HPGV_H
#define HPGV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPGV_GT_MISSING 0xFF

/* status codes */
enum {
    HPGV_OK = 0,
    HPGV_ERR_INVALID = 1,      /* bad argument */
    HPGV_ERR_NO_DEVICE = 2,    /* no usable HIP device */
    HPGV_ERR_HIP = 3,          /* a HIP runtime call failed */
    HPGV_ERR_NOMEM = 4,
    HPGV_ERR_STATE = 5,        /* e.g. scan before the cohort was set */
    HPGV_ERR_UNSUPPORTED = 6   /* shape outside the engine's limits */
};

/* enum ASSOC_task of assoc.h:55 */
enum { HPGV_TASK_NONE = 0, HPGV_TASK_CHISQ = 1, HPGV_TASK_FISHER = 2 };
/* individual->condition / ->sex by meaning (hpg-libs enums are not in the tree) */
enum { HPGV_COND_UNAFFECTED = 0, HPGV_COND_AFFECTED = 1, HPGV_COND_OTHER = 2 };
enum { HPGV_SEX_MALE = 0, HPGV_SEX_FEMALE = 1, HPGV_SEX_UNKNOWN = 2 };

typedef struct hpgv_ctx hpgv_ctx;

/* ---- lifetime ------------------------------------------------------------ */
const char *hpgv_version(void);
int  hpgv_device_count(void);                       /* <0 on HIP failure */
int  hpgv_create(int device_id, hpgv_ctx **out);
/* Several devices behind ONE context ("the context owns the devices"; the reference's runner deals its
 * batches to num_threads workers, assoc_runner.c:106-207 -- here they land on n_devices GPUs).  A group context takes
 * every call an ordinary one takes:
 *   - cohort / option calls (hpgv_set_*) are applied to every member device;
 *   - the synchronous per-batch entry points (hpgv_assoc, hpgv_tdt, hpgv_stats*, hpgv_mendel, hpgv_epi_dataset,
 *     hpgv_tokenize, hpgv_*_text) run on the member with the fewest calls in flight, so concurrent worker threads spread
 *     over the devices; variants are independent (assoc.c:38-82), so no exchange between devices is needed, and the
 *     per-sample counters of hpgv_stats_ex / hpgv_stats_text are accumulated into the caller's host arrays by every call;
 *   - device memory, streams, the device-resident entry points (hpgv_*_dev), hpgv_text_alias and the epistasis calls
 *     refer to member 0; hpgv_group_member gives the context of another member for device-resident work there.
 * The same device id may be listed more than once (two contexts on one device: what the tests do on a one-GPU box).
 * hpgv_destroy of the group destroys its members; a member must not be destroyed on its own. */
int  hpgv_create_multi(const int *device_ids, int n_devices, hpgv_ctx **out);
int  hpgv_group_size(const hpgv_ctx *ctx);                 /* 1 for an ordinary context */
hpgv_ctx *hpgv_group_member(hpgv_ctx *ctx, int i);         /* NULL when out of range */
int  hpgv_member_device(const hpgv_ctx *ctx, int i);       /* device id of member i, -1 when out of range */
void hpgv_destroy(hpgv_ctx *ctx);
/* text of the last failure on this ctx (ctx == NULL: last hpgv_create failure
 * of the calling thread) */
const char *hpgv_last_error(const hpgv_ctx *ctx);
/* Options of a context (key, value; before the cohort is set where noted):
 *   "row_align"   bytes, power of two >= 16 (default 16; before the cohort)      "row_pad"  extra bytes per row, multiple of 16
 *   "variants_per_wave" (default 2), "blocks_per_cu" (1..8), "scan_lds"          launch shape of the resident scans
 *   "profile" 0/1                HIP events around the scan / statistics kernels (hpgv_last_kernel_ms)
 *   "fisher_cut_exp" 12..300     a Fisher tail stops below 10^-this of its largest term (default 22: below one ulp of the sum)
 *   "batch_fused" 0/1, "batch_copy" 0/1   per-batch host entry points: one fused kernel reading page-locked rows in place
 *                                (default) / the kernel chain / copy the rows first
 *   "epi_complete" 0/1, "epi_triples_1pass" 0/1     epistasis scans: the shortcuts for data without missing calls / <= 10 folds
 *   "epi_pairs_mfma", "epi_triples_mfma" 0/1 (default 1)   epistasis pair / triple ranking: cell counts on the matrix cores (0: the vector-ALU scans)
 *   "group_self_exchange" 0/1    (group contexts; tests) member 0 hands its results over through the communicator too
 * Every kernel ships in ONE form.  The forms that lost their A/B comparisons (profiles/experiments_that_did_not_pay.md) are
 * compiled only into an ablation build (-DHPGV_ABLATION: tools/build_ablation.py, used by tools/ only); there the keys
 * "pipeline", "persistent", "scan_unroll", "pipe_waves", "nontemporal", "fisher_width", "inflate_wave", "tokenizer_tiles"
 * select them, and in the shipped library any value but the shipped one is refused with HPGV_ERR_UNSUPPORTED.
 *
 * Environment.  The library reads the environment in exactly two places: hpgv_create (the table below, once per context,
 * into the context) and hpgv_group_comm_init (HPGV_RCCL_LIB).  No entry point that launches work reads it.
 *   HPGV_BATCH_FUSED=0        as option "batch_fused" 0: the per-batch calls as a chain of kernels       (default 1)
 *   HPGV_BATCH_COPY=1         as option "batch_copy" 1                                                    (default 0)
 *   HPGV_STATS_ALL2=0         every stats batch through the row-staging kernel k_stats_all -- the fall-back of the shapes
 *                             k_stats_all2 does not take (unaligned rows, very wide cohorts, > 3 masked groups)  (default 1)
 *   HPGV_ASSOC_ROWS=0         text batches counted one workgroup per row (k_batch) -- the fall-back of cohorts wider than
 *                             k_assoc_rows takes                                                          (default 1)
 *   HPGV_PINNED_NONCOHERENT=1 hpgv_host_alloc asks for non-coherent page-locked memory                   (default 0)
 *   HPGV_VMM_TRACE=1          hpgv_dev_commit narrates its mappings on stderr                             (default 0)
 *   HPGV_DECODE_TILES=0       windows of text the bgzip decoder left with its tile records (hpgv_text_alias_tiles) are
 *                             tokenized with the counting sweep all the same (A/B of that short cut)     (default 1)
 *   HPGV_RCCL_LIB=<path>      librccl to load first (then the librccl beside the HIP runtime this library is bound to,
 *                             then librccl.so.1, librccl.so, /opt/rocm/lib/...)
 * (libhpgv_host.so has a table of its own: include/hpgv_host.h "Environment".) */
int  hpgv_set_option(hpgv_ctx *ctx, const char *key, long value);

/* ---- cohort description (replicated per device; tiny) --------------------- */
/* condition[j] for VCF column j (HPGV_COND_*).  Builds the assoc layout:
 * [affected columns | pad to 16 B | unaffected columns | pad], others dropped. */
int  hpgv_set_cohort(hpgv_ctx *ctx, const uint8_t *condition, int n_samples);
int  hpgv_assoc_layout(const hpgv_ctx *ctx, int *n_affected, int *n_unaffected, size_t *pitch);

/* Families in CSR form, already filtered the way tdt.c does: family f has VCF
 * columns father_col[f] / mother_col[f] (<0: founder missing or not in the VCF,
 * family skipped, tdt.c:77-95) and its counted children
 * child_off[f]..child_off[f+1]-1 = members with both parents set, AFFECTED and
 * present in the VCF (tdt.c:139-148), in family->members iteration order. */
int  hpgv_set_families(hpgv_ctx *ctx, int n_samples, int n_families,
                       const int32_t *father_col, const int32_t *mother_col,
                       const int32_t *child_off, const int32_t *child_col,
                       const uint8_t *child_sex);
int  hpgv_tdt_layout(const hpgv_ctx *ctx, int *n_trios_fast, int *n_families_slow, size_t *pitch);

/* table[i] = ln(i!) for i < n, as init_logarithm_array(num_samples*10) builds it.
 * Must cover 2*n_samples+1 entries for the Fisher pass. */
int  hpgv_set_logfact(hpgv_ctx *ctx, const double *table, size_t n);

/* stats layout: all n_samples columns in VCF order, pad to 16 B */
int  hpgv_set_stats_cohort(hpgv_ctx *ctx, int n_samples);
int  hpgv_stats_layout(const hpgv_ctx *ctx, size_t *pitch);

/* per-phenotype statistics (stats_runner.c:47-98,300-303,319-323 write one report per phenotype
 * group): group_of_sample[j] in [0, n_groups) or < 0 for "in no group".  Builds the layout
 * HPGV_LAYOUT_STATS_GROUPS = [group 0 columns | pad16 | group 1 columns | pad16 | ...]. */
int  hpgv_set_stats_groups(hpgv_ctx *ctx, const int32_t *group_of_sample, int n_samples, int n_groups);
int  hpgv_stats_groups_layout(const hpgv_ctx *ctx, size_t *pitch, int *group_sizes /* n_groups ints, may be NULL */);

/* Mendelian errors (hpg-libs check_mendel: the --mendel filter of shared_options.c:45,101-105 and the
 * per-sample counter of get_sample_stats): every child with both parents among the VCF columns is one
 * trio, whatever its phenotype.  Layout HPGV_LAYOUT_MENDEL = [father | mother | child] class planes. */
int  hpgv_set_pedigree(hpgv_ctx *ctx, int n_samples, int n_trios, const int32_t *father_col,
                       const int32_t *mother_col, const int32_t *child_col, const uint8_t *child_sex);
int  hpgv_mendel_layout(const hpgv_ctx *ctx, size_t *pitch);

/* ---- device memory + streams (thin; callers may also pass memory owned by
 *      another runtime, e.g. a torch tensor's data_ptr) ---------------------- */
int  hpgv_dev_alloc(hpgv_ctx *ctx, size_t bytes, void **dptr);
int  hpgv_dev_free(hpgv_ctx *ctx, void *dptr);
int  hpgv_memset_dev(hpgv_ctx *ctx, void *dptr, int byte_value, size_t bytes, void *stream);   /* stream NULL: done when it returns */
/* device memory that grows in place: reserve an address range of max_bytes (costs no memory), make its first `bytes` bytes
 * usable with hpgv_dev_commit (what is backed stays backed; pieces of 64 MB), give everything back with hpgv_dev_release.
 * For a text whose size is known only when its last block has been seen.  HPGV_ERR_UNSUPPORTED: no virtual memory management.
 * hpgv_dev_commit fails with HPGV_ERR_NOMEM when the memory is not there; the pieces it mapped before failing stay mapped
 * (hpgv_dev_committed).  Sizes are rounded to the device's allocation granularity (hipMemGetAllocationGranularity). */
int  hpgv_dev_reserve(hpgv_ctx *ctx, size_t max_bytes, void **dptr);
int  hpgv_dev_commit(hpgv_ctx *ctx, void *dptr, size_t bytes);
int  hpgv_dev_committed(hpgv_ctx *ctx, void *dptr, size_t *bytes);   /* how far the range is backed (a failed commit may have grown it) */
int  hpgv_dev_release(hpgv_ctx *ctx, void *dptr);
int  hpgv_memcpy_h2d(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int  hpgv_memcpy_d2h(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int  hpgv_memcpy_h2d_async(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);   /* queued only; src page-locked and left alone until hpgv_stream_sync */
int  hpgv_stream_sync(hpgv_ctx *ctx, void *stream);   /* stream NULL = default stream */
/* page-locked host memory: buffers handed to the host entry points copy to the device at full
 * PCIe rate when they come from here (pageable memory is staged by the driver, 2-4x slower) */
/* NUMA node of the device (-1: unknown).  No reference counterpart: the file runners put their staging threads and
 * page-locked buffers there (a copy out of the page cache into buffers on the other socket ran at half the rate) */
int  hpgv_device_numa_node(hpgv_ctx *ctx, int *node);
/* a non-blocking stream of the caller's own, e.g. for copies that overlap the engine's work */
int  hpgv_stream_create(hpgv_ctx *ctx, void **stream);
int  hpgv_stream_create_low(hpgv_ctx *ctx, void **stream);      /* lowest priority: its kernels give way to the other streams' */
int  hpgv_stream_destroy(hpgv_ctx *ctx, void *stream);
int  hpgv_host_alloc(hpgv_ctx *ctx, size_t bytes, void **hptr);
int  hpgv_host_free(hpgv_ctx *ctx, void *hptr);

/* ---- layout kernels: VCF-order code matrix (device) -> engine layout ------ */
/* which: 0 assoc, 1 tdt, 2 stats, 3 stats by phenotype group, 4 Mendelian-error trios,
 * 5 epistasis dataset (assoc column order, vcf2epi codes 0/1/2/255; pads 255).  d_src rows are src_pitch bytes apart and hold
 * n_samples codes in VCF column order; d_dst rows use the layout's pitch. */
enum { HPGV_LAYOUT_ASSOC = 0, HPGV_LAYOUT_TDT = 1, HPGV_LAYOUT_STATS = 2, HPGV_LAYOUT_STATS_GROUPS = 3,
       HPGV_LAYOUT_MENDEL = 4, HPGV_LAYOUT_EPI = 5 };
int  hpgv_layout_dev(hpgv_ctx *ctx, int which, const uint8_t *d_src, size_t src_pitch,
                     int n_variants, uint8_t *d_dst, void *stream);

/* ---- synthetic cohort generator, written directly in the
 *      engine layout `which`; variant ids v0..v0+n_variants-1 ------------------ */
int  hpgv_synth_dev(hpgv_ctx *ctx, int which, uint64_t v0, int n_variants,
                    uint8_t *d_dst, void *stream);
/* same genotypes in plain VCF column order (for staging tests) */
int  hpgv_synth_raw_dev(hpgv_ctx *ctx, uint64_t v0, int n_variants, int n_samples,
                        size_t pitch, uint8_t *d_dst, void *stream);

/* ---- the hot path on device-resident data (asynchronous on `stream`) ------ */
/* scan: d_gt in assoc layout -> d_counts[v] = {A1, A2, U1, U2} (int32 x4).
 * d_is_x: per-variant flag "chromosome is exactly X" (assoc.c:94) or NULL. */
int  hpgv_assoc_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants,
                         const uint8_t *d_is_x, int32_t *d_counts, void *stream);
/* statistics from the counts; SoA outputs of n_variants doubles each */
int  hpgv_assoc_chisq_dev(hpgv_ctx *ctx, const int32_t *d_counts, int n_variants,
                          double *d_odds, double *d_chisq, double *d_p, void *stream);
int  hpgv_assoc_fisher_dev(hpgv_ctx *ctx, const int32_t *d_counts, int n_variants,
                           double *d_odds, double *d_p, void *stream);

/* d_gt in tdt layout -> d_tu[v] = {t1, t2} (int32 x2) */
int  hpgv_tdt_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants,
                       const uint8_t *d_is_x, int32_t *d_tu, void *stream);
int  hpgv_tdt_stats_dev(hpgv_ctx *ctx, const int32_t *d_tu, int n_variants,
                        double *d_odds, double *d_chisq, double *d_p, void *stream);

/* d_gt in stats layout -> per variant 8 x int32:
 *   {n_00, n_01, n_10, n_11, missing_genotypes, missing_alleles, allele0, allele1}
 * (biallelic cells of genotypes_count[a1*2+a2]; allele counts include the called
 * allele of half-missing genotypes; genotypes touching an allele index >= 2 are
 * n_samples - missing_genotypes - the four cells) and Hardy-Weinberg chi2 / p on
 * (n_00, n_01+n_10, n_11). */
int  hpgv_stats_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants,
                         int32_t *d_counts8, void *stream);
int  hpgv_stats_hwe_dev(hpgv_ctx *ctx, const int32_t *d_counts8, int n_variants,
                        double *d_chi2, double *d_p, void *stream);
/* the same counters restricted to one phenotype group; d_gt in the HPGV_LAYOUT_STATS_GROUPS layout */
int  hpgv_stats_scan_group_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, int group,
                               int32_t *d_counts8, void *stream);

/* per-sample missing-genotype counts (get_sample_stats, call site stats_runner.c:197-198) over a
 * stats-layout matrix: d_missing[j] += number of listed variants in which sample j has a missing
 * allele.  The caller zeroes d_missing (n_samples ints) before the first batch. */
int  hpgv_sample_missing_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants,
                             int32_t *d_missing, void *stream);

/* full genotype table of (multi-allelic) variants from a RAW HPGV8 matrix in VCF column order:
 * d_table[i*256 + code] = samples of variant d_variant_idx[i] (or variant i when NULL) whose
 * code byte is `code`; genotypes_count[a1*num_alleles + a2] is d_table[i*256 + (a1<<4|a2)]. */
int  hpgv_genotype_table_dev(hpgv_ctx *ctx, const uint8_t *d_raw, size_t src_pitch, int n_samples,
                             const int32_t *d_variant_idx, int n_idx, int32_t *d_table, void *stream);

/* d_gt in the HPGV_LAYOUT_MENDEL layout -> d_errors[v] = trios with a Mendelian error at variant v */
int  hpgv_mendel_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x,
                          int32_t *d_errors, void *stream);
/* d_child_errors[t] += variants at which trio t (order of hpgv_set_pedigree) has a Mendelian error;
 * the caller zeroes the n_trios counters before the first batch */
int  hpgv_mendel_children_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x,
                              int32_t *d_child_errors, void *stream);

/* count-derived variant filters (--maf, --missing of shared_options.c:42-47,86-115) from the stats
 * counters: d_keep[i] = 1 iff maf >= min_maf, maf <= max_maf and missing rate <= max_missing, where a
 * negative threshold switches that test off; maf = min(allele0, allele1) / (allele0 + allele1),
 * missing rate = missing_genotypes / n_samples of the stats cohort. */
int  hpgv_stats_filter_dev(hpgv_ctx *ctx, const int32_t *d_counts8, int n_variants,
                           double min_maf, double max_maf, double max_missing,
                           uint8_t *d_keep, void *stream);

/* ---- the variant-sharded resident scan of a GROUP context: what the runner's worker loop
 *      (assoc_runner.c:106-207, tdt_runner.c:150-200, stats_runner.c:176-215) becomes when the cohort lies in the HBM of
 *      the group's devices.  Variants are independent (assoc.c:38-82, tdt.c:41-271), so member g owns the contiguous
 *      shard [g*V/G, (g+1)*V/G) of the V variants (hpgv_group_shard), scans it on a stream of its own, and the ONE
 *      exchange is the gather of the per-variant results onto member 0: grouped ncclSend / ncclRecv over a communicator
 *      the group owns (ncclCommInitAll over the members' devices, RCCL over xGMI; librccl is loaded when
 *      hpgv_group_comm_init is called, not before).  Per-sample counters (get_sample_stats, stats_runner.c:197-198) are
 *      the one sum over variants: ncclReduce onto member 0.
 *      d_gt[g] / d_is_x[g]: member g's shard on ITS device, in the tool's engine layout (d_is_x or its entries may be
 *      NULL).  Result arrays: on member 0's device, sized for all V variants, variant v at index v (the *_dev entry
 *      points' element layouts).  The calls are asynchronous: they return when everything is queued; results are
 *      complete after hpgv_group_sync.  The gather of one call overlaps the scans of the next (each member keeps two
 *      generations of result scratch), so a caller that pipelines alternates between two sets of result arrays.
 *      A device may be listed twice only if it is member 0's (the one-GPU test rig): such members hand their results
 *      over by a device-local copy instead of RCCL, which refuses one device twice. ------------------------------ */
int  hpgv_group_comm_init(hpgv_ctx *group);               /* idempotent; HPGV_ERR_UNSUPPORTED without librccl */
int  hpgv_group_comm_ranks(const hpgv_ctx *group);        /* ranks of the group's communicator (ncclCommCount); 0 before init */
/* can librccl be loaded in this process ($HPGV_RCCL_LIB, then the usual names)?  HPGV_OK or HPGV_ERR_UNSUPPORTED;
 * `why` (may be NULL) receives the loader's messages for the candidates that failed.  Needs no context and no device. */
int  hpgv_group_rccl_probe(char *why, size_t why_cap);
int  hpgv_group_shard(const hpgv_ctx *group, int64_t n_variants, int member, int64_t *lo, int64_t *hi);
int  hpgv_group_assoc(hpgv_ctx *group, int task, const uint8_t *const *d_gt, const uint8_t *const *d_is_x,
                      int64_t n_variants, int32_t *d_counts /* V x 4 */, double *d_odds,
                      double *d_chisq /* NULL for Fisher */, double *d_p);
int  hpgv_group_tdt(hpgv_ctx *group, const uint8_t *const *d_gt, const uint8_t *const *d_is_x, int64_t n_variants,
                    int32_t *d_tu /* V x 2 */, double *d_odds, double *d_chisq, double *d_p);
/* d_sample_missing (n_samples ints on member 0, may be NULL) is OVERWRITTEN with the sum over every member's shard */
int  hpgv_group_stats(hpgv_ctx *group, const uint8_t *const *d_gt, int64_t n_variants, int32_t *d_counts8 /* V x 8 */,
                      double *d_hwe_chi2, double *d_hwe_p, int32_t *d_sample_missing);
int  hpgv_group_sync(hpgv_ctx *group);
/* The epistasis scan over the group's devices (the reference deals block coordinates to its workers,
 * singlenode/epistasis_runner.c:114-145).  hpgv_epi_set_dataset / hpgv_epi_set_folds / hpgv_epi_set_fold_masks on a GROUP
 * context give every member the dataset and the folds.  A combination belongs to its first SNP; hpgv_group_epi_share tells
 * which first SNPs [i_begin, i_end) member `member` takes -- runs with (nearly) equal numbers of combinations, for pairs cut
 * at multiples of 64.  hpgv_group_epi_rank: every member ranks its share on its own device (order 2 and 3: the tile scans;
 * 4 and 5: the listed-combination kernel), the members' per-fold top lists are gathered onto member 0 over the group's
 * communicator and merged: outputs as hpgv_epi_rank_order (8 mask words per model; orders 2 and 3 use the first),
 * identical to the one-device ranking.  Synchronous.  *scan_ms (may be NULL): the slowest member's scan time. */
int  hpgv_group_epi_share(const hpgv_ctx *group, int order, int member, int *i_begin, int *i_end);
int  hpgv_group_epi_rank(hpgv_ctx *group, int order, int subset, int max_ranking_size, int32_t *combs_out,
                         double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms);

/* duration (ms) of the last scan / statistics kernel launched through this ctx
 * when option "profile" = 1 (HIP events on the launch stream; synchronises) */
int  hpgv_last_kernel_ms(hpgv_ctx *ctx, float *scan_ms, float *stats_ms);

/* ---- per-batch host entry points (synchronous; what assoc_test / tdt_test /
 *      get_variants_stats adapters call).  gt: host, variant-major, VCF column
 *      order, rows `pitch` bytes apart.  Outputs: caller-allocated arrays of
 *      n_variants elements; chisq may be NULL for Fisher.  Thread-safe: may be
 *      called concurrently from the runner's worker threads. ---------------- */
int  hpgv_assoc(hpgv_ctx *ctx, int task, const uint8_t *gt, size_t pitch, int n_variants,
                const uint8_t *is_x,
                int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2,
                double *odds, double *chisq, double *p);
int  hpgv_tdt(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants,
              const uint8_t *is_x, int32_t *t1, int32_t *t2,
              double *odds, double *chisq, double *p);
int  hpgv_stats(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants,
                int32_t *counts8 /* n_variants x 8 */, double *hwe_chi2, double *hwe_p);
/* same batch, additionally: sample_missing (n_samples ints, ACCUMULATED into; may be NULL) and, for
 * every variant whose four biallelic cells do not cover all called genotypes, its 256-bin genotype
 * table: multi_idx[k] = variant index, multi_table[k*256 + code]; *n_multi in = capacity of both
 * arrays (in variants), out = number of such variants (filled up to the capacity). */
int  hpgv_stats_ex(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants,
                   int32_t *counts8, double *hwe_chi2, double *hwe_p, int32_t *sample_missing,
                   int32_t *multi_idx, int32_t *multi_table, int *n_multi);

/* the stats counters per phenotype group of hpgv_set_stats_groups (one report per phenotype,
 * stats_runner.c:300-303,319-323): counts8[(g * n_variants + v) * 8 + k], hwe_*[g * n_variants + v]
 * (both hwe arrays may be NULL).  Samples in no group are not counted anywhere. */
int  hpgv_stats_groups(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants,
                       int32_t *counts8, double *hwe_chi2, double *hwe_p);

/* epistasis dataset rows of vcf2epi (epistasis_dataset_process_records, dataset_creator.c:241-272):
 * out[v * (n_affected + n_unaffected) + destination] with cases first then controls, each in VCF column
 * order (group_individuals_by_phenotype, :302-320), codes 0 "0/0", 1 heterozygous, 2 homozygous
 * non-reference, 255 missing.  Uses the cohort of hpgv_set_cohort. */
int  hpgv_epi_dataset(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, uint8_t *out);

/* hpgv_stats_ex (+ hpgv_mendel) on a batch of VCF text: what get_variants_stats and get_sample_stats need
 * (stats_runner.c:194-198, aggregate_runner.c:113-114), tokenized on the device.  sample_missing (n_samples ints)
 * and child_errors (one per trio of hpgv_set_pedigree) are ACCUMULATED into; mendel_errors[v] per line; any of the
 * three may be NULL.  Multi-allelic lines as in hpgv_stats_ex. */
int  hpgv_stats_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                     uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *counts8,
                     double *hwe_chi2, double *hwe_p, int32_t *sample_missing, int32_t *multi_idx,
                     int32_t *multi_table, int *n_multi, int32_t *mendel_errors, int32_t *child_errors);

/* the same with the counters of every phenotype group of hpgv_set_stats_groups in addition (one report per phenotype,
 * stats_runner.c:300-303,319-323): group_counts8[(g * max_lines + v) * 8 + k], group_hwe_*[g * max_lines + v]; all three
 * may be NULL (then this is hpgv_stats_text) */
int  hpgv_stats_text_groups(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                            uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *counts8,
                            double *hwe_chi2, double *hwe_p, int32_t *sample_missing, int32_t *multi_idx,
                            int32_t *multi_table, int *n_multi, int32_t *mendel_errors, int32_t *child_errors,
                            int32_t *group_counts8, double *group_hwe_chi2, double *group_hwe_p);

/* record filters of the text entry points, computed on the device from the tokenized matrix before the tool's own
 * scan: --maf (keep minor-allele frequency >= min_maf), --missing (keep missing-genotype rate <= max_missing),
 * --mendel (keep Mendelian errors <= max_mendel_errors) -- shared_options.c:44-46,101-115; the filter bodies live
 * in hpg-libs (definitions as hpgv_stats_filter_dev / hpgv_mendel_scan_dev).  A negative value switches a filter off.
 * A rejected line gets HPGV_LINE_FILTERED OR-ed into its status; its statistics are still computed.  The count
 * filters need hpgv_set_stats_cohort(n_samples), the Mendel filter hpgv_set_pedigree over the same columns. */
#define HPGV_LINE_FILTERED 0x100
int  hpgv_set_text_filters(hpgv_ctx *ctx, double min_maf, double max_missing, long max_mendel_errors);

/* the same rows from a batch of VCF text (tokenized on the device like hpgv_assoc_text): row v of `out` belongs to
 * line v; lines that are not records (field_off[10 v + 5] == 0xFFFFFFFF) leave their row undefined */
int  hpgv_epi_dataset_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                           uint64_t *line_off, uint32_t *field_off, int32_t *status, uint8_t *out);

/* ---- epistasis / MDR counting (hpg-var-gwas epi: src/gwas/epistasis/model.c, mdr.c, epistasis.c;
 *      runner src/gwas/epistasis/singlenode/epistasis_runner.c) -------------------------------------
 * The input is the vcf2epi dataset (dataset.c:63-76; hpgv_epi_dataset makes its rows): one row per SNP,
 * n_affected + n_unaffected bytes, cases first, 0 / 1 / 2 = genotype, anything else = missing.  The cell
 * of a combination of `order` SNPs is numbered with the LAST SNP varying fastest
 * (get_genotype_combinations, dataset.c:170-200): order 2 -> cell = g_i * 3 + g_j. */
enum { HPGV_EPI_TESTING = 0, HPGV_EPI_TRAINING = 1 };                 /* enum evaluation_subset, model.h:73 */

/* copies the dataset to the device; until folds are given all samples form one fold */
int  hpgv_epi_set_dataset(hpgv_ctx *ctx, const uint8_t *genotypes, int n_variants, int n_affected, int n_unaffected);
/* k-fold cross-validation: fold_of_sample[s] in [0, num_folds) = the fold whose TESTING part holds sample s
 * (get_k_folds, cross_validation.c:16-100); num_folds <= 16, fewer than 65536 samples per class and fold */
int  hpgv_epi_set_folds(hpgv_ctx *ctx, const int32_t *fold_of_sample, int num_folds);
/* the same from the reference's own mask array (get_k_folds_masks, cross_validation.c:247-281):
 * num_folds x num_samples_with_padding bytes, 1 = training part, both classes padded to 16.  Masks that
 * are not a partition (a sample left out of no fold, or of two) are refused with HPGV_ERR_UNSUPPORTED. */
int  hpgv_epi_set_fold_masks(hpgv_ctx *ctx, const uint8_t *fold_masks, int num_folds);
/* combination_counts (model.c:76-124) of listed combinations (order 2 to 5, 3^order cells; combs = n_combs x order
 * SNP indices): counts_*[comb * cells + cell] */
int  hpgv_epi_counts(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs,
                     int32_t *counts_aff, int32_t *counts_unaff);
/* combination_counts_all_folds (model.c:126-206), the reference's layout for n_combs combinations in a
 * row: counts_*[(fold * n_combs + comb) * cells + cell] = count over the TRAINING part of the fold */
int  hpgv_epi_counts_all_folds(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs,
                               int32_t *counts_aff, int32_t *counts_unaff);
/* process_set_of_combinations (epistasis.c:14-95) for every pair (i, j), i_begin <= i < i_end, i < j:
 * per fold the MDR high-risk cells (mdr_high_risk_combinations2 on the training counts), the confusion
 * matrix on `subset` and the balanced accuracy (test_model, model.c:320-335).  Outputs, pairs in
 * lexicographic order: accuracy[fold * *n_pairs + p], risky_mask[...] (bit c = cell c is high risk).
 * Call with accuracy = risky_mask = NULL to get *n_pairs only. */
int  hpgv_epi_scan_pairs(hpgv_ctx *ctx, int i_begin, int i_end, int subset, double *accuracy,
                         uint16_t *risky_mask, unsigned long long *n_pairs);
/* the whole scan with the per-fold ranking of the runner (add_to_model_ranking, model.c:478-517; ties:
 * higher accuracy, then smaller (i, j)): for every fold the best max_ranking_size pairs,
 * out[fold * max_ranking_size + k], n_ranked[fold] of them.  *scan_ms (may be NULL) = device time of
 * the scan kernels. */
int  hpgv_epi_rank_pairs(hpgv_ctx *ctx, int subset, int max_ranking_size, int32_t *comb_i, int32_t *comb_j,
                         double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms);
/* the same over the pairs (i, j) with i_begin <= i < i_end only (i_begin a multiple of 64): the unit of work of
 * one GPU when the triangle is cut into row bands; the bands' lists merge into the whole ranking */
int  hpgv_epi_rank_pairs_rows(hpgv_ctx *ctx, int i_begin, int i_end, int subset, int max_ranking_size,
                              int32_t *comb_i, int32_t *comb_j, double *accuracy, uint32_t *risky_mask,
                              int32_t *n_ranked, float *scan_ms);

/* order 3: the same model for every triple i < j < k (27 cells, cell = (g_i * 3 + g_j) * 3 + g_k; at most 65535
 * samples per class, no empty fold).  hpgv_epi_scan_triples is the dense form for small sets (<= 256 SNPs):
 * accuracy / risky_mask[((fold * V + i) * V + j) * V + k], NaN / 0 where (i, j, k) is no triple;
 * hpgv_epi_rank_triples the per-fold ranking as for pairs. */
int  hpgv_epi_scan_triples(hpgv_ctx *ctx, int subset, double *accuracy, uint32_t *risky_mask);
int  hpgv_epi_rank_triples(hpgv_ctx *ctx, int subset, int max_ranking_size, int32_t *comb_i, int32_t *comb_j,
                           int32_t *comb_k, double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms);
/* ... over the triples whose FIRST SNP lies in [i_begin, i_end) only (one device's share, as hpgv_epi_rank_pairs_rows) */
int  hpgv_epi_rank_triples_rows(hpgv_ctx *ctx, int i_begin, int i_end, int subset, int max_ranking_size, int32_t *comb_i,
                                int32_t *comb_j, int32_t *comb_k, double *accuracy, uint32_t *risky_mask, int32_t *n_ranked,
                                float *scan_ms);

/* ANY order the reference's --order takes (main_epistasis.c:128,142), here 2 <= order <= 5 (3^order cells, cell = the
 * genotypes of the SNPs as base-3 digits, the last SNP the lowest: get_genotype_combinations, dataset.c:170-200; at most
 * 65535 samples per class).  The pair and triple scans above are the fast forms of orders 2 and 3; these take the
 * combinations as a LIST and work one lane per cell.
 * hpgv_epi_eval_combs: process_set_of_combinations (epistasis.c:14-95) of n_combs listed combinations (n_combs x order SNP
 * indices): accuracy[comb * num_folds + fold], risky_mask[(comb * num_folds + fold) * 8 + w] (bit c % 32 of word c / 32 =
 * cell c is high risk; may be NULL).
 * hpgv_epi_rank_order: every combination i0 < i1 < ... of the dataset, per fold the best max_ranking_size (the runner's
 * add_to_model_ranking, model.c:478-517; ties: higher accuracy, then the smaller combination):
 * combs_out[(fold * max_ranking_size + k) * order + s], accuracy[fold * max_ranking_size + k],
 * risky_mask[(fold * max_ranking_size + k) * 8 + w], n_ranked[fold].  _rows: only the combinations whose FIRST SNP lies in
 * [i_begin, i_end) -- one device's share when the first SNPs are dealt out; the shares' lists merge into the whole ranking. */
int  hpgv_epi_eval_combs(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs, int subset, double *accuracy,
                         uint32_t *risky_mask);
int  hpgv_epi_rank_order(hpgv_ctx *ctx, int order, int subset, int max_ranking_size, int32_t *combs_out, double *accuracy,
                         uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms);
int  hpgv_epi_rank_order_rows(hpgv_ctx *ctx, int order, int i_begin, int i_end, int subset, int max_ranking_size,
                              int32_t *combs_out, double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms);

/* Mendelian errors of a host batch: errors[v] per variant (may be NULL) and child_errors[t] per trio of
 * hpgv_set_pedigree, ACCUMULATED into (may be NULL) */
int  hpgv_mendel(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *is_x,
                 int32_t *errors, int32_t *child_errors);

/* ---- bgzip-compressed VCF text (--compression bgzip, shared_options.c:60-61): the raw-DEFLATE payloads of BGZF blocks
 * decoded on the device, one wave per block (any number of blocks per call; option inflate_wave = 0: one lane per block,
 * which wants a hundred thousand blocks per call).  Block b occupies d_comp[in_off[b] .. + in_len[b]) and decodes to exactly
 * out_len[b] <= 65 536 bytes at d_text + out_off[b]; d_status[b] is 0, or non-zero for a block this decoder does not take (the
 * caller decodes it on the host).  d_comp must be readable for 8 bytes past the end of the last block, d_text for 8 bytes
 * past the last block's text.  Asynchronous on `stream`. */
/* "the text of the batch is on the device at d_text already" (e.g. decoded there by hpgv_inflate_blocks_dev): a *_text entry
 * point called with host_text then tokenizes d_text in place, copies nothing up, and WRITES INTO host_text (a buffer of at
 * least text_bytes bytes) the heads of the lines -- each line up to its first sample column, CHROM .. FORMAT, or the whole
 * line when it has fewer than ten fields -- back to back; line_off[i] is the offset of line i's head in host_text, field_off
 * stays relative to the line start.  d_text = NULL removes the entry. */
int  hpgv_text_alias(hpgv_ctx *ctx, const char *host_text, const char *d_text);
/* ... and d_text is a WINDOW of a text the bgzip decoder left on the device together with the tokenizer's tile records
 * (hpgv_bgzf_verify_tiles_dev): d_text_base = where that text begins (d_text lies text-bytes inside it and begins a line),
 * d_tiles its records, n_tiles how many of them are valid.  The *_text entry points then tokenize the window WITHOUT their
 * counting sweep over it -- the text is read once (the window's last tile is counted again up to the window's end; a window
 * that reaches beyond the valid records is tokenized the ordinary way).  Same results, bit for bit. */
int  hpgv_text_alias_tiles(hpgv_ctx *ctx, const char *host_text, const char *d_text, const char *d_text_base,
                           const void *d_tiles, uint64_t n_tiles);
int  hpgv_inflate_blocks_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                             const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, uint8_t *d_text,
                             int32_t *d_status, void *stream);
/* The CRC-32 check of decoded BGZF blocks (what htslib's bgzf reader / zlib's gzread do per block behind --compression bgzip,
 * shared_options.c:60-61): for every block with d_status 0, the CRC-32 of its text against the block's trailer, which follows
 * its payload (the four bytes at d_comp + in_off + in_len); a mismatch sets d_status to HPGV_BLOCK_BAD_CRC.  Same tables as
 * hpgv_inflate_blocks_dev; asynchronous on `stream`. */
#define HPGV_BLOCK_BAD_CRC 9
int  hpgv_bgzf_verify_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                          const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, const uint8_t *d_text,
                          int32_t *d_status, void *stream);
/* The same check, and -- the wave that has just read a block's text being the cheapest place to do it -- the VCF tokenizer's
 * records of that text: per 2 KiB tile of the decoded text (tile k = bytes [2048 k, 2048 k + 2048) from d_text on) its newline
 * count, the TABs behind its last newline and where that newline is, which is all the tokenizer's first sweep computes.
 * d_tiles: hpgv_text_tiles_bytes(text bytes) bytes of device memory, ZEROED before the first call for a text; n_tiles: how
 * many records it may write (blocks whose text lies beyond are only checked).  Blocks the decoder refused (d_status != 0
 * on entry) mark their tiles "count again".  See hpgv_text_alias_tiles. */
size_t hpgv_text_tiles_bytes(uint64_t text_bytes);
int  hpgv_bgzf_verify_tiles_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, const uint8_t *d_text,
                                int32_t *d_status, void *d_tiles, uint64_t n_tiles, void *stream);
/* The rows of those tables from the compressed bytes of a bgzip file on the device (bgzf.c of htslib writes the blocks the
 * reference reads with --compression bgzip, shared_options.c:60-61): the blocks that form a chain from byte `lo` (a block
 * start; 0 at first) and end at or before `hi` (bytes [0, hi) are on the device), at most max_rows of them, written from
 * index 0 on; out_off counts from text_base.  result[0] = rows, [1] = where the chain stands (the next call's lo),
 * [2] = text_base + the text bytes of the rows, [3] = block headers seen in [lo, hi).  No rows although a whole block lies
 * in the range: the file's headers are not the ones bgzip writes, and the caller walks it itself.  Returns when `result`
 * is filled.  d_scratch: hpgv_bgzf_scan_scratch_bytes(hi - lo, max_rows) bytes of device memory. */
size_t hpgv_bgzf_scan_scratch_bytes(uint64_t range_bytes, int max_rows);
int  hpgv_bgzf_scan_dev(hpgv_ctx *ctx, const uint8_t *d_comp, uint64_t lo, uint64_t hi, uint64_t text_base, int max_rows,
                        uint64_t *d_in_off, uint32_t *d_in_len, uint64_t *d_out_off, uint32_t *d_out_len,
                        void *d_scratch, size_t scratch_bytes, uint64_t *result, void *stream);

/* ---- VCF text -> HPGV8 on the GPU (replaces the per-genotype
 *      strdup + get_alleles of assoc.c:45-56 / tdt.c:97-108,150-157) ------------------
 * text: whole VCF DATA lines (no '#' header lines), TAB separated, '\n' terminated (the last
 * line may lack it), CHROM..FORMAT then n_samples sample columns.  Outputs, all optional
 * except gt:  line_off[max_lines+1] byte offset of each line start (line_off[n_lines] = end);
 * field_off[line*10 + k] offset, relative to the line start, of field k = CHROM..FORMAT (0..8)
 * and of the first sample column (9) so the host can build result records without re-scanning;
 * gt rows in VCF column order (strict != 0: genotypes get_alleles() would not report as
 * ALLELES_OK become 0xFF; a NUL byte inside a sample column ends that column's string, as it does in the
 * reference's char* samples); is_x per line (assoc.c:94 rule); status per line: 0 ok, 1 fewer
 * than 10 columns, 2 FORMAT has no GT (row all missing), 3 fewer sample columns than n_samples.
 * More than max_lines lines: the first max_lines are parsed and *n_lines reports the true count.
 * *n_lines = -1 (hpgv_tokenize_dev only, after the stream has been waited for): the one-sweep form (option tokenizer_tiles = 2)
 * gave up waiting for another workgroup's record -- set the option to 1 and call again; hpgv_tokenize and the *_text
 * entry points do that themselves. */
int  hpgv_tokenize_dev(hpgv_ctx *ctx, const char *d_text, size_t text_bytes, int n_samples, int strict,
                       int max_lines, int *d_n_lines, uint64_t *d_line_off, uint32_t *d_field_off,
                       uint8_t *d_gt, size_t pitch, uint8_t *d_is_x, int32_t *d_status, void *stream);
/* host buffers in and out (synchronous) */
int  hpgv_tokenize(hpgv_ctx *ctx, const char *text, size_t text_bytes, int n_samples, int strict,
                   int max_lines, int *n_lines, uint64_t *line_off, uint32_t *field_off,
                   uint8_t *gt, size_t pitch, uint8_t *is_x, int32_t *status);

/* text batch in, statistics out: tokenize + layout + scan + statistics without the genotype
 * matrix ever crossing PCIe.  task as hpgv_assoc; outputs sized max_lines; *n_lines = lines in
 * the text (results are filled for min(*n_lines, max_lines)); line_off / field_off / status as
 * hpgv_tokenize (may be NULL).  The cohort (hpgv_set_cohort / hpgv_set_families) fixes n_samples. */
int  hpgv_assoc_text(hpgv_ctx *ctx, int task, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                     uint64_t *line_off, uint32_t *field_off, int32_t *status,
                     int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2,
                     double *odds, double *chisq, double *p);
int  hpgv_tdt_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                   uint64_t *line_off, uint32_t *field_off, int32_t *status,
                   int32_t *t1, int32_t *t2, double *odds, double *chisq, double *p);

/* streaming-read ceiling probe: reads `bytes` from d_buf with the scan's load
 * shape and no arithmetic; returns the kernel time in ms (diagnostic) */
int  hpgv_read_probe(hpgv_ctx *ctx, const uint8_t *d_buf, size_t bytes, int iters, float *ms);

#ifdef __cplusplus
}
#endif
#endif
