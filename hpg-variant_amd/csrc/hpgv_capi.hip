// hpgv_capi.hip -- C ABI (include/hpgv.h) over the gfx950 kernels.
//
// There is no CPU path in this library: every entry point that computes
// launches HIP kernels, and hpgv_create() fails without a device.
#include "hpgv_internal.h"
#include <cstdlib>
#include "hpgv_text_kernels.h"
#include "hpgv_text2_kernels.h"
#include "hpgv_batch_kernels.h"

namespace {

// ---- the fused per-batch path (hpgv_batch_kernels.h) ---------------------------------------------------------------
// device-visible address of host pointer p when [p, p + bytes) is page-locked (hipHostMalloc / hipHostRegister) or device
// memory; nullptr for ordinary pageable memory
static const void *mapped_view(const void *p, size_t bytes) {
    if (!p || bytes == 0) return nullptr;
    const void *ends[2] = {p, (const char *)p + bytes - 1};
    const void *dev0 = nullptr;
    for (int k = 0; k < 2; ++k) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, ends[k]) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (a.type != hipMemoryTypeHost && a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged) return nullptr;
        if (!a.devicePointer) return nullptr;
        if (k == 0) dev0 = a.devicePointer;
    }
    return dev0;
}

static int ensure_result_block(hpgv_ctx *ctx, Slot *s, size_t bytes) {
    if (s->res_cap >= bytes) return HPGV_OK;
    if (s->h_res) { (void)hipHostFree(s->h_res); s->h_res = nullptr; s->d_res = nullptr; s->res_cap = 0; }
    const size_t want = round_up(bytes + bytes / 2, 4096);
    HIPCHK(ctx, hipHostMalloc(&s->h_res, want, hipHostMallocDefault));
    HIPCHK(ctx, hipHostGetDevicePointer(&s->d_res, s->h_res, 0));
    s->res_cap = want;
    return HPGV_OK;
}

static bool batch_fused_ok(const hpgv_ctx *ctx, int n_samples) {
    return ctx->batch_fused && (size_t)n_samples + 32 <= (size_t)ctx->batch_lds_max;
}

// sources of a fused call: the caller's buffers as they are when the device can read them, the slot's copies otherwise
static int batch_sources(hpgv_ctx *ctx, Slot *s, const uint8_t *gt, size_t pitch, int n_variants, int n_samples, const uint8_t *is_x,
                         hpgv::BatchArgs *A) {
    int rc;
    const size_t bytes = (size_t)(n_variants - 1) * pitch + (size_t)n_samples;     // the last row need not be a whole pitch
    // page-locked rows are read in place by the kernel -- unless batch_copy asks for the copy engine first (it moves 2 MB in
    // 35 us where the kernel's own reads over the bus take 44; the kernel then runs on device memory)
    const void *src = ctx->batch_copy ? nullptr : mapped_view(gt, bytes);
    if (!src) {
        if ((rc = ensure(ctx, s, 0, bytes + 16))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(s->buf[0], gt, bytes, hipMemcpyHostToDevice, s->stream));
        src = s->buf[0];
    }
    const void *x = nullptr;
    if (is_x) {
        x = mapped_view(is_x, (size_t)n_variants);
        if (!x) {
            if ((rc = ensure(ctx, s, 2, (size_t)n_variants))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(s->buf[2], is_x, (size_t)n_variants, hipMemcpyHostToDevice, s->stream));
            x = s->buf[2];
        }
    }
    A->src = (const uint8_t *)src; A->src_pitch = pitch; A->n_variants = n_variants; A->n_samples = n_samples;
    A->is_x = (const uint8_t *)x;
    return HPGV_OK;
}

// ---- k_stats_all on a matrix the device can read (the tokenizer's raw matrix, or a host batch through batch_sources): every
// output is optional except the per-variant counters' record block, which the kernel always produces -------------------
struct StatsAllOut {
    int32_t *counts8 = nullptr; double *hwe_chi2 = nullptr, *hwe_p = nullptr;        // [n]
    int32_t *sample_missing = nullptr;                                                // [n_samples], accumulated into
    int32_t *mendel_errors = nullptr;                                                 // [n]
    int32_t *child_errors = nullptr;                                                  // [n_trios], accumulated into
    int32_t *group_counts8 = nullptr; double *group_hwe_chi2 = nullptr, *group_hwe_p = nullptr;   // [g * group_stride + v]
    size_t group_stride = 0;
};
// LDS the kernel needs for this cohort; 0 when it cannot run (row window + column counters + trio counters too large)
static size_t stats_all_lds(const hpgv_ctx *ctx, bool mendel) {
    const size_t ns = (size_t)ctx->stats.n_samples;
    const size_t need = (ns + 32 + 15) / 16 * 16 + (ns + 15) / 16 * 16 + (mendel ? (size_t)ctx->mendel_pchunks * 16 : 0) + 16;
    return (ctx->batch_fused && need <= (size_t)ctx->batch_lds_max) ? need : 0;
}
static int stats_all_call(hpgv_ctx *ctx, Slot *s, const uint8_t *d_src, size_t src_pitch, int n_variants, const uint8_t *d_is_x,
                          const StatsAllOut &O) {
    int rc;
    const size_t n = (size_t)n_variants;
    const int ns = ctx->stats.n_samples;
    const bool want_mendel = O.mendel_errors || O.child_errors;
    const size_t ng = O.group_counts8 ? ctx->sg_off.size() : 0, nt = want_mendel ? (size_t)ctx->mendel_trios : 0;
    const size_t rec_bytes = (1 + ng) * n * sizeof(hpgv::BatchStatsRec);
    if ((rc = ensure_result_block(ctx, s, rec_bytes + n * sizeof(int32_t) + 64))) return rc;
    const bool want_sm = O.sample_missing && ns > 0, want_ce = O.child_errors && nt > 0;
    if ((rc = ensure(ctx, s, 3, ((size_t)ns + nt + 16) * sizeof(int32_t)))) return rc;
    int32_t *d_sm = (int32_t *)s->buf[3], *d_ce = d_sm + ns;
    if (want_sm || want_ce) HIPCHK(ctx, hipMemsetAsync(d_sm, 0, ((size_t)ns + nt) * sizeof(int32_t), s->stream));
    hpgv::StatsAllArgs A;
    memset(&A, 0, sizeof A);
    A.src = d_src; A.src_pitch = src_pitch; A.n_variants = n_variants; A.n_samples = ns;
    A.is_x = d_is_x;
    A.out = (hpgv::BatchStatsRec *)s->d_res;
    A.sample_missing = want_sm ? d_sm : nullptr;
    if (want_mendel) {
        A.mendel_cols = ctx->mendel.d_col_of_pos; A.pchunks = ctx->mendel_pchunks; A.n_trios = ctx->mendel_trios;
        A.luts = ctx->mendel_luts; A.male_plane = ctx->d_mendel_male;
        A.mendel_errors = O.mendel_errors ? (int32_t *)((char *)s->d_res + rec_bytes) : nullptr;
        A.child_errors = want_ce ? d_ce : nullptr;
    }
    if (ng) {
        A.group_cols = ctx->sgroups.d_col_of_pos; A.n_groups = (int)ng;
        A.group_chunk0 = ctx->d_sg_chunks; A.group_chunks = ctx->d_sg_chunks + ng;
        A.group_out = (hpgv::BatchStatsRec *)s->d_res + n;
    }
    // a band of rows per workgroup keeps the column counters in LDS across rows; short batches stay one row per workgroup
    // (one band per workgroup slot of the chip, about three per compute unit: the band's end -- its column counters' atomics --
    // costs as much as several rows, and fewer workgroups than slots leave units idle: 16 000 rows, 8 / 21 / 42 per band:
    // 105 / 65 / 84 us)
    int rows = (n_variants + 3 * ctx->n_cus - 1) / (3 * ctx->n_cus);
#ifdef HPGV_ABLATION
    if (ctx->stats_rows) rows = (int)ctx->stats_rows;               // tuning: the band length
#endif
    rows = rows < 1 ? 1 : (rows > 255 ? 255 : rows);
    A.rows_per_block = rows; A.lds_row = (int)(((size_t)ns + 32 + 15) / 16 * 16);
    // columns owned by threads across the band (hpgv_statsall_kernels.h); what that kernel does not take -- unaligned rows,
    // very wide cohorts, many groups -- goes to the row-staging kernel
    if (!ctx->stats_all2 || hpgv_launch_stats_all2(ctx, A, &s->cnt_buf, &s->cnt_cap, s->stream) != 0) {
        const size_t lds = stats_all_lds(ctx, want_mendel);
        hipLaunchKernelGGL(hpgv::k_stats_all, dim3((unsigned)((n_variants + rows - 1) / rows)), dim3(256), lds, s->stream, A);
    }
    HIPCHK(ctx, hipGetLastError());
    std::vector<int32_t> acc;
    if (want_sm || want_ce) {
        acc.resize((size_t)ns + nt);
        HIPCHK(ctx, hipMemcpyAsync(acc.data(), d_sm, acc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    const hpgv::BatchStatsRec *r = (const hpgv::BatchStatsRec *)s->h_res;
    if (O.counts8)
        for (size_t i = 0; i < n; ++i) {
            memcpy(O.counts8 + 8 * i, r[i].c8, 8 * sizeof(int32_t));
            if (O.hwe_chi2) { O.hwe_chi2[i] = r[i].hwe_chi2; O.hwe_p[i] = r[i].hwe_p; }
        }
    for (size_t k = 0; k < ng; ++k)
        for (size_t i = 0; i < n; ++i) {
            const hpgv::BatchStatsRec &q = r[n + k * n + i];
            memcpy(O.group_counts8 + (k * O.group_stride + i) * 8, q.c8, 8 * sizeof(int32_t));
            if (O.group_hwe_chi2) { O.group_hwe_chi2[k * O.group_stride + i] = q.hwe_chi2; O.group_hwe_p[k * O.group_stride + i] = q.hwe_p; }
        }
    if (O.mendel_errors) memcpy(O.mendel_errors, (const char *)s->h_res + rec_bytes, n * sizeof(int32_t));
    if (want_sm) for (int j = 0; j < ns; ++j) O.sample_missing[j] += acc[(size_t)j];
    if (want_ce) for (size_t t = 0; t < nt; ++t) O.child_errors[t] += acc[(size_t)ns + t];
    return HPGV_OK;
}

template <int KIND>
static int launch_batch(hpgv_ctx *ctx, Slot *s, const hpgv::BatchArgs &A) {
    const size_t lds = ((size_t)A.n_samples + 15 + 15) / 16 * 16 + 16;
    hipLaunchKernelGGL((hpgv::k_batch<KIND>), dim3((unsigned)A.n_variants), dim3(256), lds, s->stream, A);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

}  // namespace

extern "C" {

const char *hpgv_version(void) { return "hpgv-mi355x 0.1 (gfx950)"; }

int hpgv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

const char *hpgv_last_error(const hpgv_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// The environment is read HERE, once per context, and nowhere else in the library (include/hpgv.h "Environment").
static void read_environment(hpgv_ctx *ctx) {
    auto num = [](const char *name, long lo, long hi, long *out) { const char *e = getenv(name); if (e && *e) { long v = atol(e); *out = v < lo ? lo : v > hi ? hi : v; } };
    num("HPGV_BATCH_COPY", 0, 1, &ctx->batch_copy);
    num("HPGV_BATCH_FUSED", 0, 1, &ctx->batch_fused);
    num("HPGV_STATS_ALL2", 0, 1, &ctx->stats_all2);
    num("HPGV_ASSOC_ROWS", 0, 1, &ctx->assoc_rows);
    num("HPGV_PINNED_NONCOHERENT", 0, 1, &ctx->pinned_noncoherent);
    num("HPGV_VMM_TRACE", 0, 1, &ctx->vmm_trace);
    num("HPGV_DECODE_TILES", 0, 1, &ctx->decode_tiles);
#ifdef HPGV_ABLATION
    num("HPGV_INFLATE_WAVE", 0, 4, &ctx->inflate_wave);
    num("HPGV_TOKENIZER_TILES", 0, 2, &ctx->tokenizer_tiles);
    num("HPGV_STATS_ROWS", 1, 255, &ctx->stats_rows);
    num("HPGV_STATS_BS", 64, 1024, &ctx->stats_bs);
    num("HPGV_STATS_DEBUG", 0, 1, &ctx->stats_debug);
    num("HPGV_FISHER_LDS", 0, 160 * 1024, &ctx->fisher_lds);
    num("HPGV_INFLATE_LDS_PAD", 0, 160 * 1024, &ctx->inflate_lds_pad);
    num("HPGV_INFLATE_WAVE_WGS", 0, 64, &ctx->inflate_wave_wgs);
    num("HPGV_INFLATE_LANE_WGS", 0, 64, &ctx->inflate_lane_wgs);
#endif
}

int hpgv_create(int device_id, hpgv_ctx **out) {
    if (!out) return fail(nullptr, HPGV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, HPGV_ERR_NO_DEVICE,
                    "no usable HIP device (%s); this engine has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, HPGV_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    hpgv_ctx *ctx = new (std::nothrow) hpgv_ctx();
    if (!ctx) return fail(nullptr, HPGV_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    DeviceGuard g(device_id);
    for (int i = 0; i < 4; ++i) {
        e = hipEventCreate(&ctx->ev[i]);
        if (e != hipSuccess) {
            int rc = fail(nullptr, HPGV_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
            delete ctx;
            return rc;
        }
    }
    hipDeviceProp_t prop;
    memset(&prop, 0, sizeof prop);
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cus = prop.multiProcessorCount;
    {   // the fused per-batch kernel stages one raw row in LDS: ask for the whole 160 KiB where the device has it
        const int want = 160 * 1024 - 1024;
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void *)hpgv::k_batch<hpgv::BATCH_CHISQ>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)hpgv::k_batch<hpgv::BATCH_FISHER>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)hpgv::k_batch<hpgv::BATCH_TDT>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)hpgv::k_batch<hpgv::BATCH_STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)hpgv::k_stats_all, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        if (ok && prop.sharedMemPerBlockOptin >= (size_t)want) ctx->batch_lds_max = want;
        else if (ok && prop.maxSharedMemoryPerMultiProcessor >= (size_t)want) ctx->batch_lds_max = want;
        (void)hipGetLastError();
    }
    read_environment(ctx);
    e = hipMalloc(&ctx->d_sink, 256);
    if (e != hipSuccess) {
        int rc = fail(nullptr, HPGV_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    *out = ctx;
    return HPGV_OK;
}

int hpgv_create_multi(const int *device_ids, int n_devices, hpgv_ctx **out) {
    if (!out) return fail(nullptr, HPGV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(nullptr, HPGV_ERR_INVALID, "hpgv_create_multi needs 1..64 device ids");
    hpgv_ctx *g = new (std::nothrow) hpgv_ctx();
    if (!g) return fail(nullptr, HPGV_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_devices; ++i) {
        hpgv_ctx *m = nullptr;
        const int rc = hpgv_create(device_ids[i], &m);          // the same id twice gives two contexts on one device
        if (rc != HPGV_OK) {
            for (hpgv_ctx *c : g->members) { c->parent = nullptr; hpgv_destroy(c); }
            g->members.clear();
            delete g;
            return rc;                                          // hpgv_last_error(NULL) holds the member's text
        }
        m->parent = g;
        g->members.push_back(m);
    }
    g->device = g->members[0]->device;
    *out = g;
    return HPGV_OK;
}

int hpgv_group_size(const hpgv_ctx *ctx) { return !ctx ? 0 : (is_group(ctx) ? (int)ctx->members.size() : 1); }

hpgv_ctx *hpgv_group_member(hpgv_ctx *ctx, int i) {
    if (!ctx) return nullptr;
    if (!is_group(ctx)) return i == 0 ? ctx : nullptr;
    return (i >= 0 && i < (int)ctx->members.size()) ? ctx->members[(size_t)i] : nullptr;
}

int hpgv_member_device(const hpgv_ctx *ctx, int i) {
    if (!ctx) return -1;
    if (!is_group(ctx)) return i == 0 ? ctx->device : -1;
    return (i >= 0 && i < (int)ctx->members.size()) ? ctx->members[(size_t)i]->device : -1;
}

}  // extern "C"
namespace {
void grow_release(hpgv_ctx::GrowRange &r) {
    size_t off = 0;
    for (size_t i = 0; i < r.pieces.size(); ++i) { (void)hipMemUnmap(r.base + off, r.sizes[i]); (void)hipMemRelease(r.pieces[i]); off += r.sizes[i]; }
    if (r.base) (void)hipMemAddressFree(r.base, r.reserved);
    r = hpgv_ctx::GrowRange();
}
}  // namespace
extern "C" {

void hpgv_destroy(hpgv_ctx *ctx) {
    if (!ctx) return;
    if (is_group(ctx)) {
        hpgv_group_release(ctx);
        for (hpgv_ctx *m : ctx->members) { m->parent = nullptr; hpgv_destroy(m); }
        ctx->members.clear();
        delete ctx;
        return;
    }
    DeviceGuard g(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_mendel_male) (void)hipFree(ctx->d_mendel_male);
    if (ctx->d_sg_chunks) (void)hipFree(ctx->d_sg_chunks);
    if (ctx->d_group_of_col) (void)hipFree(ctx->d_group_of_col);
    if (ctx->d_cond) (void)hipFree(ctx->d_cond);
    for (Layout *L : {&ctx->assoc, &ctx->tdt, &ctx->stats, &ctx->sgroups, &ctx->mendel})
        if (L->d_col_of_pos) (void)hipFree(L->d_col_of_pos);
    ctx->tdt_plan.release();
    if (ctx->d_lf_base) (void)hipFree(ctx->d_lf_base);
    if (ctx->d_thr) (void)hipFree(ctx->d_thr);
    if (ctx->d_sink) (void)hipFree(ctx->d_sink);
    if (ctx->d_crc_tab) (void)hipFree(ctx->d_crc_tab);
    for (auto *t : ctx->tok_scratch) {
        if (t->d_blocks) (void)hipFree(t->d_blocks);
        if (t->d_line_off) (void)hipFree(t->d_line_off);
        if (t->d_extra) (void)hipFree(t->d_extra);
        delete t;
    }
    ctx->tok_scratch.clear();
    for (auto &r : ctx->grow) grow_release(r);
    ctx->grow.clear();
    hpgv_epi_release(ctx->epi);
    for (Slot *s : ctx->slots) {
        for (int i = 0; i < 8; ++i) if (s->buf[i]) (void)hipFree(s->buf[i]);
        if (s->h_res) (void)hipHostFree(s->h_res);
        if (s->cnt_buf) (void)hipFree(s->cnt_buf);
        if (s->stream) (void)hipStreamDestroy(s->stream);
        delete s;
    }
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
}

// the forms that lost their A/B (profiles/experiments_that_did_not_pay.md) are compiled only with -DHPGV_ABLATION (tools/): the
// shipped library holds one form of each kernel and refuses an option value that names another
#ifdef HPGV_ABLATION
#define HPGV_SHIPPED_ONLY(ok, what)
#else
#define HPGV_SHIPPED_ONLY(ok, what) if (!(ok)) return fail(ctx, HPGV_ERR_UNSUPPORTED, what ": that form is an ablation build's (-DHPGV_ABLATION, tools/build_ablation.py)");
#endif

int hpgv_set_option(hpgv_ctx *ctx, const char *key, long value) {
    if (is_group(ctx) && key && !strcmp(key, "group_self_exchange")) { ctx->group_self_exchange = value ? 1 : 0; return HPGV_OK; }
    GROUP_ALL(ctx, hpgv_set_option(m_, key, value))
    if (!ctx || !key) return HPGV_ERR_INVALID;
    if (!strcmp(key, "row_align")) {
        if (value < 16 || (value & (value - 1))) return fail(ctx, HPGV_ERR_INVALID, "row_align must be a power of two >= 16");
        if (ctx->assoc.set || ctx->tdt.set || ctx->stats.set)
            return fail(ctx, HPGV_ERR_STATE, "row_align must be set before the cohort");
        ctx->row_align = value;
    } else if (!strcmp(key, "row_pad")) {
        if (value < 0 || value % 16 || value > 65536) return fail(ctx, HPGV_ERR_INVALID, "row_pad must be a multiple of 16 in [0, 65536]");
        if (ctx->assoc.set || ctx->tdt.set || ctx->stats.set)
            return fail(ctx, HPGV_ERR_STATE, "row_pad must be set before the cohort");
        ctx->row_pad = value;
    } else if (!strcmp(key, "variants_per_wave")) {
        if (value < 1 || value > 1024) return fail(ctx, HPGV_ERR_INVALID, "variants_per_wave out of range");
        ctx->vpw = value;
    } else if (!strcmp(key, "nontemporal")) {
        HPGV_SHIPPED_ONLY(value != 0, "nontemporal = 0")
        ctx->nontemporal = value ? 1 : 0;
    } else if (!strcmp(key, "profile")) {
        ctx->profile = value ? 1 : 0;
    } else if (!strcmp(key, "scan_unroll")) {
        if (value != 4 && value != 8 && value != 10 && value != 12 && value != 16)
            return fail(ctx, HPGV_ERR_INVALID, "scan_unroll must be one of 4, 8, 10, 12, 16");
        HPGV_SHIPPED_ONLY(value == 4, "scan_unroll other than 4")
        ctx->scan_unroll = value;
    } else if (!strcmp(key, "pipeline")) {
        HPGV_SHIPPED_ONLY(value != 0, "pipeline = 0")
        ctx->pipeline = value ? 1 : 0;
    } else if (!strcmp(key, "fisher_cut_exp")) {
        if (value < 12 || value > 300) return fail(ctx, HPGV_ERR_INVALID, "fisher_cut_exp must be in [12, 300]");
        ctx->fisher_cut_exp = value;
    } else if (!strcmp(key, "epi_complete")) {
        ctx->epi_complete = value ? 1 : 0;
    } else if (!strcmp(key, "epi_triples_mfma")) {
        ctx->epi_triples_mfma = value ? 1 : 0;
    } else if (!strcmp(key, "epi_pairs_mfma")) {
        ctx->epi_pairs_mfma = value ? 1 : 0;
    } else if (!strcmp(key, "epi_triples_1pass")) {
        if (value < 0 || value > 2) return fail(ctx, HPGV_ERR_INVALID, "epi_triples_1pass must be 0 (two passes), 1 (the cells nine at a time) or 2 (one pass, one wave per SIMD)");
        HPGV_SHIPPED_ONLY(value != 2, "epi_triples_1pass = 2")
        ctx->epi_triples_1pass = value;
    } else if (!strcmp(key, "scan_lds")) {
        if (value < 0 || value > 160 * 1024) return fail(ctx, HPGV_ERR_INVALID, "scan_lds must be in [0, 163840]");
        ctx->scan_lds = value;
    } else if (!strcmp(key, "batch_copy")) {
        ctx->batch_copy = value ? 1 : 0;
    } else if (!strcmp(key, "inflate_wave")) {
        if (value < 0 || value > 4) return fail(ctx, HPGV_ERR_INVALID, "inflate_wave must be 0 (lane per block), 1 (by the number of blocks), 2 (wave per block), 4 (wave per block, several symbols per round) or 3 (lane per block, symbol tables in LDS)");
        HPGV_SHIPPED_ONLY(value == 1 || value == 4, "inflate_wave other than 1 / 4 (a wave per block, several symbols per round)")
        ctx->inflate_wave = value;
    } else if (!strcmp(key, "tokenizer_tiles")) {
        if (value < 0 || value > 2) return fail(ctx, HPGV_ERR_INVALID, "tokenizer_tiles must be 2 (one sweep), 1 (two sweeps) or 0 (line by line)");
        HPGV_SHIPPED_ONLY(value == 1, "tokenizer_tiles other than 1 (two sweeps)")
        ctx->tokenizer_tiles = value;
    } else if (!strcmp(key, "fisher_width")) {
        if (value != 64 && value != 32 && value != 16 && value != 8) return fail(ctx, HPGV_ERR_INVALID, "fisher_width must be 64, 32, 16 or 8");
        HPGV_SHIPPED_ONLY(value == 16, "fisher_width other than 16")
        ctx->fisher_width = value;
    } else if (!strcmp(key, "batch_fused")) {
        ctx->batch_fused = value ? 1 : 0;
    } else if (!strcmp(key, "pipe_waves")) {
        if (value != 4 && value != 6 && value != 8) return fail(ctx, HPGV_ERR_INVALID, "pipe_waves must be 4, 6 or 8");
        HPGV_SHIPPED_ONLY(value == 4, "pipe_waves other than 4")
        ctx->pipe_waves = value;
    } else if (!strcmp(key, "persistent")) {
        HPGV_SHIPPED_ONLY(value == 0, "persistent = 1")
        ctx->persistent = value ? 1 : 0;
    } else if (!strcmp(key, "blocks_per_cu")) {
        if (value < 1 || value > 8) return fail(ctx, HPGV_ERR_INVALID, "blocks_per_cu must be in 1..8");
        ctx->blocks_per_cu = value;
    } else {
        return fail(ctx, HPGV_ERR_INVALID, "unknown option '%s'", key);
    }
    return HPGV_OK;
}

/* ---- cohort ---------------------------------------------------------------- */

int hpgv_set_cohort(hpgv_ctx *ctx, const uint8_t *condition, int n_samples) {
    HPGV_ABI_TRY
    GROUP_ALL(ctx, hpgv_set_cohort(m_, condition, n_samples))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!condition || n_samples < 0) return fail(ctx, HPGV_ERR_INVALID, "bad cohort arguments");
    DeviceGuard g(ctx->device);
    int nA = 0, nU = 0;
    for (int j = 0; j < n_samples; ++j) {
        if (condition[j] == HPGV_COND_AFFECTED) nA++;
        else if (condition[j] == HPGV_COND_UNAFFECTED) nU++;
    }
    size_t segA = round_up((size_t)nA, 16), segU = round_up((size_t)nU, 16);
    size_t pitch = round_up(segA + segU, (size_t)ctx->row_align) + (size_t)ctx->row_pad;
    if (pitch == 0) pitch = (size_t)ctx->row_align;
    if (!pitch_supported(pitch)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "cohort of %d samples exceeds the row-length limit", n_samples);
    Layout &L = ctx->assoc;
    L.n_samples = n_samples;
    L.pitch = pitch;
    L.col_of_pos.assign(pitch, -1);
    size_t a = 0, u = segA;
    for (int j = 0; j < n_samples; ++j) {
        if (condition[j] == HPGV_COND_AFFECTED) L.col_of_pos[a++] = j;
        else if (condition[j] == HPGV_COND_UNAFFECTED) L.col_of_pos[u++] = j;
    }
    ctx->nA = nA; ctx->nU = nU; ctx->chunksA = (int)(segA / 16);
    {   // the conditions themselves, for the kernel that counts in VCF column order with masks (k_assoc_rows)
        std::vector<uint8_t> cond(round_up((size_t)n_samples, 16) + 16, (uint8_t)HPGV_COND_OTHER);
        for (int j = 0; j < n_samples; ++j) cond[(size_t)j] = condition[j] == HPGV_COND_AFFECTED ? 1 : condition[j] == HPGV_COND_UNAFFECTED ? 0 : 2;
        if (ctx->cond_cap < cond.size()) {
            if (ctx->d_cond) { (void)hipFree(ctx->d_cond); ctx->d_cond = nullptr; ctx->cond_cap = 0; }
            HIPCHK(ctx, hipMalloc(&ctx->d_cond, cond.size()));
            ctx->cond_cap = cond.size();
        }
        HIPCHK(ctx, hipMemcpy(ctx->d_cond, cond.data(), cond.size(), hipMemcpyHostToDevice));
    }
    return upload_layout(ctx, L);
    HPGV_ABI_CATCH(ctx)
}

int hpgv_assoc_layout(const hpgv_ctx *ctx, int *n_affected, int *n_unaffected, size_t *pitch) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (n_affected) *n_affected = ctx->nA;
    if (n_unaffected) *n_unaffected = ctx->nU;
    if (pitch) *pitch = ctx->assoc.pitch;
    return HPGV_OK;
}

int hpgv_set_logfact(hpgv_ctx *ctx, const double *table, size_t n) {
    GROUP_ALL(ctx, hpgv_set_logfact(m_, table, n))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!table || n == 0) return fail(ctx, HPGV_ERR_INVALID, "empty log-factorial table");
    DeviceGuard g(ctx->device);
    ctx->n_lf = 0;
    if (ctx->cap_lf < n) {
        // two doubles of padding in front and behind: the Fisher pass reads the table two entries per load, and the neighbour
        // of a term at the edge of its support may be entry -1 or n (read, never used)
        if (ctx->d_lf_base) { (void)hipFree(ctx->d_lf_base); ctx->d_lf_base = nullptr; ctx->d_lf = nullptr; ctx->cap_lf = 0; }
        HIPCHK(ctx, hipMalloc(&ctx->d_lf_base, (n + 4) * sizeof(double)));
        HIPCHK(ctx, hipMemset(ctx->d_lf_base, 0, (n + 4) * sizeof(double)));
        ctx->d_lf = ctx->d_lf_base + 2;
        ctx->cap_lf = n;
    }
    HIPCHK(ctx, hipMemcpy(ctx->d_lf, table, n * sizeof(double), hipMemcpyHostToDevice));
    ctx->n_lf = n;
    return HPGV_OK;
}

int hpgv_set_stats_cohort(hpgv_ctx *ctx, int n_samples) {
    HPGV_ABI_TRY
    GROUP_ALL(ctx, hpgv_set_stats_cohort(m_, n_samples))
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_samples < 0) return fail(ctx, HPGV_ERR_INVALID, "negative n_samples");
    DeviceGuard g(ctx->device);
    size_t pitch = round_up(round_up((size_t)n_samples, 16), (size_t)ctx->row_align);
    if (pitch == 0) pitch = (size_t)ctx->row_align;
    if (!pitch_supported(pitch)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "cohort of %d samples exceeds the row-length limit", n_samples);
    Layout &L = ctx->stats;
    L.n_samples = n_samples;
    L.pitch = pitch;
    L.col_of_pos.assign(pitch, -1);
    for (int j = 0; j < n_samples; ++j) L.col_of_pos[j] = j;
    return upload_layout(ctx, L);
    HPGV_ABI_CATCH(ctx)
}

int hpgv_set_stats_groups(hpgv_ctx *ctx, const int32_t *group_of_sample, int n_samples, int n_groups) {
    HPGV_ABI_TRY
    GROUP_ALL(ctx, hpgv_set_stats_groups(m_, group_of_sample, n_samples, n_groups))
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_samples < 0 || n_groups < 1 || n_groups > 4096 || (n_samples > 0 && !group_of_sample))
        return fail(ctx, HPGV_ERR_INVALID, "bad stats group arguments");
    DeviceGuard g(ctx->device);
    std::vector<int> size((size_t)n_groups, 0);
    for (int j = 0; j < n_samples; ++j) {
        if (group_of_sample[j] >= n_groups) return fail(ctx, HPGV_ERR_INVALID, "group %d of sample %d out of range", group_of_sample[j], j);
        if (group_of_sample[j] >= 0) size[(size_t)group_of_sample[j]]++;
    }
    std::vector<uint32_t> off((size_t)n_groups, 0);
    size_t used = 0;
    for (int k = 0; k < n_groups; ++k) { off[(size_t)k] = (uint32_t)used; used += round_up((size_t)size[(size_t)k], 16); }
    size_t pitch = round_up(used, (size_t)ctx->row_align) + (size_t)ctx->row_pad;
    if (pitch == 0) pitch = (size_t)ctx->row_align;
    if (!pitch_supported(pitch)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "cohort of %d samples exceeds the row-length limit", n_samples);
    Layout &L = ctx->sgroups;
    L.n_samples = n_samples;
    L.pitch = pitch;
    L.col_of_pos.assign(pitch, -1);
    std::vector<size_t> fill(off.begin(), off.end());
    for (int j = 0; j < n_samples; ++j)
        if (group_of_sample[j] >= 0) L.col_of_pos[fill[(size_t)group_of_sample[j]]++] = j;
    ctx->sg_off = off;
    ctx->sg_size = size;
    {   // first chunk / chunk count per group for the one-pass stats kernel
        std::vector<int32_t> tab((size_t)n_groups * 2);
        for (int k = 0; k < n_groups; ++k) {
            tab[(size_t)k] = (int32_t)(off[(size_t)k] / 16);
            tab[(size_t)n_groups + (size_t)k] = (int32_t)(round_up((size_t)size[(size_t)k], 16) / 16);
        }
        if (ctx->sg_chunks_cap < tab.size()) {
            if (ctx->d_sg_chunks) { (void)hipFree(ctx->d_sg_chunks); ctx->d_sg_chunks = nullptr; ctx->sg_chunks_cap = 0; }
            HIPCHK(ctx, hipMalloc(&ctx->d_sg_chunks, tab.size() * sizeof(int32_t)));
            ctx->sg_chunks_cap = tab.size();
        }
        HIPCHK(ctx, hipMemcpy(ctx->d_sg_chunks, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {   // the group of every column, for the kernel that counts groups with masks over the columns in VCF order (k_stats_all2)
        std::vector<uint8_t> gid(round_up((size_t)n_samples, 16) + 16, 0xFF);
        bool all = n_samples > 0;
        for (int j = 0; j < n_samples; ++j) {
            if (group_of_sample[j] >= 0 && group_of_sample[j] < 255) gid[(size_t)j] = (uint8_t)group_of_sample[j];
            else all = false;
        }
        ctx->all_grouped = all;
        if (ctx->group_of_col_cap < gid.size()) {
            if (ctx->d_group_of_col) { (void)hipFree(ctx->d_group_of_col); ctx->d_group_of_col = nullptr; ctx->group_of_col_cap = 0; }
            HIPCHK(ctx, hipMalloc(&ctx->d_group_of_col, gid.size()));
            ctx->group_of_col_cap = gid.size();
        }
        HIPCHK(ctx, hipMemcpy(ctx->d_group_of_col, gid.data(), gid.size(), hipMemcpyHostToDevice));
    }
    return upload_layout(ctx, L);
    HPGV_ABI_CATCH(ctx)
}

int hpgv_stats_groups_layout(const hpgv_ctx *ctx, size_t *pitch, int *group_sizes) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->sgroups.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_groups has not been called");
    if (pitch) *pitch = ctx->sgroups.pitch;
    if (group_sizes) for (size_t k = 0; k < ctx->sg_size.size(); ++k) group_sizes[k] = ctx->sg_size[k];
    return HPGV_OK;
}

int hpgv_stats_layout(const hpgv_ctx *ctx, size_t *pitch) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (pitch) *pitch = ctx->stats.pitch;
    return HPGV_OK;
}

int hpgv_set_families(hpgv_ctx *ctx, int n_samples, int n_families, const int32_t *father_col,
                      const int32_t *mother_col, const int32_t *child_off, const int32_t *child_col,
                      const uint8_t *child_sex) {
    HPGV_ABI_TRY
    GROUP_ALL(ctx, hpgv_set_families(m_, n_samples, n_families, father_col, mother_col, child_off, child_col, child_sex))
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_samples < 0 || n_families < 0 || (n_families > 0 && (!father_col || !mother_col || !child_off)))
        return fail(ctx, HPGV_ERR_INVALID, "bad family arguments");
    DeviceGuard g(ctx->device);
    Layout &L = ctx->tdt;
    std::string why;
    int rc = ctx->tdt_plan.build(n_samples, n_families, father_col, mother_col, child_off, child_col,
                                 child_sex, (size_t)ctx->row_align, L.col_of_pos, L.pitch, why);
    if (rc != HPGV_OK) return fail(ctx, rc, "%s", why.c_str());
    if (!pitch_supported(L.pitch)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "pedigree exceeds the row-length limit");
    L.n_samples = n_samples;
    return upload_layout(ctx, L);
    HPGV_ABI_CATCH(ctx)
}

int hpgv_set_pedigree(hpgv_ctx *ctx, int n_samples, int n_trios, const int32_t *father_col, const int32_t *mother_col,
                      const int32_t *child_col, const uint8_t *child_sex) {
    HPGV_ABI_TRY
    GROUP_ALL(ctx, hpgv_set_pedigree(m_, n_samples, n_trios, father_col, mother_col, child_col, child_sex))
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_samples < 0 || n_trios < 0 || (n_trios > 0 && (!father_col || !mother_col || !child_col || !child_sex)))
        return fail(ctx, HPGV_ERR_INVALID, "bad pedigree arguments");
    for (int t = 0; t < n_trios; ++t)
        if (father_col[t] < 0 || father_col[t] >= n_samples || mother_col[t] < 0 || mother_col[t] >= n_samples ||
            child_col[t] < 0 || child_col[t] >= n_samples)
            return fail(ctx, HPGV_ERR_INVALID, "trio %d has a column out of range", t);
    DeviceGuard g(ctx->device);
    const size_t P16 = round_up((size_t)n_trios, 16);
    size_t pitch = round_up(3 * P16, (size_t)ctx->row_align) + (size_t)ctx->row_pad;
    if (pitch == 0) pitch = (size_t)ctx->row_align;
    if (!pitch_supported(pitch)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "pedigree exceeds the row-length limit");
    Layout &L = ctx->mendel;
    L.n_samples = n_samples;
    L.pitch = pitch;
    L.col_of_pos.assign(pitch, -1);
    std::vector<uint8_t> male(P16 ? P16 : 16, 0);
    for (int t = 0; t < n_trios; ++t) {
        L.col_of_pos[(size_t)t] = father_col[t];
        L.col_of_pos[P16 + (size_t)t] = mother_col[t];
        L.col_of_pos[2 * P16 + (size_t)t] = child_col[t];
        male[(size_t)t] = child_sex[t] == HPGV_SEX_MALE ? 0xFF : 0x00;
    }
    if (ctx->d_mendel_male) { (void)hipFree(ctx->d_mendel_male); ctx->d_mendel_male = nullptr; }
    HIPCHK(ctx, hipMalloc(&ctx->d_mendel_male, male.size()));
    HIPCHK(ctx, hipMemcpy(ctx->d_mendel_male, male.data(), male.size(), hipMemcpyHostToDevice));
    hpgv::mendel_host::build_luts(ctx->mendel_luts);
    ctx->mendel_trios = n_trios;
    ctx->mendel_pchunks = (int)(P16 / 16);
    return upload_layout(ctx, L);
    HPGV_ABI_CATCH(ctx)
}

int hpgv_mendel_layout(const hpgv_ctx *ctx, size_t *pitch) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->mendel.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_pedigree has not been called");
    if (pitch) *pitch = ctx->mendel.pitch;
    return HPGV_OK;
}

int hpgv_tdt_layout(const hpgv_ctx *ctx, int *n_trios_fast, int *n_families_slow, size_t *pitch) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->tdt.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_families has not been called");
    if (n_trios_fast) *n_trios_fast = ctx->tdt_plan.n_fast;
    if (n_families_slow) *n_families_slow = ctx->tdt_plan.n_slow_families;
    if (pitch) *pitch = ctx->tdt.pitch;
    return HPGV_OK;
}

/* ---- memory ---------------------------------------------------------------- */

int hpgv_dev_alloc(hpgv_ctx *ctx, size_t bytes, void **dptr) {
    ctx = first_member(ctx);
    if (!ctx || !dptr) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 16));
    return HPGV_OK;
}
int hpgv_dev_free(hpgv_ctx *ctx, void *dptr) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    if (dptr) HIPCHK(ctx, hipFree(dptr));
    return HPGV_OK;
}
// Device memory that grows in place: an address range is reserved (costs nothing), and backed piece by piece as the caller
// learns how much it needs -- a bgzip file's text, whose size is only known when its last block header has been seen.  The
// range behaves like any device pointer (kernels, copies).  HPGV_ERR_UNSUPPORTED when the device has no virtual memory management.
// the allocation granularity of the device's virtual memory management, as a multiple of it that is at least `at_least`
static size_t vmm_granularity(int device, size_t at_least) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) { (void)hipGetLastError(); g = (size_t)2 << 20; }
    return (at_least + g - 1) / g * g;
}
int hpgv_dev_reserve(hpgv_ctx *ctx, size_t max_bytes, void **dptr) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx || !dptr || !max_bytes) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess || !vmm)
        return fail(ctx, HPGV_ERR_UNSUPPORTED, "the device has no virtual memory management");
    const size_t gran = vmm_granularity(ctx->device, (size_t)2 << 20);      // what the device asks for, at least 2 MB
    hpgv_ctx::GrowRange r;
    r.reserved = (max_bytes + gran - 1) / gran * gran;
    void *p = nullptr;
    HIPCHK(ctx, hipMemAddressReserve(&p, r.reserved, gran, nullptr, 0));
    r.base = (char *)p;
    { std::lock_guard<std::mutex> lk(ctx->mu); ctx->grow.push_back(r); }
    *dptr = p;
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}
// makes the range's first `bytes` bytes usable (a no-op when they are already); what is backed stays backed -- also the pieces
// a failing call mapped before it failed: hpgv_dev_committed says how far the range is backed
int hpgv_dev_commit(hpgv_ctx *ctx, void *dptr, size_t bytes) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx || !dptr) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &r : ctx->grow) {
        if (r.base != (char *)dptr) continue;
        if (bytes <= r.committed) return HPGV_OK;
        if (bytes > r.reserved) return fail(ctx, HPGV_ERR_INVALID, "commit of %zu bytes in a range of %zu", bytes, r.reserved);
        const size_t gran = vmm_granularity(ctx->device, (size_t)64 << 20);      // pieces of 64 MB, rounded to the device's granularity
        size_t add = (bytes - r.committed + gran - 1) / gran * gran;
        if (r.committed + add > r.reserved) add = r.reserved - r.committed;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        // one piece for all of it; should the runtime refuse that, pieces of 64 MB
        hipError_t last = hipSuccess;
        for (size_t piece = add; add > 0; piece = gran) {
            while (add > 0) {
                const size_t n = piece < add ? piece : add;
                hipMemGenericAllocationHandle_t h;
                const bool trace = ctx->vmm_trace != 0;
                last = hipMemCreate(&h, n, &prop, 0);
                if (trace) fprintf(stderr, "vmm: create %zu MB: %s\n", n >> 20, hipGetErrorString(last));
                if (last != hipSuccess) break;
                last = hipMemMap(r.base + r.committed, n, 0, h, 0);
                if (trace) fprintf(stderr, "vmm: map at +%zu MB (%p): %s\n", r.committed >> 20, (void *)(r.base + r.committed), hipGetErrorString(last));
                if (last == hipSuccess) {
                    // access is set for EVERYTHING backed so far, not for the new piece alone: the runtime (ROCm 7.2) refuses the
                    // third and later pieces of a reservation when it is asked piece by piece ("invalid argument", eleven
                    // times out of twelve; tools/exp/vmm_seq.py), and never this form
                    last = hipMemSetAccess(r.base, r.committed + n, &acc, 1);
                    if (trace) fprintf(stderr, "vmm: set-access: %s\n", hipGetErrorString(last));
                    if (last != hipSuccess) (void)hipMemUnmap(r.base + r.committed, n);
                }
                if (last != hipSuccess) { (void)hipMemRelease(h); break; }
                r.pieces.push_back(h); r.sizes.push_back(n); r.committed += n; add -= n;
            }
            if (add == 0 || piece == gran) break;
            (void)hipGetLastError();
        }
        if (add > 0) return fail(ctx, HPGV_ERR_NOMEM, "mapping device memory (%zu bytes short): %s", add, hipGetErrorString(last));
        return HPGV_OK;
    }
    return fail(ctx, HPGV_ERR_INVALID, "not a range of hpgv_dev_reserve");
    HPGV_ABI_CATCH(ctx)
}
int hpgv_dev_committed(hpgv_ctx *ctx, void *dptr, size_t *bytes) {
    ctx = first_member(ctx);
    if (!ctx || !dptr || !bytes) return HPGV_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &r : ctx->grow) if (r.base == (char *)dptr) { *bytes = r.committed; return HPGV_OK; }
    return fail(ctx, HPGV_ERR_INVALID, "not a range of hpgv_dev_reserve");
}
int hpgv_dev_release(hpgv_ctx *ctx, void *dptr) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!dptr) return HPGV_OK;
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (size_t i = 0; i < ctx->grow.size(); ++i)
        if (ctx->grow[i].base == (char *)dptr) { grow_release(ctx->grow[i]); ctx->grow.erase(ctx->grow.begin() + (long)i); return HPGV_OK; }
    return fail(ctx, HPGV_ERR_INVALID, "not a range of hpgv_dev_reserve");
    HPGV_ABI_CATCH(ctx)
}
// NUMA node the device hangs off (sysfs numa_node of its PCI function), -1 when the system does not say: a host that
// stages batches for the device does best with its staging threads and page-locked buffers on that node
int hpgv_device_numa_node(hpgv_ctx *ctx, int *node) {
    ctx = first_member(ctx);
    if (!ctx || !node) return HPGV_ERR_INVALID;
    *node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return HPGV_OK;
    for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    if (FILE *f = fopen(path, "r")) {
        int n = -1;
        if (fscanf(f, "%d", &n) == 1) *node = n;
        fclose(f);
    }
    return HPGV_OK;
}
// a stream of the caller's own (non-blocking), e.g. for copies that should overlap the engine's work
int hpgv_stream_create(hpgv_ctx *ctx, void **stream) {
    ctx = first_member(ctx);
    if (!ctx || !stream) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    hipStream_t st = nullptr;
    HIPCHK(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return HPGV_OK;
}
// a stream whose kernels give way to those of the other streams whenever the device has a choice (bulk work beside a pipeline)
int hpgv_stream_create_low(hpgv_ctx *ctx, void **stream) {
    ctx = first_member(ctx);
    if (!ctx || !stream) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    int least = 0, greatest = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st = nullptr;
    HIPCHK(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least));
    *stream = (void *)st;
    return HPGV_OK;
}
int hpgv_stream_destroy(hpgv_ctx *ctx, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    if (stream) HIPCHK(ctx, hipStreamDestroy((hipStream_t)stream));
    return HPGV_OK;
}
// "the text at host_text is already on the device at d_text": the *_text entry points then tokenize d_text in place
// instead of copying host_text over (d_text = NULL takes the entry away).  For readers that produce the text on the
// device (hpgv_inflate_blocks_dev) and keep a host copy for the result writers.
int hpgv_memset_dev(hpgv_ctx *ctx, void *dptr, int byte_value, size_t bytes, void *stream) {
    ctx = first_member(ctx);
    if (!ctx || (bytes > 0 && !dptr)) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    if (stream) HIPCHK(ctx, hipMemsetAsync(dptr, byte_value, bytes, (hipStream_t)stream));
    else HIPCHK(ctx, hipMemset(dptr, byte_value, bytes));
    return HPGV_OK;
}
int hpgv_text_alias(hpgv_ctx *ctx, const char *host_text, const char *d_text) {
    ctx = first_member(ctx);
    if (!ctx || !host_text) return HPGV_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->alias_mu);
    for (size_t i = 0; i < ctx->text_alias.size(); ++i)
        if (ctx->text_alias[i].first == host_text) {
            const char *old = ctx->text_alias[i].second;
            for (size_t k = 0; k < ctx->text_tiles.size(); ++k) if (ctx->text_tiles[k].d_text == old) { ctx->text_tiles.erase(ctx->text_tiles.begin() + (long)k); break; }
            ctx->text_alias.erase(ctx->text_alias.begin() + (long)i);
            break;
        }
    if (d_text) ctx->text_alias.emplace_back(host_text, d_text);
    return HPGV_OK;
}
int hpgv_text_alias_tiles(hpgv_ctx *ctx, const char *host_text, const char *d_text, const char *d_text_base, const void *d_tiles, uint64_t n_tiles) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx || !host_text) return HPGV_ERR_INVALID;
    if (d_text && d_tiles && (!d_text_base || d_text < d_text_base)) return fail(ctx, HPGV_ERR_INVALID, "the window lies in front of its text");
    std::lock_guard<std::mutex> lk(ctx->alias_mu);
    const char *old = nullptr;
    for (size_t i = 0; i < ctx->text_alias.size(); ++i)
        if (ctx->text_alias[i].first == host_text) { old = ctx->text_alias[i].second; ctx->text_alias.erase(ctx->text_alias.begin() + (long)i); break; }
    for (size_t i = 0; i < ctx->text_tiles.size(); ++i)
        if (ctx->text_tiles[i].d_text == (d_text ? d_text : old)) { ctx->text_tiles.erase(ctx->text_tiles.begin() + (long)i); break; }
    if (d_text) {
        ctx->text_alias.emplace_back(host_text, d_text);
        if (d_tiles && n_tiles > 0) ctx->text_tiles.push_back({d_text, d_text_base, d_tiles, n_tiles});
    }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}
static bool tiles_of_device_text(hpgv_ctx *ctx, const char *d_text, hpgv_ctx::TextTiles *out) {
    std::lock_guard<std::mutex> lk(ctx->alias_mu);
    for (const auto &t : ctx->text_tiles) if (t.d_text == d_text) { *out = t; return true; }
    return false;
}
static const char *text_on_device(hpgv_ctx *ctx, const char *host_text) {
    std::lock_guard<std::mutex> lk(ctx->alias_mu);
    for (const auto &a : ctx->text_alias) if (a.first == host_text) return a.second;
    return nullptr;
}
// the member of a group on whose device `host_text` has been declared resident (hpgv_text_alias), or nullptr
static hpgv_ctx *alias_owner(hpgv_ctx *group, const char *host_text) {
    for (hpgv_ctx *m : group->members) if (text_on_device(m, host_text)) return m;
    return nullptr;
}
int hpgv_host_alloc(hpgv_ctx *ctx, size_t bytes, void **hptr) {
    ctx = first_member(ctx);
    if (!ctx || !hptr) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    // visible to every device of a group.  (experiment: HPGV_PINNED_NONCOHERENT=1 -- no difference measured, profiles/experiments_that_did_not_pay.md)
    const unsigned flags = hipHostMallocPortable | (ctx->pinned_noncoherent ? hipHostMallocNonCoherent : 0u);
    HIPCHK(ctx, hipHostMalloc(hptr, bytes ? bytes : 16, flags));
    return HPGV_OK;
}
int hpgv_host_free(hpgv_ctx *ctx, void *hptr) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    if (hptr) HIPCHK(ctx, hipHostFree(hptr));
    return HPGV_OK;
}
int hpgv_memcpy_h2d(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return HPGV_OK;
}
// the copy is only queued (page-locked source): hpgv_stream_sync says when it is done
int hpgv_memcpy_h2d_async(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return HPGV_OK;
}
int hpgv_memcpy_d2h(hpgv_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return HPGV_OK;
}
int hpgv_stream_sync(hpgv_ctx *ctx, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return HPGV_OK;
}

/* ---- layout + synth ------------------------------------------------------- */

static Layout *pick_layout(hpgv_ctx *ctx, int which) {
    switch (which) {
        case HPGV_LAYOUT_ASSOC: return &ctx->assoc;
        case HPGV_LAYOUT_TDT: return &ctx->tdt;
        case HPGV_LAYOUT_STATS: return &ctx->stats;
        case HPGV_LAYOUT_STATS_GROUPS: return &ctx->sgroups;
        case HPGV_LAYOUT_MENDEL: return &ctx->mendel;
        case HPGV_LAYOUT_EPI: return &ctx->assoc;
        default: return nullptr;
    }
}

// per-layout recoding of the stored byte (hpgv_kernels.h "Per-tool recoding")
static void recode_of(const hpgv_ctx *ctx, int which, int *mode, int *p16) {
    *mode = hpgv::RECODE_NONE; *p16 = 0;
    if (which == HPGV_LAYOUT_TDT) { *mode = hpgv::RECODE_TDT; *p16 = ctx->tdt_plan.p16; }
    else if (which == HPGV_LAYOUT_STATS || which == HPGV_LAYOUT_STATS_GROUPS) { *mode = hpgv::RECODE_STATS; }
    else if (which == HPGV_LAYOUT_MENDEL) { *mode = hpgv::RECODE_MENDEL; }
    else if (which == HPGV_LAYOUT_EPI) { *mode = hpgv::RECODE_EPI; }
}

int hpgv_layout_dev(hpgv_ctx *ctx, int which, const uint8_t *d_src, size_t src_pitch, int n_variants,
                    uint8_t *d_dst, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    Layout *L = pick_layout(ctx, which);
    if (!L) return fail(ctx, HPGV_ERR_INVALID, "unknown layout %d", which);
    if (!L->set) return fail(ctx, HPGV_ERR_STATE, "layout %d has no cohort yet", which);
    if (n_variants < 0 || (n_variants > 0 && (!d_src || !d_dst))) return fail(ctx, HPGV_ERR_INVALID, "bad layout arguments");
    if (src_pitch < (size_t)L->n_samples) return fail(ctx, HPGV_ERR_INVALID, "src_pitch %zu < n_samples %d", src_pitch, L->n_samples);
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    const int strict = (which == HPGV_LAYOUT_STATS || which == HPGV_LAYOUT_STATS_GROUPS) ? 0 : 1;
    int mode, p16;
    recode_of(ctx, which, &mode, &p16);
    // one thread per 16-byte chunk; slabs of variants keep a launch below 2^31 threads
    const long slab = std::max(1L, (1L << 31) / (L->chunks > 0 ? L->chunks : 1));
    for (long off = 0; off < n_variants; off += slab) {
        const int n = (int)std::min(slab, (long)n_variants - off);
        const long total = (long)n * L->chunks;
        hipLaunchKernelGGL(hpgv::k_layout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           d_src + (size_t)off * src_pitch, src_pitch, n, L->pitch, L->chunks, L->d_col_of_pos, strict, mode, p16,
                           d_dst + (size_t)off * L->pitch);
    }
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

static int ensure_thr(hpgv_ctx *ctx, int n_variants) {
    size_t need = (size_t)n_variants * 3 * sizeof(uint32_t);
    if (ctx->thr_cap >= need) return HPGV_OK;
    if (ctx->d_thr) { (void)hipFree(ctx->d_thr); ctx->d_thr = nullptr; ctx->thr_cap = 0; }
    HIPCHK(ctx, hipMalloc(&ctx->d_thr, need));
    ctx->thr_cap = need;
    return HPGV_OK;
}

static int synth_common(hpgv_ctx *ctx, uint64_t v0, int n_variants, size_t pitch, int chunks,
                        const int32_t *d_col, int mode, int p16, uint8_t *d_dst, hipStream_t st) {
    // generated in slabs so the threshold scratch stays small and a launch stays below 2^31 threads (one per 16-byte
    // chunk; a grid of more than 2^32 threads does not launch whole)
    const long by_threads = (1L << 31) / (chunks > 0 ? chunks : 1);
    const int slab = (int)std::max(1L, std::min((long)(1 << 20), by_threads));
    int rc = ensure_thr(ctx, n_variants < slab ? n_variants : slab);
    if (rc) return rc;
    for (int off = 0; off < n_variants; off += slab) {
        const int n = (n_variants - off) < slab ? (n_variants - off) : slab;
        hipLaunchKernelGGL(hpgv::k_synth_thresholds, dim3((n + 255) / 256), dim3(256), 0, st,
                           v0 + (uint64_t)off, n, ctx->d_thr);
        const long total = (long)n * chunks;
        hipLaunchKernelGGL(hpgv::k_synth_layout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           v0 + (uint64_t)off, n, pitch, chunks, d_col, ctx->d_thr, mode, p16,
                           d_dst + (size_t)off * pitch);
        HIPCHK(ctx, hipGetLastError());
    }
    return HPGV_OK;
}

int hpgv_synth_dev(hpgv_ctx *ctx, int which, uint64_t v0, int n_variants, uint8_t *d_dst, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    Layout *L = pick_layout(ctx, which);
    if (!L) return fail(ctx, HPGV_ERR_INVALID, "unknown layout %d", which);
    if (!L->set) return fail(ctx, HPGV_ERR_STATE, "layout %d has no cohort yet", which);
    if (n_variants < 0 || (n_variants > 0 && !d_dst)) return fail(ctx, HPGV_ERR_INVALID, "bad synth arguments");
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    int mode, p16;
    recode_of(ctx, which, &mode, &p16);
    std::lock_guard<std::mutex> lk(ctx->mu);   // shares ctx->d_thr
    return synth_common(ctx, v0, n_variants, L->pitch, L->chunks, L->d_col_of_pos, mode, p16, d_dst, (hipStream_t)stream);
}

int hpgv_synth_raw_dev(hpgv_ctx *ctx, uint64_t v0, int n_variants, int n_samples, size_t pitch,
                       uint8_t *d_dst, void *stream) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || n_samples < 0 || pitch % 16 || pitch < (size_t)n_samples || (n_variants > 0 && !d_dst))
        return fail(ctx, HPGV_ERR_INVALID, "bad synth_raw arguments (pitch must be a multiple of 16 >= n_samples)");
    if (n_variants == 0 || pitch == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    std::vector<int32_t> col(pitch, -1);
    for (int j = 0; j < n_samples; ++j) col[j] = j;
    int32_t *d_col = nullptr;
    HIPCHK(ctx, hipMalloc(&d_col, pitch * sizeof(int32_t)));
    hipError_t e = hipMemcpy(d_col, col.data(), pitch * sizeof(int32_t), hipMemcpyHostToDevice);
    int rc = HPGV_OK;
    if (e != hipSuccess) rc = fail(ctx, HPGV_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
    if (!rc) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        rc = synth_common(ctx, v0, n_variants, pitch, (int)(pitch / 16), d_col, hpgv::RECODE_NONE, 0, d_dst, (hipStream_t)stream);
    }
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d_col);
    return rc;
    HPGV_ABI_CATCH(ctx)
}

/* ---- assoc ------------------------------------------------------------------ */

int hpgv_assoc_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x,
                        int32_t *d_counts, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_counts))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0) return HPGV_OK;
    if (((uintptr_t)d_gt & 15) || ((uintptr_t)d_counts & 15)) return fail(ctx, HPGV_ERR_INVALID, "device buffers must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    const Layout &L = ctx->assoc;
    const int vpw = (int)ctx->vpw;
    const long waves = ((long)n_variants + vpw - 1) / vpw;
    unsigned blocks = (unsigned)((waves + 3) / 4);
    if (ctx->persistent) {
        const unsigned cap = (unsigned)(ctx->n_cus * ctx->blocks_per_cu);
        const unsigned need = (unsigned)(((long)n_variants + 3) / 4);
        blocks = need < cap ? need : cap;
    }
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *gt = d_gt;
    int4 *out = (int4 *)d_counts;
    const int cA = ctx->chunksA, ch = L.chunks;
    const size_t pitch = L.pitch;
#define HPGV_LAUNCH_PIPE(NT, U, W)                                                                \
    hipLaunchKernelGGL((hpgv::k_assoc_scan_pipe<NT, U, W>), dim3(blocks), dim3(256), 0, st, gt, pitch, \
                       n_variants, cA, ch, d_is_x, out, vpw)
#ifndef HPGV_ABLATION
    // the shipped form: software-pipelined, non-temporal loads, four tiles in flight, four waves per SIMD
    return launch_profiled(ctx, st, 0, [&] { HPGV_LAUNCH_PIPE(true, 4, 4); });
#else
#define HPGV_LAUNCH_ASSOC(NT, U, S)                                                              \
    hipLaunchKernelGGL((hpgv::k_assoc_scan<NT, U, S>), dim3(blocks), dim3(256), 0, st, gt, pitch, \
                       n_variants, cA, ch, d_is_x, out, vpw)
#define HPGV_DISPATCH_U(NT, S)                                                                   \
    switch (ctx->scan_unroll) {                                                                  \
        case 4: HPGV_LAUNCH_ASSOC(NT, 4, S); break;                                              \
        case 10: HPGV_LAUNCH_ASSOC(NT, 10, S); break;                                            \
        case 12: HPGV_LAUNCH_ASSOC(NT, 12, S); break;                                            \
        case 16: HPGV_LAUNCH_ASSOC(NT, 16, S); break;                                            \
        default: HPGV_LAUNCH_ASSOC(NT, 8, S); break;                                             \
    }
#define HPGV_PIPE_W(NT, U)                                                                       \
    do {                                                                                         \
        if (ctx->pipe_waves == 8) { HPGV_LAUNCH_PIPE(NT, U, 8); }                                \
        else if (ctx->pipe_waves == 6) { HPGV_LAUNCH_PIPE(NT, U, 6); }                           \
        else { HPGV_LAUNCH_PIPE(NT, U, 4); }                                                     \
    } while (0)
    if (ctx->pipeline && !ctx->persistent)
        return launch_profiled(ctx, st, 0, [&] {
            if (ctx->nontemporal) {
                if (ctx->scan_unroll <= 4) HPGV_PIPE_W(true, 4); else HPGV_PIPE_W(true, 5);
            } else {
                if (ctx->scan_unroll <= 4) HPGV_PIPE_W(false, 4); else HPGV_PIPE_W(false, 5);
            }
        });
    return launch_profiled(ctx, st, 0, [&] {
        if (ctx->nontemporal) {
            if (ctx->persistent) { HPGV_DISPATCH_U(true, true) } else { HPGV_DISPATCH_U(true, false) }
        } else {
            if (ctx->persistent) { HPGV_DISPATCH_U(false, true) } else { HPGV_DISPATCH_U(false, false) }
        }
    });
#undef HPGV_DISPATCH_U
#undef HPGV_PIPE_W
#undef HPGV_LAUNCH_ASSOC
#endif
#undef HPGV_LAUNCH_PIPE
}

int hpgv_assoc_chisq_dev(hpgv_ctx *ctx, const int32_t *d_counts, int n_variants, double *d_odds,
                         double *d_chisq, double *d_p, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || (n_variants > 0 && (!d_counts || !d_odds || !d_chisq || !d_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad chisq arguments");
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 1, [&] {
        hipLaunchKernelGGL(hpgv::k_assoc_chisq, dim3((n_variants + 255) / 256), dim3(256), 0, st,
                           (const int4 *)d_counts, n_variants, d_odds, d_chisq, d_p);
    });
}

int hpgv_assoc_fisher_dev(hpgv_ctx *ctx, const int32_t *d_counts, int n_variants, double *d_odds,
                          double *d_p, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || (n_variants > 0 && (!d_counts || !d_odds || !d_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad fisher arguments");
    if (!ctx->d_lf) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_logfact has not been called");
    if (ctx->assoc.set && ctx->n_lf < (size_t)2 * (ctx->nA + ctx->nU) + 1)
        return fail(ctx, HPGV_ERR_STATE, "log-factorial table has %zu entries, need %d", ctx->n_lf, 2 * (ctx->nA + ctx->nU) + 1);
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const double cut = pow(10.0, -(double)ctx->fisher_cut_exp);
    return launch_profiled(ctx, st, 1, [&] {
        // fisher_width lanes per variant: 64 / width variants per wave, 4 waves per workgroup
        const long per_block = 4 * (64 / ctx->fisher_width);
        const unsigned blocks = (unsigned)(((long)n_variants + per_block - 1) / per_block);
#ifdef HPGV_ABLATION
        const unsigned pad = (unsigned)ctx->fisher_lds;             // experiment: unused LDS per workgroup caps the pass's waves per unit (room for a scan beside it)
        if (ctx->fisher_width == 64)
            hipLaunchKernelGGL(hpgv::k_assoc_fisher<64>, dim3(blocks), dim3(256), pad, st, (const int4 *)d_counts, n_variants, ctx->d_lf, d_odds, d_p, cut);
        else if (ctx->fisher_width == 8)
            hipLaunchKernelGGL(hpgv::k_assoc_fisher<8>, dim3(blocks), dim3(256), pad, st, (const int4 *)d_counts, n_variants, ctx->d_lf, d_odds, d_p, cut);
        else if (ctx->fisher_width == 32)
            hipLaunchKernelGGL(hpgv::k_assoc_fisher<32>, dim3(blocks), dim3(256), pad, st, (const int4 *)d_counts, n_variants, ctx->d_lf, d_odds, d_p, cut);
        else
#else
        const unsigned pad = 0u;
#endif
            hipLaunchKernelGGL(hpgv::k_assoc_fisher<16>, dim3(blocks), dim3(256), pad, st, (const int4 *)d_counts, n_variants, ctx->d_lf, d_odds, d_p, cut);
    });
}

/* ---- tdt ------------------------------------------------------------------- */

int hpgv_tdt_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x,
                      int32_t *d_tu, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->tdt.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_families has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_tu))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0) return HPGV_OK;
    if (((uintptr_t)d_gt & 15) || ((uintptr_t)d_tu & 7)) return fail(ctx, HPGV_ERR_INVALID, "device buffers must be aligned");
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 0, [&] {
        ctx->tdt_plan.launch_scan(d_gt, ctx->tdt.pitch, n_variants, d_is_x, (int2 *)d_tu,
                                  (int)ctx->vpw, ctx->nontemporal != 0, st);
    });
}

int hpgv_tdt_stats_dev(hpgv_ctx *ctx, const int32_t *d_tu, int n_variants, double *d_odds,
                       double *d_chisq, double *d_p, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || (n_variants > 0 && (!d_tu || !d_odds || !d_chisq || !d_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad tdt stats arguments");
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 1, [&] {
        hipLaunchKernelGGL(hpgv::k_tdt_stats, dim3((n_variants + 255) / 256), dim3(256), 0, st,
                           (const int2 *)d_tu, n_variants, d_odds, d_chisq, d_p);
    });
}

/* ---- stats ----------------------------------------------------------------- */

int hpgv_stats_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, int32_t *d_counts8, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_counts8))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0) return HPGV_OK;
    if (((uintptr_t)d_gt & 15) || ((uintptr_t)d_counts8 & 15)) return fail(ctx, HPGV_ERR_INVALID, "device buffers must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    const Layout &L = ctx->stats;
    const int vpw = (int)ctx->vpw;
    const long waves = ((long)n_variants + vpw - 1) / vpw;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    // rows of 6.5 KB or more stream best with three workgroups (12 waves) per compute unit -- 48 KB of unused LDS per workgroup:
    // 1M x 10k: 1.59 -> 1.53 ms, 7 000 samples +4 %, 50k / 100k samples +1 - 2 %; shorter rows need every wave (5 000 samples: -12 %
    // with the cap).  Option scan_lds > 0 sets the bytes.
    const size_t scan_lds = ctx->scan_lds > 0 ? (size_t)ctx->scan_lds : (L.pitch >= 6656 ? (size_t)49152 : (size_t)0);
    return launch_profiled(ctx, st, 0, [&] {
        if (ctx->pipeline) {                              // bit-sliced counting, pipelined tiles (default)
            if (ctx->nontemporal)
                hipLaunchKernelGGL((hpgv::k_stats_scan_hs<true>), dim3(blocks), dim3(256), scan_lds, st, d_gt, L.pitch, n_variants, 0u, L.chunks, (int4 *)d_counts8, vpw);
            else
                hipLaunchKernelGGL((hpgv::k_stats_scan_hs<false>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch, n_variants, 0u, L.chunks, (int4 *)d_counts8, vpw);
        } else if (ctx->nontemporal)
            hipLaunchKernelGGL((hpgv::k_stats_scan<true, kScanUnroll>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch,
                               n_variants, 0u, L.chunks, (int4 *)d_counts8, vpw);
        else
            hipLaunchKernelGGL((hpgv::k_stats_scan<false, kScanUnroll>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch,
                               n_variants, 0u, L.chunks, (int4 *)d_counts8, vpw);
    });
}

int hpgv_stats_scan_group_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, int group, int32_t *d_counts8, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->sgroups.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_groups has not been called");
    if (group < 0 || (size_t)group >= ctx->sg_off.size()) return fail(ctx, HPGV_ERR_INVALID, "group %d out of range", group);
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_counts8))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0) return HPGV_OK;
    if (((uintptr_t)d_gt & 15) || ((uintptr_t)d_counts8 & 15)) return fail(ctx, HPGV_ERR_INVALID, "device buffers must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    const Layout &L = ctx->sgroups;
    const int vpw = (int)ctx->vpw;
    const long waves = ((long)n_variants + vpw - 1) / vpw;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const uint32_t off = ctx->sg_off[(size_t)group];
    const int chunks = (int)(round_up((size_t)ctx->sg_size[(size_t)group], 16) / 16);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 0, [&] {
        if (ctx->pipeline) {
            if (ctx->nontemporal)
                hipLaunchKernelGGL((hpgv::k_stats_scan_hs<true>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch, n_variants, off, chunks, (int4 *)d_counts8, vpw);
            else
                hipLaunchKernelGGL((hpgv::k_stats_scan_hs<false>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch, n_variants, off, chunks, (int4 *)d_counts8, vpw);
        } else if (ctx->nontemporal)
            hipLaunchKernelGGL((hpgv::k_stats_scan<true, kScanUnroll>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch,
                               n_variants, off, chunks, (int4 *)d_counts8, vpw);
        else
            hipLaunchKernelGGL((hpgv::k_stats_scan<false, kScanUnroll>), dim3(blocks), dim3(256), 0, st, d_gt, L.pitch,
                               n_variants, off, chunks, (int4 *)d_counts8, vpw);
    });
}

int hpgv_stats_hwe_dev(hpgv_ctx *ctx, const int32_t *d_counts8, int n_variants, double *d_chi2,
                       double *d_p, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || (n_variants > 0 && (!d_counts8 || !d_chi2 || !d_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad hwe arguments");
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 1, [&] {
        hipLaunchKernelGGL(hpgv::k_stats_hwe, dim3((n_variants + 255) / 256), dim3(256), 0, st,
                           (const int4 *)d_counts8, n_variants, d_chi2, d_p);
    });
}

int hpgv_mendel_scan_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x, int32_t *d_errors, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->mendel.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_pedigree has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_errors))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0) return HPGV_OK;
    if ((uintptr_t)d_gt & 15) return fail(ctx, HPGV_ERR_INVALID, "device buffers must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    const int vpw = (int)ctx->vpw;
    const long waves = ((long)n_variants + vpw - 1) / vpw;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    return launch_profiled(ctx, st, 0, [&] {
        if (ctx->nontemporal)
            hipLaunchKernelGGL((hpgv::k_mendel_scan<true, 4>), dim3(blocks), dim3(256), 0, st, d_gt, ctx->mendel.pitch, n_variants,
                               ctx->mendel_pchunks, ctx->mendel_luts, ctx->d_mendel_male, d_is_x, d_errors, vpw);
        else
            hipLaunchKernelGGL((hpgv::k_mendel_scan<false, 4>), dim3(blocks), dim3(256), 0, st, d_gt, ctx->mendel.pitch, n_variants,
                               ctx->mendel_pchunks, ctx->mendel_luts, ctx->d_mendel_male, d_is_x, d_errors, vpw);
    });
}

int hpgv_mendel_children_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, const uint8_t *d_is_x,
                             int32_t *d_child_errors, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->mendel.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_pedigree has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_child_errors))) return fail(ctx, HPGV_ERR_INVALID, "bad scan arguments");
    if (n_variants == 0 || ctx->mendel_trios == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    const unsigned tiles = (unsigned)((ctx->mendel_pchunks + 63) / 64);
    dim3 grid((tiles + 3) / 4, (unsigned)((n_variants + hpgv::SAMPLE_STATS_ROWS - 1) / hpgv::SAMPLE_STATS_ROWS));
    if (grid.y > 65535u) return fail(ctx, HPGV_ERR_UNSUPPORTED, "more than %d variants per call", 65535 * hpgv::SAMPLE_STATS_ROWS);
    hipLaunchKernelGGL(hpgv::k_mendel_children, grid, dim3(256), 0, (hipStream_t)stream, d_gt, ctx->mendel.pitch, n_variants,
                       ctx->mendel_pchunks, ctx->mendel_trios, ctx->mendel_luts, ctx->d_mendel_male, d_is_x, d_child_errors);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

int hpgv_stats_filter_dev(hpgv_ctx *ctx, const int32_t *d_counts8, int n_variants, double min_maf, double max_maf,
                          double max_missing, uint8_t *d_keep, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_counts8 || !d_keep))) return fail(ctx, HPGV_ERR_INVALID, "bad filter arguments");
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(hpgv::k_stats_filter, dim3((n_variants + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const int4 *)d_counts8, n_variants, ctx->stats.n_samples, min_maf, max_maf, max_missing, d_keep);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

int hpgv_sample_missing_dev(hpgv_ctx *ctx, const uint8_t *d_gt, int n_variants, int32_t *d_missing, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!d_gt || !d_missing))) return fail(ctx, HPGV_ERR_INVALID, "bad sample stats arguments");
    if (n_variants == 0 || ctx->stats.n_samples == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    const Layout &L = ctx->stats;
    const unsigned tiles = (unsigned)((L.chunks + 63) / 64);
    dim3 grid((tiles + 3) / 4, (unsigned)((n_variants + hpgv::SAMPLE_STATS_ROWS - 1) / hpgv::SAMPLE_STATS_ROWS));
    if (grid.y > 65535u) return fail(ctx, HPGV_ERR_UNSUPPORTED, "more than %d variants per sample-stats call", 65535 * hpgv::SAMPLE_STATS_ROWS);
    hipLaunchKernelGGL(hpgv::k_sample_missing, grid, dim3(256), 0, (hipStream_t)stream, d_gt, L.pitch, n_variants,
                       L.chunks, L.n_samples, d_missing);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

int hpgv_genotype_table_dev(hpgv_ctx *ctx, const uint8_t *d_raw, size_t src_pitch, int n_samples,
                            const int32_t *d_variant_idx, int n_idx, int32_t *d_table, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_idx < 0 || n_samples < 0 || (n_idx > 0 && (!d_raw || !d_table)) || src_pitch < (size_t)n_samples)
        return fail(ctx, HPGV_ERR_INVALID, "bad genotype table arguments");
    if (n_idx == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(hpgv::k_genotype_table, dim3((unsigned)n_idx), dim3(256), 0, (hipStream_t)stream, d_raw, src_pitch,
                       n_samples, d_variant_idx, n_idx, d_table);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

int hpgv_last_kernel_ms(hpgv_ctx *ctx, float *scan_ms, float *stats_ms) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    if (scan_ms) {
        *scan_ms = -1.f;
        if (ctx->have_scan_ev) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ev[1]));
            HIPCHK(ctx, hipEventElapsedTime(scan_ms, ctx->ev[0], ctx->ev[1]));
        }
    }
    if (stats_ms) {
        *stats_ms = -1.f;
        if (ctx->have_stats_ev) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ev[3]));
            HIPCHK(ctx, hipEventElapsedTime(stats_ms, ctx->ev[2], ctx->ev[3]));
        }
    }
    return HPGV_OK;
}

/* ---- synchronous per-batch host entry points ------------------------------ */

// slot buffers: 0 raw gt, 1 laid-out gt, 2 is_x, 3 counts, 4 doubles (3n), 5 int SoA
static int stage_batch(hpgv_ctx *ctx, Slot *s, int which, const Layout &L, const uint8_t *gt, size_t pitch,
                       int n_variants, const uint8_t *is_x, const uint8_t **d_isx_out) {
    int rc;
    if ((rc = ensure(ctx, s, 0, (size_t)n_variants * pitch))) return rc;
    if ((rc = ensure(ctx, s, 1, (size_t)n_variants * L.pitch))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(s->buf[0], gt, (size_t)n_variants * pitch, hipMemcpyHostToDevice, s->stream));
    *d_isx_out = nullptr;
    if (is_x) {
        if ((rc = ensure(ctx, s, 2, (size_t)n_variants))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(s->buf[2], is_x, (size_t)n_variants, hipMemcpyHostToDevice, s->stream));
        *d_isx_out = (const uint8_t *)s->buf[2];
    }
    return hpgv_layout_dev(ctx, which, (const uint8_t *)s->buf[0], pitch, n_variants, (uint8_t *)s->buf[1], s->stream);
}

int hpgv_assoc(hpgv_ctx *ctx, int task, const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *is_x,
               int32_t *A1, int32_t *A2, int32_t *U1, int32_t *U2, double *odds, double *chisq, double *p) {
    GROUP_DEAL(ctx, hpgv_assoc(m_, task, gt, pitch, n_variants, is_x, A1, A2, U1, U2, odds, chisq, p))
    if (!ctx) return HPGV_ERR_INVALID;
    if (task != HPGV_TASK_CHISQ && task != HPGV_TASK_FISHER) return fail(ctx, HPGV_ERR_INVALID, "task must be CHISQ or FISHER");
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!gt || !A1 || !A2 || !U1 || !U2 || !odds || !p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad assoc arguments");
    if (task == HPGV_TASK_CHISQ && n_variants > 0 && !chisq) return fail(ctx, HPGV_ERR_INVALID, "chisq output is NULL");
    if (pitch < (size_t)ctx->assoc.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->assoc.n_samples);
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    const size_t n = (size_t)n_variants;
    if (batch_fused_ok(ctx, ctx->assoc.n_samples)) {
        // one kernel: raw rows (read in place from page-locked memory) -> counts -> statistics -> packed records in the
        // slot's page-locked block; the only other work of the call is unpacking them
        if (task == HPGV_TASK_FISHER) {
            if (!ctx->d_lf) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_logfact has not been called");
            if (ctx->n_lf < (size_t)2 * (ctx->nA + ctx->nU) + 1)
                return fail(ctx, HPGV_ERR_STATE, "log-factorial table has %zu entries, need %d", ctx->n_lf, 2 * (ctx->nA + ctx->nU) + 1);
        }
        hpgv::BatchArgs A;
        memset(&A, 0, sizeof A);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ctx->assoc.n_samples, is_x, &A))) return rc;
        if ((rc = ensure_result_block(ctx, s, n * sizeof(hpgv::BatchAssocRec)))) return rc;
        A.col_of_pos = ctx->assoc.d_col_of_pos; A.chunks = ctx->assoc.chunks; A.chunksA = ctx->chunksA;
        A.lf = ctx->d_lf; A.rel_cut = pow(10.0, -(double)ctx->fisher_cut_exp);
        A.out = s->d_res;
        if (task == HPGV_TASK_CHISQ) rc = launch_batch<hpgv::BATCH_CHISQ>(ctx, s, A);
        else rc = launch_batch<hpgv::BATCH_FISHER>(ctx, s, A);
        if (rc) { (void)hipStreamSynchronize(s->stream); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        const hpgv::BatchAssocRec *r = (const hpgv::BatchAssocRec *)s->h_res;
        for (size_t i = 0; i < n; ++i) {
            A1[i] = r[i].A1; A2[i] = r[i].A2; U1[i] = r[i].U1; U2[i] = r[i].U2;
            odds[i] = r[i].odds; p[i] = r[i].p;
        }
        if (task == HPGV_TASK_CHISQ) for (size_t i = 0; i < n; ++i) chisq[i] = r[i].chisq;
        return HPGV_OK;
    }
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_ASSOC, ctx->assoc, gt, pitch, n_variants, is_x, &d_isx))) return rc;
    if ((rc = ensure(ctx, s, 3, n * 16))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 3 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, s, 5, n * 4 * sizeof(int32_t)))) return rc;
    int32_t *d_counts = (int32_t *)s->buf[3];
    double *d_odds = (double *)s->buf[4], *d_chisq = d_odds + n, *d_p = d_odds + 2 * n;
    int32_t *d_soa = (int32_t *)s->buf[5];
    if ((rc = hpgv_assoc_scan_dev(ctx, (const uint8_t *)s->buf[1], n_variants, d_isx, d_counts, s->stream))) return rc;
    if (task == HPGV_TASK_CHISQ) rc = hpgv_assoc_chisq_dev(ctx, d_counts, n_variants, d_odds, d_chisq, d_p, s->stream);
    else rc = hpgv_assoc_fisher_dev(ctx, d_counts, n_variants, d_odds, d_p, s->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(hpgv::k_counts_to_soa, dim3((n_variants + 255) / 256), dim3(256), 0, s->stream,
                       (const int4 *)d_counts, n_variants, d_soa, d_soa + n, d_soa + 2 * n, d_soa + 3 * n);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(A1, d_soa, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(A2, d_soa + n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(U1, d_soa + 2 * n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(U2, d_soa + 3 * n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(odds, d_odds, n * 8, hipMemcpyDeviceToHost, s->stream));
    if (task == HPGV_TASK_CHISQ) HIPCHK(ctx, hipMemcpyAsync(chisq, d_chisq, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

int hpgv_tdt(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *is_x,
             int32_t *t1, int32_t *t2, double *odds, double *chisq, double *p) {
    HPGV_ABI_TRY
    GROUP_DEAL(ctx, hpgv_tdt(m_, gt, pitch, n_variants, is_x, t1, t2, odds, chisq, p))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->tdt.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_families has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!gt || !t1 || !t2 || !odds || !chisq || !p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad tdt arguments");
    if (pitch < (size_t)ctx->tdt.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->tdt.n_samples);
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    const size_t n = (size_t)n_variants;
    if (batch_fused_ok(ctx, ctx->tdt.n_samples)) {
        hpgv::BatchArgs A;
        memset(&A, 0, sizeof A);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ctx->tdt.n_samples, is_x, &A))) return rc;
        if ((rc = ensure_result_block(ctx, s, n * sizeof(hpgv::BatchTdtRec)))) return rc;
        const hpgv::TdtPlan &P = ctx->tdt_plan;
        A.col_of_pos = ctx->tdt.d_col_of_pos; A.chunks = ctx->tdt.chunks;
        A.pchunks = P.pchunks; A.p16 = P.p16; A.n_slow = P.n_slow_families; A.slow_base = P.slow_base; A.luts = P.luts;
        A.male_plane = P.d_male_plane; A.slow_off = P.d_slow_off; A.slow_male = P.d_slow_male;
        A.out = s->d_res;
        if ((rc = launch_batch<hpgv::BATCH_TDT>(ctx, s, A))) { (void)hipStreamSynchronize(s->stream); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        const hpgv::BatchTdtRec *r = (const hpgv::BatchTdtRec *)s->h_res;
        for (size_t i = 0; i < n; ++i) { t1[i] = r[i].t1; t2[i] = r[i].t2; odds[i] = r[i].odds; chisq[i] = r[i].chisq; p[i] = r[i].p; }
        return HPGV_OK;
    }
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_TDT, ctx->tdt, gt, pitch, n_variants, is_x, &d_isx))) return rc;
    if ((rc = ensure(ctx, s, 3, n * 8))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 3 * sizeof(double)))) return rc;
    int32_t *d_tu = (int32_t *)s->buf[3];
    double *d_odds = (double *)s->buf[4], *d_chisq = d_odds + n, *d_p = d_odds + 2 * n;
    if ((rc = hpgv_tdt_scan_dev(ctx, (const uint8_t *)s->buf[1], n_variants, d_isx, d_tu, s->stream))) return rc;
    if ((rc = hpgv_tdt_stats_dev(ctx, d_tu, n_variants, d_odds, d_chisq, d_p, s->stream))) return rc;
    std::vector<int32_t> tu(2 * n);
    HIPCHK(ctx, hipMemcpyAsync(tu.data(), d_tu, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(odds, d_odds, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(chisq, d_chisq, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < n; ++i) { t1[i] = tu[2 * i]; t2[i] = tu[2 * i + 1]; }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_stats_ex(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, int32_t *counts8,
                  double *hwe_chi2, double *hwe_p, int32_t *sample_missing, int32_t *multi_idx,
                  int32_t *multi_table, int *n_multi) {
    HPGV_ABI_TRY
    GROUP_DEAL(ctx, hpgv_stats_ex(m_, gt, pitch, n_variants, counts8, hwe_chi2, hwe_p, sample_missing, multi_idx, multi_table, n_multi))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!gt || !counts8 || !hwe_chi2 || !hwe_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad stats arguments");
    if (n_multi && *n_multi > 0 && (!multi_idx || !multi_table)) return fail(ctx, HPGV_ERR_INVALID, "multi-allelic outputs are NULL");
    if (pitch < (size_t)ctx->stats.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->stats.n_samples);
    const int cap = n_multi ? *n_multi : 0;
    if (n_multi) *n_multi = 0;
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    const size_t n = (size_t)n_variants;
    const int ns = ctx->stats.n_samples;
    if (sample_missing && ns > 0 && stats_all_lds(ctx, false) != 0) {
        // get_sample_stats' shape (stats_runner.c:197-198): counters, Hardy-Weinberg and the per-sample missing counts in one
        // pass over the batch, read in place when it lies in page-locked memory
        hpgv::BatchArgs B;
        memset(&B, 0, sizeof B);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ns, nullptr, &B))) return rc;
        StatsAllOut O;
        O.counts8 = counts8; O.hwe_chi2 = hwe_chi2; O.hwe_p = hwe_p; O.sample_missing = sample_missing;
        if ((rc = stats_all_call(ctx, s, B.src, pitch, n_variants, nullptr, O))) return rc;
        if (n_multi) {
            std::vector<int32_t> idx;
            for (size_t i = 0; i < n; ++i) {
                const int32_t *c = counts8 + 8 * i;
                if (ns - c[4] - (c[0] + c[1] + c[2] + c[3]) > 0) idx.push_back((int32_t)i);
            }
            *n_multi = (int)idx.size();
            const int m = (int)idx.size() < cap ? (int)idx.size() : cap;
            if (m > 0) {
                if ((rc = ensure(ctx, s, 6, (size_t)m * sizeof(int32_t)))) return rc;
                if ((rc = ensure(ctx, s, 7, (size_t)m * 256 * sizeof(int32_t)))) return rc;
                HIPCHK(ctx, hipMemcpyAsync(s->buf[6], idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
                if ((rc = hpgv_genotype_table_dev(ctx, B.src, pitch, ns, (const int32_t *)s->buf[6], m, (int32_t *)s->buf[7], s->stream))) return rc;
                HIPCHK(ctx, hipMemcpyAsync(multi_table, s->buf[7], (size_t)m * 256 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
                HIPCHK(ctx, hipStreamSynchronize(s->stream));
                memcpy(multi_idx, idx.data(), (size_t)m * sizeof(int32_t));
            }
        }
        return HPGV_OK;
    }
    if (batch_fused_ok(ctx, ns) && !(sample_missing && ns > 0)) {
        // get_variants_stats' shape: counters + Hardy-Weinberg per variant in one kernel; the 256-bin tables of the rare
        // multi-allelic variants are counted from the same raw rows afterwards
        hpgv::BatchArgs A;
        memset(&A, 0, sizeof A);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ns, nullptr, &A))) return rc;
        if ((rc = ensure_result_block(ctx, s, n * sizeof(hpgv::BatchStatsRec)))) return rc;
        A.col_of_pos = ctx->stats.d_col_of_pos; A.chunks = ctx->stats.chunks;
        A.out = s->d_res;
        if ((rc = launch_batch<hpgv::BATCH_STATS>(ctx, s, A))) { (void)hipStreamSynchronize(s->stream); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        const hpgv::BatchStatsRec *r = (const hpgv::BatchStatsRec *)s->h_res;
        for (size_t i = 0; i < n; ++i) {
            memcpy(counts8 + 8 * i, r[i].c8, 8 * sizeof(int32_t));
            hwe_chi2[i] = r[i].hwe_chi2; hwe_p[i] = r[i].hwe_p;
        }
        if (n_multi) {
            std::vector<int32_t> idx;
            for (size_t i = 0; i < n; ++i) {
                const int32_t *c = counts8 + 8 * i;
                if (ns - c[4] - (c[0] + c[1] + c[2] + c[3]) > 0) idx.push_back((int32_t)i);
            }
            *n_multi = (int)idx.size();
            const int m = (int)idx.size() < cap ? (int)idx.size() : cap;
            if (m > 0) {
                if ((rc = ensure(ctx, s, 6, (size_t)m * sizeof(int32_t)))) return rc;
                if ((rc = ensure(ctx, s, 7, (size_t)m * 256 * sizeof(int32_t)))) return rc;
                HIPCHK(ctx, hipMemcpyAsync(s->buf[6], idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
                if ((rc = hpgv_genotype_table_dev(ctx, A.src, pitch, ns, (const int32_t *)s->buf[6], m, (int32_t *)s->buf[7], s->stream))) {
                    (void)hipStreamSynchronize(s->stream);
                    return rc;
                }
                HIPCHK(ctx, hipMemcpyAsync(multi_table, s->buf[7], (size_t)m * 256 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
                HIPCHK(ctx, hipStreamSynchronize(s->stream));
                memcpy(multi_idx, idx.data(), (size_t)m * sizeof(int32_t));
            }
        }
        return HPGV_OK;
    }
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_STATS, ctx->stats, gt, pitch, n_variants, nullptr, &d_isx))) return rc;
    if ((rc = ensure(ctx, s, 3, n * 32))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 2 * sizeof(double)))) return rc;
    int32_t *d_c8 = (int32_t *)s->buf[3];
    double *d_chi2 = (double *)s->buf[4], *d_p = d_chi2 + n;
    if ((rc = hpgv_stats_scan_dev(ctx, (const uint8_t *)s->buf[1], n_variants, d_c8, s->stream))) return rc;
    if ((rc = hpgv_stats_hwe_dev(ctx, d_c8, n_variants, d_chi2, d_p, s->stream))) return rc;
    if (sample_missing && ns > 0) {
        if ((rc = ensure(ctx, s, 5, (size_t)ns * sizeof(int32_t)))) return rc;
        HIPCHK(ctx, hipMemsetAsync(s->buf[5], 0, (size_t)ns * sizeof(int32_t), s->stream));
        if ((rc = hpgv_sample_missing_dev(ctx, (const uint8_t *)s->buf[1], n_variants, (int32_t *)s->buf[5], s->stream))) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(counts8, d_c8, n * 32, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(hwe_chi2, d_chi2, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(hwe_p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    std::vector<int32_t> sm;
    if (sample_missing && ns > 0) {
        sm.resize((size_t)ns);
        HIPCHK(ctx, hipMemcpyAsync(sm.data(), s->buf[5], (size_t)ns * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    for (size_t j = 0; j < sm.size(); ++j) sample_missing[j] += sm[j];
    if (n_multi) {
        // variants whose biallelic cells do not cover every called genotype get their full table
        std::vector<int32_t> idx;
        for (size_t i = 0; i < n; ++i) {
            const int32_t *c = counts8 + 8 * i;
            if (ns - c[4] - (c[0] + c[1] + c[2] + c[3]) > 0) idx.push_back((int32_t)i);
        }
        *n_multi = (int)idx.size();
        const int m = (int)idx.size() < cap ? (int)idx.size() : cap;
        if (m > 0) {
            if ((rc = ensure(ctx, s, 6, (size_t)m * sizeof(int32_t)))) return rc;
            if ((rc = ensure(ctx, s, 7, (size_t)m * 256 * sizeof(int32_t)))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(s->buf[6], idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
            if ((rc = hpgv_genotype_table_dev(ctx, (const uint8_t *)s->buf[0], pitch, ns, (const int32_t *)s->buf[6], m,
                                              (int32_t *)s->buf[7], s->stream))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(multi_table, s->buf[7], (size_t)m * 256 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
            HIPCHK(ctx, hipStreamSynchronize(s->stream));
            memcpy(multi_idx, idx.data(), (size_t)m * sizeof(int32_t));
        }
    }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_stats(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, int32_t *counts8,
               double *hwe_chi2, double *hwe_p) {
    return hpgv_stats_ex(ctx, gt, pitch, n_variants, counts8, hwe_chi2, hwe_p, nullptr, nullptr, nullptr, nullptr);
}

int hpgv_stats_groups(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, int32_t *counts8,
                      double *hwe_chi2, double *hwe_p) {
    GROUP_DEAL(ctx, hpgv_stats_groups(m_, gt, pitch, n_variants, counts8, hwe_chi2, hwe_p))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->sgroups.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_groups has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!gt || !counts8))) return fail(ctx, HPGV_ERR_INVALID, "bad stats group arguments");
    if ((hwe_chi2 == nullptr) != (hwe_p == nullptr)) return fail(ctx, HPGV_ERR_INVALID, "hwe_chi2 and hwe_p go together");
    if (pitch < (size_t)ctx->sgroups.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->sgroups.n_samples);
    if (n_variants == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    if (ctx->stats.set && ctx->stats.n_samples == ctx->sgroups.n_samples && stats_all_lds(ctx, false) != 0) {
        // the counters of every phenotype group from one pass over the batch (k_stats_all gathers every group's columns)
        hpgv::BatchArgs B;
        memset(&B, 0, sizeof B);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ctx->stats.n_samples, nullptr, &B))) return rc;
        StatsAllOut O;
        O.group_counts8 = counts8; O.group_hwe_chi2 = hwe_chi2; O.group_hwe_p = hwe_p; O.group_stride = (size_t)n_variants;
        return stats_all_call(ctx, s, B.src, pitch, n_variants, nullptr, O);
    }
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_STATS_GROUPS, ctx->sgroups, gt, pitch, n_variants, nullptr, &d_isx))) return rc;
    const size_t n = (size_t)n_variants, ng = ctx->sg_off.size();
    if ((rc = ensure(ctx, s, 3, ng * n * 32))) return rc;
    if ((rc = ensure(ctx, s, 4, ng * n * 2 * sizeof(double)))) return rc;
    int32_t *d_c8 = (int32_t *)s->buf[3];
    double *d_hw = (double *)s->buf[4];
    for (size_t k = 0; k < ng; ++k) {
        if ((rc = hpgv_stats_scan_group_dev(ctx, (const uint8_t *)s->buf[1], n_variants, (int)k, d_c8 + k * n * 8, s->stream))) return rc;
        if (hwe_chi2 && (rc = hpgv_stats_hwe_dev(ctx, d_c8 + k * n * 8, n_variants, d_hw + k * n, d_hw + (ng + k) * n, s->stream))) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(counts8, d_c8, ng * n * 32, hipMemcpyDeviceToHost, s->stream));
    if (hwe_chi2) {
        HIPCHK(ctx, hipMemcpyAsync(hwe_chi2, d_hw, ng * n * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(ctx, hipMemcpyAsync(hwe_p, d_hw + ng * n, ng * n * 8, hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

int hpgv_epi_dataset(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, uint8_t *out) {
    GROUP_DEAL(ctx, hpgv_epi_dataset(m_, gt, pitch, n_variants, out))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (n_variants < 0 || (n_variants > 0 && (!gt || !out))) return fail(ctx, HPGV_ERR_INVALID, "bad epi arguments");
    if (pitch < (size_t)ctx->assoc.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->assoc.n_samples);
    const size_t nA = (size_t)ctx->nA, nU = (size_t)ctx->nU, width = nA + nU;
    if (n_variants == 0 || width == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_EPI, ctx->assoc, gt, pitch, n_variants, nullptr, &d_isx))) return rc;
    const uint8_t *d = (const uint8_t *)s->buf[1];
    const size_t segA = (size_t)ctx->chunksA * 16, dp = ctx->assoc.pitch;
    // drop the 16-byte pads: cases, then controls, straight into the caller's rows
    if (nA) HIPCHK(ctx, hipMemcpy2DAsync(out, width, d, dp, nA, (size_t)n_variants, hipMemcpyDeviceToHost, s->stream));
    if (nU) HIPCHK(ctx, hipMemcpy2DAsync(out + nA, width, d + segA, dp, nU, (size_t)n_variants, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

int hpgv_mendel(hpgv_ctx *ctx, const uint8_t *gt, size_t pitch, int n_variants, const uint8_t *is_x,
                int32_t *errors, int32_t *child_errors) {
    HPGV_ABI_TRY
    GROUP_DEAL(ctx, hpgv_mendel(m_, gt, pitch, n_variants, is_x, errors, child_errors))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->mendel.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_pedigree has not been called");
    if (n_variants < 0 || (n_variants > 0 && !gt)) return fail(ctx, HPGV_ERR_INVALID, "bad mendel arguments");
    if (pitch < (size_t)ctx->mendel.n_samples) return fail(ctx, HPGV_ERR_INVALID, "pitch %zu < n_samples %d", pitch, ctx->mendel.n_samples);
    if (n_variants == 0 || (!errors && !child_errors)) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    if (ctx->stats.set && ctx->stats.n_samples == ctx->mendel.n_samples && stats_all_lds(ctx, true) != 0) {
        hpgv::BatchArgs B;
        memset(&B, 0, sizeof B);
        if ((rc = batch_sources(ctx, s, gt, pitch, n_variants, ctx->stats.n_samples, is_x, &B))) return rc;
        StatsAllOut O;
        O.mendel_errors = errors; O.child_errors = child_errors;
        std::vector<int32_t> scratch;
        if (!errors) { scratch.resize((size_t)n_variants); O.mendel_errors = scratch.data(); }      // the kernel's switch for the Mendel pass
        return stats_all_call(ctx, s, B.src, pitch, n_variants, B.is_x, O);
    }
    const uint8_t *d_isx = nullptr;
    if ((rc = stage_batch(ctx, s, HPGV_LAYOUT_MENDEL, ctx->mendel, gt, pitch, n_variants, is_x, &d_isx))) return rc;
    const size_t n = (size_t)n_variants, nt = (size_t)ctx->mendel_trios;
    if ((rc = ensure(ctx, s, 3, n * sizeof(int32_t) + 16))) return rc;
    if ((rc = ensure(ctx, s, 5, nt * sizeof(int32_t) + 16))) return rc;
    std::vector<int32_t> ce;
    if (errors) {
        if ((rc = hpgv_mendel_scan_dev(ctx, (const uint8_t *)s->buf[1], n_variants, d_isx, (int32_t *)s->buf[3], s->stream))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(errors, s->buf[3], n * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    if (child_errors && nt) {
        HIPCHK(ctx, hipMemsetAsync(s->buf[5], 0, nt * sizeof(int32_t), s->stream));
        if ((rc = hpgv_mendel_children_dev(ctx, (const uint8_t *)s->buf[1], n_variants, d_isx, (int32_t *)s->buf[5], s->stream))) return rc;
        ce.resize(nt);
        HIPCHK(ctx, hipMemcpyAsync(ce.data(), s->buf[5], nt * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    for (size_t t = 0; t < ce.size(); ++t) child_errors[t] += ce[t];
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

/* ---- text staging ------------------------------------------------------------ */

// (hpgv_inflate_blocks_dev: hpgv_inflate_capi.hip)

int hpgv_tokenize_dev(hpgv_ctx *ctx, const char *d_text, size_t text_bytes, int n_samples, int strict,
                      int max_lines, int *d_n_lines, uint64_t *d_line_off, uint32_t *d_field_off,
                      uint8_t *d_gt, size_t pitch, uint8_t *d_is_x, int32_t *d_status, void *stream) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_samples < 0 || max_lines < 0 || !d_n_lines || (text_bytes > 0 && !d_text) ||
        (max_lines > 0 && !d_gt) || pitch < (size_t)n_samples)
        return fail(ctx, HPGV_ERR_INVALID, "bad tokenize arguments");
    if (text_bytes > ((size_t)1 << 40)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "text buffer too large for one call");
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    // a window of text the bgzip decoder left with its tile records (hpgv_text_alias_tiles): tokenized on the decoder's tile grid,
    // from the start of the tile the window begins in (grid_skip bytes in front of the window: the tail of the line before it)
    hpgv_ctx::TextTiles TT = {nullptr, nullptr, nullptr, 0};
    bool grid = false;
    size_t grid_t0 = 0, grid_skip = 0;
    if (ctx->tokenizer_tiles == 1 && ctx->decode_tiles && text_bytes > 0 && tiles_of_device_text(ctx, d_text, &TT)) {
        const size_t a = (size_t)(d_text - TT.d_base), e = a + text_bytes;
        if ((e - 1) / hpgv::TOK2_TILE < TT.n_tiles) {
            grid = true; grid_t0 = a / hpgv::TOK2_TILE; grid_skip = a - grid_t0 * hpgv::TOK2_TILE;
            d_text -= grid_skip; text_bytes += grid_skip;
        }
    }
    const size_t n_blocks = (text_bytes + hpgv::TOK_TILE - 1) / hpgv::TOK_TILE;
    if (n_blocks > 0x7FFFFFFFu) return fail(ctx, HPGV_ERR_UNSUPPORTED, "text buffer too large for one call");
    hpgv_ctx::TokScratch *ts = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->tok_mu);
        for (auto *t : ctx->tok_scratch) if (t->stream == st) ts = t;
        if (!ts) { ts = new hpgv_ctx::TokScratch(); ts->stream = st; ctx->tok_scratch.push_back(ts); }
    }
    // from here on `ts` is only touched by calls on stream `st`, which the caller does not issue concurrently
    // scratch per tile: the newline counts of the three-sweep form (4 B), or the tile records and tile states of the
    // tile-parallel form (16 B + 16 B)
    const size_t scratch_ints = (n_blocks + 1) * 8 * (hpgv::TOK_TILE / hpgv::TOK2_TILE > 1 ? hpgv::TOK_TILE / hpgv::TOK2_TILE : 1) + 64;            // (+ the one-sweep form's 64-byte head and the records of a text of a few bytes)
    if (ts->blocks_cap < scratch_ints) {
        if (ts->d_blocks) { HIPCHK(ctx, hipStreamSynchronize(st)); (void)hipFree(ts->d_blocks); ts->d_blocks = nullptr; ts->blocks_cap = 0; }
        HIPCHK(ctx, hipMalloc(&ts->d_blocks, scratch_ints * sizeof(int)));
        ts->blocks_cap = scratch_ints;
    }
    unsigned long long *line_off = (unsigned long long *)d_line_off;
    if (!line_off) {                                   // caller does not want the offsets: use scratch
        if (ts->line_cap < (size_t)max_lines + 2) {
            if (ts->d_line_off) { HIPCHK(ctx, hipStreamSynchronize(st)); (void)hipFree(ts->d_line_off); ts->d_line_off = nullptr; ts->line_cap = 0; }
            HIPCHK(ctx, hipMalloc(&ts->d_line_off, ((size_t)max_lines + 2) * sizeof(unsigned long long)));
            ts->line_cap = (size_t)max_lines + 2;
        }
        line_off = ts->d_line_off;
    }
    if (ctx->tokenizer_tiles) {
        // two sweeps of the text: tile records, tile states, then one workgroup per tile parses (hpgv_text2_kernels.h)
        const size_t n_tiles = (text_bytes + hpgv::TOK2_TILE - 1) / hpgv::TOK2_TILE;      // 2 KiB tiles
        hpgv::TokAgg *agg = (hpgv::TokAgg *)ts->d_blocks;
        hpgv::TokPre *pre = (hpgv::TokPre *)(agg + n_tiles + 1);
        const int n_groups = (int)((n_tiles + hpgv::TOK_SCAN_THREADS - 1) / hpgv::TOK_SCAN_THREADS);
        // the groups' totals and the per-line "parse again" flags live behind the line offsets' scratch
        const size_t extra = ((size_t)n_groups + 2) * sizeof(hpgv::TokState) + ((size_t)max_lines + 2) * sizeof(int);
        if (ts->extra_cap < extra) {
            if (ts->d_extra) { HIPCHK(ctx, hipStreamSynchronize(st)); (void)hipFree(ts->d_extra); ts->d_extra = nullptr; ts->extra_cap = 0; }
            HIPCHK(ctx, hipMalloc(&ts->d_extra, extra + extra / 4));
            ts->extra_cap = extra + extra / 4;
        }
        hpgv::TokState *gtot = (hpgv::TokState *)ts->d_extra;
        int *redo = (int *)(gtot + n_groups + 2), *redo_n = redo + max_lines + 1;      // the list of lines to parse again, its length
        const unsigned redo_grid = (unsigned)(max_lines < 1024 ? max_lines : 1024);
#ifdef HPGV_ABLATION
        if (ctx->tokenizer_tiles >= 2 && n_tiles > 0 && max_lines > 0) {
            // ONE sweep: count, scan and parse in one kernel, the segments' start states by look-back (k_tok_parse3).  The
            // records, the ticket and the error flag share the tile scratch (zeroed per call: 16 bytes per 32 KiB of text).
            const size_t n_seg = (text_bytes + hpgv::TOK3_SEG - 1) / hpgv::TOK3_SEG;
            unsigned *tk = (unsigned *)ts->d_blocks;
            int *err = (int *)ts->d_blocks + 1;
            redo_n = (int *)ts->d_blocks + 2;                          // (zeroed with the records)
            const size_t n_sup = (n_seg + hpgv::TOK3_SUPER - 1) / hpgv::TOK3_SUPER;
            hpgv::TokRec *rec = (hpgv::TokRec *)((char *)ts->d_blocks + 64), *sup = rec + n_seg;
            HIPCHK(ctx, hipMemsetAsync(ts->d_blocks, 0, 64 + (n_seg + n_sup) * sizeof(hpgv::TokRec), st));
            hipLaunchKernelGGL(hpgv::k_tok_parse3, dim3((unsigned)n_seg), dim3(256), 0, st, d_text, text_bytes, rec, sup, tk, err, d_n_lines,
                               max_lines, n_samples, strict, d_gt, pitch, d_is_x, line_off, d_field_off, d_status, redo, redo_n);
            hipLaunchKernelGGL(hpgv::k_tok_finish, dim3(1), dim3(1), 0, st, (const int *)err, d_n_lines);
            hipLaunchKernelGGL(hpgv::k_tok_parse_listed, dim3(redo_grid), dim3(256), 0, st, d_text, line_off,
                               (const int *)d_n_lines, max_lines, n_samples, strict, d_gt, pitch, d_is_x, d_field_off, d_status, (const int *)redo, (const int *)redo_n);
            HIPCHK(ctx, hipGetLastError());
            return HPGV_OK;
        }
#endif
        if (grid && n_tiles > 0) {
            // the decoder's records serve every tile but the window's last, which is counted again up to the window's end
            // (and the bytes in front of the window, for the number of lines that end there)
            const size_t lt = n_tiles - 1;
            hipLaunchKernelGGL(hpgv::k_tok_count2, dim3(1), dim3(256), 0, st, d_text + lt * hpgv::TOK2_TILE, text_bytes - lt * hpgv::TOK2_TILE, 1, agg);
            if (grid_skip) hipLaunchKernelGGL(hpgv::k_tok_count2, dim3(1), dim3(256), 0, st, d_text, grid_skip, 1, agg + 1);
            hipLaunchKernelGGL(hpgv::k_tok_scan2a_grid, dim3((unsigned)n_groups), dim3(hpgv::TOK_SCAN_THREADS), 0, st, (const hpgv::TokAgg2 *)TT.d_tiles, (long)grid_t0,
                               (const hpgv::TokAgg *)agg, d_text, text_bytes, (int)n_tiles, pre, gtot);
        } else if (n_tiles > 0) {
            hipLaunchKernelGGL(hpgv::k_tok_count2, dim3((unsigned)((n_tiles + hpgv::TOK2_COUNT_TILES - 1) / hpgv::TOK2_COUNT_TILES)), dim3(256), 0, st, d_text, text_bytes, (int)n_tiles, agg);
            hipLaunchKernelGGL(hpgv::k_tok_scan2a, dim3((unsigned)n_groups), dim3(hpgv::TOK_SCAN_THREADS), 0, st, (const hpgv::TokAgg *)agg, (int)n_tiles, pre, gtot);
        }
        hipLaunchKernelGGL(hpgv::k_tok_scan2b, dim3((unsigned)(n_groups > 0 ? n_groups : 1)), dim3(hpgv::TOK_SCAN_THREADS), 0, st, pre, (int)n_tiles, gtot, n_groups,
                           d_text, text_bytes, d_n_lines, line_off, max_lines, redo_n, grid_skip ? (const hpgv::TokAgg *)(agg + 1) : (const hpgv::TokAgg *)nullptr);
        if (n_tiles > 0 && max_lines > 0) {
            hipLaunchKernelGGL(hpgv::k_tok_parse2, dim3((unsigned)n_tiles), dim3(hpgv::TOK2_THREADS), 0, st, d_text, text_bytes, (const hpgv::TokPre *)pre,
                               max_lines, n_samples, strict, d_gt, pitch, d_is_x, line_off, d_field_off, d_status, redo, redo_n, (int)grid_skip);
            // the lines whose FORMAT does not begin with GT (listed by the thread that read it): once more, line by line
            hipLaunchKernelGGL(hpgv::k_tok_parse_listed, dim3(redo_grid), dim3(256), 0, st, d_text, line_off,
                               (const int *)d_n_lines, max_lines, n_samples, strict, d_gt, pitch, d_is_x, d_field_off, d_status, (const int *)redo, (const int *)redo_n);
        }
        if (grid_skip)                                               // positions counted from the first tile's start: back to the window's
            hipLaunchKernelGGL(hpgv::k_tok_grid_finish, dim3((unsigned)((max_lines + 256) / 256)), dim3(256), 0, st, line_off, (const int *)d_n_lines, max_lines, (unsigned)grid_skip);
        HIPCHK(ctx, hipGetLastError());
        return HPGV_OK;
    }
#ifndef HPGV_ABLATION
    return fail(ctx, HPGV_ERR_UNSUPPORTED, "the line-by-line tokenizer is an ablation build's");
#else
    if (n_blocks > 0)
        hipLaunchKernelGGL(hpgv::k_tok_count, dim3((unsigned)n_blocks), dim3(256), 0, st, d_text, text_bytes, ts->d_blocks);
    hipLaunchKernelGGL(hpgv::k_tok_scan, dim3(1), dim3(hpgv::TOK_SCAN_THREADS), 0, st, ts->d_blocks, (int)n_blocks, d_text, text_bytes,
                       d_n_lines, line_off, max_lines);
    if (n_blocks > 0)
        hipLaunchKernelGGL(hpgv::k_tok_mark, dim3((unsigned)n_blocks), dim3(256), 0, st, d_text, text_bytes,
                           (const int *)ts->d_blocks, line_off, max_lines);
    else
        HIPCHK(ctx, hipMemsetAsync(line_off, 0, sizeof(unsigned long long), st));
    if (max_lines > 0)
        hipLaunchKernelGGL(hpgv::k_tok_parse, dim3((unsigned)max_lines), dim3(256), 0, st, d_text, line_off,
                           (const int *)d_n_lines, max_lines, n_samples, strict, d_gt, pitch, d_is_x, d_field_off, d_status);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
#endif
    HPGV_ABI_CATCH(ctx)
}

int hpgv_tokenize(hpgv_ctx *ctx, const char *text, size_t text_bytes, int n_samples, int strict, int max_lines,
                  int *n_lines, uint64_t *line_off, uint32_t *field_off, uint8_t *gt, size_t pitch,
                  uint8_t *is_x, int32_t *status) {
    GROUP_DEAL(ctx, hpgv_tokenize(m_, text, text_bytes, n_samples, strict, max_lines, n_lines, line_off, field_off, gt, pitch, is_x, status))
    if (!ctx) return HPGV_ERR_INVALID;
    if (!n_lines || n_samples < 0 || max_lines < 0 || (text_bytes > 0 && !text) || (max_lines > 0 && !gt) ||
        pitch < (size_t)n_samples)
        return fail(ctx, HPGV_ERR_INVALID, "bad tokenize arguments");
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    const size_t ml = (size_t)max_lines;
    if ((rc = ensure(ctx, s, 0, text_bytes + 16))) return rc;
    if ((rc = ensure(ctx, s, 1, ml * pitch + 16))) return rc;
    if ((rc = ensure(ctx, s, 2, ml + 16))) return rc;
    if ((rc = ensure(ctx, s, 3, (ml + 2) * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, s, 4, ml * 10 * sizeof(uint32_t) + 16))) return rc;
    if ((rc = ensure(ctx, s, 5, ml * sizeof(int32_t) + 16))) return rc;
    if ((rc = ensure(ctx, s, 6, 16))) return rc;
    if (text_bytes) HIPCHK(ctx, hipMemcpyAsync(s->buf[0], text, text_bytes, hipMemcpyHostToDevice, s->stream));
    if ((rc = hpgv_tokenize_dev(ctx, (const char *)s->buf[0], text_bytes, n_samples, strict, max_lines, (int *)s->buf[6],
                                (uint64_t *)s->buf[3], (uint32_t *)s->buf[4], (uint8_t *)s->buf[1], pitch,
                                (uint8_t *)s->buf[2], (int32_t *)s->buf[5], s->stream))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(n_lines, s->buf[6], sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    if (*n_lines < 0) {                                             // the one-sweep tokenizer gave up a look-back: the two-sweep kernels from now on
        ctx->tokenizer_tiles = 1;
        if ((rc = hpgv_tokenize_dev(ctx, (const char *)s->buf[0], text_bytes, n_samples, strict, max_lines, (int *)s->buf[6],
                                    (uint64_t *)s->buf[3], (uint32_t *)s->buf[4], (uint8_t *)s->buf[1], pitch,
                                    (uint8_t *)s->buf[2], (int32_t *)s->buf[5], s->stream))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(n_lines, s->buf[6], sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
    }
    const size_t nl = (size_t)(*n_lines < max_lines ? *n_lines : max_lines);
    if (nl) {
        HIPCHK(ctx, hipMemcpyAsync(gt, s->buf[1], nl * pitch, hipMemcpyDeviceToHost, s->stream));
        if (is_x) HIPCHK(ctx, hipMemcpyAsync(is_x, s->buf[2], nl, hipMemcpyDeviceToHost, s->stream));
        if (field_off) HIPCHK(ctx, hipMemcpyAsync(field_off, s->buf[4], nl * 10 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        if (status) HIPCHK(ctx, hipMemcpyAsync(status, s->buf[5], nl * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    if (line_off) HIPCHK(ctx, hipMemcpyAsync(line_off, s->buf[3], (nl + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

// shared front half of the *_text entry points: text -> device, tokenize, lay out.
// slot buffers: 0 text, 1 laid-out gt, 2 is_x, 3 tallies, 4 doubles, 5 SoA ints / status,
// 6 {n_lines, line_off..., field_off...}, 7 raw gt (VCF order)
static int text_front(hpgv_ctx *ctx, Slot *s, int which, const Layout &L, const char *text, size_t text_bytes,
                      int max_lines, int *n_lines, uint64_t *line_off, uint32_t *field_off, int32_t *status,
                      int *nl_out, bool final_layout = true) {
    int rc;
    const size_t ml = (size_t)max_lines;
    const size_t raw_pitch = (size_t)(L.n_samples > 0 ? (L.n_samples + 15) / 16 * 16 : 16);
    const size_t off_lines = 16, off_fields = off_lines + (ml + 2) * sizeof(uint64_t);
    if ((rc = ensure(ctx, s, 0, text_bytes + 16))) return rc;
    if ((rc = ensure(ctx, s, 7, ml * raw_pitch + 16))) return rc;
    if ((rc = ensure(ctx, s, 1, ml * L.pitch + 16))) return rc;
    if ((rc = ensure(ctx, s, 2, ml + 16))) return rc;
    if ((rc = ensure(ctx, s, 5, ml * 4 * sizeof(int32_t) + 16))) return rc;
    if ((rc = ensure(ctx, s, 6, off_fields + ml * 10 * sizeof(uint32_t) + 16))) return rc;
    char *meta = (char *)s->buf[6];
    const char *d_src = text_on_device(ctx, text);                  // hpgv_text_alias: the text is on the device already
    const bool aliased = d_src != nullptr;
    if (!d_src) {
        if (text_bytes) HIPCHK(ctx, hipMemcpyAsync(s->buf[0], text, text_bytes, hipMemcpyHostToDevice, s->stream));
        d_src = (const char *)s->buf[0];
    }
    // the raw matrix keeps half-called genotypes ("./1"): the record filters count alleles as the stats tool does;
    // the strict layouts (assoc, tdt, epi) turn every not fully called genotype into "missing" on their way in
    if ((rc = hpgv_tokenize_dev(ctx, d_src, text_bytes, L.n_samples, 0,
                                max_lines, (int *)meta, (uint64_t *)(meta + off_lines), (uint32_t *)(meta + off_fields),
                                (uint8_t *)s->buf[7], raw_pitch, (uint8_t *)s->buf[2], (int32_t *)s->buf[5], s->stream))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(n_lines, meta, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    if (*n_lines < 0) {                                             // the one-sweep tokenizer gave up a look-back: the two-sweep kernels from now on
        ctx->tokenizer_tiles = 1;
        if ((rc = hpgv_tokenize_dev(ctx, d_src, text_bytes, L.n_samples, 0,
                                    max_lines, (int *)meta, (uint64_t *)(meta + off_lines), (uint32_t *)(meta + off_fields),
                                    (uint8_t *)s->buf[7], raw_pitch, (uint8_t *)s->buf[2], (int32_t *)s->buf[5], s->stream))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(n_lines, meta, sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
    }
    const int nl = *n_lines < max_lines ? *n_lines : max_lines;
    *nl_out = nl;
    if (nl == 0) return HPGV_OK;
    if (status) HIPCHK(ctx, hipMemcpyAsync(status, s->buf[5], (size_t)nl * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    if (aliased) {
        // the text is on the device only: the caller's host buffer gets the line heads (CHROM .. FORMAT, all it reads for its
        // result records) and line_off refers to them
        const int hb = (nl + 1023) / 1024;                          // workgroups of 1024 lines
        const size_t off_heads = (((size_t)nl + 2 + (size_t)hb + 1) * sizeof(uint64_t) + 15) / 16 * 16;
        if ((rc = ensure(ctx, s, 0, text_bytes + off_heads + 64))) return rc;
        unsigned long long *d_head_off = (unsigned long long *)s->buf[0], *d_block = d_head_off + (size_t)nl + 2;
        char *d_heads = (char *)s->buf[0] + off_heads;
        hipLaunchKernelGGL(hpgv::k_head_sums, dim3((unsigned)hb), dim3(1024), 0, s->stream, (const unsigned long long *)(meta + off_lines),
                           (const uint32_t *)(meta + off_fields), nl, d_block);
        hipLaunchKernelGGL(hpgv::k_head_bases, dim3(1), dim3(1024), 0, s->stream, d_block, hb);
        hipLaunchKernelGGL(hpgv::k_head_offsets, dim3((unsigned)hb), dim3(1024), 0, s->stream, (const unsigned long long *)(meta + off_lines),
                           (const uint32_t *)(meta + off_fields), nl, (const unsigned long long *)d_block, d_head_off);
        hipLaunchKernelGGL(hpgv::k_copy_heads, dim3((unsigned)nl), dim3(64), 0, s->stream, d_src, (const unsigned long long *)(meta + off_lines),
                           (const unsigned long long *)d_head_off, nl, d_heads);
        HIPCHK(ctx, hipGetLastError());
        unsigned long long total_heads = 0;
        HIPCHK(ctx, hipMemcpyAsync(&total_heads, d_head_off + nl, sizeof total_heads, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        if (total_heads > text_bytes) return fail(ctx, HPGV_ERR_HIP, "line heads longer than the text");
        if (total_heads) HIPCHK(ctx, hipMemcpyAsync(const_cast<char *>(text), d_heads, (size_t)total_heads, hipMemcpyDeviceToHost, s->stream));
        if (line_off) HIPCHK(ctx, hipMemcpyAsync(line_off, d_head_off, ((size_t)nl + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    } else if (line_off) HIPCHK(ctx, hipMemcpyAsync(line_off, meta + off_lines, ((size_t)nl + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    if (field_off) HIPCHK(ctx, hipMemcpyAsync(field_off, meta + off_fields, (size_t)nl * 10 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    // ---- record filters (--maf, --missing, --mendel: shared_options.c:44-46,101-115), from the same matrix ----
    const bool f_counts = ctx->filt_min_maf >= 0.0 || ctx->filt_max_missing >= 0.0, f_mendel = ctx->filt_max_mendel >= 0;
    if (status && (f_counts || f_mendel)) {
        const size_t n = (size_t)nl;
        std::vector<uint8_t> keep;
        std::vector<int32_t> merr;
        if (f_counts) {
            if (!ctx->stats.set || ctx->stats.n_samples != L.n_samples)
                return fail(ctx, HPGV_ERR_STATE, "the count filters need hpgv_set_stats_cohort(%d)", L.n_samples);
            if ((rc = ensure(ctx, s, 1, n * std::max(ctx->stats.pitch, L.pitch) + 16))) return rc;
            if ((rc = ensure(ctx, s, 3, n * 33 + 64))) return rc;
            if ((rc = hpgv_layout_dev(ctx, HPGV_LAYOUT_STATS, (const uint8_t *)s->buf[7], raw_pitch, nl, (uint8_t *)s->buf[1], s->stream))) return rc;
            if ((rc = hpgv_stats_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, (int32_t *)s->buf[3], s->stream))) return rc;
            uint8_t *d_keep = (uint8_t *)s->buf[3] + n * 32;
            if ((rc = hpgv_stats_filter_dev(ctx, (const int32_t *)s->buf[3], nl, ctx->filt_min_maf, -1.0, ctx->filt_max_missing, d_keep, s->stream))) return rc;
            keep.resize(n);
            HIPCHK(ctx, hipMemcpyAsync(keep.data(), d_keep, n, hipMemcpyDeviceToHost, s->stream));
        }
        if (f_mendel) {
            if (!ctx->mendel.set || ctx->mendel.n_samples != L.n_samples)
                return fail(ctx, HPGV_ERR_STATE, "the Mendelian error filter needs hpgv_set_pedigree over %d columns", L.n_samples);
            if ((rc = ensure(ctx, s, 1, n * std::max(ctx->mendel.pitch, L.pitch) + 16))) return rc;
            if ((rc = ensure(ctx, s, 4, n * sizeof(int32_t) + 64))) return rc;
            if ((rc = hpgv_layout_dev(ctx, HPGV_LAYOUT_MENDEL, (const uint8_t *)s->buf[7], raw_pitch, nl, (uint8_t *)s->buf[1], s->stream))) return rc;
            if ((rc = hpgv_mendel_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, (const uint8_t *)s->buf[2], (int32_t *)s->buf[4], s->stream))) return rc;
            merr.resize(n);
            HIPCHK(ctx, hipMemcpyAsync(merr.data(), s->buf[4], n * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        for (size_t i = 0; i < n; ++i) {
            const bool out = (f_counts && !keep[i]) || (f_mendel && (long)merr[i] > ctx->filt_max_mendel);
            if (out) status[i] |= HPGV_LINE_FILTERED;
        }
    }
    if (!final_layout) return HPGV_OK;                       // the caller's one-pass kernel reads the raw matrix itself
    return hpgv_layout_dev(ctx, which, (const uint8_t *)s->buf[7], raw_pitch, nl, (uint8_t *)s->buf[1], s->stream);
}

int hpgv_set_text_filters(hpgv_ctx *ctx, double min_maf, double max_missing, long max_mendel_errors) {
    GROUP_ALL(ctx, hpgv_set_text_filters(m_, min_maf, max_missing, max_mendel_errors))
    if (!ctx) return HPGV_ERR_INVALID;
    if (min_maf > 0.5 || max_missing > 1.0) return fail(ctx, HPGV_ERR_INVALID, "min_maf is at most 0.5, max_missing at most 1");
    ctx->filt_min_maf = min_maf; ctx->filt_max_missing = max_missing; ctx->filt_max_mendel = max_mendel_errors;
    return HPGV_OK;
}

int hpgv_assoc_text(hpgv_ctx *ctx, int task, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                    uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *A1, int32_t *A2,
                    int32_t *U1, int32_t *U2, double *odds, double *chisq, double *p) {
    if (is_group(ctx)) {
        if (hpgv_ctx *m_ = alias_owner(ctx, text)) return hpgv_assoc_text(m_, task, text, text_bytes, max_lines, n_lines, line_off, field_off, status, A1, A2, U1, U2, odds, chisq, p);
        Dealt d_(ctx); hpgv_ctx *m_ = d_.m; return hpgv_assoc_text(m_, task, text, text_bytes, max_lines, n_lines, line_off, field_off, status, A1, A2, U1, U2, odds, chisq, p);
    }
    if (!ctx) return HPGV_ERR_INVALID;
    if (task != HPGV_TASK_CHISQ && task != HPGV_TASK_FISHER) return fail(ctx, HPGV_ERR_INVALID, "task must be CHISQ or FISHER");
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (!n_lines || max_lines < 0 || (text_bytes > 0 && !text) ||
        (max_lines > 0 && (!A1 || !A2 || !U1 || !U2 || !odds || !p || (task == HPGV_TASK_CHISQ && !chisq))))
        return fail(ctx, HPGV_ERR_INVALID, "bad assoc_text arguments");
    *n_lines = 0;
    if (max_lines == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    int nl = 0;
    const bool fused = batch_fused_ok(ctx, ctx->assoc.n_samples);
    if (fused && task == HPGV_TASK_FISHER) {
        if (!ctx->d_lf) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_logfact has not been called");
        if (ctx->n_lf < (size_t)2 * (ctx->nA + ctx->nU) + 1)
            return fail(ctx, HPGV_ERR_STATE, "log-factorial table has %zu entries, need %d", ctx->n_lf, 2 * (ctx->nA + ctx->nU) + 1);
    }
    if ((rc = text_front(ctx, s, HPGV_LAYOUT_ASSOC, ctx->assoc, text, text_bytes, max_lines, n_lines, line_off, field_off, status, &nl, !fused))) return rc;
    if (nl == 0) { HIPCHK(ctx, hipStreamSynchronize(s->stream)); return HPGV_OK; }
    const size_t n = (size_t)nl;
    if (fused && ctx->assoc_rows) {
        // the raw matrix read once by threads that own columns (k_assoc_rows), then the scans' own statistics kernels
        const int ns = ctx->assoc.n_samples;
        if ((rc = ensure(ctx, s, 3, n * 16))) return rc;
        if ((rc = ensure(ctx, s, 4, n * 3 * sizeof(double)))) return rc;
        if ((rc = ensure_result_block(ctx, s, n * 40 + 64))) return rc;
        int32_t *d_counts = (int32_t *)s->buf[3];
        double *d_odds = (double *)s->buf[4], *d_chisq = d_odds + n, *d_p = d_odds + 2 * n;
        if (hpgv_launch_assoc_rows(ctx, (const uint8_t *)s->buf[7], (size_t)(ns > 0 ? (ns + 15) / 16 * 16 : 16), nl, (const uint8_t *)s->buf[2], d_counts, s->stream) == 0) {
            HIPCHK(ctx, hipGetLastError());
            if (task == HPGV_TASK_CHISQ) rc = hpgv_assoc_chisq_dev(ctx, d_counts, nl, d_odds, d_chisq, d_p, s->stream);
            else rc = hpgv_assoc_fisher_dev(ctx, d_counts, nl, d_odds, d_p, s->stream);
            if (rc) { (void)hipStreamSynchronize(s->stream); return rc; }
            // results come back through the slot's page-locked block (one copy each at the bus rate), then into the caller's arrays
            char *h = (char *)s->h_res;
            HIPCHK(ctx, hipMemcpyAsync(h, d_counts, n * 16, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(ctx, hipMemcpyAsync(h + n * 16, d_odds, n * 24, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(ctx, hipStreamSynchronize(s->stream));
            const int32_t *c4 = (const int32_t *)h;
            const double *dd = (const double *)(h + n * 16);
            for (size_t i = 0; i < n; ++i) { A1[i] = c4[4 * i]; A2[i] = c4[4 * i + 1]; U1[i] = c4[4 * i + 2]; U2[i] = c4[4 * i + 3]; }
            memcpy(odds, dd, n * 8);
            if (task == HPGV_TASK_CHISQ) memcpy(chisq, dd + n, n * 8);
            memcpy(p, dd + 2 * n, n * 8);
            return HPGV_OK;
        }
    }
    if (fused) {
        // the tokenizer's raw matrix is read ONCE: layout in registers, counts, statistics, packed records (hpgv_batch_kernels.h)
        const int ns = ctx->assoc.n_samples;
        hpgv::BatchArgs A;
        memset(&A, 0, sizeof A);
        A.src = (const uint8_t *)s->buf[7]; A.src_pitch = (size_t)(ns > 0 ? (ns + 15) / 16 * 16 : 16);
        A.n_variants = nl; A.n_samples = ns; A.is_x = (const uint8_t *)s->buf[2];
        if ((rc = ensure_result_block(ctx, s, n * sizeof(hpgv::BatchAssocRec)))) return rc;
        A.col_of_pos = ctx->assoc.d_col_of_pos; A.chunks = ctx->assoc.chunks; A.chunksA = ctx->chunksA;
        A.lf = ctx->d_lf; A.rel_cut = pow(10.0, -(double)ctx->fisher_cut_exp);
        A.out = s->d_res;
        if (task == HPGV_TASK_CHISQ) rc = launch_batch<hpgv::BATCH_CHISQ>(ctx, s, A);
        else rc = launch_batch<hpgv::BATCH_FISHER>(ctx, s, A);
        if (rc) { (void)hipStreamSynchronize(s->stream); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        const hpgv::BatchAssocRec *r = (const hpgv::BatchAssocRec *)s->h_res;
        for (size_t i = 0; i < n; ++i) {
            A1[i] = r[i].A1; A2[i] = r[i].A2; U1[i] = r[i].U1; U2[i] = r[i].U2;
            odds[i] = r[i].odds; p[i] = r[i].p;
        }
        if (task == HPGV_TASK_CHISQ) for (size_t i = 0; i < n; ++i) chisq[i] = r[i].chisq;
        return HPGV_OK;
    }
    if ((rc = ensure(ctx, s, 3, n * 16))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 3 * sizeof(double)))) return rc;
    int32_t *d_counts = (int32_t *)s->buf[3];
    double *d_odds = (double *)s->buf[4], *d_chisq = d_odds + n, *d_p = d_odds + 2 * n;
    int32_t *d_soa = (int32_t *)s->buf[5];        // status has been copied out (same stream, ordered)
    if ((rc = hpgv_assoc_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, (const uint8_t *)s->buf[2], d_counts, s->stream))) return rc;
    if (task == HPGV_TASK_CHISQ) rc = hpgv_assoc_chisq_dev(ctx, d_counts, nl, d_odds, d_chisq, d_p, s->stream);
    else rc = hpgv_assoc_fisher_dev(ctx, d_counts, nl, d_odds, d_p, s->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(hpgv::k_counts_to_soa, dim3((nl + 255) / 256), dim3(256), 0, s->stream,
                       (const int4 *)d_counts, nl, d_soa, d_soa + n, d_soa + 2 * n, d_soa + 3 * n);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(A1, d_soa, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(A2, d_soa + n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(U1, d_soa + 2 * n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(U2, d_soa + 3 * n, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(odds, d_odds, n * 8, hipMemcpyDeviceToHost, s->stream));
    if (task == HPGV_TASK_CHISQ) HIPCHK(ctx, hipMemcpyAsync(chisq, d_chisq, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

int hpgv_tdt_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                  uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *t1, int32_t *t2,
                  double *odds, double *chisq, double *p) {
    HPGV_ABI_TRY
    if (is_group(ctx)) {
        if (hpgv_ctx *m_ = alias_owner(ctx, text)) return hpgv_tdt_text(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, t1, t2, odds, chisq, p);
        Dealt d_(ctx); hpgv_ctx *m_ = d_.m; return hpgv_tdt_text(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, t1, t2, odds, chisq, p);
    }
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->tdt.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_families has not been called");
    if (!n_lines || max_lines < 0 || (text_bytes > 0 && !text) || (max_lines > 0 && (!t1 || !t2 || !odds || !chisq || !p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad tdt_text arguments");
    *n_lines = 0;
    if (max_lines == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    int nl = 0;
    const bool fused = batch_fused_ok(ctx, ctx->tdt.n_samples);
    if ((rc = text_front(ctx, s, HPGV_LAYOUT_TDT, ctx->tdt, text, text_bytes, max_lines, n_lines, line_off, field_off, status, &nl, !fused))) return rc;
    if (nl == 0) { HIPCHK(ctx, hipStreamSynchronize(s->stream)); return HPGV_OK; }
    const size_t n = (size_t)nl;
    if (fused) {
        const int ns = ctx->tdt.n_samples;
        hpgv::BatchArgs A;
        memset(&A, 0, sizeof A);
        A.src = (const uint8_t *)s->buf[7]; A.src_pitch = (size_t)(ns > 0 ? (ns + 15) / 16 * 16 : 16);
        A.n_variants = nl; A.n_samples = ns; A.is_x = (const uint8_t *)s->buf[2];
        if ((rc = ensure_result_block(ctx, s, n * sizeof(hpgv::BatchTdtRec)))) return rc;
        const hpgv::TdtPlan &P = ctx->tdt_plan;
        A.col_of_pos = ctx->tdt.d_col_of_pos; A.chunks = ctx->tdt.chunks;
        A.pchunks = P.pchunks; A.p16 = P.p16; A.n_slow = P.n_slow_families; A.slow_base = P.slow_base; A.luts = P.luts;
        A.male_plane = P.d_male_plane; A.slow_off = P.d_slow_off; A.slow_male = P.d_slow_male;
        A.out = s->d_res;
        if ((rc = launch_batch<hpgv::BATCH_TDT>(ctx, s, A))) { (void)hipStreamSynchronize(s->stream); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(s->stream));
        const hpgv::BatchTdtRec *r = (const hpgv::BatchTdtRec *)s->h_res;
        for (size_t i = 0; i < n; ++i) { t1[i] = r[i].t1; t2[i] = r[i].t2; odds[i] = r[i].odds; chisq[i] = r[i].chisq; p[i] = r[i].p; }
        return HPGV_OK;
    }
    if ((rc = ensure(ctx, s, 3, n * 8))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 3 * sizeof(double)))) return rc;
    int32_t *d_tu = (int32_t *)s->buf[3];
    double *d_odds = (double *)s->buf[4], *d_chisq = d_odds + n, *d_p = d_odds + 2 * n;
    if ((rc = hpgv_tdt_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, (const uint8_t *)s->buf[2], d_tu, s->stream))) return rc;
    if ((rc = hpgv_tdt_stats_dev(ctx, d_tu, nl, d_odds, d_chisq, d_p, s->stream))) return rc;
    std::vector<int32_t> tu(2 * n);
    HIPCHK(ctx, hipMemcpyAsync(tu.data(), d_tu, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(odds, d_odds, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(chisq, d_chisq, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    for (size_t i = 0; i < n; ++i) { t1[i] = tu[2 * i]; t2[i] = tu[2 * i + 1]; }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_stats_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                    uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *counts8, double *hwe_chi2,
                    double *hwe_p, int32_t *sample_missing, int32_t *multi_idx, int32_t *multi_table, int *n_multi,
                    int32_t *mendel_errors, int32_t *child_errors) {
    return hpgv_stats_text_groups(ctx, text, text_bytes, max_lines, n_lines, line_off, field_off, status, counts8, hwe_chi2, hwe_p,
                                  sample_missing, multi_idx, multi_table, n_multi, mendel_errors, child_errors, nullptr, nullptr, nullptr);
}

int hpgv_stats_text_groups(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                           uint64_t *line_off, uint32_t *field_off, int32_t *status, int32_t *counts8, double *hwe_chi2,
                           double *hwe_p, int32_t *sample_missing, int32_t *multi_idx, int32_t *multi_table, int *n_multi,
                           int32_t *mendel_errors, int32_t *child_errors, int32_t *group_counts8, double *group_hwe_chi2,
                           double *group_hwe_p) {
    HPGV_ABI_TRY
    if (is_group(ctx)) {
        if (hpgv_ctx *m_ = alias_owner(ctx, text)) return hpgv_stats_text_groups(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, counts8, hwe_chi2, hwe_p, sample_missing, multi_idx, multi_table, n_multi, mendel_errors, child_errors, group_counts8, group_hwe_chi2, group_hwe_p);
        Dealt d_(ctx); hpgv_ctx *m_ = d_.m; return hpgv_stats_text_groups(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, counts8, hwe_chi2, hwe_p, sample_missing, multi_idx, multi_table, n_multi, mendel_errors, child_errors, group_counts8, group_hwe_chi2, group_hwe_p);
    }
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->stats.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_stats_cohort has not been called");
    if (!n_lines || max_lines < 0 || (text_bytes > 0 && !text) || (max_lines > 0 && (!counts8 || !hwe_chi2 || !hwe_p)))
        return fail(ctx, HPGV_ERR_INVALID, "bad stats_text arguments");
    if (n_multi && *n_multi > 0 && (!multi_idx || !multi_table)) return fail(ctx, HPGV_ERR_INVALID, "multi-allelic outputs are NULL");
    const bool want_mendel = mendel_errors || child_errors;
    if (want_mendel && (!ctx->mendel.set || ctx->mendel.n_samples != ctx->stats.n_samples))
        return fail(ctx, HPGV_ERR_STATE, "Mendelian errors need hpgv_set_pedigree over the same %d columns", ctx->stats.n_samples);
    if (group_counts8 && (!ctx->sgroups.set || ctx->sgroups.n_samples != ctx->stats.n_samples))
        return fail(ctx, HPGV_ERR_STATE, "per-group counters need hpgv_set_stats_groups over the same %d columns", ctx->stats.n_samples);
    if ((group_hwe_chi2 == nullptr) != (group_hwe_p == nullptr) || (group_hwe_chi2 && !group_counts8))
        return fail(ctx, HPGV_ERR_INVALID, "group_hwe_chi2 and group_hwe_p go together, with group_counts8");
    const int cap = n_multi ? *n_multi : 0;
    if (n_multi) *n_multi = 0;
    *n_lines = 0;
    if (max_lines == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    int nl = 0;
    const bool fused = stats_all_lds(ctx, want_mendel) != 0;      // LDS: the row window, a byte counter per column and per trio
    if ((rc = text_front(ctx, s, HPGV_LAYOUT_STATS, ctx->stats, text, text_bytes, max_lines, n_lines, line_off, field_off, status, &nl, !fused))) return rc;
    if (nl == 0) { HIPCHK(ctx, hipStreamSynchronize(s->stream)); return HPGV_OK; }
    const size_t n = (size_t)nl;
    const int ns = ctx->stats.n_samples;
    const size_t raw_pitch = (size_t)(ns > 0 ? (ns + 15) / 16 * 16 : 16);
    if (fused) {
        // ONE pass over the tokenizer's raw matrix gives every statistic of the batch (k_stats_all): the genotype bytes
        // are read from HBM once after tokenizing
        StatsAllOut O;
        O.counts8 = counts8; O.hwe_chi2 = hwe_chi2; O.hwe_p = hwe_p; O.sample_missing = sample_missing;
        O.mendel_errors = mendel_errors; O.child_errors = child_errors;
        O.group_counts8 = group_counts8; O.group_hwe_chi2 = group_hwe_chi2; O.group_hwe_p = group_hwe_p; O.group_stride = (size_t)max_lines;
        if ((rc = stats_all_call(ctx, s, (const uint8_t *)s->buf[7], raw_pitch, nl, (const uint8_t *)s->buf[2], O))) return rc;
        if (n_multi) {
            std::vector<int32_t> idx;
            for (size_t i = 0; i < n; ++i) {
                const int32_t *c = counts8 + 8 * i;
                if (ns - c[4] - (c[0] + c[1] + c[2] + c[3]) > 0) idx.push_back((int32_t)i);
            }
            *n_multi = (int)idx.size();
            const int m = (int)idx.size() < cap ? (int)idx.size() : cap;
            if (m > 0) {
                if ((rc = ensure(ctx, s, 3, (size_t)m * 257 * sizeof(int32_t)))) return rc;
                int32_t *d_idx = (int32_t *)s->buf[3], *d_tab = d_idx + m;
                HIPCHK(ctx, hipMemcpyAsync(d_idx, idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
                if ((rc = hpgv_genotype_table_dev(ctx, (const uint8_t *)s->buf[7], raw_pitch, ns, d_idx, m, d_tab, s->stream))) return rc;
                HIPCHK(ctx, hipMemcpyAsync(multi_table, d_tab, (size_t)m * 256 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
                HIPCHK(ctx, hipStreamSynchronize(s->stream));
                memcpy(multi_idx, idx.data(), (size_t)m * sizeof(int32_t));
            }
        }
        return HPGV_OK;
    }
    if ((rc = ensure(ctx, s, 3, n * 32))) return rc;
    if ((rc = ensure(ctx, s, 4, n * 2 * sizeof(double) + 64))) return rc;
    int32_t *d_c8 = (int32_t *)s->buf[3];
    double *d_chi2 = (double *)s->buf[4], *d_p = d_chi2 + n;
    if ((rc = hpgv_stats_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, d_c8, s->stream))) return rc;
    if ((rc = hpgv_stats_hwe_dev(ctx, d_c8, nl, d_chi2, d_p, s->stream))) return rc;
    std::vector<int32_t> sm, ce;
    int32_t *d_sm = nullptr;
    if (sample_missing && ns > 0) {                                  // the line status array has been copied out: reuse its buffer
        if ((rc = ensure(ctx, s, 5, std::max((size_t)ns, n * 4) * sizeof(int32_t)))) return rc;
        d_sm = (int32_t *)s->buf[5];
        HIPCHK(ctx, hipMemsetAsync(d_sm, 0, (size_t)ns * sizeof(int32_t), s->stream));
        if ((rc = hpgv_sample_missing_dev(ctx, (const uint8_t *)s->buf[1], nl, d_sm, s->stream))) return rc;
        sm.resize((size_t)ns);
        HIPCHK(ctx, hipMemcpyAsync(sm.data(), d_sm, (size_t)ns * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(counts8, d_c8, n * 32, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(hwe_chi2, d_chi2, n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(ctx, hipMemcpyAsync(hwe_p, d_p, n * 8, hipMemcpyDeviceToHost, s->stream));
    if (want_mendel) {                                               // the stats layout in buf[1] has been consumed (same stream)
        const size_t nt = (size_t)ctx->mendel_trios;
        if ((rc = ensure(ctx, s, 1, n * std::max(ctx->mendel.pitch, ctx->stats.pitch) + 16))) return rc;
        if ((rc = hpgv_layout_dev(ctx, HPGV_LAYOUT_MENDEL, (const uint8_t *)s->buf[7], raw_pitch, nl, (uint8_t *)s->buf[1], s->stream))) return rc;
        if ((rc = ensure(ctx, s, 6, (n + nt) * sizeof(int32_t) + 64))) return rc;   // meta has been copied out
        int32_t *d_err = (int32_t *)s->buf[6], *d_child = d_err + n;
        if (mendel_errors) {
            if ((rc = hpgv_mendel_scan_dev(ctx, (const uint8_t *)s->buf[1], nl, (const uint8_t *)s->buf[2], d_err, s->stream))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(mendel_errors, d_err, n * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        }
        if (child_errors && nt) {
            HIPCHK(ctx, hipMemsetAsync(d_child, 0, nt * sizeof(int32_t), s->stream));
            if ((rc = hpgv_mendel_children_dev(ctx, (const uint8_t *)s->buf[1], nl, (const uint8_t *)s->buf[2], d_child, s->stream))) return rc;
            ce.resize(nt);
            HIPCHK(ctx, hipMemcpyAsync(ce.data(), d_child, nt * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        }
    }
    if (group_counts8) {                                             // per phenotype group: the grouped layout of the same raw matrix
        const size_t ng = ctx->sg_off.size();
        HIPCHK(ctx, hipStreamSynchronize(s->stream));                // the copies out of buf[3] / buf[4] above are done before their reuse
        if ((rc = ensure(ctx, s, 1, n * std::max(ctx->sgroups.pitch, ctx->stats.pitch) + 16))) return rc;
        if ((rc = hpgv_layout_dev(ctx, HPGV_LAYOUT_STATS_GROUPS, (const uint8_t *)s->buf[7], raw_pitch, nl, (uint8_t *)s->buf[1], s->stream))) return rc;
        if ((rc = ensure(ctx, s, 3, ng * n * 32 + 64))) return rc;
        if ((rc = ensure(ctx, s, 4, ng * n * 2 * sizeof(double) + 64))) return rc;
        int32_t *d_g8 = (int32_t *)s->buf[3];
        double *d_ghw = (double *)s->buf[4];
        for (size_t k = 0; k < ng; ++k) {
            if ((rc = hpgv_stats_scan_group_dev(ctx, (const uint8_t *)s->buf[1], nl, (int)k, d_g8 + k * n * 8, s->stream))) return rc;
            if (group_hwe_chi2 && (rc = hpgv_stats_hwe_dev(ctx, d_g8 + k * n * 8, nl, d_ghw + k * n, d_ghw + (ng + k) * n, s->stream))) return rc;
        }
        // outputs are laid out for max_lines lines per group: [g * max_lines + v]
        for (size_t k = 0; k < ng; ++k) {
            HIPCHK(ctx, hipMemcpyAsync(group_counts8 + k * (size_t)max_lines * 8, d_g8 + k * n * 8, n * 32, hipMemcpyDeviceToHost, s->stream));
            if (group_hwe_chi2) {
                HIPCHK(ctx, hipMemcpyAsync(group_hwe_chi2 + k * (size_t)max_lines, d_ghw + k * n, n * 8, hipMemcpyDeviceToHost, s->stream));
                HIPCHK(ctx, hipMemcpyAsync(group_hwe_p + k * (size_t)max_lines, d_ghw + (ng + k) * n, n * 8, hipMemcpyDeviceToHost, s->stream));
            }
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    for (size_t j = 0; j < sm.size(); ++j) sample_missing[j] += sm[j];
    for (size_t t = 0; t < ce.size(); ++t) child_errors[t] += ce[t];
    if (n_multi) {
        std::vector<int32_t> idx;
        for (size_t i = 0; i < n; ++i) {
            const int32_t *c = counts8 + 8 * i;
            if (ns - c[4] - (c[0] + c[1] + c[2] + c[3]) > 0) idx.push_back((int32_t)i);
        }
        *n_multi = (int)idx.size();
        const int m = (int)idx.size() < cap ? (int)idx.size() : cap;
        if (m > 0) {
            if ((rc = ensure(ctx, s, 3, (size_t)m * 257 * sizeof(int32_t)))) return rc;
            int32_t *d_idx = (int32_t *)s->buf[3], *d_tab = d_idx + m;
            HIPCHK(ctx, hipMemcpyAsync(d_idx, idx.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
            if ((rc = hpgv_genotype_table_dev(ctx, (const uint8_t *)s->buf[7], raw_pitch, ns, d_idx, m, d_tab, s->stream))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(multi_table, d_tab, (size_t)m * 256 * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
            HIPCHK(ctx, hipStreamSynchronize(s->stream));
            memcpy(multi_idx, idx.data(), (size_t)m * sizeof(int32_t));
        }
    }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_epi_dataset_text(hpgv_ctx *ctx, const char *text, size_t text_bytes, int max_lines, int *n_lines,
                          uint64_t *line_off, uint32_t *field_off, int32_t *status, uint8_t *out) {
    if (is_group(ctx)) {
        if (hpgv_ctx *m_ = alias_owner(ctx, text)) return hpgv_epi_dataset_text(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, out);
        Dealt d_(ctx); hpgv_ctx *m_ = d_.m; return hpgv_epi_dataset_text(m_, text, text_bytes, max_lines, n_lines, line_off, field_off, status, out);
    }
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->assoc.set) return fail(ctx, HPGV_ERR_STATE, "hpgv_set_cohort has not been called");
    if (!n_lines || max_lines < 0 || (text_bytes > 0 && !text) || (max_lines > 0 && !out))
        return fail(ctx, HPGV_ERR_INVALID, "bad epi_dataset_text arguments");
    *n_lines = 0;
    if (max_lines == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    SlotLease lease(ctx);
    int rc = acquire_slot(ctx, &lease.s);
    if (rc) return rc;
    Slot *s = lease.s;
    int nl = 0;
    if ((rc = text_front(ctx, s, HPGV_LAYOUT_EPI, ctx->assoc, text, text_bytes, max_lines, n_lines, line_off, field_off, status, &nl))) return rc;
    const size_t nA = (size_t)ctx->nA, nU = (size_t)ctx->nU, width = nA + nU;
    if (nl > 0 && width > 0) {
        const uint8_t *d = (const uint8_t *)s->buf[1];
        const size_t segA = (size_t)ctx->chunksA * 16, dp = ctx->assoc.pitch;
        if (nA) HIPCHK(ctx, hipMemcpy2DAsync(out, width, d, dp, nA, (size_t)nl, hipMemcpyDeviceToHost, s->stream));
        if (nU) HIPCHK(ctx, hipMemcpy2DAsync(out + nA, width, d + segA, dp, nU, (size_t)nl, hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(s->stream));
    return HPGV_OK;
}

int hpgv_read_probe(hpgv_ctx *ctx, const uint8_t *d_buf, size_t bytes, int iters, float *ms) {
    ctx = first_member(ctx);
    if (!ctx || !d_buf || !ms || iters <= 0 || ((uintptr_t)d_buf & 15)) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    const size_t n16 = bytes / 16;
    hipEvent_t a, b;
    HIPCHK(ctx, hipEventCreate(&a));
    HIPCHK(ctx, hipEventCreate(&b));
    const unsigned blocks = (unsigned)(ctx->n_cus * 4);      // 4 blocks x 4 waves per CU: the pipelined scan's occupancy
    auto go = [&] {
        if (ctx->nontemporal)
            hipLaunchKernelGGL((hpgv::k_read_probe<true>), dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)d_buf, n16, ctx->d_sink);
        else
            hipLaunchKernelGGL((hpgv::k_read_probe<false>), dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)d_buf, n16, ctx->d_sink);
    };
    go();
    HIPCHK(ctx, hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) go();
    HIPCHK(ctx, hipEventRecord(b, nullptr));
    HIPCHK(ctx, hipEventSynchronize(b));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, a, b));
    *ms = t / iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return HPGV_OK;
}

}  // extern "C"
