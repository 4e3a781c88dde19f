// hpgv_epi_generic_capi.hip -- C ABI of the MDR model for combinations of any order the reference's `--order` accepts
// (main_epistasis.c:128,142; model.c:76-206; epistasis.c:14-95), here 2 <= order <= 5: listed combinations evaluated
// (hpgv_epi_eval_combs), every combination of the dataset ranked per fold (hpgv_epi_rank_order[_rows]), and the in-fold
// counts behind hpgv_epi_counts / hpgv_epi_counts_all_folds for the orders the pair / triple kernels do not take.
// Kernel: hpgv_epi_generic_kernels.h (one lane per cell).  Its own translation unit: the pair and triple scans of
// hpgv_epi_capi.hip compile for minutes.
#include "hpgv_internal.h"
#include "hpgv_epi_generic_kernels.h"

namespace {

struct EventPair2 {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair2() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
struct DevFree { void *p = nullptr; ~DevFree() { if (p) (void)hipFree(p); } };

int generic_check(hpgv_ctx *ctx, int order) {
    EpiState &E = ctx->epi;
    if (!E.have_folds) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (order < 2 || order > 5) return fail(ctx, HPGV_ERR_UNSUPPORTED, "combinations of %d SNPs are not supported (2 to 5)", order);
    if (E.nA > 65535 || E.nU > 65535) return fail(ctx, HPGV_ERR_UNSUPPORTED, "the listed-combination kernel keeps 16-bit class totals: at most 65535 samples per class");
    return HPGV_OK;
}

int upload_folds(hpgv_ctx *ctx, bool training) {
    EpiState &E = ctx->epi;
    hpgv::EpiFold folds[hpgv::EPI_MAX_FOLDS];
    for (int f = 0; f < hpgv::EPI_MAX_FOLDS; ++f) {
        folds[f].test_a = E.group_size[(size_t)2 * f]; folds[f].test_u = E.group_size[(size_t)2 * f + 1];
        const int sa = training ? E.nA - folds[f].test_a : folds[f].test_a, su = training ? E.nU - folds[f].test_u : folds[f].test_u;
        folds[f].inv_a = 1.0 / (double)sa; folds[f].inv_u = 1.0 / (double)su;
    }
    HIPCHK(ctx, hipMemcpyAsync(E.d_folds, folds, sizeof folds, hipMemcpyHostToDevice, nullptr));
    return HPGV_OK;
}

// d_combs: n_combs x order SNP indices on the device.  counts only: d_counts given, everything else NULL.
int launch_combs(hpgv_ctx *ctx, int order, bool training, const int32_t *d_combs, int n_combs, int32_t *d_counts, bool evaluate,
                 double *d_acc, uint32_t *d_mask, const double *d_thr, hpgv::EpiCandN *d_cand, unsigned *d_cand_count, unsigned cap) {
    EpiState &E = ctx->epi;
    if (n_combs <= 0) return HPGV_OK;
    const hpgv::EpiFold *folds = evaluate ? E.d_folds : nullptr;
#define HPGV_COMBS(ORD, TR)                                                                                                       \
    hipLaunchKernelGGL((hpgv::k_epi_combs<ORD, TR>), dim3((unsigned)((n_combs + (256 / hpgv::EpiCells<ORD>::value) - 1) / (256 / hpgv::EpiCells<ORD>::value))), \
                       dim3(256), 0, nullptr, E.d_planes, E.W, d_combs, n_combs, E.d_group_w0, E.num_folds, folds, E.nA, E.nU, d_counts, \
                       d_acc, d_mask, d_thr, d_cand, d_cand_count, cap)
#define HPGV_COMBS_T(ORD) do { if (training) HPGV_COMBS(ORD, true); else HPGV_COMBS(ORD, false); } while (0)
    switch (order) {
        case 2: HPGV_COMBS_T(2); break;
        case 3: HPGV_COMBS_T(3); break;
        case 4: HPGV_COMBS_T(4); break;
        default: HPGV_COMBS_T(5); break;
    }
#undef HPGV_COMBS_T
#undef HPGV_COMBS
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

// the next combination in lexicographic order; false after the last
bool next_comb(int32_t *c, int order, int V) {
    int s = order - 1;
    while (s >= 0 && c[s] == V - order + s) --s;
    if (s < 0) return false;
    ++c[s];
    for (int k = s + 1; k < order; ++k) c[k] = c[k - 1] + 1;
    return true;
}

}  // namespace

// in-fold counts of listed combinations, device to device: d_out[(comb * n_groups + g) * 3^order + cell] (epi_infold_counts of
// hpgv_epi_capi.hip for the orders its own kernel does not take).  The caller holds epi_mu.
int hpgv_epi_generic_counts(hpgv_ctx *ctx, int order, const int32_t *d_combs, int n_combs, int32_t *d_out) {
    int rc = generic_check(ctx, order);
    if (rc) return rc;
    return launch_combs(ctx, order, false, d_combs, n_combs, d_out, false, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}

extern "C" {

int hpgv_epi_eval_combs(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs, int subset, double *accuracy, uint32_t *risky_mask) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    int rc = generic_check(ctx, order);
    if (rc) return rc;
    EpiState &E = ctx->epi;
    if (subset != HPGV_EPI_TESTING && subset != HPGV_EPI_TRAINING) return fail(ctx, HPGV_ERR_INVALID, "subset must be HPGV_EPI_TESTING or HPGV_EPI_TRAINING");
    if (n_combs < 0 || (n_combs > 0 && (!combs || !accuracy))) return fail(ctx, HPGV_ERR_INVALID, "bad combination list");
    for (long k = 0; k < (long)n_combs * order; ++k)
        if (combs[k] < 0 || combs[k] >= E.V) return fail(ctx, HPGV_ERR_INVALID, "SNP index %d outside the dataset", combs[k]);
    if (n_combs == 0) return HPGV_OK;
    const size_t nf = (size_t)E.num_folds, n = (size_t)n_combs;
    DevFree dc, da, dm;
    HIPCHK(ctx, hipMalloc(&dc.p, n * (size_t)order * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc(&da.p, n * nf * sizeof(double)));
    if (risky_mask) HIPCHK(ctx, hipMalloc(&dm.p, n * nf * hpgv::EPI_MASK_WORDS * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemcpy(dc.p, combs, n * (size_t)order * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemset(da.p, 0xFF, n * nf * sizeof(double)));      // (a fold without samples keeps NaN)
    if (dm.p) HIPCHK(ctx, hipMemset(dm.p, 0, n * nf * hpgv::EPI_MASK_WORDS * sizeof(uint32_t)));
    rc = upload_folds(ctx, subset == HPGV_EPI_TRAINING);
    if (rc) return rc;
    rc = launch_combs(ctx, order, subset == HPGV_EPI_TRAINING, (const int32_t *)dc.p, n_combs, nullptr, true, (double *)da.p, (uint32_t *)dm.p,
                      nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpy(accuracy, da.p, n * nf * sizeof(double), hipMemcpyDeviceToHost));
    if (risky_mask) HIPCHK(ctx, hipMemcpy(risky_mask, dm.p, n * nf * hpgv::EPI_MASK_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_epi_rank_order_rows(hpgv_ctx *ctx, int order, int i_begin, int i_end, int subset, int max_ranking_size, int32_t *combs_out,
                             double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    int rc = generic_check(ctx, order);
    if (rc) return rc;
    EpiState &E = ctx->epi;
    if (subset != HPGV_EPI_TESTING && subset != HPGV_EPI_TRAINING) return fail(ctx, HPGV_ERR_INVALID, "subset must be HPGV_EPI_TESTING or HPGV_EPI_TRAINING");
    if (i_begin < 0 || i_end < i_begin || i_end > E.V) return fail(ctx, HPGV_ERR_INVALID, "first SNPs [%d, %d) outside the dataset", i_begin, i_end);
    if (max_ranking_size < 1 || max_ranking_size > 65536 || !combs_out || !accuracy || !risky_mask || !n_ranked)
        return fail(ctx, HPGV_ERR_INVALID, "bad ranking arguments");
    const int nf = E.num_folds, N = max_ranking_size, V = E.V;
    struct Model { double accuracy; unsigned long long rank; int32_t c[5]; uint32_t risky[hpgv::EPI_MASK_WORDS]; };
    std::vector<std::vector<Model>> top((size_t)nf);
    // listed in lexicographic order, so the position in the listing IS the tie-break of add_to_model_ranking
    // (model.c:478-517: higher accuracy, then the smaller combination)
    auto better = [](const Model &a, const Model &b) { return a.accuracy != b.accuracy ? a.accuracy > b.accuracy : a.rank < b.rank; };
    for (int f = 0; f < nf; ++f) n_ranked[f] = 0;
    if (V < order || i_begin >= i_end || i_begin > V - order) { if (scan_ms) *scan_ms = 0.f; return HPGV_OK; }
    constexpr unsigned CHUNK = 1u << 17;                             // combinations per launch = capacity of a fold's candidate list
    DevFree dc, dcand, dcount, dthr;
    HIPCHK(ctx, hipMalloc(&dc.p, (size_t)CHUNK * (size_t)order * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc(&dcand.p, (size_t)nf * CHUNK * sizeof(hpgv::EpiCandN)));
    HIPCHK(ctx, hipMalloc(&dcount.p, hpgv::EPI_MAX_FOLDS * sizeof(unsigned)));
    HIPCHK(ctx, hipMalloc(&dthr.p, hpgv::EPI_MAX_FOLDS * sizeof(double)));
    rc = upload_folds(ctx, subset == HPGV_EPI_TRAINING);
    if (rc) return rc;
    std::vector<int32_t> list((size_t)CHUNK * (size_t)order);
    std::vector<double> thr(hpgv::EPI_MAX_FOLDS, -HUGE_VAL);
    std::vector<unsigned> count(hpgv::EPI_MAX_FOLDS);
    std::vector<hpgv::EpiCandN> buf;
    EventPair2 evs;
    float total_ms = 0.f;
    if (scan_ms) { HIPCHK(ctx, hipEventCreate(&evs.a)); HIPCHK(ctx, hipEventCreate(&evs.b)); }
    int32_t cur[5];
    for (int s = 0; s < order; ++s) cur[s] = i_begin + s;
    bool more = true;
    unsigned long long listed = 0;
    // the first launch has no thresholds and lists every combination it evaluates: a short one, so that the long ones that follow
    // have bounds to filter with (131 072 models x folds x 64 bytes over the bus and a sort otherwise)
    unsigned limit = std::min<unsigned>(CHUNK, (unsigned)std::max(4096, 4 * N));
    while (more) {
        unsigned n = 0;
        while (more && n < limit && cur[0] < i_end) {
            for (int s = 0; s < order; ++s) list[(size_t)n * (size_t)order + (size_t)s] = cur[s];
            ++n;
            more = next_comb(cur, order, V);
        }
        if (cur[0] >= i_end) more = false;
        if (n == 0) break;
        HIPCHK(ctx, hipMemcpyAsync(dc.p, list.data(), (size_t)n * (size_t)order * sizeof(int32_t), hipMemcpyHostToDevice, nullptr));
        HIPCHK(ctx, hipMemsetAsync(dcount.p, 0, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), nullptr));
        HIPCHK(ctx, hipMemcpyAsync(dthr.p, thr.data(), hpgv::EPI_MAX_FOLDS * sizeof(double), hipMemcpyHostToDevice, nullptr));
        if (scan_ms) HIPCHK(ctx, hipEventRecord(evs.a, nullptr));
        rc = launch_combs(ctx, order, subset == HPGV_EPI_TRAINING, (const int32_t *)dc.p, (int)n, nullptr, true, nullptr, nullptr,
                          (const double *)dthr.p, (hpgv::EpiCandN *)dcand.p, (unsigned *)dcount.p, CHUNK);
        if (rc) return rc;
        if (scan_ms) HIPCHK(ctx, hipEventRecord(evs.b, nullptr));
        HIPCHK(ctx, hipMemcpy(count.data(), dcount.p, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (scan_ms) { float ms = 0.f; HIPCHK(ctx, hipEventElapsedTime(&ms, evs.a, evs.b)); total_ms += ms; }
        for (int f = 0; f < nf; ++f) {
            const unsigned m = std::min(count[(size_t)f], CHUNK);      // (a list holds every combination of a launch: it cannot overflow)
            if (!m) continue;
            buf.resize(m);
            HIPCHK(ctx, hipMemcpy(buf.data(), (hpgv::EpiCandN *)dcand.p + (size_t)f * CHUNK, (size_t)m * sizeof(hpgv::EpiCandN), hipMemcpyDeviceToHost));
            auto &t = top[(size_t)f];
            for (const hpgv::EpiCandN &e : buf) {
                Model md;
                md.accuracy = e.accuracy; md.rank = listed + e.index;
                for (int s = 0; s < 5; ++s) md.c[s] = s < order ? list[(size_t)e.index * (size_t)order + (size_t)s] : -1;
                for (int w = 0; w < hpgv::EPI_MASK_WORDS; ++w) md.risky[w] = e.risky[w];
                t.push_back(md);
            }
            if ((int)t.size() > N) { std::partial_sort(t.begin(), t.begin() + N, t.end(), better); t.resize((size_t)N); }
            else std::sort(t.begin(), t.end(), better);
            if ((int)t.size() >= N && t.back().accuracy > thr[(size_t)f]) thr[(size_t)f] = t.back().accuracy;
        }
        listed += n;
        limit = CHUNK;
    }
    for (int f = 0; f < nf; ++f) {
        const auto &t = top[(size_t)f];
        n_ranked[f] = (int32_t)t.size();
        for (size_t e = 0; e < t.size(); ++e) {
            const size_t o = (size_t)f * (size_t)N + e;
            for (int s = 0; s < order; ++s) combs_out[o * (size_t)order + (size_t)s] = t[e].c[s];
            accuracy[o] = t[e].accuracy;
            for (int w = 0; w < hpgv::EPI_MASK_WORDS; ++w) risky_mask[o * hpgv::EPI_MASK_WORDS + (size_t)w] = t[e].risky[w];
        }
    }
    if (scan_ms) *scan_ms = total_ms;
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_epi_rank_order(hpgv_ctx *ctx, int order, int subset, int max_ranking_size, int32_t *combs_out, double *accuracy,
                        uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    const hpgv_ctx *c = first_member(ctx);
    return hpgv_epi_rank_order_rows(ctx, order, 0, c ? c->epi.V : 0, subset, max_ranking_size, combs_out, accuracy, risky_mask, n_ranked, scan_ms);
}

}  // extern "C"
