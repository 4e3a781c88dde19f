// hpgv_epi_capi.hip -- C ABI of the epistasis / MDR path (its own translation unit of libhpgv.so: the pair and triple
// scans are instantiated per fold count and compile for minutes).
#include "hpgv_internal.h"
#include "hpgv_epi_triples3_kernels.h"
#include "hpgv_epi_mfma_kernels.h"


namespace {

struct EventPair {                                                   // the two timing events of a ranking call
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

void epi_free_folds(EpiState &E) {
    if (E.d_planes) (void)hipFree(E.d_planes);
    if (E.d_marg) (void)hipFree(E.d_marg);
    E.d_marg = nullptr;
    if (E.d_chunks) (void)hipFree(E.d_chunks);
    if (E.d_chunk_cls) (void)hipFree(E.d_chunk_cls);
    E.d_chunk_cls = nullptr;
    if (E.d_folds) (void)hipFree(E.d_folds);
    if (E.d_group_w0) (void)hipFree(E.d_group_w0);
    E.rev_off = 0;
    E.d_planes = nullptr; E.d_chunks = nullptr; E.d_folds = nullptr; E.d_group_w0 = nullptr;
    E.have_folds = false;
}

void epi_free(EpiState &E) {
    epi_free_folds(E);
    if (E.d_data) (void)hipFree(E.d_data);
    if (E.d_cand) (void)hipFree(E.d_cand);
    if (E.d_cand3) (void)hipFree(E.d_cand3);
    E.d_cand3 = nullptr; E.cand3_cap = 0;
    if (E.d_cand_count) (void)hipFree(E.d_cand_count);
    if (E.d_thr) (void)hipFree(E.d_thr);
    if (E.d_tile_base) (void)hipFree(E.d_tile_base);
    E.d_tile_base = nullptr; E.tile_base_cap = 0;
    E.d_data = nullptr; E.d_cand = nullptr; E.d_cand_count = nullptr; E.d_thr = nullptr;
    E.have_data = false;
}

int epi_build_folds(hpgv_ctx *ctx, const int32_t *fold_of_sample, int num_folds) {
    EpiState &E = ctx->epi;
    const int n = E.nA + E.nU;
    epi_free_folds(E);
    // sample order: for every fold its affected, then its unaffected samples; every (fold, class) run starts on a
    // 4-word boundary (16-byte reads) and is padded with zero bits
    std::vector<std::vector<int32_t>> members((size_t)num_folds * 2);
    for (int s = 0; s < n; ++s) {
        const int f = fold_of_sample ? fold_of_sample[s] : 0;
        if (f < 0 || f >= num_folds) return fail(ctx, HPGV_ERR_INVALID, "fold %d of sample %d is outside [0, %d)", f, s, num_folds);
        members[(size_t)f * 2 + (s < E.nA ? 0 : 1)].push_back(s);
    }
    std::vector<int32_t> src;
    std::vector<uint32_t> w0((size_t)num_folds * 2 + 1);
    std::vector<int32_t> gsize(hpgv::EPI_MAX_FOLDS * 2, -1);
    std::vector<int> group_of_step;                                  // per 4-word step: the group it belongs to
    for (int g = 0; g < num_folds * 2; ++g) {
        const auto &m = members[(size_t)g];
        if (m.size() >= 65536) return fail(ctx, HPGV_ERR_UNSUPPORTED, "%zu samples of one class in one fold: the pair scan keeps 16-bit counts per (fold, class)", m.size());
        gsize[(size_t)g] = (int32_t)m.size();
        w0[(size_t)g] = (uint32_t)(src.size() / 32);
        src.insert(src.end(), m.begin(), m.end());
        while (src.size() % 128) src.push_back(-1);
        group_of_step.resize(src.size() / 128, g);
    }
    if (src.empty()) { src.assign(128, -1); group_of_step.assign(1, 0); }      // no samples at all: one step of pad bits
    if (group_of_step.size() % 2) {                                  // the scan takes two steps per turn: an even number of steps;
        src.insert(src.end(), 128, -1);                              // the extra step holds pad bits only and belongs to no group
        group_of_step.push_back(-1);
    }
    w0[(size_t)num_folds * 2] = (uint32_t)(src.size() / 32);
    // staging blocks of up to EPI_CH words; a block may hold the runs of several groups: byte s of `flush` names the
    // group whose run ends with the block's step s
    std::vector<hpgv::EpiChunk> chunks;
    std::vector<uint32_t> chunk_cls;
    const size_t n_steps = group_of_step.size(), steps_per_block = hpgv::EPI_CH / 4;
    for (size_t s0 = 0; s0 < n_steps; s0 += steps_per_block) {
        hpgv::EpiChunk c;
        const size_t ns = n_steps - s0 < steps_per_block ? n_steps - s0 : steps_per_block;
        c.w0 = (uint32_t)(s0 * 4); c.nw = (uint32_t)(ns * 4); c.flush = ~0ull;
        for (size_t k = 0; k < ns; ++k) {
            const bool last_of_group = group_of_step[s0 + k] >= 0 && (s0 + k + 1 == n_steps || group_of_step[s0 + k + 1] != group_of_step[s0 + k]);
            if (last_of_group) c.flush = (c.flush & ~(0xFFull << (8 * k))) | ((uint64_t)group_of_step[s0 + k] << (8 * k));
        }
        chunks.push_back(c);
        uint32_t m = 0;                                              // bit k: step k of the block holds controls (k_epi_pairs_mfma's first pass)
        for (size_t k = 0; k < ns; ++k) if (group_of_step[s0 + k] >= 0 && (group_of_step[s0 + k] & 1)) m |= 1u << k;
        chunk_cls.push_back(m);
    }
    E.W = (int)(src.size() / 32);
    E.num_folds = num_folds;
    E.n_chunks = (int)chunks.size();
    E.group_size = gsize;
    E.V_alloc = (E.V + 63) / 64 * 64 + 64;                           // whole tiles past the last SNP read zero planes
    if ((size_t)E.V_alloc * 3 * (size_t)E.W + hpgv::EPI_CH >= (1ull << 32))
        return fail(ctx, HPGV_ERR_UNSUPPORTED, "%d SNPs x %d samples: the genotype planes exceed 2^32 words (the scans address them with 32-bit word offsets)", E.V, n);
    int32_t *d_src = nullptr;
    HIPCHK(ctx, hipMalloc(&d_src, src.size() * sizeof(int32_t)));
    hipError_t e = hipMemcpy(d_src, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    // + slack: the LDS-DMA staging fetches whole 32-word rows.  Behind the planes, while the scans' 32-bit word offsets reach it, their
    // copy with the low seven bits of every byte reversed (hpgv_epi_mfma_kernels.h)
    const size_t plane_words = (size_t)E.V_alloc * 3 * (size_t)E.W + hpgv::EPI_CH;
    E.rev_off = 2 * plane_words < (1ull << 32) ? (uint32_t)plane_words : 0u;
    if (e == hipSuccess) {
        e = hipMalloc(&E.d_planes, (plane_words + E.rev_off) * sizeof(uint32_t));
        if (e != hipSuccess && E.rev_off) {                          // no room for the copy: the vector-ALU scans do without it
            (void)hipGetLastError();
            E.rev_off = 0;
            e = hipMalloc(&E.d_planes, plane_words * sizeof(uint32_t));
        }
    }
    if (e == hipSuccess) e = hipMalloc(&E.d_chunks, chunks.size() * sizeof(hpgv::EpiChunk));
    if (e == hipSuccess) e = hipMemcpy(E.d_chunks, chunks.data(), chunks.size() * sizeof(hpgv::EpiChunk), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&E.d_chunk_cls, chunk_cls.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(E.d_chunk_cls, chunk_cls.data(), chunk_cls.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&E.d_folds, hpgv::EPI_MAX_FOLDS * sizeof(hpgv::EpiFold));
    if (e == hipSuccess) e = hipMalloc(&E.d_group_w0, w0.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(E.d_group_w0, w0.data(), w0.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    unsigned *d_flag = nullptr;
    unsigned flag = 1;
    if (e == hipSuccess) e = hipMalloc(&E.d_marg, (size_t)E.V_alloc * 2 * hpgv::EPI_MAX_FOLDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_flag, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(d_flag, 0, sizeof(unsigned));
    if (e == hipSuccess) {
        if (n <= 65000)                                              // the row goes through LDS
            hipLaunchKernelGGL(hpgv::k_epi_planes, dim3((unsigned)E.V_alloc), dim3(256), (size_t)n + 8, nullptr, E.d_data, E.V, n, d_src, E.W, E.d_planes, d_flag);
        else
            hipLaunchKernelGGL(hpgv::k_epi_planes_gather, dim3((unsigned)E.V_alloc), dim3(256), 0, nullptr, E.d_data, E.V, n, d_src, E.W, E.d_planes, d_flag);
        // genotype counts per SNP and group: what the complete-data pair scan derives the cells with a genotype 2 from
        hipLaunchKernelGGL(hpgv::k_epi_marginals, dim3((unsigned)E.V_alloc), dim3(256), 0, nullptr, E.d_planes, E.W, E.d_group_w0, num_folds * 2, E.d_marg);
        if (E.rev_off)                                               // the column side's copy for the matrix-core pair scan
            hipLaunchKernelGGL(hpgv::k_epi_planes_rev, dim3((unsigned)((plane_words + 255) / 256)), dim3(256), 0, nullptr, E.d_planes, plane_words, E.d_planes + E.rev_off);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(&flag, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost);      // synchronises
    }
    E.complete = flag == 0;                                          // no call other than 0 / 1 / 2 anywhere
    if (d_flag) (void)hipFree(d_flag);
    (void)hipFree(d_src);
    if (e != hipSuccess) { epi_free_folds(E); return fail(ctx, HPGV_ERR_HIP, "building the genotype planes failed: %s", hipGetErrorString(e)); }
    E.have_folds = true;
    return HPGV_OK;
}

// per fold: testing sizes and the reciprocals of the evaluated part's sizes (RN(1 / y): IEEE double division on the host), for
// the scan kernels' evaluation
template <bool TRAINING>
int epi_upload_folds(hpgv_ctx *ctx, hipStream_t st) {
    EpiState &E = ctx->epi;
    hpgv::EpiFold folds[hpgv::EPI_MAX_FOLDS];
    for (int f = 0; f < hpgv::EPI_MAX_FOLDS; ++f) {
        folds[f].test_a = E.group_size[(size_t)2 * f]; folds[f].test_u = E.group_size[(size_t)2 * f + 1];
        const int sa = TRAINING ? E.nA - folds[f].test_a : folds[f].test_a, su = TRAINING ? E.nU - folds[f].test_u : folds[f].test_u;
        folds[f].inv_a = 1.0 / (double)sa; folds[f].inv_u = 1.0 / (double)su;
    }
    HIPCHK(ctx, hipMemcpyAsync(E.d_folds, folds, sizeof folds, hipMemcpyHostToDevice, st));
    return HPGV_OK;
}

// the ranking scan with the cell counts on the matrix cores (hpgv_epi_mfma_kernels.h): tiles of 16 rows x 64 columns
template <bool TRAINING, bool BALANCED>
int epi_launch_pairs_mfma(hpgv_ctx *ctx, int i_begin, int i_end, hipStream_t st) {
    EpiState &E = ctx->epi;
    const int i_first = i_begin;
    i_begin = i_begin / 64 * 64;
    const int tiles_j = (E.V + hpgv::EPI_TJ - 1) / hpgv::EPI_TJ, row_blocks = (i_end - i_begin + hpgv::EPM_TI - 1) / hpgv::EPM_TI;
    const int n_cols = tiles_j - i_begin / 64;
    if (n_cols <= 0 || row_blocks <= 0) return HPGV_OK;
    std::vector<unsigned> tile_base((size_t)n_cols + 1);
    unsigned long long total = 0;
    for (int c = 0; c < n_cols; ++c) {                               // column tile c holds pairs with the rows above its last column
        tile_base[(size_t)c] = (unsigned)total;
        total += (unsigned long long)std::min<long long>(row_blocks, (64ll / hpgv::EPM_TI) * (c + 1));
    }
    tile_base[(size_t)n_cols] = (unsigned)total;
    if (total + 8 > 0x7FFFFFFFull / 256) return fail(ctx, HPGV_ERR_UNSUPPORTED, "row band too large for one launch");
    if (total == 0) return HPGV_OK;
    if (E.tile_base_cap < tile_base.size()) {
        if (E.d_tile_base) (void)hipFree(E.d_tile_base);
        E.d_tile_base = nullptr; E.tile_base_cap = 0;
        HIPCHK(ctx, hipMalloc(&E.d_tile_base, (tile_base.size() + 64) * sizeof(unsigned)));
        E.tile_base_cap = tile_base.size() + 64;
    }
    HIPCHK(ctx, hipMemcpyAsync(E.d_tile_base, tile_base.data(), tile_base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    const unsigned n_tiles = (unsigned)total;
    const dim3 grid((n_tiles + 7u) / 8u * 8u);
    if (int rc = epi_upload_folds<TRAINING>(ctx, st)) return rc;
#define HPGV_EPM_LAUNCH(COMPLETEV)                                                                                                   \
    hipLaunchKernelGGL((hpgv::k_epi_pairs_mfma<TRAINING, BALANCED, COMPLETEV>), grid, dim3(256), 0, st, E.d_planes, E.d_marg, E.rev_off, E.W, \
                       E.V, i_begin, i_first, i_end, E.d_tile_base, n_cols, n_tiles, E.d_chunks, E.d_chunk_cls, E.n_chunks, E.d_folds, E.num_folds, E.nA, E.nU, E.d_thr, E.d_cand, E.d_cand_count, E.cand_cap)
    if (E.complete && ctx->epi_complete) HPGV_EPM_LAUNCH(true);      // no missing call in the dataset: four cells counted, five derived
    else HPGV_EPM_LAUNCH(false);
#undef HPGV_EPM_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

template <bool TRAINING, bool BALANCED>
int epi_launch_pairs2(hpgv_ctx *ctx, int i_begin, int i_end, double *d_acc, uint16_t *d_mask, unsigned long long n_pairs_out,
                      unsigned long long rank_base, bool candidates, hipStream_t st) {
    EpiState &E = ctx->epi;
    if (candidates && !d_acc && ctx->epi_pairs_mfma && E.rev_off && E.n_chunks <= hpgv::EPM_MAX_CHUNKS && E.nA < 65536 && E.nU < 65536)
        return epi_launch_pairs_mfma<TRAINING, BALANCED>(ctx, i_begin, i_end, st);
    // tiles that hold at least one pair, numbered column tile by column tile: column tile tj0 + c pairs with the row
    // blocks from the band's first row down to the diagonal, min(row_blocks, 16 (c + 1)) of them
    const int i_first = i_begin;
    i_begin = i_begin / 64 * 64;                                     // the tile numbering starts on a multiple of 64; rows before i_first emit nothing
    const int tiles_j = (E.V + hpgv::EPI_TJ - 1) / hpgv::EPI_TJ, row_blocks = (i_end - i_begin + hpgv::EPI_TI - 1) / hpgv::EPI_TI;
    const int n_cols = tiles_j - i_begin / 64;
    if (n_cols <= 0 || row_blocks <= 0) return HPGV_OK;
    std::vector<unsigned> tile_base((size_t)n_cols + 1);
    unsigned long long total = 0;
    for (int c = 0; c < n_cols; ++c) {
        tile_base[(size_t)c] = (unsigned)total;
        total += (unsigned long long)std::min<long long>(row_blocks, 16ll * (c + 1));
    }
    tile_base[(size_t)n_cols] = (unsigned)total;
    if (total + 8 > 0x7FFFFFFFull / 256) return fail(ctx, HPGV_ERR_UNSUPPORTED, "row band too large for one launch");
    if (total == 0) return HPGV_OK;
    if (E.tile_base_cap < tile_base.size()) {
        if (E.d_tile_base) (void)hipFree(E.d_tile_base);
        E.d_tile_base = nullptr; E.tile_base_cap = 0;
        HIPCHK(ctx, hipMalloc(&E.d_tile_base, (tile_base.size() + 64) * sizeof(unsigned)));
        E.tile_base_cap = tile_base.size() + 64;
    }
    HIPCHK(ctx, hipMemcpyAsync(E.d_tile_base, tile_base.data(), tile_base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    const unsigned n_tiles = (unsigned)total;
    const dim3 grid((n_tiles + 7u) / 8u * 8u);                       // eight spans, one per XCD
    if (int rc = epi_upload_folds<TRAINING>(ctx, st)) return rc;
#define HPGV_EPI_LAUNCH(KK)                                                                                                         \
    hipLaunchKernelGGL((hpgv::k_epi_pairs<KK, TRAINING, BALANCED, COMPLETEV>), grid, dim3(256), 0, st, E.d_planes, E.d_marg, E.W, E.V, i_begin, i_first, i_end, \
                       E.d_tile_base, n_cols, n_tiles, E.d_chunks, E.n_chunks, E.d_folds, E.nA, E.nU, d_acc, d_mask, n_pairs_out, rank_base,                          \
                       candidates ? E.d_thr : nullptr, candidates ? E.d_cand : nullptr, E.d_cand_count, E.cand_cap)
    const int k = E.num_folds;
    if (E.complete && ctx->epi_complete) {
        constexpr bool COMPLETEV = true;                                 // no missing call in the dataset: four cells counted, five derived
        if (k <= 2) HPGV_EPI_LAUNCH(2);
        else if (k <= 4) HPGV_EPI_LAUNCH(4);
        else if (k <= 5) HPGV_EPI_LAUNCH(5);
        else if (k <= 8) HPGV_EPI_LAUNCH(8);
        else if (k <= 10) HPGV_EPI_LAUNCH(10);
        else HPGV_EPI_LAUNCH(16);
    } else {
        constexpr bool COMPLETEV = false;
        if (k <= 2) HPGV_EPI_LAUNCH(2);
        else if (k <= 4) HPGV_EPI_LAUNCH(4);
        else if (k <= 5) HPGV_EPI_LAUNCH(5);
        else if (k <= 8) HPGV_EPI_LAUNCH(8);
        else if (k <= 10) HPGV_EPI_LAUNCH(10);
        else HPGV_EPI_LAUNCH(16);                                    // 11 .. 16 folds: two waves per SIMD
    }
#undef HPGV_EPI_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

template <bool TRAINING>
int epi_launch_pairs(hpgv_ctx *ctx, int i_begin, int i_end, double *d_acc, uint16_t *d_mask, unsigned long long n_pairs_out,
                     unsigned long long rank_base, bool candidates, hipStream_t st) {
    // equal class sizes make the MDR rule's ratio exactly 1: the rule is then an integer comparison (hpgv_epi_kernels.h)
    const bool balanced = ctx->epi.nA == ctx->epi.nU && ctx->epi.nA < (1 << 22);
    return balanced ? epi_launch_pairs2<TRAINING, true>(ctx, i_begin, i_end, d_acc, d_mask, n_pairs_out, rank_base, candidates, st)
                    : epi_launch_pairs2<TRAINING, false>(ctx, i_begin, i_end, d_acc, d_mask, n_pairs_out, rank_base, candidates, st);
}

unsigned long long epi_rank(unsigned long long V, unsigned long long i) { return i * (2ull * V - i - 1ull) / 2ull; }   // rank of (i, i+1)

}  // namespace

void hpgv_epi_release(EpiState &E) { epi_free(E); }

int hpgv_epi_set_dataset(hpgv_ctx *ctx, const uint8_t *genotypes, int n_variants, int n_affected, int n_unaffected) {
    // a group context: the dataset and the folds go to EVERY member (hpgv_group_epi_rank deals the scan out to them); the
    // scans and rankings of one context work on the first member
    GROUP_ALL(ctx, hpgv_epi_set_dataset(m_, genotypes, n_variants, n_affected, n_unaffected))
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_variants < 0 || n_affected < 0 || n_unaffected < 0 || (n_variants > 0 && n_affected + n_unaffected > 0 && !genotypes))
        return fail(ctx, HPGV_ERR_INVALID, "bad epistasis dataset arguments");
    if ((long long)n_affected + n_unaffected > 0x7FFFFFFF / 2) return fail(ctx, HPGV_ERR_UNSUPPORTED, "too many samples");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    EpiState &E = ctx->epi;
    HIPCHK(ctx, hipDeviceSynchronize());
    epi_free(E);
    E.V = n_variants; E.nA = n_affected; E.nU = n_unaffected;
    const size_t bytes = (size_t)n_variants * (size_t)(n_affected + n_unaffected);
    HIPCHK(ctx, hipMalloc(&E.d_data, bytes ? bytes : 16));
    if (bytes) HIPCHK(ctx, hipMemcpy(E.d_data, genotypes, bytes, hipMemcpyHostToDevice));
    E.have_data = true;
    return epi_build_folds(ctx, nullptr, 1);                         // until folds are given: one fold holding everybody
}

int hpgv_epi_set_folds(hpgv_ctx *ctx, const int32_t *fold_of_sample, int num_folds) {
    GROUP_ALL(ctx, hpgv_epi_set_folds(m_, fold_of_sample, num_folds))
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->epi.have_data) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (num_folds < 1 || num_folds > hpgv::EPI_MAX_FOLDS) return fail(ctx, HPGV_ERR_UNSUPPORTED, "num_folds must be in [1, %d]", hpgv::EPI_MAX_FOLDS);
    if (!fold_of_sample && ctx->epi.nA + ctx->epi.nU > 0) return fail(ctx, HPGV_ERR_INVALID, "fold_of_sample is NULL");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    HIPCHK(ctx, hipDeviceSynchronize());
    return epi_build_folds(ctx, fold_of_sample, num_folds);
}

int hpgv_epi_set_fold_masks(hpgv_ctx *ctx, const uint8_t *fold_masks, int num_folds) {
    GROUP_ALL(ctx, hpgv_epi_set_fold_masks(m_, fold_masks, num_folds))
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!ctx->epi.have_data) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (num_folds < 1 || num_folds > hpgv::EPI_MAX_FOLDS || !fold_masks) return fail(ctx, HPGV_ERR_INVALID, "bad fold mask arguments");
    const int nA = ctx->epi.nA, nU = ctx->epi.nU;
    const size_t padA = ((size_t)nA + 15) / 16 * 16, padded = padA + ((size_t)nU + 15) / 16 * 16;   // masks_info_init, model.c:212-214
    std::vector<int32_t> fold((size_t)(nA + nU), -1);
    for (int s = 0; s < nA + nU; ++s) {
        const size_t pos = s < nA ? (size_t)s : padA + (size_t)(s - nA);
        for (int f = 0; f < num_folds; ++f)
            if (!(fold_masks[(size_t)f * padded + pos] & 1)) {
                if (fold[(size_t)s] >= 0)
                    return fail(ctx, HPGV_ERR_UNSUPPORTED, "sample %d is left out of the training part of folds %d and %d: the masks must be the k-fold partition get_k_folds_masks makes", s, fold[(size_t)s], f);
                fold[(size_t)s] = f;
            }
        if (fold[(size_t)s] < 0) return fail(ctx, HPGV_ERR_UNSUPPORTED, "sample %d is in the training part of every fold", s);
    }
    return hpgv_epi_set_folds(ctx, fold.data(), num_folds);
    HPGV_ABI_CATCH(ctx)
}

static int epi_cells(int order) { int c = 1; for (int k = 0; k < order; ++k) c *= 3; return c; }

// in-fold counts of listed combinations: host vector [(comb * n_groups + g) * cells + c]
static int epi_infold_counts(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs, std::vector<int32_t> &out) {
    EpiState &E = ctx->epi;
    if (!E.have_folds) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (order < 2 || order > 5) return fail(ctx, HPGV_ERR_UNSUPPORTED, "combinations of %d SNPs are not supported (2 to 5)", order);
    if (n_combs < 0 || (n_combs > 0 && !combs)) return fail(ctx, HPGV_ERR_INVALID, "bad combination list");
    for (int k = 0; k < n_combs * order; ++k)
        if (combs[k] < 0 || combs[k] >= E.V) return fail(ctx, HPGV_ERR_INVALID, "SNP index %d outside the dataset", combs[k]);
    const int cells = epi_cells(order), ng = E.num_folds * 2;
    out.assign((size_t)n_combs * ng * cells, 0);
    if (n_combs == 0) return HPGV_OK;
    int32_t *d_combs = nullptr, *d_out = nullptr;
    HIPCHK(ctx, hipMalloc(&d_combs, (size_t)n_combs * order * sizeof(int32_t)));
    hipError_t e = hipMalloc(&d_out, out.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d_combs, combs, (size_t)n_combs * order * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((n_combs + 3) / 4));
        int rc4 = HPGV_OK;
        if (order == 2) hipLaunchKernelGGL((hpgv::k_epi_counts<2>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, d_combs, n_combs, E.d_group_w0, ng, d_out);
        else if (order == 3) hipLaunchKernelGGL((hpgv::k_epi_counts<3>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, d_combs, n_combs, E.d_group_w0, ng, d_out);
        else rc4 = hpgv_epi_generic_counts(ctx, order, d_combs, n_combs, d_out);      // one lane per cell (hpgv_epi_generic_kernels.h)
        e = hipGetLastError();
        if (rc4) { (void)hipFree(d_combs); (void)hipFree(d_out); return rc4; }
    }
    if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_combs);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, HPGV_ERR_HIP, "epistasis counts failed: %s", hipGetErrorString(e));
    return HPGV_OK;
}

int hpgv_epi_counts(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs, int32_t *counts_aff, int32_t *counts_unaff) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_combs > 0 && (!counts_aff || !counts_unaff)) return fail(ctx, HPGV_ERR_INVALID, "count outputs are NULL");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    std::vector<int32_t> in;
    int rc = epi_infold_counts(ctx, order, combs, n_combs, in);
    if (rc) return rc;
    const int cells = epi_cells(order), ng = ctx->epi.num_folds * 2;
    for (int k = 0; k < n_combs; ++k)
        for (int c = 0; c < cells; ++c) {
            int a = 0, u = 0;
            for (int gi = 0; gi < ng; gi += 2) { a += in[((size_t)k * ng + gi) * cells + c]; u += in[((size_t)k * ng + gi + 1) * cells + c]; }
            counts_aff[(size_t)k * cells + c] = a; counts_unaff[(size_t)k * cells + c] = u;
        }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_epi_counts_all_folds(hpgv_ctx *ctx, int order, const int32_t *combs, int n_combs, int32_t *counts_aff, int32_t *counts_unaff) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_combs > 0 && (!counts_aff || !counts_unaff)) return fail(ctx, HPGV_ERR_INVALID, "count outputs are NULL");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    std::vector<int32_t> in;
    int rc = epi_infold_counts(ctx, order, combs, n_combs, in);
    if (rc) return rc;
    const int cells = epi_cells(order), nf = ctx->epi.num_folds, ng = nf * 2;
    for (int k = 0; k < n_combs; ++k)
        for (int c = 0; c < cells; ++c) {
            int a = 0, u = 0;
            for (int gi = 0; gi < ng; gi += 2) { a += in[((size_t)k * ng + gi) * cells + c]; u += in[((size_t)k * ng + gi + 1) * cells + c]; }
            for (int f = 0; f < nf; ++f) {                           // training part of fold f = everybody but its own group
                const size_t o = ((size_t)f * n_combs + k) * cells + c;   // model.c:166-168
                counts_aff[o] = a - in[((size_t)k * ng + 2 * f) * cells + c];
                counts_unaff[o] = u - in[((size_t)k * ng + 2 * f + 1) * cells + c];
            }
        }
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

int hpgv_epi_scan_pairs(hpgv_ctx *ctx, int i_begin, int i_end, int subset, double *accuracy, uint16_t *risky_mask,
                        unsigned long long *n_pairs) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    EpiState &E = ctx->epi;
    if (!E.have_folds) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (subset != HPGV_EPI_TESTING && subset != HPGV_EPI_TRAINING) return fail(ctx, HPGV_ERR_INVALID, "subset must be HPGV_EPI_TESTING or HPGV_EPI_TRAINING");
    if (i_begin < 0 || i_end < i_begin || i_end > E.V) return fail(ctx, HPGV_ERR_INVALID, "rows [%d, %d) outside the dataset", i_begin, i_end);
    const unsigned long long V = (unsigned long long)E.V;
    const unsigned long long base = epi_rank(V, (unsigned long long)i_begin);
    const unsigned long long np = (i_end >= E.V ? V * (V - 1) / 2 : epi_rank(V, (unsigned long long)i_end)) - (E.V > 0 ? base : 0);
    if (n_pairs) *n_pairs = E.V > 1 ? np : 0;
    if (E.V < 2 || np == 0 || (!accuracy && !risky_mask)) return HPGV_OK;
    if (!accuracy || !risky_mask) return fail(ctx, HPGV_ERR_INVALID, "accuracy and risky_mask go together");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    const size_t nf = (size_t)E.num_folds;
    double *d_acc = nullptr; uint16_t *d_mask = nullptr;
    HIPCHK(ctx, hipMalloc(&d_acc, nf * np * sizeof(double)));
    hipError_t e = hipMalloc(&d_mask, nf * np * sizeof(uint16_t));
    int rc = HPGV_OK;
    if (e == hipSuccess) {
        rc = subset == HPGV_EPI_TRAINING ? epi_launch_pairs<true>(ctx, i_begin, i_end, d_acc, d_mask, np, base, false, nullptr)
                                         : epi_launch_pairs<false>(ctx, i_begin, i_end, d_acc, d_mask, np, base, false, nullptr);
        if (!rc) e = hipMemcpy(accuracy, d_acc, nf * np * sizeof(double), hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess) e = hipMemcpy(risky_mask, d_mask, nf * np * sizeof(uint16_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_acc);
    if (d_mask) (void)hipFree(d_mask);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, HPGV_ERR_HIP, "epistasis pair scan failed: %s", hipGetErrorString(e));
    return HPGV_OK;
}

int hpgv_epi_rank_pairs(hpgv_ctx *ctx, int subset, int max_ranking_size, int32_t *comb_i, int32_t *comb_j, double *accuracy,
                        uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    return hpgv_epi_rank_pairs_rows(ctx, 0, ctx->epi.V, subset, max_ranking_size, comb_i, comb_j, accuracy, risky_mask, n_ranked, scan_ms);
}

int hpgv_epi_rank_pairs_rows(hpgv_ctx *ctx, int i_begin, int i_end, int subset, int max_ranking_size, int32_t *comb_i,
                             int32_t *comb_j, double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    EpiState &E = ctx->epi;
    if (i_begin < 0 || i_end < i_begin || i_end > E.V || (i_begin % 64 && i_begin != i_end))
        return fail(ctx, HPGV_ERR_INVALID, "rows [%d, %d): the band must lie in the dataset and start on a multiple of 64", i_begin, i_end);
    if (!E.have_folds) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (subset != HPGV_EPI_TESTING && subset != HPGV_EPI_TRAINING) return fail(ctx, HPGV_ERR_INVALID, "subset must be HPGV_EPI_TESTING or HPGV_EPI_TRAINING");
    if (max_ranking_size < 1 || max_ranking_size > 65536 || !comb_i || !comb_j || !accuracy || !risky_mask || !n_ranked)
        return fail(ctx, HPGV_ERR_INVALID, "bad ranking arguments");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    const int nf = E.num_folds, N = max_ranking_size;
    const unsigned cap = (unsigned)std::max<long long>(1ll << 20, 64ll * E.V);      // a band of 64 rows with no threshold yet fits
    if (!E.d_cand || E.cand_cap != cap) {
        if (E.d_cand) (void)hipFree(E.d_cand);
        E.d_cand = nullptr;
        HIPCHK(ctx, hipMalloc(&E.d_cand, (size_t)hpgv::EPI_MAX_FOLDS * cap * sizeof(hpgv::EpiCand)));
        E.cand_cap = cap;
    }
    if (!E.d_cand_count) HIPCHK(ctx, hipMalloc(&E.d_cand_count, hpgv::EPI_MAX_FOLDS * sizeof(unsigned)));
    if (!E.d_thr) HIPCHK(ctx, hipMalloc(&E.d_thr, hpgv::EPI_MAX_FOLDS * sizeof(double)));
    std::vector<std::vector<hpgv::EpiCand>> top((size_t)nf);
    std::vector<double> thr(hpgv::EPI_MAX_FOLDS, -HUGE_VAL);
    std::vector<unsigned> count(hpgv::EPI_MAX_FOLDS);
    std::vector<hpgv::EpiCand> buf;
    auto better = [](const hpgv::EpiCand &a, const hpgv::EpiCand &b) {
        if (a.accuracy != b.accuracy) return a.accuracy > b.accuracy;
        if (a.i != b.i) return a.i < b.i;
        return a.j < b.j;
    };
    EventPair evs;                                                   // destroyed on every return path
    hipEvent_t &ev0 = evs.a, &ev1 = evs.b;
    float total_ms = 0.f;
    if (scan_ms) { HIPCHK(ctx, hipEventCreate(&ev0)); HIPCHK(ctx, hipEventCreate(&ev1)); }
    const long long V = E.V;
    const long long last = std::min<long long>(V - 1, i_end);          // row V - 1 has no pair
    // Bands from the LAST rows up: row r begins V - 1 - r pairs, so the first launches list a few thousand models and leave
    // thresholds behind for the long rows (no pre-pass for starting thresholds; the order of the launches does not show in the
    // ranking: ties go by (i, j)).  Pairs per launch: a small first band, growing while the candidate lists stay short.
    long long band_pairs = std::max<long long>(8192, 2ll * N);
    long long hi = last;
    int rc = HPGV_OK;
    while (hi > i_begin && !rc) {
        long long pairs = 0;
        long long lo = hi;
        // whole blocks of 64 rows (the tile numbering wants bands that start on a multiple of 64), at least one
        while (lo > i_begin) {
            const long long nxt = std::max<long long>(i_begin, (lo - 1) / 64 * 64);
            long long add = 0;
            for (long long r = nxt; r < lo; ++r) add += V - 1 - r;
            if (lo < hi && pairs + add > band_pairs) break;
            pairs += add; lo = nxt;
        }
        const int i = (int)lo, e_row = (int)hi;
        HIPCHK(ctx, hipMemsetAsync(E.d_cand_count, 0, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), nullptr));
        HIPCHK(ctx, hipMemcpyAsync(E.d_thr, thr.data(), hpgv::EPI_MAX_FOLDS * sizeof(double), hipMemcpyHostToDevice, nullptr));
        if (scan_ms) HIPCHK(ctx, hipEventRecord(ev0, nullptr));
        rc = subset == HPGV_EPI_TRAINING ? epi_launch_pairs<true>(ctx, i, e_row, nullptr, nullptr, 0, 0, true, nullptr)
                                         : epi_launch_pairs<false>(ctx, i, e_row, nullptr, nullptr, 0, 0, true, nullptr);
        if (rc) break;
        if (scan_ms) HIPCHK(ctx, hipEventRecord(ev1, nullptr));
        HIPCHK(ctx, hipMemcpy(count.data(), E.d_cand_count, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (scan_ms) { float ms = 0.f; HIPCHK(ctx, hipEventElapsedTime(&ms, ev0, ev1)); total_ms += ms; }
        unsigned worst = 0;
        for (int f = 0; f < nf; ++f) worst = count[(size_t)f] > worst ? count[(size_t)f] : worst;
        if (worst > cap) {                                           // some list overflowed: this band again, in smaller pieces
            if (e_row - i <= 64) { rc = fail(ctx, HPGV_ERR_UNSUPPORTED, "more than %u models of 64 rows reach a fold's threshold: too many SNPs for the candidate lists", cap); break; }
            band_pairs = pairs / 2 > 0 ? pairs / 2 : 1;
            continue;
        }
        for (int f = 0; f < nf && !rc; ++f) {
            const unsigned n = count[(size_t)f];
            if (!n) continue;
            buf.resize(n);
            HIPCHK(ctx, hipMemcpy(buf.data(), E.d_cand + (size_t)f * cap, (size_t)n * sizeof(hpgv::EpiCand), hipMemcpyDeviceToHost));
            auto &t = top[(size_t)f];
            t.insert(t.end(), buf.begin(), buf.end());
            if ((int)t.size() > N) {
                std::partial_sort(t.begin(), t.begin() + N, t.end(), better);
                t.resize((size_t)N);
            } else {
                std::sort(t.begin(), t.end(), better);
            }
            if ((int)t.size() >= N && t.back().accuracy > thr[(size_t)f]) thr[(size_t)f] = t.back().accuracy;
        }
        // (a launch of a few tiles takes as long as one workgroup's whole scan: once thresholds exist and the lists stay
        // nearly empty the bands grow fast)
        if (worst < cap / 64 && band_pairs < (long long)cap * 1024) band_pairs *= 32;
        else if (worst < cap / 8 && band_pairs < (long long)cap * 1024) band_pairs *= 4;
        hi = lo;
    }
    if (rc) return rc;
    for (int f = 0; f < nf; ++f) {
        const auto &t = top[(size_t)f];
        n_ranked[f] = (int32_t)t.size();
        for (size_t k = 0; k < t.size(); ++k) {
            comb_i[(size_t)f * N + k] = t[k].i; comb_j[(size_t)f * N + k] = t[k].j;
            accuracy[(size_t)f * N + k] = t[k].accuracy; risky_mask[(size_t)f * N + k] = t[k].risky;
        }
    }
    if (scan_ms) *scan_ms = total_ms;
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

// ---- order 3 -------------------------------------------------------------------------------------------------

namespace {

template <bool TRAINING>
int epi_launch_triples(hpgv_ctx *ctx, int i_first, int n_i, double *d_acc, uint32_t *d_mask, bool candidates,
                       hpgv::EpiCand3 *d_cand, unsigned cap) {
    EpiState &E = ctx->epi;
    if (int rc = epi_upload_folds<TRAINING>(ctx, nullptr)) return rc;
    // tiles that hold a triple: for the j block jb (rows 4 jb .. 4 jb + 3) the k tiles from the one that holds 4 jb + 1
    // on; a first SNP i takes the j blocks from (i + 1) / 4 on
    const int n_kt = (E.V + hpgv::EPI_TJ - 1) / hpgv::EPI_TJ, n_jb = (E.V + hpgv::EPI_TI - 1) / hpgv::EPI_TI;
    std::vector<unsigned> jbp((size_t)n_jb + 1), rb((size_t)n_i + 1);
    unsigned long long acc = 0;
    for (int jb = 0; jb < n_jb; ++jb) { jbp[(size_t)jb] = (unsigned)acc; acc += (unsigned long long)std::max(0, n_kt - ((4 * jb + 1) >> 6)); }
    jbp[(size_t)n_jb] = (unsigned)acc;
    if (acc > 0x7FFFFFFFull) return fail(ctx, HPGV_ERR_UNSUPPORTED, "too many SNPs for the triple scan");
    unsigned long long total = 0;
    for (int r = 0; r < n_i; ++r) {
        rb[(size_t)r] = (unsigned)total;
        const int jb_min = std::min(n_jb, (i_first + r + 1) >> 2);
        total += acc - jbp[(size_t)jb_min];
    }
    rb[(size_t)n_i] = (unsigned)total;
    if (total > (0x7FFFFFFFull >> 8)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "too many first SNPs for one launch of the triple scan");
    if (total == 0) return HPGV_OK;
    const size_t need = jbp.size() + rb.size();
    if (E.tile_base_cap < need) {
        if (E.d_tile_base) (void)hipFree(E.d_tile_base);
        E.d_tile_base = nullptr; E.tile_base_cap = 0;
        HIPCHK(ctx, hipMalloc(&E.d_tile_base, (need + 64) * sizeof(unsigned)));
        E.tile_base_cap = need + 64;
    }
    unsigned *d_jbp = E.d_tile_base, *d_rb = E.d_tile_base + jbp.size();
    HIPCHK(ctx, hipMemcpyAsync(d_jbp, jbp.data(), jbp.size() * sizeof(unsigned), hipMemcpyHostToDevice, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(d_rb, rb.data(), rb.size() * sizeof(unsigned), hipMemcpyHostToDevice, nullptr));
    const bool balanced = E.nA == E.nU && E.nA < (1 << 22);
    // ranking, classes below 65 536 samples: the cell counts on the matrix cores (hpgv_epi_mfma_kernels.h:
    // k_epi_triples_mfma), tiles of one first SNP x 16 second x 64 third
    if (ctx->epi_triples_mfma && candidates && !d_acc && E.rev_off && E.n_chunks <= hpgv::EPM_MAX_CHUNKS && E.nA < 65536 && E.nU < 65536) {
        const int n_jb16 = (E.V + hpgv::EPM_TI - 1) / hpgv::EPM_TI;
        std::vector<unsigned> jbp16((size_t)n_jb16 + 1), rb16((size_t)n_i + 1);
        unsigned long long acc16 = 0;
        for (int jb = 0; jb < n_jb16; ++jb) { jbp16[(size_t)jb] = (unsigned)acc16; acc16 += (unsigned long long)std::max(0, n_kt - ((hpgv::EPM_TI * jb + 1) >> 6)); }
        jbp16[(size_t)n_jb16] = (unsigned)acc16;
        unsigned long long total16 = 0;
        for (int r = 0; r < n_i; ++r) {
            rb16[(size_t)r] = (unsigned)total16;
            total16 += acc16 - jbp16[(size_t)std::min(n_jb16, (i_first + r + 1) >> 4)];
        }
        rb16[(size_t)n_i] = (unsigned)total16;
        if (total16 == 0) return HPGV_OK;
        if (total16 > (0x7FFFFFFFull >> 8)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "too many first SNPs for one launch of the triple scan");
        HIPCHK(ctx, hipMemcpyAsync(d_jbp, jbp16.data(), jbp16.size() * sizeof(unsigned), hipMemcpyHostToDevice, nullptr));    // (no longer than the lists above: same buffer)
        HIPCHK(ctx, hipMemcpyAsync(d_rb, rb16.data(), rb16.size() * sizeof(unsigned), hipMemcpyHostToDevice, nullptr));
#define HPGV_EPM3_LAUNCH(KK, BAL)                                                                                                               \
        hipLaunchKernelGGL((hpgv::k_epi_triples_mfma<KK, TRAINING, BAL>), dim3((unsigned)total16), dim3(256), 0, nullptr, E.d_planes, E.rev_off, E.W, E.V, i_first, d_rb, n_i, \
                           d_jbp, n_jb16, E.d_chunks, E.n_chunks, E.d_folds, E.num_folds, E.nA, E.nU, E.d_thr, d_cand, E.d_cand_count, cap)
        if (E.num_folds <= 5) { if (balanced) HPGV_EPM3_LAUNCH(5, true); else HPGV_EPM3_LAUNCH(5, false); }
        else if (E.num_folds <= 10) { if (balanced) HPGV_EPM3_LAUNCH(10, true); else HPGV_EPM3_LAUNCH(10, false); }
        else { if (balanced) HPGV_EPM3_LAUNCH(16, true); else HPGV_EPM3_LAUNCH(16, false); }      // one wave per SIMD
#undef HPGV_EPM3_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return HPGV_OK;
    }
    const dim3 grid((unsigned)total);
    // ranking, at most 10 folds, classes below 65 536 samples: the 27 cells nine at a time (hpgv_epi_triples3_kernels.h): three walks
    // over the samples with a third of the state each, three waves per SIMD
    if (ctx->epi_triples_1pass == 1 && candidates && !d_acc && E.num_folds <= 10 && E.nA < 65536 && E.nU < 65536) {
#define HPGV_EPI3B_LAUNCH(KK, BAL)                                                                                                              \
        hipLaunchKernelGGL((hpgv::k_epi_triples3<KK, TRAINING, BAL>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, E.V, i_first, d_rb, n_i, d_jbp, n_jb, \
                           E.d_chunks, E.n_chunks, E.d_folds, E.nA, E.nU, E.d_thr, d_cand, E.d_cand_count, cap)
        if (E.num_folds <= 5) { if (balanced) HPGV_EPI3B_LAUNCH(5, true); else HPGV_EPI3B_LAUNCH(5, false); }
        else { if (balanced) HPGV_EPI3B_LAUNCH(10, true); else HPGV_EPI3B_LAUNCH(10, false); }
#undef HPGV_EPI3B_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return HPGV_OK;
    }
#ifdef HPGV_ABLATION
    // (option epi_triples_1pass = 2: the one-pass kernel it replaced -- all 27 K counts in one lane, one wave per SIMD.  Not for unequal
    // classes above 5 folds: that instantiation does not fit the register file)
    if (ctx->epi_triples_1pass == 2 && candidates && !d_acc && E.num_folds <= 10 && E.nA < 65536 && E.nU < 65536 && (balanced || E.num_folds <= 5)) {
#define HPGV_EPI3_LAUNCH(KK, BAL)                                                                                                               \
        hipLaunchKernelGGL((hpgv::k_epi_triples1<KK, TRAINING, BAL>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, E.V, i_first, d_rb, n_i, d_jbp, n_jb, \
                           E.d_chunks, E.n_chunks, E.d_folds, E.nA, E.nU, E.d_thr, d_cand, E.d_cand_count, cap)
        if (E.num_folds <= 5) { if (balanced) HPGV_EPI3_LAUNCH(5, true); else HPGV_EPI3_LAUNCH(5, false); }
        else HPGV_EPI3_LAUNCH(10, true);                              // (above 5 folds only equal classes come here)
#undef HPGV_EPI3_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return HPGV_OK;
    }
#endif
    if (balanced)
        hipLaunchKernelGGL((hpgv::k_epi_triples<TRAINING, true>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, E.V, i_first, d_rb, n_i, d_jbp, n_jb, E.d_chunks, E.n_chunks,
                           E.num_folds, E.d_folds, E.nA, E.nU, d_acc, d_mask, candidates ? E.d_thr : nullptr, d_cand, E.d_cand_count, cap);
    else
        hipLaunchKernelGGL((hpgv::k_epi_triples<TRAINING, false>), grid, dim3(256), 0, nullptr, E.d_planes, E.W, E.V, i_first, d_rb, n_i, d_jbp, n_jb, E.d_chunks, E.n_chunks,
                           E.num_folds, E.d_folds, E.nA, E.nU, d_acc, d_mask, candidates ? E.d_thr : nullptr, d_cand, E.d_cand_count, cap);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

int epi_triples_check(hpgv_ctx *ctx, int subset) {
    EpiState &E = ctx->epi;
    if (!E.have_folds) return fail(ctx, HPGV_ERR_STATE, "hpgv_epi_set_dataset has not been called");
    if (subset != HPGV_EPI_TESTING && subset != HPGV_EPI_TRAINING) return fail(ctx, HPGV_ERR_INVALID, "subset must be HPGV_EPI_TESTING or HPGV_EPI_TRAINING");
    if (E.nA > 65535 || E.nU > 65535) return fail(ctx, HPGV_ERR_UNSUPPORTED, "the triple scan keeps 16-bit totals: at most 65535 samples per class");
    for (int f = 0; f < E.num_folds; ++f)
        if (E.group_size[(size_t)2 * f] + E.group_size[(size_t)2 * f + 1] == 0)
            return fail(ctx, HPGV_ERR_UNSUPPORTED, "fold %d has no samples", f);
    return HPGV_OK;
}

}  // namespace

int hpgv_epi_scan_triples(hpgv_ctx *ctx, int subset, double *accuracy, uint32_t *risky_mask) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    int rc = epi_triples_check(ctx, subset);
    if (rc) return rc;
    EpiState &E = ctx->epi;
    if (!accuracy || !risky_mask) return fail(ctx, HPGV_ERR_INVALID, "outputs are NULL");
    if (E.V > 256) return fail(ctx, HPGV_ERR_UNSUPPORTED, "the dense triple scan is for small sets (<= 256 SNPs); use hpgv_epi_rank_triples");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    const size_t V = (size_t)E.V, total = (size_t)E.num_folds * V * V * V;
    if (total == 0) return HPGV_OK;
    double *d_acc = nullptr; uint32_t *d_mask = nullptr;
    HIPCHK(ctx, hipMalloc(&d_acc, total * sizeof(double)));
    hipError_t e = hipMalloc(&d_mask, total * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(d_acc, 0xFF, total * sizeof(double));            // cells that are no triple i < j < k read as NaN
    if (e == hipSuccess) e = hipMemset(d_mask, 0, total * sizeof(uint32_t));
    if (e == hipSuccess) {
        rc = subset == HPGV_EPI_TRAINING ? epi_launch_triples<true>(ctx, 0, E.V, d_acc, d_mask, false, nullptr, 0)
                                         : epi_launch_triples<false>(ctx, 0, E.V, d_acc, d_mask, false, nullptr, 0);
        if (!rc) e = hipMemcpy(accuracy, d_acc, total * sizeof(double), hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess) e = hipMemcpy(risky_mask, d_mask, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_acc);
    if (d_mask) (void)hipFree(d_mask);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, HPGV_ERR_HIP, "epistasis triple scan failed: %s", hipGetErrorString(e));
    return HPGV_OK;
}

int hpgv_epi_rank_triples(hpgv_ctx *ctx, int subset, int max_ranking_size, int32_t *comb_i, int32_t *comb_j, int32_t *comb_k,
                          double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    const hpgv_ctx *c = first_member(ctx);
    return hpgv_epi_rank_triples_rows(ctx, 0, c ? c->epi.V : 0, subset, max_ranking_size, comb_i, comb_j, comb_k, accuracy, risky_mask, n_ranked, scan_ms);
}

int hpgv_epi_rank_triples_rows(hpgv_ctx *ctx, int i_begin, int i_end, int subset, int max_ranking_size, int32_t *comb_i, int32_t *comb_j,
                               int32_t *comb_k, double *accuracy, uint32_t *risky_mask, int32_t *n_ranked, float *scan_ms) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    int rc = epi_triples_check(ctx, subset);
    if (rc) return rc;
    EpiState &E = ctx->epi;
    if (i_begin < 0 || i_end < i_begin || i_end > E.V) return fail(ctx, HPGV_ERR_INVALID, "first SNPs [%d, %d) outside the dataset", i_begin, i_end);
    if (max_ranking_size < 1 || max_ranking_size > 65536 || !comb_i || !comb_j || !comb_k || !accuracy || !risky_mask || !n_ranked)
        return fail(ctx, HPGV_ERR_INVALID, "bad ranking arguments");
    DeviceGuard g(ctx->device);
    std::lock_guard<std::mutex> lk(ctx->epi_mu);
    const int nf = E.num_folds, N = max_ranking_size;
    const long long V = E.V;
    // one first SNP i gives (V - i - 1)(V - i - 2) / 2 triples; a launch without thresholds lists them all
    const unsigned long long per_i0 = V > 2 ? (unsigned long long)(V - 1) * (unsigned long long)(V - 2) / 2 : 1;
    if (per_i0 > (1ull << 28)) return fail(ctx, HPGV_ERR_UNSUPPORTED, "too many SNPs for the candidate lists of the triple scan");
    const unsigned cap = (unsigned)std::max<unsigned long long>(1ull << 20, per_i0);
    if (!E.d_cand3 || E.cand3_cap < (size_t)nf * cap) {              // (kept between calls: 240 MB at 1 024 SNPs)
        if (E.d_cand3) (void)hipFree(E.d_cand3);
        E.d_cand3 = nullptr; E.cand3_cap = 0;
        HIPCHK(ctx, hipMalloc(&E.d_cand3, (size_t)nf * cap * sizeof(hpgv::EpiCand3)));
        E.cand3_cap = (size_t)nf * cap;
    }
    hpgv::EpiCand3 *d_cand = E.d_cand3;
    if (!E.d_cand_count) HIPCHK(ctx, hipMalloc(&E.d_cand_count, hpgv::EPI_MAX_FOLDS * sizeof(unsigned)));
    if (!E.d_thr) HIPCHK(ctx, hipMalloc(&E.d_thr, hpgv::EPI_MAX_FOLDS * sizeof(double)));
    std::vector<std::vector<hpgv::EpiCand3>> top((size_t)nf);
    std::vector<double> thr(hpgv::EPI_MAX_FOLDS, -HUGE_VAL);
    std::vector<unsigned> count(hpgv::EPI_MAX_FOLDS);
    std::vector<hpgv::EpiCand3> buf;
    auto better = [](const hpgv::EpiCand3 &a, const hpgv::EpiCand3 &b) {
        if (a.accuracy != b.accuracy) return a.accuracy > b.accuracy;
        if (a.i != b.i) return a.i < b.i;
        if (a.j != b.j) return a.j < b.j;
        return a.k < b.k;
    };
    EventPair evs;                                                   // destroyed on every return path
    hipEvent_t &ev0 = evs.a, &ev1 = evs.b;
    float total_ms = 0.f;
    if (scan_ms) { HIPCHK(ctx, hipEventCreate(&ev0)); HIPCHK(ctx, hipEventCreate(&ev1)); }
    // a launch stays below 2^23 tiles: at most n_jb * n_kt tiles per first SNP
    const long long per_row_tiles = ((V + hpgv::EPI_TI - 1) / hpgv::EPI_TI) * ((V + hpgv::EPI_TJ - 1) / hpgv::EPI_TJ) + 1;
    const int max_rows = (int)std::max<long long>(1, (1ll << 23) / per_row_tiles);
    const long long i_last = std::min<long long>(V - 2, i_end);    // first SNPs V - 2 and V - 1 begin no triple
    // From the LAST first SNP down: a first SNP i begins (V - i - 1)(V - i - 2) / 2 triples, so the first launches list a handful
    // of models each and leave thresholds behind for the long rows (from the first SNP up, the very first launch listed every
    // triple of SNP 0 -- half a million models per fold at 1 024 SNPs, 125 MB over the bus and a sort: 50 ms of a 130 ms call).
    // The order of the launches does not show in the ranking: ties go by (i, j, k).
    long long hi = i_last;
    int step = 1;
    while (hi > i_begin && !rc) {
        const int n_i = (int)std::min<long long>(std::min<long long>(step, max_rows), hi - i_begin);
        const int i = (int)(hi - n_i);
        HIPCHK(ctx, hipMemsetAsync(E.d_cand_count, 0, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), nullptr));
        HIPCHK(ctx, hipMemcpyAsync(E.d_thr, thr.data(), hpgv::EPI_MAX_FOLDS * sizeof(double), hipMemcpyHostToDevice, nullptr));
        if (scan_ms) HIPCHK(ctx, hipEventRecord(ev0, nullptr));
        rc = subset == HPGV_EPI_TRAINING ? epi_launch_triples<true>(ctx, i, n_i, nullptr, nullptr, true, d_cand, cap)
                                         : epi_launch_triples<false>(ctx, i, n_i, nullptr, nullptr, true, d_cand, cap);
        if (rc) break;
        if (scan_ms) HIPCHK(ctx, hipEventRecord(ev1, nullptr));
        HIPCHK(ctx, hipMemcpy(count.data(), E.d_cand_count, hpgv::EPI_MAX_FOLDS * sizeof(unsigned), hipMemcpyDeviceToHost));
        if (scan_ms) { float ms = 0.f; HIPCHK(ctx, hipEventElapsedTime(&ms, ev0, ev1)); total_ms += ms; }
        unsigned worst = 0;
        for (int f = 0; f < nf; ++f) worst = std::max(worst, count[(size_t)f]);
        if (worst > cap) {                                           // a list overflowed: the same first SNPs again, fewer at a time
            if (n_i <= 1) { rc = fail(ctx, HPGV_ERR_UNSUPPORTED, "more than %u models of one first SNP reach a fold's threshold", cap); break; }
            step = std::max(1, n_i / 2);
            continue;
        }
        for (int f = 0; f < nf && !rc; ++f) {
            const unsigned n = count[(size_t)f];
            if (!n) continue;
            buf.resize(n);
            HIPCHK(ctx, hipMemcpy(buf.data(), d_cand + (size_t)f * cap, (size_t)n * sizeof(hpgv::EpiCand3), hipMemcpyDeviceToHost));
            auto &t = top[(size_t)f];
            t.insert(t.end(), buf.begin(), buf.end());
            if ((int)t.size() > N) { std::partial_sort(t.begin(), t.begin() + N, t.end(), better); t.resize((size_t)N); }
            else std::sort(t.begin(), t.end(), better);
            if ((int)t.size() >= N && t.back().accuracy > thr[(size_t)f]) thr[(size_t)f] = t.back().accuracy;
        }
        hi = i;
        if (worst < cap / 8 && step < 4096) step *= 2;
    }
    if (rc) return rc;
    for (int f = 0; f < nf; ++f) {
        const auto &t = top[(size_t)f];
        n_ranked[f] = (int32_t)t.size();
        for (size_t e = 0; e < t.size(); ++e) {
            comb_i[(size_t)f * N + e] = t[e].i; comb_j[(size_t)f * N + e] = t[e].j; comb_k[(size_t)f * N + e] = t[e].k;
            accuracy[(size_t)f * N + e] = t[e].accuracy; risky_mask[(size_t)f * N + e] = t[e].risky;
        }
    }
    if (scan_ms) *scan_ms = total_ms;
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}
