// hpgv_statsall_capi.hip -- the launcher of k_stats_all2 (hpgv_statsall_kernels.h): its own translation unit, because the
// kernel is instantiated per (chunks per thread, masked group sets, Mendel on / off).
#include "hpgv_internal.h"
#include "hpgv_batch_kernels.h"
#include "hpgv_statsall_kernels.h"

namespace {

// what = 0: launch; 1: workgroups of this instantiation a compute unit holds (returned)
template <int CPT, int NM>
int run_cn(int what, bool mendel, unsigned grid, unsigned bs, size_t lds, hipStream_t st, const hpgv::StatsAllArgs &A, const hpgv::StatsAll2Cfg &G) {
    if (what == 1) {
        int n = 0;
        const hipError_t e = mendel ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hpgv::k_stats_all2<CPT, NM, true>, (int)bs, lds)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hpgv::k_stats_all2<CPT, NM, false>, (int)bs, lds);
        if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
        return n;
    }
    if (mendel) hipLaunchKernelGGL((hpgv::k_stats_all2<CPT, NM, true>), dim3(grid), dim3(bs), lds, st, A, G);
    else        hipLaunchKernelGGL((hpgv::k_stats_all2<CPT, NM, false>), dim3(grid), dim3(bs), lds, st, A, G);
    return 0;
}
template <int CPT>
int run_c(int nm, int what, bool mendel, unsigned grid, unsigned bs, size_t lds, hipStream_t st, const hpgv::StatsAllArgs &A, const hpgv::StatsAll2Cfg &G) {
    switch (nm) {
        case 0: return run_cn<CPT, 0>(what, mendel, grid, bs, lds, st, A, G);
        case 1: return run_cn<CPT, 1>(what, mendel, grid, bs, lds, st, A, G);
        case 2: return run_cn<CPT, 2>(what, mendel, grid, bs, lds, st, A, G);
        default: return run_cn<CPT, 3>(what, mendel, grid, bs, lds, st, A, G);
    }
}
int run(int cpt, int nm, int what, bool mendel, unsigned grid, unsigned bs, size_t lds, hipStream_t st, const hpgv::StatsAllArgs &A, const hpgv::StatsAll2Cfg &G) {
    switch (cpt) {
        case 1: return run_c<1>(nm, what, mendel, grid, bs, lds, st, A, G);
        case 2: return run_c<2>(nm, what, mendel, grid, bs, lds, st, A, G);
        case 3: return run_c<3>(nm, what, mendel, grid, bs, lds, st, A, G);
        default: return run_c<4>(nm, what, mendel, grid, bs, lds, st, A, G);
    }
}

}  // namespace

// 0: launched; 1: this batch is not one the kernel takes (the caller runs k_stats_all); the rows of A are set by the caller;
// cnt_buf / cnt_cap: device scratch of the calling slot for the rows' counters (grown here)
int hpgv_launch_stats_all2(hpgv_ctx *ctx, hpgv::StatsAllArgs &A, void **cnt_buf, size_t *cnt_cap, hipStream_t st) {
    const int ns = A.n_samples, chunks = (ns + 15) / 16;
    if (ns <= 0 || chunks * 16 > 16384) return 1;                  // (the class buffers' fixed LDS offsets; columns as 16-bit indices)
    if (((uintptr_t)A.src & 15) || (A.src_pitch & 15) || A.src_pitch < (size_t)chunks * 16) return 1;
    const bool mendel = A.mendel_errors || A.child_errors;
    if (mendel && A.pchunks <= 0) return 1;
    hpgv::StatsAll2Cfg G = {nullptr, 0, 0, nullptr};
    if (A.group_out) {
        const int ng = A.n_groups;
        if (!ctx->d_group_of_col || ng < 1) return 1;
        G.derive_last = ctx->all_grouped ? 1 : 0;
        G.n_masked = ng - G.derive_last;
        if (G.n_masked > 3) return 1;
        G.group_of_col = ctx->d_group_of_col;
    }
    // chunks per thread (c) and threads per workgroup (the fewest that cover the row with c chunks each; with trios, sixteen
    // of them per thread at most): the pair that keeps the most USEFUL threads on a compute unit -- workgroups the unit holds
    // (the kernel's registers decide) x threads x the share of their chunks that exist.  10 k samples = 625 chunks with
    // trios and three groups: 3 x 256 threads 155 us; 2 x 320 threads, which wastes fewer chunks, 196 - 217 us.
    int cpt = 0, per_cu = 1;
    unsigned bs = 0;
    double best = 0.0;
    const size_t lds32 = hpgv::stats_all2_lds(32);
#ifdef HPGV_ABLATION
    const long fb = ctx->stats_bs;                                  // tuning: threads per workgroup (0: chosen below)
#endif
    for (int c = 1; c <= 4; ++c) {
        // with trios (a barrier per row) 1, 2, 4 or 8 waves: a workgroup of 5 waves puts two on one SIMD, and every row waits
        // for that SIMD; without, the fewest waves that cover the row
        unsigned b = (unsigned)(((chunks + c - 1) / c + 63) / 64 * 64);
        if (mendel) { b = 64; while (b < 512 && (int)b * c < chunks) b *= 2; }
        if ((int)b * c < chunks) continue;
#ifdef HPGV_ABLATION
        if (fb) { b = (unsigned)fb; if (b < 64 || b % 64 || (int)((chunks + b - 1) / b) != c) continue; }
#endif
        if (b > 512 || (mendel && (long)b * 16 < (long)A.n_trios)) continue;
        const int occ = run(c, G.n_masked, 1, mendel, 0, b, lds32, st, A, G);
        const double useful = (double)occ * (double)chunks / (double)c;      // = occ x b x chunks / (b x c)
        if (occ > 0 && useful > best * 1.02) { best = useful; cpt = c; bs = b; per_cu = occ; }
    }
    if (!cpt) return 1;
#ifdef HPGV_ABLATION
    if (ctx->stats_debug) fprintf(stderr, "k_stats_all2<%d, %d, %d>: %u threads, %d workgroups per unit\n", cpt, G.n_masked, (int)mendel, bs, per_cu);
#endif
    // the band length: the grid is ONE round of the workgroups the chip holds (or k rounds, bands of at most 32 rows).  A band's
    // end -- its column counters' atomics -- costs as much as several rows, and a round that is not full leaves units idle
    // (16 000 rows of 10 k samples, counters + per-sample missing: 8 / 21 / 42 rows per band 105 / 65 / 84 us).
#ifdef HPGV_ABLATION
    if (!ctx->stats_rows)
#endif
    {
        const long slots = (long)per_cu * ctx->n_cus;
        long k = 1;
        while (((long)A.n_variants + slots * k - 1) / (slots * k) > 32) ++k;
        const long rows = ((long)A.n_variants + slots * k - 1) / (slots * k);
        A.rows_per_block = (int)(rows < 1 ? 1 : rows);
    }
    const size_t lds = hpgv::stats_all2_lds(A.rows_per_block);
    if (lds > 64 * 1024) return 1;
    {   // the rows' packed counters between the two kernels (scratch of the calling slot)
        const size_t need = (size_t)A.n_variants * (size_t)hpgv::STATS2_W * sizeof(uint32_t);
        if (*cnt_cap < need) {
            if (*cnt_buf) { (void)hipFree(*cnt_buf); *cnt_buf = nullptr; *cnt_cap = 0; }
            if (hipMalloc(cnt_buf, need + need / 4) != hipSuccess) { (void)hipGetLastError(); return 1; }
            *cnt_cap = need + need / 4;
        }
        G.row_counters = (uint32_t *)*cnt_buf;
    }
    const unsigned grid = (unsigned)((A.n_variants + A.rows_per_block - 1) / A.rows_per_block);
    (void)run(cpt, G.n_masked, 0, mendel, grid, bs, lds, st, A, G);
    const long recs = (long)A.n_variants * (1 + (A.group_out ? G.n_masked + G.derive_last : 0));
    hipLaunchKernelGGL(hpgv::k_stats_all2_records, dim3((unsigned)((recs + 255) / 256)), dim3(256), 0, st, (const uint32_t *)G.row_counters,
                       A.n_variants, G.n_masked, G.derive_last, A.out, A.group_out, A.mendel_errors);
    return 0;
}

// k_assoc_rows on the tokenizer's raw rows (0 = launched, 1 = not a batch it takes: the caller runs k_batch)
int hpgv_launch_assoc_rows(hpgv_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int n_variants, const uint8_t *d_is_x, int32_t *d_counts, hipStream_t st) {
    const int ns = ctx->assoc.n_samples, chunks = (ns + 15) / 16;
    if (ns <= 0 || ns > 65535 || !ctx->d_cond) return 1;
    if (((uintptr_t)d_src & 15) || (src_pitch & 15) || src_pitch < (size_t)chunks * 16 || ((uintptr_t)d_counts & 15)) return 1;
    int cpt = 0;
    unsigned bs = 0;
    for (int c = 1; c <= 4 && !cpt; ++c) {                          // the fewest chunks per thread that 512 threads cover
        const unsigned b = (unsigned)(((chunks + c - 1) / c + 63) / 64 * 64);
        if (b <= 512) { cpt = c; bs = b; }
    }
    if (!cpt) return 1;
    // one round of workgroups over the chip, bands of at most 32 rows (as k_stats_all2)
    const long slots = 4L * ctx->n_cus;
    long k = 1;
    while (((long)n_variants + slots * k - 1) / (slots * k) > 32) ++k;
    long rows = ((long)n_variants + slots * k - 1) / (slots * k);
    if (rows < 1) rows = 1;
    const unsigned grid = (unsigned)((n_variants + rows - 1) / rows);
    const size_t lds = (size_t)rows * 16 + 16;
    switch (cpt) {
        case 1: hipLaunchKernelGGL(hpgv::k_assoc_rows<1>, dim3(grid), dim3(bs), lds, st, d_src, src_pitch, n_variants, ns, (int)rows, d_is_x, (const uint8_t *)ctx->d_cond, (int4 *)d_counts); break;
        case 2: hipLaunchKernelGGL(hpgv::k_assoc_rows<2>, dim3(grid), dim3(bs), lds, st, d_src, src_pitch, n_variants, ns, (int)rows, d_is_x, (const uint8_t *)ctx->d_cond, (int4 *)d_counts); break;
        case 3: hipLaunchKernelGGL(hpgv::k_assoc_rows<3>, dim3(grid), dim3(bs), lds, st, d_src, src_pitch, n_variants, ns, (int)rows, d_is_x, (const uint8_t *)ctx->d_cond, (int4 *)d_counts); break;
        default: hipLaunchKernelGGL(hpgv::k_assoc_rows<4>, dim3(grid), dim3(bs), lds, st, d_src, src_pitch, n_variants, ns, (int)rows, d_is_x, (const uint8_t *)ctx->d_cond, (int4 *)d_counts); break;
    }
    return 0;
}
