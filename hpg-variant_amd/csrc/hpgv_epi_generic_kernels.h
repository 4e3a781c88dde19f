// hpgv_epi_generic_kernels.h -- the MDR model of process_set_of_combinations (epistasis.c:14-95) for combinations of ANY
// order the caller lists (2 <= order <= 5): combination_counts_all_folds (model.c:76-206) over 3^order cells, the high-risk
// cells of every fold from its training counts (mdr_high_risk_combinations2, mdr.c:45-76), the confusion matrix on the
// chosen part and the balanced accuracy (test_model / evaluate_model, model.c:320-476).
//
// The pair and triple scans (hpgv_epi_kernels.h) are built around their tile shape; this is the form for every other order
// (`--order` is any integer in the reference, main_epistasis.c:128,142), and a second, independent implementation of
// orders 2 and 3 that the tests hold against the scans.
//
// Work split: ONE LANE PER CELL.  A combination of order n has 3^n cells (9, 27, 81, 243); a workgroup of 256 threads takes
// floor(256 / 3^n) combinations (28, 9, 3, 1).  A lane knows its n plane rows (SNP s of the combination, genotype = digit s
// of the cell, last SNP fastest: get_genotype_combinations, dataset.c:170-200) and walks the samples' words group by
// group -- the (fold, class) groups are contiguous word runs of the planes (hpgv_epi_set_folds) -- four words per step:
// n 16-byte loads (lanes of a wave read few distinct rows: the loads are served by the L1), n - 1 ANDs and one
// v_bcnt per word.  The count of a cell in a group therefore ends in ONE lane's register with no reduction across lanes;
// a lane keeps its K (fold) x 2 (class) in-fold counts packed two to a register.  The evaluation of a fold needs sums over
// the cells of a combination (TP, FP) and the OR of their high-risk bits: LDS atomics, one barrier, then one thread per
// (combination, fold) forms the accuracy exactly as the scans do (same quotient sequence).
#pragma once
#include "hpgv_epi_kernels.h"

namespace hpgv {

constexpr int EPI_MASK_WORDS = 8;                                    // 243 cells of order 5 in 8 x 32 bits

struct EpiCandN {                                                    // a listed combination that reached a fold's threshold
    double accuracy;
    uint32_t index;                                                  // its index in the launch's list
    uint32_t pad;
    uint32_t risky[EPI_MASK_WORDS];                                  // bit c = cell c is high risk
};

template <int ORDER> struct EpiCells { static constexpr int value = 3 * EpiCells<ORDER - 1>::value; };
template <> struct EpiCells<0> { static constexpr int value = 1; };

// counts_out (may be NULL): the in-fold counts, counts_out[(comb * n_groups + g) * cells + c] (what k_epi_counts gives for
// orders 2 and 3).  With folds == NULL only the counts are made.
template <int ORDER, bool TRAINING>
__global__ void __launch_bounds__(256) k_epi_combs(const uint32_t *__restrict__ planes, int W, const int32_t *__restrict__ combs, int n_combs,
                                                   const uint32_t *__restrict__ group_w0 /* n_groups + 1 */, int num_folds,
                                                   const EpiFold *__restrict__ folds, int n_affected, int n_unaffected,
                                                   int32_t *__restrict__ counts_out,
                                                   double *__restrict__ acc_out, uint32_t *__restrict__ mask_out,
                                                   const double *__restrict__ thr, EpiCandN *__restrict__ cand,
                                                   unsigned *__restrict__ cand_count, unsigned cand_cap) {
    constexpr int CELLS = EpiCells<ORDER>::value, CPW = 256 / CELLS;      // combinations per workgroup
    __shared__ int s_tp[CPW][EPI_MAX_FOLDS], s_fp[CPW][EPI_MAX_FOLDS];
    __shared__ uint32_t s_mask[CPW][EPI_MAX_FOLDS][EPI_MASK_WORDS];
    const int t = threadIdx.x, lc = t / CELLS, cell = t - lc * CELLS;
    const long comb = (long)blockIdx.x * CPW + lc;
    const bool live = lc < CPW && comb < n_combs;
    for (int k = t; k < CPW * EPI_MAX_FOLDS; k += 256) { (&s_tp[0][0])[k] = 0; (&s_fp[0][0])[k] = 0; }
    for (int k = t; k < CPW * EPI_MAX_FOLDS * EPI_MASK_WORDS; k += 256) (&s_mask[0][0][0])[k] = 0;
    __syncthreads();

    // this lane's plane rows: SNP s of the combination, genotype = digit s of the cell (the last SNP varies fastest)
    const uint32_t *row[ORDER];
    {
        int c = cell;
        #pragma unroll
        for (int s = ORDER - 1; s >= 0; --s) {
            const int digit = c % 3; c /= 3;
            const int snp = live ? combs[comb * ORDER + s] : 0;
            row[s] = planes + ((size_t)snp * 3 + (size_t)digit) * (size_t)W;
        }
    }
    // in-fold counts: fold f's affected (low half) and unaffected (high half) samples in this cell
    uint32_t in[EPI_MAX_FOLDS];
    #pragma unroll
    for (int f = 0; f < EPI_MAX_FOLDS; ++f) {
        in[f] = 0;
        if (f < num_folds && live) {
            #pragma unroll
            for (int cls = 0; cls < 2; ++cls) {
                uint32_t cnt = 0;
                const uint32_t w_lo = group_w0[2 * f + cls], w_hi = group_w0[2 * f + cls + 1];      // whole 4-word steps
                for (uint32_t w = w_lo; w < w_hi; w += 4) {
                    uint4 x = *reinterpret_cast<const uint4 *>(row[0] + w);
                    #pragma unroll
                    for (int s = 1; s < ORDER; ++s) {
                        const uint4 y = *reinterpret_cast<const uint4 *>(row[s] + w);
                        x.x &= y.x; x.y &= y.y; x.z &= y.z; x.w &= y.w;
                    }
                    cnt = bcnt_acc(x.x, cnt); cnt = bcnt_acc(x.y, cnt); cnt = bcnt_acc(x.z, cnt); cnt = bcnt_acc(x.w, cnt);
                }
                in[f] |= cnt << (16 * cls);
                if (counts_out) counts_out[((size_t)comb * (size_t)(2 * num_folds) + (size_t)(2 * f + cls)) * CELLS + cell] = (int32_t)cnt;
            }
        }
    }
    if (!folds) return;                                              // (uniform: counts only)

    // ---- per fold: training counts of this cell, its high-risk bit, what it adds to TP / FP ----
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    uint32_t tot = 0;                                                // halves cannot carry: a class holds fewer than 65536 samples
    #pragma unroll
    for (int f = 0; f < EPI_MAX_FOLDS; ++f) tot += in[f];
    #pragma unroll
    for (int f = 0; f < EPI_MAX_FOLDS; ++f) {
        if (f >= num_folds || !live) continue;
        if (folds[f].test_a < 0) continue;
        const uint32_t tr = tot - in[f];
        const int tr_a = (int)(tr & 0xFFFFu), tr_u = (int)(tr >> 16);
        if (mdr_high_risk<false>(tr_a, tr_u, ratio, f_na, f_nu)) {
            const int add_a = TRAINING ? tr_a : (int)(in[f] & 0xFFFFu), add_u = TRAINING ? tr_u : (int)(in[f] >> 16);
            if (add_a) atomicAdd(&s_tp[lc][f], add_a);
            if (add_u) atomicAdd(&s_fp[lc][f], add_u);
            atomicOr(&s_mask[lc][f][cell >> 5], 1u << (cell & 31));
        }
    }
    __syncthreads();
    // ---- one thread per (combination of the workgroup, fold): the confusion matrix's accuracy ----
    for (int k = t; k < CPW * EPI_MAX_FOLDS; k += 256) {
        const int c2 = k / EPI_MAX_FOLDS, f = k - c2 * EPI_MAX_FOLDS;
        const long cb = (long)blockIdx.x * CPW + c2;
        if (cb >= n_combs || f >= num_folds) continue;
        const EpiFold fo = folds[f];
        if (fo.test_a < 0) continue;
        const int tp = s_tp[c2][f], fp = s_fp[c2][f];
        const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
        // evaluate_model BA (model.c:466-467), the quotients formed as in k_epi_pairs (Markstein: the correctly rounded x / y)
        const double TP = (double)tp, TN = (double)(size_u - fp), ya = (double)size_a, yu = (double)size_u;
        double qa = TP * fo.inv_a, qu = TN * fo.inv_u;
        qa = __builtin_fma(__builtin_fma(-qa, ya, TP), fo.inv_a, qa);
        qu = __builtin_fma(__builtin_fma(-qu, yu, TN), fo.inv_u, qu);
        const double acc = (qa + qu) / 2;
        if (acc_out) {
            acc_out[(size_t)cb * (size_t)num_folds + (size_t)f] = acc;
            if (mask_out)
                for (int w = 0; w < EPI_MASK_WORDS; ++w) mask_out[((size_t)cb * (size_t)num_folds + (size_t)f) * EPI_MASK_WORDS + w] = s_mask[c2][f][w];
        }
        if (cand && acc >= thr[f]) {                                 // (a NaN accuracy ranks nowhere)
            const unsigned slot = atomicAdd(&cand_count[f], 1u);
            if (slot < cand_cap) {
                EpiCandN e;
                e.accuracy = acc; e.index = (uint32_t)cb; e.pad = 0;
                for (int w = 0; w < EPI_MASK_WORDS; ++w) e.risky[w] = s_mask[c2][f][w];
                cand[(size_t)f * cand_cap + slot] = e;
            }
        }
    }
}

}  // namespace hpgv
