// hpgv_epi_kernels.h -- CDNA4 (gfx950) kernels of the epistasis / MDR counting path
// (src/gwas/epistasis/model.c:76-206 combination_counts[_all_folds], mdr.c:45-76,
// model.c:320-476 test_model / confusion_matrix / evaluate_model, driven per combination by
// epistasis.c:14-95).
//
// Data: the vcf2epi dataset (one row per SNP, cases first, codes 0/1/2, anything else =
// missing) is turned once into three BIT PLANES per SNP ("genotype is 0 / 1 / 2"), with the
// samples re-ordered so that every (fold, class) group -- the samples of one class whose
// TESTING fold is f -- is a contiguous run of whole 32-bit words (runs are padded to 4 words
// with zero bits, which no plane counts).  The cell counts the reference gets from k passes
// of byte masks ANDed with k fold masks then come from ONE pass: popcount(plane_i[a] &
// plane_j[b]) over a group's words is that group's count of cell (a, b); the training counts
// of fold f are the totals minus the fold's own group.
//
// This path is a dense binary contraction (V x V pairs x N samples): VALU bound, 18 integer
// operations (9 v_and + 9 v_bcnt with accumulate) per pair and 32 samples.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

constexpr int EPI_CH = 32;            // words of one LDS chunk (1024 samples)
constexpr int EPI_TJ = 64;            // tile: 64 columns (one per lane) ...
constexpr int EPI_TI = 4;             // ... x 4 rows (one per wave)
constexpr int EPI_MAX_FOLDS = 16;

struct EpiChunk {                     // one staging block: up to EPI_CH consecutive words, possibly of several groups
    uint32_t w0;                      // first word
    uint32_t nw;                      // words (multiple of 8, <= EPI_CH)
    uint64_t flush;                   // byte s = the group (fold * 2 + class, class 0 = affected) that ends with the
                                      // block's 4-word step s, or 0xFF when no group ends there
};

struct EpiFold {                      // per fold constants of the evaluation
    int32_t test_a, test_u;           // samples of the fold's testing part (affected, unaffected); test_a < 0: fold unused
    double inv_a, inv_u;              // RN(1 / size) of the evaluated part, for the two divisions of the balanced accuracy
};

struct EpiCand {                      // a model that reached a fold's current threshold
    double accuracy;
    int32_t i, j;
    uint32_t risky;                   // bit c = cell c is high risk
    uint32_t pad;
};

// ---------------------------------------------------------------------------
// dataset rows -> bit planes.  planes[(snp * 3 + g) * W + w]; src_of_pos[p] = dataset column of
// the sample at bit position p, or -1 for a pad bit.  One workgroup per SNP row (rows >= V are zero).
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_epi_planes_gather(const uint8_t *__restrict__ data, int n_variants, int n_samples,
                                                     const int32_t *__restrict__ src_of_pos, int W,
                                                     uint32_t *__restrict__ planes, unsigned *__restrict__ any_missing) {
    const int snp = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *out = planes + (size_t)snp * 3 * W;
    const uint8_t *row = data + (size_t)snp * n_samples;
    bool any = false;
    for (int w2 = wave; w2 * 2 < W; w2 += 4) {                       // 64 bit positions per wave step
        const int p = w2 * 64 + lane;
        uint32_t g = 255;
        bool missing = false;
        if (snp < n_variants && p < W * 32) {
            const int s = src_of_pos[p];
            if (s >= 0) { g = row[s]; missing = g > 2; }
        }
        const unsigned long long b0 = __ballot(g == 0), b1 = __ballot(g == 1), b2 = __ballot(g == 2);
        any |= __ballot(missing) != 0ull;                            // a call that is none of 0 / 1 / 2
        if (lane == 0) {
            out[0 * W + 2 * w2] = (uint32_t)b0; out[1 * W + 2 * w2] = (uint32_t)b1; out[2 * W + 2 * w2] = (uint32_t)b2;
            if (2 * w2 + 1 < W) {
                out[0 * W + 2 * w2 + 1] = (uint32_t)(b0 >> 32); out[1 * W + 2 * w2 + 1] = (uint32_t)(b1 >> 32);
                out[2 * W + 2 * w2 + 1] = (uint32_t)(b2 >> 32);
            }
        }
    }
    if (any && lane == 0) atomicOr(any_missing, 1u);
}

// The same with the SNP's row brought into LDS first (rows of at most 65 000 samples): the samples are taken in (fold, class)
// order, i.e. as a gather over the row.  The row is read in aligned dwords (its edges byte by
// byte) into an image that keeps the row's alignment: row byte p sits at lds[a0 + p].
static __global__ void __launch_bounds__(256) k_epi_planes(const uint8_t *__restrict__ data, int n_variants, int n_samples,
                                                     const int32_t *__restrict__ src_of_pos, int W,
                                                     uint32_t *__restrict__ planes, unsigned *__restrict__ any_missing) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_row[];  // n_samples + 8 bytes
    const int snp = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t *out = planes + (size_t)snp * 3 * W;
    int a0 = 0;
    if (snp < n_variants) {
        const uint8_t *row = data + (size_t)snp * n_samples;
        a0 = (int)((uintptr_t)row & 3u);
        const int head = a0 ? (4 - a0 < n_samples ? 4 - a0 : n_samples) : 0;             // bytes in front of the first aligned dword
        const int n_dw = (n_samples - head) >> 2, tail0 = head + 4 * n_dw;
        if (t < head) s_row[a0 + t] = row[t];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(row + head);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s_row + a0 + head);                 // (a0 + head is 0 or 4)
        for (int q = t; q < n_dw; q += 256) dst[q] = src[q];
        if (t < n_samples - tail0) s_row[a0 + tail0 + t] = row[tail0 + t];
    }
    __syncthreads();
    bool any = false;                                                // (one atomic per wave at most: one per 64 positions, all on one word, took 13 of this kernel's 13.8 ms)
    for (int w2 = wave; w2 * 2 < W; w2 += 4) {                       // 64 bit positions per wave step
        const int p = w2 * 64 + lane;
        uint32_t g = 255;
        bool missing = false;
        if (snp < n_variants && p < W * 32) {
            const int s = src_of_pos[p];
            if (s >= 0) { g = s_row[a0 + s]; missing = g > 2; }
        }
        const unsigned long long b0 = __ballot(g == 0), b1 = __ballot(g == 1), b2 = __ballot(g == 2);
        any |= __ballot(missing) != 0ull;                            // a call that is none of 0 / 1 / 2
        if (lane == 0) {
            out[0 * W + 2 * w2] = (uint32_t)b0; out[1 * W + 2 * w2] = (uint32_t)b1; out[2 * W + 2 * w2] = (uint32_t)b2;
            if (2 * w2 + 1 < W) {
                out[0 * W + 2 * w2 + 1] = (uint32_t)(b0 >> 32); out[1 * W + 2 * w2 + 1] = (uint32_t)(b1 >> 32);
                out[2 * W + 2 * w2 + 1] = (uint32_t)(b2 >> 32);
            }
        }
    }
    if (any && lane == 0) atomicOr(any_missing, 1u);
}

// per SNP and (fold, class) group: how many of the group's samples have genotype 0 and genotype 1 (low / high 16 bits).
// marg[snp * (2 * EPI_MAX_FOLDS) + g].  One workgroup per SNP row, wave w takes the groups w, w + 4, ...
static __global__ void __launch_bounds__(256) k_epi_marginals(const uint32_t *__restrict__ planes, int W, const uint32_t *__restrict__ group_w0,
                                                        int n_groups, uint32_t *__restrict__ marg) {
    const int snp = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *p0 = planes + (size_t)snp * 3 * W, *p1 = p0 + W;
    for (int g = wave; g < 2 * EPI_MAX_FOLDS; g += 4) {
        int c0 = 0, c1 = 0;
        if (g < n_groups)
            for (uint32_t w = group_w0[g] + lane; w < group_w0[g + 1]; w += 64) { c0 += __popc(p0[w]); c1 += __popc(p1[w]); }
        c0 = wave_sum(c0); c1 = wave_sum(c1);
        if (lane == 0) marg[(size_t)snp * (2 * EPI_MAX_FOLDS) + g] = (uint32_t)c0 | ((uint32_t)c1 << 16);
    }
}

// acc + popcount(x) in ONE instruction (v_bcnt_u32_b32 has an accumulator operand; written as a sum of four
// popcounts the compiler emits four v_bcnt + two v_add3 instead)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
    return acc;
}

// the MDR rule of the runner, mdr_high_risk_combinations2 (mdr.c:45-76): single precision, operation by
// operation (the library is built with -ffp-contract=off; float division is correctly rounded)
__device__ __forceinline__ bool mdr_high_risk_exact(float ca, float cu, float ratio) {
    const float total = ca + cu;
    const float prop = cu * ratio;
    const float red = total / (prop + ca);
    const float norm_unaff = prop * red;
    const float norm_aff = total - norm_unaff;
    return norm_aff >= norm_unaff;                                   // an empty cell is 0/0 = NaN: false
}

// The same decision without the division wherever it is safe.  In exact arithmetic the rule is
// count_aff * num_unaffected >= count_unaff * num_affected (norm_aff - norm_unaff = total * d / s with
// d = ca*nU - cu*nA, s = ca*nU + cu*nA); the six float operations above move norm_aff - norm_unaff by less than
// 2^-20 * total, so when |d| > 2^-16 * s the sign of d decides.  Cells closer to the boundary (ties included, and
// the empty cell) take the exact sequence.  With num_affected == num_unaffected the sequence is exact for counts
// below 2^23 (ratio = 1, red = 1) and reduces to count_aff >= count_unaff on a non-empty cell.
template <bool BALANCED>
__device__ __forceinline__ bool mdr_high_risk(int count_aff, int count_unaff, float ratio, float f_na, float f_nu) {
    if constexpr (BALANCED) {
        return (count_aff >= count_unaff) & ((count_aff | count_unaff) != 0);
    } else {
        const float ca = (float)count_aff, cu = (float)count_unaff;
        const float p1 = ca * f_nu, p2 = cu * f_na;
        const float d = p1 - p2, tol = (p1 + p2) * 0x1p-16f;
        if (__builtin_fabsf(d) > tol) return d > 0.0f;
        if ((count_aff | count_unaff) == 0) return false;            // the empty cell (0 / 0 = NaN in the sequence: false) -- common enough that
        return mdr_high_risk_exact(ca, cu, ratio);                   // without this line some lane of nearly every wave takes the division
    }
}

// ---------------------------------------------------------------------------
// listed combinations of 2 or 3 SNPs: in-fold cell counts, out[(comb * n_groups + g) * cells + c]
// (g = fold * 2 + class).  One wave per combination; utility / parity kernel, not the fast path.
// ---------------------------------------------------------------------------
template <int ORDER>
__global__ void __launch_bounds__(256) k_epi_counts(const uint32_t *__restrict__ planes, int W, const int32_t *__restrict__ combs,
                                                     int n_combs, const uint32_t *__restrict__ group_w0 /* n_groups + 1 */,
                                                     int n_groups, int32_t *__restrict__ out) {
    constexpr int CELLS = ORDER == 2 ? 9 : 27;
    const int lane = threadIdx.x & 63, comb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (comb >= n_combs) return;
    const uint32_t *r0 = planes + (size_t)combs[comb * ORDER] * 3 * W;
    const uint32_t *r1 = planes + (size_t)combs[comb * ORDER + 1] * 3 * W;
    const uint32_t *r2 = ORDER == 3 ? planes + (size_t)combs[comb * ORDER + 2] * 3 * W : r1;
    for (int g = 0; g < n_groups; g++) {
        int cnt[CELLS];
        #pragma unroll
        for (int c = 0; c < CELLS; c++) cnt[c] = 0;
        for (uint32_t w = group_w0[g] + lane; w < group_w0[g + 1]; w += 64) {
            #pragma unroll
            for (int a = 0; a < 3; a++) {
                const uint32_t x = r0[a * W + w];
                #pragma unroll
                for (int b = 0; b < 3; b++) {
                    const uint32_t xy = x & r1[b * W + w];
                    if constexpr (ORDER == 2) {
                        cnt[a * 3 + b] += __popc(xy);
                    } else {
                        #pragma unroll
                        for (int c = 0; c < 3; c++) cnt[(a * 3 + b) * 3 + c] += __popc(xy & r2[c * W + w]);
                    }
                }
            }
        }
        #pragma unroll
        for (int c = 0; c < CELLS; c++) {
            const int s = wave_sum(cnt[c]);
            if (lane == 0) out[((size_t)comb * n_groups + g) * CELLS + c] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// every pair (i, j), i < j, of a band of rows: cell counts per (fold, class) group, MDR high-risk cells
// per fold from the training counts, confusion matrix on the training or the testing part, balanced
// accuracy.  Workgroup = 4 waves = 4 rows x 64 columns; the planes of the 64 + 4 SNPs stream through LDS
// in chunks of 32 words (double buffered, one barrier per chunk); every lane keeps its pair's counts in
// registers: 9 running counts of the current group + K x 9 finished groups, two 16-bit counts per register
// (affected low, unaffected high: a (fold, class) group holds fewer than 65536 samples).
// ---------------------------------------------------------------------------
// COMPLETE: a dataset WITHOUT missing calls, where only the four cells of genotypes {0, 1} x {0, 1} are counted (16 instead
// of 36 AND + popcount pairs per step, two planes per SNP staged) and the other five follow at a group's end from the
// per-SNP, per-group genotype counts `marg` (k_epi_marginals): n(a, 2) = n_i(a) - n(a, 0) - n(a, 1),
// n(2, b) = n_j(b) - n(0, b) - n(1, b), n(2, 2) = the rest of the group.
// K <= 10: at most 168 registers, three waves per SIMD; K = 16 (more than 10 folds): two.
template <int K, bool TRAINING, bool BALANCED, bool COMPLETE>
__global__ void __launch_bounds__(256, K <= 10 ? 3 : 2) k_epi_pairs(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ marg, int W, int n_variants, int i_begin, int i_first, int i_end,
                                                    const unsigned *__restrict__ tile_base, int n_cols, unsigned n_tiles,
                                                    const EpiChunk *__restrict__ chunks, int n_chunks,
                                                    const EpiFold *__restrict__ folds /* K */, int n_affected, int n_unaffected,
                                                    double *__restrict__ acc_out, uint16_t *__restrict__ mask_out, unsigned long long n_pairs_out,
                                                    unsigned long long rank_base,
                                                    const double *__restrict__ thr, EpiCand *__restrict__ cand,
                                                    unsigned *__restrict__ cand_count, unsigned cand_cap) {
    // staging by LDS-DMA: 26 (17) x 8 linear rows of 32 words, the 16-byte pieces of a row swizzled by the column index so
    // that the per-lane reads of 16 lanes fall on 16 different slots
    constexpr int NP = COMPLETE ? 2 : 3, NC = NP * NP;               // planes staged per SNP, cells counted
    constexpr int RP = EPI_CH;
    constexpr int LROWS = COMPLETE ? 136 : 208;
    constexpr int NDMA = COMPLETE ? 17 : 26;                         // LDS-DMA instructions per chunk (8 rows each)
    // two separate arrays, not lds[2][...]: the compiler then sees that the LDS-DMA writes into one buffer cannot touch the
    // other and does not put an s_waitcnt vmcnt(0) in front of the counting loop's reads
    __shared__ __attribute__((aligned(16))) uint32_t lds_a[LROWS * RP];
    __shared__ __attribute__((aligned(16))) uint32_t lds_b[LROWS * RP];
    // Tiles that hold at least one pair are numbered COLUMN tile by column tile (tile_base[c] = tiles before column tile
    // tj0 + c; inside a column tile the row blocks 0 .. n - 1 from the band's first row down to the diagonal), and the
    // numbering is dealt to the XCDs in eight contiguous spans: workgroup b runs on XCD b % 8 (round-robin dispatch) and
    // takes tile (b % 8) * span + b / 8.  An XCD therefore works through whole column tiles: their 64 x 3 plane rows stay
    // in its L2 while only the four row-SNPs of each tile stream.
    const unsigned span = (n_tiles + 7u) / 8u;
    const unsigned tile = (blockIdx.x & 7u) * span + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    int c_lo = 0, c_hi = n_cols;
    while (c_hi - c_lo > 1) { const int mid = (c_lo + c_hi) >> 1; if (tile_base[mid] <= tile) c_lo = mid; else c_hi = mid; }
    const int j0 = ((i_begin >> 6) + c_lo) * EPI_TJ, i0 = i_begin + (int)(tile - tile_base[c_lo]) * EPI_TI;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = i0 + wave, j = j0 + lane;

    // ---- staging: piece q = one 16-byte piece of one (snp, plane) row of the chunk ----
    constexpr int ROWS = (EPI_TJ + EPI_TI) * NP;
    // LDS-DMA: one global_load_lds_dwordx4 = 64 lanes x 16 B = 8 rows of the image; lane l fetches row 8k + l / 8, physical
    // piece l % 8, i.e. the logical piece (l % 8) ^ swizzle(row's SNP): the swizzle sits on the source address.  Kept per
    // instruction: the word offset of the lane's piece inside the planes (below 2^32 words).  Always whole 32-word rows
    // (the words past a short last chunk are fetched and never read: the planes carry 32 words of slack); the four rows
    // of the image past the 204 used ones re-fetch row 0.  The complete-data image is plane-major (row = plane * 68 + SNP,
    // 136 rows = 17 instructions): with two planes per SNP an SNP-major image would put all 16 lanes of a read on 32 banks.
    uint32_t dma_off[7];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);         // the wave index as a scalar
    {
        #pragma unroll
        for (int r = 0; r < 7; r++) {
            const int k = wave + 4 * r, row8 = 8 * k + (lane >> 3), row = row8 < ROWS ? row8 : 0;
            const int snp_idx = COMPLETE ? row % (EPI_TJ + EPI_TI) : row / 3, plane = COMPLETE ? row / (EPI_TJ + EPI_TI) : row % 3;
            const int piece = (lane & 7) ^ ((snp_idx >> 1) & 7);
            const int snp = snp_idx < EPI_TJ ? j0 + snp_idx : i0 + (snp_idx - EPI_TJ);
            dma_off[r] = ((uint32_t)snp * 3u + (uint32_t)plane) * (uint32_t)W + (uint32_t)piece * 4u;
        }
    }
    auto load_chunk = [&](int c, uint32_t *dst) {
        const uint32_t w0 = chunks[c].w0;
        #pragma unroll
        for (int r = 0; r < 7; r++) {
            const int k = wave_u + 4 * r;
            if (k < NDMA)
                __builtin_amdgcn_global_load_lds(planes + (dma_off[r] + w0), (__attribute__((address_space(3))) uint32_t *)(dst + 8 * k * EPI_CH), 16, 0, 0);
        }
    };

    uint32_t packed[K][9];
    uint32_t run[NC];
    #pragma unroll
    for (int f = 0; f < K; f++)
        #pragma unroll
        for (int c = 0; c < 9; c++) packed[f][c] = 0;
    // complete data: genotype counts of the tile's 64 + 4 SNPs per group, [g][0..63 columns | 64..67 rows | 68 group size]
    __shared__ uint32_t lds_marg[COMPLETE ? 2 * K * 69 : 1];
    if constexpr (COMPLETE) {
        for (int q = t; q < 2 * K * 69; q += 256) {
            const int g = q / 69, e = q % 69;
            uint32_t v;
            if (e < EPI_TJ + EPI_TI) v = marg[(size_t)(e < EPI_TJ ? j0 + e : i0 + (e - EPI_TJ)) * (2 * EPI_MAX_FOLDS) + g];
            else { const int n = (g & 1) ? folds[g >> 1].test_u : folds[g >> 1].test_a; v = n > 0 ? (uint32_t)n : 0u; }
            lds_marg[q] = v;
        }
    }

    load_chunk(0, lds_a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int swz_j = (lane >> 1) & 7, swz_i = ((EPI_TJ + wave_u) >> 1) & 7;
    constexpr int PS = COMPLETE ? (EPI_TJ + EPI_TI) * RP * 4 : RP * 4;      // bytes from one plane of an SNP to the next
    uint4 xa[NP], ya[NP], xb[NP], yb[NP];
    bool fresh = true;
#define HPGV_EPI_FETCH(X, Y, S)                                                                          \
        _Pragma("unroll") for (int a = 0; a < NP; a++) {                                                 \
            X[a] = *reinterpret_cast<const uint4 *>(cur_bytes + ((ioff ^ ((S) << 2)) + a * PS));         \
            Y[a] = *reinterpret_cast<const uint4 *>(cur_bytes + ((joff ^ ((S) << 2)) + a * PS));         \
        }
#define HPGV_EPI_COUNT1(X, Y, FIRST)                                                                     \
        _Pragma("unroll") for (int a = 0; a < NP; a++)                                                   \
            _Pragma("unroll") for (int b = 0; b < NP; b++) {                                             \
                uint32_t r = (FIRST) ? (uint32_t)__popc(X[a].x & Y[b].x) : bcnt_acc(X[a].x & Y[b].x, run[a * NP + b]); \
                r = bcnt_acc(X[a].y & Y[b].y, r);                                                        \
                r = bcnt_acc(X[a].z & Y[b].z, r); r = bcnt_acc(X[a].w & Y[b].w, r);                      \
                run[a * NP + b] = r;                                                                     \
            }
    // `fresh` (wave-uniform): the step starts a group, its counts start from zero -- the running counts are never cleared
#define HPGV_EPI_COUNT(X, Y, S)                                                                          \
        if (fresh) { HPGV_EPI_COUNT1(X, Y, true) } else { HPGV_EPI_COUNT1(X, Y, false) }                 \
        {                                                                                                \
            const int g = (int)((flush >> (2 * (S))) & 0xFFu);       /* byte S / 4 */                    \
            fresh = g != 0xFF;                                                                           \
            if (fresh) {                     /* a (fold, class) group ends here: bank its nine counts */ \
                const int f = g >> 1, sh = (g & 1) * 16;                                                 \
                uint32_t cell[9];                                                                        \
                if constexpr (COMPLETE) {    /* the five cells with a genotype 2 from the genotype counts */ \
                    const uint32_t mj = lds_marg[g * 69 + lane], mi = lds_marg[g * 69 + EPI_TJ + wave_u], ng = lds_marg[g * 69 + 68]; \
                    const uint32_t mj0 = mj & 0xFFFFu, mj1 = mj >> 16, mi0 = mi & 0xFFFFu, mi1 = mi >> 16;  \
                    cell[0] = run[0]; cell[1] = run[1]; cell[2] = mi0 - run[0] - run[1];                 \
                    cell[3] = run[2]; cell[4] = run[3]; cell[5] = mi1 - run[2] - run[3];                 \
                    cell[6] = mj0 - run[0] - run[2]; cell[7] = mj1 - run[1] - run[3];                    \
                    cell[8] = ng - mi0 - mi1 - cell[6] - cell[7];                                        \
                } else {                                                                                 \
                    _Pragma("unroll") for (int cc = 0; cc < 9; cc++) cell[cc] = run[cc < NC ? cc : 0];   \
                }                                                                                        \
                _Pragma("unroll") for (int ff = 0; ff < K; ff++)                                         \
                    if (ff == f) {                                                                       \
                        _Pragma("unroll") for (int cc = 0; cc < 9; cc++) packed[ff][cc] += cell[cc] << sh; \
                    }                                                                                    \
            }                                                                                            \
        }
    // one chunk: the next chunk's loads fly into NXT during the counting over CUR
#define HPGV_EPI_CHUNK(CUR, NXT)                                                                         \
    {                                                                                                    \
        if (c + 1 < n_chunks) load_chunk(c + 1, NXT);                                                    \
        const int nw = (int)chunks[c].nw;                                                                \
        const uint64_t flush = chunks[c].flush;                      /* wave-uniform */                  \
        /* byte offsets of the lane's column and of the wave's own row (the same in every lane: broadcast) in the buffer;     \
           with the swizzle folded in, the 16-byte piece of step S sits at offset ^ (S * 4): rows are 128-byte aligned */    \
        const char *cur_bytes = reinterpret_cast<const char *>(CUR);                                                       \
        int joff = (lane * (COMPLETE ? 1 : 3) * RP + (swz_j << 2)) * 4,                                                    \
            ioff = ((EPI_TJ + wave_u) * (COMPLETE ? 1 : 3) * RP + (swz_i << 2)) * 4;                                        \
        asm("" : "+v"(joff)); asm("" : "+s"(ioff));   /* opaque: keeps the compiler from pulling the * 4 out of the ^ */     \
        if constexpr (K > 5) {                                                                           \
            /* three waves per SIMD hide the LDS latency; one register set keeps the kernel within 168 VGPRs */ \
            for (int s = 0; s < nw; s += 4) {                                                            \
                HPGV_EPI_FETCH(xa, ya, s)                                                                \
                HPGV_EPI_COUNT(xa, ya, s)                                                                \
            }                                                                                            \
            (void)xb; (void)yb;                                                                          \
        } else {                                                                                         \
            HPGV_EPI_FETCH(xa, ya, 0)                                                                    \
            for (int s = 0; s < nw; s += 8) {                                                            \
                HPGV_EPI_FETCH(xb, yb, s + 4)                                                            \
                HPGV_EPI_COUNT(xa, ya, s)                                                                \
                if (s + 8 < nw) { HPGV_EPI_FETCH(xa, ya, s + 8) }                                        \
                HPGV_EPI_COUNT(xb, yb, s + 4)                                                            \
            }                                                                                            \
        }                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             /* this wave's part of NXT has landed */ \
        __syncthreads();                                                                                 \
    }
    for (int c = 0; c < n_chunks; c++) {
        HPGV_EPI_CHUNK(lds_a, lds_b)
        if (++c >= n_chunks) break;
        HPGV_EPI_CHUNK(lds_b, lds_a)
    }
#undef HPGV_EPI_CHUNK
#undef HPGV_EPI_FETCH
#undef HPGV_EPI_COUNT
#undef HPGV_EPI_COUNT1

    if (i < i_first || i >= i_end || i >= n_variants || j >= n_variants || j <= i) return;

    // ---- per fold: training counts, high-risk cells, confusion matrix, balanced accuracy ----
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    if (!acc_out && cand && n_affected < 65536 && n_unaffected < 65536) {
        // Ranking, classes below 65 536 samples: the whole evaluation stays on the packed pairs (cases low, controls high
        // half: neither half of a difference or of a sum of selected cells can borrow or carry), and the high-risk mask is
        // formed only for the rare model that reaches its fold's threshold.
        uint32_t totp[9];
        #pragma unroll
        for (int c = 0; c < 9; c++) {
            totp[c] = 0;
            #pragma unroll
            for (int f = 0; f < K; f++) totp[c] += packed[f][c];
        }
        #pragma unroll
        for (int f = 0; f < K; f++) {
            const EpiFold fo = folds[f];
            if (fo.test_a < 0) continue;
            uint32_t sel = 0;                                            // TP (low half), FP (high half)
            #pragma unroll
            for (int c = 0; c < 9; c++) {
                const uint32_t in = packed[f][c], tr = totp[c] - in;
                bool high;
                // balanced: cases >= controls, compared as (cases:controls) >= (controls:cases).  An empty training cell is not
                // high risk: on the training part it adds nothing either way; on the testing part its samples must stay out,
                // so the comparison is made against max(tr, 1), which the swapped halves of tr = 0 cannot reach.
                if constexpr (BALANCED) high = __builtin_amdgcn_alignbit(tr, tr, 16) >= (TRAINING ? tr : (tr > 1u ? tr : 1u));
                else high = mdr_high_risk<false>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu);
                sel += high ? (TRAINING ? tr : in) : 0u;
            }
            const int tp = (int)(sel & 0xFFFFu), fp = (int)(sel >> 16);
            const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
            const double TP = (double)tp, TN = (double)(size_u - fp), ya = (double)size_a, yu = (double)size_u;
            double qa = TP * fo.inv_a, qu = TN * fo.inv_u;               // the two quotients as below (Markstein)
            qa = __builtin_fma(__builtin_fma(-qa, ya, TP), fo.inv_a, qa);
            qu = __builtin_fma(__builtin_fma(-qu, yu, TN), fo.inv_u, qu);
            const double acc = (qa + qu) / 2;
            if (acc >= thr[f]) {
                uint32_t mask = 0;
                #pragma unroll
                for (int c = 0; c < 9; c++) {
                    const uint32_t tr = totp[c] - packed[f][c];
                    if (mdr_high_risk<BALANCED>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu)) mask |= 1u << c;
                }
                const unsigned slot = atomicAdd(&cand_count[f], 1u);
                if (slot < cand_cap) {
                    EpiCand e;
                    e.accuracy = acc; e.i = i; e.j = j; e.risky = mask; e.pad = 0;
                    cand[(size_t)f * cand_cap + slot] = e;
                }
            }
        }
        return;
    }
    int tot_a[9], tot_u[9];
    #pragma unroll
    for (int c = 0; c < 9; c++) { tot_a[c] = 0; tot_u[c] = 0; }
    #pragma unroll
    for (int f = 0; f < K; f++)
        #pragma unroll
        for (int c = 0; c < 9; c++) { tot_a[c] += (int)(packed[f][c] & 0xFFFFu); tot_u[c] += (int)(packed[f][c] >> 16); }
    const unsigned long long vi = (unsigned long long)i;
    const unsigned long long p = vi * (2ull * (unsigned long long)n_variants - vi - 1ull) / 2ull + (unsigned long long)(j - i - 1) - rank_base;
    #pragma unroll
    for (int f = 0; f < K; f++) {
        const EpiFold fo = folds[f];
        if (fo.test_a < 0) continue;                                 // fold beyond the run's num_folds
        int tp = 0, fp = 0;
        uint32_t mask = 0;
        #pragma unroll
        for (int c = 0; c < 9; c++) {
            const int in_a = (int)(packed[f][c] & 0xFFFFu), in_u = (int)(packed[f][c] >> 16);
            const int tr_a = tot_a[c] - in_a, tr_u = tot_u[c] - in_u;
            if (mdr_high_risk<BALANCED>(tr_a, tr_u, ratio, f_na, f_nu)) {
                mask |= 1u << c;
                tp += TRAINING ? tr_a : in_a;
                fp += TRAINING ? tr_u : in_u;
            }
        }
        const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
        // evaluate_model BA (model.c:466-467) on {TP, FN, FP, TN} = {tp, size_a - tp, fp, size_u - fp}:
        // ((TP / (TP + FN)) + (TN / (TN + FP))) / 2 with TP + FN = size_a, TN + FP = size_u.  Each quotient x / y is
        // formed as q = x * r, q' = fma(fma(-q, y, x), r, q) with r = RN(1 / y) from the host: the correctly rounded
        // quotient (Markstein), i.e. the very double the division gives; y = 0 gives 0 * inf = NaN like 0 / 0.
        const double TP = (double)tp, TN = (double)(size_u - fp), ya = (double)size_a, yu = (double)size_u;
        double qa = TP * fo.inv_a, qu = TN * fo.inv_u;
        qa = __builtin_fma(__builtin_fma(-qa, ya, TP), fo.inv_a, qa);
        qu = __builtin_fma(__builtin_fma(-qu, yu, TN), fo.inv_u, qu);
        const double acc = (qa + qu) / 2;
        if (acc_out) {
            acc_out[(unsigned long long)f * n_pairs_out + p] = acc;
            mask_out[(unsigned long long)f * n_pairs_out + p] = (uint16_t)mask;
        }
        if (cand && acc >= thr[f]) {
            const unsigned slot = atomicAdd(&cand_count[f], 1u);
            if (slot < cand_cap) {
                EpiCand e;
                e.accuracy = acc; e.i = i; e.j = j; e.risky = mask; e.pad = 0;
                cand[(size_t)f * cand_cap + slot] = e;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// every triple (i, j, k), i < j < k, of a band of first SNPs i: the order-3 model of process_set_of_combinations
// (27 cells, cell = (g_i * 3 + g_j) * 3 + g_k).  Workgroup = 4 waves; a wave owns one (i, j) pair, its lanes 64
// third SNPs k.  Per 32-bit word a lane forms the nine x_a & y_b words of its pair and 27 x (v_and + v_bcnt) =
// 63 operations.  With 27 cells the per-fold counts of all K folds do not fit in registers, so the scan makes TWO
// passes over the samples: the first leaves the 27 totals (cases low, controls high 16 bits), the second visits the
// groups in (fold, class) order and, each time a fold's two groups are complete, evaluates that fold at once from
// totals - fold counts.  Classes are limited to 65 535 samples here.
// ---------------------------------------------------------------------------
struct EpiCand3 {
    double accuracy;
    int32_t i, j, k;
    uint32_t risky;                   // bit c = cell c of the 27 is high risk
};

template <bool TRAINING, bool BALANCED>
__global__ void __launch_bounds__(256, 3) k_epi_triples(const uint32_t *__restrict__ planes, int W, int n_variants, int i_first,
                                                         const unsigned *__restrict__ row_base /* n_i + 1 */, int n_i,
                                                         const unsigned *__restrict__ jb_prefix /* n_jb + 1 */, int n_jb,
                                                         const EpiChunk *__restrict__ chunks, int n_chunks, int num_folds,
                                                         const EpiFold *__restrict__ folds, int n_affected, int n_unaffected,
                                                         double *__restrict__ acc_out, uint32_t *__restrict__ mask_out,
                                                         const double *__restrict__ thr, EpiCand3 *__restrict__ cand,
                                                         unsigned *__restrict__ cand_count, unsigned cand_cap) {
    // LDS-DMA staging: 26 x 8 linear rows of 32 words, pieces swizzled by the SNP index on the source address (as k_epi_pairs)
    constexpr int RP = EPI_CH;
    constexpr int LROWS = 208;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][LROWS * RP];
    // only tiles that hold a triple are launched: blockIdx.x -> first SNP (bisection on row_base), then its j block
    // (bisection on jb_prefix, the count of k tiles per j block, which does not depend on i), then the k tile
    int r_lo = 0, r_hi = n_i;
    while (r_hi - r_lo > 1) { const int mid = (r_lo + r_hi) >> 1; if (row_base[mid] <= blockIdx.x) r_lo = mid; else r_hi = mid; }
    const int i = i_first + r_lo;
    const int jb_min = (i + 1) >> 2;                                 // first j block with a row j > i
    const unsigned want = (blockIdx.x - row_base[r_lo]) + jb_prefix[jb_min];
    int b_lo = jb_min, b_hi = n_jb;
    while (b_hi - b_lo > 1) { const int mid = (b_lo + b_hi) >> 1; if (jb_prefix[mid] <= want) b_lo = mid; else b_hi = mid; }
    const int j0 = b_lo * EPI_TI;
    const int k0 = (((j0 + 1) >> 6) + (int)(want - jb_prefix[b_lo])) * EPI_TJ;     // k tiles from the one that holds j0 + 1 on
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = j0 + wave, k = k0 + lane;
    // LDS rows: 0..63 the k columns, 64..67 the four j rows, 68 the i row
    constexpr int SNPS = EPI_TJ + EPI_TI + 1, ROWS = SNPS * 3;
    auto load_chunk = [&](int c, int buf) {
        const uint32_t w0 = chunks[c].w0;
        const int nw = (int)chunks[c].nw;
        #pragma unroll
        for (int r = 0; r < 7; r++) {
            const int kk = wave + 4 * r;
            if (kk < 26) {
                const int row = 8 * kk + (lane >> 3), sidx = row / 3;
                const int piece = (lane & 7) ^ ((sidx >> 1) & 7);
                if (row < ROWS && piece * 4 < nw) {
                    const int snp = sidx < EPI_TJ ? k0 + sidx : sidx < EPI_TJ + EPI_TI ? j0 + (sidx - EPI_TJ) : i;
                    const uint32_t *src = planes + ((size_t)snp * 3 + row % 3) * W + w0 + piece * 4;
                    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) uint32_t *)&lds[buf][8 * kk * EPI_CH], 16, 0, 0);
                }
            }
        }
    };
    const bool live = j > i && k > j && j < n_variants && k < n_variants;
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    uint32_t tot[27], cur[27], run[27];
    #pragma unroll
    for (int c = 0; c < 27; c++) { tot[c] = 0; cur[c] = 0; run[c] = 0; }

    for (int pass = 0; pass < 2; pass++) {
        __syncthreads();
        load_chunk(0, 0);
        __syncthreads();
        for (int c = 0; c < n_chunks; c++) {
            const int buf = c & 1;
            if (c + 1 < n_chunks) load_chunk(c + 1, buf ^ 1);
            const int nw = (int)chunks[c].nw;
            const uint64_t flush = chunks[c].flush;
            const uint32_t *zrow = &lds[buf][lane * 3 * RP];
            const uint32_t *yrow = &lds[buf][(EPI_TJ + wave) * 3 * RP];
            const uint32_t *xrow = &lds[buf][(EPI_TJ + EPI_TI) * 3 * RP];
            const int swz_z = (lane >> 1) & 7, swz_y = ((EPI_TJ + wave) >> 1) & 7, swz_x = ((EPI_TJ + EPI_TI) >> 1) & 7;
            for (int s = 0; s < nw; s += 4) {
                uint4 x[3], y[3], z[3];
                #pragma unroll
                for (int a = 0; a < 3; a++) {
                    x[a] = *reinterpret_cast<const uint4 *>(xrow + a * RP + (((s >> 2) ^ swz_x) << 2));
                    y[a] = *reinterpret_cast<const uint4 *>(yrow + a * RP + (((s >> 2) ^ swz_y) << 2));
                    z[a] = *reinterpret_cast<const uint4 *>(zrow + a * RP + (((s >> 2) ^ swz_z) << 2));
                }
                #pragma unroll
                for (int a = 0; a < 3; a++)
                    #pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const uint4 xy = make_uint4(x[a].x & y[b].x, x[a].y & y[b].y, x[a].z & y[b].z, x[a].w & y[b].w);
                        #pragma unroll
                        for (int d = 0; d < 3; d++) {
                            uint32_t r = run[(a * 3 + b) * 3 + d];
                            r = bcnt_acc(xy.x & z[d].x, r); r = bcnt_acc(xy.y & z[d].y, r);
                            r = bcnt_acc(xy.z & z[d].z, r); r = bcnt_acc(xy.w & z[d].w, r);
                            run[(a * 3 + b) * 3 + d] = r;
                        }
                    }
                const int g = (int)((flush >> (2 * s)) & 0xFFu);
                if (g != 0xFF) {                                     // a (fold, class) group ends with this step
                    const int sh = (g & 1) * 16;
                    if (pass == 0) {
                        #pragma unroll
                        for (int cc = 0; cc < 27; cc++) { tot[cc] += run[cc] << sh; run[cc] = 0; }
                    } else {
                        #pragma unroll
                        for (int cc = 0; cc < 27; cc++) { cur[cc] += run[cc] << sh; run[cc] = 0; }
                        // groups come as (fold 0 cases, fold 0 controls, fold 1 cases, ...); a fold is complete with its
                        // controls, or with its cases when it has no controls (then no group g | 1 ever ends)
                        const int f = g >> 1;
                        const EpiFold fo = folds[f];
                        const bool fold_done = (g & 1) || fo.test_u == 0;
                        if (fold_done) {
                            if (live) {
                                int tp = 0, fp = 0;
                                uint32_t mask = 0;
                                #pragma unroll
                                for (int cc = 0; cc < 27; cc++) {
                                    const int in_a = (int)(cur[cc] & 0xFFFFu), in_u = (int)(cur[cc] >> 16);
                                    const int tr_a = (int)(tot[cc] & 0xFFFFu) - in_a, tr_u = (int)(tot[cc] >> 16) - in_u;
                                    if (mdr_high_risk<BALANCED>(tr_a, tr_u, ratio, f_na, f_nu)) {
                                        mask |= 1u << cc;
                                        tp += TRAINING ? tr_a : in_a;
                                        fp += TRAINING ? tr_u : in_u;
                                    }
                                }
                                const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
                                const double TP = (double)tp, TN = (double)(size_u - fp), ya = (double)size_a, yu = (double)size_u;
                                double qa = TP * fo.inv_a, qu = TN * fo.inv_u;
                                qa = __builtin_fma(__builtin_fma(-qa, ya, TP), fo.inv_a, qa);
                                qu = __builtin_fma(__builtin_fma(-qu, yu, TN), fo.inv_u, qu);
                                const double acc = (qa + qu) / 2;
                                if (acc_out) {
                                    const size_t V = (size_t)n_variants;
                                    const size_t o = (((size_t)f * V + (size_t)i) * V + (size_t)j) * V + (size_t)k;
                                    acc_out[o] = acc; mask_out[o] = mask;
                                }
                                if (cand && acc >= thr[f]) {
                                    const unsigned slot = atomicAdd(&cand_count[f], 1u);
                                    if (slot < cand_cap) {
                                        EpiCand3 e;
                                        e.accuracy = acc; e.i = i; e.j = j; e.k = k; e.risky = mask;
                                        cand[(size_t)f * cand_cap + slot] = e;
                                    }
                                }
                            }
                            #pragma unroll
                            for (int cc = 0; cc < 27; cc++) cur[cc] = 0;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    (void)num_folds;
}

// ---------------------------------------------------------------------------
// The triple scan of the ranking in ONE pass over the samples (at most K folds, classes below 65 536 samples): every
// lane keeps the 27 counts of all K folds, two 16-bit counts per register (cases low, controls high) = 27 K registers
// besides the 27 running counts and two sets of operands.  That takes the whole register file of a SIMD (one wave per
// SIMD, up to 512 registers: what does not fit in the 256 architectural VGPRs the compiler parks in the accumulation
// registers), so latency is hidden inside the wave: the operands of step s + 1 are read while step s counts, the next
// chunk arrives by LDS-DMA.  Half the counting work of the two-pass form above.  Evaluation as in k_epi_pairs'
// ranking mode: packed differences and sums, the 27-bit mask only for a model that reaches its fold's threshold.
// ---------------------------------------------------------------------------
template <int K, bool TRAINING, bool BALANCED>
__global__ void __launch_bounds__(256, 1) k_epi_triples1(const uint32_t *__restrict__ planes, int W, int n_variants, int i_first,
                                                          const unsigned *__restrict__ row_base /* n_i + 1 */, int n_i,
                                                          const unsigned *__restrict__ jb_prefix /* n_jb + 1 */, int n_jb,
                                                          const EpiChunk *__restrict__ chunks, int n_chunks,
                                                          const EpiFold *__restrict__ folds, int n_affected, int n_unaffected,
                                                          const double *__restrict__ thr, EpiCand3 *__restrict__ cand,
                                                          unsigned *__restrict__ cand_count, unsigned cand_cap) {
    constexpr int RP = EPI_CH;
    __shared__ __attribute__((aligned(16))) uint32_t lds_a[208 * RP];
    __shared__ __attribute__((aligned(16))) uint32_t lds_b[208 * RP];
    // tile -> (i, j block, k tile) as in k_epi_triples
    int r_lo = 0, r_hi = n_i;
    while (r_hi - r_lo > 1) { const int mid = (r_lo + r_hi) >> 1; if (row_base[mid] <= blockIdx.x) r_lo = mid; else r_hi = mid; }
    const int i = i_first + r_lo;
    const int jb_min = (i + 1) >> 2;
    const unsigned want = (blockIdx.x - row_base[r_lo]) + jb_prefix[jb_min];
    int b_lo = jb_min, b_hi = n_jb;
    while (b_hi - b_lo > 1) { const int mid = (b_lo + b_hi) >> 1; if (jb_prefix[mid] <= want) b_lo = mid; else b_hi = mid; }
    const int j0 = b_lo * EPI_TI;
    const int k0 = (((j0 + 1) >> 6) + (int)(want - jb_prefix[b_lo])) * EPI_TJ;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = j0 + wave, k = k0 + lane;
    // LDS rows: 0..63 the k columns, 64..67 the four j rows, 68 the i row; staging as in k_epi_pairs
    constexpr int ROWS = (EPI_TJ + EPI_TI + 1) * 3;
    uint32_t dma_off[7];
    #pragma unroll
    for (int r = 0; r < 7; r++) {
        const int kk = wave + 4 * r, row8 = 8 * kk + (lane >> 3), row = row8 < ROWS ? row8 : 0, sidx = row / 3;
        const int piece = (lane & 7) ^ ((sidx >> 1) & 7);
        const int snp = sidx < EPI_TJ ? k0 + sidx : sidx < EPI_TJ + EPI_TI ? j0 + (sidx - EPI_TJ) : i;
        dma_off[r] = ((uint32_t)snp * 3u + (uint32_t)(row % 3)) * (uint32_t)W + (uint32_t)piece * 4u;
    }
    auto load_chunk = [&](int c, uint32_t *dst) {
        const uint32_t w0 = chunks[c].w0;
        #pragma unroll
        for (int r = 0; r < 7; r++) {
            const int kk = wave_u + 4 * r;
            if (kk < 26)
                __builtin_amdgcn_global_load_lds(planes + (dma_off[r] + w0), (__attribute__((address_space(3))) uint32_t *)(dst + 8 * kk * EPI_CH), 16, 0, 0);
        }
    };

    uint32_t packed[K][27];
    uint32_t run[27];
    #pragma unroll
    for (int f = 0; f < K; f++)
        #pragma unroll
        for (int c = 0; c < 27; c++) packed[f][c] = 0;
    bool fresh = true;
    const int swz_z = (lane >> 1) & 7, swz_y = ((EPI_TJ + wave_u) >> 1) & 7, swz_x = ((EPI_TJ + EPI_TI) >> 1) & 7;

    load_chunk(0, lds_a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint4 xa[3], ya[3], za[3], xb[3], yb[3], zb[3];
#define HPGV_EPI3_FETCH(X, Y, Z, S)                                                                      \
        _Pragma("unroll") for (int a = 0; a < 3; a++) {                                                  \
            X[a] = *reinterpret_cast<const uint4 *>(cur_bytes + ((xoff ^ ((S) << 2)) + a * RP * 4));     \
            Y[a] = *reinterpret_cast<const uint4 *>(cur_bytes + ((yoff ^ ((S) << 2)) + a * RP * 4));     \
            Z[a] = *reinterpret_cast<const uint4 *>(cur_bytes + ((zoff ^ ((S) << 2)) + a * RP * 4));     \
        }
    // nine cells at a time: first the nine three-way ANDs, then the nine popcount-accumulates, so that no instruction
    // waits on the one just before it (with one wave per SIMD a dependent pair issues ~1.7x slower than an independent one).
#define HPGV_EPI3_W(V, C) ((C) == 0 ? (V).x : (C) == 1 ? (V).y : (C) == 2 ? (V).z : (V).w)
#define HPGV_EPI3_COUNT1(X, Y, Z, FIRST)                                                 \
        _Pragma("unroll") for (int a = 0; a < 3; a++) {                                                  \
            _Pragma("unroll") for (int w = 0; w < 4; w++) {                                              \
                uint32_t t9[9];                                                                          \
                _Pragma("unroll") for (int b = 0; b < 3; b++)                                            \
                    _Pragma("unroll") for (int d = 0; d < 3; d++)                                        \
                        t9[b * 3 + d] = HPGV_EPI3_W(X[a], w) & HPGV_EPI3_W(Y[b], w) & HPGV_EPI3_W(Z[d], w); \
                _Pragma("unroll") for (int q = 0; q < 9; q++)                                            \
                    run[a * 9 + q] = ((FIRST) && w == 0) ? (uint32_t)__popc(t9[q]) : bcnt_acc(t9[q], run[a * 9 + q]); \
            }                                                                                            \
        }
#define HPGV_EPI3_COUNT(X, Y, Z, S)                                                      \
        if (fresh) { HPGV_EPI3_COUNT1(X, Y, Z, true) } else { HPGV_EPI3_COUNT1(X, Y, Z, false) }         \
        {                                                                                                \
            const int g = (int)((flush >> (2 * (S))) & 0xFFu);                                           \
            fresh = g != 0xFF;                                                                           \
            if (fresh) {                     /* a (fold, class) group ends here: bank its 27 counts */   \
                const int f = g >> 1, sh = (g & 1) * 16;                                                 \
                _Pragma("unroll") for (int ff = 0; ff < K; ff++)                                         \
                    if (ff == f) {                                                                       \
                        _Pragma("unroll") for (int cc = 0; cc < 27; cc++) packed[ff][cc] += run[cc] << sh; \
                    }                                                                                    \
            }                                                                                            \
        }
#define HPGV_EPI3_CHUNK(CUR, NXT)                                                                        \
    {                                                                                                    \
        if (c + 1 < n_chunks) load_chunk(c + 1, NXT);                                                    \
        const int nw = (int)chunks[c].nw;                                                                \
        const uint64_t flush = chunks[c].flush;                                                          \
        const char *cur_bytes = reinterpret_cast<const char *>(CUR);                                     \
        int zoff = (lane * 3 * RP + (swz_z << 2)) * 4, yoff = ((EPI_TJ + wave_u) * 3 * RP + (swz_y << 2)) * 4,  \
            xoff = ((EPI_TJ + EPI_TI) * 3 * RP + (swz_x << 2)) * 4;                                      \
        asm("" : "+v"(zoff)); asm("" : "+s"(yoff)); asm("" : "+s"(xoff));                                \
        HPGV_EPI3_FETCH(xa, ya, za, 0)                                                                   \
        for (int s = 0; s < nw; s += 8) {        /* reads past the chunk's end: a neighbouring row, never used */ \
            HPGV_EPI3_FETCH(xb, yb, zb, s + 4)                                                           \
            HPGV_EPI3_COUNT(xa, ya, za, s)                                                               \
            HPGV_EPI3_FETCH(xa, ya, za, s + 8)                                                           \
            HPGV_EPI3_COUNT(xb, yb, zb, s + 4)                                                           \
        }                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                                 \
    }
    for (int c = 0; c < n_chunks; c++) {
        HPGV_EPI3_CHUNK(lds_a, lds_b)
        if (++c >= n_chunks) break;
        HPGV_EPI3_CHUNK(lds_b, lds_a)
    }
#undef HPGV_EPI3_CHUNK
#undef HPGV_EPI3_COUNT
#undef HPGV_EPI3_COUNT1
#undef HPGV_EPI3_W
#undef HPGV_EPI3_FETCH

    if (!(j > i && k > j && j < n_variants && k < n_variants)) return;
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    uint32_t totp[27];
    #pragma unroll
    for (int c = 0; c < 27; c++) {
        totp[c] = 0;
        #pragma unroll
        for (int f = 0; f < K; f++) totp[c] += packed[f][c];
    }
    #pragma unroll
    for (int f = 0; f < K; f++) {
        const EpiFold fo = folds[f];
        if (fo.test_a < 0) continue;                                 // fold beyond the run's num_folds
        uint32_t sel = 0;                                            // TP (low half), FP (high half)
        #pragma unroll
        for (int c = 0; c < 27; c++) {
            const uint32_t in = packed[f][c], tr = totp[c] - in;
            bool high;
            if constexpr (BALANCED) high = __builtin_amdgcn_alignbit(tr, tr, 16) >= (TRAINING ? tr : (tr > 1u ? tr : 1u));   // as k_epi_pairs
            else high = mdr_high_risk<false>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu);
            sel += high ? (TRAINING ? tr : in) : 0u;
        }
        const int tp = (int)(sel & 0xFFFFu), fp = (int)(sel >> 16);
        const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
        const double TP = (double)tp, TN = (double)(size_u - fp), ya_ = (double)size_a, yu_ = (double)size_u;
        double qa = TP * fo.inv_a, qu = TN * fo.inv_u;               // the two quotients as in k_epi_pairs (Markstein)
        qa = __builtin_fma(__builtin_fma(-qa, ya_, TP), fo.inv_a, qa);
        qu = __builtin_fma(__builtin_fma(-qu, yu_, TN), fo.inv_u, qu);
        const double acc = (qa + qu) / 2;
        if (acc >= thr[f]) {
            uint32_t mask = 0;
            #pragma unroll
            for (int c = 0; c < 27; c++) {
                const uint32_t tr = totp[c] - packed[f][c];
                if (mdr_high_risk<BALANCED>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu)) mask |= 1u << c;
            }
            const unsigned slot = atomicAdd(&cand_count[f], 1u);
            if (slot < cand_cap) {
                EpiCand3 e;
                e.accuracy = acc; e.i = i; e.j = j; e.k = k; e.risky = mask;
                cand[(size_t)f * cand_cap + slot] = e;
            }
        }
    }
}

}  // namespace hpgv
