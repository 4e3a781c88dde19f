// hpgv_epi_mfma_kernels.h -- the pair and triple rankings of the epistasis / MDR path with the cell counts on the MATRIX
// cores (model.c:76-206 combination_counts_all_folds, mdr.c:45-76, model.c:320-476; what k_epi_pairs / k_epi_triples3 of
// hpgv_epi_kernels.h / hpgv_epi_triples3_kernels.h do on the vector ALU).
//
// The nine cell counts of the pairs of 16 row SNPs x 16 column SNPs over 128 samples are nine products of 0/1 matrices:
// count(a, b)[i][j] = sum over samples of plane_i[a] * plane_j[b], i.e. v_mfma_scale_f32_16x16x128_f8f6f4 with the genotype
// planes as FP4 operands (scales 1) -- A holds plane a of the 16 row SNPs, B plane b of the 16 column SNPs; sums of 0 / 1 are
// exact in the f32 accumulators far beyond a group's 65 535 samples.  The result tile of that instruction has its column on the
// lane (lane & 15) and rows 4 (lane >> 4) + 0 .. 3 in the lane's four registers, so with the nine (a, b) tiles side by side every
// lane ends up with the WHOLE 3 x 3 table of four pairs: (i0 + 4 (lane >> 4) + q, j0 + (lane & 15)), q = 0 .. 3, and the
// evaluation needs no exchange between lanes.
//
// Operands: the planes are bits.  Lane (r, h) = (lane & 15, lane >> 4) supplies 32 samples of SNP r to each MFMA -- the
// instruction sums over the four h groups and the lane's 32 four-bit values, in an order that does not matter to a count as
// long as A and B agree.  Per 128-sample step (the unit the (fold, class) groups are padded to) a lane reads ONE 32-bit word
// per plane (word h of its SNP's four, out of the LDS image that k_epi_pairs' staging scheme keeps ahead): exactly one MFMA's
// worth.  The four-bit values are NOT 0 / 1: an operand register is the word ANDed with one bit of every nibble,
// x & 0x22222222 = eight samples as E2M1 1.0 where the bit is set, x & 0x11111111 = 0.5, x & 0x44444444 = 2.0 -- ONE
// instruction per eight samples (the nibble's top bit is the sign: those samples are shifted down to 0.5).  The other side
// must weigh the same samples 2.0, 1.0, 0.5, 2.0 so that every product is 1: the column side reads a second copy of the planes
// with bits 0 and 2 of every nibble swapped, again one instruction per register.  Five instructions per word and plane, 30
// per step, against nine MFMAs.  (The first form used v_mfma_i32_16x16x64_i8 with one bit of every BYTE per
// register: 54 instructions and 18 MFMAs per step -- and back to back that instruction issues every 45 cycles
// (tools/exp/mfma_rate.hip; the FP4 form every 33): 800 cycles of matrix core per step, no faster than k_epi_pairs.)
//
// State and passes: see k_epi_pairs_mfma below (two passes over the samples, two waves per SIMD).
#pragma once
#include "hpgv_epi_kernels.h"

namespace hpgv {

constexpr int EPM_MAX_CHUNKS = 128;   // chunk descriptors kept in LDS (131 072 samples and their padding)
constexpr int EPM_TI = 16;            // rows of a workgroup's tile (its 64 columns: 16 per wave)
typedef int epm_v8i __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(3))) uint32_t *epm_lds_u32;           // a word of LDS by its 32-bit address
typedef float epm_v4f __attribute__((ext_vector_type(4)));

// the column side's copy of the planes: bits 0 and 2 of every nibble swapped (same layout, right behind the planes)
__device__ __forceinline__ uint32_t epm_swap02(uint32_t x) {
    return (x & 0xAAAAAAAAu) | ((x & 0x11111111u) << 2) | ((x >> 2) & 0x11111111u);
}
static __global__ void __launch_bounds__(256) k_epi_planes_rev(const uint32_t *__restrict__ planes, size_t n_words, uint32_t *__restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx < n_words) out[idx] = epm_swap02(planes[idx]);
}

// the operand registers of one MFMA (FP4: the first four of the eight): samples 4 m + 0 / 1 / 2 / 3 of the word weighted
// 0.5 / 1 / 2 / 0.5 on the row side and 2 / 1 / 0.5 / 2 on the column side (from the swapped copy)
__device__ __forceinline__ epm_v8i epm_row_operand(uint32_t x) {
    return epm_v8i{(int)(x & 0x11111111u), (int)(x & 0x22222222u), (int)(x & 0x44444444u), (int)((x >> 3) & 0x11111111u), 0, 0, 0, 0};
}
__device__ __forceinline__ epm_v8i epm_col_operand(uint32_t y) {
    return epm_v8i{(int)(y & 0x44444444u), (int)(y & 0x22222222u), (int)(y & 0x11111111u), (int)((y >> 1) & 0x44444444u), 0, 0, 0, 0};
}
#define HPGV_EPM_MFMA(A, B, C) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, C, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)   /* FP4 x FP4, scales 2^0 */

// Ranking only (thresholds + candidate lists), classes below 65 536 samples, any number of folds.  TWO passes over the
// samples, like the triple scan's: the first leaves every pair's nine totals (cases and controls in accumulators of their own,
// chosen step by step, packed cases low / controls high at its end), the second visits
// the groups in (fold, class) order and, each time a fold's groups are complete, evaluates that fold at once from totals -
// fold counts.  What this buys over one pass that keeps all folds' counts: the state is 36 + 36 + 36 registers instead of
// 36 (K + 1), so TWO waves share a SIMD -- one wave's vector work (operands, banking, evaluation) runs under the other's
// MFMAs, which a single wave's own instruction stream does not do -- there is one copy of the evaluation whatever the fold
// (its counts sit in fixed registers), and neither the fold count nor unequal classes change the code.  The second pass of
// MFMAs is the cheap part.
// Tiles: 16 rows x 64 columns, numbered column tile by column tile and dealt to the XCDs in spans like k_epi_pairs'
// (tile_base[c] = tiles before column tile i_begin / 64 + c; inside it the 16-row blocks from the band's first row down to the
// diagonal).  Staging as k_epi_pairs: an LDS image of (64 + 16) SNPs x 3 planes x 32 words per chunk, rows of 128 bytes,
// the 16-byte pieces of a row swizzled by the SNP so that the 64 lanes of a read (16 SNPs x 4 words) fall on 64 different
// banks; double buffered, one barrier per chunk.  The columns' rows come from the swapped copy (rev_off words behind the planes).
// COMPLETE: a dataset WITHOUT missing calls -- only the four cells of genotypes {0, 1} x {0, 1} are counted (two planes per SNP
// staged, four MFMAs per step) and the other five follow at a group's end from the per-SNP, per-group genotype counts `marg`
// (k_epi_marginals) exactly as in k_epi_pairs: n(a, 2) = n_i(a) - n(a, 0) - n(a, 1), n(2, b) = n_j(b) - n(0, b) - n(1, b), n(2, 2) =
// the rest of the group.  Its image is plane-major (row = plane * 80 + SNP: with two planes per SNP an SNP-major image would
// put the 16 SNPs of a read on 32 banks).
template <bool TRAINING, bool BALANCED, bool COMPLETE>
__global__ void __launch_bounds__(256, COMPLETE ? 3 : 2) k_epi_pairs_mfma(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ marg, uint32_t rev_off, int W,
                                                         int n_variants, int i_begin, int i_first, int i_end,
                                                         const unsigned *__restrict__ tile_base, int n_cols, unsigned n_tiles,
                                                         const EpiChunk *__restrict__ chunks, const uint32_t *__restrict__ chunk_cls /* bit k: step k of the chunk holds controls */, int n_chunks,
                                                         const EpiFold *__restrict__ folds, int num_folds, int n_affected, int n_unaffected,
                                                         const double *__restrict__ thr, EpiCand *__restrict__ cand,
                                                         unsigned *__restrict__ cand_count, unsigned cand_cap) {
    constexpr int NP = COMPLETE ? 2 : 3, NC = NP * NP;               // planes staged per SNP, cells counted
    constexpr int SNPS = EPI_TJ + EPM_TI, ROWS = SNPS * NP, NDMA = ROWS / 8;     // 80 SNPs, 240 (160) rows, 30 (20) LDS-DMA instructions per chunk
    static_assert(ROWS % 8 == 0, "whole LDS-DMA instructions");
    // two separate arrays, not lds[2][...]: the compiler then sees that the LDS-DMA writes into one buffer cannot touch the other
    // (128-byte aligned: the reads below fold the piece swizzle and the step into ONE xor on the word's LDS address)
    __shared__ __attribute__((aligned(128))) uint32_t lds_a[ROWS * EPI_CH];
    __shared__ __attribute__((aligned(128))) uint32_t lds_b[ROWS * EPI_CH];
    // the chunk descriptors and the folds' constants in LDS (a scalar load that misses costs a microsecond)
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[EPM_MAX_CHUNKS * 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_fold[EPI_MAX_FOLDS * 8];
    __shared__ uint32_t s_cls[EPM_MAX_CHUNKS];
    __shared__ uint32_t s_marg[COMPLETE ? 2 * EPI_MAX_FOLDS * SNPS : 1];        // complete data: genotype counts of the tile's SNPs per group, [g][SNP]
    const unsigned span = (n_tiles + 7u) / 8u;
    const unsigned tile = (blockIdx.x & 7u) * span + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    int c_lo = 0, c_hi = n_cols;
    while (c_hi - c_lo > 1) { const int mid = (c_lo + c_hi) >> 1; if (tile_base[mid] <= tile) c_lo = mid; else c_hi = mid; }
    for (int q = threadIdx.x; q < n_chunks * 4; q += 256) s_chunk[q] = reinterpret_cast<const uint32_t *>(chunks)[q];
    for (int q = threadIdx.x; q < n_chunks; q += 256) s_cls[q] = chunk_cls[q];
    for (int q = threadIdx.x; q < num_folds * 8; q += 256) {         // per fold: test_a, test_u, inv_a, inv_u, the threshold
        const int f = q >> 3, e = q & 7;
        s_fold[q] = e < 6 ? reinterpret_cast<const uint32_t *>(folds + f)[e] : reinterpret_cast<const uint32_t *>(thr + f)[e - 6];
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    const int jt = ((i_begin >> 6) + c_lo) * EPI_TJ, j0 = jt + 16 * wave, i0 = i_begin + (int)(tile - tile_base[c_lo]) * EPM_TI;
    const bool active = !(j0 + 15 <= i0 || i0 >= i_end || i0 + 15 < i_first);       // is any pair of this wave's block asked for
    if constexpr (COMPLETE)
        for (int q = threadIdx.x; q < 2 * num_folds * SNPS; q += 256) {
            const int g = q / SNPS, e = q % SNPS;
            s_marg[q] = marg[(size_t)(e < EPI_TJ ? jt + e : i0 + (e - EPI_TJ)) * (2 * EPI_MAX_FOLDS) + g];
        }

    // LDS-DMA: one global_load_lds_dwordx4 = 8 rows of the image; lane l fetches row 8 k + l / 8, physical piece l % 8 = the
    // logical piece (l % 8) ^ swizzle(row's SNP).  Kept per instruction: the word offset of the lane's piece in the planes.
    uint32_t dma_off[8];
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const int k = wave + 4 * q, row8 = 8 * k + (lane >> 3), row = row8 < ROWS ? row8 : 0;
        const int snp_idx = COMPLETE ? row % SNPS : row / 3, plane = COMPLETE ? row / SNPS : row % 3;
        const int piece = (lane & 7) ^ ((snp_idx >> 1) & 7);
        const int snp = snp_idx < EPI_TJ ? jt + snp_idx : i0 + (snp_idx - EPI_TJ);
        dma_off[q] = ((uint32_t)snp * 3u + (uint32_t)plane) * (uint32_t)W + (uint32_t)piece * 4u + (snp_idx < EPI_TJ ? rev_off : 0u);
    }
    auto load_chunk = [&](uint32_t w0, uint32_t *dst) {
        #pragma unroll
        for (int q = 0; q < 8; q++) {
            const int k = wave + 4 * q;
            if (k < NDMA)
                __builtin_amdgcn_global_load_lds(planes + (dma_off[q] + w0), (__attribute__((address_space(3))) uint32_t *)(dst + 8 * k * EPI_CH), 16, 0, 0);
        }
    };
    load_chunk(chunks[0].w0, lds_a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                 // (the tables above are in place too)

    // byte offsets of the lane's words in the image: row (SNP * 3 + plane) * 128 + piece (step ^ swizzle) * 16 + h * 4
    const int sa = EPI_TJ + r, sb = 16 * wave + r;
    constexpr int PS = COMPLETE ? SNPS * EPI_CH * 4 : EPI_CH * 4;    // bytes from one plane of an SNP to the next
    const int base_a = sa * (COMPLETE ? 1 : 3) * (EPI_CH * 4) + h * 4, base_b = sb * (COMPLETE ? 1 : 3) * (EPI_CH * 4) + h * 4;
    const int swz_a = (sa >> 1) & 7, swz_b = (sb >> 1) & 7;
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    const int j = j0 + r;
    bool asked[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = i0 + 4 * h + q;
        asked[q] = !(i < i_first || i >= i_end || i >= n_variants || j >= n_variants || j <= i);
    }

    epm_v4f acc[NC], acc_u[NC];                                       // (acc_u: the first pass's controls; dead when `part` comes to life)
    uint32_t totp[9][4], part[9][4];                                 // totals; the fold under way (cases low, controls high halves)
    #pragma unroll
    for (int c = 0; c < NC; c++) { acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f}; acc_u[c] = epm_v4f{0.f, 0.f, 0.f, 0.f}; }
    #pragma unroll
    for (int c = 0; c < 9; c++)
        #pragma unroll
        for (int q = 0; q < 4; q++) { totp[c][q] = 0; part[c][q] = 0; }
    // the nine cells of one pair in one group from what was counted (all nine, or the four of genotypes 0 / 1 and the genotype
    // counts of the pair's SNPs in that group: mi, mj = count of genotype 0 | count of genotype 1 << 16; ng = the group's size)
    auto cells_of = [&](const epm_v4f (&x)[NC], int q, uint32_t mi, uint32_t mj, uint32_t ng, uint32_t (&cell)[9]) {
        auto at = [&](int c) { return (uint32_t)(q == 0 ? x[c].x : q == 1 ? x[c].y : q == 2 ? x[c].z : x[c].w); };
        if constexpr (COMPLETE) {
            const uint32_t n00 = at(0), n01 = at(1), n10 = at(2), n11 = at(3);
            const uint32_t mi0 = mi & 0xFFFFu, mi1 = mi >> 16, mj0 = mj & 0xFFFFu, mj1 = mj >> 16;
            cell[0] = n00; cell[1] = n01; cell[2] = mi0 - n00 - n01;
            cell[3] = n10; cell[4] = n11; cell[5] = mi1 - n10 - n11;
            cell[6] = mj0 - n00 - n10; cell[7] = mj1 - n01 - n11;
            cell[8] = ng - mi0 - mi1 - cell[6] - cell[7];
        } else {
            #pragma unroll
            for (int c = 0; c < 9; c++) cell[c] = at(c < NC ? c : 0);
        }
    };

    // the word of step k: row * 128 + ((k ^ swizzle) << 4) + h * 4 bytes into the buffer = (buffer + row * 128 + (swizzle << 4) + h * 4) ^ (k << 4)
#define HPGV_EPM_READ(X, Y, KSTEP)                                                                       \
    {                                                                                                    \
        const uint32_t kx = (uint32_t)((KSTEP) < 7 ? (KSTEP) : 7) << 4;      /* (past the chunk's last step: any step, never used) */ \
        _Pragma("unroll") for (int a = 0; a < NP; a++) {                                                 \
            X[a] = *(epm_lds_u32)(uintptr_t)((cur_a ^ kx) + (uint32_t)(a * PS)); Y[a] = *(epm_lds_u32)(uintptr_t)((cur_b ^ kx) + (uint32_t)(a * PS)); \
        }                                                                                                \
    }
    // a (fold, class) group has ended: the accumulators hold its counts.  First pass: into the totals.  Second pass: into the
    // fold under way; when the fold has no further group, it is evaluated and the fold under way starts empty again.
    auto bank_fold = [&](int g) {
        const int f = g >> 1, sh = (g & 1) * 16;
        {
            {
                uint32_t mj = 0, ng = 0;
                if constexpr (COMPLETE) {
                    mj = s_marg[g * SNPS + sb];
                    const int n = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + (g & 1)]);
                    ng = n > 0 ? (uint32_t)n : 0u;
                }
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t cell[9];
                    cells_of(acc, q, COMPLETE ? s_marg[g * SNPS + EPI_TJ + 4 * h + q] : 0u, mj, ng, cell);
                    #pragma unroll
                    for (int c = 0; c < 9; c++) part[c][q] += cell[c] << sh;
                }
                #pragma unroll
                for (int c = 0; c < NC; c++) acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f};
            }
            EpiFold fo;
            fo.test_a = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8]); fo.test_u = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 1]);
            if ((g & 1) || fo.test_u <= 0) {                         // the fold's last group (its controls, or its cases when it has no controls)
                fo.inv_a = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 2]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 3]) << 32));
                fo.inv_u = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 4]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 5]) << 32));
                const double thr_f = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 6]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 7]) << 32));
                const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
                const float finv_a = (float)fo.inv_a, finv_u = (float)fo.inv_u, fthr = (float)thr_f;
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (asked[q]) {
                        uint32_t sel = 0;                            // TP (low half), FP (high half)
                        #pragma unroll
                        for (int c = 0; c < 9; c++) {
                            const uint32_t in = part[c][q], tr = totp[c][q] - in;
                            bool high;
                            // balanced: cases >= controls, compared as (cases:controls) >= (controls:cases); an empty training cell is not
                            // high risk (on the testing part its samples must stay out: compared against max(tr, 1))
                            if constexpr (BALANCED) high = __builtin_amdgcn_alignbit(tr, tr, 16) >= (TRAINING ? tr : (tr > 1u ? tr : 1u));
                            else high = mdr_high_risk<false>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu);
                            sel += high ? (TRAINING ? tr : in) : 0u;
                        }
                        const int tp = (int)(sel & 0xFFFFu), fp = (int)(sel >> 16);
                        // nearly every model is far below its fold's threshold: single precision (error below 1e-6) says so at a fraction of the cost
                        if (!(0.5f * ((float)tp * finv_a + (float)(size_u - fp) * finv_u) + 1e-5f < fthr)) {
                            const double TP = (double)tp, TN = (double)(size_u - fp), ya = (double)size_a, yu = (double)size_u;
                            double qa = TP * fo.inv_a, qu = TN * fo.inv_u;       // the two quotients, correctly rounded (Markstein)
                            qa = __builtin_fma(__builtin_fma(-qa, ya, TP), fo.inv_a, qa);
                            qu = __builtin_fma(__builtin_fma(-qu, yu, TN), fo.inv_u, qu);
                            const double accy = (qa + qu) / 2;
                            if (accy >= thr_f) {
                                uint32_t mask = 0;
                                #pragma unroll
                                for (int c = 0; c < 9; c++) {
                                    const uint32_t tr = totp[c][q] - part[c][q];
                                    if (mdr_high_risk<BALANCED>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu)) mask |= 1u << c;
                                }
                                const unsigned slot = atomicAdd(&cand_count[f], 1u);
                                if (slot < cand_cap) {
                                    EpiCand e;
                                    e.accuracy = accy; e.i = i0 + 4 * h + q; e.j = j; e.risky = mask; e.pad = 0;
                                    cand[(size_t)f * cand_cap + slot] = e;
                                }
                            }
                        }
                    }
                }
                #pragma unroll
                for (int c = 0; c < 9; c++)
                    #pragma unroll
                    for (int q = 0; q < 4; q++) part[c][q] = 0;
            }
        }
    };

    // n_chunks chunks, twice (a copy of the code per pass: what a group's end does differs; written as one loop the compiler
    // merges the two with a select per register); the next chunk's loads fly into the other buffer during the work on this
    // one.  Inside a chunk the words are read from the image a step ahead and the next step's operand registers are made
    // beside this step's MFMAs.
    uint32_t *cur = lds_a, *nxt = lds_b;
#define HPGV_EPM_PASS(PASS)                                                                              \
    for (int c = 0; c < n_chunks; c++) {                                                                 \
        if (PASS == 0 || c + 1 < n_chunks)                                                               \
            load_chunk((uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[(c + 1 < n_chunks ? c + 1 : 0) * 4]), nxt); \
        if (active) {                                                                                    \
            const int ns = __builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 1]) >> 2;                 \
            const uint32_t clsm = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cls[c]);             \
            const uint64_t flush = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 2])      /* wave-uniform */ \
                                   | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 3]) << 32); \
            const uint32_t cur_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)cur;     \
            const uint32_t cur_a = cur_lds + (uint32_t)(base_a | (swz_a << 4)), cur_b = cur_lds + (uint32_t)(base_b | (swz_b << 4)); \
            uint32_t xa[NP], xb[NP];                                                                     \
            HPGV_EPM_READ(xa, xb, 0)                                                                     \
            epm_v8i A0[NP], B0[NP];                                                                      \
            _Pragma("unroll") for (int a = 0; a < NP; a++) { A0[a] = epm_row_operand(xa[a]); B0[a] = epm_col_operand(xb[a]); } \
            for (int k = 0; k < ns; k++) {                                                               \
                uint32_t na[NP], nb[NP];                                                                 \
                HPGV_EPM_READ(na, nb, k + 1)                                                             \
                if (PASS == 0 && ((clsm >> k) & 1u)) {               /* first pass: cases and controls each into accumulators of their own, no banking */ \
                    _Pragma("unroll") for (int a = 0; a < NP; a++)                                       \
                        _Pragma("unroll") for (int b = 0; b < NP; b++) acc_u[a * NP + b] = HPGV_EPM_MFMA(A0[a], B0[b], acc_u[a * NP + b]); \
                } else {                                                                                 \
                    _Pragma("unroll") for (int a = 0; a < NP; a++)                                       \
                        _Pragma("unroll") for (int b = 0; b < NP; b++) acc[a * NP + b] = HPGV_EPM_MFMA(A0[a], B0[b], acc[a * NP + b]); \
                }                                                                                        \
                _Pragma("unroll") for (int a = 0; a < NP; a++) { A0[a] = epm_row_operand(na[a]); B0[a] = epm_col_operand(nb[a]); } \
                const int g = (int)((flush >> (8 * k)) & 0xFFu);     /* the same in every lane */         \
                /* (the banking clears the accumulators: starting a group's first MFMAs from a zero operand instead needs two \
                   copies of the step, and measured slower) */                                           \
                if (PASS == 1 && g != 0xFF) bank_fold(g);                                                \
            }                                                                                            \
        }                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             /* this wave's part of the next chunk has landed */ \
        __syncthreads();                                                                                 \
        uint32_t *t = cur; cur = nxt; nxt = t;                                                           \
    }
    HPGV_EPM_PASS(0)
    {                                                                // the totals, packed; the second pass starts from zero
        uint32_t mj[2] = {0, 0}, mi[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        if constexpr (COMPLETE)                                      // the SNPs' genotype counts per class: sums over the class's groups (below 65 536 each)
            for (int g = 0; g < 2 * num_folds; g++) {
                mj[g & 1] += s_marg[g * SNPS + sb];
                #pragma unroll
                for (int q = 0; q < 4; q++) mi[q][g & 1] += s_marg[g * SNPS + EPI_TJ + 4 * h + q];
            }
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t ca[9], cu[9];
            cells_of(acc, q, mi[q][0], mj[0], (uint32_t)n_affected, ca);
            cells_of(acc_u, q, mi[q][1], mj[1], (uint32_t)n_unaffected, cu);
            #pragma unroll
            for (int c = 0; c < 9; c++) totp[c][q] = ca[c] + (cu[c] << 16);
        }
        #pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f};
    }
    HPGV_EPM_PASS(1)
#undef HPGV_EPM_PASS
#undef HPGV_EPM_READ
}

// ---------------------------------------------------------------------------------------------------------------------
// Triples (i, j, k), i < j < k, on the matrix cores.  For a fixed first SNP i and a fixed genotype a of it, the nine cells
// (a, b, c) of the triples of 16 second SNPs j x 16 third SNPs k are nine products again: A = plane_i[a] & plane_j[b]
// (one more AND per word), B = plane_k[c].  A wave owns (i, 16 j, 16 k) -- four triples per lane --, walks the samples once
// per genotype a like k_epi_triples3 (a cell's verdict is its own: it only adds to its fold's TP | FP sum and high-risk
// bits), and every walk is k_epi_pairs_mfma's two passes.  Across the walks a lane keeps K x 4 sums and masks; everything
// else is the pair kernel's state, so two waves share a SIMD here too (up to 10 folds; with 11 to 16 the sums and masks take a
// wave the whole register file of its SIMD).  Image: 64 k columns (swapped copy) + 16 j rows + the
// i row, three planes each.  Tiles: first SNP i, then the j blocks of 16 from (i + 1) / 16 on, then the k tiles of 64 from the
// one that holds 16 jb + 1 on (row_base / jb_prefix as k_epi_triples3's, with blocks of 16).
template <int K, bool TRAINING, bool BALANCED>
__global__ void __launch_bounds__(256, K <= 10 ? 2 : 1) k_epi_triples_mfma(const uint32_t *__restrict__ planes, uint32_t rev_off, int W, int n_variants, int i_first,
                                                           const unsigned *__restrict__ row_base /* n_i + 1 */, int n_i,
                                                           const unsigned *__restrict__ jb_prefix /* n_jb + 1 */, int n_jb,
                                                           const EpiChunk *__restrict__ chunks, int n_chunks,
                                                           const EpiFold *__restrict__ folds, int num_folds, int n_affected, int n_unaffected,
                                                           const double *__restrict__ thr, EpiCand3 *__restrict__ cand,
                                                           unsigned *__restrict__ cand_count, unsigned cand_cap) {
    constexpr int SNPS = EPI_TJ + EPM_TI + 1, ROWS = SNPS * 3, NDMA = (ROWS + 7) / 8;     // 81 SNPs, 243 rows, 31 LDS-DMA instructions per chunk
    __shared__ __attribute__((aligned(16))) uint32_t lds_a[NDMA * 8 * EPI_CH];
    __shared__ __attribute__((aligned(16))) uint32_t lds_b[NDMA * 8 * EPI_CH];
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[EPM_MAX_CHUNKS * 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_fold[EPI_MAX_FOLDS * 8];
    int r_lo = 0, r_hi = n_i;
    while (r_hi - r_lo > 1) { const int mid = (r_lo + r_hi) >> 1; if (row_base[mid] <= blockIdx.x) r_lo = mid; else r_hi = mid; }
    const int i = i_first + r_lo;
    const int jb_min = (i + 1) >> 4;
    const unsigned want = (blockIdx.x - row_base[r_lo]) + jb_prefix[jb_min];
    int b_lo = jb_min, b_hi = n_jb;
    while (b_hi - b_lo > 1) { const int mid = (b_lo + b_hi) >> 1; if (jb_prefix[mid] <= want) b_lo = mid; else b_hi = mid; }
    const int j0 = b_lo * EPM_TI;
    const int kt = (((j0 + 1) >> 6) + (int)(want - jb_prefix[b_lo])) * EPI_TJ;
    for (int q = threadIdx.x; q < n_chunks * 4; q += 256) s_chunk[q] = reinterpret_cast<const uint32_t *>(chunks)[q];
    for (int q = threadIdx.x; q < num_folds * 8; q += 256) {
        const int f = q >> 3, e = q & 7;
        s_fold[q] = e < 6 ? reinterpret_cast<const uint32_t *>(folds + f)[e] : reinterpret_cast<const uint32_t *>(thr + f)[e - 6];
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, h = lane >> 4;
    const int k0 = kt + 16 * wave, k = k0 + r;
    const bool active = k0 + 15 > j0 && j0 + 15 > i;                 // is any triple of this wave's block in order
    uint32_t dma_off[8];
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const int d = wave + 4 * q, row8 = 8 * d + (lane >> 3), row = row8 < ROWS ? row8 : 0;
        const int snp_idx = row / 3, plane = row % 3;
        const int piece = (lane & 7) ^ ((snp_idx >> 1) & 7);
        const int snp = snp_idx < EPI_TJ ? kt + snp_idx : snp_idx < EPI_TJ + EPM_TI ? j0 + (snp_idx - EPI_TJ) : i;
        dma_off[q] = ((uint32_t)snp * 3u + (uint32_t)plane) * (uint32_t)W + (uint32_t)piece * 4u + (snp_idx < EPI_TJ ? rev_off : 0u);
    }
    auto load_chunk = [&](uint32_t w0, uint32_t *dst) {
        #pragma unroll
        for (int q = 0; q < 8; q++) {
            const int d = wave + 4 * q;
            if (d < NDMA)
                __builtin_amdgcn_global_load_lds(planes + (dma_off[q] + w0), (__attribute__((address_space(3))) uint32_t *)(dst + 8 * d * EPI_CH), 16, 0, 0);
        }
    };
    load_chunk(chunks[0].w0, lds_a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int sj = EPI_TJ + r, sk = 16 * wave + r, si = EPI_TJ + EPM_TI;
    const int base_j = sj * 3 * (EPI_CH * 4) + h * 4, base_k = sk * 3 * (EPI_CH * 4) + h * 4, base_i = si * 3 * (EPI_CH * 4) + h * 4;
    const int swz = (sj >> 1) & 7, swz_i = (si >> 1) & 7;            // ((sk >> 1) & 7 is the same value)
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;
    bool asked[4];
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = j0 + 4 * h + q;
        asked[q] = j > i && k > j && j < n_variants && k < n_variants;
    }
    uint32_t sel[K][4], mask[K][4];                                  // per fold and triple: TP (low half) | FP (high half); the high-risk cells
    #pragma unroll
    for (int f = 0; f < K; f++)
        #pragma unroll
        for (int q = 0; q < 4; q++) { sel[f][q] = 0; mask[f][q] = 0; }

    uint32_t *cur = lds_a, *nxt = lds_b;
    #pragma unroll 1
    for (int a = 0; a < 3; a++) {
        epm_v4f acc[9];
        uint32_t totp[9][4], part[9][4];
        #pragma unroll
        for (int c = 0; c < 9; c++) {
            acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int q = 0; q < 4; q++) { totp[c][q] = 0; part[c][q] = 0; }
        }
        auto bank_totals = [&](int g) {
            const int sh = (g & 1) * 16;
            #pragma unroll
            for (int c = 0; c < 9; c++) {
                totp[c][0] += (uint32_t)acc[c].x << sh; totp[c][1] += (uint32_t)acc[c].y << sh;
                totp[c][2] += (uint32_t)acc[c].z << sh; totp[c][3] += (uint32_t)acc[c].w << sh;
                acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto bank_fold = [&](int g) {
            const int f = g >> 1, sh = (g & 1) * 16;
            #pragma unroll
            for (int c = 0; c < 9; c++) {
                part[c][0] += (uint32_t)acc[c].x << sh; part[c][1] += (uint32_t)acc[c].y << sh;
                part[c][2] += (uint32_t)acc[c].z << sh; part[c][3] += (uint32_t)acc[c].w << sh;
                acc[c] = epm_v4f{0.f, 0.f, 0.f, 0.f};
            }
            const int test_u = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 1]);
            if ((g & 1) || test_u <= 0) {                            // the fold's last group: the nine cells' verdicts
                uint32_t ds[4], dm[4];
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    ds[q] = 0; dm[q] = 0;
                    #pragma unroll
                    for (int c = 0; c < 9; c++) {
                        const uint32_t in = part[c][q], tr = totp[c][q] - in;
                        bool high;
                        if constexpr (BALANCED) high = __builtin_amdgcn_alignbit(tr, tr, 16) >= (TRAINING ? tr : (tr > 1u ? tr : 1u));   // as k_epi_pairs
                        else high = mdr_high_risk<false>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu);
                        ds[q] += high ? (TRAINING ? tr : in) : 0u;
                        // (balanced, training part: an EMPTY cell passes the packed comparison and adds nothing to the sums; it is not high risk)
                        dm[q] |= (high && tr != 0u) ? 1u << c : 0u;
                        part[c][q] = 0;
                    }
                    dm[q] <<= 9 * a;
                }
                switch (f) {
#define HPGV_EPM_CASE(FF)                                                                                \
                    case FF:                                                                             \
                        if constexpr (FF < K) {                                                          \
                            _Pragma("unroll") for (int q = 0; q < 4; q++) { sel[FF < K ? FF : 0][q] += ds[q]; mask[FF < K ? FF : 0][q] |= dm[q]; } \
                        }                                                                                \
                        break;
                    HPGV_EPM_CASE(0) HPGV_EPM_CASE(1) HPGV_EPM_CASE(2) HPGV_EPM_CASE(3) HPGV_EPM_CASE(4)
                    HPGV_EPM_CASE(5) HPGV_EPM_CASE(6) HPGV_EPM_CASE(7) HPGV_EPM_CASE(8) HPGV_EPM_CASE(9)
                    HPGV_EPM_CASE(10) HPGV_EPM_CASE(11) HPGV_EPM_CASE(12) HPGV_EPM_CASE(13) HPGV_EPM_CASE(14) HPGV_EPM_CASE(15)
#undef HPGV_EPM_CASE
                    default: break;
                }
            }
        };
#define HPGV_EPM3_READ(XI, YJ, ZK, KSTEP)                                                                \
        {                                                                                                \
            const int k_ = (KSTEP) < 7 ? (KSTEP) : 7;                                                    \
            const char *qj = cur_bytes + (base_j + ((k_ ^ swz) << 4)), *qk = cur_bytes + (base_k + ((k_ ^ swz) << 4)); \
            XI = *reinterpret_cast<const uint32_t *>(cur_bytes + (base_i + ((k_ ^ swz_i) << 4)) + a * (EPI_CH * 4)); \
            _Pragma("unroll") for (int b = 0; b < 3; b++) {                                              \
                YJ[b] = *reinterpret_cast<const uint32_t *>(qj + b * (EPI_CH * 4)); ZK[b] = *reinterpret_cast<const uint32_t *>(qk + b * (EPI_CH * 4)); \
            }                                                                                            \
        }
#define HPGV_EPM3_PASS(PASS)                                                                             \
        for (int c = 0; c < n_chunks; c++) {                                                             \
            if (!(PASS == 1 && a == 2 && c + 1 == n_chunks))                                             \
                load_chunk((uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[(c + 1 < n_chunks ? c + 1 : 0) * 4]), nxt); \
            if (active) {                                                                                \
                const int ns = __builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 1]) >> 2;             \
                const uint64_t flush = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 2]) \
                                       | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_chunk[c * 4 + 3]) << 32); \
                const char *cur_bytes = reinterpret_cast<const char *>(cur);                             \
                uint32_t xi, yj[3], zk[3];                                                               \
                HPGV_EPM3_READ(xi, yj, zk, 0)                                                            \
                epm_v8i A0[3], B0[3];                                                                    \
                _Pragma("unroll") for (int b = 0; b < 3; b++) { A0[b] = epm_row_operand(xi & yj[b]); B0[b] = epm_col_operand(zk[b]); } \
                for (int st = 0; st < ns; st++) {                                                        \
                    uint32_t nxi, nyj[3], nzk[3];                                                        \
                    HPGV_EPM3_READ(nxi, nyj, nzk, st + 1)                                                \
                    _Pragma("unroll") for (int b = 0; b < 3; b++)                                        \
                        _Pragma("unroll") for (int d = 0; d < 3; d++) acc[b * 3 + d] = HPGV_EPM_MFMA(A0[b], B0[d], acc[b * 3 + d]); \
                    _Pragma("unroll") for (int b = 0; b < 3; b++) { A0[b] = epm_row_operand(nxi & nyj[b]); B0[b] = epm_col_operand(nzk[b]); } \
                    const int g = (int)((flush >> (8 * st)) & 0xFFu);                                    \
                    if (g != 0xFF) { if (PASS == 0) bank_totals(g); else bank_fold(g); }                 \
                }                                                                                        \
            }                                                                                            \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
            __syncthreads();                                                                             \
            uint32_t *t = cur; cur = nxt; nxt = t;                                                       \
        }
        HPGV_EPM3_PASS(0)
        HPGV_EPM3_PASS(1)
#undef HPGV_EPM3_PASS
#undef HPGV_EPM3_READ
    }
    if (!active) return;

    #pragma unroll 1
    for (int f = 0; f < K; f++) {
        if (f >= num_folds) break;
        uint32_t s4[4], m4[4];
        switch (f) {
#define HPGV_EPM_CASE(FF)                                                                                \
            case FF:                                                                                     \
                if constexpr (FF < K) { _Pragma("unroll") for (int q = 0; q < 4; q++) { s4[q] = sel[FF < K ? FF : 0][q]; m4[q] = mask[FF < K ? FF : 0][q]; } } \
                break;
            HPGV_EPM_CASE(0) HPGV_EPM_CASE(1) HPGV_EPM_CASE(2) HPGV_EPM_CASE(3) HPGV_EPM_CASE(4)
            HPGV_EPM_CASE(5) HPGV_EPM_CASE(6) HPGV_EPM_CASE(7) HPGV_EPM_CASE(8) HPGV_EPM_CASE(9)
            HPGV_EPM_CASE(10) HPGV_EPM_CASE(11) HPGV_EPM_CASE(12) HPGV_EPM_CASE(13) HPGV_EPM_CASE(14) HPGV_EPM_CASE(15)
#undef HPGV_EPM_CASE
            default:
                #pragma unroll
                for (int q = 0; q < 4; q++) { s4[q] = 0; m4[q] = 0; }
                break;
        }
        EpiFold fo;
        fo.test_a = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8]); fo.test_u = __builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 1]);
        if (fo.test_a < 0) continue;
        fo.inv_a = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 2]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 3]) << 32));
        fo.inv_u = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 4]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 5]) << 32));
        const double thr_f = __builtin_bit_cast(double, (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 6]) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)s_fold[f * 8 + 7]) << 32));
        const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!asked[q]) continue;
            const int tp = (int)(s4[q] & 0xFFFFu), fp = (int)(s4[q] >> 16);
            const double TP = (double)tp, TN = (double)(size_u - fp), ya_ = (double)size_a, yu_ = (double)size_u;
            double qa = TP * fo.inv_a, qu = TN * fo.inv_u;           // the two quotients as in k_epi_pairs (Markstein)
            qa = __builtin_fma(__builtin_fma(-qa, ya_, TP), fo.inv_a, qa);
            qu = __builtin_fma(__builtin_fma(-qu, yu_, TN), fo.inv_u, qu);
            const double accy = (qa + qu) / 2;
            if (accy >= thr_f) {
                const unsigned slot = atomicAdd(&cand_count[f], 1u);
                if (slot < cand_cap) {
                    EpiCand3 e;
                    e.accuracy = accy; e.i = i; e.j = j0 + 4 * h + q; e.k = k; e.risky = m4[q];
                    cand[(size_t)f * cand_cap + slot] = e;
                }
            }
        }
    }
}

#undef HPGV_EPM_MFMA

}  // namespace hpgv
