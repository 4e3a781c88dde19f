// hpgv_epi_triples3_kernels.h -- the triple scan of the ranking with the 27 cells taken NINE AT A TIME (round 4).
//
// The one-pass kernel (k_epi_triples1) keeps the 27 counts of all K folds in one lane: 27 K / 2 registers besides the
// running counts and the operands -- the whole register file of a SIMD, ONE wave per SIMD, which issues a vector instruction
// every ~5 cycles whatever it is (profiles/r03_valu_instruction_costs.txt): 38 - 45 % of the count's own time.
//
// A cell's verdict is its own: whether cell c is high risk in fold f depends on the training counts of c alone, and what it
// adds to the fold's TP / FP is its own evaluated count.  So the cells need not be together.  Here a wave still owns one
// (i, j) pair and its lanes 64 third SNPs k, but it walks the samples THREE times, once per genotype a of SNP i, with the nine
// cells (a, b, c) of that genotype: 9 K / 2 + 9 count registers.  After each walk the nine cells' verdicts go into per-fold
// sums (TP | FP packed) and per-fold 27-bit masks, the counts are dropped.  Same number of counting operations as one pass
// (three walks of nine cells), a third of the state: three waves per SIMD.
//
// Operands: the planes of i (one) and j (three) are the same for the whole wave.  They come through the SCALAR cache
// (s_load_dwordx4 from the planes, L2 resident), their products x_a & y_b are three s_and_b32 per word, and a cell costs the
// lane one v_and_b32 with a scalar operand (the 2-cycle class) and one v_bcnt_u32_b32 with accumulate.  Only the 64 x 3 planes
// of the k columns stream through LDS (LDS-DMA, double buffered), a third of the one-pass kernel's LDS reads per walk.
#pragma once
#include "hpgv_epi_kernels.h"

namespace hpgv {

// wave-uniform operands through the scalar cache: a load from the CONSTANT address space at a uniform address is an
// s_load_dwordx4 (a plain global load here would be a vector load of 64 identical addresses -- and four of those per step
// keep the texture path busier than the counting keeps the SIMD).  The planes are not written while a scan runs.
typedef uint32_t epi_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) epi_u32x4 *epi_const4;

template <int K, bool TRAINING, bool BALANCED>
__global__ void __launch_bounds__(256, 3) k_epi_triples3(const uint32_t *__restrict__ planes, int W, int n_variants, int i_first,
                                                          const unsigned *__restrict__ row_base /* n_i + 1 */, int n_i,
                                                          const unsigned *__restrict__ jb_prefix /* n_jb + 1 */, int n_jb,
                                                          const EpiChunk *__restrict__ chunks, int n_chunks,
                                                          const EpiFold *__restrict__ folds, int n_affected, int n_unaffected,
                                                          const double *__restrict__ thr, EpiCand3 *__restrict__ cand,
                                                          unsigned *__restrict__ cand_count, unsigned cand_cap) {
    constexpr int RP = EPI_CH;
    constexpr int ZROWS = EPI_TJ * 3;                                // 192 rows of 32 words: the k columns' planes
    __shared__ __attribute__((aligned(16))) uint32_t lds_a[ZROWS * RP];
    __shared__ __attribute__((aligned(16))) uint32_t lds_b[ZROWS * RP];
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
    const int j = j0 + wave_u, k = k0 + lane;
    // staging of the k columns: one global_load_lds_dwordx4 = 8 rows of 32 words; lane l fetches row 8 q + l / 8, piece l % 8,
    // the 16-byte pieces of a row swizzled by the column index so that the per-lane reads of 16 lanes fall on 16 slots
    uint32_t dma_off[6];
    #pragma unroll
    for (int r = 0; r < 6; r++) {
        const int q = wave + 4 * r, row = 8 * q + (lane >> 3), sidx = row / 3;      // 24 instructions x 8 rows = 192 rows
        const int piece = (lane & 7) ^ ((sidx >> 1) & 7);
        dma_off[r] = ((uint32_t)(k0 + sidx) * 3u + (uint32_t)(row % 3)) * (uint32_t)W + (uint32_t)piece * 4u;
    }
    auto load_chunk = [&](int c, uint32_t *dst) {
        const uint32_t w0 = chunks[c].w0;
        #pragma unroll
        for (int r = 0; r < 6; r++)
            __builtin_amdgcn_global_load_lds(planes + (dma_off[r] + w0), (__attribute__((address_space(3))) uint32_t *)(dst + 8 * (wave_u + 4 * r) * EPI_CH), 16, 0, 0);
    };
    // the wave's scalar operands: plane a of SNP i, the three planes of SNP j (rows past the dataset are zero planes)
    const epi_const4 yrow = (epi_const4)(planes + (size_t)j * 3u * (size_t)W);
    const int W4 = W >> 2;                                           // (the planes' rows are whole 4-word steps)
    const int swz_z = (lane >> 1) & 7;

    uint32_t sel[K], mask[K];                                        // per fold: TP (low half) | FP (high half); the high-risk cells
    #pragma unroll
    for (int f = 0; f < K; f++) { sel[f] = 0; mask[f] = 0; }
    const float f_na = (float)(unsigned)n_affected, f_nu = (float)(unsigned)n_unaffected;
    const float ratio = f_na / f_nu;

    #pragma unroll 1
    for (int a = 0; a < 3; a++) {
        const epi_const4 xrow = (epi_const4)(planes + ((size_t)i * 3u + (size_t)a) * (size_t)W);
        uint32_t packed[K][9], run[9];
        #pragma unroll
        for (int f = 0; f < K; f++)
            #pragma unroll
            for (int c = 0; c < 9; c++) packed[f][c] = 0;
        #pragma unroll
        for (int c = 0; c < 9; c++) run[c] = 0;

        __syncthreads();                                             // the walk before this one has read its last chunk
        load_chunk(0, lds_a);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#define HPGV_EPI3B_CHUNK(CUR, NXT)                                                                       \
        {                                                                                                \
            if (c + 1 < n_chunks) load_chunk(c + 1, NXT);                                                \
            const int nw = (int)chunks[c].nw;                                                            \
            const uint32_t w0 = chunks[c].w0;                                                            \
            const uint64_t flush = chunks[c].flush;                                                      \
            const char *cur_bytes = reinterpret_cast<const char *>(CUR);                                 \
            const int zoff = (lane * 3 * RP + (swz_z << 2)) * 4;                                         \
            for (int s = 0; s < nw; s += 4) {                                                            \
                const epi_u32x4 x = xrow[(w0 + s) >> 2];                                                 \
                uint4 z[3];                                                                              \
                _Pragma("unroll") for (int d = 0; d < 3; d++)                                            \
                    z[d] = *reinterpret_cast<const uint4 *>(cur_bytes + ((zoff ^ (s << 2)) + d * RP * 4)); \
                _Pragma("unroll") for (int b = 0; b < 3; b++) {                                          \
                    const epi_u32x4 y = yrow[b * W4 + ((w0 + s) >> 2)];                                  \
                    const uint32_t xy0 = x.x & y.x, xy1 = x.y & y.y, xy2 = x.z & y.z, xy3 = x.w & y.w;   \
                    _Pragma("unroll") for (int d = 0; d < 3; d++) {                                      \
                        uint32_t r = run[b * 3 + d];                                                     \
                        r = bcnt_acc(xy0 & z[d].x, r); r = bcnt_acc(xy1 & z[d].y, r);                    \
                        r = bcnt_acc(xy2 & z[d].z, r); r = bcnt_acc(xy3 & z[d].w, r);                    \
                        run[b * 3 + d] = r;                                                              \
                    }                                                                                    \
                }                                                                                        \
                const int g = (int)((flush >> (2 * s)) & 0xFFu);                                         \
                if (g != 0xFF) {                 /* a (fold, class) group ends here: bank its nine counts */ \
                    const int f = g >> 1, sh = (g & 1) * 16;                                             \
                    _Pragma("unroll") for (int ff = 0; ff < K; ff++)                                     \
                        if (ff == f) {                                                                   \
                            _Pragma("unroll") for (int cc = 0; cc < 9; cc++) packed[ff][cc] += run[cc] << sh; \
                        }                                                                                \
                    _Pragma("unroll") for (int cc = 0; cc < 9; cc++) run[cc] = 0;                        \
                }                                                                                        \
            }                                                                                            \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
            __syncthreads();                                                                             \
        }
        for (int c = 0; c < n_chunks; c++) {
            HPGV_EPI3B_CHUNK(lds_a, lds_b)
            if (++c >= n_chunks) break;
            HPGV_EPI3B_CHUNK(lds_b, lds_a)
        }
#undef HPGV_EPI3B_CHUNK
        // ---- the nine cells' verdicts, fold by fold (as k_epi_pairs' ranking mode: packed differences and sums) ----
        uint32_t totp[9];
        #pragma unroll
        for (int c = 0; c < 9; c++) {
            totp[c] = 0;
            #pragma unroll
            for (int f = 0; f < K; f++) totp[c] += packed[f][c];
        }
        #pragma unroll
        for (int f = 0; f < K; f++) {
            #pragma unroll
            for (int c = 0; c < 9; c++) {
                const uint32_t in = packed[f][c], tr = totp[c] - in;
                bool high;
                if constexpr (BALANCED) high = __builtin_amdgcn_alignbit(tr, tr, 16) >= (TRAINING ? tr : (tr > 1u ? tr : 1u));   // as k_epi_pairs
                else high = mdr_high_risk<false>((int)(tr & 0xFFFFu), (int)(tr >> 16), ratio, f_na, f_nu);
                sel[f] += high ? (TRAINING ? tr : in) : 0u;
                // (balanced, training part: an EMPTY cell passes the packed comparison and adds nothing to the sums; it is not high risk)
                mask[f] |= (high && tr != 0u) ? (1u << c) << (9 * a) : 0u;
            }
        }
    }

    if (!(j > i && k > j && j < n_variants && k < n_variants)) return;
    #pragma unroll
    for (int f = 0; f < K; f++) {
        const EpiFold fo = folds[f];
        if (fo.test_a < 0) continue;                                 // fold beyond the run's num_folds
        const int tp = (int)(sel[f] & 0xFFFFu), fp = (int)(sel[f] >> 16);
        const int size_a = TRAINING ? n_affected - fo.test_a : fo.test_a, size_u = TRAINING ? n_unaffected - fo.test_u : fo.test_u;
        const double TP = (double)tp, TN = (double)(size_u - fp), ya_ = (double)size_a, yu_ = (double)size_u;
        double qa = TP * fo.inv_a, qu = TN * fo.inv_u;               // the two quotients as in k_epi_pairs (Markstein)
        qa = __builtin_fma(__builtin_fma(-qa, ya_, TP), fo.inv_a, qa);
        qu = __builtin_fma(__builtin_fma(-qu, yu_, TN), fo.inv_u, qu);
        const double acc = (qa + qu) / 2;
        if (acc >= thr[f]) {
            const unsigned slot = atomicAdd(&cand_count[f], 1u);
            if (slot < cand_cap) {
                EpiCand3 e;
                e.accuracy = acc; e.i = i; e.j = j; e.k = k; e.risky = mask[f];
                cand[(size_t)f * cand_cap + slot] = e;
            }
        }
    }
}

}  // namespace hpgv
