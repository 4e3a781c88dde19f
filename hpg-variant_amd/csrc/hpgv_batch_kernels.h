// hpgv_batch_kernels.h -- the per-batch call (assoc_test / tdt_test: assoc_runner.c:192-195,
// tdt_runner.c:184-185) as ONE kernel.
//
// The reference's batch is small (batch_lines = 200 variants, hpg-variant.conf:33), so the call is bound by
// latency and by the bus, not by HBM.  One workgroup per variant:
//   1. the RAW row (HPGV8, VCF column order, as the host staged it) is pulled into LDS with aligned 16-byte loads
//      -- straight out of the caller's page-locked host buffer when it has one (the loads then ARE the PCIe
//      transfer: no separate copy, no device staging buffer), or out of the slot's device copy otherwise;
//   2. every thread builds the 16-byte chunks of the tool's row layout it owns in registers -- the same column
//      gather + strict rule + class recoding as k_layout (hpgv_kernels.h), reading LDS instead of HBM -- and
//      counts them with the same functions the big scans use;
//   3. the workgroup's sums are reduced and the statistics (chi-square / Fisher / TDT) are computed right there;
//      one packed record per variant goes to the slot's page-locked result block.
// So a batch costs one launch, the bus time of its genotypes, and one synchronisation.
#pragma once
#include "hpgv_kernels.h"
#include "hpgv_tdt_stats_kernels.h"

namespace hpgv {

// packed result records (one per variant, written by one lane)
struct BatchAssocRec { int32_t A1, A2, U1, U2; double odds, chisq, p; };          // 40 B; Fisher leaves chisq unset
struct BatchTdtRec { int32_t t1, t2; double odds, chisq, p; };                    // 32 B
struct BatchStatsRec { int32_t c8[8]; double hwe_chi2, hwe_p; };                  // 48 B

enum { BATCH_CHISQ = 0, BATCH_FISHER = 1, BATCH_TDT = 2, BATCH_STATS = 3 };

struct BatchArgs {
    const uint8_t *src;          // raw rows (device-visible: device memory or mapped page-locked host memory)
    size_t src_pitch;
    int n_variants, n_samples;
    const uint8_t *is_x;         // per variant, or null
    const int32_t *col_of_pos;   // the layout's column permutation (device)
    int chunks;                  // 16-byte chunks of the laid-out row
    int chunksA;                 // assoc: chunks of the affected segment
    const double *lf;            // Fisher: ln(i!) table
    double rel_cut;
    // tdt
    int pchunks, p16, n_slow, slow_base;
    TdtLuts luts;
    const uint8_t *male_plane;
    const int32_t *slow_off;
    const uint8_t *slow_male;
    void *out;                   // packed records
};

// one laid-out chunk out of the raw row in LDS: dst byte i = recode(strict(raw[col_of_pos[16 c + i]]))
__device__ __forceinline__ uint4 batch_gather_chunk(const uint8_t *raw /* LDS, byte j of the row at raw[j] */,
                                                    const int32_t *__restrict__ col_of_pos, int c, bool strict, int mode, int p16) {
    const int4 *cp = reinterpret_cast<const int4 *>(col_of_pos + (size_t)c * 16);
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int4 cols = cp[k];
        const int col[4] = {cols.x, cols.y, cols.z, cols.w};
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t g = (col[j] < 0) ? 0xFFu : (uint32_t)raw[col[j]];
            if (strict && (((g & 0xF) == 0xF) || ((g >> 4) == 0xF))) g = 0xFFu;
            g = recode_byte(g, mode, p16, c * 16 + k * 4 + j, col[j] < 0);
            acc |= g << (8 * j);
        }
        w[k] = acc;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// workgroup-wide integer sum of up to 8 per-thread values: every thread gets every total
template <int K>
__device__ __forceinline__ void block_sum(int (&v)[K], int *red /* LDS: 4 x K ints */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = wave_sum(v[k]);
        if (lane == 0) red[wv * K + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = red[k] + red[K + k] + red[2 * K + k] + red[3 * K + k];
    __syncthreads();
}

// the eight flag counts of a row -> {n_00 n_01 n_10 n_11 missing_genotypes missing_alleles allele0 allele1} and the
// Hardy-Weinberg test on (n_00, n_01 + n_10, n_11): the record k_stats_scan + k_stats_hwe produce
__device__ __forceinline__ BatchStatsRec stats_record(const int (&cnt)[8]) {
    BatchStatsRec r;
    r.c8[0] = cnt[0]; r.c8[1] = cnt[1]; r.c8[2] = cnt[2]; r.c8[3] = cnt[3];
    r.c8[4] = cnt[4]; r.c8[5] = cnt[4] + cnt[5];
    r.c8[6] = 2 * cnt[0] + cnt[1] + cnt[2] + cnt[6]; r.c8[7] = 2 * cnt[3] + cnt[1] + cnt[2] + cnt[7];
    const int n_AA = cnt[0], n_Aa = cnt[1] + cnt[2], n_aa = cnt[3], tot = n_AA + n_Aa + n_aa;
    if (tot == 0) { r.hwe_chi2 = __builtin_nan(""); r.hwe_p = __builtin_nan(""); }
    else {                                                     // the body of k_stats_hwe
        const double pf = (2.0 * n_AA + n_Aa) / (2.0 * tot);
        const double qf = 1.0 - pf;
        const double e_AA = pf * pf * tot, e_Aa = 2.0 * pf * qf * tot, e_aa = qf * qf * tot;
        double x = 0.0;
        if (e_AA > 0.0) x += ((n_AA - e_AA) * (n_AA - e_AA)) / e_AA;
        if (e_Aa > 0.0) x += ((n_Aa - e_Aa) * (n_Aa - e_Aa)) / e_Aa;
        if (e_aa > 0.0) x += ((n_aa - e_aa) * (n_aa - e_aa)) / e_aa;
        r.hwe_chi2 = x; r.hwe_p = chisq_p_value(x);
    }
    return r;
}

// the raw row of variant v into LDS (aligned 16-byte loads; an aligned granule never crosses a page, so the bytes before
// the row's start / after its end that come along are harmless); returns the row's offset inside the window
__device__ __forceinline__ int batch_stage_row(const uint8_t *src, size_t src_pitch, int v, int n_samples, uint8_t *lds_raw) {
    const int tid = threadIdx.x;
    const uintptr_t a = (uintptr_t)(src + (size_t)v * src_pitch);
    const uintptr_t a0 = a & ~(uintptr_t)15;
    const int shift = (int)(a - a0);
    const int n16 = (shift + n_samples + 15) >> 4;
    for (int c0 = 0; c0 < n16; c0 += 4 * 256) {                   // four loads in flight per thread: one bus round trip per 16 KiB
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 256 + tid;
            if (c < n16) q[u] = load16<false>(reinterpret_cast<const uint4 *>(a0) + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 256 + tid;
            if (c < n16) reinterpret_cast<uint4 *>(lds_raw)[c] = q[u];
        }
    }
    return shift;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_batch(BatchArgs A) {
    extern __shared__ __align__(16) uint8_t lds_raw[];          // the aligned window of the raw row
    __shared__ int red[4 * 8];
    __shared__ double exp_tab[64];
    const int v = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    if (KIND == BATCH_FISHER && tid < 64) exp_tab[tid] = k_exp2_j64[tid];

    // ---- 1. raw row -> LDS ---------------------------------------------------------------------------------------
    const int shift = batch_stage_row(A.src, A.src_pitch, v, A.n_samples, lds_raw);
    const bool x_row = (A.is_x != nullptr) && (A.is_x[v] != 0);
    __syncthreads();
    const uint8_t *raw = lds_raw + shift;

    if constexpr (KIND == BATCH_CHISQ || KIND == BATCH_FISHER) {
        // ---- 2. assoc layout chunks [affected | pad16 | unaffected | pad], counted as in k_assoc_scan ------------
        AssocAcc acc = {0u, 0u, 0u};
        for (int c = tid; c < A.chunks; c += 256) {
            uint4 q[1] = {batch_gather_chunk(raw, A.col_of_pos, c, true, RECODE_NONE, 0)};
            if (x_row) acc = assoc_count_tile<1, true>(q, c - lane, lane, A.chunksA, acc);
            else       acc = assoc_count_tile<1, false>(q, c - lane, lane, A.chunksA, acc);
        }
        int s[6] = {(int)(acc.pnz & 0xFFFFu), (int)(acc.pnz >> 16), (int)(acc.pnnf & 0xFFFFu), (int)(acc.pnnf >> 16),
                    (int)(acc.pbnz & 0xFFFFu), (int)(acc.pbnz >> 16)};
        block_sum<6>(s, red);
        const int TA = 32 * A.chunksA, TU = 32 * (A.chunks - A.chunksA);
        int A1, A2, U1, U2;
        if (!x_row) {                                          // assoc.c:108-125
            A1 = TA - s[0]; A2 = s[0] + s[2] - TA; U1 = TU - s[1]; U2 = s[1] + s[3] - TU;
        } else {                                               // assoc.c:94-107
            const int validA = s[2] >> 1, validU = s[3] >> 1;
            const int xxA = s[4] - (TA / 2 - validA), xxU = s[5] - (TU / 2 - validU);
            A1 = (TA - s[0]) - validA + xxA; A2 = xxA; U1 = (TU - s[1]) - validU + xxU; U2 = xxU;
        }
        // ---- 3. statistics + record -----------------------------------------------------------------------------
        BatchAssocRec *out = reinterpret_cast<BatchAssocRec *>(A.out) + v;
        if constexpr (KIND == BATCH_CHISQ) {
            if (tid == 0) {
                const double x = assoc_chisq_value(A1, A2, U1, U2);
                BatchAssocRec r = {A1, A2, U1, U2, assoc_odds(A1, A2, U1, U2), x, chisq_p_value(x)};
                *out = r;
            }
        } else {
            if (tid < 64) {                                    // wave 0 (exp_tab was filled before the first barrier)
                const double p = fisher_wave(A1, A2, U1, U2, A.lf, exp_tab, A.rel_cut, lane);
                if (tid == 0) { BatchAssocRec r = {A1, A2, U1, U2, assoc_odds(A1, A2, U1, U2), 0.0, p}; *out = r; }
            }
        }
    } else if constexpr (KIND == BATCH_TDT) {
        // ---- 2. [father | mother | child] class planes of the single-child families, 4 trios per dword -----------
        int t[2] = {0, 0};
        for (int c = tid; c < A.pchunks; c += 256) {
            const uint4 qf = batch_gather_chunk(raw, A.col_of_pos, c, true, RECODE_TDT, A.p16);
            const uint4 qm = batch_gather_chunk(raw, A.col_of_pos, c + A.pchunks, true, RECODE_TDT, A.p16);
            const uint4 qc = batch_gather_chunk(raw, A.col_of_pos, c + 2 * A.pchunks, true, RECODE_TDT, A.p16);
            if (!x_row) {
                { const int2 d = tdt4<false>(A.luts, qf.x, qm.x, qc.x, 0); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<false>(A.luts, qf.y, qm.y, qc.y, 0); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<false>(A.luts, qf.z, qm.z, qc.z, 0); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<false>(A.luts, qf.w, qm.w, qc.w, 0); t[0] += d.x; t[1] += d.y; }
            } else {
                const uint4 ml = reinterpret_cast<const uint4 *>(A.male_plane)[c];
                { const int2 d = tdt4<true>(A.luts, qf.x, qm.x, qc.x, ml.x); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<true>(A.luts, qf.y, qm.y, qc.y, ml.y); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<true>(A.luts, qf.z, qm.z, qc.z, ml.z); t[0] += d.x; t[1] += d.y; }
                { const int2 d = tdt4<true>(A.luts, qf.w, qm.w, qc.w, ml.w); t[0] += d.x; t[1] += d.y; }
            }
        }
        // families with several counted children: the scalar rule at family scope (tdt.c:128-132), one family per thread
        for (int k = tid; k < A.n_slow; k += 256) {
            const int off = A.slow_off[k], n_children = A.slow_off[k + 1] - off - 2;
            const int32_t *cols = A.col_of_pos + A.slow_base + off;
            auto code = [&](int i) -> uint32_t {
                uint32_t g = (uint32_t)raw[cols[i]];
                if (((g & 0xF) == 0xF) || ((g >> 4) == 0xF)) g = 0xFFu;
                return g;
            };
            const uint32_t fb = code(0), mb = code(1);
            const int f1 = fb >> 4, f2 = fb & 0xF, m1 = mb >> 4, m2 = mb & 0xF;
            if (!tdt_parents_usable(f1, f2, m1, m2)) continue;
            TdtState st = {0, 0, 0, 0};
            for (int j = 0; j < n_children; ++j) {
                const uint32_t cb = code(2 + j);
                st = tdt_child(f1, f2, m1, m2, (int)(cb >> 4), (int)(cb & 0xF), x_row && A.slow_male[off + 2 + j], st);
            }
            t[0] += st.t1; t[1] += st.t2;
        }
        block_sum<2>(t, red);
        if (tid == 0) {                                        // tdt.c:255-260, 288-292
            const int t1 = t[0], t2 = t[1];
            double x = -1;
            if (t1 + t2 > 0) x = ((double)((t1 - t2) * (t1 - t2))) / (t1 + t2);
            const double d1 = t1, d2 = t2;
            BatchTdtRec r = {t1, t2, (d2 == 0.0) ? __builtin_nan("") : (d1 / d2), x, chisq_p_value(x)};
            reinterpret_cast<BatchTdtRec *>(A.out)[v] = r;
        }
    } else {
        // ---- stats: VCF column order, one-hot flag bytes, eight masked popcounts per dword ------------------------
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = tid; c < A.chunks; c += 256) {
            const uint4 q = batch_gather_chunk(raw, A.col_of_pos, c, false, RECODE_STATS, 0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 8; ++b) cnt[b] += __builtin_popcount(w[k] & (0x01010101u << b));
        }
        block_sum<8>(cnt, red);
        if (tid == 0) reinterpret_cast<BatchStatsRec *>(A.out)[v] = stats_record(cnt);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Everything the stats tool wants of a batch (get_variants_stats + get_sample_stats, stats_runner.c:194-198) in ONE pass
// over the raw matrix the tokenizer wrote: per variant the counters + Hardy-Weinberg record, the per-sample missing
// counts, the Mendelian errors per variant and per child, and the counters of every phenotype group.  A workgroup owns a
// band of consecutive variants; each row is read from HBM once into LDS and every statistic is computed from that copy:
//   A. all columns in VCF order, four genotypes per instruction: every byte becomes its PAIR CLASS idx = 4 c1 + c2 with
//      c = class of an allele nibble (0, 1, other, missing) through 16-entry v_perm_b32 tables, written back over the raw
//      byte; the one-hot stats flags are a 16-entry table of idx; eight masked popcounts per dword; the "some allele
//      missing" bits are added to per-column byte counters kept in LDS across the band;
//   B. the trios' [father | mother | child] planes gather idx bytes by column, a 16-entry table gives the zero-ness class
//      check_mendel looks at, the error table of k_mendel_scan the error bit per trio; per-trio byte counters in LDS;
//   C. every phenotype group gathers idx bytes in its own column order and counts the flags.
// The per-sample / per-child counters are column sums over variants: at the band's end the LDS counters go to the device
// totals by atomics -- coalesced full-wave atomics over ALL columns when the band saw many events (a dense wave atomic
// costs what one scattered lane costs), only the non-zero counters when it saw few.
// ---------------------------------------------------------------------------------------------------------------------
struct StatsAllArgs {
    const uint8_t *src; size_t src_pitch; int n_variants, n_samples;
    int rows_per_block;              // band length (<= 255: byte counters)
    int lds_row;                     // bytes of the row window in LDS (multiple of 16)
    const uint8_t *is_x;
    BatchStatsRec *out;              // [n_variants]
    int32_t *sample_missing;         // [n_samples] device, accumulated into; or null
    // Mendelian errors over the trios of hpgv_set_pedigree: planes [father | mother | child] of col_of_pos
    const int32_t *mendel_cols; int pchunks, n_trios; MendelLuts luts; const uint8_t *male_plane;
    int32_t *mendel_errors;          // [n_variants]; or null
    int32_t *child_errors;           // [n_trios] device, accumulated into; or null
    // phenotype groups of hpgv_set_stats_groups: the grouped layout's permutation, first chunk and chunk count per group
    const int32_t *group_cols; int n_groups; const int32_t *group_chunk0; const int32_t *group_chunks;
    BatchStatsRec *group_out;        // [g * n_variants + v]; or null
};

// 16-entry byte table look-up on four selectors (0..15) at once: two 8-entry v_perm_b32 + one v_perm_b32 to pick by bit 3
__device__ __forceinline__ uint32_t lut16x4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t sel) {
    const uint32_t s7 = sel & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(t1, t0, s7), hi = __builtin_amdgcn_perm(t3, t2, s7);
    return __builtin_amdgcn_perm(hi, lo, ((sel >> 1) & 0x04040404u) | 0x03020100u);
}
// class of an allele nibble: 0 -> 0, 1 -> 1, 2..14 -> 2, 15 (missing) -> 3
__device__ __forceinline__ uint32_t nib_class4(uint32_t n) { return lut16x4(0x02020100u, 0x02020202u, 0x02020202u, 0x03020202u, n); }
// four HPGV8 bytes -> four pair classes idx = 4 * class(allele1) + class(allele2)
__device__ __forceinline__ uint32_t pair_class4(uint32_t g) {
    return (nib_class4((g >> 4) & 0x0F0F0F0Fu) << 2) | nib_class4(g & 0x0F0F0F0Fu);
}
// stats_flags (hpgv_kernels.h) as a table of the pair class: rows c1 = 0, 1, other, missing; columns c2 likewise
__device__ __forceinline__ uint32_t flags_of_class4(uint32_t idx) { return lut16x4(0x50400201u, 0x90800804u, 0x10008040u, 0x30109050u, idx); }
// mendel_class (hpgv_kernels.h) as a table of the pair class: 0 "0/0", 1 one zero allele, 2 no zero allele, 3 not fully called
__device__ __forceinline__ uint32_t mendel_of_class4(uint32_t idx) { return lut16x4(0x03010100u, 0x03020201u, 0x03020201u, 0x03030303u, idx); }
constexpr uint32_t PAIR_CLASS_NOFLAGS = 10u;   // (other, other): no stats flag -- the pad of a group's segment
constexpr uint32_t PAIR_CLASS_MISSING = 15u;   // (missing, missing): the pad of a trio plane

// 16 pair-class bytes gathered by column from the row's LDS copy
__device__ __forceinline__ uint4 gather_class_chunk(const uint8_t *cls /* LDS */, const int32_t *__restrict__ cols, int c, uint32_t pad) {
    const int4 *cp = reinterpret_cast<const int4 *>(cols + (size_t)c * 16);
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int4 q = cp[k];
        const uint32_t b0 = q.x < 0 ? pad : (uint32_t)cls[q.x], b1 = q.y < 0 ? pad : (uint32_t)cls[q.y];
        const uint32_t b2 = q.z < 0 ? pad : (uint32_t)cls[q.z], b3 = q.w < 0 ? pad : (uint32_t)cls[q.w];
        w[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void stats_count_flags(const uint4 q, int (&cnt)[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 8; ++b) cnt[b] += __builtin_popcount(w[k] & (0x01010101u << b));
}

static __global__ __launch_bounds__(256) void k_stats_all(StatsAllArgs A) {
    extern __shared__ __align__(16) uint8_t lds[];
    __shared__ int red[4 * 8];
    const int tid = threadIdx.x;
    const int chunks = (A.n_samples + 15) >> 4;
    uint8_t *lds_raw = lds;
    uint32_t *miss_cnt = reinterpret_cast<uint32_t *>(lds + A.lds_row);           // [chunks * 4]: byte counters per column
    uint32_t *trio_cnt = miss_cnt + (size_t)chunks * 4;                            // [pchunks * 4]: byte counters per trio
    const bool want_sm = A.sample_missing != nullptr, want_ce = A.child_errors != nullptr;
    const bool want_me = A.mendel_errors != nullptr || want_ce;
    if (want_sm) for (int i = tid; i < chunks * 4; i += 256) miss_cnt[i] = 0u;
    if (want_ce) for (int i = tid; i < A.pchunks * 4; i += 256) trio_cnt[i] = 0u;
    const int v0 = blockIdx.x * A.rows_per_block;
    const int v1 = v0 + A.rows_per_block < A.n_variants ? v0 + A.rows_per_block : A.n_variants;
    int events[2] = {0, 0};                                                        // missing genotypes / child errors of the band

    for (int v = v0; v < v1; ++v) {
        __syncthreads();                                                           // the previous row's readers are done
        const int shift = batch_stage_row(A.src, A.src_pitch, v, A.n_samples, lds_raw);
        const bool x_row = (A.is_x != nullptr) && (A.is_x[v] != 0);
        __syncthreads();
        uint8_t *row = lds_raw + shift;                                            // raw bytes now, pair classes after pass A

        // ---- A. all columns in VCF order ---------------------------------------------------------------------------------
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = tid; c < chunks; c += 256) {
            uint32_t g[4];
            if (shift == 0) {                                                      // rows of a 16-byte pitch (the tokenizer's): vector reads
                const uint4 q = reinterpret_cast<const uint4 *>(row)[c];
                g[0] = q.x; g[1] = q.y; g[2] = q.z; g[3] = q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint8_t *b = row + c * 16 + k * 4;
                    g[k] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
                }
            }
            uint32_t idx[4], f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                idx[k] = pair_class4(g[k]);
                const int left = A.n_samples - (c * 16 + k * 4);                   // columns of this dword that exist
                const uint32_t valid = left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
                f[k] = flags_of_class4(idx[k]) & valid;
            }
            if (shift == 0) reinterpret_cast<uint4 *>(row)[c] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) row[c * 16 + k * 4 + j] = (uint8_t)(idx[k] >> (8 * j));
            }
            stats_count_flags(make_uint4(f[0], f[1], f[2], f[3]), cnt);
            if (want_sm) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = (f[k] >> 4) & 0x01010101u;                  // flag bit 4: some allele missing
                    miss_cnt[c * 4 + k] += m;                                      // this thread's own columns: no atomics
                    events[0] += __builtin_popcount(m);
                }
            }
        }
        block_sum<8>(cnt, red);                                                    // (its barriers also publish the pair classes)
        if (tid == 0) A.out[v] = stats_record(cnt);

        // ---- B. Mendelian errors: 4 trios per dword ---------------------------------------------------------------------
        if (want_me) {
            int n[1] = {0};
            for (int c = tid; c < A.pchunks; c += 256) {
                const uint4 qf = gather_class_chunk(row, A.mendel_cols, c, PAIR_CLASS_MISSING);
                const uint4 qm = gather_class_chunk(row, A.mendel_cols, c + A.pchunks, PAIR_CLASS_MISSING);
                const uint4 qc = gather_class_chunk(row, A.mendel_cols, c + 2 * A.pchunks, PAIR_CLASS_MISSING);
                const uint32_t ff[4] = {mendel_of_class4(qf.x), mendel_of_class4(qf.y), mendel_of_class4(qf.z), mendel_of_class4(qf.w)};
                const uint32_t mm[4] = {mendel_of_class4(qm.x), mendel_of_class4(qm.y), mendel_of_class4(qm.z), mendel_of_class4(qm.w)};
                const uint32_t cc[4] = {mendel_of_class4(qc.x), mendel_of_class4(qc.y), mendel_of_class4(qc.z), mendel_of_class4(qc.w)};
                uint4 ml = make_uint4(0, 0, 0, 0);
                if (x_row) ml = reinterpret_cast<const uint4 *>(A.male_plane)[c];
                const uint32_t mlw[4] = {ml.x, ml.y, ml.z, ml.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t e = x_row ? mendel4<true>(A.luts, ff[k], mm[k], cc[k], mlw[k]) : mendel4<false>(A.luts, ff[k], mm[k], cc[k], 0);
                    n[0] += __builtin_popcount(e);
                    if (want_ce) trio_cnt[c * 4 + k] += e;
                }
            }
            events[1] += n[0];
            block_sum<1>(n, red);
            if (tid == 0 && A.mendel_errors) A.mendel_errors[v] = n[0];
        }

        // ---- C. per phenotype group: the grouped layout's chunks, one group after the other ------------------------------
        if (A.group_out) {
            for (int gk = 0; gk < A.n_groups; ++gk) {
                int gc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const int c0 = A.group_chunk0[gk], nc = A.group_chunks[gk];
                for (int c = tid; c < nc; c += 256) {
                    const uint4 q = gather_class_chunk(row, A.group_cols, c0 + c, PAIR_CLASS_NOFLAGS);
                    stats_count_flags(make_uint4(flags_of_class4(q.x), flags_of_class4(q.y), flags_of_class4(q.z), flags_of_class4(q.w)), gc);
                }
                block_sum<8>(gc, red);
                if (tid == 0) A.group_out[(size_t)gk * (size_t)A.n_variants + (size_t)v] = stats_record(gc);
            }
        }
    }

    // ---- the band's column counters -> device totals -------------------------------------------------------------------------
    if (want_sm || want_ce) {
        block_sum<2>(events, red);                                                 // (barrier: every thread's counters are written)
        if (want_sm && events[0] > 0) {
            const bool dense = events[0] * 64 >= A.n_samples;                      // as many events as a dense sweep has wave instructions
            const uint8_t *cb = reinterpret_cast<const uint8_t *>(miss_cnt);
            for (int d = tid; d < A.n_samples; d += 256) {
                const int val = cb[d];
                if (dense || val) atomicAdd(A.sample_missing + d, val);
            }
        }
        if (want_ce && events[1] > 0) {
            const bool dense = events[1] * 64 >= A.n_trios;
            const uint8_t *cb = reinterpret_cast<const uint8_t *>(trio_cnt);
            for (int d = tid; d < A.n_trios; d += 256) {
                const int val = cb[d];
                if (dense || val) atomicAdd(A.child_errors + d, val);
            }
        }
    }
}

}  // namespace hpgv
