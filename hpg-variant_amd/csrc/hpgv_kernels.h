// hpgv_kernels.h -- CDNA4 (gfx950) kernels of the per-variant statistics path.
//
// All scans are HBM-read bound integer/byte work: one variant row per
// wavefront, 16 B per lane per load (1 KiB per wave instruction), several loads
// in flight, SWAR nibble counting in VGPRs, DPP + readlane wave reduction, one
// small store per variant.  No LDS, no MFMA: there is no reuse and no dense
// contraction on this path (DESIGN.md "Kernels").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

// ---------------------------------------------------------------------------
// SWAR helpers on 4 packed genotype bytes (8 nibbles)
// ---------------------------------------------------------------------------
// bit 3 of every nibble set <=> that nibble is non-zero (2 x v_lshl_or_b32 + and)
__device__ __forceinline__ uint32_t nib_nonzero(uint32_t x) {
    uint32_t y = x | (x << 1);
    y = y | (y << 2);
    return y & 0x88888888u;
}
// bit 3 of every nibble set <=> that nibble is not 0xF
__device__ __forceinline__ uint32_t nib_not_f(uint32_t x) { return nib_nonzero(~x); }

// ---------------------------------------------------------------------------
// wave-wide integer sum -> wave-uniform value (SGPR).  4 DPP adds reduce each
// row of 16 lanes, 4 readlanes + scalar adds combine the 4 rows.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// 16-byte load from a wave-uniform base + 32-bit byte offset: this form lets the compiler keep
// the base in SGPRs (global_load ... saddr) instead of a 64-bit VGPR address per load
template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4 *p);
template <bool NT>
__device__ __forceinline__ uint4 load16o(const uint8_t *__restrict__ base, uint32_t byte_off) {
    return load16<NT>(reinterpret_cast<const uint4 *>(base + byte_off));
}
template <bool NT>
__device__ __forceinline__ uint4 load16(const uint4 *p) {
    if constexpr (NT) {
        // streamed once: keep it out of the way of the small cached vectors
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}

// ---------------------------------------------------------------------------
// assoc scan.  Row layout: [affected | pad16 | unaffected | pad .. pitch), pad
// bytes = 0xFF.  chunks = pitch/16 counted chunks, chunksA = affected chunks.
//
// Per 16-byte chunk two counts are taken:
//   NZ  = nibbles != 0        (0xFF bytes give 2)
//   NNF = nibbles != 0xF      (0xFF bytes give 0; strict rows have no half-missing)
// and, for chromosome "X" rows only, BNZ = bytes whose two nibbles are both != 0.
// With T = 2 * (bytes looked at, real or virtual 0xFF):
//   autosome (assoc.c:108-125): allele1 count = zero nibbles of valid bytes = T - NZ
//                               allele2 count = NZ + NNF - T
//   chr X    (assoc.c:94-107) : n_valid = NNF/2, n_missing = T/2 - n_valid,
//                               n_xx = BNZ - n_missing, n_00 = (T - NZ) - n_valid + n_xx
// Per-lane partial sums of the two phenotype groups share one register
// (affected in bits 0..15, unaffected in bits 16..31).
// ---------------------------------------------------------------------------
// one row, all lanes of the wave; X selects the chromosome-"X" extra count
template <bool NT, int U, bool X>
__device__ __forceinline__ void assoc_row(const uint8_t *__restrict__ row, int lane, int chunksA,
                                          int chunks, uint32_t &pnz, uint32_t &pnnf, uint32_t &pbnz) {
    for (int base = 0; base < chunks; base += 64 * U) {
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = base + u * 64 + lane;
            q[u] = make_uint4(~0u, ~0u, ~0u, ~0u);     // virtual chunk = all missing
            if (c < chunks) q[u] = load16o<NT>(row, (uint32_t)c * 16u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = base + u * 64 + lane;
            const uint32_t sh = (c < chunksA) ? 0u : 16u;
            const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
            uint32_t nz = 0, nnf = 0, bnz = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ind = nib_nonzero(w[k]);
                nz += __builtin_popcount(ind);
                nnf += __builtin_popcount(nib_not_f(w[k]));
                if constexpr (X) bnz += __builtin_popcount(ind & (ind >> 4) & 0x08080808u);
            }
            pnz += nz << sh;
            pnnf += nnf << sh;
            if constexpr (X) pbnz += bnz << sh;
        }
    }
}

// reduce one row's packed partial sums and store {A1, A2, U1, U2}
__device__ __forceinline__ void assoc_row_finish(bool x_row, int lane, int TA, int TU, uint32_t pnz,
                                                 uint32_t pnnf, uint32_t pbnz, int4 *__restrict__ dst) {
    const int nzA = wave_sum((int)(pnz & 0xFFFFu)), nzU = wave_sum((int)(pnz >> 16));
    const int nfA = wave_sum((int)(pnnf & 0xFFFFu)), nfU = wave_sum((int)(pnnf >> 16));
    int4 out;
    if (!x_row) {
        out.x = TA - nzA;            // A1
        out.y = nzA + nfA - TA;      // A2
        out.z = TU - nzU;            // U1
        out.w = nzU + nfU - TU;      // U2
    } else {
        const int bA = wave_sum((int)(pbnz & 0xFFFFu)), bU = wave_sum((int)(pbnz >> 16));
        const int validA = nfA >> 1, validU = nfU >> 1;
        const int xxA = bA - (TA / 2 - validA), xxU = bU - (TU / 2 - validU);
        out.x = (TA - nzA) - validA + xxA;
        out.y = xxA;
        out.z = (TU - nzU) - validU + xxU;
        out.w = xxU;
    }
    if (lane == 0) *dst = out;
}

// STRIDED = false: wave w owns the vpw consecutive rows [w*vpw, (w+1)*vpw) and the
//                  grid covers all rows (the hardware dispatcher balances the load);
// STRIDED = true : persistent grid, wave w takes rows w, w+W, w+2W, ... (W = waves in
//                  the grid), so at any moment the chip streams one contiguous window.
template <bool NT, int U, bool STRIDED>
__global__ __launch_bounds__(256) void k_assoc_scan(const uint8_t *__restrict__ gt, size_t pitch,
                                                    int n_variants, int chunksA, int chunks,
                                                    const uint8_t *__restrict__ is_x,
                                                    int4 *__restrict__ counts, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // every wave looks at `slots` chunk slots per row; slots past `chunks` are
    // virtual all-0xFF chunks booked on the unaffected group
    const int slots = ((chunks + 64 * U - 1) / (64 * U)) * (64 * U);
    const int TA = 32 * chunksA;
    const int TU = 32 * (slots - chunksA);
    long v, v_end, v_step;
    if constexpr (STRIDED) { v = wave; v_end = n_variants; v_step = (long)gridDim.x * (blockDim.x >> 6); }
    else { v = wave * vpw; v_end = v + vpw < n_variants ? v + vpw : n_variants; v_step = 1; }

    for (; v < v_end; v += v_step) {                       // wave-uniform
        const uint8_t *row = gt + (size_t)v * pitch;
        const bool x_row = (is_x != nullptr) && (__builtin_amdgcn_readfirstlane((int)is_x[v]) != 0);
        uint32_t pnz = 0, pnnf = 0, pbnz = 0;
        if (x_row) assoc_row<NT, U, true>(row, lane, chunksA, chunks, pnz, pnnf, pbnz);
        else       assoc_row<NT, U, false>(row, lane, chunksA, chunks, pnz, pnnf, pbnz);
        assoc_row_finish(x_row, lane, TA, TU, pnz, pnnf, pbnz, counts + v);
    }
}

// Software-pipelined variant: the wave's rows are cut into tiles of 64*U chunks and
// the loads of tile t+1 (which may belong to the NEXT row) are issued before tile t
// is counted and, at a row end, reduced and stored.  Two register sets.
struct AssocAcc { uint32_t pnz, pnnf, pbnz; };

template <int U, bool X>
__device__ __forceinline__ AssocAcc assoc_count_tile(const uint4 (&q)[U], int base, int lane, int chunksA,
                                                     AssocAcc a) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = base + u * 64 + lane;
        const uint32_t sh = (c < chunksA) ? 0u : 16u;
        const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
        uint32_t nz = 0, nnf = 0, bnz = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ind = nib_nonzero(w[k]);
            nz += __builtin_popcount(ind);
            nnf += __builtin_popcount(nib_not_f(w[k]));
            if constexpr (X) bnz += __builtin_popcount(ind & (ind >> 4) & 0x08080808u);
        }
        a.pnz += nz << sh;
        a.pnnf += nnf << sh;
        if constexpr (X) a.pbnz += bnz << sh;
    }
    return a;
}

template <bool NT, int U>
__device__ __forceinline__ void assoc_issue_tile(uint4 (&q)[U], const uint8_t *__restrict__ g0, size_t pitch,
                                                 int t, int ipr, int chunks, int lane) {
    const int r = t / ipr, base = (t - r * ipr) * 64 * U;
    const uint8_t *row = g0 + (size_t)r * pitch;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = base + u * 64 + lane;
        q[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (c < chunks) q[u] = load16o<NT>(row, (uint32_t)c * 16u);
    }
}

template <int U>
__device__ __forceinline__ AssocAcc assoc_consume_tile(const uint4 (&q)[U], AssocAcc a, int t, int ipr, long v_begin,
                                                       int lane, int chunksA, int TA, int TU,
                                                       const uint8_t *__restrict__ is_x, int4 *__restrict__ counts) {
    const int r = t / ipr, it = t - r * ipr;
    const long v = v_begin + r;
    const bool x_row = (is_x != nullptr) && (__builtin_amdgcn_readfirstlane((int)is_x[v]) != 0);
    if (x_row) a = assoc_count_tile<U, true>(q, it * 64 * U, lane, chunksA, a);
    else       a = assoc_count_tile<U, false>(q, it * 64 * U, lane, chunksA, a);
    if (it == ipr - 1) {
        assoc_row_finish(x_row, lane, TA, TU, a.pnz, a.pnnf, a.pbnz, counts + v);
        a.pnz = a.pnnf = a.pbnz = 0;
    }
    return a;
}

template <bool NT, int U, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_assoc_scan_pipe(const uint8_t *__restrict__ gt, size_t pitch,
                                                         int n_variants, int chunksA, int chunks,
                                                         const uint8_t *__restrict__ is_x,
                                                         int4 *__restrict__ counts, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    if (v_begin >= n_variants) return;
    const int rows = (int)((v_begin + vpw <= n_variants) ? vpw : (n_variants - v_begin));
    const int ipr = (chunks + 64 * U - 1) / (64 * U);       // tiles per row
    const int slots = ipr * 64 * U;
    const int TA = 32 * chunksA, TU = 32 * (slots - chunksA);
    const int n_tiles = rows * ipr;
    const uint8_t *g0 = gt + (size_t)v_begin * pitch;
    AssocAcc acc = {0u, 0u, 0u};
    uint4 qa[U], qb[U];
    assoc_issue_tile<NT, U>(qa, g0, pitch, 0, ipr, chunks, lane);
    for (int t = 0; t < n_tiles; t += 2) {
        if (t + 1 < n_tiles) assoc_issue_tile<NT, U>(qb, g0, pitch, t + 1, ipr, chunks, lane);
        acc = assoc_consume_tile<U>(qa, acc, t, ipr, v_begin, lane, chunksA, TA, TU, is_x, counts);
        if (t + 1 < n_tiles) {
            if (t + 2 < n_tiles) assoc_issue_tile<NT, U>(qa, g0, pitch, t + 2, ipr, chunks, lane);
            acc = assoc_consume_tile<U>(qb, acc, t + 1, ipr, v_begin, lane, chunksA, TA, TU, is_x, counts);
        }
    }
}

// ---------------------------------------------------------------------------
// chi-square statistics from the counts (assoc_basic_test.c:23-41,58-61).
// One thread per variant; same operation order as the reference (the file is
// built with -ffp-contract=off so no FMA is formed).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double chisq_p_value(double x) {
    // 1 - gsl_cdf_chisq_P(x, 1): see oracle/hpgv_oracle.c orc_chisq_p_value
    if (x != x) return x;
    if (x <= 0.0) return 1.0;
    const double y = x / 2.0;
    double P;
    if (y > 0.5) P = 1.0 - erfc(sqrt(y));
    else         P = erf(sqrt(y));
    return 1.0 - P;
}

__device__ __forceinline__ double assoc_odds(int A1, int A2, int U1, int U2) {
    return (A2 == 0 || U1 == 0) ? __builtin_nan("") : ((double)A1 / A2) * ((double)U2 / U1);
}

// (a, c, b, d) = (A1, A2, U1, U2): the argument order of assoc.c:61 is (A1, U1, A2, U2)
__device__ __forceinline__ double assoc_chisq_value(int a, int c, int b, int d) {
    const double total = a + c + b + d;
    const double t_aff = a + c, t_un = b + d, t_1 = a + b, t_2 = c + d;
    const double e_a1 = (t_aff * t_1) / total;
    const double e_a2 = (t_aff * t_2) / total;
    const double e_u1 = (t_un * t_1) / total;
    const double e_u2 = (t_un * t_2) / total;
    return ((a - e_a1) * (a - e_a1)) / e_a1 + ((c - e_a2) * (c - e_a2)) / e_a2 +
           ((b - e_u1) * (b - e_u1)) / e_u1 + ((d - e_u2) * (d - e_u2)) / e_u2;
}

static __global__ __launch_bounds__(256) void k_assoc_chisq(const int4 *__restrict__ counts, int n,
                                                     double *__restrict__ odds,
                                                     double *__restrict__ chisq,
                                                     double *__restrict__ pval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c4 = counts[i];
    const int a = c4.x, c = c4.y, b = c4.z, d = c4.w;   // assoc.c:61: (A1, U1, A2, U2)
    const double x = assoc_chisq_value(a, c, b, d);
    odds[i] = assoc_odds(a, c, b, d);
    chisq[i] = x;
    pval[i] = chisq_p_value(x);
}

// ---------------------------------------------------------------------------
// Fisher's exact test, two-sided (assoc_fisher_test.c:24-26; definition in
// oracle/hpgv_oracle.c orc_fisher_two_sided: sum of P(x) over all admissible
// tables x with P(x) <= P_obs * (1 + 1e-7), P through the log-factorial table).
// One wave per variant.  The hypergeometric pmf is unimodal, so the summed set is
// a left tail [lo, xL] and a right tail [xR, hi].  Short ranges are scanned whole;
// for long ranges the wave finds xL and xR -- one round of two 64-wide windows around
// where they almost always are (next to the observed table; near its mirror image
// about the mode), a 64-ary search over what is left when a window misses -- and
// sums each tail outwards, 64 terms per round, until a whole round is below 1e-22 of
// the tail's largest term (the terms dropped are further out and decay faster than
// geometrically, i.e. far below one ulp of the result).  Every term is the
// definition's exp(konst - lf[..] - lf[..] - lf[..] - lf[..]) on the reference's own
// table: the table's rounding noise (1e-10 relative per term at 100 k alleles) is part
// of the reference's result, and a sum built from ratios of neighbouring terms --
// 2x fewer instructions, measured -- drifts from it by more than the 1e-10 bar.
// The table (a few MB at most) stays L1 / L2 resident (92 % L1 hits measured).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);   // fixed order: deterministic
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

// exp for the Fisher terms (arguments are log-probabilities: <= 0 up to rounding, down to underflow): e = (64 m + j) ln2/64 + r,
// |r| <= ln2/128, exp(e) = 2^m * 2^(j/64) * (1 + r + r^2/2 + ... + r^5/120); the 64 values 2^(j/64) sit in LDS (filled by
// the kernel).  Relative error below 3e-16 (the truncated r^6/720 is 3.5e-17); about 18 instructions against the ~45 of the
// library's full-range exp.  Results below the denormal range flush to zero through v_ldexp_f64.
__device__ const double k_exp2_j64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double fisher_exp(double e, const double *tab /* LDS */) {
    const double kf = __builtin_rint(e * 0x1.71547652b82fep+6);
    double r = __builtin_fma(-kf, 0x1.62e42fe000000p-7, e);
    r = __builtin_fma(-kf, 0x1.f473de6af278fp-36, r);
    const int ki = (int)kf;
    const double tj = tab[ki & 63];
    double q = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = __builtin_fma(r, q, 1.0 / 6.0);
    q = __builtin_fma(r, q, 0.5);
    q = __builtin_fma(r * r, q, r);                                  // exp(r) - 1
    const double y = __builtin_fma(tj, q, tj);
    return e < -1100.0 ? 0.0 : __builtin_ldexp(y, ki >> 6);
}

struct FisherTab {
    const double *lf; const double *xt /* 2^(j/64), LDS */; double konst; int r1, c1, d0;    // d0 = r2 - c1
    // ln P(x): the exponent of the definition's term; comparisons between terms are made on it (exp is monotone),
    // so the boundary searches need no exp at all
    // the four table reads address the table by unsigned 32-bit byte offsets off a scalar base (one VALU op per address)
    __device__ __forceinline__ double at(uint32_t byte_off) const { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lf) + byte_off); }
    __device__ __forceinline__ double e(int x) const {
        const uint32_t x8 = (uint32_t)x << 3;
        return konst - at(x8) - at(((uint32_t)r1 << 3) - x8) - at(((uint32_t)c1 << 3) - x8) - at(((uint32_t)d0 << 3) + x8);
    }
    __device__ __forceinline__ double p(int x) const { return fisher_exp(e(x), xt); }
    // e(x) and e(x + 1) from FOUR 16-byte loads instead of eight 8-byte ones: the pass is bound by its vector memory
    // instructions (48 gathers of 64 x 8 bytes per variant, profiles/r03_pmc_fisher.json), and neighbouring terms read
    // neighbouring table entries -- lf[x], lf[x + 1]; lf[r1 - x - 1], lf[r1 - x]; lf[c1 - x - 1], lf[c1 - x]; lf[d0 + x],
    // lf[d0 + x + 1] (the table is 8-byte aligned: dwordx4 loads need no more).  Same subtractions in the same order as e().
    // At the edge of the support one of the two neighbours may be entry -1 or one past the table: the allocation is padded.
    __device__ __forceinline__ void e2(int x, double *e0, double *e1) const {
        // (offsets are UNSIGNED 32-bit off a scalar base, so they are taken from entry -2, the start of the padding: entry -1 is
        // offset 8, not 2^32 - 8)
        const char *b = reinterpret_cast<const char *>(lf) - 16;
        double2 q0, q1, q2, q3;
        __builtin_memcpy(&q0, b + ((uint32_t)(x + 2) << 3), 16);
        __builtin_memcpy(&q1, b + ((uint32_t)(r1 - x + 1) << 3), 16);
        __builtin_memcpy(&q2, b + ((uint32_t)(c1 - x + 1) << 3), 16);
        __builtin_memcpy(&q3, b + ((uint32_t)(d0 + x + 2) << 3), 16);
        *e0 = konst - q0.x - q1.y - q2.y - q3.x;
        *e1 = konst - q0.y - q1.x - q2.x - q3.y;
    }
};

// ---- sub-waves: W = 64, 32 or 16 consecutive lanes work on one variant (a wave holds 64 / W variants).  Everything the
// lanes of a sub-wave decide together goes through these; values that are uniform within a sub-wave may differ between
// the sub-waves of a wave, and the hardware's lane masking takes care of loops that end at different times.
template <int W> __device__ __forceinline__ int sub_lane(int lane) { return lane & (W - 1); }
template <int W> __device__ __forceinline__ uint64_t sub_ballot(bool pred, int lane) {
    const uint64_t b = __ballot(pred);
    if constexpr (W == 64) { (void)lane; return b; }
    else return (b >> (lane & ~(W - 1))) & ((1ull << W) - 1ull);
}
template <int W> __device__ __forceinline__ double sub_sum_f64(double v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);   // fixed order: deterministic
    return v;
}
template <int W> __device__ __forceinline__ double sub_first_f64(double v, int lane) { return __shfl(v, lane & ~(W - 1)); }

// Boundary of a prefix-true predicate over [L, R) by W-ary search: W probes per round; UP (left of the mode, p
// increasing): pred(x) = p(x) <= thr; right of the mode (p decreasing): pred(x) = p(x) > thr.  Returns the first x where
// the predicate is false (R if none).
// Both boundary searches in lockstep: the two searches are independent, so every round issues the table
// reads and the exp of BOTH probes before either result is needed -- half as many latency-bound rounds.
template <int W>
__device__ __forceinline__ void fisher_boundaries(const FisherTab &T, double thr /* ln */, int L1, int R1, int L2, int R2, int lane,
                                                  int *xL, int *xR) {
    const int sl = sub_lane<W>(lane);
    while (R1 - L1 > W || R2 - L2 > W) {                            // uniform within the sub-wave
        const bool go1 = R1 - L1 > W, go2 = R2 - L2 > W;
        const int step1 = (R1 - L1 + W - 1) / W, step2 = (R2 - L2 + W - 1) / W;
        const int x1 = L1 + sl * step1, x2 = L2 + sl * step2;
        const double v1 = (go1 && x1 < R1) ? T.e(x1) : 0.0;
        const double v2 = (go2 && x2 < R2) ? T.e(x2) : 0.0;
        if (go1) {
            const int j = __builtin_popcountll(sub_ballot<W>(x1 < R1 && v1 <= thr, lane));
            const int xj = L1 + j * step1;
            const int nL = (j == 0) ? L1 : L1 + (j - 1) * step1 + 1, nR = (j == W || xj >= R1) ? R1 : xj;
            L1 = nL; R1 = nR;
        }
        if (go2) {
            const int j = __builtin_popcountll(sub_ballot<W>(x2 < R2 && v2 > thr, lane));
            const int xj = L2 + j * step2;
            const int nL = (j == 0) ? L2 : L2 + (j - 1) * step2 + 1, nR = (j == W || xj >= R2) ? R2 : xj;
            L2 = nL; R2 = nR;
        }
    }
    const int x1 = L1 + sl, x2 = L2 + sl;
    const double v1 = x1 < R1 ? T.e(x1) : 0.0, v2 = x2 < R2 ? T.e(x2) : 0.0;
    *xL = L1 + __builtin_popcountll(sub_ballot<W>(x1 < R1 && v1 <= thr, lane));
    *xR = L2 + __builtin_popcountll(sub_ballot<W>(x2 < R2 && v2 > thr, lane));
}

// Both tails in lockstep (left from xL - 1 downwards, right from xR upwards), W tables per side and turn: the
// table reads and exps of a turn are independent of each other.  Lanes keep private partial sums (one reduction at the
// end); a tail ends after a turn whose W tables are all below rel_cut x its own first (largest) table (the tables
// dropped are further out and decay faster than geometrically).
template <int W>
__device__ __forceinline__ double fisher_tails(const FisherTab &T, int xL, int xR, int lo, int hi, int lane, double rel_cut) {
    // two neighbouring tables per lane and side and turn (FisherTab::e2): 2 W tables per side and turn
    const int sl = sub_lane<W>(lane);
    double partL = 0.0, partR = 0.0, cutL = 0.0, cutR = 0.0;
    bool onL = true, onR = true;
    for (int k = 0; onL || onR; ++k) {                              // uniform within the sub-wave
        const int firstL = xL - 1 - 2 * W * k, firstR = xR + 2 * W * k;
        if (firstL < lo) onL = false;
        if (firstR > hi) onR = false;
        const int a = firstL - 2 * sl, b = firstR + 2 * sl;         // this lane's tables: a, a - 1 on the left; b, b + 1 on the right
        double eL0 = -2000.0, eL1 = -2000.0, eR0 = -2000.0, eR1 = -2000.0;
        if (onL && a >= lo) { double lo_e, hi_e; T.e2(a - 1, &lo_e, &hi_e); eL0 = hi_e; if (a - 1 >= lo) eL1 = lo_e; }      // all the table reads first
        if (onR && b <= hi) { double lo_e, hi_e; T.e2(b, &lo_e, &hi_e); eR0 = lo_e; if (b + 1 <= hi) eR1 = hi_e; }
        const double sL0 = fisher_exp(eL0, T.xt), sL1 = fisher_exp(eL1, T.xt), sR0 = fisher_exp(eR0, T.xt), sR1 = fisher_exp(eR1, T.xt);   // exp(-2000) = 0
        partL += sL0 + sL1; partR += sR0 + sR1;
        // the reference value of a tail: its first table (lane 0 of the first turn), the largest of the tail
        if (k == 0) { cutL = rel_cut * sub_first_f64<W>(sL0, lane); cutR = rel_cut * sub_first_f64<W>(sR0, lane); }
        if (onL && sub_ballot<W>(sL0 > cutL || sL1 > cutL, lane) == 0ull) onL = false;
        if (onR && sub_ballot<W>(sR0 > cutR || sR1 > cutR, lane) == 0ull) onR = false;
    }
    return sub_sum_f64<W>(partL + partR);
}

// The two boundaries are almost always where a guess puts them: on the observed side right after the observed table
// (further only by the 1e-7 slack, i.e. next to the mode), on the other side near the mirror image of the observed table
// about the mode (off by the skew: a few tables).  One round probes a W-wide window around each guess; a window that
// does not bracket its boundary leaves a one-sided range to the W-ary search.  Returns true when both were found.
template <int W>
__device__ __forceinline__ bool fisher_boundary_windows(const FisherTab &T, double thr /* ln */, int mode, int obs, int lane,
                                                        int *L1, int *R1, int *L2, int *R2) {
    // left: pred(x) = e(x) <= thr is prefix-true over [lo, mode + 1); right: pred(x) = e(x) > thr over [mode + 1, hi + 1)
    const int sl = sub_lane<W>(lane);
    const int g1 = obs <= mode ? obs + 1 : 2 * mode - obs + 1, g2 = obs <= mode ? 2 * mode - obs : obs;
    int w1 = g1 - W / 2, w2 = g2 - W / 2;
    w1 = w1 > *R1 - W ? *R1 - W : w1; w1 = w1 < *L1 ? *L1 : w1;
    w2 = w2 > *R2 - W ? *R2 - W : w2; w2 = w2 < *L2 ? *L2 : w2;
    const int x1 = w1 + sl, x2 = w2 + sl;
    const bool in1 = x1 < *R1, in2 = x2 < *R2;
    const double v1 = in1 ? T.e(x1) : 0.0, v2 = in2 ? T.e(x2) : 0.0;
    const int n1 = __builtin_popcountll(sub_ballot<W>(in1, lane)), n2 = __builtin_popcountll(sub_ballot<W>(in2, lane));
    const int c1 = __builtin_popcountll(sub_ballot<W>(in1 && v1 <= thr, lane)), c2 = __builtin_popcountll(sub_ballot<W>(in2 && v2 > thr, lane));
    bool ok = true;
    // count == 0: the boundary is at or left of the window's start; count == all: at or right of its end
    if (c1 == 0 && w1 > *L1) { *R1 = w1; ok = false; }
    else if (c1 == n1 && w1 + n1 < *R1) { *L1 = w1 + n1; ok = false; }
    else { *L1 = w1 + c1; *R1 = w1 + c1; }
    if (c2 == 0 && w2 > *L2) { *R2 = w2; ok = false; }
    else if (c2 == n2 && w2 + n2 < *R2) { *L2 = w2 + n2; ok = false; }
    else { *L2 = w2 + c2; *R2 = w2 + c2; }
    return ok;
}

// a sub-wave of W lanes computes the two-sided p of the table (a, b, c, d) = (A1, A2, U1, U2) (assoc.c:70); every lane of the
// sub-wave returns it
template <int W>
__device__ __forceinline__ double fisher_sub(int a, int b, int c, int d, const double *__restrict__ lf,
                                             const double *exp_tab /* LDS */, double rel_cut, int lane) {
    const int sl = sub_lane<W>(lane);
    const int r1 = a + b, r2 = c + d, c1 = a + c, nn = r1 + r2;
    const int lo = (c1 - r2) > 0 ? (c1 - r2) : 0;
    const int hi = r1 < c1 ? r1 : c1;
    FisherTab T;
    T.lf = lf; T.xt = exp_tab; T.r1 = r1; T.c1 = c1; T.d0 = r2 - c1;
    T.konst = lf[r1] + lf[r2] + lf[c1] + lf[nn - c1] - lf[nn];
    // included tables: P(x) <= P_obs * (1 + 1e-7), tested as ln P(x) <= ln P_obs + ln(1 + 1e-7)
    const double thr = T.e(a) + 9.9999995000000333e-08;
    double sum;
    if (hi - lo < 8 * W) {
        double part = 0.0;
        for (int x = lo + sl; x <= hi; x += W) {
            const double ex = T.e(x);
            if (ex <= thr) part += fisher_exp(ex, exp_tab);
        }
        sum = sub_sum_f64<W>(part);
    } else {
        // mode of the hypergeometric distribution, floor((r1 + 1)(c1 + 1) / (n + 2)), clamped to the support: the quotient
        // in double (the product is below 2^53), put right with the exact remainder
        const double prod = (double)(r1 + 1) * (double)(c1 + 1), den = (double)(nn + 2);      // exact: below 2^53
        double md = __builtin_floor(prod / den);
        const double rem = __builtin_fma(-md, den, prod);           // exact remainder of the candidate
        md += rem >= den ? 1.0 : (rem < 0.0 ? -1.0 : 0.0);
        int mode = (int)md;
        mode = mode < lo ? lo : (mode > hi ? hi : mode);
        // left of (and including) the mode p grows with x: included x are a prefix [lo, xL)
        // right of the mode p falls: p > thr on a prefix [mode+1, xR), included x are [xR, hi]
        int L1 = lo, R1 = mode + 1, L2 = mode + 1, R2 = hi + 1, xL, xR;
        if (fisher_boundary_windows<W>(T, thr, mode, a, lane, &L1, &R1, &L2, &R2)) { xL = L1; xR = L2; }
        else fisher_boundaries<W>(T, thr, L1, R1, L2, R2, lane, &xL, &xR);
        sum = fisher_tails<W>(T, xL, xR, lo, hi, lane, rel_cut);
    }
    return sum > 1.0 ? 1.0 : sum;
}

// the whole wave on one table (the per-batch kernel's form)
__device__ __forceinline__ double fisher_wave(int a, int b, int c, int d, const double *__restrict__ lf,
                                              const double *exp_tab /* LDS */, double rel_cut, int lane) {
    return fisher_sub<64>(a, b, c, d, lf, exp_tab, rel_cut, lane);
}

// W lanes per variant: a wave computes 64 / W variants side by side.  The fixed part of a variant (setup, mode, the two
// boundary windows: about a third of its instructions at W = 64) is then shared by 64 / W variants, and the last,
// partly useful turn of each tail wastes W / 2 tables on average instead of 32.
template <int W>
__global__ __launch_bounds__(256) void k_assoc_fisher(const int4 *__restrict__ counts, int n,
                                                      const double *__restrict__ lf,
                                                      double *__restrict__ odds,
                                                      double *__restrict__ pval, double rel_cut) {
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = k_exp2_j64[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    constexpr int PER_WAVE = 64 / W;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v = wave * PER_WAVE + lane / W;
    if (v >= n) return;
    const int4 c4 = counts[v];
    const int a = c4.x, b = c4.y, c = c4.z, d = c4.w;   // assoc.c:70: (A1, A2, U1, U2)
    const double p = fisher_sub<W>(a, b, c, d, lf, exp_tab, rel_cut, lane);
    if (sub_lane<W>(lane) == 0) {
        odds[v] = assoc_odds(a, b, c, d);
        pval[v] = p;
    }
}

// ---------------------------------------------------------------------------
// synthetic cohort: bit-reproducible with oracle/hpgv_oracle.c
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
#define HPGV_SYNTH_SEED 0x4850475631ULL

static __global__ void k_synth_thresholds(uint64_t v0, int n, uint32_t *__restrict__ thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = v0 + (uint64_t)i;
    const uint64_t h = splitmix64(HPGV_SYNTH_SEED ^ 0xA11E1EULL ^ v);
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    const double q = 0.05 + 0.45 * u;
    const uint32_t t_miss = 167772u;
    const double rest = (double)(16777216u - t_miss);
    const double omq = 1.0 - q;
    const double p00 = omq * omq;
    const double p01 = 2.0 * q * omq;
    const uint32_t t00 = t_miss + (uint32_t)floor(p00 * rest);
    const uint32_t t01 = t00 + (uint32_t)floor(p01 * rest);
    thr[3 * i + 0] = t_miss; thr[3 * i + 1] = t00; thr[3 * i + 2] = t01;
}

__device__ __forceinline__ uint32_t synth_gt(uint64_t vterm, uint64_t s, uint32_t tm, uint32_t t0,
                                             uint32_t t1) {
    const uint32_t r = (uint32_t)(splitmix64(vterm + s) >> 40);
    return (r < tm) ? 0xFFu : (r < t0) ? 0x00u : (r < t1) ? 0x01u : 0x11u;
}

// ---------------------------------------------------------------------------
// Per-tool recoding applied when a row is written in an engine layout.  The
// canonical exchange format stays HPGV8 (one byte per genotype); a layout may
// store, still one byte per genotype, the CLASS of that genotype for its scan:
//   RECODE_NONE  : HPGV8 as is (assoc; slow-family groups of the tdt layout)
//   RECODE_TDT   : positions [0,p16) father / [p16,2p16) mother planes hold the
//                  parent class, [2p16,3p16) the child class (tdt classes below)
//   RECODE_STATS : one-hot genotype cell / missing / extra-allele flags
//   RECODE_EPI   : the epistasis dataset code of vcf2epi (dataset_creator.c:255-266):
//                  0 "0/0", 1 heterozygous (a1 != a2), 2 homozygous non-reference, 255 not ALLELES_OK
//   RECODE_MENDEL: zero-ness class 0 "0/0", 1 one zero allele, 2 no zero allele, 3 not fully called
//                  (what check_mendel looks at), in father / mother / child planes
// Every class is a function of that ONE genotype and of the column's fixed role.
// ---------------------------------------------------------------------------
enum { RECODE_NONE = 0, RECODE_TDT = 1, RECODE_STATS = 2, RECODE_MENDEL = 3, RECODE_EPI = 4 };

// parent classes (tdt.c:113-123 tests): 0 "0/0", 1 "0/x", 2 "x/x" (equal, non-zero),
// 3 "x/y" (both non-zero, different); unusable = missing or "x/0" (tdt.c:103-108,119)
constexpr uint32_t TDT_F_UNUSABLE = 0x80u, TDT_M_UNUSABLE = 0x20u;
// child classes (tdt.c:175,182,203): 0 "0/0", 1 "0/x", 2 "x/0", 3 both non-zero, 4 missing
constexpr uint32_t TDT_C_INVALID = 4u;

__host__ __device__ __forceinline__ uint32_t tdt_parent_class(uint32_t g, uint32_t unusable) {
    const uint32_t a1 = g >> 4, a2 = g & 0xFu;
    if (a1 == 0xFu || a2 == 0xFu) return unusable;
    if (a1 && !a2) return unusable;
    if (!a1 && !a2) return 0u;
    if (!a1) return 1u;
    return (a1 == a2) ? 2u : 3u;
}
__host__ __device__ __forceinline__ uint32_t tdt_child_class(uint32_t g) {
    const uint32_t a1 = g >> 4, a2 = g & 0xFu;
    if (a1 == 0xFu || a2 == 0xFu) return TDT_C_INVALID;
    if (!a1 && !a2) return 0u;
    if (!a1) return 1u;
    if (!a2) return 2u;
    return 3u;
}
// stats flags: bit0..3 genotype is 0/0, 0/1, 1/0, 1/1; bit4 some allele missing; bit5 both
// missing; bit6 / bit7: a called allele 0 / 1 outside those four cells (e.g. "0/2", "./1")
__host__ __device__ __forceinline__ uint32_t stats_flags(uint32_t g) {
    const uint32_t a1 = g >> 4, a2 = g & 0xFu;
    uint32_t f = 0;
    if (g == 0x00u) f |= 1u; else if (g == 0x01u) f |= 2u; else if (g == 0x10u) f |= 4u; else if (g == 0x11u) f |= 8u;
    if (a1 == 0xFu || a2 == 0xFu) f |= 0x10u;
    if (a1 == 0xFu && a2 == 0xFu) f |= 0x20u;
    if (!(f & 0xFu)) {
        if (a1 == 0u || a2 == 0u) f |= 0x40u;
        if (a1 == 1u || a2 == 1u) f |= 0x80u;
    }
    return f;
}

__host__ __device__ __forceinline__ uint32_t mendel_class(uint32_t g) {
    const uint32_t a1 = g >> 4, a2 = g & 0xFu;
    if (a1 == 0xFu || a2 == 0xFu) return 3u;
    if (!a1 && !a2) return 0u;
    return (a1 && a2) ? 2u : 1u;
}

// g: HPGV8 byte (0xFF for padding), pos: byte position in the row
__device__ __forceinline__ uint32_t recode_byte(uint32_t g, int mode, int p16, int pos, bool is_pad) {
    if (mode == RECODE_TDT) {
        if (p16 <= 0 || pos >= 3 * p16) return g;
        if (pos < p16) return tdt_parent_class(g, TDT_F_UNUSABLE);
        if (pos < 2 * p16) return tdt_parent_class(g, TDT_M_UNUSABLE);
        return tdt_child_class(g);
    }
    if (mode == RECODE_STATS) return is_pad ? 0u : stats_flags(g);
    if (mode == RECODE_MENDEL) return mendel_class(g);
    if (mode == RECODE_EPI) {
        const uint32_t a1 = g >> 4, a2 = g & 0xFu;
        if (a1 == 0xFu || a2 == 0xFu) return 255u;          // dataset_creator.c:255-257
        if (!a1 && !a2) return 0u;                          // :259
        if (a1 != a2) return 1u;                            // :261
        return 2u;                                          // :263
    }
    return g;
}

// one thread per 16-byte chunk of the destination row; col_of_pos[p] = VCF column
// stored at row position p, or -1 for padding (0xFF before recoding).
static __global__ __launch_bounds__(256) void k_synth_layout(uint64_t v0, int n_variants, size_t pitch,
                                                      int chunks, const int32_t *__restrict__ col_of_pos,
                                                      const uint32_t *__restrict__ thr, int mode, int p16,
                                                      uint8_t *__restrict__ dst) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_variants * chunks;
    if (t >= total) return;
    const int i = (int)(t / chunks), c = (int)(t % chunks);
    const uint64_t v = v0 + (uint64_t)i;
    const uint64_t vterm = HPGV_SYNTH_SEED + v * 0x9E3779B97F4A7C15ULL;
    const uint32_t tm = thr[3 * i], t0 = thr[3 * i + 1], t1 = thr[3 * i + 2];
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = c * 16 + k * 4 + j;
            const int col = col_of_pos[pos];
            uint32_t g = (col < 0) ? 0xFFu : synth_gt(vterm, (uint64_t)col, tm, t0, t1);
            g = recode_byte(g, mode, p16, pos, col < 0);
            acc |= g << (8 * j);
        }
        w[k] = acc;
    }
    *reinterpret_cast<uint4 *>(dst + (size_t)i * pitch + (size_t)c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------------------
// layout (column gather + recode) kernel: dst[v][p] = recode(src[v][col_of_pos[p]]).
// strict != 0 turns any byte with a missing allele into 0xFF first (assoc / tdt
// drop such genotypes: assoc.c:53, tdt.c:103-108,154).
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_layout(const uint8_t *__restrict__ src, size_t src_pitch,
                                                int n_variants, size_t pitch, int chunks,
                                                const int32_t *__restrict__ col_of_pos, int strict,
                                                int mode, int p16, uint8_t *__restrict__ dst) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_variants * chunks;
    if (t >= total) return;
    const int i = (int)(t / chunks), c = (int)(t % chunks);
    const uint8_t *row = src + (size_t)i * src_pitch;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = c * 16 + k * 4 + j;
            const int col = col_of_pos[pos];
            uint32_t g = (col < 0) ? 0xFFu : (uint32_t)row[col];
            if (strict && (((g & 0xF) == 0xF) || ((g >> 4) == 0xF))) g = 0xFFu;
            g = recode_byte(g, mode, p16, pos, col < 0);
            acc |= g << (8 * j);
        }
        w[k] = acc;
    }
    *reinterpret_cast<uint4 *>(dst + (size_t)i * pitch + (size_t)c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// de-interleave helpers for the host entry points
static __global__ void k_counts_to_soa(const int4 *__restrict__ counts, int n, int32_t *A1, int32_t *A2,
                                int32_t *U1, int32_t *U2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = counts[i];
    A1[i] = c.x; A2[i] = c.y; U1[i] = c.z; U2[i] = c.w;
}

// ---------------------------------------------------------------------------
// streaming-read probe: a pure read of the buffer with the scan's machine shape
// (one wave = one contiguous 8 KiB piece per step, 8 x 16 B per lane in flight, two
// steps of look-ahead, non-temporal) and a single OR per dword, so the measured
// time is the memory system's, not the ALU's.  This is the ceiling the scans are
// compared with.
// ---------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void k_read_probe(const uint4 *__restrict__ buf, size_t n16,
                                                    uint32_t *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t piece = 512;                            // 16-byte chunks per wave step (8 KiB)
    uint32_t acc = 0;
    uint4 qa[8], qb[8];
    size_t p = wave * piece;
    auto issue = [&](uint4 (&q)[8], size_t base) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = base + (size_t)u * 64 + lane;
            q[u] = make_uint4(0, 0, 0, 0);
            if (i < n16) q[u] = load16<NT>(buf + i);
        }
    };
    auto eat = [&](const uint4 (&q)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= q[u].x | q[u].y | q[u].z | q[u].w;
    };
    if (p < n16) issue(qa, p);
    while (p < n16) {
        const size_t p1 = p + n_waves * piece;
        if (p1 < n16) issue(qb, p1);
        eat(qa);
        if (p1 >= n16) break;
        const size_t p2 = p1 + n_waves * piece;
        if (p2 < n16) issue(qa, p2);
        eat(qb);
        p = p2;
    }
    if (acc == 0x12345678u) sink[0] = acc;   // practically never; keeps the loads alive
}

}  // namespace hpgv
