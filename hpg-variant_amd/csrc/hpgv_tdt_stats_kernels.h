// hpgv_tdt_stats_kernels.h -- TDT trio scan, variant-stats scan and their
// FP64 statistics kernels (gfx950).  Same machine model as hpgv_kernels.h:
// one variant row per wavefront, 16-byte lane loads, bit-sliced SWAR logic.
#pragma once
#include "hpgv_kernels.h"

#include <string>
#include <vector>

namespace hpgv {

constexpr uint32_t K8 = 0x08080808u;   // "flag" bit of every byte in the SWAR masks

// per-byte flags (bit 3 of each byte) derived from 4 packed genotype bytes
struct GtFlags {
    uint32_t a1nz, a2nz;   // allele1 / allele2 non-zero (missing counts as non-zero)
    uint32_t ne;           // allele1 != allele2
    uint32_t valid;        // neither nibble is 0xF
};
__device__ __forceinline__ GtFlags gt_flags(uint32_t x) {
    GtFlags g;
    const uint32_t ind = nib_nonzero(x);
    g.a2nz = ind & K8;
    g.a1nz = (ind >> 4) & K8;
    g.ne = nib_nonzero((x ^ (x >> 4)) & 0x0F0F0F0Fu) & K8;
    const uint32_t nf = nib_not_f(x);
    g.valid = nf & (nf >> 4) & K8;
    return g;
}

// ---------------------------------------------------------------------------
// TDT on 4 trios at once (bytes of f, m, c at the same position form a trio).
// Bit-sliced statement of tdt.c:103-123 (family filters), the Mendel check
// (hpg-libs check_mendel as pinned by test/test_checks_family.c, classes by
// zero-ness) and the transmission table tdt.c:175-213 for families with ONE
// counted child (so trA/trB start at 0; multi-child families take the slow
// path below).  male: bit 3 set for male children, used on chr "X" rows only.
// ---------------------------------------------------------------------------
template <bool X>
__device__ __forceinline__ void tdt4(uint32_t f, uint32_t m, uint32_t c, uint32_t male, int &t1, int &t2) {
    const GtFlags F = gt_flags(f), M = gt_flags(m), C = gt_flags(c);
    // tdt.c:103-108 parents genotyped; :113 at least one het (a1 != a2); :119 no "x/0" parent
    uint32_t ok = F.valid & M.valid & C.valid & (F.ne | M.ne);
    ok &= ~(F.a1nz & ~F.a2nz) & ~(M.a1nz & ~M.a2nz);
    const uint32_t f_ref = ~(F.a1nz | F.a2nz) & K8, f_alt = F.a1nz & F.a2nz;
    const uint32_t m_ref = ~(M.a1nz | M.a2nz) & K8, m_alt = M.a1nz & M.a2nz;
    const uint32_t c_ref = ~(C.a1nz | C.a2nz) & K8, c_alt = C.a1nz & C.a2nz;
    const uint32_t c_het = ~(c_ref | c_alt) & K8;
    uint32_t err = (c_het & ((f_ref & m_ref) | (f_alt & m_alt))) | (c_ref & (f_alt | m_alt)) |
                   (c_alt & (f_ref | m_ref));
    if constexpr (X) {
        const uint32_t err_x = (c_alt & m_ref) | (c_ref & m_alt);
        err = (male & err_x) | (~male & err);
    }
    ok &= ~err;
    const uint32_t fh = ~F.a1nz & F.a2nz, mh = ~M.a1nz & M.a2nz;    // parent is "0/x"
    const uint32_t c0x = ~C.a1nz & C.a2nz;
    const uint32_t both_h = fh & mh;
    // trA == 1  (tdt.c:175-181 kid 00; :182-202 kid 0x)
    const uint32_t a1 = c_ref | (c0x & ((F.ne & (M.ne | M.a1nz)) | (~F.ne & F.a1nz)));
    const uint32_t A1 = ok & a1, A2 = ok & ~a1;
    const uint32_t B1 = ok & c_ref & both_h;
    const uint32_t B2 = ok & ((c0x & F.ne & M.ne) | (~(c_ref | c0x) & both_h));
    t1 += __builtin_popcount(A1) + __builtin_popcount(B1);
    t2 += __builtin_popcount(A2) + __builtin_popcount(B2);
}

// scalar statement of the same rules for one family with several counted
// children, keeping trA/trB at family scope (tdt.c:128-132)
__device__ __forceinline__ int mendel_code(bool is_x_male, int f1, int f2, int m1, int m2, int c1, int c2) {
    const bool f_ref = !f1 && !f2, f_alt = f1 && f2, m_ref = !m1 && !m2, m_alt = m1 && m2;
    const bool c_ref = !c1 && !c2, c_alt = c1 && c2;
    if (is_x_male) return (c_alt && m_ref) ? 9 : (c_ref && m_alt) ? 10 : 0;
    if (!c_ref && !c_alt) return (f_ref && m_ref) ? 1 : (f_alt && m_alt) ? 2 : 0;
    if (c_ref) return (f_alt && m_alt) ? 5 : m_alt ? 3 : f_alt ? 4 : 0;
    return (f_ref && m_ref) ? 8 : f_ref ? 6 : m_ref ? 7 : 0;
}

__device__ __forceinline__ void tdt_family_slow(const uint8_t *__restrict__ grp, int n_children,
                                                const uint8_t *__restrict__ male, bool x_row,
                                                int &t1, int &t2) {
    const uint32_t fb = grp[0], mb = grp[1];
    const int f1 = fb >> 4, f2 = fb & 0xF, m1 = mb >> 4, m2 = mb & 0xF;
    if (f1 == 0xF || f2 == 0xF || m1 == 0xF || m2 == 0xF) return;       // tdt.c:103-108
    if (f1 == f2 && m1 == m2) return;                                    // tdt.c:113-117
    if ((f1 && !f2) || (m1 && !m2)) return;                              // tdt.c:119-123
    int trA = 0, trB = 0;                                                // tdt.c:128-132
    for (int k = 0; k < n_children; ++k) {
        const uint32_t cb = grp[2 + k];
        const int c1 = cb >> 4, c2 = cb & 0xF;
        if (c1 == 0xF || c2 == 0xF) continue;                            // tdt.c:154
        if (mendel_code(x_row && male[2 + k], f1, f2, m1, m2, c1, c2)) continue;   // tdt.c:161-166
        const bool fh = !f1 && f2, mh = !m1 && m2;
        if (!c1 && !c2) {
            if (fh && mh) { trA = 1; trB = 1; } else { trA = 1; }
        } else if (!c1 && c2) {
            if (f1 != f2) {
                if (m1 != m2) { trA = 1; trB = 2; }
                else if (!m1) { trA = 2; }
                else { trA = 1; }
            } else if (!f1) { trA = 2; }
            else { trA = 1; }
        } else {
            if (fh && mh) { trA = 2; trB = 2; } else { trA = 2; }
        }
        if (trA == 1) t1++; else if (trA == 2) t2++;                     // tdt.c:235-239
        if (trB == 1) t1++; else if (trB == 2) t2++;
    }
}

// Row layout: [F plane P16 | M plane P16 | C plane P16 | slow groups | pad].
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_tdt_scan(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                  int pchunks /* P16/16 */, const uint8_t *__restrict__ male_plane,
                                                  int n_slow, const int32_t *__restrict__ slow_off,
                                                  const uint8_t *__restrict__ slow_male, int slow_base,
                                                  const uint8_t *__restrict__ is_x, int2 *__restrict__ tu, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    for (int i = 0; i < vpw; ++i) {
        const long v = v_begin + i;
        if (v >= n_variants) break;
        const uint8_t *rowb = gt + (size_t)v * pitch;
        const uint4 *rowF = reinterpret_cast<const uint4 *>(rowb);
        const uint4 *rowM = rowF + pchunks, *rowC = rowF + 2 * pchunks;
        const bool x_row = (is_x != nullptr) && (__builtin_amdgcn_readfirstlane((int)is_x[v]) != 0);
        int t1 = 0, t2 = 0;
        for (int base = 0; base < pchunks; base += 64 * U) {
            uint4 qf[U], qm[U], qc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                qf[u] = qm[u] = qc[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (c < pchunks) {
                    qf[u] = load16<NT>(rowF + c);
                    qm[u] = load16<NT>(rowM + c);
                    qc[u] = load16<NT>(rowC + c);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                if (!x_row) {
                    tdt4<false>(qf[u].x, qm[u].x, qc[u].x, 0, t1, t2);
                    tdt4<false>(qf[u].y, qm[u].y, qc[u].y, 0, t1, t2);
                    tdt4<false>(qf[u].z, qm[u].z, qc[u].z, 0, t1, t2);
                    tdt4<false>(qf[u].w, qm[u].w, qc[u].w, 0, t1, t2);
                } else {
                    uint4 ml = make_uint4(0, 0, 0, 0);
                    if (c < pchunks) ml = reinterpret_cast<const uint4 *>(male_plane)[c];
                    tdt4<true>(qf[u].x, qm[u].x, qc[u].x, ml.x, t1, t2);
                    tdt4<true>(qf[u].y, qm[u].y, qc[u].y, ml.y, t1, t2);
                    tdt4<true>(qf[u].z, qm[u].z, qc[u].z, ml.z, t1, t2);
                    tdt4<true>(qf[u].w, qm[u].w, qc[u].w, ml.w, t1, t2);
                }
            }
        }
        for (int k = lane; k < n_slow; k += 64) {
            const int off = slow_off[k], n_children = slow_off[k + 1] - off - 2;
            tdt_family_slow(rowb + slow_base + off, n_children, slow_male + off, x_row, t1, t2);
        }
        const int s1 = wave_sum(t1), s2 = wave_sum(t2);
        if (lane == 0) tu[v] = make_int2(s1, s2);
    }
}

// tdt.c:255-260 (integer square before the cast) and tdt.c:288-292
__global__ __launch_bounds__(256) void k_tdt_stats(const int2 *__restrict__ tu, int n, double *__restrict__ odds,
                                                   double *__restrict__ chisq, double *__restrict__ pval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t1 = tu[i].x, t2 = tu[i].y;
    double x = -1;
    if (t1 + t2 > 0) x = ((double)((t1 - t2) * (t1 - t2))) / (t1 + t2);
    const double d1 = t1, d2 = t2;
    odds[i] = (d2 == 0.0) ? __builtin_nan("") : (d1 / d2);
    chisq[i] = x;
    pval[i] = chisq_p_value(x);
}

// ---------------------------------------------------------------------------
// host-side plan of the TDT row layout
// ---------------------------------------------------------------------------
struct TdtPlan {
    int n_fast = 0, n_slow_families = 0;
    int pchunks = 0;
    int slow_base = 0;
    uint8_t *d_male_plane = nullptr;
    int32_t *d_slow_off = nullptr;
    uint8_t *d_slow_male = nullptr;

    void release() {
        if (d_male_plane) (void)hipFree(d_male_plane);
        if (d_slow_off) (void)hipFree(d_slow_off);
        if (d_slow_male) (void)hipFree(d_slow_male);
        d_male_plane = nullptr; d_slow_off = nullptr; d_slow_male = nullptr;
        n_fast = n_slow_families = pchunks = slow_base = 0;
    }

    int build(int n_samples, int n_families, const int32_t *father_col, const int32_t *mother_col,
              const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
              size_t row_align, std::vector<int32_t> &col_of_pos, size_t &pitch, std::string &why) {
        release();
        std::vector<int> fast, slow;
        for (int f = 0; f < n_families; ++f) {
            const int nc = child_off[f + 1] - child_off[f];
            if (nc < 0) { why = "child_off is not non-decreasing"; return 1; }
            if (father_col[f] >= n_samples || mother_col[f] >= n_samples) { why = "parent column out of range"; return 1; }
            for (int k = child_off[f]; k < child_off[f + 1]; ++k)
                if (!child_col || !child_sex || child_col[k] < 0 || child_col[k] >= n_samples) { why = "child column out of range"; return 1; }
            if (father_col[f] < 0 || mother_col[f] < 0 || nc == 0) continue;   // tdt.c:77-95: contributes nothing
            (nc == 1 ? fast : slow).push_back(f);
        }
        n_fast = (int)fast.size();
        n_slow_families = (int)slow.size();
        const size_t P16 = ((size_t)n_fast + 15) / 16 * 16;
        pchunks = (int)(P16 / 16);
        std::vector<int32_t> slow_off(1, 0);
        size_t slow_bytes = 0;
        for (int f : slow) { slow_bytes += 2 + (size_t)(child_off[f + 1] - child_off[f]); slow_off.push_back((int32_t)slow_bytes); }
        size_t used = 3 * P16 + slow_bytes;
        pitch = (used + row_align - 1) / row_align * row_align;
        if (pitch == 0) pitch = row_align;
        if (pitch > 0x7FFFFFFFu) { why = "row too long"; return 6; }
        slow_base = (int)(3 * P16);
        col_of_pos.assign(pitch, -1);
        std::vector<uint8_t> male_plane(P16 ? P16 : 16, 0), slow_male(slow_bytes ? slow_bytes : 16, 0);
        for (int t = 0; t < n_fast; ++t) {
            const int f = fast[t], k = child_off[f];
            col_of_pos[t] = father_col[f];
            col_of_pos[P16 + t] = mother_col[f];
            col_of_pos[2 * P16 + t] = child_col[k];
            male_plane[t] = (child_sex[k] == 0 /* HPGV_SEX_MALE */) ? 0x08 : 0x00;
        }
        for (size_t s = 0; s < slow.size(); ++s) {
            const int f = slow[s];
            size_t o = (size_t)slow_base + (size_t)slow_off[s];
            col_of_pos[o] = father_col[f];
            col_of_pos[o + 1] = mother_col[f];
            int j = 2;
            for (int k = child_off[f]; k < child_off[f + 1]; ++k, ++j) {
                col_of_pos[o + j] = child_col[k];
                slow_male[(size_t)slow_off[s] + j] = (child_sex[k] == 0) ? 1 : 0;
            }
        }
        auto up = [&](void **d, const void *h, size_t bytes) -> bool {
            if (hipMalloc(d, bytes) != hipSuccess) return false;
            return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
        };
        if (!up((void **)&d_male_plane, male_plane.data(), male_plane.size()) ||
            !up((void **)&d_slow_off, slow_off.data(), slow_off.size() * sizeof(int32_t)) ||
            !up((void **)&d_slow_male, slow_male.data(), slow_male.size())) {
            why = "hipMalloc/hipMemcpy of the TDT plan failed";
            release();
            return 3;
        }
        return 0;
    }

    void launch_scan(const uint8_t *d_gt, size_t pitch, int n_variants, const uint8_t *d_is_x, int2 *d_tu,
                     int vpw, bool nt, hipStream_t st) const {
        const long waves = ((long)n_variants + vpw - 1) / vpw;
        const unsigned blocks = (unsigned)((waves + 3) / 4);
        constexpr int U = 4;
        if (nt)
            hipLaunchKernelGGL((k_tdt_scan<true, U>), dim3(blocks), dim3(256), 0, st, d_gt, pitch, n_variants, pchunks,
                               d_male_plane, n_slow_families, d_slow_off, d_slow_male, slow_base, d_is_x, d_tu, vpw);
        else
            hipLaunchKernelGGL((k_tdt_scan<false, U>), dim3(blocks), dim3(256), 0, st, d_gt, pitch, n_variants, pchunks,
                               d_male_plane, n_slow_families, d_slow_off, d_slow_male, slow_base, d_is_x, d_tu, vpw);
    }
};

// ---------------------------------------------------------------------------
// variant stats scan (hpg-libs get_variants_stats, call site stats_runner.c:194;
// counting rules: oracle/hpgv_oracle.c orc_variant_stats).  Output per variant,
// 8 x int32: n_00 n_01 n_10 n_11 missing_genotypes missing_alleles allele0 allele1
// (allele counts include the called allele of half-missing genotypes).
// Pad / virtual bytes are 0xFF and are subtracted using the known slot count.
// ---------------------------------------------------------------------------
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_stats_scan(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                    int chunks, int n_samples, int4 *__restrict__ out8, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    const int slots = ((chunks + 64 * U - 1) / (64 * U)) * (64 * U);
    const int fake = slots * 16 - n_samples;          // 0xFF bytes that are not samples
    for (int i = 0; i < vpw; ++i) {
        const long v = v_begin + i;
        if (v >= n_variants) break;
        const uint4 *row = reinterpret_cast<const uint4 *>(gt + (size_t)v * pitch);
        int n00 = 0, n01 = 0, n10 = 0, n11 = 0, mg = 0, nnf = 0, a0 = 0, a1 = 0;
        for (int base = 0; base < chunks; base += 64 * U) {
            uint4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                q[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (c < chunks) q[u] = load16<NT>(row + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t x = w[k];
                    const uint32_t z = ~nib_nonzero(x);                 // bit3/bit7: nibble == 0
                    const uint32_t e = ~nib_nonzero(x ^ 0x11111111u);   // bit3/bit7: nibble == 1
                    const uint32_t zl = z & K8, zh = (z >> 4) & K8, el = e & K8, eh = (e >> 4) & K8;
                    a0 += __builtin_popcount(z & 0x88888888u);
                    a1 += __builtin_popcount(e & 0x88888888u);
                    n00 += __builtin_popcount(zl & zh);
                    n01 += __builtin_popcount(zh & el);
                    n10 += __builtin_popcount(eh & zl);
                    n11 += __builtin_popcount(eh & el);
                    const uint32_t nf = nib_not_f(x);
                    nnf += __builtin_popcount(nf);
                    mg += __builtin_popcount(~(nf & (nf >> 4)) & K8);
                }
            }
        }
        const int s00 = wave_sum(n00), s01 = wave_sum(n01), s10 = wave_sum(n10), s11 = wave_sum(n11);
        const int smg = wave_sum(mg) - fake;
        const int sma = (slots * 32 - wave_sum(nnf)) - 2 * fake;
        const int sa0 = wave_sum(a0), sa1 = wave_sum(a1);
        if (lane == 0) {
            out8[2 * v] = make_int4(s00, s01, s10, s11);
            out8[2 * v + 1] = make_int4(smg, sma, sa0, sa1);
        }
    }
}

// Hardy-Weinberg chi-square on (n_AA, n_Aa, n_aa) = (n_00, n_01 + n_10, n_11);
// definition: oracle/hpgv_oracle.c orc_hwe (hpg-libs body absent: unpinned)
__global__ __launch_bounds__(256) void k_stats_hwe(const int4 *__restrict__ in8, int n, double *__restrict__ chi2,
                                                   double *__restrict__ pval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = in8[2 * i];
    const int n_AA = c.x, n_Aa = c.y + c.z, n_aa = c.w;
    const int tot = n_AA + n_Aa + n_aa;
    if (tot == 0) { chi2[i] = __builtin_nan(""); pval[i] = __builtin_nan(""); return; }
    const double pf = (2.0 * n_AA + n_Aa) / (2.0 * tot);
    const double qf = 1.0 - pf;
    const double e_AA = pf * pf * tot, e_Aa = 2.0 * pf * qf * tot, e_aa = qf * qf * tot;
    double x = 0.0;
    if (e_AA > 0.0) x += ((n_AA - e_AA) * (n_AA - e_AA)) / e_AA;
    if (e_Aa > 0.0) x += ((n_Aa - e_Aa) * (n_Aa - e_Aa)) / e_Aa;
    if (e_aa > 0.0) x += ((n_aa - e_aa) * (n_aa - e_aa)) / e_aa;
    chi2[i] = x;
    pval[i] = chisq_p_value(x);
}

}  // namespace hpgv
