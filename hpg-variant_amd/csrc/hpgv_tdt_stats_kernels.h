// hpgv_tdt_stats_kernels.h -- TDT trio scan, variant-stats scan and their
// FP64 statistics kernels (gfx950).  Same machine model as hpgv_kernels.h:
// one variant row per wavefront, 16-byte lane loads, a handful of VALU ops per
// dword, DPP + readlane reduction, one small store per variant.
#pragma once
#include "hpgv_kernels.h"

#include <cstring>
#include <string>
#include <vector>

namespace hpgv {

// ---------------------------------------------------------------------------
// The TDT rule for one counted child, scalar, on allele values.  This is the
// product's statement of tdt.c:103-123 (family filters), the Mendel check
// (hpg-libs check_mendel as pinned by test/test_checks_family.c:16-111; classes
// by zero-ness) and the transmission table tdt.c:175-213.  It runs
//   - on the device for families with several counted children (slow groups),
//     where trA / trB live at family scope (tdt.c:128-132), and
//   - on the host, once per pedigree, over class representatives, to BUILD the
//     byte look-up tables the fast path evaluates with v_perm_b32.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ int mendel_code(bool is_x_male, int f1, int f2, int m1, int m2, int c1, int c2) {
    const bool f_ref = !f1 && !f2, f_alt = f1 && f2, m_ref = !m1 && !m2, m_alt = m1 && m2;
    const bool c_ref = !c1 && !c2, c_alt = c1 && c2;
    if (is_x_male) return (c_alt && m_ref) ? 9 : (c_ref && m_alt) ? 10 : 0;
    if (!c_ref && !c_alt) return (f_ref && m_ref) ? 1 : (f_alt && m_alt) ? 2 : 0;
    if (c_ref) return (f_alt && m_alt) ? 5 : m_alt ? 3 : f_alt ? 4 : 0;
    return (f_ref && m_ref) ? 8 : f_ref ? 6 : m_ref ? 7 : 0;
}

// parents usable? (tdt.c:103-123); alleles 0..14, 15 = missing
__host__ __device__ __forceinline__ bool tdt_parents_usable(int f1, int f2, int m1, int m2) {
    if (f1 == 0xF || f2 == 0xF || m1 == 0xF || m2 == 0xF) return false;   // tdt.c:103-108
    if (f1 == f2 && m1 == m2) return false;                                // tdt.c:113-117
    if ((f1 && !f2) || (m1 && !m2)) return false;                          // tdt.c:119-123
    return true;
}

// family-scope state of tdt.c:51-52,128-132 (values, not references: keeps it in registers)
struct TdtState { int trA, trB, t1, t2; };

// one child against usable parents; updates the family-scope trA / trB and the tallies
__host__ __device__ __forceinline__ TdtState tdt_child(int f1, int f2, int m1, int m2, int c1, int c2, bool x_male,
                                                       TdtState st) {
    int trA = st.trA, trB = st.trB, t1 = st.t1, t2 = st.t2;
    if (c1 == 0xF || c2 == 0xF) return st;                                 // tdt.c:154
    if (mendel_code(x_male, f1, f2, m1, m2, c1, c2)) return st;            // tdt.c:161-166
    const bool fh = !f1 && f2, mh = !m1 && m2;
    if (!c1 && !c2) {                                                      // tdt.c:175-181
        if (fh && mh) { trA = 1; trB = 1; } else { trA = 1; }
    } else if (!c1 && c2) {                                                // tdt.c:182-202
        if (f1 != f2) {
            if (m1 != m2) { trA = 1; trB = 2; }
            else if (!m1) { trA = 2; }
            else { trA = 1; }
        } else if (!f1) { trA = 2; }
        else { trA = 1; }
    } else {                                                               // tdt.c:203-213
        if (fh && mh) { trA = 2; trB = 2; } else { trA = 2; }
    }
    if (trA == 1) t1++; else if (trA == 2) t2++;                           // tdt.c:235-239
    if (trB == 1) t1++; else if (trB == 2) t2++;
    return TdtState{trA, trB, t1, t2};
}

// returns the family's (t1, t2) contribution
__device__ __forceinline__ int2 tdt_family_slow(const uint8_t *__restrict__ grp, int n_children,
                                                const uint8_t *__restrict__ male, bool x_row) {
    const uint32_t fb = grp[0], mb = grp[1];
    const int f1 = fb >> 4, f2 = fb & 0xF, m1 = mb >> 4, m2 = mb & 0xF;
    if (!tdt_parents_usable(f1, f2, m1, m2)) return make_int2(0, 0);
    TdtState st = {0, 0, 0, 0};                                            // tdt.c:128-132
    for (int k = 0; k < n_children; ++k) {
        const uint32_t cb = grp[2 + k];
        st = tdt_child(f1, f2, m1, m2, (int)(cb >> 4), (int)(cb & 0xF), x_row && male[2 + k], st);
    }
    return make_int2(st.t1, st.t2);
}

// ---------------------------------------------------------------------------
// Fast path: byte look-up tables evaluated 4 trios at a time with v_perm_b32.
//
// The father / mother / child planes hold CLASS bytes (hpgv_kernels.h
// tdt_parent_class / tdt_child_class).  For a single counted child the tallies
// depend only on (father class, mother class, child class):
//   pair  = PAIR[m*4 + f]      one-hot "pair class" byte (16-entry table = two
//                              8-byte v_perm tables + one v_perm to pick by bit 3)
//   t1   += popcount(pair & C1[child]),  t2 += popcount(pair & C2[child])
// where C1/C2 are 8-entry tables of the pair classes that add to t1 / t2 for that
// child class; a pair class that adds 2 owns a second ("dup") bit.  Unusable
// parents carry a code >= 0x20, which (a) is killed by K = v_perm(0, 0, f|m):
// selector bytes >= 13 give 0xFF, selectors 0..3 give the 0x00 table bytes.
// ---------------------------------------------------------------------------
struct TdtLut {
    uint32_t pair_lo[2];   // PAIR[0..7]   (idx = mother*4 + father)
    uint32_t pair_hi[2];   // PAIR[8..15]
    uint32_t c1[2];        // child class -> pair-class mask counted into t1
    uint32_t c2[2];        // child class -> pair-class mask counted into t2
};
struct TdtLuts {
    TdtLut autosome;       // joint encoding: one pair byte serves t1 and t2
    TdtLut xmale_t1;       // chr "X", male child: separate encodings for t1 ...
    TdtLut xmale_t2;       // ... and t2 (8 pair classes + a dup bit do not fit one byte)
};

__device__ __forceinline__ uint32_t lut8(uint32_t t0, uint32_t t1, uint32_t sel) {
    return __builtin_amdgcn_perm(t1, t0, sel);              // selector 0..3 -> t0 bytes, 4..7 -> t1 bytes
}
__device__ __forceinline__ uint32_t tdt_pair(const TdtLut &L, uint32_t idx) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t lo = lut8(L.pair_lo[0], L.pair_lo[1], sel), hi = lut8(L.pair_hi[0], L.pair_hi[1], sel);
    const uint32_t pick = ((idx >> 1) & 0x04040404u) | 0x03020100u;   // byte i: i (lo) or i + 4 (hi)
    return __builtin_amdgcn_perm(hi, lo, pick);
}

template <bool X>
__device__ __forceinline__ int2 tdt4(const TdtLuts &L, uint32_t f, uint32_t m, uint32_t c, uint32_t male) {
    const uint32_t idx = (m << 2) | f;
    const uint32_t kill = __builtin_amdgcn_perm(0u, 0u, f | m);       // 0xFF where a parent is unusable
    uint32_t w1, w2;
    if constexpr (!X) {
        const uint32_t pair = tdt_pair(L.autosome, idx) & ~kill;
        w1 = pair & lut8(L.autosome.c1[0], L.autosome.c1[1], c);
        w2 = pair & lut8(L.autosome.c2[0], L.autosome.c2[1], c);
    } else {
        const uint32_t pa = tdt_pair(L.autosome, idx);
        const uint32_t a1 = pa & lut8(L.autosome.c1[0], L.autosome.c1[1], c), a2 = pa & lut8(L.autosome.c2[0], L.autosome.c2[1], c);
        const uint32_t x1 = tdt_pair(L.xmale_t1, idx) & lut8(L.xmale_t1.c1[0], L.xmale_t1.c1[1], c);
        const uint32_t x2 = tdt_pair(L.xmale_t2, idx) & lut8(L.xmale_t2.c2[0], L.xmale_t2.c2[1], c);
        w1 = ((male & x1) | (~male & a1)) & ~kill;                    // male: 0xFF per male child
        w2 = ((male & x2) | (~male & a2)) & ~kill;
    }
    return make_int2(__builtin_popcount(w1), __builtin_popcount(w2));
}

// Row layout: [F plane P16 | M plane P16 | C plane P16 | slow groups (HPGV8) | pad].
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_tdt_scan(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                  int pchunks /* P16/16 */, TdtLuts luts,
                                                  const uint8_t *__restrict__ male_plane,
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
        const uint32_t plane = (uint32_t)pchunks * 16u;
        const bool x_row = (is_x != nullptr) && (__builtin_amdgcn_readfirstlane((int)is_x[v]) != 0);
        int t1 = 0, t2 = 0;
        for (int base = 0; base < pchunks; base += 64 * U) {
            uint4 qf[U], qm[U], qc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                // virtual trio: unusable parents, missing child
                qf[u] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
                qm[u] = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
                qc[u] = make_uint4(0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u);
                if (c < pchunks) {
                    qf[u] = load16o<NT>(rowb, (uint32_t)c * 16u);
                    qm[u] = load16o<NT>(rowb, (uint32_t)c * 16u + plane);
                    qc[u] = load16o<NT>(rowb, (uint32_t)c * 16u + 2u * plane);
                }
            }
            if (!x_row) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    { const int2 d = tdt4<false>(luts, qf[u].x, qm[u].x, qc[u].x, 0); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<false>(luts, qf[u].y, qm[u].y, qc[u].y, 0); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<false>(luts, qf[u].z, qm[u].z, qc[u].z, 0); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<false>(luts, qf[u].w, qm[u].w, qc[u].w, 0); t1 += d.x; t2 += d.y; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = base + u * 64 + lane;
                    uint4 ml = make_uint4(0, 0, 0, 0);
                    if (c < pchunks) ml = reinterpret_cast<const uint4 *>(male_plane)[c];
                    { const int2 d = tdt4<true>(luts, qf[u].x, qm[u].x, qc[u].x, ml.x); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<true>(luts, qf[u].y, qm[u].y, qc[u].y, ml.y); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<true>(luts, qf[u].z, qm[u].z, qc[u].z, ml.z); t1 += d.x; t2 += d.y; }
                    { const int2 d = tdt4<true>(luts, qf[u].w, qm[u].w, qc[u].w, ml.w); t1 += d.x; t2 += d.y; }
                }
            }
        }
        for (int k = lane; k < n_slow; k += 64) {
            const int off = slow_off[k], n_children = slow_off[k + 1] - off - 2;
            const int2 d = tdt_family_slow(rowb + slow_base + off, n_children, slow_male + off, x_row);
            t1 += d.x; t2 += d.y;
        }
        const int s1 = wave_sum(t1), s2 = wave_sum(t2);
        if (lane == 0) tu[v] = make_int2(s1, s2);
    }
}

// tdt.c:255-260 (integer square before the cast) and tdt.c:288-292
static __global__ __launch_bounds__(256) void k_tdt_stats(const int2 *__restrict__ tu, int n, double *__restrict__ odds,
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
// host side: pedigree -> row layout + look-up tables
// ---------------------------------------------------------------------------
namespace tdt_host {

// weights[f][m][c] in {0,1,2} -> one-hot pair bytes (idx = m*4+f) and child masks.
// Pair classes = distinct non-zero weight vectors over the 4 child classes; a class
// with a weight 2 somewhere owns a second bit.  Returns false if 8 bits do not suffice.
inline bool encode(const int w1[4][4][4], const int w2[4][4][4], bool use1, bool use2,
                   uint8_t pair[16], uint8_t c1[8], uint8_t c2[8]) {
    struct Cls { int v1[4], v2[4]; int bit, dup; };
    std::vector<Cls> cls;
    int next_bit = 0;
    memset(pair, 0, 16); memset(c1, 0, 8); memset(c2, 0, 8);
    for (int m = 0; m < 4; ++m)
        for (int f = 0; f < 4; ++f) {
            Cls k{};
            bool zero = true, has2 = false;
            for (int c = 0; c < 4; ++c) {
                k.v1[c] = use1 ? w1[f][m][c] : 0;
                k.v2[c] = use2 ? w2[f][m][c] : 0;
                if (k.v1[c] || k.v2[c]) zero = false;
                if (k.v1[c] == 2 || k.v2[c] == 2) has2 = true;
                if (k.v1[c] > 2 || k.v2[c] > 2) return false;
            }
            if (zero) continue;
            int found = -1;
            for (size_t i = 0; i < cls.size(); ++i)
                if (!memcmp(cls[i].v1, k.v1, sizeof k.v1) && !memcmp(cls[i].v2, k.v2, sizeof k.v2)) { found = (int)i; break; }
            if (found < 0) {
                k.bit = next_bit++;
                k.dup = has2 ? next_bit++ : -1;
                if (next_bit > 8) return false;
                cls.push_back(k);
                found = (int)cls.size() - 1;
            }
            const Cls &q = cls[found];
            pair[m * 4 + f] = (uint8_t)((1u << q.bit) | (q.dup >= 0 ? (1u << q.dup) : 0u));
        }
    for (const Cls &q : cls)
        for (int c = 0; c < 4; ++c) {
            if (q.v1[c] >= 1) c1[c] |= (uint8_t)(1u << q.bit);
            if (q.v1[c] == 2) c1[c] |= (uint8_t)(1u << q.dup);
            if (q.v2[c] >= 1) c2[c] |= (uint8_t)(1u << q.bit);
            if (q.v2[c] == 2) c2[c] |= (uint8_t)(1u << q.dup);
        }
    return true;
}

inline void pack(const uint8_t pair[16], const uint8_t c1[8], const uint8_t c2[8], TdtLut &L) {
    auto dw = [](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
    L.pair_lo[0] = dw(pair); L.pair_lo[1] = dw(pair + 4);
    L.pair_hi[0] = dw(pair + 8); L.pair_hi[1] = dw(pair + 12);
    L.c1[0] = dw(c1); L.c1[1] = dw(c1 + 4);
    L.c2[0] = dw(c2); L.c2[1] = dw(c2 + 4);
}

// Runs the scalar rule over class representatives and builds the three tables.
inline bool build_luts(TdtLuts &out) {
    // representatives: parent classes 0 "0/0", 1 "0/1", 2 "1/1", 3 "1/2"; child 0 "0/0", 1 "0/1", 2 "1/0", 3 "1/1"
    static const int P[4][2] = {{0, 0}, {0, 1}, {1, 1}, {1, 2}};
    static const int C[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
    int w1[2][4][4][4], w2[2][4][4][4];
    for (int x = 0; x < 2; ++x)
        for (int f = 0; f < 4; ++f)
            for (int m = 0; m < 4; ++m)
                for (int c = 0; c < 4; ++c) {
                    TdtState st = {0, 0, 0, 0};
                    if (tdt_parents_usable(P[f][0], P[f][1], P[m][0], P[m][1]))
                        st = tdt_child(P[f][0], P[f][1], P[m][0], P[m][1], C[c][0], C[c][1], x == 1, st);
                    w1[x][f][m][c] = st.t1; w2[x][f][m][c] = st.t2;
                }
    uint8_t pair[16], c1[8], c2[8];
    if (!encode(w1[0], w2[0], true, true, pair, c1, c2)) return false;
    pack(pair, c1, c2, out.autosome);
    if (!encode(w1[1], w2[1], true, false, pair, c1, c2)) return false;
    pack(pair, c1, c2, out.xmale_t1);
    if (!encode(w1[1], w2[1], false, true, pair, c1, c2)) return false;
    pack(pair, c1, c2, out.xmale_t2);
    return true;
}

}  // namespace tdt_host

struct TdtPlan {
    int n_fast = 0, n_slow_families = 0;
    int pchunks = 0;
    int p16 = 0;
    int slow_base = 0;
    TdtLuts luts{};
    uint8_t *d_male_plane = nullptr;
    int32_t *d_slow_off = nullptr;
    uint8_t *d_slow_male = nullptr;

    void release() {
        if (d_male_plane) (void)hipFree(d_male_plane);
        if (d_slow_off) (void)hipFree(d_slow_off);
        if (d_slow_male) (void)hipFree(d_slow_male);
        d_male_plane = nullptr; d_slow_off = nullptr; d_slow_male = nullptr;
        n_fast = n_slow_families = pchunks = p16 = slow_base = 0;
    }

    // return codes are hpgv status values (1 invalid, 3 hip, 6 unsupported)
    int build(int n_samples, int n_families, const int32_t *father_col, const int32_t *mother_col,
              const int32_t *child_off, const int32_t *child_col, const uint8_t *child_sex,
              size_t row_align, std::vector<int32_t> &col_of_pos, size_t &pitch, std::string &why) {
        release();
        if (!tdt_host::build_luts(luts)) { why = "TDT look-up tables do not fit one byte"; return 6; }
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
        p16 = (int)P16;
        std::vector<int32_t> slow_off(1, 0);
        size_t slow_bytes = 0;
        for (int f : slow) { slow_bytes += 2 + (size_t)(child_off[f + 1] - child_off[f]); slow_off.push_back((int32_t)slow_bytes); }
        size_t used = 3 * P16 + slow_bytes;
        pitch = (used + row_align - 1) / row_align * row_align;
        if (pitch == 0) pitch = row_align;
        if (pitch > 0x3FFFFFFFu) { why = "row too long"; return 6; }
        slow_base = (int)(3 * P16);
        col_of_pos.assign(pitch, -1);
        std::vector<uint8_t> male_plane(P16 ? P16 : 16, 0), slow_male(slow_bytes ? slow_bytes : 16, 0);
        for (int t = 0; t < n_fast; ++t) {
            const int f = fast[t], k = child_off[f];
            col_of_pos[t] = father_col[f];
            col_of_pos[P16 + t] = mother_col[f];
            col_of_pos[2 * P16 + t] = child_col[k];
            male_plane[t] = (child_sex[k] == 0 /* HPGV_SEX_MALE */) ? 0xFF : 0x00;
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
            hipLaunchKernelGGL((k_tdt_scan<true, U>), dim3(blocks), dim3(256), 0, st, d_gt, pitch, n_variants, pchunks, luts,
                               d_male_plane, n_slow_families, d_slow_off, d_slow_male, slow_base, d_is_x, d_tu, vpw);
        else
            hipLaunchKernelGGL((k_tdt_scan<false, U>), dim3(blocks), dim3(256), 0, st, d_gt, pitch, n_variants, pchunks, luts,
                               d_male_plane, n_slow_families, d_slow_off, d_slow_male, slow_base, d_is_x, d_tu, vpw);
    }
};

// ---------------------------------------------------------------------------
// variant stats scan (hpg-libs get_variants_stats, call site stats_runner.c:194;
// counting rules: oracle/hpgv_oracle.c orc_variant_stats).  Rows hold the
// one-hot flag bytes of hpgv_kernels.h stats_flags (pad bytes are 0), so every
// counter is a masked popcount.  Output per variant, 8 x int32:
//   n_00 n_01 n_10 n_11 missing_genotypes missing_alleles allele0 allele1
// (allele counts include the called allele of half-missing genotypes).
// ---------------------------------------------------------------------------
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_stats_scan(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                    uint32_t row_off /* start of the scanned segment */,
                                                    int chunks, int4 *__restrict__ out8, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    for (int i = 0; i < vpw; ++i) {
        const long v = v_begin + i;
        if (v >= n_variants) break;
        const uint8_t *row = gt + (size_t)v * pitch + row_off;
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int base = 0; base < chunks; base += 64 * U) {
            uint4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                q[u] = make_uint4(0u, 0u, 0u, 0u);
                if (c < chunks) q[u] = load16o<NT>(row, (uint32_t)c * 16u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int b = 0; b < 8; ++b) cnt[b] += __builtin_popcount(w[k] & (0x01010101u << b));
            }
        }
        int s[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) s[b] = wave_sum(cnt[b]);
        if (lane == 0) {
            out8[2 * v] = make_int4(s[0], s[1], s[2], s[3]);
            out8[2 * v + 1] = make_int4(s[4], s[4] + s[5], 2 * s[0] + s[1] + s[2] + s[6], 2 * s[3] + s[1] + s[2] + s[7]);
        }
    }
}

// ---------------------------------------------------------------------------
// The same scan with bit-sliced counting.  The eight flag counters are eight positional popcounts over the
// row's dwords (bit b of every byte); counting them one masked popcount at a time costs 16 VALU operations
// per dword and made the scan VALU-limited (about two thirds of the issue slots at 15 k samples).  A
// carry-save adder tree (Harley-Seal) folds 16 dwords into ones / twos / fours / eights words plus one
// "sixteens" word with 15 adders of 3 operations (xor, xor, v_bfi_b32) and only the sixteens word is popcounted
// (8 masked popcounts per 16 dwords): 3.8 operations per dword.  The four residual words are popcounted once at
// the row's end.  Tiles of 4 chunks (16 dwords per lane = one adder-tree round) are software-pipelined across
// tiles and rows like the assoc scan: the loads of tile t + 1 are issued before tile t is counted.
// ---------------------------------------------------------------------------
struct StatsAcc { uint32_t ones, twos, fours, eights; int c16[8]; };

__device__ __forceinline__ void csa(uint32_t &h, uint32_t &l, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t u = a ^ b;
    l = u ^ c;
    h = (u & c) | (~u & a);                                    // majority(a, b, c): v_bfi_b32
}

__device__ __forceinline__ void stats_count_tile(const uint4 (&q)[4], StatsAcc &a) {
    const uint32_t w[16] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                            q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
    uint32_t t2a, t2b, t4a, t4b, t8a, t8b, s16;
    csa(t2a, a.ones, a.ones, w[0], w[1]);   csa(t2b, a.ones, a.ones, w[2], w[3]);   csa(t4a, a.twos, a.twos, t2a, t2b);
    csa(t2a, a.ones, a.ones, w[4], w[5]);   csa(t2b, a.ones, a.ones, w[6], w[7]);   csa(t4b, a.twos, a.twos, t2a, t2b);
    csa(t8a, a.fours, a.fours, t4a, t4b);
    csa(t2a, a.ones, a.ones, w[8], w[9]);   csa(t2b, a.ones, a.ones, w[10], w[11]); csa(t4a, a.twos, a.twos, t2a, t2b);
    csa(t2a, a.ones, a.ones, w[12], w[13]); csa(t2b, a.ones, a.ones, w[14], w[15]); csa(t4b, a.twos, a.twos, t2a, t2b);
    csa(t8b, a.fours, a.fours, t4a, t4b);
    csa(s16, a.eights, a.eights, t8a, t8b);
#pragma unroll
    for (int b = 0; b < 8; ++b) a.c16[b] += __builtin_popcount(s16 & (0x01010101u << b));
}

template <bool NT>
__device__ __forceinline__ void stats_issue_tile(uint4 (&q)[4], const uint8_t *__restrict__ g0, size_t pitch, int t, int ipr,
                                                 int chunks, int lane) {
    const int r = t / ipr, base = (t - r * ipr) * 256;
    const uint8_t *row = g0 + (size_t)r * pitch;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = base + u * 64 + lane;
        q[u] = make_uint4(0u, 0u, 0u, 0u);                       // beyond the row: no flags
        if (c < chunks) q[u] = load16o<NT>(row, (uint32_t)c * 16u);
    }
}

__device__ __forceinline__ void stats_consume_tile(const uint4 (&q)[4], StatsAcc &a, int t, int ipr, long v_begin, int lane,
                                                   int4 *__restrict__ out8) {
    stats_count_tile(q, a);
    const int r = t / ipr;
    if (t - r * ipr == ipr - 1) {                               // the row's last tile: weigh the residual words, reduce, store
        int s[8], c[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t m = 0x01010101u << b;
            c[b] = 16 * a.c16[b] + 8 * __builtin_popcount(a.eights & m) + 4 * __builtin_popcount(a.fours & m) +
                   2 * __builtin_popcount(a.twos & m) + __builtin_popcount(a.ones & m);
            a.c16[b] = 0;
        }
        if (ipr * 4096 < 65536) {                               // a row of fewer than 65 536 bytes: two counters share a wave sum
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                const int two = wave_sum(c[b] | (c[b + 1] << 16));
                s[b] = two & 0xFFFF; s[b + 1] = (int)((uint32_t)two >> 16);
            }
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) s[b] = wave_sum(c[b]);
        }
        a.ones = a.twos = a.fours = a.eights = 0u;
        const long v = v_begin + r;
        if (lane == 0) {
            out8[2 * v] = make_int4(s[0], s[1], s[2], s[3]);
            out8[2 * v + 1] = make_int4(s[4], s[4] + s[5], 2 * s[0] + s[1] + s[2] + s[6], 2 * s[3] + s[1] + s[2] + s[7]);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_stats_scan_hs(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                       uint32_t row_off, int chunks, int4 *__restrict__ out8, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    if (v_begin >= n_variants) return;
    const int rows = (int)((v_begin + vpw <= n_variants) ? vpw : (n_variants - v_begin));
    const int ipr = chunks > 0 ? (chunks + 255) / 256 : 1;       // tiles per row (an empty segment still stores its zeros)
    const int n_tiles = rows * ipr;
    const uint8_t *g0 = gt + (size_t)v_begin * pitch + row_off;
    StatsAcc acc;
    acc.ones = acc.twos = acc.fours = acc.eights = 0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) acc.c16[b] = 0;
    uint4 qa[4], qb[4];
    stats_issue_tile<NT>(qa, g0, pitch, 0, ipr, chunks, lane);
    for (int t = 0; t < n_tiles; t += 2) {
        if (t + 1 < n_tiles) stats_issue_tile<NT>(qb, g0, pitch, t + 1, ipr, chunks, lane);
        stats_consume_tile(qa, acc, t, ipr, v_begin, lane, out8);
        if (t + 1 < n_tiles) {
            if (t + 2 < n_tiles) stats_issue_tile<NT>(qa, g0, pitch, t + 2, ipr, chunks, lane);
            stats_consume_tile(qb, acc, t + 1, ipr, v_begin, lane, out8);
        }
    }
}

// ---------------------------------------------------------------------------
// per-sample missing-genotype counts (hpg-libs get_sample_stats, call site
// stats_runner.c:197-198) over the same flag rows: a column-wise sum.  Wave =
// one tile of 1024 samples (64 lanes x 16 B) x SB variants; bytes accumulate in
// SWAR lanes (SB <= 255 so no byte overflows), then 16 atomics per lane.
// ---------------------------------------------------------------------------
constexpr int SAMPLE_STATS_ROWS = 128;
static __global__ __launch_bounds__(256) void k_sample_missing(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                        int chunks, int n_samples, int32_t *__restrict__ missing) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // column tile
    const int c = tile * 64 + lane;                                            // 16-byte chunk in the row
    const long v0 = (long)blockIdx.y * SAMPLE_STATS_ROWS;
    if (tile * 64 >= chunks) return;
    uint32_t acc[4] = {0, 0, 0, 0};
    const int rows = (int)((v0 + SAMPLE_STATS_ROWS <= n_variants) ? SAMPLE_STATS_ROWS : (n_variants - v0));
    if (c < chunks) {
#pragma unroll 4
        for (int r = 0; r < rows; ++r) {
            const uint4 q = load16o<true>(gt + (size_t)(v0 + r) * pitch, (uint32_t)c * 16u);
            acc[0] += (q.x >> 4) & 0x01010101u;      // flag bit 4: some allele missing
            acc[1] += (q.y >> 4) & 0x01010101u;
            acc[2] += (q.z >> 4) & 0x01010101u;
            acc[3] += (q.w >> 4) & 0x01010101u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = c * 16 + k * 4 + j;
                const int n = (int)((acc[k] >> (8 * j)) & 0xFFu);
                if (n && col < n_samples) atomicAdd(missing + col, n);
            }
    }
}

// ---------------------------------------------------------------------------
// full genotype table of one variant from a RAW HPGV8 row (any alleles 0..14):
// 256-bin histogram of the code byte, one workgroup per listed variant, LDS
// atomics.  Meant for the few multi-allelic variants the flag scan reports
// (cells 0/0..1/1 do not add up), not for the bulk.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_genotype_table(const uint8_t *__restrict__ raw, size_t src_pitch,
                                                        int n_samples, const int32_t *__restrict__ variant_idx,
                                                        int n_idx, int32_t *__restrict__ table) {
    __shared__ int hist[256];
    const int i = blockIdx.x;
    if (i >= n_idx) return;
    const long v = variant_idx ? (long)variant_idx[i] : (long)i;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *row = raw + (size_t)v * src_pitch;
    for (int j = threadIdx.x; j < n_samples; j += blockDim.x) atomicAdd(&hist[row[j]], 1);
    __syncthreads();
    table[(size_t)i * 256 + threadIdx.x] = hist[threadIdx.x];
}

// ---------------------------------------------------------------------------
// Mendelian errors per variant and per child (hpg-libs check_mendel, used by the
// --mendel filter and by get_sample_stats; rule = mendel_code above).  Rows hold three
// planes [father | mother | child] of zero-ness classes.  ERR[m*4+f] is a 3-bit mask:
// bit c set <=> a child of class c is an error for that parent pair; the child byte
// selects its bit through an 8-entry v_perm table.  Chr "X" male children use the
// second table (only the mother matters).
// ---------------------------------------------------------------------------
struct MendelLut { uint32_t lo[2], hi[2]; };            // ERR[0..7], ERR[8..15]
struct MendelLuts { MendelLut autosome, xmale; };

__device__ __forceinline__ uint32_t mendel_err4(const MendelLut &L, uint32_t idx, uint32_t c) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t lo = lut8(L.lo[0], L.lo[1], sel), hi = lut8(L.hi[0], L.hi[1], sel);
    const uint32_t pair = __builtin_amdgcn_perm(hi, lo, ((idx >> 1) & 0x04040404u) | 0x03020100u);
    // child class -> its bit (class 3 = not called -> 0)
    const uint32_t cbit = lut8(0x00040201u, 0u, c);
    return pair & cbit;
}
// 0 / 1 per byte: is one of bits 0..2 set?
__device__ __forceinline__ uint32_t any3(uint32_t z) { return (z | (z >> 1) | (z >> 2)) & 0x01010101u; }

template <bool X>
__device__ __forceinline__ uint32_t mendel4(const MendelLuts &L, uint32_t f, uint32_t m, uint32_t c, uint32_t male) {
    const uint32_t idx = (m << 2) | f;
    uint32_t z = mendel_err4(L.autosome, idx, c);
    if constexpr (X) {
        const uint32_t zx = mendel_err4(L.xmale, idx, c);
        z = (male & zx) | (~male & z);
    }
    return any3(z);
}

template <bool NT, int U>
__global__ __launch_bounds__(256) void k_mendel_scan(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                     int pchunks, MendelLuts luts, const uint8_t *__restrict__ male_plane,
                                                     const uint8_t *__restrict__ is_x, int32_t *__restrict__ errors, int vpw) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long v_begin = wave * vpw;
    const uint32_t plane = (uint32_t)pchunks * 16u;
    for (int i = 0; i < vpw; ++i) {
        const long v = v_begin + i;
        if (v >= n_variants) break;
        const uint8_t *rowb = gt + (size_t)v * pitch;
        const bool x_row = (is_x != nullptr) && (__builtin_amdgcn_readfirstlane((int)is_x[v]) != 0);
        int n = 0;
        for (int base = 0; base < pchunks; base += 64 * U) {
            uint4 qf[U], qm[U], qc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                qf[u] = qm[u] = qc[u] = make_uint4(0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u);   // not called
                if (c < pchunks) {
                    qf[u] = load16o<NT>(rowb, (uint32_t)c * 16u);
                    qm[u] = load16o<NT>(rowb, (uint32_t)c * 16u + plane);
                    qc[u] = load16o<NT>(rowb, (uint32_t)c * 16u + 2u * plane);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = base + u * 64 + lane;
                uint4 ml = make_uint4(0, 0, 0, 0);
                if (x_row && c < pchunks) ml = reinterpret_cast<const uint4 *>(male_plane)[c];
                if (!x_row) {
                    n += __builtin_popcount(mendel4<false>(luts, qf[u].x, qm[u].x, qc[u].x, 0)) +
                         __builtin_popcount(mendel4<false>(luts, qf[u].y, qm[u].y, qc[u].y, 0)) +
                         __builtin_popcount(mendel4<false>(luts, qf[u].z, qm[u].z, qc[u].z, 0)) +
                         __builtin_popcount(mendel4<false>(luts, qf[u].w, qm[u].w, qc[u].w, 0));
                } else {
                    n += __builtin_popcount(mendel4<true>(luts, qf[u].x, qm[u].x, qc[u].x, ml.x)) +
                         __builtin_popcount(mendel4<true>(luts, qf[u].y, qm[u].y, qc[u].y, ml.y)) +
                         __builtin_popcount(mendel4<true>(luts, qf[u].z, qm[u].z, qc[u].z, ml.z)) +
                         __builtin_popcount(mendel4<true>(luts, qf[u].w, qm[u].w, qc[u].w, ml.w));
                }
            }
        }
        const int tot = wave_sum(n);
        if (lane == 0) errors[v] = tot;
    }
}

// per-child error counts: tile of 1024 trios x SAMPLE_STATS_ROWS variants, SWAR byte lanes, atomics at the end
static __global__ __launch_bounds__(256) void k_mendel_children(const uint8_t *__restrict__ gt, size_t pitch, int n_variants,
                                                         int pchunks, int n_trios, MendelLuts luts,
                                                         const uint8_t *__restrict__ male_plane,
                                                         const uint8_t *__restrict__ is_x, int32_t *__restrict__ child_errors) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int c = tile * 64 + lane;
    const long v0 = (long)blockIdx.y * SAMPLE_STATS_ROWS;
    if (tile * 64 >= pchunks || c >= pchunks) return;
    const uint32_t plane = (uint32_t)pchunks * 16u;
    const int rows = (int)((v0 + SAMPLE_STATS_ROWS <= n_variants) ? SAMPLE_STATS_ROWS : (n_variants - v0));
    const uint4 ml = reinterpret_cast<const uint4 *>(male_plane)[c];
    uint32_t acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < rows; ++r) {
        const uint8_t *rowb = gt + (size_t)(v0 + r) * pitch;
        const uint4 f = load16o<true>(rowb, (uint32_t)c * 16u), m = load16o<true>(rowb, (uint32_t)c * 16u + plane);
        const uint4 k = load16o<true>(rowb, (uint32_t)c * 16u + 2u * plane);
        if (is_x != nullptr && is_x[v0 + r]) {
            acc[0] += mendel4<true>(luts, f.x, m.x, k.x, ml.x); acc[1] += mendel4<true>(luts, f.y, m.y, k.y, ml.y);
            acc[2] += mendel4<true>(luts, f.z, m.z, k.z, ml.z); acc[3] += mendel4<true>(luts, f.w, m.w, k.w, ml.w);
        } else {
            acc[0] += mendel4<false>(luts, f.x, m.x, k.x, 0); acc[1] += mendel4<false>(luts, f.y, m.y, k.y, 0);
            acc[2] += mendel4<false>(luts, f.z, m.z, k.z, 0); acc[3] += mendel4<false>(luts, f.w, m.w, k.w, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = c * 16 + q * 4 + j;
            const int n = (int)((acc[q] >> (8 * j)) & 0xFFu);
            if (n && t < n_trios) atomicAdd(child_errors + t, n);
        }
}

namespace mendel_host {
inline void build_luts(MendelLuts &out) {
    static const int G[3][2] = {{0, 0}, {0, 1}, {1, 1}};          // class representatives
    for (int x = 0; x < 2; ++x) {
        uint8_t err[16];
        for (int m = 0; m < 4; ++m)
            for (int f = 0; f < 4; ++f) {
                uint8_t mask = 0;
                if (f < 3 && m < 3)
                    for (int c = 0; c < 3; ++c)
                        if (mendel_code(x == 1, G[f][0], G[f][1], G[m][0], G[m][1], G[c][0], G[c][1])) mask |= (uint8_t)(1u << c);
                err[m * 4 + f] = mask;
            }
        auto dw = [](const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); };
        MendelLut &L = x ? out.xmale : out.autosome;
        L.lo[0] = dw(err); L.lo[1] = dw(err + 4); L.hi[0] = dw(err + 8); L.hi[1] = dw(err + 12);
    }
}
}  // namespace mendel_host

// ---------------------------------------------------------------------------
// count-derived variant filters (shared_options.c:42-47,86-115: --maf, --missing;
// predicates live in hpg-libs, absent: directions are explicit parameters here).
// keep[i] = 1 iff every enabled test passes; a negative threshold disables a test.
//   maf        = min(allele0, allele1) / (allele0 + allele1)      (0 when no allele is called)
//   missing    = missing_genotypes / n_samples
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_stats_filter(const int4 *__restrict__ in8, int n, int n_samples,
                                                      double min_maf, double max_maf, double max_missing,
                                                      uint8_t *__restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 hi = in8[2 * i + 1];                       // missing_gt, missing_alleles, allele0, allele1
    const int a0 = hi.z, a1 = hi.w, tot = a0 + a1;
    const double maf = tot > 0 ? (double)(a0 < a1 ? a0 : a1) / (double)tot : 0.0;
    const double miss = n_samples > 0 ? (double)hi.x / (double)n_samples : 0.0;
    bool ok = true;
    if (min_maf >= 0.0) ok = ok && (maf >= min_maf);
    if (max_maf >= 0.0) ok = ok && (maf <= max_maf);
    if (max_missing >= 0.0) ok = ok && (miss <= max_missing);
    keep[i] = ok ? 1 : 0;
}

// Hardy-Weinberg chi-square on (n_AA, n_Aa, n_aa) = (n_00, n_01 + n_10, n_11);
// definition: oracle/hpgv_oracle.c orc_hwe (hpg-libs body absent: unpinned)
static __global__ __launch_bounds__(256) void k_stats_hwe(const int4 *__restrict__ in8, int n, double *__restrict__ chi2,
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
