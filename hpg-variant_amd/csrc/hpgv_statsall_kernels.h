// hpgv_statsall_kernels.h -- k_stats_all2: everything the stats tool wants of a batch (get_variants_stats + get_sample_stats,
// stats_runner.c:194-198) from ONE read of the raw matrix, without a row's worth of barriers and index traffic.
//
// k_stats_all (hpgv_batch_kernels.h) stages every row in LDS, reduces every counter set across the workgroup with two
// barriers each, and gathers the trios' and the phenotype groups' columns through index tables it re-reads from memory for
// every row: ten barriers and 80 KB of indices per 10 KB row -- 254 GB/s (round 2).  Here a THREAD OWNS COLUMNS for the whole
// band of rows its workgroup scans, so everything that depends on the column only is loaded once per band and lives in
// registers:
//   * its CPT 16-byte chunks of the row come straight from memory (coalesced, the next row's loads issued before this row
//     is counted); pair classes and flags as in k_stats_all (16-entry v_perm_b32 tables, four genotypes per instruction);
//   * phenotype groups are MASKS over the thread's own columns (a group id per column, turned into byte masks once per
//     band): a group's counters are masked popcounts of the flags the thread has anyway -- no gather;
//   * per-sample missing counts are SWAR byte counters in registers over the band (rows <= 255), as k_sample_missing keeps
//     them, flushed once per band;
//   * Mendelian errors need three arbitrary columns per trio: the row's pair classes go to LDS (double-buffered: ONE barrier
//     per row), and every thread gathers the 16 trios it owns with column indices it keeps packed in registers;
//   * a row's counters are reduced inside each wave (DPP) and added to the band's per-row counters in LDS by one lane; the
//     records (counters, Hardy-Weinberg) are computed for all rows of the band at its end.
#pragma once
#include "hpgv_batch_kernels.h"
#include <type_traits>

namespace hpgv {

// bytes equal to g -> 0xFF, others 0x00 (exact, byte by byte)
__device__ __forceinline__ uint32_t byte_eq_mask(uint32_t w, uint32_t g) {
    const uint32_t x = w ^ (g * 0x01010101u);
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 where the byte of x is zero
    return z | (z - (z >> 7));
}

// Positional popcount by a bit-matrix transpose.  Eight dwords of one-hot flag bytes are, byte lane by byte lane, 8 x 8 bit
// matrices (row j = dword j, column b = flag bit b).  Three butterfly stages (4-, 2-, 1-bit blocks; shift + v_bfi_b32 each)
// transpose them: afterwards dword b holds flag bit b of all 32 genotypes -- bit j of byte lane k is bit b of input dword j,
// byte k -- so ONE popcount per flag bit counts 32 genotypes, and a phenotype group is ONE mask word over the transposed
// layout (which genotypes of the thread are in the group) instead of a mask per dword: 48 + 8 (1 + groups) instructions per
// 32 genotypes against 20 per dword and group before.
__device__ __forceinline__ void bit_transpose8(uint32_t (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                  // rows j, j + 4: swap the off-diagonal 4 x 4 blocks
        const uint32_t lo = x[j], hi = x[j + 4];
        x[j] = (lo & 0x0F0F0F0Fu) | ((hi << 4) & 0xF0F0F0F0u);
        x[j + 4] = ((lo >> 4) & 0x0F0F0F0Fu) | (hi & 0xF0F0F0F0u);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {                              // rows j, j + 2 inside each group of four
            const uint32_t lo = x[4 * q + j], hi = x[4 * q + j + 2];
            x[4 * q + j] = (lo & 0x33333333u) | ((hi << 2) & 0xCCCCCCCCu);
            x[4 * q + j + 2] = ((lo >> 2) & 0x33333333u) | (hi & 0xCCCCCCCCu);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                  // neighbouring rows
        const uint32_t lo = x[2 * q], hi = x[2 * q + 1];
        x[2 * q] = (lo & 0x55555555u) | ((hi << 1) & 0xAAAAAAAAu);
        x[2 * q + 1] = ((lo >> 1) & 0x55555555u) | (hi & 0xAAAAAAAAu);
    }
}
// the mask of a byte-granular selection (0xFF / 0x00 per byte of dword j) in the transposed layout
__device__ __forceinline__ uint32_t transposed_mask(const uint32_t (&m)[8]) {
    uint32_t t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t |= (m[j] & 0x01010101u) << j;
    return t;
}

// Sixteen per-lane values -> the sums over each row of 16 lanes, one value per lane, by halving: a lane hands the half it does
// not keep to its partner (lane ^ 1, ^ 2, ^ 8 by DPP) and adds the half it gets; a symmetric step adds the neighbouring quad.
// Lane l of every row ends with the row's sum of slot 8 i + 4 bit3(l) + 2 bit1(l) + bit0(l) in out[i]; lanes l and l ^ 4
// hold the same.  14 select-and-add steps + 2 instead of 16 full reductions.
__device__ __forceinline__ void row_reduce16(const uint32_t (&v)[16], uint32_t (&out)[2], int lane) {
    const bool s0 = lane & 1, s1 = lane & 2, s3 = lane & 8;
    uint32_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t keep = s0 ? v[2 * i + 1] : v[2 * i], give = s0 ? v[2 * i] : v[2 * i + 1];
        a[i] = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t keep = s1 ? a[2 * i + 1] : a[2 * i], give = s1 ? a[2 * i] : a[2 * i + 1];
        uint32_t t = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        // the neighbouring quad (row_ror:4) holds the same slots: a symmetric add, BEFORE bit 3 becomes a selector (afterwards
        // lanes l and l ^ 8 hold different slots and a rotation would mix them); with the ^ 8 step below the four quads are covered
        t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x124, 0xF, 0xF, true);
        b[i] = t;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t keep = s3 ? b[2 * i + 1] : b[2 * i], give = s3 ? b[2 * i] : b[2 * i + 1];
        out[i] = keep + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0x128, 0xF, 0xF, true);       // row_ror:8 = lane ^ 8
    }
}

struct StatsAll2Cfg {
    const uint8_t *group_of_col;   // [chunks * 16] group id per column, 0xFF = in no group (or no groups at all: null)
    int n_masked;                  // groups counted with masks (the template's NM)
    int derive_last;               // every column is in a group: group n_masked is "all minus the masked ones"
    uint32_t *row_counters;        // [n_variants][STATS2_W] device scratch: the rows' packed counters
};
enum { STATS2_W = 18,              // per row: 16 slots (set s, flag pair j at 4 s + j: two 16-bit counts each), Mendel errors, spare
       STATS2_CLS = 16384 + 64 };  // bytes between the two class buffers (a compile-time LDS offset)

// CPT: 16-byte chunks per thread, thread t owns chunks t, t + blockDim, ...; NM: masked group sets (0..3); MENDEL: the trios'
// errors are wanted.  blockDim.x: a multiple of 64, at most 512, at least n_trios / 16 with MENDEL.
template <int CPT, int NM, bool MENDEL>
__global__ __launch_bounds__(512) void k_stats_all2(StatsAllArgs A, StatsAll2Cfg G) {
    extern __shared__ __align__(16) uint8_t lds[];                 // class buffer 0 | class buffer 1 | row counters | band events
    constexpr int SETS = 1 + NM, W = STATS2_W, NB = (CPT + 1) / 2;  // NB blocks of eight dwords per thread
    const int tid = threadIdx.x, lane = tid & 63, BS = blockDim.x;
    const int chunks = (A.n_samples + 15) >> 4;
    uint32_t *rowcnt = reinterpret_cast<uint32_t *>(lds + 2 * STATS2_CLS);
    uint32_t *band_events = rowcnt + (size_t)A.rows_per_block * W;  // [2]: missing genotypes, child errors of the band
    const int v0 = blockIdx.x * A.rows_per_block;
    const int rows = (v0 + A.rows_per_block <= A.n_variants) ? A.rows_per_block : A.n_variants - v0;
    const bool want_sm = A.sample_missing != nullptr, want_ce = A.child_errors != nullptr;

    for (int i = tid; i < rows * W + 2; i += BS) rowcnt[i] = 0u;
    // the pad slot the trios' pads point at reads "not called" (Mendel class 3)
    if (MENDEL && tid < 32) lds[(tid >> 4) * STATS2_CLS + chunks * 16 + (tid & 15)] = 3;

    // ---- what depends on the thread's columns only -------------------------------------------------------------------
    // only the row's last chunk can hold columns that do not exist: one mask (vlast) for the chunk of this thread that is the
    // last one (ilast; none: -1).  Chunks past the row's end are not loaded but filled with 0x22 bytes -- allele 2 / allele 2,
    // the pair class without a stats flag -- so they count nothing.
    uint32_t vlast[4], tmask[NB][NM > 0 ? NM : 1];
    bool own[CPT];
    int ilast = -1;
    {
        uint32_t gm[NB][NM > 0 ? NM : 1][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int left = A.n_samples - ((chunks - 1) * 16 + k * 4);
            vlast[k] = left <= 0 ? 0u : (left >= 4 ? 0xFFFFFFFFu : ((1u << (8 * left)) - 1u));
        }
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) {
            const int c = tid + BS * i;
            const bool mine = i < CPT && c < chunks;
            if (i < CPT) own[i] = mine;
            if (mine && c == chunks - 1) ilast = i;
            uint4 gq = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (NM > 0 && mine) gq = reinterpret_cast<const uint4 *>(G.group_of_col)[c];
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int s = 0; s < NM; ++s) gm[i / 2][s][(i & 1) * 4 + k] = mine ? byte_eq_mask(gw[k], (uint32_t)s) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int s = 0; s < NM; ++s) tmask[q][s] = transposed_mask(gm[q][s]);
    }
    // the (up to) 16 trios this thread owns -- trios tid, tid + BS, tid + 2 BS, ...: neighbouring lanes read neighbouring trios,
    // whose members are neighbouring columns in the usual pedigree order, so a wave's byte gathers fall into few LDS words
    // (trios 16 t .. 16 t + 15 per thread: lanes 48 bytes apart, four-way bank conflicts, two thirds of the LDS cycles) --
    // column of father / mother / child, two per register; a pad points at the pad slot
    uint32_t tf[8], tm[8], tc[8];
    uint32_t male[4] = {0u, 0u, 0u, 0u};
    const int trio_groups = MENDEL ? (((A.n_trios + BS - 1) / BS) + 3) / 4 : 0;      // groups of four trios per thread that hold any
    if (MENDEL) {
        const uint32_t pad = (uint32_t)(chunks * 16);
        const size_t plane = (size_t)A.pchunks * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int id = j * BS + tid;
            uint32_t cf = pad, cm = pad, cc = pad, ml = 0u;
            if (id < A.n_trios) {
                const int32_t a = A.mendel_cols[id], b = A.mendel_cols[plane + id], c2 = A.mendel_cols[2 * plane + id];
                cf = a < 0 ? pad : (uint32_t)a; cm = b < 0 ? pad : (uint32_t)b; cc = c2 < 0 ? pad : (uint32_t)c2;
                if (A.is_x) ml = A.male_plane[id];
            }
            if (j & 1) { tf[j / 2] |= cf << 16; tm[j / 2] |= cm << 16; tc[j / 2] |= cc << 16; }
            else { tf[j / 2] = cf; tm[j / 2] = cm; tc[j / 2] = cc; }
            male[j / 4] |= ml << (8 * (j & 3));
        }
    }
    uint32_t miss[CPT][4], trio[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) miss[i][k] = 0u;
    uint32_t ev_missing = 0u, ev_errors = 0u;

    // ---- the band's rows ----------------------------------------------------------------------------------------------
    uint4 cur[CPT], nxt[CPT];
    const uint8_t *rowp = A.src + (size_t)v0 * A.src_pitch;
#pragma unroll
    for (int i = 0; i < CPT; ++i) cur[i] = own[i] ? reinterpret_cast<const uint4 *>(rowp)[tid + BS * i] : make_uint4(0x22222222u, 0x22222222u, 0x22222222u, 0x22222222u);
    __syncthreads();                                               // the counters are zero, the pad slots set

    // one row; CB = the row's class buffer (a compile-time LDS offset: the gathers' addresses are index + constant)
    auto do_row = [&](const int r, auto cb_tag) {
        constexpr uint32_t CB = decltype(cb_tag)::value;
        // the next row's chunks: without trios a whole row ahead, into registers of their own; with trios into the registers this
        // row's chunks leave free after they are counted -- the reduction, the barrier and the gathers cover the loads' latency,
        // and eight or twelve registers fewer are another workgroup per unit
        const uint8_t *np = rowp + (size_t)(r + 1) * A.src_pitch;
        if (!MENDEL && r + 1 < rows) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) nxt[i] = own[i] ? reinterpret_cast<const uint4 *>(np)[tid + BS * i] : make_uint4(0x22222222u, 0x22222222u, 0x22222222u, 0x22222222u);
        }
        const bool x_row = MENDEL && (A.is_x != nullptr) && (A.is_x[v0 + r] != 0);
        uint32_t cnt[SETS][8];
#pragma unroll
        for (int s = 0; s < SETS; ++s)
#pragma unroll
            for (int b = 0; b < 8; ++b) cnt[s][b] = 0u;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            uint32_t f[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * q + h;
                if (i >= CPT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[4 * h + k] = 0u;
                    continue;
                }
                const uint32_t g[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
                uint32_t mc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t idx = pair_class4(g[k]);
                    f[4 * h + k] = flags_of_class4(idx) & (i == ilast ? vlast[k] : 0xFFFFFFFFu);
                    if (MENDEL) mc[k] = mendel_of_class4(idx);
                    if (want_sm) miss[i][k] += (f[4 * h + k] >> 4) & 0x01010101u;      // flag bit 4: some allele missing
                }
                if (MENDEL && own[i]) *reinterpret_cast<uint4 *>(lds + CB + 16 * (tid + BS * i)) = make_uint4(mc[0], mc[1], mc[2], mc[3]);
            }
            bit_transpose8(f);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                cnt[0][b] += (uint32_t)__builtin_popcount(f[b]);
#pragma unroll
                for (int s = 0; s < NM; ++s) cnt[1 + s][b] += (uint32_t)__builtin_popcount(f[b] & tmask[q][s]);
            }
            __builtin_amdgcn_sched_barrier(0);                     // one block of eight dwords at a time: interleaved blocks double the temporaries
        }
        if (MENDEL && r + 1 < rows) {
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                if (own[i]) cur[i] = reinterpret_cast<const uint4 *>(np)[tid + BS * i];
        }
        ev_missing += cnt[0][4];
        uint32_t v16[16], red[2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) v16[4 * s + j] = s < SETS ? (cnt[s][2 * j] | (cnt[s][2 * j + 1] << 16)) : 0u;
        row_reduce16(v16, red, lane);
        uint32_t *rc = rowcnt + (size_t)r * W;
        if (!(lane & 4)) {
            const int slot = ((lane >> 1) & 4) | (lane & 3);       // 4 bit3 + 2 bit1 + bit0
            if (slot < 4 * SETS && red[0]) atomicAdd(&rc[slot], red[0]);
            if (8 + slot < 4 * SETS && red[1]) atomicAdd(&rc[8 + slot], red[1]);
        }
        if (MENDEL) {
            __syncthreads();                                       // the row's Mendel classes are in its buffer
            int n = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < trio_groups) {                             // (wave-uniform: the same for every thread)
                    auto four = [&](const uint32_t (&t)[8]) {
                        // the unpacking is asm volatile so that it STAYS here: as plain C it is loop-invariant, the compiler hoists the
                        // 48 unpacked indices out of the row loop, and those registers cost a workgroup per unit
                        uint32_t i0, i1, i2, i3;
                        asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(i0) : "v"(t[2 * k]));
                        asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(i1) : "v"(t[2 * k]));
                        asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(i2) : "v"(t[2 * k + 1]));
                        asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(i3) : "v"(t[2 * k + 1]));
                        return (uint32_t)lds[CB + i0] | ((uint32_t)lds[CB + i1] << 8) | ((uint32_t)lds[CB + i2] << 16) | ((uint32_t)lds[CB + i3] << 24);
                    };
                    const uint32_t ff = four(tf), mm = four(tm), cc = four(tc);
                    const uint32_t e = x_row ? mendel4<true>(A.luts, ff, mm, cc, male[k]) : mendel4<false>(A.luts, ff, mm, cc, 0);
                    n += __builtin_popcount(e);
                    trio[k] += e;
                }
                __builtin_amdgcn_sched_barrier(0);                 // twelve byte reads in flight, not forty-eight: their registers decide the waves per unit
            }
            ev_errors += (uint32_t)n;
            const int tot = wave_sum(n);
            if (lane == 0 && tot) atomicAdd(&rc[16], (uint32_t)tot);
        }
        if (!MENDEL) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) cur[i] = nxt[i];
        }
    };
    for (int r = 0; r < rows; r += 2) {
        do_row(r, std::integral_constant<uint32_t, 0u>{});
        if (r + 1 < rows) do_row(r + 1, std::integral_constant<uint32_t, (uint32_t)STATS2_CLS>{});
    }

    // ---- the band's counters leave ---------------------------------------------------------------------------------------
    if (want_sm) { const int s = wave_sum((int)ev_missing); if (lane == 0 && s) atomicAdd(&band_events[0], (uint32_t)s); }
    if (want_ce) { const int s = wave_sum((int)ev_errors); if (lane == 0 && s) atomicAdd(&band_events[1], (uint32_t)s); }
    __syncthreads();                                               // every row's counters are complete; the class buffers are free
    // the rows' counters as they are (k_stats_all2_records turns them into records: its double arithmetic would cost this kernel
    // half its waves in registers)
    for (int i = tid; i < rows * W; i += BS) G.row_counters[(size_t)v0 * W + i] = rowcnt[i];
    if (!(want_sm || want_ce)) return;
    // the threads' byte counters -> LDS in column order, then to the device totals: coalesced full-wave atomics over ALL columns
    // when the band saw many events (a dense wave atomic costs what one scattered lane costs), only the non-zero ones otherwise
    if (want_sm)
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (own[i]) *reinterpret_cast<uint4 *>(lds + 16 * (tid + BS * i)) = make_uint4(miss[i][0], miss[i][1], miss[i][2], miss[i][3]);
    if (MENDEL && want_ce)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int id = j * BS + tid;
            if (id < A.n_trios) lds[STATS2_CLS + id] = (uint8_t)(trio[j / 4] >> (8 * (j & 3)));
        }
    __syncthreads();
    if (want_sm && band_events[0] > 0) {
        const bool dense = (int)band_events[0] * 64 >= A.n_samples;
        for (int d = tid; d < A.n_samples; d += BS) {
            const int val = lds[d];
            if (dense || val) atomicAdd(A.sample_missing + d, val);
        }
    }
    if (MENDEL && want_ce && band_events[1] > 0) {
        const bool dense = (int)band_events[1] * 64 >= A.n_trios;
        for (int d = tid; d < A.n_trios; d += BS) {
            const int val = lds[STATS2_CLS + d];
            if (dense || val) atomicAdd(A.child_errors + d, val);
        }
    }
}

// the rows' packed counters -> records: one thread per (variant, counter set); set 0 = all samples, 1.. = the masked groups,
// 1 + n_masked = the derived last group
static __global__ __launch_bounds__(256) void k_stats_all2_records(const uint32_t *__restrict__ row_counters, int n_variants, int n_masked,
                                                             int derive_last, BatchStatsRec *__restrict__ out,
                                                             BatchStatsRec *__restrict__ group_out, int32_t *__restrict__ mendel_errors) {
    const int W = STATS2_W, sets = 1 + (group_out ? n_masked + derive_last : 0);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)n_variants * sets) return;
    const int s = (int)(t / n_variants);
    const size_t v = (size_t)(t % n_variants);
    const uint32_t *rc = row_counters + v * (size_t)W;
    int c[8];
    auto unpack = [&](int set, int sign) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[2 * j] += sign * (int)(rc[4 * set + j] & 0xFFFFu); c[2 * j + 1] += sign * (int)(rc[4 * set + j] >> 16); }
    };
#pragma unroll
    for (int b = 0; b < 8; ++b) c[b] = 0;
    if (s <= n_masked) unpack(s, 1);
    else { unpack(0, 1); for (int k = 1; k <= n_masked; ++k) unpack(k, -1); }
    if (s == 0) {
        out[v] = stats_record(c);
        if (mendel_errors) mendel_errors[v] = (int32_t)rc[16];
    } else group_out[(size_t)(s - 1) * (size_t)n_variants + v] = stats_record(c);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_assoc_rows: the allele counts of assoc_count_individual (assoc.c:87-134) from the tokenizer's raw rows (VCF column order,
// not strict), threads owning columns across a band of rows as in k_stats_all2: the affected / unaffected columns are two
// masks in the transposed layout, a genotype's contribution a 3-bit flag of its pair class -- "0/0", exactly one zero allele,
// both alleles non-zero; anything with a missing allele counts nothing, which is the strict rule of assoc.c:53 -- and
//     A1 = 2 n00 + nhet, A2 = 2 nxx + nhet   (chromosome "X", assoc.c:94-107: A1 = n00, A2 = nxx)
// per phenotype group.  No LDS staging, no barrier per row, no column gather (k_batch<BATCH_CHISQ> on the same rows: one
// workgroup per row, layout gathered through col_of_pos: 110 us per 16 000 x 10 k).  The counts go to an int4 array; the
// statistics are the scans' own kernels (k_assoc_chisq / k_assoc_fisher).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t assoc_flags_of_class4(uint32_t idx) { return lut16x4(0x00020201u, 0x00040402u, 0x00040402u, 0x00000000u, idx); }

template <int CPT>
__global__ __launch_bounds__(512) void k_assoc_rows(const uint8_t *__restrict__ src, size_t src_pitch, int n_variants, int n_samples, int rows_per_block,
                                                     const uint8_t *__restrict__ is_x, const uint8_t *__restrict__ cond /* per column: 1 affected, 0 unaffected, else neither; padded */,
                                                     int4 *__restrict__ counts) {
    extern __shared__ __align__(16) uint8_t lds[];
    constexpr int NB = (CPT + 1) / 2;
    uint32_t *rowcnt = reinterpret_cast<uint32_t *>(lds);                 // [rows][4]: n00A | nhetA << 16, nxxA | n00U << 16, nhetU | nxxU << 16, -
    const int tid = threadIdx.x, lane = tid & 63, BS = blockDim.x;
    const int chunks = (n_samples + 15) >> 4;
    const int v0 = blockIdx.x * rows_per_block;
    const int rows = (v0 + rows_per_block <= n_variants) ? rows_per_block : n_variants - v0;
    for (int i = tid; i < rows * 4; i += BS) rowcnt[i] = 0u;
    uint32_t tA[NB], tU[NB];
    bool own[CPT];
    {
        uint32_t mA[NB][8], mU[NB][8];
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) {
            const int c = tid + BS * i;
            const bool mine = i < CPT && c < chunks;
            if (i < CPT) own[i] = mine;
            uint4 cq = make_uint4(0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u);
            if (mine) cq = reinterpret_cast<const uint4 *>(cond)[c];
            const uint32_t cw[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { mA[i / 2][(i & 1) * 4 + k] = byte_eq_mask(cw[k], 1u); mU[i / 2][(i & 1) * 4 + k] = byte_eq_mask(cw[k], 0u); }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) { tA[q] = transposed_mask(mA[q]); tU[q] = transposed_mask(mU[q]); }
    }
    uint4 cur[CPT], nxt[CPT];
    const uint8_t *rowp = src + (size_t)v0 * src_pitch;
#pragma unroll
    for (int i = 0; i < CPT; ++i) cur[i] = own[i] ? reinterpret_cast<const uint4 *>(rowp)[tid + BS * i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
        if (r + 1 < rows) {
            const uint8_t *np = rowp + (size_t)(r + 1) * src_pitch;
#pragma unroll
            for (int i = 0; i < CPT; ++i) nxt[i] = own[i] ? reinterpret_cast<const uint4 *>(np)[tid + BS * i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        uint32_t cA[3] = {0u, 0u, 0u}, cU[3] = {0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            uint32_t f[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * q + h;
                if (i >= CPT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) f[4 * h + k] = 0u;
                    continue;
                }
                const uint32_t g[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) f[4 * h + k] = assoc_flags_of_class4(pair_class4(g[k]));
            }
            bit_transpose8(f);                                       // (only rows 0 - 2 are used: the rest of the network is dead code)
#pragma unroll
            for (int b = 0; b < 3; ++b) { cA[b] += (uint32_t)__builtin_popcount(f[b] & tA[q]); cU[b] += (uint32_t)__builtin_popcount(f[b] & tU[q]); }
        }
        uint32_t v16[16], red[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) v16[j] = 0u;
        v16[0] = cA[0] | (cA[1] << 16); v16[1] = cA[2] | (cU[0] << 16); v16[2] = cU[1] | (cU[2] << 16);
        row_reduce16(v16, red, lane);
        if (!(lane & 4)) {
            const int slot = ((lane >> 1) & 4) | (lane & 3);
            if (slot < 3 && red[0]) atomicAdd(&rowcnt[r * 4 + slot], red[0]);
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) cur[i] = nxt[i];
    }
    __syncthreads();
    for (int r = tid; r < rows; r += BS) {
        const uint32_t a = rowcnt[r * 4], b = rowcnt[r * 4 + 1], c = rowcnt[r * 4 + 2];
        const int n00A = (int)(a & 0xFFFFu), nhetA = (int)(a >> 16), nxxA = (int)(b & 0xFFFFu);
        const int n00U = (int)(b >> 16), nhetU = (int)(c & 0xFFFFu), nxxU = (int)(c >> 16);
        const bool x = is_x != nullptr && is_x[v0 + r] != 0;
        int4 o;
        o.x = x ? n00A : 2 * n00A + nhetA; o.y = x ? nxxA : 2 * nxxA + nhetA;
        o.z = x ? n00U : 2 * n00U + nhetU; o.w = x ? nxxU : 2 * nxxU + nhetU;
        counts[v0 + r] = o;
    }
}

// LDS bytes of a launch
inline size_t stats_all2_lds(int rows_per_block) {
    return 2 * (size_t)STATS2_CLS + ((size_t)rows_per_block * STATS2_W + 2) * sizeof(uint32_t) + 16;
}

}  // namespace hpgv
