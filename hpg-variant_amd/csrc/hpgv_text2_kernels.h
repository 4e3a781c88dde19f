// hpgv_text2_kernels.h -- tile-parallel VCF tokenizer (round 2): two sweeps of the text instead of three, and no
// workgroup walks a line tile by tile (k_tok_parse of hpgv_text_kernels.h does: one memory round trip per 4 KiB of a line).
//
//   k_tok_count2 : per 4 KiB tile its newline count, the TABs after its last newline (all its TABs when it has none) and
//                  where that last newline is;
//   k_tok_scan2a/b: the state at every tile's start -- lines before it, TABs since the current line began, where that line
//                  began -- as a two-level scan (1024 tiles per workgroup, then the workgroups' totals), and the line count;
//   k_tok_parse2 : one workgroup per TILE.  A segmented block scan gives every thread the (line, TABs so far, line start)
//                  at its 16 bytes; a second one the GT position of its line (defined by whoever holds the line's 8th
//                  TAB, i.e. the start of FORMAT).  A tile in the middle of a line whose FORMAT lies in an earlier tile
//                  ASSUMES GT is the first FORMAT key (the VCF specification requires it when GT is present); the thread
//                  that does see a FORMAT with GT elsewhere or absent flags its line, and k_tok_parse (the line-by-line
//                  kernel) re-does exactly the flagged lines afterwards.  Then every thread walks its TABs and newlines: a TAB from the ninth on starts
//                  a sample field and is encoded out of the thread's registers, a newline closes its line (status, the
//                  0xFF tail of a short row), the first nine TABs give CHROM .. FORMAT, the first one chromosome "X".
// Same outputs as k_tok_count / k_tok_scan / k_tok_mark / k_tok_parse, bit for bit (tests/test_gpu_text.py runs both).
// Reference: the per-genotype strdup + get_alleles of assoc.c:45-56 / tdt.c:97-108,150-157 (what is being replaced).
#pragma once
#include "hpgv_text_kernels.h"

namespace hpgv {

constexpr int TOK2_TB = 32;                                          // bytes per thread: 32 TAB / newline bits in one register
constexpr int TOK2_NW = TOK2_TB / 8 + 1;                             // 8-byte words a thread holds: its bytes and the 8 after them
constexpr int TOK2_TILE = 256 * TOK2_TB;                             // bytes per workgroup tile (8 KiB)
struct TokAgg { int nl, tabs, last_nl, pad; };                       // last_nl: offset inside the tile, -1 when none
struct TokPre { int lines, tabs; unsigned long long line_start; };   // state at the tile's first byte
constexpr int TOK_GT_UNDEF = -2;                                     // "FORMAT of this line not seen yet"

// TAB and newline bits of a thread's TOK2_TB bytes (bit j = byte base + j), and the bytes themselves (+ the 8 after them) when
// they all exist
__device__ __forceinline__ void tok_masks(const char *__restrict__ t, size_t base, size_t n, uint32_t *tabs, uint32_t *nls,
                                          uint64_t (&w)[TOK2_NW], bool *wide) {
    *wide = base + TOK2_TB + 8 <= n;
    uint32_t a = 0, b = 0;
    if (*wide) {
#pragma unroll
        for (int k = 0; k < TOK2_NW; ++k) __builtin_memcpy(&w[k], t + base + 8 * k, 8);
#pragma unroll
        for (int k = 0; k < TOK2_TB / 8; ++k) { a |= tok_byte_mask(w[k], '\t') << (8 * k); b |= tok_byte_mask(w[k], '\n') << (8 * k); }
    } else {
#pragma unroll
        for (int k = 0; k < TOK2_NW; ++k) w[k] = 0;
        for (int j = 0; j < TOK2_TB; ++j)
            if (base + j < n) { const char c = t[base + j]; if (c == '\t') a |= 1u << j; else if (c == '\n') b |= 1u << j; }
    }
    *tabs = a; *nls = b;
}
// TABs behind bit `last` (the thread's last newline): a shift by 32 is not a shift
__device__ __forceinline__ int tok_tabs_after(uint32_t tabs, int last) { return last >= 31 ? 0 : __popc(tabs >> (last + 1)); }

static __global__ __launch_bounds__(256) void k_tok_count2(const char *__restrict__ text, size_t n, TokAgg *__restrict__ agg) {
    __shared__ int s_nl[4], s_last[4], s_tabs[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t base = (size_t)blockIdx.x * TOK2_TILE + (size_t)tid * TOK2_TB;
    uint32_t tabs, nls; uint64_t ww[TOK2_NW]; bool wide;
    tok_masks(text, base, n, &tabs, &nls, ww, &wide);
    const int nl = __popc(nls);
    const int last_bit = nl ? 31 - __clz((int)nls) : -1;
    const int tabs_after = nl ? tok_tabs_after(tabs, last_bit) : __popc(tabs);
    // the last thread of the workgroup that holds a newline
    int c = nl, key = nl ? tid : -1;
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off); const int k2 = __shfl_xor(key, off); key = k2 > key ? k2 : key; }
    if (lane == 0) { s_nl[w] = c; s_last[w] = key; }
    __syncthreads();
    const int tlast = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    int mine = tid > tlast ? __popc(tabs) : (tid == tlast ? tabs_after : 0);
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) s_tabs[w] = mine;
    __syncthreads();
    if (tid == (tlast < 0 ? 0 : tlast)) {                             // one thread writes the whole record
        TokAgg a;
        a.nl = s_nl[0] + s_nl[1] + s_nl[2] + s_nl[3];
        a.tabs = s_tabs[0] + s_tabs[1] + s_tabs[2] + s_tabs[3];
        a.last_nl = tlast < 0 ? -1 : tid * TOK2_TB + last_bit;
        a.pad = 0;
        agg[blockIdx.x] = a;
    }
}

// The state at every tile's start.  The combine is associative but not commutative (a newline resets the TAB count and
// moves the line start): (lines, has_nl, tabs, line_start) o (...) as in tok_fold below.  Two levels: k_tok_scan2a scans
// 1024 tiles per workgroup (one tile per thread: wave scans by shuffles, the 16 wave totals folded through LDS) and writes
// every tile's state RELATIVE to its workgroup plus the workgroup's total; k_tok_scan2b folds the workgroups' totals (one
// thread: there are n_tiles / 1024 of them), counts the lines and makes every tile's state absolute in place.
struct TokState { int lines, tabs; long long ls; };                 // ls < 0: no newline so far (tabs add up)
__device__ __forceinline__ TokState tok_fold(TokState a, TokState b) {       // a happened before b
    TokState r;
    r.lines = a.lines + b.lines;
    if (b.ls >= 0) { r.tabs = b.tabs; r.ls = b.ls; } else { r.tabs = a.tabs + b.tabs; r.ls = a.ls; }
    return r;
}

static __global__ __launch_bounds__(TOK_SCAN_THREADS) void k_tok_scan2a(const TokAgg *__restrict__ agg, int n_tiles, TokPre *__restrict__ pre,
                                                                      TokState *__restrict__ group_total) {
    __shared__ int w_lines[16], w_tabs[16];
    __shared__ long long w_ls[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x * TOK_SCAN_THREADS + tid;
    TokState me = {0, 0, -1};
    if (i < n_tiles) {
        const TokAgg a = agg[i];
        me.lines = a.nl; me.tabs = a.tabs; me.ls = a.last_nl >= 0 ? (long long)i * TOK2_TILE + a.last_nl + 1 : -1;
    }
    TokState inc = me;                                               // inclusive scan within the wave
    for (int off = 1; off < 64; off <<= 1) {
        TokState o; o.lines = __shfl_up(inc.lines, off); o.tabs = __shfl_up(inc.tabs, off); o.ls = __shfl_up(inc.ls, off);
        if (lane >= off) inc = tok_fold(o, inc);
    }
    if (lane == 63) { w_lines[w] = inc.lines; w_tabs[w] = inc.tabs; w_ls[w] = inc.ls; }
    TokState exc; exc.lines = __shfl_up(inc.lines, 1); exc.tabs = __shfl_up(inc.tabs, 1); exc.ls = __shfl_up(inc.ls, 1);
    if (lane == 0) { exc.lines = 0; exc.tabs = 0; exc.ls = -1; }
    __syncthreads();
    TokState before = {0, 0, -1};
    for (int k = 0; k < w; ++k) { TokState o = {w_lines[k], w_tabs[k], w_ls[k]}; before = tok_fold(before, o); }
    const TokState st = tok_fold(before, exc);
    if (i < n_tiles) { TokPre p; p.lines = st.lines; p.tabs = st.tabs; p.line_start = (unsigned long long)st.ls; pre[i] = p; }
    if (tid == TOK_SCAN_THREADS - 1) group_total[blockIdx.x] = tok_fold(st, me);
}

static __global__ __launch_bounds__(TOK_SCAN_THREADS) void k_tok_scan2b(TokPre *__restrict__ pre, int n_tiles, TokState *__restrict__ group_total,
                                                                      int n_groups, const char *__restrict__ text, size_t n,
                                                                      int *__restrict__ n_lines, unsigned long long *__restrict__ line_off, int max_lines) {
    // grid-stride over the groups' totals would need a second scan; n_groups = n_tiles / 1024 is small (150 for 640 MB of text),
    // so thread 0 of workgroup 0 folds them, and every workgroup then fixes up its own group
    __shared__ TokState base;
    if (threadIdx.x == 0) {
        TokState c = {0, 0, 0};
        for (int k = 0; k < (int)blockIdx.x && k < n_groups; ++k) c = tok_fold(c, group_total[k]);
        base = c;
        if (blockIdx.x == 0) {
            TokState all = c;
            for (int k = 0; k < n_groups; ++k) all = tok_fold(all, group_total[k]);
            const int tail = (n > 0 && text[n - 1] != '\n') ? 1 : 0;      // unterminated last line
            *n_lines = all.lines + tail;
            if (tail && all.lines + 1 <= max_lines) line_off[all.lines + 1] = n;
            if (n == 0) line_off[0] = 0;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * TOK_SCAN_THREADS + threadIdx.x;
    if (i < n_tiles) {
        const TokPre q = pre[i];
        TokState rel = {q.lines, q.tabs, (long long)q.line_start};
        const TokState st = tok_fold(base, rel);
        TokPre p; p.lines = st.lines; p.tabs = st.tabs; p.line_start = (unsigned long long)st.ls;
        pre[i] = p;
    }
}

// GT position inside a FORMAT field that starts at p (ends at the next TAB / newline / end of text): -1 when absent
__device__ __forceinline__ int tok_format_gtpos(const char *__restrict__ t, size_t p, size_t n) {
    int pos = 0;
    for (;;) {
        size_t q = p;
        while (q < n && t[q] != ':' && t[q] != '\t' && t[q] != '\n') q++;
        if (q - p == 2 && t[p] == 'G' && t[p + 1] == 'T') return pos;
        if (q >= n || t[q] != ':') return -1;
        p = q + 1; pos++;
    }
}

// tok_encode for a field whose line end is not known: the field ends at the next TAB, newline or end of text
__device__ __forceinline__ uint32_t tok_encode_open(const char *__restrict__ t, size_t p, size_t n, int gt_position, int strict) {
    size_t fe = p;
    while (fe < n && t[fe] != '\t' && t[fe] != '\n') fe++;
    return tok_encode(t, p, fe, gt_position, strict);
}

// what a line's end settles: chromosome "X" of a line without TABs, the unset field offsets, the 0xFF tail of the row, status
__device__ __forceinline__ void tok_close_line(const char *__restrict__ t, int line, int ntab, int gtpos, size_t ls, size_t end,
                                               int n_samples, uint8_t *__restrict__ gt, size_t pitch, uint8_t *__restrict__ is_x,
                                               uint32_t *__restrict__ field_off, int *__restrict__ status) {
    if (ntab == 0 && is_x) { const size_t clen = end - ls; is_x[line] = (clen == 0 || (clen == 1 && t[ls] == 'X')) ? 1 : 0; }
    if (field_off) for (int k = ntab + 1; k <= 9; ++k) field_off[(size_t)line * 10 + k] = 0xFFFFFFFFu;
    const int found = (ntab >= 9 && gtpos >= 0) ? (ntab - 8 < n_samples ? ntab - 8 : n_samples) : 0;
    uint8_t *row = gt + (size_t)line * pitch;
    for (int s2 = found; s2 < n_samples; ++s2) row[s2] = 0xFF;
    if (status) status[line] = ntab < 9 ? 1 : (gtpos < 0 ? 2 : (ntab - 8 < n_samples ? 3 : 0));
}

// one lane-shift of a wave scan as a DPP move (one instruction per value; a lane without a source keeps `identity`):
// row_shr:1/2/4/8 inside the rows of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; wave_shr:1 for
// the exclusive value
template <int CTRL, int ROWS> __device__ __forceinline__ int tok_dpp(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWS, 0xF, false);
}
#define TOK_SCAN_STEPS(STEP) STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)

static __global__ __launch_bounds__(256) void k_tok_parse2(const char *__restrict__ text, size_t n, const TokPre *__restrict__ pre,
                                                    int max_lines, int n_samples, int strict, uint8_t *__restrict__ gt, size_t pitch,
                                                    uint8_t *__restrict__ is_x, unsigned long long *__restrict__ line_off,
                                                    uint32_t *__restrict__ field_off, int *__restrict__ status,
                                                    int *__restrict__ redo /* per line: 1 = parse again line by line */) {
    __shared__ int s_f[4], s_v[4], s_n[4], s_d[4], s_g[4];
    __shared__ unsigned long long s_p[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const char *t = text;
    const size_t base = (size_t)blockIdx.x * TOK2_TILE + (size_t)tid * TOK2_TB;
    const TokPre P = pre[blockIdx.x];
    uint32_t tabs, nls; uint64_t ww[TOK2_NW]; bool wide;
    tok_masks(t, base, n, &tabs, &nls, ww, &wide);

    // ---- the line in progress when its FORMAT (8th TAB) lies before this tile: GT is ASSUMED to be the first FORMAT key, as the
    //      VCF specification requires; the thread that sees a FORMAT where it is not flags the line for k_tok_parse ----------
    const int gt0 = P.tabs >= 8 ? 0 : TOK_GT_UNDEF;

    // ---- scan A: (line, TABs since the line began, where it began) at every thread's first byte --------------------------
    const int nl = __popc(nls);
    const int last_bit = nl ? 31 - __clz((int)nls) : -1;
    int f = nl ? 1 : 0, v = nl ? tok_tabs_after(tabs, last_bit) : __popc(tabs), cn = nl;
    unsigned long long p = nl ? (unsigned long long)(base + last_bit + 1) : 0ull;
    int plo = (int)(uint32_t)p, phi = (int)(uint32_t)(p >> 32);
#define TOK_STEP_A(CTRL, ROWS) {                                                                                          \
        const int f2 = tok_dpp<CTRL, ROWS>(0, f), v2 = tok_dpp<CTRL, ROWS>(0, v), n2 = tok_dpp<CTRL, ROWS>(0, cn);         \
        const int l2 = tok_dpp<CTRL, ROWS>(0, plo), h2 = tok_dpp<CTRL, ROWS>(0, phi);                                       \
        if (!f) { v += v2; plo = l2; phi = h2; }                                                                           \
        f |= f2; cn += n2; }
    TOK_SCAN_STEPS(TOK_STEP_A)                                       // inclusive, within the wave
#undef TOK_STEP_A
    p = ((unsigned long long)(uint32_t)phi << 32) | (uint32_t)plo;
    if (lane == 63) { s_f[w] = f; s_v[w] = v; s_p[w] = p; s_n[w] = cn; }
    const int ef = tok_dpp<0x138, 0xF>(0, f), ev = tok_dpp<0x138, 0xF>(0, v), en = tok_dpp<0x138, 0xF>(0, cn);      // exclusive within the wave
    const unsigned long long ep = ((unsigned long long)(uint32_t)tok_dpp<0x138, 0xF>(0, phi) << 32) | (uint32_t)tok_dpp<0x138, 0xF>(0, plo);
    __syncthreads();
    int bv = P.tabs, bn = P.lines;                                   // state at the wave's first byte
    unsigned long long bp = P.line_start;
    for (int k = 0; k < w; ++k) { if (s_f[k]) { bv = s_v[k]; bp = s_p[k]; } else bv += s_v[k]; bn += s_n[k]; }
    int line = bn + en, ntab = ef ? ev : bv + ev;
    size_t ls = (size_t)(ef ? ep : bp);

    // ---- scan B: the GT position of the line at every thread's first byte.  Events: a newline (undefined again), a line's
    //      8th TAB (FORMAT begins: parsed by the thread that holds it; at most one per 16 bytes) ---------------------------
    int d = 0, g = TOK_GT_UNDEF, g8 = TOK_GT_UNDEF;
    bool fmt_here = false;
    if (nls != 0 || (ntab < 8 && ntab + __popc(tabs) >= 8)) {        // (most threads hold neither a newline nor a line's 8th TAB)
        uint32_t m = tabs | nls;
        int k = ntab;
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1;
            if ((nls >> j) & 1u) { k = 0; d = 1; g = TOK_GT_UNDEF; }
            else if (++k == 8) { g8 = tok_format_gtpos(t, base + j + 1, n); d = 1; g = g8; fmt_here = true; }
        }
    }
#define TOK_STEP_B(CTRL, ROWS) {                                                                                          \
        const int d2 = tok_dpp<CTRL, ROWS>(0, d), g2 = tok_dpp<CTRL, ROWS>(0, g);                                          \
        if (!d) g = g2;                                                                                                    \
        d |= d2; }
    TOK_SCAN_STEPS(TOK_STEP_B)
#undef TOK_STEP_B
    if (lane == 63) { s_d[w] = d; s_g[w] = g; }
    const int ed = tok_dpp<0x138, 0xF>(0, d);
    int eg = tok_dpp<0x138, 0xF>(0, g);
    if (lane == 0) eg = TOK_GT_UNDEF;
    __syncthreads();
    int bg = gt0;
    for (int k = 0; k < w; ++k) if (s_d[k]) bg = s_g[k];
    int gtpos = ed ? eg : bg;

    // ---- the walk: every TAB and newline of the thread's bytes, in order ----------------------------------------------
    if (blockIdx.x == 0 && tid == 0 && n > 0) {
        line_off[0] = 0;
        if (field_off && max_lines > 0) field_off[0] = 0;
    }
    // ---- the everyday stretch: 32 bytes inside the sample columns of one line, a TAB every fourth byte (genotypes of the form
    //      d/d, d|d or ./.), GT first: the eight genotypes that begin here are taken out of the thread's words in one go --
    //      the window is shifted so that they lie dword by dword, each is checked and encoded, and the eight codes leave as
    //      one 8-byte store.  Anything else about the thread's bytes sends it through the walk below.
    bool everyday = false;
    if (wide && nls == 0 && tabs != 0 && ntab >= 9 && gtpos == 0 && line < max_lines) {
        const int phi = __ffs((int)tabs) - 1;
        const int s0 = ntab - 8;                                       // the sample whose field begins after the first TAB
        if (phi < 4 && tabs == (0x11111111u << phi) && s0 + 8 <= n_samples) {
            const int sh = 8 * (phi + 1);                              // 8 .. 32: the first field begins at byte phi + 1
            uint64_t out = 0;
            bool all_ok = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t a = (ww[k] >> sh) | (ww[k + 1] << (64 - sh));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t q = (uint32_t)(a >> (32 * h));
                    const uint32_t b0 = q & 0xFF, b1 = (q >> 8) & 0xFF, b2 = (q >> 16) & 0xFF, b3 = q >> 24;
                    const uint32_t d0 = b0 - '0', d1 = b2 - '0';
                    const bool form = (b1 == '/' || b1 == '|') && (b3 == '\t' || b3 == ':' || b3 == '\n');
                    const bool digits = d0 <= 9 && d1 <= 9, dots = b0 == '.' && b2 == '.';
                    all_ok = all_ok && form && (digits || dots);
                    const uint32_t code = dots ? 0xFFu : ((d0 << 4) | d1) & 0xFFu;
                    out |= (uint64_t)code << (8 * (2 * k + h));
                }
            }
            if (all_ok) {
                __builtin_memcpy(gt + (size_t)line * pitch + s0, &out, 8);
                everyday = true;
            }
        }
    }
    uint32_t m = everyday ? 0u : (tabs | nls);
    while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1;
        const size_t pos = base + j;
        const bool ok_line = line < max_lines;
        if ((tabs >> j) & 1u) {
            ntab++;
            if (ntab == 1 && ok_line && is_x) {                     // CHROM ends here: assoc.c:94
                const size_t clen = pos - ls;
                is_x[line] = (clen == 0 || (clen == 1 && t[ls] == 'X')) ? 1 : 0;
            }
            if (ntab <= 9 && ok_line && field_off) field_off[(size_t)line * 10 + ntab] = (uint32_t)(pos + 1 - ls);
            if (ntab == 8) { gtpos = g8; if (fmt_here && g8 != 0 && ok_line) redo[line] = 1; }      // later tiles assumed GT first
            if (ntab >= 9 && ok_line && gtpos >= 0) {
                const int sample = ntab - 9;
                if (sample < n_samples) {
                    uint32_t code = 0x100u;                         // "not decided"
                    if (wide && gtpos == 0) {
                        const int k = j + 1;                        // the four bytes after the TAB: k .. k + 3 < TOK2_TB + 8
                        const int wi = k >> 3, sh = (k & 7) * 8;
                        uint64_t a = ww[0], b = ww[1];
#pragma unroll
                        for (int q2 = 1; q2 < TOK2_NW; ++q2) if (wi == q2) { a = ww[q2]; b = q2 + 1 < TOK2_NW ? ww[q2 + 1] : 0; }
                        const uint32_t q = (uint32_t)(sh ? (a >> sh) | (b << (64 - sh)) : a);
                        const uint32_t b0 = q & 0xFF, b1 = (q >> 8) & 0xFF, b2 = (q >> 16) & 0xFF, b3 = q >> 24;
                        if ((b1 == '/' || b1 == '|') && (b3 == '\t' || b3 == ':' || b3 == '\n')) {
                            const uint32_t d0 = b0 - '0', d1 = b2 - '0';
                            if (d0 <= 9 && d1 <= 9) code = (d0 << 4) | d1;
                            else if (b0 == '.' && b2 == '.') code = 0xFFu;      // both alleles missing: 0xFF strict or not
                        }
                    }
                    if (code == 0x100u) code = tok_encode_open(t, pos + 1, n, gtpos, strict);
                    gt[(size_t)line * pitch + sample] = (uint8_t)code;
                }
            }
        } else {                                                     // newline: the line ends at pos
            if (ok_line) tok_close_line(t, line, ntab, gtpos, ls, pos, n_samples, gt, pitch, is_x, field_off, status);
            line++; ntab = 0; ls = pos + 1; gtpos = TOK_GT_UNDEF;
            if (line <= max_lines) line_off[line] = pos + 1;
            if (line < max_lines && pos + 1 < n && field_off) field_off[(size_t)line * 10] = 0;
        }
    }
    // the unterminated last line ends at n: closed by the thread that holds the text's last byte
    if (n > 0 && base <= n - 1 && n - 1 < base + TOK2_TB && t[n - 1] != '\n' && line < max_lines)
        tok_close_line(t, line, ntab, gtpos, ls, n, n_samples, gt, pitch, is_x, field_off, status);
}

}  // namespace hpgv
