// hpgv_text2_kernels.h -- tile-parallel VCF tokenizer (round 2; the shape recogniser and the walk by kind: round 3): two sweeps of
// the text instead of three, and no workgroup walks a line tile by tile (k_tok_parse of hpgv_text_kernels.h does: one memory
// round trip per 4 KiB of a line).  A tile is 8 KiB: 256 threads x 32 bytes.
//
//   k_tok_count2 : per tile its newline count, the TABs after its last newline (all its TABs when it has none) and where that
//                  last newline is;
//   k_tok_scan2a/b: the state at every tile's start -- lines before it, TABs since the current line began, where that line
//                  began -- as a two-level scan (1024 tiles per workgroup, then the workgroups' totals), and the line count;
//   k_tok_parse2 : one workgroup per TILE.  tok_read recognises the EVERYDAY SHAPE of a thread's 32 bytes from the bytes themselves
//                  (eight genotypes d/d, d|d, ./. or .|. behind TABs four bytes apart: tok_pattern) and encodes them at once; other
//                  threads compute exact TAB / newline masks.  A segmented block scan gives every thread the (line, TABs so far,
//                  line start) at its first byte; a second one the GT position of its line (defined by whoever holds the line's
//                  8th TAB, i.e. the start of FORMAT).  A tile in the middle of a line whose FORMAT lies in an earlier tile
//                  ASSUMES GT is the first FORMAT key (the VCF specification requires it when GT is present); the thread that
//                  does see a FORMAT with GT elsewhere or absent LISTS its line, and k_tok_parse_listed (the line-by-line
//                  parse) re-does exactly the listed lines afterwards.  A thread with the shape stores its eight codes; any other
//                  handles its TABs and newlines by kind -- a TAB from the ninth on starts a sample field, a newline closes its
//                  line (status, the 0xFF tail of a short row), the first nine TABs give CHROM .. FORMAT, the first one
//                  chromosome "X" -- each from the state at its byte, which follows from the masks (tok_parse_tile);
//   k_tok_parse3 : count + scan + parse in ONE sweep, the tiles' start states by decoupled look-back (an option, see below).
// Same outputs as k_tok_count / k_tok_scan / k_tok_mark / k_tok_parse, bit for bit (tests/test_gpu_text.py runs all three forms
// against the oracle, down to random bytes from the tokenizer's own alphabet).
// Reference: the per-genotype strdup + get_alleles of assoc.c:45-56 / tdt.c:97-108,150-157 (what is being replaced).
#pragma once
#include "hpgv_text_kernels.h"

namespace hpgv {

constexpr int TOK2_TB = 32;                                          // bytes per thread: 32 TAB / newline bits in one register
constexpr int TOK2_NW = TOK2_TB / 8 + 1;                             // 8-byte words a thread holds: its bytes and the 8 after them
// A tile of the two sweeps is 2 KiB: 64 threads, ONE wave per parsing workgroup.  The parse meets two workgroup barriers per tile and
// a workgroup's place is free again only when its slowest wave is done (a wave with a line's header costs several times a wave of
// sample columns): 16 000 x 10 k samples took 0.314 ms with four waves per workgroup (8 KiB tiles), 0.295 with two, 0.287 with one;
// 64 000 x 2 504 samples 0.419 / 0.36 / 0.34 ms; 800 000 x 200 0.72 / 0.71 / 0.70 ms (same box, profiles/r03_tokenizer_bench.jsonl).
// The counting sweep keeps workgroups of 256 threads, which count four tiles: with smaller ones it is slower.
#ifndef HPGV_TOK2_THREADS
#define HPGV_TOK2_THREADS 64
#endif
constexpr int TOK2_THREADS = HPGV_TOK2_THREADS;
constexpr int TOK2_TILE = TOK2_THREADS * TOK2_TB;                    // bytes per tile (2 KiB)
struct TokAgg { int nl, tabs, last_nl, pad; };                       // last_nl: offset inside the tile, -1 when none
struct TokPre { int lines, tabs; unsigned long long line_start; };   // state at the tile's first byte
constexpr int TOK_GT_UNDEF = -2;                                     // "FORMAT of this line not seen yet"

// 0x80 in every byte of w that equals the byte repeated in c4 (exact: no carry crosses a byte)
__device__ __forceinline__ uint32_t tok_eq_flags(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4, k7 = 0x7F7F7F7Fu;
    return ~(((x & k7) + k7) | x | k7);
}

// What a thread knows about its TOK2_TB bytes: the TAB and newline bits (bit j = byte base + j), the bytes themselves and
// the 8 after them (when they all exist: `wide`), and whether they have the EVERYDAY SHAPE `pat`:
//     [not TAB, not newline]{phi}  (TAB g s g){8}  e        phi = 0 .. 3;  g s g one of d/d d|d ./. .|.;  e a TAB, ':' or newline
// i.e. 32 bytes inside the sample columns of one line with eight everyday genotypes beginning in them.  The shape is
// recognised from the bytes, not from the masks: phi from the first dword's TAB flags, then the ten dwords are realigned so that
// every genotype is one dword [TAB g s g] (v_alignbyte_b32) and checked whole -- a digit is a byte whose high nibble is 3 both as
// it is and after + 6 (a byte that wraps loses its 3, so a carry into the next byte only ever follows a failure);
// the code (first allele << 4 | second) comes out of two shifts in byte 1 and four of them are packed by three v_perm_b32.  A
// thread with the shape has tabs = 0x11111111 << phi and no newline BY CONSTRUCTION, and `codes` holds its eight HPGV8 bytes; any
// other thread computes the masks exactly (SWAR zero-byte test + multiply gather, 3 quarter-rate multiplies per 8 bytes and
// character: what every thread used to pay).  Whether the codes may be stored is the state's business (tok_parse_tile).
struct TokThread { uint32_t tabs, nls; uint64_t w[TOK2_NW]; bool wide, pat; uint64_t codes; };

__device__ __forceinline__ bool tok_pattern(const uint64_t (&w)[TOK2_NW], int *phi_out, uint64_t *codes) {
    uint32_t d[2 * TOK2_NW];
#pragma unroll
    for (int k = 0; k < TOK2_NW; ++k) { d[2 * k] = (uint32_t)w[k]; d[2 * k + 1] = (uint32_t)(w[k] >> 32); }
    const uint32_t zt = tok_eq_flags(d[0], 0x09090909u), zn = tok_eq_flags(d[0], 0x0A0A0A0Au);
    const uint32_t below = (zt & (0u - zt)) - 1u;                     // the bits below the first TAB's flag
    const int phi = ((__ffs((int)zt) - 1) >> 3) & 3;
    // no branch and one verdict at the end: what is wrong with any of the eight genotypes is OR-ed into `wrong` (under M: the TAB byte
    // and the high nibbles of the allele bytes), the separators are OR-ed and AND-ed -- all '/' (0 after the XOR) or all '|' (0x53);
    // a thread that mixes the two walks.  "./." and ".|." become the pseudo-digits 0x3F: the code byte comes out 0xFF by itself, and
    // '.' + 6 = 0x34 passes the second digit test, which is taken on the bytes as they were.
    constexpr uint32_t C = 0x302F3009u, M = 0xF000F0FFu;
    uint32_t wrong = 0, sep_and = 0xFFFFFFFFu;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t a = __builtin_amdgcn_alignbyte(d[k + 1], d[k], (uint32_t)phi);      // [TAB g s g]
        const uint32_t kd = (a & 0xFF00FF00u) == 0x2E002E00u ? 0x11001100u : 0u;
        const uint32_t x = a ^ kd ^ C;                                 // allele bytes 0 .. 9 (15: '.'), TAB byte 0, separator 0 or 0x53
        const uint32_t t = a + 0x06000600u;                            // '0' .. '9' + 6 stays 0x3?; so does 0x2A .. 0x2F, which x catches
        wrong |= x | (t ^ C);
        sep_and &= x;
        v[k] = (x << 4) | (x >> 16);                                   // byte 1: first allele << 4 | second allele
    }
    const uint32_t e = __builtin_amdgcn_alignbyte(d[9], d[8], (uint32_t)phi) & 0xFFu;       // what ends the eighth genotype
    const uint32_t so = wrong & 0x00FF0000u, sa = sep_and & 0x00FF0000u;
    const bool ok = (zt != 0) & ((zn & below) == 0) & ((wrong & M) == 0) & ((so == 0) | ((so == 0x00530000u) & (sa == 0x00530000u))) &
                    ((e == '\t') | (e == ':') | (e == '\n'));
    const uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0501u) | __builtin_amdgcn_perm(v[3], v[2], 0x05010C0Cu);
    const uint32_t hi = __builtin_amdgcn_perm(v[5], v[4], 0x0C0C0501u) | __builtin_amdgcn_perm(v[7], v[6], 0x05010C0Cu);
    *codes = ((uint64_t)hi << 32) | lo;
    *phi_out = phi;
    return ok;
}

__device__ __forceinline__ void tok_read(const char *__restrict__ t, size_t base, size_t n, TokThread &T) {
    T.wide = base + TOK2_TB + 8 <= n;
    T.pat = false; T.codes = 0;
    uint32_t a = 0, b = 0;
    if (T.wide) {
#pragma unroll
        for (int k = 0; k < TOK2_NW; ++k) __builtin_memcpy(&T.w[k], t + base + 8 * k, 8);
        int phi;
        T.pat = tok_pattern(T.w, &phi, &T.codes);
        if (T.pat) a = 0x11111111u << phi;
        else {
#pragma unroll
            for (int k = 0; k < TOK2_TB / 8; ++k) { a |= tok_byte_mask(T.w[k], '\t') << (8 * k); b |= tok_byte_mask(T.w[k], '\n') << (8 * k); }
        }
    } else {
#pragma unroll
        for (int k = 0; k < TOK2_NW; ++k) T.w[k] = 0;
        for (int j = 0; j < TOK2_TB; ++j)
            if (base + j < n) { const char c = t[base + j]; if (c == '\t') a |= 1u << j; else if (c == '\n') b |= 1u << j; }
    }
    T.tabs = a; T.nls = b;
}
// the counting sweep's view of the same bytes: newlines, the TABs behind the last newline (all the TABs when there is none) and
// that newline's bit (-1).  Flags stay where the SWAR test leaves them (0x80 per byte) and are only counted -- 6 operations per
// dword and character, no gather; a thread that does hold a newline (one in 1 250 at 10 k samples) takes the exact masks
__device__ __forceinline__ void tok_count_thread(const char *__restrict__ t, size_t base, size_t n, int *nl, int *tabs_after, int *last_bit) {
    uint32_t tabs = 0, nls = 0;
    if (base + TOK2_TB <= n) {
        uint32_t d[TOK2_TB / 4];
        __builtin_memcpy(d, t + base, TOK2_TB);
        int ct = 0; uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < TOK2_TB / 4; ++k) { ct += __popc(tok_eq_flags(d[k], 0x09090909u)); any |= tok_eq_flags(d[k], 0x0A0A0A0Au); }
        if (any == 0) { *nl = 0; *tabs_after = ct; *last_bit = -1; return; }
#pragma unroll
        for (int k = 0; k < TOK2_TB / 8; ++k) {
            const uint64_t w = ((uint64_t)d[2 * k + 1] << 32) | d[2 * k];
            tabs |= tok_byte_mask(w, '\t') << (8 * k); nls |= tok_byte_mask(w, '\n') << (8 * k);
        }
    } else {
        for (int j = 0; j < TOK2_TB; ++j)
            if (base + j < n) { const char c = t[base + j]; if (c == '\t') tabs |= 1u << j; else if (c == '\n') nls |= 1u << j; }
    }
    *nl = __popc(nls);
    *last_bit = *nl ? 31 - __clz((int)nls) : -1;
    *tabs_after = *nl ? (*last_bit >= 31 ? 0 : __popc(tabs >> (*last_bit + 1))) : __popc(tabs);
}
// TABs behind bit `last` (the thread's last newline): a shift by 32 is not a shift
__device__ __forceinline__ int tok_tabs_after(uint32_t tabs, int last) { return last >= 31 ? 0 : __popc(tabs >> (last + 1)); }

// one lane-shift of a wave scan as a DPP move (one instruction per value; a lane without a source keeps `identity`):
// row_shr:1/2/4/8 inside the rows of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; wave_shr:1 for
// the exclusive value
template <int CTRL, int ROWS> __device__ __forceinline__ int tok_dpp(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWS, 0xF, false);
}
#define TOK_RFL(x) __builtin_amdgcn_readfirstlane(x)                  // a value every lane holds alike, into a scalar register
#define TOK_SCAN_STEPS(STEP) STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)

constexpr int TOK2_WPT = TOK2_THREADS / 64;                          // waves per tile
constexpr int TOK2_COUNT_TILES = 256 / TOK2_THREADS;                 // tiles a workgroup of the counting sweep takes
static __global__ __launch_bounds__(256) void k_tok_count2(const char *__restrict__ text, size_t n, int n_tiles, TokAgg *__restrict__ agg) {
    __shared__ int s_nl[4], s_last[4], s_tabs[4];
    // waves [g WPT, (g + 1) WPT) count tile (tiles per workgroup) b + g (b = the workgroup's index); `tid` is the thread's place inside its tile
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), g = w / TOK2_WPT, w0 = g * TOK2_WPT;
    const int tid = (int)threadIdx.x & (TOK2_THREADS - 1);
    const int tile = TOK2_COUNT_TILES * (int)blockIdx.x + g;
    const size_t base = (size_t)tile * TOK2_TILE + (size_t)tid * TOK2_TB;
    int nl, tabs_after, last_bit;                                     // tabs_after: all the thread's TABs when it holds no newline
    tok_count_thread(text, base, n, &nl, &tabs_after, &last_bit);
    const int all_tabs = nl ? 0 : tabs_after;                         // (only asked of threads behind the tile's last newline: they hold none)
    // the last thread of the tile that holds a newline, and how many newlines there are: a wave without one (19 of 20 at 10 k
    // samples) knows from its ballot; sums over the wave are DPP scans (the total arrives in lane 63)
    const unsigned long long has = __ballot(nl != 0);
    int c = 0;
    if (has) {
        c = nl;
#define TOK_STEP_C(CTRL, ROWS) c += tok_dpp<CTRL, ROWS>(0, c);
        TOK_SCAN_STEPS(TOK_STEP_C)
#undef TOK_STEP_C
    }
    if (lane == 63) { s_nl[w] = c; s_last[w] = has ? (w - w0) * 64 + 63 - __clzll(has) : -1; }
    __syncthreads();
    int tl = -1;
#pragma unroll
    for (int k = 0; k < TOK2_WPT; ++k) tl = max(tl, s_last[w0 + k]);
    const int tlast = __builtin_amdgcn_readfirstlane(tl);
    int mine = tid > tlast ? all_tabs : (tid == tlast ? tabs_after : 0);
#define TOK_STEP_M(CTRL, ROWS) mine += tok_dpp<CTRL, ROWS>(0, mine);
    TOK_SCAN_STEPS(TOK_STEP_M)
#undef TOK_STEP_M
    if (lane == 63) s_tabs[w] = mine;
    __syncthreads();
    if (tid == (tlast < 0 ? 0 : tlast) && tile < n_tiles) {           // one thread writes the tile's whole record
        TokAgg a;
        a.nl = 0; a.tabs = 0;
#pragma unroll
        for (int k = 0; k < TOK2_WPT; ++k) { a.nl += s_nl[w0 + k]; a.tabs += s_tabs[w0 + k]; }
        a.last_nl = tlast < 0 ? -1 : tid * TOK2_TB + last_bit;
        a.pad = 0;
        agg[tile] = a;
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

// the fold of the first `count` group totals onto the text's start, by the whole workgroup (the same value in every thread):
// a thousand totals per round, a wave scan by shuffles and the 16 wave totals through LDS
__device__ __forceinline__ TokState tok_fold_groups(const TokState *__restrict__ group_total, int count, int *w_lines, int *w_tabs, long long *w_ls,
                                                    const TokAgg *__restrict__ prefix = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the text's start begins a line; on the decoder's grid the first tile begins `skip` bytes IN FRONT of the window, inside
    // the lines before it: prefix->nl of them end there -- lines -nl .. -1, the first of which is taken to be past its FORMAT
    // (nine TABs: what follows are sample columns).  Lines below 0 are kept nowhere.
    TokState c = {prefix ? -prefix->nl : 0, prefix ? 9 : 0, 0};
    for (int g0 = 0; g0 < count; g0 += TOK_SCAN_THREADS) {
        TokState inc = {0, 0, -1};
        if (g0 + tid < count) inc = group_total[g0 + tid];
        for (int off = 1; off < 64; off <<= 1) {
            TokState o; o.lines = __shfl_up(inc.lines, off); o.tabs = __shfl_up(inc.tabs, off); o.ls = __shfl_up(inc.ls, off);
            if (lane >= off) inc = tok_fold(o, inc);
        }
        if (lane == 63) { w_lines[w] = inc.lines; w_tabs[w] = inc.tabs; w_ls[w] = inc.ls; }
        __syncthreads();
        for (int k = 0; k < TOK_SCAN_THREADS / 64; ++k) { const TokState o = {w_lines[k], w_tabs[k], w_ls[k]}; c = tok_fold(c, o); }
        __syncthreads();
    }
    return c;
}

static __global__ __launch_bounds__(TOK_SCAN_THREADS) void k_tok_scan2b(TokPre *__restrict__ pre, int n_tiles, TokState *__restrict__ group_total,
                                                                      int n_groups, const char *__restrict__ text, size_t n,
                                                                      int *__restrict__ n_lines, unsigned long long *__restrict__ line_off, int max_lines,
                                                                      int *__restrict__ redo_n, const TokAgg *__restrict__ prefix = nullptr) {
    // every workgroup folds the totals of the groups in front of its own (n_groups = n_tiles / 1024: 77 for 640 MB of text) and
    // fixes up its group; workgroup 0 folds them all for the line count (one thread doing this took 11 us per call)
    __shared__ int w_lines[TOK_SCAN_THREADS / 64], w_tabs[TOK_SCAN_THREADS / 64];
    __shared__ long long w_ls[TOK_SCAN_THREADS / 64];
    const int mine = (int)blockIdx.x < n_groups ? (int)blockIdx.x : n_groups;
    const TokState base = tok_fold_groups(group_total, mine, w_lines, w_tabs, w_ls, prefix);
    if (blockIdx.x == 0) {
        const TokState all = tok_fold_groups(group_total, n_groups, w_lines, w_tabs, w_ls, prefix);
        if (threadIdx.x == 0) {
            const int tail = (n > 0 && text[n - 1] != '\n') ? 1 : 0;      // unterminated last line
            *n_lines = all.lines + tail;
            if (tail && all.lines + 1 <= max_lines) line_off[all.lines + 1] = n;
            if (n == 0) line_off[0] = 0;
            if (redo_n) *redo_n = 0;                                 // (k_tok_parse2 runs behind this kernel)
        }
    }
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

// what a tile's parse writes
struct TokOut {
    int max_lines, n_samples, strict;
    uint8_t *gt; size_t pitch; uint8_t *is_x;
    unsigned long long *line_off; uint32_t *field_off; int *status;
    int *redo, *redo_n;                                               // the lines to parse again line by line: a list and its length
};
struct TokShared { int s_f[4], s_v[4], s_n[4], s_d[4], s_g[4]; unsigned long long s_p[4]; };

// One tile (TOK2_TILE bytes from tile_base) parsed by the workgroup: P = the state at the tile's first byte, (tabs, nls, ww, wide) =
// tok_read of this thread's bytes.  Two workgroup barriers; the shared arrays may be used again right after the call.
__device__ __forceinline__ void tok_parse_tile(const char *__restrict__ t, const size_t n, const size_t tile_base, const bool first_tile, const TokPre P,
                                               const TokThread &T, TokShared &S, const TokOut &O) {
    // line indices below 0 exist on the decoder's tile grid only (k_tok_parse2 with skip > 0): the bytes of the window's first tile
    // that lie in front of the window are the tail of the line before it, parsed as line -1 and kept nowhere
#define TOK_LINE_OK(L) ((unsigned)(L) < (unsigned)max_lines)
    const uint32_t tabs = T.tabs, nls = T.nls;
    const int tid = threadIdx.x, lane = tid & 63, w = TOK_RFL(tid >> 6);
    const size_t base = tile_base + (size_t)tid * TOK2_TB;
    const int max_lines = O.max_lines, n_samples = O.n_samples, strict = O.strict;
    uint8_t *__restrict__ gt = O.gt; const size_t pitch = O.pitch; uint8_t *__restrict__ is_x = O.is_x;
    unsigned long long *__restrict__ line_off = O.line_off; uint32_t *__restrict__ field_off = O.field_off; int *__restrict__ status = O.status;
    int (&s_f)[4] = S.s_f, (&s_v)[4] = S.s_v, (&s_n)[4] = S.s_n, (&s_d)[4] = S.s_d, (&s_g)[4] = S.s_g;
    unsigned long long (&s_p)[4] = S.s_p;

    // ---- the line in progress when its FORMAT (8th TAB) lies before this tile: GT is ASSUMED to be the first FORMAT key, as the
    //      VCF specification requires; the thread that sees a FORMAT where it is not flags the line for k_tok_parse ----------
    const int gt0 = P.tabs >= 8 ? 0 : TOK_GT_UNDEF;

    // ---- scan A: (line, TABs since the line began, where it began) at every thread's first byte --------------------------
    // A wave without a newline (19 of 20 at 10 k samples) only adds up its TABs: one DPP addition per step instead of five moves and
    // their selects.  (The branch is uniform: the ballot is the wave's.)
    int f, v, cn, ef, ev, en;
    unsigned long long p, ep;
    if (__ballot(nls != 0) == 0) {
        f = 0; cn = 0; p = 0; ef = 0; en = 0; ep = 0;
        v = __popc(tabs);
#define TOK_STEP_A0(CTRL, ROWS) v += tok_dpp<CTRL, ROWS>(0, v);
        TOK_SCAN_STEPS(TOK_STEP_A0)
#undef TOK_STEP_A0
        ev = tok_dpp<0x138, 0xF>(0, v);
    } else {
        // three values per lane: "a newline so far" in the sign bit over the TABs since it (or since the wave began), the newlines, and
        // where the line in progress began INSIDE THE TILE (0: not inside) -- a lane without a newline adds the TABs and takes the
        // place of what lies in front of it
        const int nl = __popc(nls);
        const int last_bit = nl ? 31 - __clz((int)nls) : -1;
        int fv = nl ? (int)(0x80000000u | (uint32_t)tok_tabs_after(tabs, last_bit)) : __popc(tabs);
        int rp = nl ? tid * TOK2_TB + last_bit + 1 : 0;
        cn = nl;
#define TOK_STEP_A(CTRL, ROWS) {                                                                                          \
        const int fv2 = tok_dpp<CTRL, ROWS>(0, fv), r2 = tok_dpp<CTRL, ROWS>(0, rp);                                        \
        cn += tok_dpp<CTRL, ROWS>(0, cn);                                                                                  \
        if (fv >= 0) { fv += fv2; rp = r2; } }
        TOK_SCAN_STEPS(TOK_STEP_A)                                   // inclusive, within the wave
#undef TOK_STEP_A
        f = fv < 0; v = fv & 0x7FFFFFFF;
        p = f ? (unsigned long long)(tile_base + (size_t)rp) : 0ull;
        const int efv = tok_dpp<0x138, 0xF>(0, fv), erp = tok_dpp<0x138, 0xF>(0, rp);                                 // exclusive within the wave
        en = tok_dpp<0x138, 0xF>(0, cn);
        ef = efv < 0; ev = efv & 0x7FFFFFFF;
        ep = (unsigned long long)(tile_base + (size_t)erp);
    }
    if (lane == 63) { s_f[w] = f; s_v[w] = v; s_p[w] = p; s_n[w] = cn; }
    __syncthreads();
    int bv = P.tabs, bn = P.lines;                                   // state at the wave's first byte
    unsigned long long bp = P.line_start;
    for (int k = 0; k < w; ++k) {                                    // (the waves in front of this one: values every lane reads alike)
        const int kf = TOK_RFL(s_f[k]), kv = TOK_RFL(s_v[k]), kn = TOK_RFL(s_n[k]);
        if (kf) {
            bv = kv;
            bp = ((unsigned long long)(uint32_t)TOK_RFL((int)(s_p[k] >> 32)) << 32) | (uint32_t)TOK_RFL((int)(uint32_t)s_p[k]);
        } else bv += kv;
        bn += kn;
    }
    int line = bn + en, ntab = ef ? ev : bv + ev;
    size_t ls = (size_t)(ef ? ep : bp);

    // ---- scan B: the GT position of the line at every thread's first byte.  Events: a newline (undefined again), a line's
    //      8th TAB (FORMAT begins: parsed by the thread that holds it; at most one per 16 bytes) ---------------------------
    //      The line in progress reaches its 8th TAB inside this thread when it has fewer than eight in front of it and enough here:
    //      `from8` = its TABs from the 8th on.  Only a thread with eight or more TABs BEHIND its first newline (a second FORMAT inside
    //      the same 32 bytes: lines of empty fields) goes through its TABs and newlines one by one, here and below.
    int d = 0, g = TOK_GT_UNDEF, g8 = TOK_GT_UNDEF;
    bool fmt_here = false;
    const int n_nl = __popc(nls);
    const uint32_t seg0 = n_nl ? ((1u << (__ffs((int)nls) - 1)) - 1u) : 0xFFFFFFFFu;      // the bits in front of the first newline
    const uint32_t t0 = tabs & seg0;                                 // the TABs of the line in progress
    const bool in_order = n_nl != 0 && __popc(tabs & ~seg0) >= 8;
    uint32_t from8 = 0;
    if (!in_order) {
        if (ntab < 8 && ntab + __popc(t0) >= 8) {
            from8 = t0;
            for (int drop = 7 - ntab; drop > 0; --drop) from8 &= from8 - 1;
            g8 = tok_format_gtpos(t, base + (__ffs((int)from8) - 1) + 1, n);
            fmt_here = true;
        }
        d = (n_nl != 0 || fmt_here) ? 1 : 0;
        g = n_nl != 0 ? TOK_GT_UNDEF : g8;
    } else {
        uint32_t m = tabs | nls;
        int k = ntab;
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1;
            if ((nls >> j) & 1u) { k = 0; d = 1; g = TOK_GT_UNDEF; }
            else if (++k == 8) { g8 = tok_format_gtpos(t, base + j + 1, n); d = 1; g = g8; fmt_here = true; }
        }
    }
    const bool wave_events = __ballot(d != 0) != 0;                    // (uniform; most waves hold neither a newline nor a line's 8th TAB)
    int ed = 0, eg = TOK_GT_UNDEF;
    if (wave_events) {
        int e = d ? g + 3 : 0;                                       // one value: 0 = no event, else the GT position + 3 (GT_UNDEF + 3 = 1)
#define TOK_STEP_B(CTRL, ROWS) { const int e2 = tok_dpp<CTRL, ROWS>(0, e); if (!e) e = e2; }
        TOK_SCAN_STEPS(TOK_STEP_B)
#undef TOK_STEP_B
        const int ee = tok_dpp<0x138, 0xF>(0, e);
        d = e != 0; g = e ? e - 3 : TOK_GT_UNDEF;
        ed = ee != 0; eg = ee ? ee - 3 : TOK_GT_UNDEF;
    }
    if (lane == 63) { s_d[w] = d; s_g[w] = g; }
    __syncthreads();
    int bg = gt0;
    for (int k = 0; k < w; ++k) if (TOK_RFL(s_d[k])) bg = TOK_RFL(s_g[k]);
    int gtpos = ed ? eg : bg;

    // ---- the walk: every TAB and newline of the thread's bytes, in order ----------------------------------------------
    if (first_tile && tid == 0 && n > 0) {
        line_off[0] = 0;
        if (field_off && max_lines > 0) field_off[0] = 0;
    }
    // ---- the everyday stretch: 32 bytes inside the sample columns of one line, a TAB every fourth byte (genotypes of the form
    //      d/d, d|d, ./. or .|.): tok_read has recognised the shape and encoded the eight genotypes that begin here; with GT first
    //      in FORMAT and all eight samples inside the row they leave as one 8-byte store.  Anything else walks.
    bool everyday = false;
    if (T.pat && ntab >= 9 && gtpos == 0 && TOK_LINE_OK(line)) {
        const int s0 = ntab - 8;                                       // the sample whose field begins after the first TAB
        if (s0 + 8 <= n_samples) {
            __builtin_memcpy(gt + (size_t)line * pitch + s0, &T.codes, 8);
            everyday = true;
        }
    }
    // ---- everything else: the thread's TABs and newlines BY KIND.  What a TAB or a newline does depends on (line, TABs so far, line
    //      start, GT position) at its byte, and all four follow from the thread's start state and its two masks without walking: so
    //      the sample TABs, the header TABs (a line's first nine) and the newlines get a loop each, every one with a short body of its
    //      own.  One loop over all of them in order ran every body in every round of a wave that holds all three kinds -- and at 200
    //      samples every wave does: 1 390 vector instructions per wave, 97 % of the issue rate (profiles/r03_tokenizer_pmc.json).
    //      Only a thread with eight or more TABs behind its first newline (a second FORMAT inside the same 32 bytes: lines of empty
    //      fields) walks in order, as before.
    const uint32_t m_all = everyday ? 0u : (tabs | nls);
    if (m_all != 0 && !in_order) {
        const int gt_seg0 = ntab >= 8 ? gtpos : (fmt_here ? g8 : TOK_GT_UNDEF);            // that line's GT position once its FORMAT has gone by
        const uint32_t ts = ntab >= 8 ? t0 : (from8 & (from8 - 1));   // its sample TABs: from the line's ninth TAB on
        uint32_t th = tabs & ~ts;                                    // header TABs: they set field offsets (the ninth is both)
        if (ntab < 9) th |= ts & (0u - ts);
        // -- sample fields
        if (TOK_LINE_OK(line) && gt_seg0 >= 0) {
            uint8_t *row = gt + (size_t)line * pitch;
            for (uint32_t m = ts; m; m &= m - 1) {
                const int j = __ffs((int)m) - 1;
                const int sample = ntab + __popc(t0 & ((2u << j) - 1u)) - 9;
                if (sample >= n_samples) break;                      // (the samples only grow)
                const size_t q0 = base + j + 1;
                uint32_t code = 0x100u;                              // "not decided"
                if (gt_seg0 == 0 && q0 + 4 <= n) {                   // the everyday forms, out of the four bytes behind the TAB
                    uint32_t q;
                    __builtin_memcpy(&q, t + q0, 4);
                    const uint32_t b0 = q & 0xFF, b1 = (q >> 8) & 0xFF, b2 = (q >> 16) & 0xFF, b3 = q >> 24;
                    if ((b1 == '/' || b1 == '|') && (b3 == '\t' || b3 == ':' || b3 == '\n')) {
                        const uint32_t d0 = b0 - '0', d1 = b2 - '0';
                        if (d0 <= 9 && d1 <= 9) code = (d0 << 4) | d1;
                        else if (b0 == '.' && b2 == '.') code = 0xFFu;      // both alleles missing: 0xFF strict or not
                    }
                }
                if (code == 0x100u) code = tok_encode_open(t, q0, n, gt_seg0, strict);
                row[sample] = (uint8_t)code;
            }
        }
        // -- header TABs: CHROM's end (chromosome "X": assoc.c:94), the field offsets, the FORMAT that does not begin with GT
        for (uint32_t m = th; m; m &= m - 1) {
            const int j = __ffs((int)m) - 1;
            const uint32_t below = (1u << j) - 1u, nb = nls & below;
            const int line_j = line + __popc(nb);
            int ntab_j; size_t ls_j;
            if (nb == 0) { ntab_j = ntab + __popc(tabs & below) + 1; ls_j = ls; }
            else { const int last = 31 - __clz((int)nb); ntab_j = __popc(tabs & below & ~((2u << last) - 1u)) + 1; ls_j = base + last + 1; }
            if (line_j >= max_lines) break;                          // (the lines only grow)
            if (line_j < 0) continue;                                // (the tail of the line before the window: nothing of it is kept)
            const size_t pos = base + j;
            if (ntab_j == 1 && is_x) { const size_t clen = pos - ls_j; is_x[line_j] = (clen == 0 || (clen == 1 && t[ls_j] == 'X')) ? 1 : 0; }
            if (field_off) field_off[(size_t)line_j * 10 + ntab_j] = (uint32_t)(pos + 1 - ls_j);
            if (ntab_j == 8 && fmt_here && g8 != 0) O.redo[atomicAdd(O.redo_n, 1)] = line_j;      // later tiles assumed GT first (one thread per line: no line is listed twice)
        }
        // -- newlines: what a line's end settles, and where the next line begins
        for (uint32_t m = nls; m; m &= m - 1) {
            const int j = __ffs((int)m) - 1;
            const uint32_t below = (1u << j) - 1u, nb = nls & below;
            const int line_j = line + __popc(nb);
            int ntab_j, g_j; size_t ls_j;
            if (nb == 0) { ntab_j = ntab + __popc(tabs & below); ls_j = ls; g_j = gt_seg0; }
            else { const int last = 31 - __clz((int)nb); ntab_j = __popc(tabs & below & ~((2u << last) - 1u)); ls_j = base + last + 1; g_j = TOK_GT_UNDEF; }
            const size_t pos = base + j;
            if (TOK_LINE_OK(line_j)) tok_close_line(t, line_j, ntab_j, g_j, ls_j, pos, n_samples, gt, pitch, is_x, field_off, status);
            if (line_j + 1 >= 0 && line_j + 1 <= max_lines) line_off[line_j + 1] = pos + 1;      // (line_j = -1: where the window's first line begins)
            if (TOK_LINE_OK(line_j + 1) && pos + 1 < n && field_off) field_off[(size_t)(line_j + 1) * 10] = 0;
        }
    }
    // the state behind the thread's last byte (all the unterminated last line needs)
    if (!in_order) {
        if (n_nl == 0) { gtpos = ntab >= 8 ? gtpos : (fmt_here ? g8 : TOK_GT_UNDEF); ntab += __popc(tabs); }
        else { const int last = 31 - __clz((int)nls); line += n_nl; ntab = __popc(tabs & ~((2u << last) - 1u)); ls = base + last + 1; gtpos = TOK_GT_UNDEF; }
    }
    uint32_t m = in_order ? m_all : 0u;
    while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1;
        const size_t pos = base + j;
        const bool ok_line = TOK_LINE_OK(line);
        if ((tabs >> j) & 1u) {
            ntab++;
            if (ntab == 1 && ok_line && is_x) {                     // CHROM ends here: assoc.c:94
                const size_t clen = pos - ls;
                is_x[line] = (clen == 0 || (clen == 1 && t[ls] == 'X')) ? 1 : 0;
            }
            if (ntab <= 9 && ok_line && field_off) field_off[(size_t)line * 10 + ntab] = (uint32_t)(pos + 1 - ls);
            if (ntab == 8) {                                         // (several FORMATs may lie in this thread: g8 is only the last one's)
                gtpos = tok_format_gtpos(t, pos + 1, n);
                if (gtpos != 0 && ok_line) O.redo[atomicAdd(O.redo_n, 1)] = line;
            }
            if (ntab >= 9 && ok_line && gtpos >= 0) {
                const int sample = ntab - 9;
                if (sample < n_samples) gt[(size_t)line * pitch + sample] = (uint8_t)tok_encode_open(t, pos + 1, n, gtpos, strict);
            }
        } else {                                                     // newline: the line ends at pos
            if (ok_line) tok_close_line(t, line, ntab, gtpos, ls, pos, n_samples, gt, pitch, is_x, field_off, status);
            line++; ntab = 0; ls = pos + 1; gtpos = TOK_GT_UNDEF;
            if (line >= 0 && line <= max_lines) line_off[line] = pos + 1;
            if (TOK_LINE_OK(line) && pos + 1 < n && field_off) field_off[(size_t)line * 10] = 0;
        }
    }
    // the unterminated last line ends at n: closed by the thread that holds the text's last byte
    if (n > 0 && base <= n - 1 && n - 1 < base + TOK2_TB && t[n - 1] != '\n' && TOK_LINE_OK(line))
        tok_close_line(t, line, ntab, gtpos, ls, n, n_samples, gt, pitch, is_x, field_off, status);
#undef TOK_LINE_OK
}

static __global__ __launch_bounds__(TOK2_THREADS) void k_tok_parse2(const char *__restrict__ text, size_t n, const TokPre *__restrict__ pre,
                                                    int max_lines, int n_samples, int strict, uint8_t *__restrict__ gt, size_t pitch,
                                                    uint8_t *__restrict__ is_x, unsigned long long *__restrict__ line_off,
                                                    uint32_t *__restrict__ field_off, int *__restrict__ status,
                                                    int *__restrict__ redo, int *__restrict__ redo_n /* the lines to parse again line by line */,
                                                    int skip = 0 /* the decoder's grid: bytes of the first tile in front of the window */) {
    __shared__ TokShared S;
    const size_t tile_base = (size_t)blockIdx.x * TOK2_TILE;
    TokThread T;
    tok_read(text, tile_base + (size_t)threadIdx.x * TOK2_TB, n, T);
    const TokOut O = {max_lines, n_samples, strict, gt, pitch, is_x, line_off, field_off, status, redo, redo_n};
    tok_parse_tile(text, n, tile_base, blockIdx.x == 0 && skip == 0, pre[blockIdx.x], T, S, O);
}

// ---------------------------------------------------------------------------------------------------------------------
// Windows of text the bgzip decoder left on the device (round 4): the counting sweep is NOT run.  The CRC kernel of the decoder
// (hpgv_crc_kernels.h, which reads every decoded byte anyway) has left, per 2 KiB tile of an ABSOLUTE grid over the decoded
// text, the tile's record in two halves: TokAgg2.h[0] from the block that holds the tile's first byte, h[1] from the block that
// follows it inside the tile (BGZF blocks are 65 280 bytes: their seams fall inside tiles).  A window begins at a line start,
// anywhere: its first tile is taken WHOLE -- the `skip` bytes in front of the window are the ends of the lines before it, which the
// parse walks as lines -k .. -1 and keeps nowhere (k = the newlines among them: one more small count) --, its last tile is counted
// again up to the window's end (one workgroup).  Everything else -- the scan, the parse -- is the two-sweep form's, on
// positions that count from the first tile's start; k_tok_grid_finish moves line_off to the window's start at the end.
// A tile that more than two blocks touch, or that holds text the decoder did not write (a block handed back to the host), is
// marked (TOK_AGG_COMPLEX) and counted again by the scan's thread.
struct TokAgg2 { TokAgg h[2]; };                                     // TokAgg.last_nl holds offset + 1 here (0 = none: zeroed memory is "nothing yet"); .pad: flags
enum { TOK_AGG_WRITTEN = 1, TOK_AGG_COMPLEX = 2 };

// lane's view of its 32 bytes of a tile, clipped to [lo, hi): newlines, the TABs behind the last one (all when none), its bit
__device__ __forceinline__ void tok_count_thread_clip(const char *__restrict__ t, size_t base, size_t lo, size_t hi, int *nl, int *tabs_after, int *last_bit) {
    if (base >= lo && base + TOK2_TB <= hi) { tok_count_thread(t, base, hi, nl, tabs_after, last_bit); return; }
    uint32_t tabs = 0, nls = 0;
    if (((uintptr_t)(t + base) & 3u) == 0 && hi > lo) {
        // eight aligned dwords, all in flight at once: one that holds no byte of [lo, hi) is replaced by one that does (an aligned
        // dword with a byte of the text in it lies inside the text's pages), and the bits outside the range are dropped
        const size_t safe = ((size_t)(uintptr_t)(t + lo) & ~(size_t)3) - (size_t)(uintptr_t)t;
        uint32_t d[TOK2_TB / 4];
        #pragma unroll
        for (int k = 0; k < TOK2_TB / 4; ++k) {
            const size_t a = base + 4 * (size_t)k;
            d[k] = *reinterpret_cast<const uint32_t *>(t + (a < hi && a + 4 > lo ? a : safe));
        }
        #pragma unroll
        for (int k = 0; k < TOK2_TB / 8; ++k) {
            const uint64_t w = ((uint64_t)d[2 * k + 1] << 32) | d[2 * k];
            tabs |= tok_byte_mask(w, '\t') << (8 * k); nls |= tok_byte_mask(w, '\n') << (8 * k);
        }
        const size_t first = lo > base ? (lo - base < (size_t)TOK2_TB ? lo - base : (size_t)TOK2_TB) : 0;
        const size_t last = hi > base ? (hi - base < (size_t)TOK2_TB ? hi - base : (size_t)TOK2_TB) : 0;
        const uint32_t m = last > first ? (uint32_t)(((1ull << last) - 1ull) & ~((1ull << first) - 1ull)) : 0u;
        tabs &= m; nls &= m;
    } else
        for (int j = 0; j < TOK2_TB; ++j)
            if (base + j >= lo && base + j < hi) { const char c = t[base + j]; if (c == '\t') tabs |= 1u << j; else if (c == '\n') nls |= 1u << j; }
    *nl = __popc(nls);
    *last_bit = *nl ? 31 - __clz((int)nls) : -1;
    *tabs_after = *nl ? tok_tabs_after(tabs, *last_bit) : __popc(tabs);
}

// a lane's 32 bytes already in registers (eight dwords, all inside the text): newlines, the TABs behind the last one (all when
// none), its bit -- tok_count_thread's fast path without its load
__device__ __forceinline__ void tok_count_regs(const uint32_t (&d)[TOK2_TB / 4], int *nl, int *tabs_after, int *last_bit) {
    int ct = 0; uint32_t any = 0;
    #pragma unroll
    for (int k = 0; k < TOK2_TB / 4; ++k) { ct += __popc(tok_eq_flags(d[k], 0x09090909u)); any |= tok_eq_flags(d[k], 0x0A0A0A0Au); }
    if (any == 0) { *nl = 0; *tabs_after = ct; *last_bit = -1; return; }
    uint32_t tabs = 0, nls = 0;
    #pragma unroll
    for (int k = 0; k < TOK2_TB / 8; ++k) {
        const uint64_t w = ((uint64_t)d[2 * k + 1] << 32) | d[2 * k];
        tabs |= tok_byte_mask(w, '\t') << (8 * k); nls |= tok_byte_mask(w, '\n') << (8 * k);
    }
    *nl = __popc(nls);
    *last_bit = 31 - __clz((int)nls);
    *tabs_after = tok_tabs_after(tabs, *last_bit);
}

// ONE WAVE: the lanes' counts of a tile folded into its record, which LANE 63 returns (last_nl = offset + 1, 0 = none).  A tile
// without a newline (nineteen in twenty at 10 k samples) is one DPP sum of the TABs.
__device__ __forceinline__ TokAgg tok_wave_fold(int nl, int tabs_after, int last_bit) {
    const int lane = threadIdx.x & 63;
    const unsigned long long has = __ballot(nl != 0);
    TokAgg a;
    a.pad = TOK_AGG_WRITTEN;
    if (has == 0) {
        int v = tabs_after;
#define TOK_STEP_F0(CTRL, ROWS) v += tok_dpp<CTRL, ROWS>(0, v);
        TOK_SCAN_STEPS(TOK_STEP_F0)
#undef TOK_STEP_F0
        a.nl = 0; a.tabs = v; a.last_nl = 0;
        return a;
    }
    const int tlast = 63 - __clzll(has);
    int c = nl, mine = lane >= tlast ? tabs_after : 0;               // (a lane behind the last newline holds none: all its TABs)
#define TOK_STEP_F1(CTRL, ROWS) { c += tok_dpp<CTRL, ROWS>(0, c); mine += tok_dpp<CTRL, ROWS>(0, mine); }
    TOK_SCAN_STEPS(TOK_STEP_F1)
#undef TOK_STEP_F1
    const int lb = __builtin_amdgcn_readlane(last_bit, tlast);
    a.nl = c; a.tabs = mine; a.last_nl = tlast * TOK2_TB + lb + 1;
    return a;
}

// ONE WAVE: the record of the bytes [lo, hi) of the tile that begins at tile_base (lane 63 returns it)
__device__ __forceinline__ TokAgg tok_wave_agg(const char *__restrict__ t, size_t tile_base, size_t lo, size_t hi) {
    const int lane = threadIdx.x & 63;
    int nl, tabs_after, last_bit;
    tok_count_thread_clip(t, tile_base + (size_t)lane * TOK2_TB, lo, hi, &nl, &tabs_after, &last_bit);
    return tok_wave_fold(nl, tabs_after, last_bit);
}

// as k_tok_scan2a, the tiles' records taken from the decoder's grid: tile i of the window = grid tile t0 + i; the last tile's
// record comes from `last` (counted again up to the window's end)
static __global__ __launch_bounds__(TOK_SCAN_THREADS) void k_tok_scan2a_grid(const TokAgg2 *__restrict__ agg2, long t0, const TokAgg *__restrict__ last,
                                                                           const char *__restrict__ text /* from the first tile's start */, size_t n,
                                                                           int n_tiles, TokPre *__restrict__ pre, TokState *__restrict__ group_total) {
    __shared__ int w_lines[16], w_tabs[16];
    __shared__ long long w_ls[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x * TOK_SCAN_THREADS + tid;
    TokState me = {0, 0, -1};
    if (i < n_tiles) {
        TokAgg a;
        if (i == n_tiles - 1) { a = last[0]; a.last_nl += 1; }      // (k_tok_count2 writes the offset itself, -1 = none)
        else {
            const TokAgg2 r = agg2[t0 + i];
            if ((r.h[0].pad | r.h[1].pad) & TOK_AGG_COMPLEX) {       // rare: counted again, byte by byte
                a.nl = 0; a.tabs = 0; a.last_nl = 0;
                const size_t b0 = (size_t)i * TOK2_TILE;
                for (int j = 0; j < TOK2_TILE && b0 + j < n; ++j) {
                    const char ch = text[b0 + j];
                    if (ch == '\n') { a.nl++; a.tabs = 0; a.last_nl = j + 1; } else if (ch == '\t') a.tabs++;
                }
            } else if (r.h[1].nl > 0) { a.nl = r.h[0].nl + r.h[1].nl; a.tabs = r.h[1].tabs; a.last_nl = r.h[1].last_nl; }
            else { a.nl = r.h[0].nl; a.tabs = r.h[0].tabs + r.h[1].tabs; a.last_nl = r.h[0].last_nl; }
        }
        me.lines = a.nl; me.tabs = a.tabs; me.ls = a.last_nl > 0 ? (long long)i * TOK2_TILE + a.last_nl : -1;
    }
    TokState inc = me;
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

// line_off counts from the first tile's start: moved to the window's start (skip bytes further on)
static __global__ void k_tok_grid_finish(unsigned long long *__restrict__ line_off, const int *__restrict__ n_lines, int max_lines, unsigned skip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *n_lines < max_lines ? *n_lines : max_lines;
    if (i <= n) line_off[i] -= skip;
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE sweep of the text (round 3): k_tok_parse3 = count + scan + parse in one kernel, the states at the segments' starts by
// decoupled look-back.  A workgroup takes a SEGMENT of TOK3_TILES tiles (32 KiB): every thread loads its 32 + 8 bytes of each
// of the four tiles ONCE, into registers, and keeps them (and their TAB / newline masks) for both phases:
//   1. the tiles' aggregates (newlines, TABs behind the last newline, where that newline is -- what k_tok_count2 writes) by
//      workgroup reductions; their fold is the segment's aggregate, PUBLISHED at once: it depends on no other workgroup;
//   2. look-back in two levels.  64 segments form a SUPER (2 MB of text) with a record of its own: its aggregate is published by
//      its last segment's workgroup as soon as that has read the other 63 aggregates (which it needs anyway), its prefix when that
//      workgroup has its own.  A workgroup reads the aggregates of the segments between its super's start and itself (wave 0:
//      at most 63 records) and the supers' records backwards from there (waves 1 - 3: 192 records, more rounds if none of them
//      carries a prefix yet), folds them in order onto the nearest prefix (the combine is associative, not commutative: a
//      newline resets the TAB count) -- about a hundred 16-byte records per 32 KiB of text, one round as a rule;
//   3. the four tiles are parsed by tok_parse_tile, each from its own start state.
// Why two levels: prefixes spread by one look-back's reach per round trip (4 - 10 us with agent-scope loads and a workgroup
// fold).  The text moves at ~60 segments per us, a one-level look-back of 256 segment records falls behind -- every segment looks
// back further than the one before it (measured: 1.45 ms for 640 MB, three times the two-sweep form) -- and reading a thousand
// records per segment would be half the text's bytes again.  With supers the reach of one round is 192 x 2 MB.
// Records: two 64-bit words per segment, each carrying the record's status in its top two bits (0 nothing, 1 aggregate,
// 2 prefix), written and read with relaxed agent-scope atomics (a record is only what these two loads return: no other
// memory is handed over, so no fence); a reader takes a record when both words show the same status.
// Segments are numbered by a ticket taken at the workgroup's start, so every segment a workgroup waits for has started: it
// publishes its aggregate without waiting for anyone.  Every wait is bounded all the same: a workgroup that gives up raises the
// error flag, the kernel drains, k_tok_finish reports -1 lines, and the caller falls back to the two-sweep kernels.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TOK3_TILES = 2;
constexpr int TOK3_TILE = 256 * TOK2_TB;                            // this form's tile: 8 KiB, the 256 threads of its workgroup
constexpr int TOK3_SEG = TOK3_TILES * TOK3_TILE;
constexpr int TOK3_SUPER = 64;                      // segments per super (2 MB of text)
struct TokRec { unsigned long long w0, w1; };           // w0: status:2 | lines:31 | tabs:31;  w1: status:2 | (line start + 1):62 (0: no newline)
__device__ __forceinline__ void tok_rec_store(TokRec *r, unsigned status, const TokState st) {
    const unsigned long long w0 = ((unsigned long long)status << 62) | ((unsigned long long)(uint32_t)st.lines << 31) | (unsigned long long)(uint32_t)st.tabs;
    const unsigned long long w1 = ((unsigned long long)status << 62) | (unsigned long long)(st.ls + 1);
    __hip_atomic_store(&r->w0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&r->w1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// status of the record (0 while nothing consistent is there), the state in *st
__device__ __forceinline__ unsigned tok_rec_load(const TokRec *r, TokState *st) {
    const unsigned long long w0 = __hip_atomic_load(&r->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(&r->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned s0 = (unsigned)(w0 >> 62), s1 = (unsigned)(w1 >> 62);
    if (s0 != s1) return 0u;
    st->lines = (int)((w0 >> 31) & 0x7FFFFFFFu); st->tabs = (int)(w0 & 0x7FFFFFFFu);
    st->ls = (long long)(w1 & 0x3FFFFFFFFFFFFFFFull) - 1;
    return s0;
}
// (has a prefix been reached, state): `older` happened before `newer`; a prefix in `newer` supersedes everything older
struct TokLook { int pre; TokState st; };
__device__ __forceinline__ TokLook tok_look_fold(const TokLook older, const TokLook newer) {
    if (newer.pre) return newer;
    TokLook r; r.pre = older.pre; r.st = tok_fold(older.st, newer.st);
    return r;
}

static __global__ __launch_bounds__(256) void k_tok_parse3(const char *__restrict__ text, size_t n, TokRec *__restrict__ rec, TokRec *__restrict__ super_rec,
                                                    unsigned *__restrict__ ticket,
                                                    int *__restrict__ err, int *__restrict__ n_lines,
                                                    int max_lines, int n_samples, int strict, uint8_t *__restrict__ gt, size_t pitch,
                                                    uint8_t *__restrict__ is_x, unsigned long long *__restrict__ line_off,
                                                    uint32_t *__restrict__ field_off, int *__restrict__ status, int *__restrict__ redo, int *__restrict__ redo_n) {
    __shared__ TokShared S;
    __shared__ int s_nl[TOK3_TILES][4], s_last[TOK3_TILES][4], s_tabs[TOK3_TILES][4];
    __shared__ unsigned s_seg;
    __shared__ int l_pre[4], l_lines[4], l_tabs[4];
    __shared__ long long l_ls[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // The segment is the workgroup's index: workgroups are dispatched in index order, so every segment this one waits for has
    // started (and publishes without waiting for anyone).  A ticket taken at the start (atomicAdd on one counter) would make that
    // order certain instead of customary -- and cost more than the whole kernel: 19 550 agent-scope atomics on one address are
    // performed one after the other at the memory side, 1.4 ms for 640 MB of text.  Should the order ever fail, the bounded waits
    // below end the kernel with the error flag set and the caller goes back to two sweeps.
    (void)ticket; (void)s_seg;
    const unsigned seg = blockIdx.x;
    const size_t seg_base = (size_t)seg * TOK3_SEG;
    if (seg_base >= n) return;                                       // (the grid is exact: cannot happen)

    // ---- 1. the segment's bytes (once) and its tiles' aggregates -------------------------------------------------------
    TokThread T[TOK3_TILES];
    int last_bit[TOK3_TILES];
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) tok_read(text, seg_base + (size_t)k * TOK3_TILE + (size_t)tid * TOK2_TB, n, T[k]);
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) {
        const int nl = __popc(T[k].nls);
        last_bit[k] = nl ? 31 - __clz((int)T[k].nls) : -1;
        int c = nl, key = nl ? tid : -1;
        for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off); const int k2 = __shfl_xor(key, off); key = k2 > key ? k2 : key; }
        if (lane == 0) { s_nl[k][w] = c; s_last[k][w] = key; }
    }
    __syncthreads();
    int tlast[TOK3_TILES];
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) {
        tlast[k] = max(max(s_last[k][0], s_last[k][1]), max(s_last[k][2], s_last[k][3]));
        int mine = tid > tlast[k] ? __popc(T[k].tabs) : (tid == tlast[k] ? tok_tabs_after(T[k].tabs, last_bit[k]) : 0);
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) s_tabs[k][w] = mine;
    }
    __syncthreads();
    // every thread folds the four tile aggregates (same values everywhere): rel[k] = the state at tile k's start relative to the segment's
    TokState rel[TOK3_TILES + 1];
    rel[0].lines = 0; rel[0].tabs = 0; rel[0].ls = -1;
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) {
        TokState a;
        a.lines = s_nl[k][0] + s_nl[k][1] + s_nl[k][2] + s_nl[k][3];
        a.tabs = s_tabs[k][0] + s_tabs[k][1] + s_tabs[k][2] + s_tabs[k][3];
        // where the tile's last newline is: known to the thread that holds it; the others take it from that thread's lane... every
        // thread can compute it from tlast and that thread's last_bit only if it IS that thread, so it goes through LDS below
        a.ls = -1;
        rel[k + 1] = a;                                              // (ls filled in below)
    }
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k)
        if (tid == tlast[k]) s_last[k][0] = last_bit[k];             // (s_last is free again: every thread has read it)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) {
        TokState a = rel[k + 1];
        a.ls = tlast[k] >= 0 ? (long long)(seg_base + (size_t)k * TOK3_TILE + (size_t)tlast[k] * TOK2_TB + (size_t)s_last[k][0] + 1) : -1;
        rel[k + 1] = tok_fold(rel[k], a);
    }
    const TokState seg_agg = rel[TOK3_TILES];
    if (tid == 0) tok_rec_store(&rec[seg], 1u, seg_agg);

    // ---- 2. look-back, two levels: the segments of this segment's SUPER (64 segments, 2 MB) back to the super's start, and
    //         from there the supers' own records back to one that carries a prefix.  One round as a rule: 63 + 192 records. ------
    const long long sup = (long long)(seg / TOK3_SUPER);
    const int r_in = (int)(seg % TOK3_SUPER);                        // segments of this super in front of this one
    TokLook acc; acc.pre = 0; acc.st.lines = 0; acc.st.tabs = 0; acc.st.ls = -1;      // what lies between the prefix found so far and this segment
    bool gave_up = false;
    long long next_super = sup - 1;                                  // the nearest super not yet read
    for (int round = 0; !acc.pre; ++round) {
        // round 0: wave 0 reads the segments seg - 1 .. of this super, waves 1 - 3 the supers sup - 1 ..; later rounds: supers only.
        // Thread order = age: tid 0 the nearest record, tid 255 the oldest.
        const TokRec *src = nullptr;
        TokLook me; me.pre = 0; me.st.lines = 0; me.st.tabs = 0; me.st.ls = -1;      // (no record: the identity)
        if (round == 0 && tid < 64) { if (tid < r_in) src = &rec[(long long)seg - 1 - tid]; }
        else {
            // (round 0: 64 supers, wave 1; waves 2 and 3 hold the identity -- the records read are memory traffic too)
            const long long q = next_super - (round == 0 ? tid - 64 : tid);
            if (round == 0 && tid >= 128) { }
            else if (q < 0) { me.pre = 1; me.st.ls = 0; }            // the text's start is a prefix: no lines, no TABs, the line starts at 0
            else src = &super_rec[q];
        }
        if (src) {
            unsigned stt = 0;
            for (int spin = 0; spin < (1 << 20); ++spin) {
                stt = tok_rec_load(src, &me.st);
                if (stt || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!stt) { gave_up = true; me.st.lines = 0; me.st.tabs = 0; me.st.ls = -1; me.pre = 1; }
            else me.pre = stt == 2u;
        }
        // ordered fold over the 256 threads, oldest (tid 255) first: within the wave by shuffles from the higher lanes, across the waves in LDS
        TokLook inc = me;
        for (int off = 1; off < 64; off <<= 1) {
            TokLook o;
            o.pre = __shfl_down(inc.pre, off); o.st.lines = __shfl_down(inc.st.lines, off); o.st.tabs = __shfl_down(inc.st.tabs, off); o.st.ls = __shfl_down(inc.st.ls, off);
            if (lane + off < 64) inc = tok_look_fold(o, inc);        // lane + off is OLDER
        }
        // the super's last segment: wave 0 has just read the aggregates of all the others -- the super's aggregate goes out NOW, before
        // the workgroup's barrier, which waits for the waves that are still waiting for OLDER supers (behind the barrier every
        // super's aggregate would wait for the one before it: a chain through all 300 supers of a 640 MB text, 1.6 ms)
        if (round == 0 && tid == 0 && r_in == TOK3_SUPER - 1) tok_rec_store(&super_rec[sup], 1u, tok_fold(inc.st, seg_agg));
        if (lane == 0) { l_pre[w] = inc.pre; l_lines[w] = inc.st.lines; l_tabs[w] = inc.st.tabs; l_ls[w] = inc.st.ls; }
        __syncthreads();
        TokLook rnd; rnd.pre = l_pre[3]; rnd.st.lines = l_lines[3]; rnd.st.tabs = l_tabs[3]; rnd.st.ls = l_ls[3];
        for (int k = 2; k >= 0; --k) { TokLook nw; nw.pre = l_pre[k]; nw.st.lines = l_lines[k]; nw.st.tabs = l_tabs[k]; nw.st.ls = l_ls[k]; rnd = tok_look_fold(rnd, nw); }
        acc = tok_look_fold(rnd, acc);                               // this round's records are older than what has been folded
        next_super -= round == 0 ? 64 : 256;
        __syncthreads();
    }
    if (__syncthreads_or(gave_up ? 1 : 0)) { if (tid == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    const TokState start = acc.st;                                   // the state at the segment's first byte
    const TokState incl = tok_fold(start, seg_agg);
    if (tid == 0 && r_in == TOK3_SUPER - 1) tok_rec_store(&super_rec[sup], 2u, incl);      // the prefix at the next super's start
    if (seg_base + TOK3_SEG >= n && tid == 0) {                      // the text's last segment: the line count (k_tok_scan2b's part)
        const int tail = (n > 0 && text[n - 1] != '\n') ? 1 : 0;     // unterminated last line
        *n_lines = incl.lines + tail;
        if (tail && incl.lines + 1 <= max_lines) line_off[incl.lines + 1] = n;
    }

    // ---- 3. the tiles ------------------------------------------------------------------------------------------------------
    const TokOut O = {max_lines, n_samples, strict, gt, pitch, is_x, line_off, field_off, status, redo, redo_n};
#pragma unroll
    for (int k = 0; k < TOK3_TILES; ++k) {
        const size_t tile_base = seg_base + (size_t)k * TOK3_TILE;
        if (tile_base >= n) break;                                   // (uniform)
        const TokState st = tok_fold(start, rel[k]);
        TokPre P; P.lines = st.lines; P.tabs = st.tabs; P.line_start = (unsigned long long)st.ls;
        tok_parse_tile(text, n, tile_base, seg == 0 && k == 0, P, T[k], S, O);
    }
}
// the fused kernel's verdict: a look-back that gave up makes the line count -1 (the caller runs the two-sweep kernels instead)
static __global__ void k_tok_finish(const int *__restrict__ err, int *__restrict__ n_lines) { if (*err) *n_lines = -1; }

}  // namespace hpgv
