// hpgv_crc_kernels.h -- CRC-32 of decoded BGZF blocks on the device (the check htslib's bgzf reader and zlib's gzread make on
// every block: a stream that inflates to ISIZE bytes of the WRONG text must not reach the statistics).
//
// One wave per block.  CRC-32 is linear over GF(2): the register after a message is the XOR of what every piece of the message
// contributes, each piece advanced over the bytes that follow it.  So the 64 lanes read the text coalesced -- lane i takes
// dwords i, i + 64, i + 128, ... -- and each keeps a register of its own: XOR the dword in, advance by 256 bytes (the distance to
// the lane's next dword) with four table look-ups, exactly as slice-by-4 advances by 4 bytes, only with tables built for
// x^2048 instead of x^32.  At the end lane i's register stands 64 - i dwords before the end of the interleaved part: it is
// advanced by multiplying with x^(32 (64 - i)) mod P, one conditional multiplication per bit of 64 - i with the constants
// x^(32 * 2^b), and the 64 registers are XORed together.  The bytes in front of the first aligned dword (at most three) and
// behind the last full 256-byte row are taken byte by byte.  Reflected polynomial 0xEDB88320, as zlib.
//
// Round 4: the same wave, which has just read its block's text, also leaves the TOKENIZER's tile records of that text
// (k_bgzf_crc<true>: TokAgg2 per 2 KiB tile of the absolute grid over the decoded text, hpgv_text2_kernels.h) -- the second read
// comes out of the L2, and the tokenizer's counting sweep over windows of this text is not run at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hpgv_text2_kernels.h"

namespace hpgv {

enum { CRC_T0 = 0, CRC_W = 256, CRC_X2N = 256 + 1024, CRC_TAB_WORDS = 256 + 1024 + 8 };
enum { BGZF_STATUS_BAD_CRC = 9 };       // status of a block whose text does not have the CRC-32 its trailer gives

// a * b mod P, reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

// host: t0 = the byte table; w[j][b] = (b << 8 j) advanced by 256 bytes; x2n[b] = x^(32 * 2^b), b = 0 .. 6
inline void crc_build_tables(uint32_t *tab) {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t c = b;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        tab[CRC_T0 + b] = c;
    }
    uint32_t x = 0x40000000u;                                   // x^1
    uint32_t x2n[16];
    for (int k = 0; k < 16; ++k) { x2n[k] = x; x = crc_multmodp(x, x); }      // x^(2^k)
    for (int b = 0; b < 7; ++b) tab[CRC_X2N + b] = x2n[5 + b];  // x^(32 * 2^b)
    tab[CRC_X2N + 7] = 0;
    const uint32_t x2048 = x2n[11];
    for (int j = 0; j < 4; ++j)
        for (uint32_t b = 0; b < 256; ++b) tab[CRC_W + 256 * j + b] = crc_multmodp(x2048, b << (8 * j));
}

// the tile records of block b's text [o, o + len): per tile of the absolute grid the half this block owns -- h[0] when the block
// holds the tile's first byte, h[1] when it begins inside the tile.  A second block that begins inside the same tile (blocks
// shorter than a tile) marks the tile TOK_AGG_COMPLEX: the tokenizer's scan counts such a tile again.  One wave.
__device__ __forceinline__ void bgzf_tile_store(TokAgg2 *__restrict__ agg2, size_t tile, bool first_half, const TokAgg &a) {
    if (first_half) {                                                // (the flags only ever by atomics: another block may be marking the tile)
        agg2[tile].h[0].nl = a.nl; agg2[tile].h[0].tabs = a.tabs; agg2[tile].h[0].last_nl = a.last_nl;
        atomicOr(&agg2[tile].h[0].pad, (int)TOK_AGG_WRITTEN);
    } else {
        // this block begins inside the tile: it owns h[1] -- unless a block before it did too
        const int old = atomicOr(&agg2[tile].h[1].pad, (int)TOK_AGG_WRITTEN);
        if (old & TOK_AGG_WRITTEN) atomicOr(&agg2[tile].h[0].pad, (int)TOK_AGG_COMPLEX);
        else { agg2[tile].h[1].nl = a.nl; agg2[tile].h[1].tabs = a.tabs; agg2[tile].h[1].last_nl = a.last_nl; }
    }
}
__device__ __forceinline__ void bgzf_block_agg(const char *__restrict__ text, size_t o, size_t len, TokAgg2 *__restrict__ agg2, long agg_tiles) {
    if (len == 0) return;
    const int lane = threadIdx.x & 63;
    const size_t end = o + len;
    size_t t_lo = o / TOK2_TILE, t_hi = (end - 1) / TOK2_TILE;        // first and last tile the block touches
    if ((long)t_lo >= agg_tiles) return;
    if ((long)t_hi >= agg_tiles) t_hi = (size_t)agg_tiles - 1;
    // the tiles at the block's ends: clipped to the block (the first one is the other block's tile when the block begins inside it)
    {
        const size_t tb = t_lo * TOK2_TILE, hi = end < tb + TOK2_TILE ? end : tb + TOK2_TILE;
        const TokAgg a = tok_wave_agg(text, tb, o > tb ? o : tb, hi);
        if (lane == 63) bgzf_tile_store(agg2, t_lo, tb >= o, a);
    }
    if (t_hi > t_lo && end < (t_hi + 1) * TOK2_TILE) {
        const TokAgg a = tok_wave_agg(text, t_hi * TOK2_TILE, t_hi * TOK2_TILE, end);
        if (lane == 63) bgzf_tile_store(agg2, t_hi, true, a);
        --t_hi;
    }
    // the tiles in between lie whole inside the block: the next tile's 32 bytes per lane are on their way while this one is counted
    if (t_hi <= t_lo) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(text + (t_lo + 1) * TOK2_TILE + (size_t)lane * TOK2_TB);      // (a tile is 16-byte aligned in the text... see below)
    const bool aligned = ((uintptr_t)text & 15u) == 0;
    uint4 n0, n1;
    if (aligned) { n0 = p[0]; n1 = p[1]; }
    for (size_t tile = t_lo + 1; tile <= t_hi; ++tile) {
        int nl, tabs_after, last_bit;
        if (aligned) {
            const uint4 c0 = n0, c1 = n1;
            if (tile < t_hi) { p += TOK2_TILE / 16; n0 = p[0]; n1 = p[1]; }
            const uint32_t d[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            tok_count_regs(d, &nl, &tabs_after, &last_bit);
        } else tok_count_thread(text, tile * TOK2_TILE + (size_t)lane * TOK2_TB, end, &nl, &tabs_after, &last_bit);
        const TokAgg a = tok_wave_fold(nl, tabs_after, last_bit);
        if (lane == 63) bgzf_tile_store(agg2, tile, true, a);
    }
}
// text of block b that the decoder did not write (status != 0: the host decodes it and patches the text): its tiles are counted
// again by the tokenizer
__device__ __forceinline__ void bgzf_block_agg_void(size_t o, size_t len, TokAgg2 *__restrict__ agg2, long agg_tiles) {
    if (len == 0) return;
    const size_t t_lo = o / TOK2_TILE, t_hi = (o + len - 1) / TOK2_TILE;
    for (size_t tile = t_lo + (threadIdx.x & 63); tile <= t_hi && (long)tile < agg_tiles; tile += 64) atomicOr(&agg2[tile].h[0].pad, (int)TOK_AGG_COMPLEX);
}

// a tile that lies whole inside the block, from what the CRC loop saw of it: the lanes' TAB counts and newline flags.  Without a
// newline (nineteen tiles in twenty at 10 k samples) the record is one DPP sum; with one the tile's 32 bytes per lane are read
// again, in the tokenizer's order, out of the cache the loop has just filled.
__device__ __forceinline__ void bgzf_tile_finish(const char *__restrict__ text, size_t tile, int tabs_l, uint32_t nl_l, TokAgg2 *__restrict__ agg2) {
    const int lane = threadIdx.x & 63;
    TokAgg a;
    if (__ballot(nl_l != 0u) == 0ull) a = tok_wave_fold(0, tabs_l, -1);
    else {
        const uint4 *p = reinterpret_cast<const uint4 *>(text + tile * TOK2_TILE + (size_t)lane * TOK2_TB);
        const uint4 c0 = p[0], c1 = p[1];
        const uint32_t d[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        int nl, tabs_after, last_bit;
        tok_count_regs(d, &nl, &tabs_after, &last_bit);
        a = tok_wave_fold(nl, tabs_after, last_bit);
    }
    if (lane == 63) bgzf_tile_store(agg2, tile, true, a);
}
// the tiles [t_from, t_to] of the block's text [o, end), clipped to it, one at a time (the few at the block's ends)
__device__ __forceinline__ void bgzf_tiles_clipped(const char *__restrict__ text, size_t o, size_t end, size_t t_from, size_t t_to,
                                                   TokAgg2 *__restrict__ agg2, long agg_tiles) {
    const int lane = threadIdx.x & 63;
    for (size_t tile = t_from; tile <= t_to && (long)tile < agg_tiles; ++tile) {
        const size_t tb = tile * TOK2_TILE, hi = end < tb + TOK2_TILE ? end : tb + TOK2_TILE;
        const TokAgg a = tok_wave_agg(text, tb, o > tb ? o : tb, hi);
        if (lane == 63) bgzf_tile_store(agg2, tile, tb >= o, a);
    }
}

template <bool AGG>
static __global__ __launch_bounds__(256) void k_bgzf_crc(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                   const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                   const uint32_t *__restrict__ out_len, int n_blocks,
                                                   const uint8_t *__restrict__ text, int32_t *__restrict__ status,
                                                   const uint32_t *__restrict__ tab, TokAgg2 *__restrict__ agg2, long agg_tiles) {
    __shared__ uint32_t s_tab[CRC_TAB_WORDS];
    for (int i = threadIdx.x; i < CRC_TAB_WORDS; i += 256) s_tab[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // (one per wave: its offsets and lengths are scalars)
    if (b >= n_blocks) return;
    if (status[b] != 0) { if constexpr (AGG) bgzf_block_agg_void((size_t)out_off[b], out_len[b], agg2, agg_tiles); return; }
    // the tile records come out of the CRC loop's own loads when the text is 16-byte aligned (every buffer of the runtime is);
    // otherwise out of a sweep of their own
    const bool fused = AGG && ((uintptr_t)text & 15u) == 0 && out_len[b] != 0;
    if constexpr (AGG) if (!fused) bgzf_block_agg((const char *)text, (size_t)out_off[b], out_len[b], agg2, agg_tiles);
    const size_t o = (size_t)out_off[b];
    const uint8_t *p = text + o;
    uint32_t L = out_len[b];
    const size_t end = o + L;
    const uint8_t *t = comp + in_off[b] + in_len[b];            // the block's trailer: CRC32, ISIZE
    const uint32_t stored = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t *t0 = s_tab + CRC_T0;
    uint32_t s = 0xFFFFFFFFu;
    uint32_t head = (uint32_t)((4 - ((uintptr_t)p & 3)) & 3);
    if (head > L) head = L;
    for (uint32_t i = 0; i < head; ++i) s = t0[(s ^ p[i]) & 0xFFu] ^ (s >> 8);
    p += head; L -= head;
    const uint32_t steps = L >> 8;                              // dwords per lane
    // Rows are taken eight at a time: 2 KiB that straddle the SAME point of the tile grid in every group of the block -- the
    // first c bytes of a group end tile T0 + g, the rest begin the next.  Group g therefore completes tile T0 + g (from g = 1
    // on, or from 0 when the groups ARE tiles); the tiles at the block's ends are counted apart, clipped.
    const size_t P = o + head;
    const uint32_t c = (uint32_t)TOK2_TILE - (uint32_t)(P & (TOK2_TILE - 1));
    const size_t T0 = P / TOK2_TILE;
    const uint32_t ng = fused ? steps >> 3 : 0u, g_a = c == (uint32_t)TOK2_TILE ? 0u : 1u;
    size_t t_lo = 0, t_hi = 0;
    if constexpr (AGG) if (fused) {
        t_lo = o / TOK2_TILE; t_hi = (end - 1) / TOK2_TILE;
        const size_t upto = ng > g_a ? T0 + g_a : t_hi + 1;           // the first tile the loop completes (none: all of them here)
        if (upto > t_lo) bgzf_tiles_clipped((const char *)text, o, end, t_lo, upto - 1 < t_hi ? upto - 1 : t_hi, agg2, agg_tiles);
    }
    if (steps) {
        const uint32_t *q = (const uint32_t *)p + lane;
        const uint32_t *w0 = s_tab + CRC_W, *w1 = w0 + 256, *w2 = w0 + 512, *w3 = w0 + 768;
        uint32_t a = lane == 0 ? s : 0u;
        uint32_t k = 0;
        if constexpr (AGG) if (ng) {
            // the grid's point falls in row jc of a group, cin bytes into it (jc = 8: the groups are tiles) -- only in that row do
            // the lanes stand on different sides of it.  A newline is looked for, not counted: (x - 01..) & ~x has its 0x80 bits
            // set somewhere exactly when a byte of x is zero.
            const uint32_t myoff = 4u * (uint32_t)lane, jc = c >> 8, cin = c & 255u;
            const bool early = myoff < cin;
            int carry_t = 0; uint32_t carry_n = 0;
            uint32_t d[8], nx[8];
            #pragma unroll
            for (int j = 0; j < 8; ++j) { d[j] = q[64 * j]; nx[j] = 0; }
            for (uint32_t g = 0; g < ng; ++g) {
                {   // the next group's rows are on their way while this one's are worked on (always eight loads -- the last
                    // group asks for itself again -- so that the waits can be counted: a load that may or may not have been
                    // issued makes every wait a wait for everything)
                    const uint32_t *qn = q + 512 * (size_t)(g + 1 < ng ? g + 1 : g);
                    #pragma unroll
                    for (int j = 0; j < 8; ++j) nx[j] = qn[64 * j];
                }
                int acc_t = 0, te = 0; uint32_t acc_n = 0, ne = 0;
                #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t x = d[j] ^ 0x0A0A0A0Au, hn = (x - 0x01010101u) & ~x;
                    const int pc = __popc(tok_eq_flags(d[j], 0x09090909u));
                    if ((uint32_t)j == jc) {
                        te = acc_t + (early ? pc : 0); ne = acc_n | (early ? hn : 0u);
                        acc_t = early ? 0 : pc; acc_n = early ? 0u : hn;
                    } else { acc_t += pc; acc_n |= hn; }
                    a ^= d[j];
                    if (j < 7 || 8 * g + 8 < steps) a = w0[a & 0xFFu] ^ w1[(a >> 8) & 0xFFu] ^ w2[(a >> 16) & 0xFFu] ^ w3[a >> 24];
                }
                if (jc == 8u) { te = acc_t; ne = acc_n; acc_t = 0; acc_n = 0; }
                if (g >= g_a && (long)(T0 + g) < agg_tiles)
                    bgzf_tile_finish((const char *)text, T0 + g, carry_t + te, (carry_n | ne) & 0x80808080u, agg2);
                carry_t = acc_t; carry_n = acc_n;
                #pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = nx[j];
            }
            k = 8 * ng;
        }
        if (k < steps) {                                         // (the rows that do not fill a group; every row without the records)
            uint32_t d = q[64 * (size_t)k];
            for (++k; k < steps; ++k) {
                const uint32_t nx = q[64 * (size_t)k];
                a ^= d;
                a = w0[a & 0xFFu] ^ w1[(a >> 8) & 0xFFu] ^ w2[(a >> 16) & 0xFFu] ^ w3[a >> 24];
                d = nx;
            }
            a ^= d;
        }
        // a stands 64 - lane dwords before the end of the rows
        const uint32_t adv = 64u - (uint32_t)lane;
        #pragma unroll 1
        for (int bit = 0; bit < 7; ++bit)
            if ((adv >> bit) & 1u) a = crc_multmodp(s_tab[CRC_X2N + bit], a);
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) a ^= __shfl_xor(a, off);
        s = a;
        p += (size_t)steps << 8; L -= steps << 8;
    }
    for (uint32_t i = 0; i < L; ++i) s = t0[(s ^ p[i]) & 0xFFu] ^ (s >> 8);
    if (lane == 0 && ~s != stored) status[b] = BGZF_STATUS_BAD_CRC;
    if constexpr (AGG) if (fused && ng > g_a) {                  // the tiles behind the last one the loop completed
        const size_t from = T0 + ng;
        if (from <= t_hi) bgzf_tiles_clipped((const char *)text, o, end, from, t_hi, agg2, agg_tiles);
    }
}

}  // namespace hpgv
