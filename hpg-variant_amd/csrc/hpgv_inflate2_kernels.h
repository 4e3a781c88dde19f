// hpgv_inflate2_kernels.h -- raw DEFLATE (RFC 1951) on the GPU, one WAVE per BGZF block (`--compression bgzip`,
// shared_options.c:60-61).
//
// The lane-per-block decoder (hpgv_inflate_kernels.h) keeps every lane's code tables in private memory and writes its
// text a byte at a time 64 KiB apart from its neighbours: it moves ten times the bytes it decodes and a block takes a lane
// 38 ms.  Here the 64 lanes of a wave work on ONE block:
//   * the headers' bits come in through the scalar cache, a dword at a time, two loads ahead of a wave-uniform bit buffer;
//   * the Huffman codes live in LDS as look-up tables (8 bits for literals / lengths, 8 for distances; a longer code takes
//     the canonical walk over the code's length histogram), built by the whole wave: every lane decodes its share of the
//     table's indices with that same walk;
//   * the decoding: SEVERAL symbols per round of the loop (MULTI, the default).  The next 2 048 bits of the stream lie in
//     one vector register; every lane takes the 64 bits from "position + lane" on out of it and decodes the symbol that
//     would start there (two LDS reads for the whole wave), and the lanes on the true chain -- lane 0, the lane its symbol
//     ends at, ... -- are visited one after the other with two v_readlane and a bit test each: 25 scalar + 24 vector
//     instructions per symbol.  (The form before it, kept for A/B: ONE symbol per round, serial and uniform -- scalar
//     registers and scalar branches; one LDS read serves a length code and the distance code behind it: lane 0 reads the
//     literal / length table at the buffer's low bits while lane i reads the distance table at the bits from i on, and the
//     entry of the lane the length code ends at is picked with v_readlane: 41 + 39 instructions per symbol.)
//   * the last 4 KiB of the block's text are also kept in LDS (a ring): nineteen matches in twenty of genotype text reach
//     back less than that (zlib follows its hash chains from the nearest candidate), so a match is copied by the lanes
//     side by side out of LDS -- byte k from k mod distance -- into the ring and into global memory, and the wave never
//     waits for global memory; the ring write and the store of a match's last 64 bytes are held back until the next symbol
//     has been decoded, so the LDS read's latency lies behind that symbol's table look-up.  A match from further back
//     reads the block's text in global memory: loads and stores of one wave to global memory are performed in program
//     order, which is all a match reading the wave's own output needs;
//   * NOTHING in the kernel branches on a lane's value: a lane that has nothing to write is pointed at 256 spare bytes of
//     LDS, or at an offset that the buffer descriptor of the block's text rejects.  With only wave-uniform branches the
//     compiler leaves the control flow as written (scalar compares and branches) instead of structurizing it.
// A block is decoded in a fraction of a millisecond, and only its compressed bytes, its text and the far match sources
// move.  Same contract as the lane kernel: anything irregular ends with a non-zero status and the host decodes that
// block.  The compressed bytes must be readable up to 4 bytes past the last block's end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

enum { INF2_ROOT_L = 8, INF2_ROOT_D = 8, INF2_ROOT_C = 7 };
// table entry: bits 0-3 code length (0 = not in the table), 4-7 extra bits, 8-9 kind (0 literal, 1 base + extra bits,
// 2 end of block, 3 not a symbol of the format), 10-14 code length + extra bits, 16-31 literal / base
// LDS map (bytes); the code-length code's tables lie where the literal / length table is built afterwards
enum { INF2_LUT_L = 0, INF2_LUT_D = INF2_LUT_L + (4 << INF2_ROOT_L), INF2_LUT_C = INF2_LUT_L, INF2_SYM_C = INF2_LUT_C + (4 << INF2_ROOT_C),
       INF2_CNT_C = INF2_SYM_C + 64, INF2_CLEN = INF2_CNT_C + 32,
       INF2_SYM_L = INF2_LUT_D + (4 << INF2_ROOT_D), INF2_SYM_D = INF2_SYM_L + 576, INF2_CNT_L = INF2_SYM_D + 64, INF2_CNT_D = INF2_CNT_L + 32,
       INF2_LENS = INF2_CNT_D + 32, INF2_RUN = INF2_LENS + 384, INF2_DUMP = INF2_RUN + 32, INF2_RING = INF2_DUMP + 256,
       INF2_WINDOW = 4096, INF2_LDS = INF2_RING + INF2_WINDOW };
static_assert(INF2_CLEN + 32 <= INF2_LUT_D, "the code-length code's tables lie inside the literal / length table's room");

__device__ __forceinline__ uint32_t inf2_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// LDS accessors by byte offset; `on ? at : spare` keeps a lane's store without a branch
__device__ __forceinline__ uint32_t inf2_sel(bool on, uint32_t at) { return on ? at : (uint32_t)INF2_DUMP + (threadIdx.x << 2); }
__device__ __forceinline__ uint32_t inf2_r8(const uint8_t *smem, uint32_t at) { return smem[at]; }
__device__ __forceinline__ uint32_t inf2_r16(const uint8_t *smem, uint32_t at) { return *(const uint16_t *)(smem + at); }
__device__ __forceinline__ uint32_t inf2_r32(const uint8_t *smem, uint32_t at) { return *(const uint32_t *)(smem + at); }
__device__ __forceinline__ void inf2_w8(uint8_t *smem, uint32_t at, uint32_t v) { smem[at] = (uint8_t)v; }
__device__ __forceinline__ void inf2_w16(uint8_t *smem, uint32_t at, uint32_t v) { *(uint16_t *)(smem + at) = (uint16_t)v; }
__device__ __forceinline__ void inf2_w32(uint8_t *smem, uint32_t at, uint32_t v) { *(uint32_t *)(smem + at) = v; }

__device__ __forceinline__ uint32_t inf2_entry_litlen(uint32_t sym, uint32_t len) {
    const uint32_t s = sym - 257;
    const bool wide = s >= 8 && s != 28;
    const uint32_t xb = wide ? ((s >> 2) - 1) & 7 : 0u;
    const uint32_t base = s == 28 ? 258u : wide ? 3 + ((4 + (s & 3)) << xb) : 3 + s;
    const uint32_t lit = len | (sym << 16), eob = len | (2u << 8), bad = len | (3u << 8);
    const uint32_t mat = len | (xb << 4) | (1u << 8) | ((len + xb) << 10) | (base << 16);
    return sym < 256 ? lit : sym == 256 ? eob : s >= 29 ? bad : mat;
}
__device__ __forceinline__ uint32_t inf2_entry_dist(uint32_t ds, uint32_t len) {
    const bool wide = ds >= 4;
    const uint32_t xb = wide ? ((ds >> 1) - 1) & 15 : 0u;
    const uint32_t base = wide ? 1 + ((2 + (ds & 1)) << xb) : 1 + ds;
    return ds >= 30 ? len | (3u << 8) : len | (xb << 4) | (1u << 8) | ((len + xb) << 10) | (base << 16);
}

// lanes whose value v (0 .. 15) equals this lane's
__device__ __forceinline__ uint64_t inf2_same(uint32_t v) {
    uint64_t m = ~0ull;
    #pragma unroll
    for (int bit = 0; bit < 4; bit++) {
        const uint64_t bb = __ballot((v >> bit) & 1u);
        m &= ((v >> bit) & 1u) ? bb : ~bb;
    }
    return m;
}

// code lengths at lens .. lens + n (LDS bytes) -> the histogram at cnt (16 x u16) and the symbols ordered by (length,
// symbol) at sym (u16 each); INF2_RUN is scratch.  Returns 0 for a complete code, > 0 for an incomplete one, < 0 for an
// over-subscribed one; *used = symbols with a code
template <int NG>
__device__ __forceinline__ int inf2_sort(uint8_t *smem, uint32_t lens, int n, uint32_t sym, uint32_t cnt, int *used) {
    const uint32_t lane = threadIdx.x;
    inf2_w16(smem, inf2_sel(lane < 16, cnt + 2 * lane), 0);
    uint32_t l[NG];
    #pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t s = g * 64 + lane;
        const uint32_t v = inf2_r8(smem, lens + (s < (uint32_t)n ? s : 0u));
        l[g] = s < (uint32_t)n ? v : 0u;
    }
    const uint64_t lt = (1ull << lane) - 1;
    #pragma unroll
    for (int g = 0; g < NG; g++) {                      // histogram: the first lane of every length adds its lanes
        const uint64_t m = inf2_same(l[g]);
        const bool lead = l[g] != 0 && (int)lane == __ffsll((unsigned long long)m) - 1;
        const uint32_t at = cnt + 2 * l[g];
        inf2_w16(smem, inf2_sel(lead, at), inf2_r16(smem, at) + (uint32_t)__popcll(m));
    }
    const bool mine = lane >= 1 && lane < 16;
    const uint32_t cr = inf2_r16(smem, cnt + 2 * (lane & 15));
    const uint32_t c = mine ? cr : 0u;
    uint32_t kraft = c << ((15 - lane) & 15), nz = c, incl = c;
    #pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        kraft += (uint32_t)__shfl_xor((int)kraft, d);
        nz += (uint32_t)__shfl_xor((int)nz, d);
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        incl += (int)lane >= d ? t : 0u;
    }
    kraft = inf2_u(kraft); nz = inf2_u(nz);
    *used = (int)nz;
    inf2_w16(smem, inf2_sel(lane < 16, (uint32_t)INF2_RUN + 2 * lane), incl - c);       // symbols with a shorter code
    #pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint64_t m = inf2_same(l[g]);
        const bool has = l[g] != 0, lead = has && (int)lane == __ffsll((unsigned long long)m) - 1;
        const uint32_t ra = (uint32_t)INF2_RUN + 2 * l[g];
        const uint32_t at = inf2_r16(smem, ra);
        inf2_w16(smem, inf2_sel(has, sym + 2 * (at + (uint32_t)__popcll(m & lt))), (uint32_t)g * 64 + lane);
        inf2_w16(smem, inf2_sel(lead, ra), at + (uint32_t)__popcll(m));
    }
    if (kraft > (1u << 15)) return -1;
    return nz == 0 ? 0 : (int)((1u << 15) - kraft);
}

// the look-up table of a code: every lane runs the canonical walk on its share of the 2^ROOT bit patterns
template <int ROOT, int KIND>                          // KIND 0 literal / length, 1 distance, 2 code-length code
__device__ __forceinline__ void inf2_fill(uint8_t *smem, uint32_t cnt, uint32_t sym, uint32_t lut) {
    const uint32_t lane = threadIdx.x;
    uint32_t c[ROOT + 1];
    #pragma unroll
    for (int L = 1; L <= ROOT; L++) c[L] = inf2_r16(smem, cnt + 2 * L);
    #pragma unroll 1
    for (uint32_t j = 0; j < (1u << ROOT); j += 64) {
        const uint32_t i = j + lane;
        uint32_t code = 0, first = 0, index = 0, fl = 0, fidx = 0;
        #pragma unroll
        for (int L = 1; L <= ROOT; L++) {
            code |= (i >> (L - 1)) & 1u;
            const uint32_t count = c[L];
            const bool hit = fl == 0 && (int)(code - count) < (int)first;
            fidx = hit ? index + (code - first) : fidx;
            fl = hit ? (uint32_t)L : fl;
            index += count; first += count; first <<= 1; code <<= 1;
        }
        const uint32_t s = inf2_r16(smem, sym + 2 * fidx);
        const uint32_t e = KIND == 0 ? inf2_entry_litlen(s, fl) : KIND == 1 ? inf2_entry_dist(s, fl) : (fl | (s << 16));
        inf2_w32(smem, lut + 4 * i, fl ? e : 0u);
    }
}

// a code longer than the table's index: the canonical walk, uniform, on the low bits of buf (at least 15 valid)
template <int KIND>
__device__ __forceinline__ uint32_t inf2_slow(const uint8_t *smem, uint64_t buf, uint32_t cnt, uint32_t sym) {
    uint32_t code = 0, first = 0, index = 0;
    #pragma unroll 1
    for (int L = 1; L <= 15; L++) {
        code |= (uint32_t)(buf >> (L - 1)) & 1u;
        const uint32_t count = inf2_u(inf2_r16(smem, cnt + 2 * L));
        if ((int)(code - count) < (int)first) {
            const uint32_t s = inf2_u(inf2_r16(smem, sym + 2 * (index + (code - first))));
            return KIND == 0 ? inf2_entry_litlen(s, (uint32_t)L) : inf2_entry_dist(s, (uint32_t)L);
        }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return 0;
}

// the compressed stream of one block: wave-uniform bit buffer fed through the scalar cache, a dword at a time, two dwords
// loaded ahead (a scalar load's result is only waited for when it moves up, one refill later)
struct Inf2In {
    uint64_t buf; int cnt;                             // cnt valid bits in buf
    uint32_t n0, n1;                                   // the next dword, the one after it
    // (the constant address space: the kernel never writes the compressed bytes, and this makes the loads scalar ones)
    const __attribute__((address_space(4))) char *base; uint32_t next_off, last_off;     // byte offsets of dwords from base
    int taken, len_bits, skip;
    __device__ __forceinline__ uint32_t load(uint32_t off) const {
        return *(const __attribute__((address_space(4))) uint32_t *)(base + (off < last_off ? off : last_off));
    }
    // the stream starts at byte `start` of comp and has `len` bytes (len = 0: one dword is read all the same)
    __device__ __forceinline__ void open(const uint8_t *comp, uint64_t start, uint32_t len) {
        const uint64_t a = start & ~3ull;
        base = (const __attribute__((address_space(4))) char *)(uintptr_t)(comp + a);
        skip = (int)(start - a) * 8;                                  // 0 .. 24 bits of the first dword are not the stream's
        const uint32_t n_words = ((uint32_t)(start - a) + len + 3) >> 2;
        last_off = n_words ? (n_words - 1) * 4 : 0;
        len_bits = (int)len * 8;
        n0 = load(0); n1 = load(4); next_off = 8; taken = 0; buf = 0; cnt = 0;
        refill();
        buf >>= skip; cnt -= skip;
        refill();
    }
    // at least 33 valid bits afterwards
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            buf |= (uint64_t)n0 << cnt; cnt += 32;
            n0 = n1; n1 = load(next_off); next_off += 4; taken++;
        }
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1)); buf >>= n; cnt -= n; return v; }
    __device__ __forceinline__ int consumed_bits() const { return taken * 32 - skip - cnt; }
};

// LDS address of text byte p for the lanes that write (on), the spare bytes for the others
__device__ __forceinline__ uint32_t inf2_ring_at(bool on, uint32_t p) { return inf2_sel(on, (uint32_t)INF2_RING + (p & (INF2_WINDOW - 1))); }

// the ring write and the store of the match read last
#define INF2_COMPLETE() do {                                                                                          \
    if (pend_n) {                                                                                                      \
        const bool on_ = lane < pend_n;                                                                                \
        inf2_w8(smem, inf2_ring_at(on_, pend_pos + lane), pend_v);                                                     \
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pend_v, ors, on_ ? (int)(pend_pos + lane) : -1, 0, 0);           \
        pend_n = 0;                                                                                                    \
    }                                                                                                                  \
} while (0)

// a match of `len_` bytes from `dist_` bytes back, as the single-symbol loop below copies it (rc = 17: it reaches back further
// than the text goes, or the text would overflow)
#define INF2_MATCH(len_, dist_) do {                                                                                   \
    const uint32_t ml_ = (len_), md_ = (dist_);                                                                        \
    if (md_ > n_out || ml_ > cap - n_out) { rc = 17; break; }                                                          \
    {                                                                                                                  \
        const bool on_ = lane < pend_n;                                                                                \
        inf2_w8(smem, inf2_ring_at(on_, pend_pos + lane), pend_v);                                                     \
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pend_v, ors, on_ ? (int)(pend_pos + lane) : -1, 0, 0);           \
        pend_n = 0;                                                                                                    \
    }                                                                                                                  \
    const uint32_t dvz_ = md_ | vz, lvz_ = ml_ | vz;                                                                   \
    if (!inf2_u((uint32_t)(dvz_ > (uint32_t)INF2_WINDOW) | (uint32_t)(dvz_ < lvz_) | (uint32_t)(lvz_ > 64u))) {        \
        pend_pos = n_out; pend_n = ml_;                                                                                \
        pend_v = inf2_r8(smem, (uint32_t)INF2_RING + ((n_out - md_ + lane) & (INF2_WINDOW - 1)));                      \
        n_out += ml_;                                                                                                  \
        break;                                                                                                         \
    }                                                                                                                  \
    if (md_ > INF2_WINDOW) {                                                                                           \
        _Pragma("unroll 1")                                                                                            \
        for (uint32_t done_ = 0; done_ < ml_; done_ += 64) {                                                           \
            const uint32_t k_ = done_ + lane;                                                                          \
            const bool on_ = k_ < ml_;                                                                                 \
            const uint8_t v_ = __builtin_amdgcn_raw_buffer_load_b8(ors, on_ ? (int)(n_out - md_ + k_) : -1, 0, 0);     \
            inf2_w8(smem, inf2_ring_at(on_, n_out + k_), v_);                                                          \
            __builtin_amdgcn_raw_buffer_store_b8(v_, ors, on_ ? (int)(n_out + k_) : -1, 0, 0);                         \
        }                                                                                                              \
        n_out += ml_;                                                                                                  \
        break;                                                                                                         \
    }                                                                                                                  \
    const uint32_t from_ = n_out - md_;                                                                                \
    const bool wrap_ = md_ < ml_;                                                                                      \
    const float rcp_ = __builtin_amdgcn_rcpf((float)md_);                                                              \
    uint32_t done_ = 0;                                                                                                \
    _Pragma("unroll 1")                                                                                                \
    for (; done_ + 64 < ml_; done_ += 64) {                                                                            \
        const uint32_t k_ = done_ + lane;                                                                              \
        const uint32_t r_ = wrap_ ? inf2_mod(k_, md_, rcp_) : k_;                                                      \
        const uint32_t v_ = inf2_r8(smem, (uint32_t)INF2_RING + ((from_ + r_) & (INF2_WINDOW - 1)));                   \
        inf2_w8(smem, (uint32_t)INF2_RING + ((n_out + k_) & (INF2_WINDOW - 1)), v_);                                   \
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v_, ors, (int)(n_out + k_), 0, 0);                               \
    }                                                                                                                  \
    {                                                                                                                  \
        const uint32_t k_ = done_ + lane;                                                                              \
        const uint32_t r_ = wrap_ ? inf2_mod(k_, md_, rcp_) : k_;                                                      \
        pend_pos = n_out + done_; pend_n = ml_ - done_;                                                                \
        pend_v = inf2_r8(smem, (uint32_t)INF2_RING + ((from_ + r_) & (INF2_WINDOW - 1)));                              \
    }                                                                                                                  \
    n_out += ml_;                                                                                                      \
} while (0)

// byte k of a match that repeats its own output is byte k mod dist of the dist bytes before it
__device__ __forceinline__ uint32_t inf2_mod(uint32_t k, uint32_t dist, float rcp) {
    const uint32_t q = (uint32_t)((float)k * rcp);
    uint32_t r = k - q * dist;
    r += (int)r < 0 ? dist : 0u;
    r -= r >= dist ? dist : 0u;
    return r;
}

// block b of the table, by the whole wave.  MULTI: several symbols per round of the symbol loop (below)
template <bool MULTI>
__device__ __forceinline__ void inf2_one_block(uint8_t *smem, const int b, const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                               const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                               const uint32_t *__restrict__ out_len, uint8_t *text, int32_t *__restrict__ status) {
    const uint32_t lane = threadIdx.x;
    uint64_t start = in_off[b];                         // of what is left of the stream (a stored block moves it on)
    uint32_t clen_bytes = in_len[b];
    uint8_t *const out0 = text + out_off[b];
    const uint32_t cap = out_len[b];
    // the block's text as a buffer: a store at an offset of cap or more (-1 for the lanes that have nothing to write) is dropped
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)out0, 0, (int)cap, 0x00020000);
    Inf2In B;
    B.open(comp, start, clen_bytes);
    uint32_t n_out = 0;
    int rc = 0, last = 0;
    uint32_t pend_v = 0, pend_pos = 0, pend_n = 0;      // the (last 64 bytes of a) match read from the ring, not yet written
    // the symbol loop's table read, per lane: which table, from which bit of the buffer on, how many index bits
    const uint32_t vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);
    const uint32_t look_base = lane == 0 ? (uint32_t)INF2_LUT_L : (uint32_t)INF2_LUT_D, look_shift = lane & 31,
                   look_mask = lane == 0 ? (1u << INF2_ROOT_L) - 1 : (1u << INF2_ROOT_D) - 1;
    while (!last && !rc) {
        if (B.consumed_bits() > B.len_bits) { rc = 2; break; }
        B.refill();
        last = (int)B.take(1);
        const int type = (int)B.take(2);
        if (type == 3) { rc = 4; break; }
        if (type == 0) {                                             // stored: the bytes follow, byte-aligned
            B.take(B.cnt & 7);
            B.refill();
            const uint32_t len = B.take(16), nlen = B.take(16);
            if ((len ^ nlen) != 0xFFFFu) { rc = 3; break; }
            const int at = B.consumed_bits() >> 3;                   // bytes of the stream before the stored data
            if ((uint32_t)at + len > clen_bytes || n_out + len > cap) { rc = 2; break; }
            INF2_COMPLETE();
            const uint8_t *src = comp + start + at;
            #pragma unroll 1
            for (uint32_t done = 0; done < len; done += 64) {
                const uint32_t k = done + lane;
                const bool on = k < len;
                const uint8_t v = src[on ? k : 0u];
                inf2_w8(smem, inf2_ring_at(on, n_out + k), v);
                __builtin_amdgcn_raw_buffer_store_b8(v, ors, on ? (int)(n_out + k) : -1, 0, 0);
            }
            n_out += len;
            start += (uint64_t)at + len; clen_bytes -= (uint32_t)at + len;
            B.open(comp, start, clen_bytes);
            continue;
        }
        int nlen = 288, ndist = 30;
        if (type == 1) {
            #pragma unroll 1
            for (uint32_t g = 0; g < 5; g++) {
                const uint32_t s = g * 64 + lane;
                inf2_w8(smem, inf2_sel(s < 288, (uint32_t)INF2_LENS + s), s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u);
            }
            inf2_w8(smem, inf2_sel(lane < 30, (uint32_t)INF2_LENS + 288 + lane), 5);
        } else {
            B.refill();
            nlen = (int)B.take(5) + 257; ndist = (int)B.take(5) + 1;
            const int ncode = (int)B.take(4) + 4;
            if (nlen > 286 || ndist > 30) { rc = 5; break; }
            inf2_w8(smem, inf2_sel(lane < 32, (uint32_t)INF2_CLEN + lane), 0);
            __syncthreads();
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            #pragma unroll 1
            for (int k = 0; k < ncode; k++) {
                B.refill();
                inf2_w8(smem, inf2_sel(lane == 0, (uint32_t)INF2_CLEN + order[k]), B.take(3));
            }
            __syncthreads();
            int used = 0;
            if (inf2_sort<1>(smem, INF2_CLEN, 19, INF2_SYM_C, INF2_CNT_C, &used) != 0) { rc = 6; break; }
            __syncthreads();
            inf2_fill<INF2_ROOT_C, 2>(smem, INF2_CNT_C, INF2_SYM_C, INF2_LUT_C);
            __syncthreads();
            int idx = 0;
            uint32_t prev = 0;
            const int total = nlen + ndist;
            while (idx < total) {                                     // the code lengths, run-length coded
                B.refill();
                const uint32_t e = inf2_u(inf2_r32(smem, (uint32_t)INF2_LUT_C + 4 * ((uint32_t)B.buf & ((1u << INF2_ROOT_C) - 1))));
                const int l = (int)(e & 15);
                if (!l) { rc = 7; break; }
                B.buf >>= l; B.cnt -= l;
                const uint32_t sym = e >> 16;
                int rep = 1; uint32_t val = sym;
                if (sym == 16) { if (idx == 0) { rc = 8; break; } val = prev; rep = 3 + (int)B.take(2); }
                else if (sym == 17) { val = 0; rep = 3 + (int)B.take(3); }
                else if (sym == 18) { val = 0; rep = 11 + (int)B.take(7); }
                if (idx + rep > total) { rc = 9; break; }
                #pragma unroll 1
                for (int k = 0; k < rep; k += 64)
                    inf2_w8(smem, inf2_sel((int)lane + k < rep, (uint32_t)INF2_LENS + (uint32_t)(idx + k) + lane), val);
                idx += rep; prev = val;
            }
            if (rc) break;
            __syncthreads();
            if (inf2_u(inf2_r8(smem, INF2_LENS + 256)) == 0) { rc = 10; break; }
        }
        __syncthreads();
        int used_l = 0, used_d = 0;
        const int el = inf2_sort<5>(smem, INF2_LENS, nlen, INF2_SYM_L, INF2_CNT_L, &used_l);
        if (el < 0 || (el > 0 && used_l != 1)) { rc = 11; break; }
        __syncthreads();
        inf2_fill<INF2_ROOT_L, 0>(smem, INF2_CNT_L, INF2_SYM_L, INF2_LUT_L);
        const int ed = inf2_sort<1>(smem, (uint32_t)INF2_LENS + (uint32_t)nlen, ndist, INF2_SYM_D, INF2_CNT_D, &used_d);
        if (type == 2 && (ed < 0 || (ed > 0 && used_d != 1))) { rc = 12; break; }       // the fixed distance code has 30 of its 32 codes
        __syncthreads();
        inf2_fill<INF2_ROOT_D, 1>(smem, INF2_CNT_D, INF2_SYM_D, INF2_LUT_D);
        __syncthreads();
        if (MULTI) {
            // ---- several symbols per round.  The next 2 048 bits of the stream lie in ONE vector register (lane j: dword
            // W0 + j); the 128 bits from the stream's position P on are picked out of it with five v_readlane, and lane i
            // decodes the symbol that would start i bits further on -- its length code, extra bits, distance code, extra bits,
            // all inside its own 64-bit window -- with two LDS reads for the whole wave.  The lanes on the true chain are then
            // visited one after the other (lane 0, then the lane its symbol ends at, ...): two v_readlane per symbol give the
            // scalar unit its length and distance, and the copy is the one of the single-symbol loop.  A symbol that is not an
            // everyday one (a code longer than the table's index, the end of the block, a code that is none) ends the chain
            // and is decoded on its own from the window of lane 0.
            const uint32_t *wsrc = (const uint32_t *)(uintptr_t)B.base;
            const uint32_t w_last = B.last_off;
            const uint32_t gp = (uint32_t)(B.consumed_bits() + B.skip);      // bits from the dword-aligned base on
            uint32_t W0 = gp >> 5, P = gp & 31;
            #define INF2_WLOAD(w) wsrc[(((w) + lane) * 4 < w_last ? ((w) + lane) * 4 : w_last) >> 2]
            uint32_t win = INF2_WLOAD(W0), nxt = INF2_WLOAD(W0 + 32);
            int done = 0;
            #pragma unroll 1
            while (!done) {
                if (P >= 1024) { win = nxt; W0 += 32; P -= 1024; nxt = INF2_WLOAD(W0 + 32); }
                // the 64 bits from bit P + lane on: three dwords of the window register (ds_bpermute: lane q's value), two funnel shifts
                const uint32_t g = P + lane, qa = (g >> 3) & ~3u, r = g & 31;
                const uint32_t da = (uint32_t)__builtin_amdgcn_ds_bpermute((int)qa, (int)win), db = (uint32_t)__builtin_amdgcn_ds_bpermute((int)qa + 4, (int)win),
                               dc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)qa + 8, (int)win);
                const uint32_t xl0 = __builtin_amdgcn_alignbit(db, da, r), xh0 = __builtin_amdgcn_alignbit(dc, db, r);
                const uint64_t X = (uint64_t)xl0 | ((uint64_t)xh0 << 32);
                const uint32_t xl = (uint32_t)X;
                const uint32_t e = inf2_r32(smem, (uint32_t)INF2_LUT_L + ((xl & ((1u << INF2_ROOT_L) - 1)) << 2));
                const uint32_t cl = e & 15, used1 = (e >> 10) & 31;
                const bool is_lit = cl != 0 && (e & 0x300u) == 0, is_len = cl != 0 && (e & 0x300u) == 0x100u;
                const uint32_t len = (e >> 16) + ((xl >> cl) & ((1u << ((e >> 4) & 15)) - 1));
                const uint32_t y = (uint32_t)(X >> (is_len ? used1 : 0u));
                const uint32_t d = inf2_r32(smem, (uint32_t)INF2_LUT_D + ((y & ((1u << INF2_ROOT_D) - 1)) << 2));
                const bool d_ok = (d & 15) != 0 && (d & 0x300u) == 0x100u;
                const uint32_t dist = (d >> 16) + ((y >> (d & 15)) & ((1u << ((d >> 4) & 15)) - 1));
                // one word per lane for the walk along the chain: bits 0-8 the match's length (or the literal's byte), 9-14 the
                // symbol's bits (0: not an everyday symbol, the chain ends before it), 15 "an everyday match" (out of the ring, not
                // longer than its distance, 64 bytes at most, not from before the text as it stands now -- and, 64 symbols of 64
                // bytes at most, with 4 096 bytes of room left no such match needs its own look at the room), 16-31 the distance
                // (0: a literal)
                const uint32_t tb = is_lit ? cl : (is_len && d_ok) ? used1 + ((d >> 10) & 31) : 0u;
                const bool easy = !is_lit & !((dist > (uint32_t)INF2_WINDOW) | (dist < len) | (len > 64u) | (dist > n_out)) & (cap - n_out >= 4096u);
                const uint32_t infov = tb == 0 ? 0u : ((is_lit ? (e >> 16) : (len | (dist << 16))) | (tb << 9) | (easy ? 0x8000u : 0u));
                uint32_t pos = 0;
                #pragma unroll 1
                while (pos < 64) {
                    const uint32_t info = (uint32_t)__builtin_amdgcn_readlane((int)infov, (int)pos);
                    const uint32_t t = (info >> 9) & 63;
                    if (t == 0) break;
                    pos += t;
                    const uint32_t mlen = info & 0x1FFu, mdist = info >> 16;
                    if (!(info & 0x8000u)) {                          // a literal, or a match that is not an everyday one
                        if (mdist == 0) {
                            INF2_COMPLETE();
                            if (n_out >= cap) rc = 14;
                            else {
                                inf2_w8(smem, inf2_ring_at(lane == 0, n_out), info);
                                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)info, ors, lane == 0 ? (int)n_out : -1, 0, 0);
                                n_out++;
                            }
                        } else {
                            INF2_MATCH(mlen, mdist);
                        }
                        if (rc) break;
                        continue;
                    }
                    const bool on_ = lane < pend_n;                  // the ring write and the store of the match before
                    inf2_w8(smem, inf2_ring_at(on_, pend_pos + lane), pend_v);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pend_v, ors, on_ ? (int)(pend_pos + lane) : -1, 0, 0);
                    pend_pos = n_out; pend_n = mlen;
                    pend_v = inf2_r8(smem, (uint32_t)INF2_RING + ((n_out - mdist + lane) & (INF2_WINDOW - 1)));
                    n_out += mlen;
                }
                if (rc) break;
                if (pos == 0) {                                       // the symbol at the position itself is not an everyday one
                    const uint64_t sb = (uint64_t)inf2_u(xl0) | ((uint64_t)inf2_u(xh0) << 32);      // lane 0's window
                    uint32_t e0 = inf2_u(e);
                    if ((e0 & 15) == 0) { e0 = inf2_slow<0>(smem, sb, INF2_CNT_L, INF2_SYM_L); if (!e0) { rc = 13; break; } }
                    if ((e0 & 0x300u) == 0) {                         // a literal with a long code
                        pos = e0 & 15;
                        INF2_COMPLETE();
                        if (n_out >= cap) { rc = 14; break; }
                        inf2_w8(smem, inf2_ring_at(lane == 0, n_out), e0 >> 16);
                        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(e0 >> 16), ors, lane == 0 ? (int)n_out : -1, 0, 0);
                        n_out++;
                    } else if ((e0 & 0x300u) != 0x100u) {             // end of block, or not a symbol
                        pos = e0 & 15;
                        if ((e0 & 0x300u) == 0x300u) rc = 15;
                        done = 1;
                    } else {
                        const uint32_t used = (e0 >> 10) & 31;
                        const uint32_t slen = (e0 >> 16) + ((uint32_t)(sb >> (e0 & 15)) & ((1u << ((e0 >> 4) & 15)) - 1));
                        const uint64_t sb2 = sb >> used;
                        uint32_t d0 = inf2_u(inf2_r32(smem, (uint32_t)INF2_LUT_D + 4 * ((uint32_t)sb2 & ((1u << INF2_ROOT_D) - 1))));
                        if ((d0 & 0x300u) != 0x100u) {
                            if ((d0 & 15) == 0) d0 = inf2_slow<1>(smem, sb2, INF2_CNT_D, INF2_SYM_D);
                            if ((d0 & 0x300u) != 0x100u) { rc = 16; break; }
                        }
                        const uint32_t sdist = (d0 >> 16) + (((uint32_t)sb2 >> (d0 & 15)) & ((1u << ((d0 >> 4) & 15)) - 1));
                        pos = used + ((d0 >> 10) & 31);
                        INF2_MATCH(slen, sdist);
                        if (rc) break;
                    }
                }
                P += pos;
            }
            #undef INF2_WLOAD
            if (rc) break;
            // the bit buffer takes over where the symbols ended (the next block's header, or the end of the stream)
            const uint32_t bit = (W0 << 5) + P - (uint32_t)B.skip, byte = bit >> 3;
            if (byte > clen_bytes) { rc = 2; break; }
            start += byte; clen_bytes -= byte;
            B.open(comp, start, clen_bytes);
            B.take((int)(bit & 7));
            continue;
        }
        for (;;) {                                                   // the block's symbols
            B.refill();
            const uint32_t lo = (uint32_t)B.buf;
            // lane 0: the literal / length code at the low bits; lane i: the distance code that starts i bits on
            const uint32_t ent = inf2_r32(smem, look_base + (((lo >> look_shift) & look_mask) << 2));
            uint32_t e = inf2_u(ent);
            uint32_t len, dist;
            if ((e & 0x300u) == 0x100u) {                             // a length code out of the table: the common case
                const uint32_t used = (e >> 10) & 31;                 // the code and its extra bits; the distance code follows
                // (the scalar unit is what bounds this loop: the fields of the two entries are taken apart in the vector unit --
                // `vz` is 0 in every lane but not known to be uniform, so what is computed from it stays there)
                const uint32_t ev = e | vz;
                len = inf2_u((ev >> 16) + (((lo | vz) >> (ev & 15)) & ((1u << ((ev >> 4) & 15)) - 1)));
                uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)ent, (int)used);
                B.buf >>= used; B.cnt -= (int)used;
                B.refill();
                if ((d & 0x300u) != 0x100u) {                         // not in the distance table (a long code), or no distance
                    if ((d & 15) == 0) d = inf2_slow<1>(smem, B.buf, INF2_CNT_D, INF2_SYM_D);
                    if ((d & 0x300u) != 0x100u) { rc = 16; break; }
                }
                const uint32_t dv = d | vz;
                dist = inf2_u((dv >> 16) + ((((uint32_t)B.buf | vz) >> (dv & 15)) & ((1u << ((dv >> 4) & 15)) - 1)));
                const uint32_t du = (d >> 10) & 31;
                B.buf >>= du; B.cnt -= (int)du;
            } else {
                if ((e & 15) == 0) { e = inf2_slow<0>(smem, B.buf, INF2_CNT_L, INF2_SYM_L); if (!e) { rc = 13; break; } }
                if ((e & 0x300u) == 0) {                              // literal
                    const uint32_t l = e & 15;
                    B.buf >>= l; B.cnt -= (int)l;
                    INF2_COMPLETE();
                    if (n_out >= cap) { rc = 14; break; }
                    inf2_w8(smem, inf2_ring_at(lane == 0, n_out), e >> 16);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(e >> 16), ors, lane == 0 ? (int)n_out : -1, 0, 0);
                    n_out++;
                    continue;
                }
                if ((e & 0x300u) != 0x100u) {                         // end of block, or not a symbol
                    const uint32_t l = e & 15;
                    B.buf >>= l; B.cnt -= (int)l;
                    if ((e & 0x300u) == 0x300u) rc = 15;
                    break;
                }
                // a length code longer than the table's index: the distance code is looked up on its own
                const uint32_t used = (e >> 10) & 31;
                len = (e >> 16) + ((uint32_t)(B.buf >> (e & 15)) & ((1u << ((e >> 4) & 15)) - 1));
                B.buf >>= used; B.cnt -= (int)used;
                B.refill();
                uint32_t d = inf2_u(inf2_r32(smem, (uint32_t)INF2_LUT_D + 4 * ((uint32_t)B.buf & ((1u << INF2_ROOT_D) - 1))));
                if ((d & 0x300u) != 0x100u) {
                    if ((d & 15) == 0) d = inf2_slow<1>(smem, B.buf, INF2_CNT_D, INF2_SYM_D);
                    if ((d & 0x300u) != 0x100u) { rc = 16; break; }
                }
                dist = (d >> 16) + (((uint32_t)B.buf >> (d & 15)) & ((1u << ((d >> 4) & 15)) - 1));
                const uint32_t du = (d >> 10) & 31;
                B.buf >>= du; B.cnt -= (int)du;
            }
            {
                if (dist > n_out || len > cap - n_out) { rc = 17; break; }      // further back than the text goes, or the text would overflow
                {                                                     // the ring write and the store of the match before
                    const bool on_ = lane < pend_n;
                    inf2_w8(smem, inf2_ring_at(on_, pend_pos + lane), pend_v);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pend_v, ors, on_ ? (int)(pend_pos + lane) : -1, 0, 0);
                    pend_n = 0;
                }
                // the everyday match: from the ring, not longer than the distance, 64 bytes at most -- one test (in the vector unit)
                const uint32_t dvz = dist | vz, lvz = len | vz;
                if (!inf2_u((uint32_t)(dvz > (uint32_t)INF2_WINDOW) | (uint32_t)(dvz < lvz) | (uint32_t)(lvz > 64u))) {
                    pend_pos = n_out; pend_n = len;
                    pend_v = inf2_r8(smem, (uint32_t)INF2_RING + ((n_out - dist + lane) & (INF2_WINDOW - 1)));
                    n_out += len;
                    continue;
                }
                if (dist > INF2_WINDOW) {                             // from further back than the ring holds: out of global memory
                    #pragma unroll 1
                    for (uint32_t done = 0; done < len; done += 64) { // (dist > len: no byte of the match is its own source)
                        const uint32_t k = done + lane;
                        const bool on = k < len;
                        const uint8_t v = __builtin_amdgcn_raw_buffer_load_b8(ors, on ? (int)(n_out - dist + k) : -1, 0, 0);
                        inf2_w8(smem, inf2_ring_at(on, n_out + k), v);
                        __builtin_amdgcn_raw_buffer_store_b8(v, ors, on ? (int)(n_out + k) : -1, 0, 0);
                    }
                    n_out += len;
                    continue;
                }
                const uint32_t from = n_out - dist;
                const bool wrap = dist < len;
                const float rcp = __builtin_amdgcn_rcpf((float)dist);
                uint32_t done = 0;
                #pragma unroll 1
                for (; done + 64 < len; done += 64) {                 // all but the last 64 bytes: written at once
                    const uint32_t k = done + lane;
                    const uint32_t r = wrap ? inf2_mod(k, dist, rcp) : k;
                    const uint32_t v = inf2_r8(smem, (uint32_t)INF2_RING + ((from + r) & (INF2_WINDOW - 1)));
                    inf2_w8(smem, (uint32_t)INF2_RING + ((n_out + k) & (INF2_WINDOW - 1)), v);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, ors, (int)(n_out + k), 0, 0);
                }
                {
                    const uint32_t k = done + lane;
                    const uint32_t r = wrap ? inf2_mod(k, dist, rcp) : k;
                    pend_pos = n_out + done; pend_n = len - done;
                    pend_v = inf2_r8(smem, (uint32_t)INF2_RING + ((from + r) & (INF2_WINDOW - 1)));
                }
                n_out += len;
                continue;
            }
        }
    }
    INF2_COMPLETE();
    if (!rc && B.consumed_bits() > B.len_bits) rc = 2;              // the codes ran past the block's last byte
    if (!rc && n_out != cap) rc = 18;
    status[b] = rc;
}

// a wave per block; with fewer workgroups than blocks (a grid of so many per compute unit) a wave goes on to further blocks
template <bool MULTI>
static __global__ void __launch_bounds__(64) k_inflate_wave(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                     const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                     const uint32_t *__restrict__ out_len, int n_blocks,
                                                     uint8_t *text, int32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[INF2_LDS];
    #pragma unroll 1
    for (int b = (int)blockIdx.x; b < n_blocks; b += (int)gridDim.x) {
        inf2_one_block<MULTI>(smem, b, comp, in_off, in_len, out_off, out_len, text, status);
        __syncthreads();
    }
}

}  // namespace hpgv
