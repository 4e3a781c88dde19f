// hpgv_bgzf_kernels.h -- the block table of a bgzip file, built on the device from the compressed bytes as they arrive
// (`--compression bgzip`, shared_options.c:60-61).
//
// A BGZF file is a chain: a block's header gives its size, which gives the next header.  Walking 490 000 headers of a
// 4.6 GB file on the host costs 0.08 - 0.26 s of dependent reads before the first block can be decoded.  On the device the
// chain is FOUND instead of followed: every byte position is tested for the 16 fixed bytes of the header bgzip writes
// (1f 8b 08 04 .. .. .. .. .. .. 06 00 'B' 'C' 02 00 + BSIZE), the hits are put in file order, and they are accepted as
// long as each one begins where its predecessor ends, starting from a position that is known to be a block start.  A hit
// inside compressed data (2^-80 per position) breaks that chain and is skipped by the next call, which starts at the end
// of the accepted part; a file whose headers look different (other extra fields) yields no chain and the host walks it.
//
//   k_bgzf_count   hits per 16 KiB tile                                      (reads the bytes once)
//   k_bgzf_scan    exclusive prefix of the tile counts                       (one workgroup)
//   k_bgzf_list    the hits again, ranked inside their tile, written densely (reads the bytes a second time)
//   k_bgzf_chain   chain check, text offsets, the decoder's table rows       (one workgroup)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

enum { BGZF_TILE = 16384, BGZF_TPB = 256, BGZF_TILE_MAX = 640 };      // a block is at least 26 bytes: at most 631 starts per tile

struct BgzfHit { uint64_t pos; uint32_t bsize, isize; };

// is there a bgzip-written block header at byte p whose block ends at or before hi?  (lo .. hi are uploaded bytes)
__device__ __forceinline__ bool bgzf_header_at(const uint8_t *comp, uint64_t p, uint64_t hi, uint32_t *bsize, uint32_t *isize) {
    if (p + 28 > hi) return false;
    const uint8_t *h = comp + p;
    if (h[4 + 6] != 6 || h[4 + 7] != 0 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) return false;
    const uint32_t bs = ((uint32_t)h[16] | ((uint32_t)h[17] << 8)) + 1;
    if (bs < 28 || p + bs > hi) return false;
    const uint8_t *t = comp + p + bs - 4;
    const uint32_t is = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    if (is > 65536u) return false;
    *bsize = bs; *isize = is;
    return true;
}

// the tile's hits: every thread tests its 64 byte positions for the magic dword, then the rest of the header
template <bool LIST>
__device__ __forceinline__ void bgzf_tile(const uint8_t *comp, uint64_t base, uint64_t lo, uint64_t hi, uint32_t *s_n, BgzfHit *s_hit) {
    const uint64_t t0 = base + (uint64_t)blockIdx.x * BGZF_TILE + (uint64_t)threadIdx.x * 64;      // 16-byte aligned
    if (t0 >= hi) return;
    uint32_t w[17];
    const uint4 *q = (const uint4 *)(comp + t0);
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 v = (t0 + 16 * (uint64_t)k + 16 <= ((hi + 15) & ~15ull)) ? q[k] : make_uint4(0, 0, 0, 0);
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    w[16] = (t0 + 64 + 4 <= ((hi + 15) & ~15ull)) ? *(const uint32_t *)(comp + t0 + 64) : 0u;
    #pragma unroll
    for (int k = 0; k < 16; k++) {
        #pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t m = s == 0 ? w[k] : __builtin_amdgcn_alignbyte(w[k + 1], w[k], s);
            if (m == 0x04088b1fu) {
                const uint64_t p = t0 + 4 * (uint64_t)k + (uint64_t)s;
                uint32_t bs, is;
                if (p >= lo && bgzf_header_at(comp, p, hi, &bs, &is)) {
                    const uint32_t slot = atomicAdd(s_n, 1u);
                    if (LIST && slot < BGZF_TILE_MAX) { s_hit[slot].pos = p; s_hit[slot].bsize = bs; s_hit[slot].isize = is; }
                }
            }
        }
    }
}

// bytes [base, hi) in tiles from the 16-byte aligned `base` (<= lo); only positions >= lo count
static __global__ void __launch_bounds__(BGZF_TPB) k_bgzf_count(const uint8_t *__restrict__ comp, uint64_t base, uint64_t lo, uint64_t hi,
                                                         uint32_t *__restrict__ tile_n) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bgzf_tile<false>(comp, base, lo, hi, &s_n, nullptr);
    __syncthreads();
    if (threadIdx.x == 0) tile_n[blockIdx.x] = s_n;
}

// tile_n[0 .. n_tiles) -> exclusive prefix in place; total[0] = the sum
static __global__ void __launch_bounds__(1024) k_bgzf_scan(uint32_t *__restrict__ tile_n, int n_tiles, uint32_t *__restrict__ total) {
    __shared__ uint32_t s_part[1024];
    const int per = (n_tiles + 1023) / 1024, a = (int)threadIdx.x * per, b = min(a + per, n_tiles);
    uint32_t sum = 0;
    for (int i = a; i < b; i++) sum += tile_n[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int i = a; i < b; i++) { const uint32_t c = tile_n[i]; tile_n[i] = run; run += c; }
    if (threadIdx.x == 1023) total[0] = s_part[1023];
}

// the hits in file order: hit[tile_base[tile] + rank inside the tile], as far as cap goes
static __global__ void __launch_bounds__(BGZF_TPB) k_bgzf_list(const uint8_t *__restrict__ comp, uint64_t base, uint64_t lo, uint64_t hi,
                                                        const uint32_t *__restrict__ tile_base, BgzfHit *__restrict__ hit, uint32_t cap) {
    __shared__ uint32_t s_n;
    __shared__ BgzfHit s_hit[BGZF_TILE_MAX];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bgzf_tile<true>(comp, base, lo, hi, &s_n, s_hit);
    __syncthreads();
    const uint32_t n = min(s_n, (uint32_t)BGZF_TILE_MAX), at = tile_base[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += BGZF_TPB) {
        const BgzfHit h = s_hit[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) rank += s_hit[j].pos < h.pos ? 1u : 0u;
        if (at + rank < cap) hit[at + rank] = h;
    }
}

// result: [0] rows written, [1] where the accepted chain ends (the next call's lo), [2] text bytes after these rows,
// [3] hits seen in the range
static __global__ void __launch_bounds__(1024) k_bgzf_chain(const BgzfHit *__restrict__ hit, const uint32_t *__restrict__ total, uint32_t cap,
                                                      uint64_t lo, uint64_t text_base, uint32_t max_rows,
                                                      uint64_t *__restrict__ in_off, uint32_t *__restrict__ in_len,
                                                      uint64_t *__restrict__ out_off, uint32_t *__restrict__ out_len,
                                                      uint64_t *__restrict__ result) {
    __shared__ uint32_t s_first_bad;
    __shared__ uint64_t s_part[1024];
    const uint32_t n_hit = min(min(total[0], cap), max_rows);
    if (threadIdx.x == 0) s_first_bad = n_hit;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_hit; i += 1024) {
        const bool ok = i == 0 ? hit[0].pos == lo : hit[i - 1].pos + hit[i - 1].bsize == hit[i].pos;
        if (!ok) atomicMin(&s_first_bad, i);
    }
    __syncthreads();
    const uint32_t n = s_first_bad;
    const uint32_t per = (n + 1023) / 1024, a = min(threadIdx.x * per, n), b = min(a + per, n);
    uint64_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += hit[i].isize;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t v = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0ull;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = text_base + s_part[threadIdx.x] - sum;
    for (uint32_t i = a; i < b; i++) {
        const BgzfHit h = hit[i];
        in_off[i] = h.pos + 18; in_len[i] = h.bsize - 26; out_off[i] = run; out_len[i] = h.isize;
        run += h.isize;
    }
    if (threadIdx.x == 1023) {
        result[0] = n;
        result[1] = n ? hit[n - 1].pos + hit[n - 1].bsize : lo;
        result[2] = text_base + s_part[1023];
        result[3] = total[0];
    }
}

}  // namespace hpgv
