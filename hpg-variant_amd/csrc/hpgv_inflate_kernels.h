// hpgv_inflate_kernels.h -- raw DEFLATE (RFC 1951) on the GPU for BGZF-compressed VCF text
// (`--compression bgzip`, shared_options.c:60-61).
//
// A bgzip file is a sequence of independent DEFLATE streams of at most 64 KiB of text each, so a file of G gigabytes is
// 16 000 G independent decoding jobs: one lane per block, a whole file's blocks in one launch (a block is a serial job
// of a few milliseconds per lane; only tens of thousands of them at once fill the chip).  The compressed bytes cross
// PCIe instead of the text (a seventh of it for genotype text).
//
// Per lane: a 64-bit bit buffer and canonical-code decoding from the count / symbol arrays of the code (no large
// tables: a lane's state is 1.4 KB of private memory).  Every irregularity ends the lane with a non-zero status; the
// host decodes such blocks itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

struct InflateCode { uint16_t count[16]; uint16_t symbol[288]; };
// the same with the symbols in LDS, one column per lane (entry k of lane l at [k * 64 + l]): a look-up in private memory
// fetches a cache line for two bytes, and 250 000 lanes' tables do not stay in any cache
template <typename T> struct InflateCodeLds {
    uint16_t count[16]; T *symbol;                       // symbol: this lane's column
    __device__ __forceinline__ T get(int k) const { return symbol[k * 64]; }
    __device__ __forceinline__ void put(int k, T v) { symbol[k * 64] = v; }
};

struct InflateBits {
    const uint8_t *in, *end;
    uint64_t buf; int cnt;
    __device__ __forceinline__ void refill() {
        if (in + 8 <= end) {                                        // one 8-byte load instead of up to seven dependent byte loads
            uint64_t w;
            __builtin_memcpy(&w, in, 8);
            buf |= w << cnt;
            in += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && in < end) { buf |= (uint64_t)(*in++) << cnt; cnt += 8; }
        }
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1)); buf >>= n; cnt -= n; return v; }
};

// the code-length histogram of a code held in registers (the decode loop below is unrolled over the lengths, so every
// index is static): without it each of the up to 15 steps per symbol waits for a load from private memory
struct InflateCnt { uint32_t v[16]; };
__device__ __forceinline__ InflateCnt inflate_counts(const InflateCode &h) {
    InflateCnt c;
    #pragma unroll
    for (int l = 0; l < 16; l++) c.v[l] = h.count[l];
    return c;
}

// canonical decode, one bit at a time (codes are packed MSB first): returns the symbol or -1
__device__ __forceinline__ int inflate_decode(InflateBits &B, const InflateCnt &c, const uint16_t *symbol) {
    int code = 0, first = 0, index = 0;
    if (B.cnt < 15) B.refill();
    uint64_t bits = B.buf;
    #pragma unroll
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1); bits >>= 1;
        const int count = (int)c.v[len];
        if (code - count < first) {
            if (len > B.cnt) return -1;
            B.buf >>= len; B.cnt -= len;
            return symbol[index + (code - first)];
        }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

template <typename T>
__device__ __forceinline__ int inflate_decode_lds(InflateBits &B, const InflateCnt &c, const InflateCodeLds<T> &h) {
    int code = 0, first = 0, index = 0;
    if (B.cnt < 15) B.refill();
    uint64_t bits = B.buf;
    #pragma unroll
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1); bits >>= 1;
        const int count = (int)c.v[len];
        if (code - count < first) {
            if (len > B.cnt) return -1;
            B.buf >>= len; B.cnt -= len;
            return (int)h.get(index + (code - first));
        }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}
template <typename T>
__device__ __forceinline__ InflateCnt inflate_counts_lds(const InflateCodeLds<T> &h) {
    InflateCnt c;
    #pragma unroll
    for (int l = 0; l < 16; l++) c.v[l] = h.count[l];
    return c;
}
template <typename T>
__device__ __forceinline__ int inflate_construct_lds(InflateCodeLds<T> &h, const uint8_t *length, int n) {
    for (int l = 0; l <= 15; l++) h.count[l] = 0;
    for (int s = 0; s < n; s++) h.count[length[s]]++;
    if (h.count[0] == n) return 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) return left; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; s++) if (length[s]) h.put(offs[length[s]]++, (T)s);
    return left;
}

// code lengths -> count / symbol arrays; returns 0 for a complete code, >0 incomplete, <0 over-subscribed
__device__ __forceinline__ int inflate_construct(InflateCode &h, const uint8_t *length, int n) {
    for (int l = 0; l <= 15; l++) h.count[l] = 0;
    for (int s = 0; s < n; s++) h.count[length[s]]++;
    if (h.count[0] == n) return 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) return left; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; s++) if (length[s]) h.symbol[offs[length[s]]++] = (uint16_t)s;
    return left;
}

__device__ __forceinline__ void inflate_one_block(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                  const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                  const uint32_t *__restrict__ out_len, int b,
                                                  uint8_t *__restrict__ text, int32_t *__restrict__ status) {
    const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    const uint8_t len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    const uint8_t dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    InflateBits B;
    B.in = comp + in_off[b]; B.end = B.in + in_len[b]; B.buf = 0; B.cnt = 0;
    uint8_t *const out0 = text + out_off[b];
    const uint32_t cap = out_len[b];
    uint32_t n_out = 0;
    int rc = 0, last = 0;
    InflateCode lencode, distcode;
    uint8_t lengths[320];
    while (!last && !rc) {
        B.refill();
        if (B.cnt < 3) { rc = 2; break; }
        last = (int)B.take(1);
        const int type = (int)B.take(2);
        if (type == 0) {                                            // stored
            B.take(B.cnt & 7);
            B.refill();
            if (B.cnt < 32) { rc = 2; break; }
            const uint32_t len = B.take(16), nlen = B.take(16);
            if ((len ^ nlen) != 0xFFFFu) { rc = 3; break; }
            for (uint32_t k = 0; k < len; k++) {
                if (B.cnt < 8) B.refill();
                if (B.cnt < 8 || n_out >= cap) { rc = 2; break; }
                out0[n_out++] = (uint8_t)B.take(8);
            }
            continue;
        }
        if (type == 3) { rc = 4; break; }
        if (type == 1) {
            for (int s = 0; s < 144; s++) lengths[s] = 8;
            for (int s = 144; s < 256; s++) lengths[s] = 9;
            for (int s = 256; s < 280; s++) lengths[s] = 7;
            for (int s = 280; s < 288; s++) lengths[s] = 8;
            inflate_construct(lencode, lengths, 288);
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            inflate_construct(distcode, lengths, 30);
        } else {
            B.refill();
            if (B.cnt < 14) { rc = 2; break; }
            const int nlen = (int)B.take(5) + 257, ndist = (int)B.take(5) + 1, ncode = (int)B.take(4) + 4;
            if (nlen > 286 || ndist > 30) { rc = 5; break; }
            for (int k = 0; k < 19; k++) lengths[k] = 0;
            for (int k = 0; k < ncode; k++) { if (B.cnt < 3) B.refill(); if (B.cnt < 3) { rc = 2; break; } lengths[order[k]] = (uint8_t)B.take(3); }
            if (rc) break;
            if (inflate_construct(lencode, lengths, 19) != 0) { rc = 6; break; }
            const InflateCnt clc = inflate_counts(lencode);
            int idx = 0;
            while (idx < nlen + ndist && !rc) {
                const int sym = inflate_decode(B, clc, lencode.symbol);
                if (sym < 0) { rc = 7; break; }
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (B.cnt < 7) B.refill();
                if (sym == 16) { if (idx == 0) { rc = 8; break; } val = lengths[idx - 1]; rep = 3 + (int)B.take(2); }
                else if (sym == 17) rep = 3 + (int)B.take(3);
                else rep = 11 + (int)B.take(7);
                if (B.cnt < 0 || idx + rep > nlen + ndist) { rc = 9; break; }
                while (rep--) lengths[idx++] = (uint8_t)val;
            }
            if (rc) break;
            if (lengths[256] == 0) { rc = 10; break; }
            int e = inflate_construct(lencode, lengths, nlen);
            if (e < 0 || (e > 0 && nlen - lencode.count[0] != 1)) { rc = 11; break; }
            e = inflate_construct(distcode, lengths + nlen, ndist);
            if (e < 0 || (e > 0 && ndist - distcode.count[0] != 1)) { rc = 12; break; }
        }
        const InflateCnt lc = inflate_counts(lencode), dc = inflate_counts(distcode);
        for (;;) {                                                  // the block's symbols
            int sym = inflate_decode(B, lc, lencode.symbol);
            if (sym < 0) { rc = 13; break; }
            if (sym < 256) { if (n_out >= cap) { rc = 14; break; } out0[n_out++] = (uint8_t)sym; continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { rc = 15; break; }
            if (B.cnt < 5) B.refill();
            const uint32_t len = len_base[sym] + B.take(len_extra[sym]);
            const int ds = inflate_decode(B, dc, distcode.symbol);
            if (ds < 0 || ds >= 30) { rc = 16; break; }
            if (B.cnt < 13) B.refill();
            const uint32_t dist = dist_base[ds] + B.take(dist_extra[ds]);
            if (B.cnt < 0 || dist > n_out || n_out + len > cap) { rc = 17; break; }
            uint8_t *d = out0 + n_out;
            const uint8_t *src = d - dist;
            // a byte loop makes every byte wait for a load of bytes just stored (a round trip to L2 each); eight bytes at a
            // time, and for a period below eight from a pattern held in a register
            if (n_out + len + 8 <= cap) {
                uint32_t k = 0;
                if (dist >= 8) {
                    for (; k < len; k += 8) { uint64_t w; __builtin_memcpy(&w, src + k, 8); __builtin_memcpy(d + k, &w, 8); }
                } else {
                    uint64_t w = 0;                                  // dist bytes repeated up to eight
                    for (uint32_t q = 0; q < 8; q++) w |= (uint64_t)src[q % dist] << (8 * q);
                    const uint32_t step = dist * (8 / dist);         // whole periods per store
                    for (; k < len; k += step) __builtin_memcpy(d + k, &w, 8);
                }
            } else {
                for (uint32_t k = 0; k < len; k++) d[k] = src[k];
            }
            n_out += len;
        }
    }
    if (!rc && n_out != cap) rc = 18;
    status[b] = rc;
}

static __global__ void __launch_bounds__(64) k_inflate_blocks(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                       const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                       const uint32_t *__restrict__ out_len, int n_blocks,
                                                       uint8_t *__restrict__ text, int32_t *__restrict__ status) {
    // one workgroup per 64 blocks, or (a grid shorter than that) workgroups that go on to further blocks: fewer waves on
    // the compute units, which leaves registers for the kernels of a pipeline beside the decoder
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x)
        inflate_one_block(comp, in_off, in_len, out_off, out_len, b, text, status);
}

// the same decoder with the symbol tables in LDS (38 KB per workgroup: four workgroups per compute unit) -- measured, not the
// default: 153 against 122 GB/s at 32 768 blocks, 230 against 217 at 125 000, 232 against 240 at 500 000 (the decoder is bound
// by its scattered text writes and match copies, not by the look-ups), and its LDS keeps other kernels off the compute units
static __global__ void __launch_bounds__(64) k_inflate_blocks_lds(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ in_off,
                                                       const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                       const uint32_t *__restrict__ out_len, int n_blocks,
                                                       uint8_t *__restrict__ text, int32_t *__restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;                           // (no barrier in this kernel: every lane is on its own)
    const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    const uint8_t len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    const uint8_t dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    InflateBits B;
    B.in = comp + in_off[b]; B.end = B.in + in_len[b]; B.buf = 0; B.cnt = 0;
    uint8_t *const out0 = text + out_off[b];
    const uint32_t cap = out_len[b];
    uint32_t n_out = 0;
    int rc = 0, last = 0;
    __shared__ uint16_t s_lsym[288 * 64];
    __shared__ uint8_t s_dsym[32 * 64];
    InflateCodeLds<uint16_t> lencode; InflateCodeLds<uint8_t> distcode;
    lencode.symbol = s_lsym + threadIdx.x; distcode.symbol = s_dsym + threadIdx.x;
    uint8_t lengths[320];
    while (!last && !rc) {
        B.refill();
        if (B.cnt < 3) { rc = 2; break; }
        last = (int)B.take(1);
        const int type = (int)B.take(2);
        if (type == 0) {                                            // stored
            B.take(B.cnt & 7);
            B.refill();
            if (B.cnt < 32) { rc = 2; break; }
            const uint32_t len = B.take(16), nlen = B.take(16);
            if ((len ^ nlen) != 0xFFFFu) { rc = 3; break; }
            for (uint32_t k = 0; k < len; k++) {
                if (B.cnt < 8) B.refill();
                if (B.cnt < 8 || n_out >= cap) { rc = 2; break; }
                out0[n_out++] = (uint8_t)B.take(8);
            }
            continue;
        }
        if (type == 3) { rc = 4; break; }
        if (type == 1) {
            for (int s = 0; s < 144; s++) lengths[s] = 8;
            for (int s = 144; s < 256; s++) lengths[s] = 9;
            for (int s = 256; s < 280; s++) lengths[s] = 7;
            for (int s = 280; s < 288; s++) lengths[s] = 8;
            inflate_construct_lds(lencode, lengths, 288);
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            inflate_construct_lds(distcode, lengths, 30);
        } else {
            B.refill();
            if (B.cnt < 14) { rc = 2; break; }
            const int nlen = (int)B.take(5) + 257, ndist = (int)B.take(5) + 1, ncode = (int)B.take(4) + 4;
            if (nlen > 286 || ndist > 30) { rc = 5; break; }
            for (int k = 0; k < 19; k++) lengths[k] = 0;
            for (int k = 0; k < ncode; k++) { if (B.cnt < 3) B.refill(); if (B.cnt < 3) { rc = 2; break; } lengths[order[k]] = (uint8_t)B.take(3); }
            if (rc) break;
            if (inflate_construct_lds(lencode, lengths, 19) != 0) { rc = 6; break; }
            const InflateCnt clc = inflate_counts_lds(lencode);
            int idx = 0;
            while (idx < nlen + ndist && !rc) {
                const int sym = inflate_decode_lds(B, clc, lencode);
                if (sym < 0) { rc = 7; break; }
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (B.cnt < 7) B.refill();
                if (sym == 16) { if (idx == 0) { rc = 8; break; } val = lengths[idx - 1]; rep = 3 + (int)B.take(2); }
                else if (sym == 17) rep = 3 + (int)B.take(3);
                else rep = 11 + (int)B.take(7);
                if (B.cnt < 0 || idx + rep > nlen + ndist) { rc = 9; break; }
                while (rep--) lengths[idx++] = (uint8_t)val;
            }
            if (rc) break;
            if (lengths[256] == 0) { rc = 10; break; }
            int e = inflate_construct_lds(lencode, lengths, nlen);
            if (e < 0 || (e > 0 && nlen - lencode.count[0] != 1)) { rc = 11; break; }
            e = inflate_construct_lds(distcode, lengths + nlen, ndist);
            if (e < 0 || (e > 0 && ndist - distcode.count[0] != 1)) { rc = 12; break; }
        }
        const InflateCnt lc = inflate_counts_lds(lencode), dc = inflate_counts_lds(distcode);
        for (;;) {                                                  // the block's symbols
            int sym = inflate_decode_lds(B, lc, lencode);
            if (sym < 0) { rc = 13; break; }
            if (sym < 256) { if (n_out >= cap) { rc = 14; break; } out0[n_out++] = (uint8_t)sym; continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { rc = 15; break; }
            if (B.cnt < 5) B.refill();
            const uint32_t len = len_base[sym] + B.take(len_extra[sym]);
            const int ds = inflate_decode_lds(B, dc, distcode);
            if (ds < 0 || ds >= 30) { rc = 16; break; }
            if (B.cnt < 13) B.refill();
            const uint32_t dist = dist_base[ds] + B.take(dist_extra[ds]);
            if (B.cnt < 0 || dist > n_out || n_out + len > cap) { rc = 17; break; }
            uint8_t *d = out0 + n_out;
            const uint8_t *src = d - dist;
            // a byte loop makes every byte wait for a load of bytes just stored (a round trip to L2 each); eight bytes at a
            // time, and for a period below eight from a pattern held in a register
            if (n_out + len + 8 <= cap) {
                uint32_t k = 0;
                if (dist >= 8) {
                    for (; k < len; k += 8) { uint64_t w; __builtin_memcpy(&w, src + k, 8); __builtin_memcpy(d + k, &w, 8); }
                } else {
                    uint64_t w = 0;                                  // dist bytes repeated up to eight
                    for (uint32_t q = 0; q < 8; q++) w |= (uint64_t)src[q % dist] << (8 * q);
                    const uint32_t step = dist * (8 / dist);         // whole periods per store
                    for (; k < len; k += step) __builtin_memcpy(d + k, &w, 8);
                }
            } else {
                for (uint32_t k = 0; k < len; k++) d[k] = src[k];
            }
            n_out += len;
        }
    }
    if (!rc && n_out != cap) rc = 18;
    status[b] = rc;
}


}  // namespace hpgv
