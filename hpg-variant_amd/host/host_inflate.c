/* host_inflate.c -- raw DEFLATE (RFC 1951) on the host: the decoder of BGZF blocks the device path hands back, and of the host path.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ---- raw DEFLATE (RFC 1951) decoder for BGZF blocks ------------------------------------------------------------
 * zlib's inflate does 0.7 GB/s per thread on VCF text and bounded the bgzip runs; this one keeps a 64-bit bit
 * buffer, decodes through two-level tables (10-bit first level for literals / lengths, 8-bit for distances) and copies
 * matches eight bytes at a time.  Input and output sizes are known up front (a BGZF block states both).  It returns
 * non-zero on ANY irregularity -- the caller then decodes the block again with zlib, so this is an accelerator, never
 * the arbiter of what a valid block is. */
typedef struct { uint32_t e[1024 + 592 + 8]; } fi_lit_t;        /* entry: bits 0-7 code length, 8-15 kind / extra bits, 16-31 value */
typedef struct { uint32_t e[256 + 400 + 8]; } fi_dist_t;
enum { FI_LIT = 0x00, FI_LEN = 0x40, FI_EOB = 0x80, FI_SUB = 0xC0, FI_BAD = 0xFF };

/* canonical Huffman code lengths -> decode table with `root` first-level bits.  Entries: len in bits 0-7 (for a
 * sub-table link: the first level's bits), kind | extra bits in 8-15, value in 16-31.  Returns 0 when the code is
 * complete (or the single-code case deflate allows for distances), non-zero otherwise. */
static int fi_build(uint32_t *tab, int tab_cap, int root, const uint8_t *lens, int n, int is_dist) {
    static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint8_t len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint8_t dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    int count[16] = {0}, maxlen = 0, used = 0;
    for (int i = 0; i < n; i++) { count[lens[i]]++; if (lens[i] > maxlen) maxlen = lens[i]; if (lens[i]) used++; }
    const int first = 1 << root;
    for (int i = 0; i < first; i++) tab[i] = FI_BAD << 8;
    if (used == 0) return is_dist ? 0 : 1;                          /* no distance codes at all: literals only */
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return 1; }
    if (left > 0 && !(used == 1 && maxlen == 1)) return 1;           /* incomplete (the one-code case is allowed) */
    int next_code[16], code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next_code[l] = code; }
    int sub_next = first;                                           /* where the next sub-table goes */
    /* sub-tables: one per distinct first-level prefix among the long codes, sized for the longest code below it.
     * Symbols are visited in (length, symbol) order, i.e. in canonical code order: prefixes arrive grouped. */
    int sub_prefix = -1, sub_bits = 0, sub_base = 0;
    int remaining[16];
    for (int l = 0; l <= 15; l++) remaining[l] = count[l];
    for (int l = 1; l <= 15; l++) {
        for (int sym = 0; sym < n; sym++) {
            if (lens[sym] != l) continue;
            const int c = next_code[l]++;
            uint32_t rev = 0;                                       /* deflate packs codes starting from the MSB: reverse */
            for (int b = 0; b < l; b++) rev |= (uint32_t)((c >> b) & 1) << (l - 1 - b);
            uint32_t kind, value;
            if (is_dist) { if (sym >= 30) { kind = FI_BAD; value = 0; } else { kind = FI_LEN | dist_extra[sym]; value = dist_base[sym]; } }
            else if (sym < 256) { kind = FI_LIT; value = (uint32_t)sym; }
            else if (sym == 256) { kind = FI_EOB; value = 0; }
            else if (sym > 285) { kind = FI_BAD; value = 0; }         /* 286, 287: in the fixed code, never valid in data */
            else { kind = FI_LEN | len_extra[sym - 257]; value = len_base[sym - 257]; }
            if (l <= root) {
                const uint32_t ent = (value << 16) | (kind << 8) | (uint32_t)l;
                for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) tab[i] = ent;
            } else {
                const int prefix = (int)(rev & (uint32_t)(first - 1));
                if (prefix != sub_prefix) {
                    /* bits below this prefix: as many as the codes still to come under it need (zlib's inflate_table rule;
                     * remaining[] counts the codes of each length not placed yet, this one included) */
                    int bits = l - root, room = 1 << bits;
                    while (bits + root < maxlen) {
                        room -= remaining[bits + root];
                        if (room <= 0) break;
                        bits++; room <<= 1;
                    }
                    if (sub_next + (1 << bits) > tab_cap) return 1;
                    sub_prefix = prefix; sub_bits = bits; sub_base = sub_next; sub_next += 1 << bits;
                    for (int i = 0; i < (1 << bits); i++) tab[sub_base + i] = FI_BAD << 8;
                    tab[prefix] = ((uint32_t)sub_base << 16) | ((uint32_t)(FI_SUB | sub_bits) << 8) | (uint32_t)root;
                }
                const uint32_t ent = (value << 16) | (kind << 8) | (uint32_t)(l - root);
                for (uint32_t i = rev >> root; i < (1u << sub_bits); i += 1u << (l - root)) tab[sub_base + i] = ent;
            }
            remaining[l]--;
        }
    }
    return 0;
}

#define FI_REFILL() do {                                                                             \
        if (in + 8 <= in_end) { uint64_t w; memcpy(&w, in, 8); bitbuf |= w << bitcnt; in += (63 - bitcnt) >> 3; bitcnt |= 56; } \
        else while (bitcnt <= 56 && in < in_end) { bitbuf |= (uint64_t)*in++ << bitcnt; bitcnt += 8; } \
    } while (0)
#define FI_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))

int fast_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    const unsigned char *in_end = in + in_len;
    unsigned char *const out0 = out, *const out_end = out + out_len;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    fi_lit_t lit_tab; fi_dist_t dist_tab;                          /* 9 KB of stack */
    fi_lit_t *const LT = &lit_tab;
    fi_dist_t *const DT = &dist_tab;
    int rc = 1, last = 0;
    while (!last) {
        FI_REFILL();
        if (bitcnt < 3) goto done;
        last = (int)(bitbuf & 1); const int type = (int)((bitbuf >> 1) & 3); FI_TAKE(3);
        if (type == 0) {                                            /* stored */
            FI_TAKE(bitcnt & 7);
            FI_REFILL();
            if (bitcnt < 32) goto done;
            const unsigned len = (unsigned)(bitbuf & 0xFFFF), nlen = (unsigned)((bitbuf >> 16) & 0xFFFF);
            FI_TAKE(32);
            if ((len ^ nlen) != 0xFFFF) goto done;
            in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0;              /* whole bytes still in the buffer go back */
            if ((size_t)(in_end - in) < len || (size_t)(out_end - out) < len) goto done;
            memcpy(out, in, len); out += len; in += len;
            continue;
        }
        if (type == 3) goto done;
        uint8_t lens[320];
        int nlit, ndist;
        if (type == 1) {
            nlit = 288; ndist = 32;                                 /* the fixed distance code has 32 five-bit codes (30, 31 never valid) */
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            for (int i = 0; i < 32; i++) lens[288 + i] = 5;
        } else {
            FI_REFILL();
            if (bitcnt < 14) goto done;
            nlit = (int)(bitbuf & 31) + 257; ndist = (int)((bitbuf >> 5) & 31) + 1; const int ncode = (int)((bitbuf >> 10) & 15) + 4;
            FI_TAKE(14);
            if (nlit > 286 || ndist > 30) goto done;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) { FI_REFILL(); if (bitcnt < 3) goto done; cl[order[i]] = (uint8_t)(bitbuf & 7); FI_TAKE(3); }
            uint32_t ct[128 + 8];
            if (fi_build(ct, 128, 7, cl, 19, 0) ) {                  /* code-length code: at most 7 bits, one level; kinds unused */
                /* fi_build classifies symbols as literals here (all < 256): a code with a single symbol is rejected */
                goto done;
            }
            int i = 0;
            while (i < nlit + ndist) {
                FI_REFILL();
                const uint32_t ent = ct[bitbuf & 127];
                const int l = (int)(ent & 0xFF);
                if (((ent >> 8) & 0xFF) == FI_BAD || l > bitcnt) goto done;
                FI_TAKE(l);
                const int sym = (int)(ent >> 16);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0 || bitcnt < 2) goto done; val = lens[i - 1]; rep = 3 + (int)(bitbuf & 3); FI_TAKE(2); }
                else if (sym == 17) { if (bitcnt < 3) goto done; rep = 3 + (int)(bitbuf & 7); FI_TAKE(3); }
                else { if (bitcnt < 7) goto done; rep = 11 + (int)(bitbuf & 127); FI_TAKE(7); }
                if (i + rep > nlit + ndist) goto done;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) goto done;                           /* no end-of-block code */
            memmove(lens + 288, lens + nlit, (size_t)ndist);        /* distances at a fixed place */
            for (int k = nlit; k < 288; k++) lens[k] = 0;
        }
        if (fi_build(LT->e, 1024 + 592, 10, lens, type == 1 ? 288 : nlit, 0)) goto done;
        if (fi_build(DT->e, 256 + 400, 8, lens + 288, ndist, 1)) goto done;
        for (;;) {                                                  /* symbols of this block */
            FI_REFILL();
            uint32_t ent = LT->e[bitbuf & 1023];
            if ((ent >> 8 & 0xC0) == FI_SUB && (ent >> 8 & 0xFF) != FI_BAD) {
                const int sb = (int)(ent >> 8) & 0x3F;
                ent = LT->e[(ent >> 16) + ((bitbuf >> 10) & ((1u << sb) - 1))];
                FI_TAKE(10);
            }
            int kind = (int)(ent >> 8) & 0xFF, l = (int)(ent & 0xFF);
            if (kind == FI_BAD || l > bitcnt) goto done;
            FI_TAKE(l);
            if (kind == FI_LIT) {
                if (out >= out_end) goto done;
                *out++ = (unsigned char)(ent >> 16);
                /* up to two more first-level literals out of the same refill (56 bits hold three 15-bit codes) */
                ent = LT->e[bitbuf & 1023]; kind = (int)(ent >> 8) & 0xFF; l = (int)(ent & 0xFF);
                if (kind != FI_LIT || l > bitcnt || out >= out_end) continue;
                FI_TAKE(l); *out++ = (unsigned char)(ent >> 16);
                ent = LT->e[bitbuf & 1023]; kind = (int)(ent >> 8) & 0xFF; l = (int)(ent & 0xFF);
                if (kind != FI_LIT || l > bitcnt || out >= out_end) continue;
                FI_TAKE(l); *out++ = (unsigned char)(ent >> 16);
                continue;
            }
            if (kind == FI_EOB) break;
            const int xb = kind & 0x3F;                             /* a length: extra bits, then the distance */
            if (xb > bitcnt) goto done;
            const unsigned len = (unsigned)(ent >> 16) + (unsigned)(bitbuf & ((1u << xb) - 1));
            FI_TAKE(xb);
            FI_REFILL();
            uint32_t de = DT->e[bitbuf & 255];
            if ((de >> 8 & 0xC0) == FI_SUB && (de >> 8 & 0xFF) != FI_BAD) {
                const int sb = (int)(de >> 8) & 0x3F;
                de = DT->e[(de >> 16) + ((bitbuf >> 8) & ((1u << sb) - 1))];
                FI_TAKE(8);
            }
            const int dk = (int)(de >> 8) & 0xFF, dl = (int)(de & 0xFF);
            if (dk == FI_BAD || (dk & 0xC0) != FI_LEN || dl > bitcnt) goto done;
            FI_TAKE(dl);
            const int dxb = dk & 0x3F;
            if (dxb > bitcnt) goto done;
            const size_t dist = (size_t)(de >> 16) + (size_t)(bitbuf & ((1u << dxb) - 1));
            FI_TAKE(dxb);
            if (dist > (size_t)(out - out0) || (size_t)(out_end - out) < len) goto done;
            const unsigned char *src = out - dist;
            if ((size_t)(out_end - out) >= len + 8) {               /* eight bytes at a time (may write up to 7 past len) */
                unsigned char *d = out;
                const unsigned char *e = out + len;
                if (dist < 8) {                                      /* a short period ("0/0\t"...): lay down a multiple of it that */
                    static const uint8_t big_of[8] = {0, 8, 8, 9, 8, 10, 12, 14};      /* is >= 8 byte by byte, then copy from there */
                    const unsigned big = big_of[dist], pre = len < big ? len : big;
                    for (unsigned k = 0; k < pre; k++) d[k] = src[k];
                    d += pre; src = d - big;
                }
                while (d < e) { uint64_t w; memcpy(&w, src, 8); memcpy(d, &w, 8); src += 8; d += 8; }
            } else for (unsigned k = 0; k < len; k++) out[k] = src[k];
            out += len;
        }
    }
    rc = out == out_end ? 0 : 1;
done:
    return rc;
}
#undef FI_REFILL
#undef FI_TAKE

/* exported for the tests: raw DEFLATE `in` -> exactly out_len bytes; 0 = done, non-zero = not decodable by the fast path */
int hpgv_host_inflate_raw(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
    return fast_inflate(in, in_len, out, out_len);
}
