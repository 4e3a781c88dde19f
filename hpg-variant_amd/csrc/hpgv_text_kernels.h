// hpgv_text_kernels.h -- VCF data lines (text) -> HPGV8 genotype matrix on the GPU
// (the step right before the scan; the reference does it
// per genotype with strdup + get_alleles, assoc.c:45-56).
//
//   k_tok_count / k_tok_scan / k_tok_mark : line starts (newline compaction)
//   k_tok_parse                           : one workgroup per line
// Text arrives over PCIe at ~55 GB/s, far below what these kernels sustain, so
// they are written for clarity: 16 bytes per thread per tile, block-wide prefix
// count of TABs gives every TAB its field index, the thread owning a TAB parses
// the genotype that follows it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpgv {

constexpr int TOK_TILE = 4096;          // bytes per workgroup tile (256 threads x 16 B)

// bit j of the result = byte j of w equals c (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t tok_byte_mask(uint64_t w, char c) {
    const uint64_t x = w ^ (0x0101010101010101ull * (uint8_t)c);
    const uint64_t k7 = 0x7F7F7F7F7F7F7F7Full;
    const uint64_t z = ~(((x & k7) + k7) | x | k7);          // 0x80 in every byte of x that is zero
    return (uint32_t)((z * 0x0002040810204081ull) >> 56);          // the eight 0x80 flags gathered into one byte
}

// the newline bits of a thread's 16 bytes at `base` (two 8-byte loads; byte loop at the end of the text)
__device__ __forceinline__ uint32_t tok_newlines16(const char *__restrict__ text, size_t base, size_t n) {
    if (base + 16 <= n) {
        uint64_t w0, w1;
        __builtin_memcpy(&w0, text + base, 8); __builtin_memcpy(&w1, text + base + 8, 8);
        return tok_byte_mask(w0, '\n') | (tok_byte_mask(w1, '\n') << 8);
    }
    uint32_t m = 0;
    for (int j = 0; j < 16; ++j) if (base + j < n && text[base + j] == '\n') m |= 1u << j;
    return m;
}

static __global__ __launch_bounds__(256) void k_tok_count(const char *__restrict__ text, size_t n, int *__restrict__ block_counts) {
    __shared__ int s[4];
    const size_t base = (size_t)blockIdx.x * TOK_TILE + (size_t)threadIdx.x * 16;
    int c = __popc(tok_newlines16(text, base, n));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// single workgroup of 1024 threads: exclusive scan of block_counts in place; n_lines = newlines (+1 for an unterminated
// tail).  Every thread owns a contiguous run of the counts: sum it, scan the 1024 sums (wave scans + 16 wave totals), write
// the run's prefixes -- two sweeps over an array that sits in L2 (a scan of 256 counts per round with two barriers per
// doubling step took 0.69 ms for the 156 k tiles of a 640 MB batch: as long as parsing it).
constexpr int TOK_SCAN_THREADS = 1024;
static __global__ __launch_bounds__(TOK_SCAN_THREADS) void k_tok_scan(int *__restrict__ block_counts, int n_blocks, const char *__restrict__ text,
                                                                    size_t n, int *__restrict__ n_lines,
                                                                    unsigned long long *__restrict__ line_off, int max_lines) {
    __shared__ int wave_tot[TOK_SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (n_blocks + TOK_SCAN_THREADS - 1) / TOK_SCAN_THREADS;
    const int lo = tid * per < n_blocks ? tid * per : n_blocks;
    const int hi = lo + per < n_blocks ? lo + per : n_blocks;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_counts[i];
    int x = sum;                                                     // inclusive scan within the wave
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    int before = x - sum, total = 0;
    for (int k = 0; k < TOK_SCAN_THREADS / 64; ++k) { if (k < w) before += wave_tot[k]; total += wave_tot[k]; }
    int run = before;
    for (int i = lo; i < hi; ++i) { const int v = block_counts[i]; block_counts[i] = run; run += v; }
    if (tid == 0) {
        const int tail = (n > 0 && text[n - 1] != '\n') ? 1 : 0;     // unterminated last line
        *n_lines = total + tail;
        if (tail && total + 1 <= max_lines) line_off[total + 1] = n;
    }
}

// line_off[k] = start of line k; line_off[n_lines] = end of text (one past the last newline or n)
static __global__ __launch_bounds__(256) void k_tok_mark(const char *__restrict__ text, size_t n, const int *__restrict__ block_offsets,
                                                  unsigned long long *__restrict__ line_off, int max_lines) {
    __shared__ int s[4];
    const size_t base = (size_t)blockIdx.x * TOK_TILE + (size_t)threadIdx.x * 16;
    uint32_t nls = tok_newlines16(text, base, n);
    const int c = __popc(nls);
    int x = c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) s[w] = x;
    __syncthreads();
    int before = block_offsets[blockIdx.x] + x - c;
    for (int k = 0; k < w; ++k) before += s[k];
    while (nls) {
        const int j = __ffs((int)nls) - 1;
        nls &= nls - 1;
        before++;
        if (before <= max_lines) line_off[before] = base + j + 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) line_off[0] = 0;
}

// atoi() on [p, e): optional blanks, sign, digits (what get_alleles applies to an allele token) -- as the C library does it:
// (int) strtol(): the digits accumulate in 64 bits and stick at LONG_MAX / LONG_MIN, the result is cut to 32 bits.  A VCF
// writer never produces an allele index of ten digits; a text that holds one anyway ("11111111111/0") gets what the reference's
// atoi gives it (found by tests/test_gpu_text.py::test_random_bytes_from_a_small_alphabet)
__device__ __forceinline__ int tok_atoi(const char *__restrict__ t, size_t p, size_t e) {
    while (p < e && (t[p] == ' ' || (t[p] >= '\t' && t[p] <= '\r'))) p++;
    bool neg = false;
    if (p < e && (t[p] == '+' || t[p] == '-')) { neg = t[p] == '-'; p++; }
    unsigned long long v = 0;
    bool stuck = false;
    while (p < e && t[p] >= '0' && t[p] <= '9') {
        const unsigned d = (unsigned)(t[p] - '0');
        if (v > 922337203685477580ull || (v == 922337203685477580ull && d > 7u)) stuck = true;      // v * 10 + d > LONG_MAX
        else v = v * 10 + d;
        p++;
    }
    const long long r = stuck ? (neg ? (long long)0x8000000000000000ull : 0x7FFFFFFFFFFFFFFFll) : (neg ? -(long long)v : (long long)v);
    return (int)r;
}

// one sample field starting at p (line ends at e, exclusive): the product's statement of
// get_alleles + the HPGV8 encoding (host twin: host/host_stage.c encode_gt_general)
__device__ __forceinline__ uint32_t tok_encode(const char *__restrict__ t, size_t p, size_t e, int gt_position, int strict) {
    size_t fe = p;                                      // end of this sample field: the next TAB -- or a NUL byte before it: the
    while (fe < e && t[fe] != '\t' && t[fe] != 0) fe++;  // reference's sample fields are C strings (vcf_record_t.samples), get_alleles stops there
    for (int i = 0; i < gt_position; ++i) {             // skip to the GT sub-field
        while (p < fe && t[p] != ':') p++;
        if (p >= fe) return 0xFFu;                      // sub-field absent: all alleles missing
        p++;
    }
    size_t ge = p;
    while (ge < fe && t[ge] != ':') ge++;
    size_t sep = p;
    while (sep < ge && t[sep] != '/' && t[sep] != '|') sep++;
    int status = 0, a1 = -1, a2 = -1;
    if (sep == p || (sep - p == 1 && t[p] == '.')) status += 1; else a1 = tok_atoi(t, p, sep);
    if (sep == ge) { status = (status == 0) ? 4 : 3; }
    else {
        const size_t q = sep + 1;
        if (q == ge || (ge - q == 1 && t[q] == '.')) status += 2; else a2 = tok_atoi(t, q, ge);
    }
    if (strict && status != 0) return 0xFFu;
    uint32_t n1 = (a1 < 0) ? 0xFu : (a1 > 14 ? 14u : (uint32_t)a1);
    const uint32_t n2 = (a2 < 0) ? 0xFu : (a2 > 14 ? 14u : (uint32_t)a2);
    if (n1 == 14u && n2 == 14u && a1 != a2) n1 = 13u;    // two different alleles above 14 stay different (host twin: encode_gt)
    return (n1 << 4) | n2;
}

// block-wide exclusive prefix of v; *total = block sum.  s4: 4 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int *s4, int *total) {
    int x = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    __syncthreads();
    if (lane == 63) s4[w] = x;
    __syncthreads();
    int before = x - v;
    for (int k = 0; k < w; ++k) before += s4[k];
    *total = s4[0] + s4[1] + s4[2] + s4[3];
    return before;
}

// status per line: 0 ok, 1 fewer than 9 TABs (no sample columns), 2 FORMAT has no GT,
// 3 fewer sample fields than n_samples (the missing ones are 0xFF)
// one line, by the whole workgroup (every `return` is the workgroup's)
__device__ __forceinline__ void tok_parse_line(const int line, const char *__restrict__ text, const unsigned long long *__restrict__ line_off,
                                               const int n_lines, int n_samples, int strict,
                                               uint8_t *__restrict__ gt, size_t pitch, uint8_t *__restrict__ is_x,
                                               uint32_t *__restrict__ field_off /* n_lines x 10 */, int *__restrict__ status) {
    __shared__ int s4[4];
    __shared__ unsigned int s_field[10];
    __shared__ int s_gtpos;
    if (line >= n_lines) return;
    const size_t ls = line_off[line];
    size_t le = line_off[line + 1];
    if (le > ls && text[le - 1] == '\n') le--;              // exclusive end, newline dropped
    const char *t = text;
    if (threadIdx.x < 10) s_field[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();

    // ---- phase 1: the first nine TABs -> starts of CHROM..FORMAT and of the sample columns
    int carry = 0;
    for (size_t base = ls; base < le && carry < 9; base += TOK_TILE) {
        const size_t p0 = base + (size_t)threadIdx.x * 16;
        int c = 0;
        for (int j = 0; j < 16; ++j) if (p0 + j < le && t[p0 + j] == '\t') c++;
        int total;
        int idx = carry + block_excl_scan(c, s4, &total);
        for (int j = 0; j < 16; ++j)
            if (p0 + j < le && t[p0 + j] == '\t') {
                idx++;
                if (idx <= 9) s_field[idx] = (unsigned int)(p0 + j + 1 - ls);
            }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s_field[0] = 0;
        int gtpos = -1;
        if (s_field[9] != 0xFFFFFFFFu) {                   // FORMAT = [s_field[8], s_field[9] - 1)
            size_t p = ls + s_field[8];
            const size_t fe = ls + s_field[9] - 1;
            int pos = 0;
            while (p <= fe) {
                size_t q = p;
                while (q < fe && t[q] != ':') q++;
                if (q - p == 2 && t[p] == 'G' && t[p + 1] == 'T') { gtpos = pos; break; }
                if (q >= fe) break;
                p = q + 1; pos++;
            }
        }
        s_gtpos = gtpos;
        const size_t clen = (s_field[1] != 0xFFFFFFFFu) ? (size_t)s_field[1] - 1 : (le - ls);
        // assoc.c:94: !strncmp("X", chromosome, chromosome_len)
        if (is_x) is_x[line] = (clen == 0 || (clen == 1 && t[ls] == 'X')) ? 1 : 0;
        if (field_off) for (int k = 0; k < 10; ++k) field_off[(size_t)line * 10 + k] = s_field[k];
    }
    __syncthreads();
    uint8_t *row = gt + (size_t)line * pitch;
    const int gtpos = s_gtpos;
    if (s_field[9] == 0xFFFFFFFFu || gtpos < 0) {
        for (int j = threadIdx.x; j < n_samples; j += 256) row[j] = 0xFF;
        if (threadIdx.x == 0 && status) status[line] = (s_field[9] == 0xFFFFFFFFu) ? 1 : 2;
        return;
    }

    // ---- phase 2: every TAB from the ninth on starts a sample field
    const size_t r0 = ls + s_field[9] - 1;                  // position of the ninth TAB
    int before = 8;                                         // TABs in front of r0
    for (size_t base = r0; base < le; base += TOK_TILE) {
        const size_t p0 = base + (size_t)threadIdx.x * 16;
        // a thread's 16 bytes and the 8 after them come in three 8-byte loads (the byte loops below cost 32 loads); TABs
        // by a bit trick; a genotype of the everyday form -- digit, '/' or '|', digit (or ". ." both missing), then TAB
        // or ':' -- is encoded out of these registers; anything else, and the last bytes of a line, take the general way
        const bool wide = p0 + 24 <= le;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        uint32_t tabs = 0;
        if (wide) {
            __builtin_memcpy(&w0, t + p0, 8); __builtin_memcpy(&w1, t + p0 + 8, 8); __builtin_memcpy(&w2, t + p0 + 16, 8);
            tabs = tok_byte_mask(w0, '\t') | (tok_byte_mask(w1, '\t') << 8);
        } else {
            for (int j = 0; j < 16; ++j) if (p0 + j < le && t[p0 + j] == '\t') tabs |= 1u << j;
        }
        const int c = __popc(tabs);
        int total;
        int idx = before + block_excl_scan(c, s4, &total);  // TABs before this thread's bytes
        while (tabs) {
            const int j = __ffs((int)tabs) - 1;
            tabs &= tabs - 1;
            const int sample = idx - 8;                     // field index idx+1, samples start at field 9
            idx++;
            if (sample >= n_samples) continue;
            uint32_t code = 0x100u;                         // "not decided"
            if (wide && gtpos == 0) {
                const int k = j + 1;                        // the four bytes after the TAB: k .. k + 3 <= 19
                const int wi = k >> 3, sh = (k & 7) * 8;
                const uint64_t a = wi == 0 ? w0 : wi == 1 ? w1 : w2, b = wi == 0 ? w1 : w2;
                const uint32_t g = (uint32_t)(sh ? (a >> sh) | (b << (64 - sh)) : a);
                const uint32_t b0 = g & 0xFF, b1 = (g >> 8) & 0xFF, b2 = (g >> 16) & 0xFF, b3 = g >> 24;
                if ((b1 == '/' || b1 == '|') && (b3 == '\t' || b3 == ':')) {
                    const uint32_t d0 = b0 - '0', d1 = b2 - '0';
                    if (d0 <= 9 && d1 <= 9) code = (d0 << 4) | d1;
                    else if (b0 == '.' && b2 == '.') code = 0xFFu;      // both alleles missing: 0xFF strict or not
                }
            }
            if (code == 0x100u) code = tok_encode(t, p0 + j + 1, le, gtpos, strict);
            row[sample] = (uint8_t)code;
        }
        before += total;
        __syncthreads();
    }
    const int found = before - 8;                           // sample fields present on the line
    for (int j = (found < 0 ? 0 : found) + threadIdx.x; j < n_samples; j += 256) row[j] = 0xFF;
    if (threadIdx.x == 0 && status) status[line] = found < n_samples ? 3 : 0;
}

// every line, one workgroup each (the three-sweep form's parse)
static __global__ __launch_bounds__(256) void k_tok_parse(const char *__restrict__ text, const unsigned long long *__restrict__ line_off,
                                                   const int *__restrict__ n_lines_p, int max_lines, int n_samples, int strict,
                                                   uint8_t *__restrict__ gt, size_t pitch, uint8_t *__restrict__ is_x,
                                                   uint32_t *__restrict__ field_off, int *__restrict__ status) {
    tok_parse_line((int)blockIdx.x, text, line_off, *n_lines_p < max_lines ? *n_lines_p : max_lines, n_samples, strict, gt, pitch, is_x, field_off, status);
}
// the LISTED lines only (the tile-parallel forms flag the lines whose FORMAT does not begin with GT: as a rule none), a fixed
// grid striding over the list: an empty list costs a thousand workgroups that read one counter, not one workgroup per line
static __global__ __launch_bounds__(256) void k_tok_parse_listed(const char *__restrict__ text, const unsigned long long *__restrict__ line_off,
                                                          const int *__restrict__ n_lines_p, int max_lines, int n_samples, int strict,
                                                          uint8_t *__restrict__ gt, size_t pitch, uint8_t *__restrict__ is_x,
                                                          uint32_t *__restrict__ field_off, int *__restrict__ status,
                                                          const int *__restrict__ list, const int *__restrict__ list_n) {
    const int n_lines = *n_lines_p < max_lines ? *n_lines_p : max_lines;
    const int n = *list_n < max_lines ? *list_n : max_lines;
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        tok_parse_line(list[i], text, line_off, n_lines, n_samples, strict, gt, pitch, is_x, field_off, status);
        __syncthreads();                                     // the line's shared arrays are written again by the next one
    }
}


// ---------------------------------------------------------------------------
// line heads: when the text lives on the device only (bgzip decoded there), the host still needs CHROM .. FORMAT of
// every line for its result records.  head_off[i] = exclusive prefix sum of the head lengths (a head = the line up to
// the first sample column, or the whole line when it has fewer than ten fields), head_off[n] = total; then the bytes.
// ---------------------------------------------------------------------------
// Three launches over all the lines (one workgroup walking them took 3.2 ms for 800 000 lines, a tenth of such a call):
// k_head_sums: every workgroup of 1024 lines sums its head lengths; k_head_bases: one workgroup scans those sums (a thousand
// of them per million lines); k_head_offsets: every workgroup scans its 1024 lengths from its base.
__device__ __forceinline__ unsigned long long head_len(const unsigned long long *__restrict__ line_off, const uint32_t *__restrict__ field_off, int i) {
    const uint32_t f9 = field_off[(size_t)i * 10 + 9];
    return f9 != 0xFFFFFFFFu ? (unsigned long long)f9 : line_off[i + 1] - line_off[i];
}
// inclusive scan of one value per thread over a workgroup of 1024 (wave scans by shuffles, the 16 wave totals through LDS)
__device__ __forceinline__ unsigned long long head_block_scan(unsigned long long v, unsigned long long *s_w /* [16] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long o = __shfl_up(v, off); if (lane >= off) v += o; }
    if (lane == 63) s_w[w] = v;
    __syncthreads();
    unsigned long long before = 0;
    for (int k = 0; k < w; ++k) before += s_w[k];
    __syncthreads();
    return v + before;
}
static __global__ __launch_bounds__(1024) void k_head_sums(const unsigned long long *__restrict__ line_off, const uint32_t *__restrict__ field_off,
                                                    int n_lines, unsigned long long *__restrict__ block_sum) {
    __shared__ unsigned long long s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const unsigned long long inc = head_block_scan(i < n_lines ? head_len(line_off, field_off, i) : 0ull, s_w);
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = inc;
}
static __global__ __launch_bounds__(1024) void k_head_bases(unsigned long long *__restrict__ block_sum, int n_blocks) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {                   // exclusive, in place, 1024 block sums per turn
        const int b = b0 + threadIdx.x;
        const unsigned long long v = b < n_blocks ? block_sum[b] : 0ull;
        const unsigned long long inc = head_block_scan(v, s_w);
        const unsigned long long base = carry;
        if (b < n_blocks) block_sum[b] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + inc;
        __syncthreads();
    }
}
static __global__ __launch_bounds__(1024) void k_head_offsets(const unsigned long long *__restrict__ line_off, const uint32_t *__restrict__ field_off,
                                                       int n_lines, const unsigned long long *__restrict__ block_base,
                                                       unsigned long long *__restrict__ head_off) {
    __shared__ unsigned long long s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const unsigned long long len = i < n_lines ? head_len(line_off, field_off, i) : 0ull;
    const unsigned long long inc = head_block_scan(len, s_w) + block_base[blockIdx.x];
    if (i < n_lines) head_off[i] = inc - len;
    if (i == n_lines - 1) head_off[n_lines] = inc;
}
static __global__ __launch_bounds__(64) void k_copy_heads(const char *__restrict__ text, const unsigned long long *__restrict__ line_off,
                                                   const unsigned long long *__restrict__ head_off, int n_lines, char *__restrict__ heads) {
    const int i = blockIdx.x;
    if (i >= n_lines) return;
    const unsigned long long n = head_off[i + 1] - head_off[i];
    const char *src = text + line_off[i];
    char *dst = heads + head_off[i];
    for (unsigned long long k = threadIdx.x; k < n; k += 64) dst[k] = src[k];
}

}  // namespace hpgv
