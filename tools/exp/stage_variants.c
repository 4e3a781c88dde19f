/* Which shape of the staging loop suits the host: the sample strings of a batch (one heap block per sample, 32 bytes
 * apart) are walked (a) summing first bytes only = the floor the memory system sets, (b) eight branch-free probes per step
 * (what hpgv_host_stage_records does), (c) the same with software prefetch 8 / 16 / 32 strings ahead, (d) sixteen per
 * step.  ns per genotype, single thread, four batches in rotation (320 MB: not cache resident on a small L3).
 *   gcc -O2 -o /tmp/stage_variants tools/exp/stage_variants.c && /tmp/stage_variants [samples] */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static uint8_t TB[256];
static const char Z[8];
static inline unsigned enc(const char *s, unsigned *bad) {
    const char *p = (((uintptr_t)s & 4095u) <= 4092u) ? s : Z;
    uint32_t w; memcpy(&w, p, 4);
    const uint32_t e = w & 0xFF00FF00u;
    const unsigned n0 = TB[w & 0xFF], n2 = TB[(w >> 16) & 0xFF], m = n0 | n2;
    *bad |= (((e == 0x2F00u) | (e == 0x7C00u) | (e == 0x3A002F00u) | (e == 0x3A007C00u)) & ((m >> 7) ^ 1u)) ^ 1u;
    return ((((n0 << 4) | (n2 & 15u)) & 0xFFu) | ((0u - ((m >> 6) & 1u)) & 0xFFu));
}
/* the hot words through a perfect hash: 484 four-byte words "a/b\0", "a|b\0", "a/b:", "a|b:" (a, b in 0..9 or '.'), one
 * multiply and shift into a 1024-entry table of (word, strict code, loose code); the word is read as the ALIGNED dword at
 * s & ~3 (an aligned dword never crosses a page; a string that does not start on one goes the general way) */
#define GT_HASH_MAGIC 0xe9c3deefu
static uint64_t HT[1024];
static void ht_build(void) {
    static const char A[] = "0123456789.";
    for (int i = 0; i < 1024; i++) HT[i] = 0xFFFFFFFFull;             /* a key no aligned hot word has (byte 1 = 0xFF) */
    for (int a = 0; a < 11; a++) for (int b = 0; b < 11; b++) for (int sp = 0; sp < 2; sp++) for (int e = 0; e < 2; e++) {
        const uint32_t w = (uint32_t)A[a] | (uint32_t)(sp ? '|' : '/') << 8 | (uint32_t)A[b] << 16 | (uint32_t)(e ? ':' : 0) << 24;
        const unsigned n0 = a < 10 ? a : 15, n2 = b < 10 ? b : 15, loose = n0 << 4 | n2, strict = (a == 10 || b == 10) ? 0xFF : loose;
        const uint32_t i = (w * GT_HASH_MAGIC) >> 22;
        if ((HT[i] & 0xFFFFFFFFull) != 0xFFFFFFFFull) { fprintf(stderr, "collision\n"); exit(1); }
        HT[i] = (uint64_t)w | (uint64_t)strict << 32 | (uint64_t)loose << 40;
    }
    /* an empty slot's key must not hash to its own slot */
    for (uint32_t i = 0; i < 1024; i++) if ((HT[i] & 0xFFFFFFFFull) == 0xFFFFFFFFull) { uint32_t k = 0xFFFFFFFFu; while (((k * GT_HASH_MAGIC) >> 22) == i) k -= 0x100; HT[i] = k; }
}
static inline unsigned enc_h(const char *s, uint32_t *bad) {
    const uint32_t w = *(const uint32_t *)((uintptr_t)s & ~(uintptr_t)3);
    const uint64_t e = HT[(w * GT_HASH_MAGIC) >> 22];
    *bad |= ((uint32_t)e ^ w) | ((uint32_t)(uintptr_t)s & 3u);
    return (unsigned)(e >> 32) & 0xFFu;
}
static inline unsigned enc_a(const char *s, unsigned *bad) {          /* the nibble tables on the aligned word */
    const uint32_t w = *(const uint32_t *)((uintptr_t)s & ~(uintptr_t)3);
    const uint32_t e = w & 0xFF00FF00u;
    const unsigned n0 = TB[w & 0xFF], n2 = TB[(w >> 16) & 0xFF], m = n0 | n2;
    *bad |= ((((e == 0x2F00u) | (e == 0x7C00u) | (e == 0x3A002F00u) | (e == 0x3A007C00u)) & ((m >> 7) ^ 1u)) ^ 1u) | ((unsigned)(uintptr_t)s & 3u);
    return ((((n0 << 4) | (n2 & 15u)) & 0xFFu) | ((0u - ((m >> 6) & 1u)) & 0xFFu));
}
#define ROWA(U)                                                                                             \
    for (; j + U <= N; j += U) {                                                                            \
        uint64_t v[U / 8];                                                                                  \
        _Pragma("GCC unroll 16") for (int k = 0; k < U; k++) { if (!(k & 7)) v[k / 8] = 0; v[k / 8] |= (uint64_t)enc_a(s[j + k], &bad) << (8 * (k & 7)); } \
        memcpy(row + j, v, U);                                                                              \
    }
#define ROWH(U)                                                                                             \
    for (; j + U <= N; j += U) {                                                                            \
        uint64_t v[U / 8];                                                                                  \
        _Pragma("GCC unroll 16") for (int k = 0; k < U; k++) { if (!(k & 7)) v[k / 8] = 0; v[k / 8] |= (uint64_t)enc_h(s[j + k], &bad) << (8 * (k & 7)); } \
        memcpy(row + j, v, U);                                                                              \
    }
#define ROW(U, PF)                                                                                          \
    for (; j + U <= N; j += U) {                                                                            \
        if (PF) { _Pragma("GCC unroll 16") for (int k = 0; k < U; k += 2) __builtin_prefetch(s[j + k + PF < N ? j + k + PF : j], 0, 0); } \
        uint64_t v[U / 8];                                                                                  \
        _Pragma("GCC unroll 16") for (int k = 0; k < U; k++) { if (!(k & 7)) v[k / 8] = 0; v[k / 8] |= (uint64_t)enc(s[j + k], &bad) << (8 * (k & 7)); } \
        memcpy(row + j, v, U);                                                                              \
    }
int main(int argc, char **argv) {
    const int B = 200, N = argc > 1 ? atoi(argv[1]) : 10000, K = 4;
    for (int i = 0; i < 256; i++) TB[i] = 0x80;
    for (int c = '0'; c <= '9'; c++) TB[c] = (uint8_t)(c - '0');
    TB['.'] = 0x4F;
    static const char *const calls[8] = {"0/0", "0/0", "0/0", "0/1", "0/1", "1/1", "1/0", "./."};
    char ****bt = malloc(sizeof(void *) * K);
    unsigned long long st = 88172645463325252ULL;
    for (int k = 0; k < K; k++) { bt[k] = malloc(sizeof(void *) * B); for (int i = 0; i < B; i++) { bt[k][i] = malloc(sizeof(void *) * N); for (int j = 0; j < N; j++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; bt[k][i][j] = strdup(calls[(st >> 20) & 7]); } } }
    uint8_t *out = malloc((size_t)B * N);
    static const char *const names[9] = {"first bytes only (floor)", "8 per step", "8 per step, prefetch 8 ahead", "8 per step, prefetch 16 ahead", "8 per step, prefetch 32 ahead", "16 per step",
                                         "perfect hash of the aligned word, 8 per step", "perfect hash of the aligned word, 16 per step",
                                         "nibble tables on the aligned word, 8 per step"};
    ht_build();
    for (int mode = 0; mode < 9; mode++) {
        double t0 = now(); long c = 0; unsigned tb = 0;
        while (now() - t0 < 1.0) {
            char ***b = bt[c % K];
            for (int i = 0; i < B; i++) {
                char **s = b[i]; uint8_t *row = out + (size_t)i * N; unsigned bad = 0; int j = 0;
                if (mode == 0) { unsigned a = 0; for (; j < N; j++) a += (unsigned char)s[j][0]; row[0] = (uint8_t)a; }
                else if (mode == 1) { ROW(8, 0) }
                else if (mode == 2) { ROW(8, 8) }
                else if (mode == 3) { ROW(8, 16) }
                else if (mode == 4) { ROW(8, 32) }
                else if (mode == 5) { ROW(16, 0) }
                else if (mode == 6) { ROWH(8) }
                else if (mode == 7) { ROWH(16) }
                else { ROWA(8) }
                for (; j < N; j++) row[j] = (uint8_t)enc(s[j], &bad);
                tb += bad;
            }
            c++;
        }
        printf("{\"loop\": \"%s\", \"samples\": %d, \"ns_per_genotype\": %.3f, \"check\": %u}\n", names[mode], N, (now() - t0) / c / ((double)B * N) * 1e9, tb + out[7]);
    }
    return 0;
}
