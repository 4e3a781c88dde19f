/* host_bgzf.c -- a bgzip file staged on the device(s): upload, block table, decode, CRC check, in parts over a group.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* HPGV_BGZF_VERIFY (default 1): every BGZF block's text is checked against the CRC-32 of its trailer, as htslib's bgzf reader
 * and zlib's gzread do -- a damaged stream can still inflate to ISIZE bytes.  On the device path the check runs on the device
 * (hpgv_bgzf_verify_dev); a block it rejects comes here like any block the device decoder refused, and fails the run. */
char g_input_err[192];
static int bgzf_verify_on(void) { return g_env.bgzf_verify != 0; }
/* the tokenizer's tile records of a text of about `bytes` bytes that the device is going to decode and check: zeroed device
 * memory (hpgv.h "hpgv_bgzf_verify_tiles_dev").  Not having them only means the windows are tokenized the ordinary way. */
static void tiles_alloc(source_t *s, size_t bytes) {
    s->d_tiles = NULL; s->n_tiles = 0; s->d_tiles_cap = 0;
    if (!bgzf_verify_on() || g_env.no_decode_tiles) return;
    const size_t nb = hpgv_text_tiles_bytes((uint64_t)bytes);
    size_t cap = 0;
    void *p = dev_tiles_get(nb, &cap);
    if (!p) return;
    if (hpgv_memset_dev(CTX, p, 0, nb, NULL) != HPGV_OK) { dev_tiles_put(p, cap); return; }
    s->d_tiles = p; s->n_tiles = nb / 32 - 1; s->d_tiles_cap = cap;
}
static void tiles_free(source_t *s) { if (s->d_tiles) { dev_tiles_put(s->d_tiles, s->d_tiles_cap); s->d_tiles = NULL; s->n_tiles = 0; s->d_tiles_cap = 0; } }
/* in[clen .. clen + 4) is the block's CRC-32 (the BGZF trailer follows the payload) */
static int block_crc_bad(const unsigned char *in, size_t clen, const unsigned char *out, size_t isize) {
    const unsigned char *t = in + clen;
    const uint32_t stored = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)isize) == stored) return 0;
    /* written by a reader / stager thread while the run may be reporting what it saw fail downstream: kept apart, and put in
     * front when the run ends with an error (run_file) */
    snprintf(g_input_err, sizeof g_input_err, "bgzip input: a block's text does not have the CRC-32 its trailer gives (damaged file)");
    return 1;
}
int inflate_block(const unsigned char *in, size_t clen, unsigned char *out, size_t isize) {
    if (!g_env.zlib_inflate && fast_inflate(in, clen, out, isize) == 0)       /* anything unusual: zlib decides */
        return bgzf_verify_on() ? block_crc_bad(in, clen, out, isize) : 0;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return 1;
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)clen;
    zs.next_out = (Bytef *)out; zs.avail_out = (uInt)isize;
    int bad = inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != isize;
    inflateEnd(&zs);
    if (!bad && bgzf_verify_on()) bad = block_crc_bad(in, clen, out, isize);
    return bad;
}

void source_task_pread(void *v, int k) {
    source_t *s = (source_t *)v;
    size_t off = (size_t)k * PREAD_SEG, len = off + PREAD_SEG <= s->job_want ? (size_t)PREAD_SEG : s->job_want - off;
    while (len > 0) {
        ssize_t got = pread(s->fd, s->job_buf + off, len, s->pos + (off_t)off);
        if (got <= 0) { __atomic_store_n(&s->job_bad, 1, __ATOMIC_RELAXED); return; }
        off += (size_t)got; len -= (size_t)got;
    }
}
void source_task_inflate(void *v, int g) {
    source_t *s = (source_t *)v;
    const size_t *b_in = s->blk, *b_clen = s->blk + MAXB, *b_out = s->blk + 2 * MAXB, *b_isize = s->blk + 3 * MAXB;
    const int nb = (int)s->job_want;
    for (int k = g * INFLATE_GROUP; k < nb && k < (g + 1) * INFLATE_GROUP; k++) {
        if (b_isize[k] == 0) continue;                                /* e.g. the BGZF end-of-file marker */
        if (inflate_block(s->map + b_in[k], b_clen[k], (unsigned char *)s->job_buf + b_out[k], b_isize[k]))
            __atomic_store_n(&s->job_bad, 1, __ATOMIC_RELAXED);
    }
}

/* BGZF on the GPU: the file's blocks are decoded on the device (hpgv_inflate_blocks_dev: one lane per block, tens of
 * thousands of blocks per launch; 8 GB of VCF text in 0.1 s) and the text stays in device memory.  The reader copies it
 * out window by window for the result writers and the engine tokenizes the device copy in place (hpgv_text_alias) --
 * the compressed bytes are all that goes up the bus, read from the file with pread into a page-locked buffer (the
 * mapping is not touched: faulting a gigabyte in and unmapping it costs more than the decoding).  A stager thread
 * uploads and decodes the file in stretches: 4 096 blocks first, decoded alone, so that the header reader and the
 * pipeline start after the time one block takes; then 32 768 and more, several launches side by side on streams of
 * their own.  A block the device decoder refuses is decoded by the host and patched in.  No memory, a file of more
 * than 48 GB of text or fewer than 256 blocks leave the CPU path in charge.  HPGV_NO_GPU_INFLATE=1 switches it off. */
enum { GPU_STRETCH = 32768, GPU_AHEAD_BYTES = 768 << 20 };
/* The upload: out of pageable memory a copy runs at a fifth of the bus rate, so the file goes through a page-locked ring
 * (pread, not the mapping: faulting a gigabyte in and unmapping it again costs ~90 ms).  Readers take the file's 4 MB
 * segments in turn and fill the ring's slots; the uploader copies the slots up in file order and announces every segment
 * that has arrived (s->up_done, under g_mu).  Nobody waits at a barrier: a reader waits only for its slot to be free, the
 * copier only for the next segment to be filled.  (Halves of a buffer filled by a team, with a barrier per half, left
 * the bus at 18 - 28 GB/s beside the pipeline's own threads.) */
enum { UP_SEG_DEFAULT = 4 << 20, UP_SLOTS = 16, UP_READERS_MAX = 12, UP_INFLIGHT = 2 };
static size_t g_up_seg = UP_SEG_DEFAULT;                 /* bytes per segment (HPGV_UPLOAD_SEGMENT_MB: diagnosis) */
#define UP_SEG g_up_seg
static size_t pread_full(int fd, void *buf, size_t n, size_t pos);
typedef struct {
    source_t *s; char *pin; size_t n_seg;
    pthread_mutex_t mu; pthread_cond_t cv;
    size_t next;                                         /* the next segment a reader takes */
    size_t copied;                                       /* segments [0, copied) are on the device: slot i % UP_SLOTS is free for segment i < copied + UP_SLOTS */
    unsigned char filled[UP_SLOTS];
    int bad, stop;
} up_ring_t;
static void *up_reader(void *v) {
    up_ring_t *r = (up_ring_t *)v;
    for (;;) {
        pthread_mutex_lock(&r->mu);
        const size_t i = r->next < r->n_seg ? r->next++ : (size_t)-1;
        while (i != (size_t)-1 && !r->stop && !r->bad && i >= r->copied + UP_SLOTS) pthread_cond_wait(&r->cv, &r->mu);
        const int quit = i == (size_t)-1 || r->stop || r->bad;
        pthread_mutex_unlock(&r->mu);
        if (quit) return NULL;
        const size_t off = i * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)r->s->size ? (size_t)UP_SEG : (size_t)r->s->size - off;
        const int ok = pread_full(r->s->fd, r->pin + (i % UP_SLOTS) * (size_t)UP_SEG, len, off + (size_t)r->s->file_off) == len;
        pthread_mutex_lock(&r->mu);
        if (ok) r->filled[i % UP_SLOTS] = 1; else r->bad = 1;
        pthread_cond_broadcast(&r->cv);
        pthread_mutex_unlock(&r->mu);
    }
}
static void *bgzf_uploader(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    void *up = NULL;
    int ok = stream_get(0, &up) == HPGV_OK;
    g_up_seg = g_env.upload_segment_mb >= 1 && g_env.upload_segment_mb <= 64 ? (size_t)g_env.upload_segment_mb << 20 : (size_t)UP_SEG_DEFAULT;
    const size_t pin_cap = (size_t)UP_SEG * UP_SLOTS;
    char *pin = text_buf_get(pin_cap + 1);                          /* from the runs' cache of page-locked buffers */
    up_ring_t r;
    memset(&r, 0, sizeof r);
    r.s = s; r.pin = pin; r.n_seg = ((size_t)s->size + UP_SEG - 1) / UP_SEG;
    pthread_mutex_init(&r.mu, NULL); pthread_cond_init(&r.cv, NULL);
    pthread_t th[UP_READERS_MAX];
    int n_th = 0;
    if (ok && pin) {
        int want = default_io_threads() * 3 / 4;
        if (s->is_part || s->mp) want = want / 2 > 2 ? want / 2 : 2;    /* several parts go up side by side: they share the host's threads */
        want = want < 1 ? 1 : want > UP_READERS_MAX ? UP_READERS_MAX : want;
        for (; n_th < want; n_th++) if (pthread_create(&th[n_th], NULL, up_reader, &r) != 0) break;
    }
    ok = ok && pin && n_th > 0;
    double t_wait = 0, t_copy = 0; const double t_begin = now_s();
    /* UP_INFLIGHT copies in flight: one is waited for while the others are queued or running */
    enum { UP_INFLIGHT_MAX = 8 };
    const int depth = g_env.upload_inflight >= 1 && g_env.upload_inflight <= UP_INFLIGHT_MAX ? (int)g_env.upload_inflight : UP_INFLIGHT;
    void *stq[UP_INFLIGHT_MAX] = { up };
    for (int k = 1; ok && k < depth; k++) ok = stream_get(0, &stq[k]) == HPGV_OK;
    for (size_t i = 0; ok && i < r.n_seg + (size_t)depth - 1; i++) {
        double t0 = now_s();
        if (i < r.n_seg) {
            pthread_mutex_lock(&r.mu);
            while (!r.filled[i % UP_SLOTS] && !r.bad) pthread_cond_wait(&r.cv, &r.mu);
            ok = !r.bad;
            pthread_mutex_unlock(&r.mu);
            t_wait += now_s() - t0; t0 = now_s();
            if (!ok) break;
            const size_t off = i * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)s->size ? (size_t)UP_SEG : (size_t)s->size - off;
            ok = hpgv_memcpy_h2d_async(CTX, (char *)s->d_comp + off, pin + (i % UP_SLOTS) * (size_t)UP_SEG, len, stq[i % (size_t)depth]) == HPGV_OK;
            if (!ok) break;
        }
        if (i + 1 < (size_t)depth) continue;
        const size_t j = i + 1 - (size_t)depth, off = j * (size_t)UP_SEG, len = off + UP_SEG <= (size_t)s->size ? (size_t)UP_SEG : (size_t)s->size - off;
        ok = hpgv_stream_sync(CTX, stq[j % (size_t)depth]) == HPGV_OK;
        t_copy += now_s() - t0;
        pthread_mutex_lock(&r.mu);
        r.filled[j % UP_SLOTS] = 0; r.copied = j + 1;
        pthread_cond_broadcast(&r.cv);
        pthread_mutex_unlock(&r.mu);
        if (!ok) break;
        if (j == 0 && g_env.run_trace) fprintf(stderr, "uploader: first segment up %.4f s after its start\n", now_s() - t_begin);
        pthread_mutex_lock(&s->g_mu);
        s->up_done = off + len;
        const int cancel = s->u_cancel;
        pthread_cond_broadcast(&s->g_cv);
        pthread_mutex_unlock(&s->g_mu);
        if (cancel) { ok = 0; break; }
    }
    for (int k = 1; k < depth; k++) if (stq[k]) { (void)hpgv_stream_sync(CTX, stq[k]); stream_put(0, stq[k]); }
    if (up) (void)hpgv_stream_sync(CTX, up);
    if (g_env.run_trace)
        fprintf(stderr, "uploader: %.1f MB in %.4f s: %.4f s waiting for the readers (%d), %.4f s in copies\n", s->size / 1e6, now_s() - t_begin, t_wait, n_th, t_copy);
    pthread_mutex_lock(&r.mu); r.stop = 1; pthread_cond_broadcast(&r.cv); pthread_mutex_unlock(&r.mu);
    for (int k = 0; k < n_th; k++) pthread_join(th[k], NULL);
    pthread_mutex_destroy(&r.mu); pthread_cond_destroy(&r.cv);
    if (pin) text_buf_put(pin, pin_cap + 1);
    stream_put(0, up);
    pthread_mutex_lock(&s->g_mu);
    if (!ok && !s->u_cancel) s->u_err = 1;
    if (!ok && s->up_done < (size_t)s->size) s->u_err = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    return NULL;
}
#undef UP_SEG
/* the stager waits until the file's first `hi` bytes are on the device; 0 when the uploader failed */
static int wait_uploaded(source_t *s, size_t hi) {
    pthread_mutex_lock(&s->g_mu);
    while (s->up_done < hi && !s->u_err) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const int ok = s->up_done >= hi;
    pthread_mutex_unlock(&s->g_mu);
    return ok;
}

static void *bgzf_gpu_stager(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    const size_t nb = s->g_nb;
    char *t = (char *)s->d_tab;
    int ok = 1;
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * 4 * GPU_STRETCH);
    unsigned char *tmp = (unsigned char *)malloc(65536);
    ok = ok && st && tmp;
    /* (the host's table, HPGV_BGZF_HOST_TABLE=1)  A short first stretch, which the header reader and the pipeline wait for
     * (a wave per block: 1.5 ms), then stretches that double up to 131 072 blocks, decoding side by side, up to GPU_INFLIGHT
     * launches on streams of their own.  Published in file order. */
    enum { GPU_INFLIGHT = 4, GPU_FIRST = 4096 };
    const int inflight = GPU_INFLIGHT;
    void *cs[GPU_INFLIGHT] = { s->cstream, NULL, NULL, NULL };
    for (int q = 1; ok && q < inflight; q++) ok = stream_get(0, &cs[q]) == HPGV_OK;
    size_t q_hi[GPU_INFLIGHT];                           /* the stretches in flight end at these blocks; the oldest starts at g_done */
    int qh = 0, qn = 0;
    const int dbg = (g_env.run_trace != 0); const double T0 = now_s();
    size_t up_hi = 0;                                    /* compressed bytes [0, up_hi) are on the device */
    for (size_t first = 0; ok && (first < nb || qn > 0);) {
        if (first < nb && qn < inflight) {               /* upload the next stretch while the earlier ones decode */
            /* a short first stretch, which the header reader waits for, then 32 768 blocks doubling up to four times that */
            const size_t stretch = first == 0 ? (nb <= 3 * (size_t)GPU_FIRST ? nb : GPU_FIRST)      /* a small file: one launch */
                                 : first < GPU_FIRST + (size_t)GPU_STRETCH ? GPU_STRETCH
                                 : first < GPU_FIRST + 3 * (size_t)GPU_STRETCH ? 2 * (size_t)GPU_STRETCH : 4 * (size_t)GPU_STRETCH;
            const size_t next = first + stretch < nb ? first + stretch : nb;
            const size_t lo = (size_t)s->g_in_off[first], hi = (size_t)s->g_in_off[next - 1] + s->g_in_len[next - 1] + 8;   /* + the last block's trailer: its CRC-32 is checked */
            (void)lo;
            if (hi > up_hi) {                             /* the uploader has been at it since the file was opened */
                ok = wait_uploaded(s, hi);
                up_hi = hi;
                if (dbg) fprintf(stderr, "stager: [%zu,%zu) is up, to byte %.1f MB, at %.4f\n", first, next, hi / 1e6, now_s() - T0);
            }
            const int q = (qh + qn) % inflight;
            ok = ok && hpgv_inflate_blocks_dev(CTX, (const uint8_t *)s->d_comp, (const uint64_t *)t + first, (const uint32_t *)(t + nb * 16) + first,
                                               (const uint64_t *)(t + nb * 8) + first, (const uint32_t *)(t + nb * 20) + first, (int)(next - first),
                                               (uint8_t *)s->d_text, (int32_t *)s->d_status + first, cs[q]) == HPGV_OK;
            if (bgzf_verify_on())                         /* the blocks' CRC-32, on the device behind the decoder */
                ok = ok && hpgv_bgzf_verify_tiles_dev(CTX, (const uint8_t *)s->d_comp, (const uint64_t *)t + first, (const uint32_t *)(t + nb * 16) + first,
                                                      (const uint64_t *)(t + nb * 8) + first, (const uint32_t *)(t + nb * 20) + first, (int)(next - first),
                                                      (const uint8_t *)s->d_text, (int32_t *)s->d_status + first, s->d_tiles, (uint64_t)s->n_tiles, cs[q]) == HPGV_OK;
            q_hi[q] = next; qn++;
            if (ok && first == 0 && next < nb) {
                /* launches that run side by side finish together, so the first stretch decodes alone (the time of one
                 * block, ~40 ms) while the uploader carries on; the next stretches are launched as soon as it is done */
                first = next;
            } else {
                first = next;
                if (first < nb && qn < inflight) continue;
            }
        }
        if (ok && qn > 0) {                              /* the oldest stretch in flight: wait, check, publish */
            const size_t a = s->g_done, n = q_hi[qh] - a;
            ok = hpgv_memcpy_d2h(CTX, st, (char *)s->d_status + a * 4, n * 4, cs[qh]) == HPGV_OK;      /* synchronises that stream */
            const size_t refuse_every = (size_t)g_env.test_refuse_every;     /* tests: exercise the host patch path */
            for (size_t k = 0; ok && k < n; k++)
                if (st[k] || (refuse_every && (a + k) % refuse_every == 0)) {                             /* not taken by the device decoder: the host decodes it, the text is patched */
                    const size_t bb = a + k;
                    ok = !inflate_block(s->map + s->g_in_off[bb], s->g_in_len[bb], tmp, s->g_out_len[bb])
                      && hpgv_memcpy_h2d(CTX, (char *)s->d_text + s->g_out_off[bb], tmp, s->g_out_len[bb], cs[qh]) == HPGV_OK;
                }
            if (dbg) fprintf(stderr, "stager: decoded up to %zu at %.4f\n", q_hi[qh], now_s() - T0);
            if (ok) {
                pthread_mutex_lock(&s->g_mu);
                s->g_done = q_hi[qh];
                s->dev_ready = q_hi[qh] == nb ? s->dev_len : (size_t)s->g_out_off[q_hi[qh]];
                pthread_cond_broadcast(&s->g_cv);
                pthread_mutex_unlock(&s->g_mu);
            }
            qh = (qh + 1) % inflight; qn--;
        }
    }
    for (int q = 0; q < GPU_INFLIGHT; q++) if (cs[q]) (void)hpgv_stream_sync(CTX, cs[q]);      /* after a failure launches may still be running */
    for (int q = 1; q < GPU_INFLIGHT; q++) stream_put(0, cs[q]);
    pthread_mutex_lock(&s->g_mu);
    if (!ok) s->g_err = 1;
    s->g_finished = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    free(st); free(tmp);
    if (s->u_started) {                                  /* the compressed bytes are freed below: the uploader must be through */
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (dbg) fprintf(stderr, "stager: finished %.4f\n", now_s() - T0);
    /* only the text is needed from here on */
    if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    if (s->d_status) { (void)hpgv_dev_free(CTX, s->d_status); s->d_status = NULL; }
    if (dbg) fprintf(stderr, "stager: freed %.4f\n", now_s() - T0);
    return NULL;
}

/* The BGZF header walk is a chain of dependent cache misses (122 000 blocks: 25 ms), so a team walks the file in
 * segments: every segment but the first finds a place where three valid block headers follow one another, walks from
 * there to the first block at or past its end, and the pieces are accepted only if every walk ends exactly where
 * the next one began -- then their concatenation IS the chain from offset 0.  Anything else: the serial walk.  The
 * team reads the file with pread (some 60 bytes per block), so that no page of the mapping is touched. */
typedef struct {
    int fd; size_t size, seg; int nseg;
    size_t *start, *end, *cnt;                           /* per segment: first block, where the walk stopped, blocks */
    uint64_t **in_off; uint32_t **in_len, **out_len; int *bad;
} bgzf_walk_t;
static size_t pread_full(int fd, void *buf, size_t n, size_t pos) {
    size_t have = 0;
    while (have < n) {
        const ssize_t got = pread(fd, (char *)buf + have, n - have, (off_t)(pos + have));
        if (got <= 0) break;
        have += (size_t)got;
    }
    return have;
}
/* bgzf_block without the trailer: the first `have` bytes of a block that has `avail` bytes of file left */
static int bgzf_header(const unsigned char *p, size_t have, size_t avail, size_t *bsize, size_t *cdata_off) {
    if (have < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8), end = 12 + xlen;
    size_t off = 12, bs = 0;
    if (end > have) return 0;                            /* more extra fields than the walk reads: the serial walk takes the file */
    while (off + 4 <= end) {
        const size_t slen = (size_t)p[off + 2] | ((size_t)p[off + 3] << 8);
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2 && off + 6 <= end) bs = ((size_t)p[off + 4] | ((size_t)p[off + 5] << 8)) + 1;
        off += 4 + slen;
    }
    if (bs < end + 8 || bs > avail) return 0;
    *bsize = bs; *cdata_off = end;
    return 1;
}
static void bgzf_walk_task(void *v, int k) {
    enum { WINDOW = 4 * 65536 + 256, HEAD = 60 };
    bgzf_walk_t *w = (bgzf_walk_t *)v;
    const size_t lim = k + 1 == w->nseg ? w->size : (size_t)(k + 1) * w->seg;
    size_t pos = (size_t)k * w->seg, bs, co, is;
    w->bad[k] = 1; w->cnt[k] = 0;
    if (k > 0) {                                         /* the first place in the segment where three valid blocks follow one another */
        unsigned char *win = (unsigned char *)malloc(WINDOW);
        if (!win) return;
        const size_t base = pos, wn = pread_full(w->fd, win, WINDOW, base), slim = lim - base < wn ? lim - base : wn;
        int found = 0;
        size_t o = 0;
        while (o < slim) {
            const unsigned char *c = (const unsigned char *)memchr(win + o, 31, slim - o);
            if (!c) break;
            o = (size_t)(c - win);
            size_t q = o;
            int chain = 0;
            while (chain < 3 && q < wn && bgzf_block(win + q, wn - q, &bs, &co, &is) && is <= 65536) { q += bs; chain++; }
            if (chain == 3 || (chain > 0 && base + q == w->size)) { found = 1; break; }
            o++;
        }
        free(win);
        if (!found) {
            if (slim < lim - base) return;               /* a segment longer than the window with no block start in the window: not BGZF as we know it */
            w->start[k] = w->end[k] = lim; w->bad[k] = 0; return;      /* no block begins in this segment */
        }
        pos = base + o;
    }
    w->start[k] = pos;
    size_t cap = w->seg / 8192 + 1024, n = 0;
    uint64_t *a = (uint64_t *)malloc(cap * 8);
    uint32_t *b = (uint32_t *)malloc(cap * 4), *c = (uint32_t *)malloc(cap * 4);
    int ok = a && b && c;
    unsigned char hb[4 + HEAD];
    size_t hn = ok && pos < lim ? pread_full(w->fd, hb + 4, HEAD, pos) : 0;      /* hb + 4: this block's first bytes */
    while (ok && pos < lim) {
        if (!bgzf_header(hb + 4, hn, w->size - pos, &bs, &co)) { ok = 0; break; }
        /* one read gets this block's last four bytes (ISIZE) and the next block's first ones */
        const size_t got = pread_full(w->fd, hb, 4 + HEAD, pos + bs - 4);
        if (got < 4) { ok = 0; break; }
        is = (size_t)hb[0] | ((size_t)hb[1] << 8) | ((size_t)hb[2] << 16) | ((size_t)hb[3] << 24);
        hn = got - 4;
        if (is > 65536) { ok = 0; break; }
        if (n == cap) {
            cap *= 2;
            uint64_t *a2 = (uint64_t *)realloc(a, cap * 8);
            uint32_t *b2 = (uint32_t *)realloc(b, cap * 4), *c2 = (uint32_t *)realloc(c, cap * 4);
            if (a2) a = a2;
            if (b2) b = b2;
            if (c2) c = c2;
            if (!a2 || !b2 || !c2) { ok = 0; break; }
        }
        a[n] = pos + co; b[n] = (uint32_t)(bs - co - 8); c[n] = (uint32_t)is;
        n++; pos += bs;
    }
    w->in_off[k] = a; w->in_len[k] = b; w->out_len[k] = c; w->cnt[k] = n; w->end[k] = pos; w->bad[k] = !ok;
}
/* 0 = the tables are filled (malloc'ed, *nb_out blocks, *text_out bytes of text); non-zero = walk serially */
static int bgzf_walk_parallel(int fd, size_t size, uint64_t **in_off, uint64_t **out_off, uint32_t **in_len, uint32_t **out_len,
                              size_t *nb_out, size_t *text_out) {
    enum { NSEG = 64 };
    if (size < (size_t)NSEG * 4096) return 1;
    bgzf_walk_t w;
    size_t start[NSEG], end[NSEG], cnt[NSEG];
    uint64_t *a[NSEG]; uint32_t *b[NSEG], *c[NSEG]; int bad[NSEG];
    memset(a, 0, sizeof a); memset(b, 0, sizeof b); memset(c, 0, sizeof c);
    w.fd = fd; w.size = size; w.seg = size / NSEG; w.nseg = NSEG;
    w.start = start; w.end = end; w.cnt = cnt; w.in_off = a; w.in_len = b; w.out_len = c; w.bad = bad;
    io_pool_t tp;
    pool_init(&tp, default_io_threads());
    pool_run(&tp, bgzf_walk_task, &w, NSEG);
    pool_destroy(&tp);
    int ok = 1;
    size_t nb = 0, expect = 0;
    for (int k = 0; k < NSEG && ok; k++) {
        if (bad[k]) ok = 0;
        else if (cnt[k] == 0) { if (start[k] != end[k]) ok = 0; }          /* an empty segment: the chain passes over it */
        else { if (start[k] != expect) ok = 0; expect = end[k]; nb += cnt[k]; }
    }
    if (ok && expect != size) ok = 0;
    uint64_t *io = NULL, *oo = NULL; uint32_t *il = NULL, *ol = NULL;
    if (ok) {
        io = (uint64_t *)malloc((nb + 1) * 8); oo = (uint64_t *)malloc((nb + 1) * 8);
        il = (uint32_t *)malloc((nb + 1) * 4); ol = (uint32_t *)malloc((nb + 1) * 4);
        ok = io && oo && il && ol;
    }
    if (ok) {
        size_t n = 0, text = 0;
        for (int k = 0; k < NSEG; k++) {
            if (cnt[k] == 0) continue;
            memcpy(io + n, a[k], cnt[k] * 8); memcpy(il + n, b[k], cnt[k] * 4); memcpy(ol + n, c[k], cnt[k] * 4);
            n += cnt[k];
        }
        for (size_t i = 0; i < nb; i++) { oo[i] = text; text += ol[i]; }
        *in_off = io; *out_off = oo; *in_len = il; *out_len = ol; *nb_out = nb; *text_out = text;
    } else { free(io); free(oo); free(il); free(ol); }
    for (int k = 0; k < NSEG; k++) { free(a[k]); free(b[k]); free(c[k]); }
    return !ok;
}

/* ---- the streaming form of the device path: the block table comes from the device too ----------------------------------
 * Walking the 490 000 block headers of a 4.6 GB file on the host takes 0.08 - 0.26 s of dependent reads (beside the
 * uploader's own reads of the same file) before the first block can be decoded.  Here the stager asks the device for the
 * blocks in what has been uploaded so far (hpgv_bgzf_scan_dev finds the headers in the compressed bytes and checks that
 * they form a chain), decodes them, and goes on where the chain stands: the blocks of the first 8 MB are decoded a few
 * milliseconds after the file was opened, then stretches of 16 384 and 32 768 blocks as their bytes arrive.  The text's size is not known in advance, so the text lies in a range of device
 * addresses that is backed as the table grows (dev_text_grow); the reader learns the text's end when the stager has seen
 * the file's last block.  A stretch of the file whose headers are not the ones bgzip writes is walked on the host (through
 * the mapping).  HPGV_BGZF_HOST_TABLE=1 keeps the table on the host (the form above). */
enum { SCAN_ROWS_MAX = 131072, SCAN_SLOTS = 4 };
#define SCAN_RANGE_MAX ((size_t)2 << 30)
typedef struct {
    uint64_t *d_in_off, *d_out_off; uint32_t *d_in_len, *d_out_len; int32_t *d_status;       /* device rows of this stretch */
    uint64_t *h_in_off, *h_out_off; uint32_t *h_in_len, *h_out_len;                          /* and their host copy (for blocks the device refuses) */
    size_t n, text_end;
    void *stream;
} scan_slot_t;
typedef struct {
    scan_slot_t slot[SCAN_SLOTS];
    void *d_scratch; size_t scratch_bytes;
    size_t chain_pos, text_pos, blocks;                   /* the chain stands at this file offset; text bytes and blocks before it */
    size_t first_n;                                       /* rows already in slot 0 (found when the path was chosen) */
    size_t rows_cap;                                      /* tests (HPGV_TEST_SCAN_ROWS): no stretch longer than this */
    size_t host_rows;                                     /* blocks the next walk on the host may take (doubles while the device finds no chain) */
} scan_state_t;

/* the uploader's progress: waits until at least `want` bytes are up (or all of the file); returns how many are, 0 on failure */
static size_t wait_uploaded_some(source_t *s, size_t want) {
    if (want > (size_t)s->size) want = (size_t)s->size;
    pthread_mutex_lock(&s->g_mu);
    while (s->up_done < want && !s->u_err) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const size_t have = s->u_err ? 0 : s->up_done;
    pthread_mutex_unlock(&s->g_mu);
    return have;
}
/* rows of the blocks from file offset `pos` on, walked through the mapping (headers the device scan does not know);
 * returns the number of rows (0: not a block), *end = where the walk stands */
static size_t bgzf_host_rows(source_t *s, size_t pos, size_t hi, size_t max_rows, size_t text, scan_slot_t *q, size_t *end, size_t *text_end) {
    size_t n = 0;
    while (n < max_rows && pos < hi) {
        size_t bs, co, is;
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536 || pos + bs > hi) break;
        q->h_in_off[n] = pos + co; q->h_in_len[n] = (uint32_t)(bs - co - 8); q->h_out_off[n] = text; q->h_out_len[n] = (uint32_t)is;
        text += is; pos += bs; n++;
    }
    *end = pos; *text_end = text;
    return n;
}
static int scan_slot_rows_to_host(scan_slot_t *q) {
    return hpgv_memcpy_d2h(CTX, q->h_in_off, q->d_in_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_out_off, q->d_out_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_in_len, q->d_in_len, q->n * 4, q->stream) == HPGV_OK
        && hpgv_memcpy_d2h(CTX, q->h_out_len, q->d_out_len, q->n * 4, q->stream) == HPGV_OK;
}
static int scan_slot_rows_to_device(scan_slot_t *q) {
    return hpgv_memcpy_h2d(CTX, q->d_in_off, q->h_in_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_out_off, q->h_out_off, q->n * 8, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_in_len, q->h_in_len, q->n * 4, q->stream) == HPGV_OK
        && hpgv_memcpy_h2d(CTX, q->d_out_len, q->h_out_len, q->n * 4, q->stream) == HPGV_OK;
}
/* the next stretch's rows into slot q: up to max_rows blocks from the chain's position among the bytes that are up.
 * 1 = q->n rows (0 rows: the file has ended), 0 = failure, 2 = (only with wait = 0) the bytes for that many blocks are
 * not up yet */
static int scan_next_rows(source_t *s, scan_state_t *S, scan_slot_t *q, size_t max_rows, int wait, int dbg, double T0) {
    q->n = 0; q->text_end = S->text_pos;
    const size_t avg = S->blocks ? S->chain_pos / S->blocks + 1 : 16384;
    size_t want = S->chain_pos + max_rows * avg;                      /* bytes that should hold that many blocks */
    /* the file's first stretch is whatever the first two segments hold (1 400 blocks of level 6): the header reader and
     * the pipeline wait for it, and a launch of 1 400 blocks takes as long as one of 4 096 (the time of one block) */
    if (S->blocks == 0 && want > ((size_t)8 << 20)) want = (size_t)8 << 20;
    for (;;) {
        if (S->chain_pos >= (size_t)s->size) return 1;
        if (!wait) {
            const size_t now_up = wait_uploaded_some(s, 0);
            if (now_up < (want < (size_t)s->size ? want : (size_t)s->size)) return 2;
        }
        const size_t have = wait_uploaded_some(s, want);
        if (!have) return 0;
        size_t hi = have;
        if (hi - S->chain_pos > SCAN_RANGE_MAX) hi = S->chain_pos + SCAN_RANGE_MAX;
        uint64_t res[4] = { 0, 0, 0, 0 };
        if (hpgv_bgzf_scan_dev(CTX, (const uint8_t *)s->d_comp, S->chain_pos, hi, S->text_pos, (int)max_rows, q->d_in_off, q->d_in_len,
                               q->d_out_off, q->d_out_len, S->d_scratch, S->scratch_bytes, res, q->stream) != HPGV_OK) return 0;
        if (res[0] > 0) {
            q->n = (size_t)res[0]; q->text_end = (size_t)res[2];
            if (!scan_slot_rows_to_host(q)) return 0;
            S->chain_pos = (size_t)res[1]; S->text_pos = q->text_end; S->blocks += q->n; S->host_rows = 64;
            if (dbg) fprintf(stderr, "stager: %zu blocks found up to byte %.1f MB (%.1f MB are up) at %.4f\n", q->n, S->chain_pos / 1e6, have / 1e6, now_s() - T0);
            return 1;
        }
        /* no block at the chain's position among the bytes that are up: it is not all there yet, or its header is not bgzip's */
        if (hi < (size_t)s->size && hi - S->chain_pos < ((size_t)1 << 17)) { want = hi + ((size_t)1 << 20); continue; }
        size_t end = 0, tend = 0;
        if (S->host_rows < 64) S->host_rows = 64;
        q->n = bgzf_host_rows(s, S->chain_pos, hi, S->host_rows < max_rows ? S->host_rows : max_rows, S->text_pos, q, &end, &tend);
        S->host_rows *= 2;                                            /* a few blocks, then the device again; more if it still finds none */
        if (q->n == 0) {
            if (hi < (size_t)s->size) { want = hi + ((size_t)1 << 20); continue; }      /* a block that ends beyond what is up */
            return 0;                                                 /* bytes that are no block */
        }
        if (dbg) fprintf(stderr, "stager: %zu blocks walked on the host from byte %zu at %.4f\n", q->n, S->chain_pos, now_s() - T0);
        q->text_end = tend;
        if (!scan_slot_rows_to_device(q)) return 0;
        S->chain_pos = end; S->text_pos = tend; S->blocks += q->n;
        return 1;
    }
}

/* the stretches go through a ring of SCAN_SLOTS slots: the stager finds and launches them as their bytes arrive, the
 * publisher waits for them in file order, patches what the device refused and hands the text to the reader -- neither
 * waits for the other's event */
typedef struct {
    source_t *s; scan_state_t *S;
    pthread_mutex_t mu; pthread_cond_t cv;
    size_t n_launched, n_published;                       /* stretches; slot k % SCAN_SLOTS holds stretch k */
    int launch_done, bad;
    double T0; int dbg;
} scan_ring_t;
static void *bgzf_publisher(void *v) {
    scan_ring_t *R = (scan_ring_t *)v;
    source_t *s = R->s;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    unsigned char *tmp = (unsigned char *)malloc(65536);
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * SCAN_ROWS_MAX);
    int ok = tmp && st;
    const size_t refuse_every = (size_t)g_env.test_refuse_every;       /* tests: exercise the host patch path */
    size_t done_blocks = 0;
    for (size_t k = 0; ok; k++) {
        pthread_mutex_lock(&R->mu);
        while (R->n_launched <= k && !R->launch_done && !R->bad) pthread_cond_wait(&R->cv, &R->mu);
        const int have = R->n_launched > k && !R->bad;
        pthread_mutex_unlock(&R->mu);
        if (!have) break;
        scan_slot_t *q = &R->S->slot[k % SCAN_SLOTS];
        ok = hpgv_memcpy_d2h(CTX, st, q->d_status, q->n * 4, q->stream) == HPGV_OK;           /* synchronises that stream */
        for (size_t i = 0; ok && i < q->n; i++)
            if (st[i] || (refuse_every && (done_blocks + i) % refuse_every == 0)) {             /* not taken by the device decoder: the host decodes it, the text is patched */
                ok = !inflate_block(s->map + q->h_in_off[i], q->h_in_len[i], tmp, q->h_out_len[i])
                  && hpgv_memcpy_h2d(CTX, (char *)s->d_text + q->h_out_off[i], tmp, q->h_out_len[i], q->stream) == HPGV_OK;
            }
        done_blocks += q->n;
        if (R->dbg) fprintf(stderr, "stager: decoded up to block %zu at %.4f\n", done_blocks, now_s() - R->T0);
        if (ok) {
            pthread_mutex_lock(&s->g_mu);
            s->g_done = done_blocks;
            s->dev_ready = q->text_end;
            pthread_cond_broadcast(&s->g_cv);
            pthread_mutex_unlock(&s->g_mu);
        }
        pthread_mutex_lock(&R->mu);
        R->n_published = k + 1;
        if (!ok) R->bad = 1;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->mu);
    }
    if (!ok) { pthread_mutex_lock(&R->mu); R->bad = 1; pthread_cond_broadcast(&R->cv); pthread_mutex_unlock(&R->mu); }
    free(tmp); free(st);
    return NULL;
}
static void *bgzf_gpu_stream_stager(void *v) {
    source_t *s = (source_t *)v;
    (void)SRC_CTX(s);                                               /* this thread works on the part's member device from here on */
    scan_state_t *S = (scan_state_t *)s->blk;                        /* (handed over in the field the host path uses for its block list) */
    s->blk = NULL;
    scan_ring_t R;
    memset(&R, 0, sizeof R);
    R.s = s; R.S = S; R.dbg = (g_env.run_trace != 0); R.T0 = now_s();
    const int dbg = R.dbg; const double T0 = R.T0;
    pthread_mutex_init(&R.mu, NULL); pthread_cond_init(&R.cv, NULL);
    pthread_t pub;
    int ok = pthread_create(&pub, NULL, bgzf_publisher, &R) == 0;
    const int have_pub = ok;
    size_t launched = 0;
    for (size_t k = 0; ok; k++) {
        pthread_mutex_lock(&R.mu);                                   /* a free slot */
        while (k >= R.n_published + SCAN_SLOTS && !R.bad) pthread_cond_wait(&R.cv, &R.mu);
        ok = !R.bad;
        pthread_mutex_unlock(&R.mu);
        if (!ok) break;
        scan_slot_t *q = &S->slot[k % SCAN_SLOTS];
        /* a short first stretch, which the header reader and the pipeline wait for, then 16 384 and 32 768 blocks at a time:
         * a stretch is launched when its bytes are up, and the decoder (23 GB/s of bgzip's level-6 bytes) is not much slower than
         * the bus -- behind stretches that double (up to 131 072 blocks) the device waited for the next one's bytes */
        size_t rows = launched == 0 ? 4096 : launched < 4096 + 16384 ? 16384 : 32768;
        if (S->rows_cap && rows > S->rows_cap) rows = S->rows_cap;
        if (k == 0 && S->first_n) q->n = S->first_n;                 /* found when the path was chosen */
        else ok = scan_next_rows(s, S, q, rows, 1, dbg, T0) == 1;
        if (!ok || q->n == 0) break;                                 /* (no rows: the file has ended) */
        if (k == 1) {                                                /* launches side by side finish together: the first stretch, which the */
            pthread_mutex_lock(&R.mu);                               /* header reader waits for, decodes alone (1.5 ms instead of 6) */
            while (R.n_published < 1 && !R.bad) pthread_cond_wait(&R.cv, &R.mu);
            ok = !R.bad;
            pthread_mutex_unlock(&R.mu);
            if (!ok) break;
        }
        ok = dev_text_grow(s->d_text, q->text_end + 16 + (q->text_end >> 4), &s->d_text_cap)      /* some room ahead: growing waits for the kernels that run */
          || dev_text_grow(s->d_text, q->text_end + 16, &s->d_text_cap);
        ok = ok && hpgv_inflate_blocks_dev(CTX, (const uint8_t *)s->d_comp, q->d_in_off, q->d_in_len, q->d_out_off, q->d_out_len,
                                           (int)q->n, (uint8_t *)s->d_text, q->d_status, q->stream) == HPGV_OK;
        if (bgzf_verify_on())                                        /* the blocks' CRC-32, on the device behind the decoder */
            ok = ok && hpgv_bgzf_verify_tiles_dev(CTX, (const uint8_t *)s->d_comp, q->d_in_off, q->d_in_len, q->d_out_off, q->d_out_len,
                                                  (int)q->n, (const uint8_t *)s->d_text, q->d_status, s->d_tiles, (uint64_t)s->n_tiles, q->stream) == HPGV_OK;
        if (!ok) break;
        if (dbg && k == 0) fprintf(stderr, "stager: first stretch launched at %.4f\n", now_s() - T0);
        launched += q->n;
        pthread_mutex_lock(&R.mu);
        R.n_launched = k + 1;
        pthread_cond_broadcast(&R.cv);
        pthread_mutex_unlock(&R.mu);
        if (S->chain_pos >= (size_t)s->size) break;
    }
    pthread_mutex_lock(&R.mu);
    R.launch_done = 1;
    if (!ok) R.bad = 1;
    pthread_cond_broadcast(&R.cv);
    pthread_mutex_unlock(&R.mu);
    if (have_pub) pthread_join(pub, NULL);
    ok = ok && !R.bad;
    pthread_mutex_destroy(&R.mu); pthread_cond_destroy(&R.cv);
    for (int q = 0; q < SCAN_SLOTS; q++) if (S->slot[q].stream) (void)hpgv_stream_sync(CTX, S->slot[q].stream);      /* after a failure launches may still be running */
    pthread_mutex_lock(&s->g_mu);
    if (!ok) s->g_err = 1;
    else { s->dev_len = S->text_pos; s->dev_len_known = 1; s->g_nb = S->blocks; }
    s->g_finished = 1;
    pthread_cond_broadcast(&s->g_cv);
    pthread_mutex_unlock(&s->g_mu);
    if (s->u_started) {                                  /* the compressed bytes are freed below: the uploader must be through */
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (dbg) fprintf(stderr, "stager: finished (%zu blocks, %.1f MB of text) at %.4f\n", S->blocks, S->text_pos / 1e6, now_s() - T0);
    for (int q = 1; q < SCAN_SLOTS; q++) stream_put(s->c_low, S->slot[q].stream);
    for (int q = 0; q < SCAN_SLOTS; q++) free(S->slot[q].h_in_off);
    free(S);
    if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }      /* only the text is needed from here on */
    if (s->d_scan) { (void)hpgv_dev_free(CTX, s->d_scan); s->d_scan = NULL; }
    return NULL;
}

/* 0 = the streaming stager has the file; 1 = not taken (the caller goes on with the host's table; nothing is left behind) */
static int bgzf_stream_stage(source_t *s) {
    const int dbg = (g_env.run_trace != 0); const double T0 = now_s();
    scan_state_t *S = (scan_state_t *)calloc(1, sizeof *S);
    if (!S) return 1;
    /* the decoder's streams have the lowest priority: the batches' kernels go first whenever a compute unit has room */
    const int low = !g_env.no_low_priority;
    s->c_low = low;
    int ok = stream_get(0, &s->rstream) == HPGV_OK && stream_get(low, &s->cstream) == HPGV_OK;
    const size_t slot_bytes = (size_t)SCAN_ROWS_MAX * 28;
    S->scratch_bytes = hpgv_bgzf_scan_scratch_bytes(SCAN_RANGE_MAX + 16, SCAN_ROWS_MAX);
    if (ok) ok = hpgv_dev_alloc(CTX, slot_bytes * SCAN_SLOTS + S->scratch_bytes + 256, &s->d_scan) == HPGV_OK;
    for (int k = 0; ok && k < SCAN_SLOTS; k++) {
        scan_slot_t *q = &S->slot[k];
        char *d = (char *)s->d_scan + (size_t)k * slot_bytes;
        q->d_in_off = (uint64_t *)d; q->d_out_off = (uint64_t *)(d + (size_t)SCAN_ROWS_MAX * 8);
        q->d_in_len = (uint32_t *)(d + (size_t)SCAN_ROWS_MAX * 16); q->d_out_len = (uint32_t *)(d + (size_t)SCAN_ROWS_MAX * 20);
        q->d_status = (int32_t *)(d + (size_t)SCAN_ROWS_MAX * 24);
        char *h = (char *)malloc((size_t)SCAN_ROWS_MAX * 24);
        ok = h != NULL;
        q->h_in_off = (uint64_t *)h; q->h_out_off = (uint64_t *)(h + (size_t)SCAN_ROWS_MAX * 8);
        q->h_in_len = (uint32_t *)(h + (size_t)SCAN_ROWS_MAX * 16); q->h_out_len = (uint32_t *)(h + (size_t)SCAN_ROWS_MAX * 20);
        if (k == 0) q->stream = s->cstream; else ok = ok && stream_get(low, &q->stream) == HPGV_OK;
    }
    if (ok) S->d_scratch = (char *)s->d_scan + slot_bytes * SCAN_SLOTS;
    if (dbg) fprintf(stderr, "stage: streams and tables at %.4f\n", now_s() - T0);
    /* the first blocks, from the file's first megabytes: is this a file the device can chain, and how much text is it? */
    S->rows_cap = g_env.test_scan_rows > 0 ? (size_t)g_env.test_scan_rows : 0;
    if (ok) ok = scan_next_rows(s, S, &S->slot[0], S->rows_cap && S->rows_cap < 4096 ? S->rows_cap : 4096, 1, dbg, T0) == 1 && S->slot[0].n > 0;
    size_t est = 0;
    if (ok) {
        /* committed at once, with room (growing later waits for the kernels that are running) */
        est = (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size * 1.10) + ((size_t)128 << 20);
        if (S->chain_pos >= (size_t)s->size) est = S->text_pos + 16;
        if (est > ((size_t)48 << 30)) ok = 0;                        /* as with the host's table: such a text stays on the host path, */
        if (S->chain_pos >= (size_t)s->size && S->blocks < 256 && !s->is_part && !s->mp) ok = 0;      /* and a small file is as quick there */
    }
    const long tp = g_env.test_text_estimate_percent;              /* tests: a text that outgrows what was committed for it */
    if (ok && tp > 0 && S->chain_pos < (size_t)s->size) {
        est = (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size) / 100 * (size_t)tp;
        if (est < S->text_pos + 16) est = S->text_pos + 16;
        dev_text_drop_cached();
    }
    if (ok) {
        s->text_est = S->chain_pos >= (size_t)s->size ? S->text_pos : (size_t)((double)S->text_pos / (double)S->chain_pos * (double)s->size);
        s->d_text = dev_text_get(est, &s->d_text_cap, &s->d_text_kind);
        ok = s->d_text != NULL && ((s->d_text_kind == DEV_TEXT_GROWS && !g_env.no_growing_text) || S->chain_pos >= (size_t)s->size);
        if (!ok && s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
        if (ok) tiles_alloc(s, est);
    }
    if (dbg) fprintf(stderr, "stage: first %zu blocks found, text estimate %.1f MB, %s at %.4f\n", S->slot[0].n, est / 1e6, ok ? "streaming" : "not taken", now_s() - T0);
    if (ok) {
        S->first_n = S->slot[0].n;
        s->dev_len = 0; s->dev_len_known = 0; s->dev_pos = 0; s->dev_ready = 0; s->g_done = 0; s->g_err = 0; s->g_finished = 0; s->g_nb = 0;
        s->blk = (size_t *)S;
        ok = pthread_create(&s->g_thread, NULL, bgzf_gpu_stream_stager, s) == 0;
        if (ok) s->g_started = 1; else s->blk = NULL;
    }
    if (!ok) {
        for (int k = 1; k < SCAN_SLOTS; k++) stream_put(low, S->slot[k].stream);
        for (int k = 0; k < SCAN_SLOTS; k++) free(S->slot[k].h_in_off);
        free(S);
        if (s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
        tiles_free(s);
        if (s->d_scan) { (void)hpgv_dev_free(CTX, s->d_scan); s->d_scan = NULL; }
        stream_put(0, s->rstream); s->rstream = NULL;
        stream_put(low, s->cstream); s->cstream = NULL; s->c_low = 0;
        return 1;
    }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

/* ---- a bgzip file on SEVERAL devices (a group context: HPGV_DEVICES, hpgv_host_init_devices; --num-threads / the devices
 * option of shared_options.c:60-61 has no such notion: the reference reads the file with one thread).  BGZF blocks are
 * independent, so the file is cut at block starts into one contiguous PART per member: every part goes up ITS device's own
 * link (the upload is what a run of the device path waits for), is decoded and kept there, and its windows are tokenized and
 * scanned there.  Each part is staged by the streaming stager exactly as a whole file is (a source_t whose map / size /
 * file_off describe the part).  The reader walks the parts in file order; a part's text ends inside a line as a rule: the
 * line's head (the end of part k) and tail (the start of part k + 1) come back to the host, and the joined line goes through
 * the ordinary host-text entry as a batch of one line between the two parts' windows (read_lines_dev). ---- */
static int bgzf_stream_stage(source_t *s);
static void *bgzf_uploader(void *v);
/* one part (or, when the parts are given up, nothing): 0 = the streaming stager has it */
static int part_stream_stage(source_t *s) {
    const ctx_saved_t saved = SRC_CTX(s);
    int taken = 0;
    s->gpu_tried = 1;
    pthread_mutex_init(&s->g_mu, NULL); pthread_cond_init(&s->g_cv, NULL);
    s->g_sync = 1; s->up_done = 0; s->u_cancel = 0; s->u_err = 0;
    if (hpgv_dev_alloc(CTX, (size_t)s->size + 16, &s->d_comp) == HPGV_OK) {
        if (pthread_create(&s->u_thread, NULL, bgzf_uploader, s) == 0) s->u_started = 1;
        else { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    }
    if (s->u_started && bgzf_stream_stage(s) == 0) taken = 1;
    if (!taken) {                                                    /* nothing is left behind */
        if (s->u_started) {
            pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
            pthread_join(s->u_thread, NULL); s->u_started = 0;
        }
        if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
        pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0;
    }
    ctx_back(saved);
    return taken ? 0 : 1;
}
/* the first block start at or after `from` from which four blocks chain (a header's magic inside compressed data does not) */
static size_t bgzf_find_block_start(const source_t *s, size_t from) {
    const size_t size = (size_t)s->size, lim = from + ((size_t)1 << 20) < size ? from + ((size_t)1 << 20) : size;
    for (size_t pos = from; pos + 28 <= lim; pos++) {
        if (s->map[pos] != 31 || s->map[pos + 1] != 139) continue;
        size_t q = pos;
        int n = 0;
        while (n < 4 && q < size) {
            size_t bs, co, is;
            if (!bgzf_block(s->map + q, size - q, &bs, &co, &is) || is > 65536) break;
            q += bs; n++;
        }
        if (n == 4 || (n > 0 && q == size)) return pos;
    }
    return 0;
}
/* 0 = the file is staged in parts (s is part 0); 1 = not taken, s is as it was */
static int bgzf_parts_stage(source_t *s) {
    const int G = g_ctx ? hpgv_group_size(g_ctx) : 1;
    if (G < 2 || g_env.bgzf_one_device || g_env.no_device_windows || g_env.bgzf_host_table || g_env.serial_bgzf_walk || g_env.no_growing_text) return 1;
    const size_t part_min = g_env.bgzf_part_min_kb > 0 ? (size_t)g_env.bgzf_part_min_kb << 10 : (size_t)64 << 20;      /* (tests: parts of small files) */
    int n = G < MEMBERS_MAX ? G : MEMBERS_MAX;
    if ((size_t)s->size / part_min < (size_t)n) n = (int)((size_t)s->size / part_min);
    if (n < 2) return 1;
    size_t b[MEMBERS_MAX + 1];
    b[0] = 0; b[n] = (size_t)s->size;
    for (int k = 1; k < n; k++) {
        b[k] = bgzf_find_block_start(s, (size_t)s->size / (size_t)n * (size_t)k);
        if (b[k] == 0 || b[k] <= b[k - 1]) return 1;
    }
    src_parts_t *mp = (src_parts_t *)calloc(1, sizeof *mp);
    if (!mp) return 1;
    mp->n = n; mp->p[0] = s; mp->whole_size = s->size;
    const int dbg = (g_env.run_trace != 0);
    int ok = 1;
    for (int k = 1; ok && k < n; k++) {                              /* the later parts first: if one of them is not taken, part 0 is still the whole file */
        source_t *p = (source_t *)calloc(1, sizeof *p);
        if (!p) { ok = 0; break; }
        p->kind = SRC_BGZF; p->fd = dup(s->fd); p->is_part = 1;
        p->map_base = s->map_base; p->map_len = s->map_len;
        p->map = s->map + b[k]; p->size = (off_t)(b[k + 1] - b[k]); p->file_off = (off_t)b[k];
        p->ctx = hpgv_group_member(g_ctx, k); p->member = k;
        mp->p[k] = p;
        ok = p->fd >= 0 && p->ctx && part_stream_stage(p) == 0;
        if (ok) p->map_pos = (size_t)p->size;
    }
    if (ok) {
        s->ctx = hpgv_group_member(g_ctx, 0); s->member = 0; s->mp = mp; s->size = (off_t)b[1];
        ok = part_stream_stage(s) == 0;
        if (!ok) { s->ctx = NULL; s->mp = NULL; s->size = mp->whole_size; s->gpu_tried = 0; }
    }
    if (!ok) {
        for (int k = 1; k < n; k++) if (mp->p[k]) { source_close(mp->p[k]); free(mp->p[k]); }
        free(mp);
        if (dbg) fprintf(stderr, "stage: the file is not taken in parts\n");
        return 1;
    }
    if (dbg) { fprintf(stderr, "stage: %d parts, one per device, cut at bytes", n); for (int k = 1; k < n; k++) fprintf(stderr, " %zu", b[k]); fprintf(stderr, "\n"); }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

int bgzf_gpu_stage(source_t *s) {
    s->gpu_tried = 1;
    if (g_env.no_gpu_inflate || !g_ctx || s->map_pos != 0) return 1;
    if (!s->is_part && !s->mp && (size_t)s->size >= ((size_t)64 << 10) && bgzf_parts_stage(s) == 0) return 0;
    const int dbg = (g_env.run_trace != 0); double T0 = now_s();
    size_t nb = 0, cap = 1 << 16, text = 0;
    uint64_t *in_off = NULL, *out_off = NULL;
    uint32_t *in_len = NULL, *out_len = NULL;
    int ok = 1;
    if ((size_t)s->size < ((size_t)64 << 10)) return 1;             /* a tiny file is as quick on the host (the block count decides below) */
    /* the compressed bytes start going up NOW, beside the walk of the block headers (0.08 s for the 490 000 blocks of a
     * 4.6 GB file): the decoder needs the table, the bus does not */
    pthread_mutex_init(&s->g_mu, NULL); pthread_cond_init(&s->g_cv, NULL);
    s->g_sync = 1; s->up_done = 0; s->u_cancel = 0; s->u_err = 0;
    if (hpgv_dev_alloc(CTX, (size_t)s->size + 16, &s->d_comp) == HPGV_OK) {
        if (dbg) fprintf(stderr, "stage: room for the compressed file at %.4f\n", now_s() - T0);
        if (pthread_create(&s->u_thread, NULL, bgzf_uploader, s) == 0) s->u_started = 1;
        else { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
    }
    if (!s->u_started) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0; return 1; }
    if (!g_env.bgzf_host_table && !g_env.serial_bgzf_walk && bgzf_stream_stage(s) == 0) return 0;
    if (g_env.serial_bgzf_walk || bgzf_walk_parallel(s->fd, (size_t)s->size, &in_off, &out_off, &in_len, &out_len, &nb, &text)) {
        nb = 0; text = 0;
        if (dbg) fprintf(stderr, "stage: serial walk\n");
        in_off = (uint64_t *)malloc(cap * 8); out_off = (uint64_t *)malloc(cap * 8);
        in_len = (uint32_t *)malloc(cap * 4); out_len = (uint32_t *)malloc(cap * 4);
        ok = in_off && out_off && in_len && out_len;
        size_t pos = 0;
        while (ok && pos < (size_t)s->size) {
            size_t bs, co, is;
            if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536) { ok = 0; break; }
            if (nb == cap) {
                cap *= 2;
                uint64_t *a = (uint64_t *)realloc(in_off, cap * 8), *b = (uint64_t *)realloc(out_off, cap * 8);
                uint32_t *c = (uint32_t *)realloc(in_len, cap * 4), *d = (uint32_t *)realloc(out_len, cap * 4);
                if (a) in_off = a;
                if (b) out_off = b;
                if (c) in_len = c;
                if (d) out_len = d;
                if (!a || !b || !c || !d) { ok = 0; break; }
            }
            in_off[nb] = pos + co; in_len[nb] = (uint32_t)(bs - co - 8); out_off[nb] = text; out_len[nb] = (uint32_t)is;
            text += is; pos += bs; nb++;
        }
    }
    if (dbg) fprintf(stderr, "stage: walk %.4f\n", now_s() - T0);
    if (ok && (nb < 256 || nb > 0x7FFFFFFFu || text > ((size_t)48 << 30))) ok = 0;     /* a small file is as quick on the host */
    if (ok) { s->c_low = 0; ok = stream_get(0, &s->rstream) == HPGV_OK && stream_get(0, &s->cstream) == HPGV_OK; }
    if (ok) ok = hpgv_dev_alloc(CTX, nb * 24 + 64, &s->d_tab) == HPGV_OK;
    if (ok) { s->text_est = text; s->d_text = dev_text_get(text + 16, &s->d_text_cap, &s->d_text_kind); ok = s->d_text != NULL; }
    if (ok) tiles_alloc(s, text + 16);
    if (ok) ok = hpgv_dev_alloc(CTX, nb * 4 + 16, &s->d_status) == HPGV_OK;
    if (dbg) fprintf(stderr, "stage: alloc %.4f\n", now_s() - T0);
    if (ok) {
        char *t = (char *)s->d_tab;
        ok = hpgv_memcpy_h2d(CTX, t, in_off, nb * 8, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 8, out_off, nb * 8, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 16, in_len, nb * 4, s->cstream) == HPGV_OK
          && hpgv_memcpy_h2d(CTX, t + nb * 20, out_len, nb * 4, s->cstream) == HPGV_OK;
    }
    if (dbg) fprintf(stderr, "stage: tab %.4f\n", now_s() - T0);
    if (ok) {
        s->g_in_off = in_off; s->g_out_off = out_off; s->g_in_len = in_len; s->g_out_len = out_len; s->g_nb = nb;
        s->dev_len = text; s->dev_len_known = 1; s->dev_pos = 0; s->dev_ready = 0; s->g_done = 0; s->g_err = 0; s->g_finished = 0;
        ok = pthread_create(&s->g_thread, NULL, bgzf_gpu_stager, s) == 0;
        if (ok) s->g_started = 1;
    }
    if (!ok) {
        if (s->u_started) {                              /* not a file for the device path after all: call the uploader back */
            pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
            pthread_join(s->u_thread, NULL); s->u_started = 0;
        }
        free(in_off); free(out_off); free(in_len); free(out_len);
        s->g_in_off = s->g_out_off = NULL; s->g_in_len = s->g_out_len = NULL;
        if (s->g_sync) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); s->g_sync = 0; }
            if (s->d_comp) { (void)hpgv_dev_free(CTX, s->d_comp); s->d_comp = NULL; }
        if (s->d_tab) { (void)hpgv_dev_free(CTX, s->d_tab); s->d_tab = NULL; }
        if (s->d_status) { (void)hpgv_dev_free(CTX, s->d_status); s->d_status = NULL; }
        if (s->d_text) { dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind); s->d_text = NULL; }
        tiles_free(s);
        stream_put(0, s->rstream); s->rstream = NULL;
        stream_put(0, s->cstream); s->cstream = NULL;
        return 1;
    }
    s->map_pos = (size_t)s->size;                                    /* the CPU path has nothing left to do */
    return 0;
}

/* appends up to cap bytes of (decompressed) data to buf; 0 = end of data, (size_t)-1 = error.  RAW and GZIP
 * fill the room unless the data ends; BGZF hands out the whole blocks that fit (inflated in parallel, straight
 * into buf), or -- when not even the next block fits -- the part of it that does, so callers loop. */
