/* host_reader.c -- whole-line batches out of a byte source (host text or windows of device text), the VCF header.
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

static size_t source_read(source_t *s, char *buf, size_t cap) {
    if (s->kind == SRC_RAW) {
        if (s->pos >= s->size) return 0;
        size_t want = (size_t)(s->size - s->pos) < cap ? (size_t)(s->size - s->pos) : cap;
        s->job_buf = buf; s->job_want = want; s->job_bad = 0;
        pool_run(s->pool, source_task_pread, s, (int)((want + PREAD_SEG - 1) / PREAD_SEG));
        const int bad = s->job_bad;
        if (bad) return (size_t)-1;
        s->pos += (off_t)want;
        return want;
    }
    if (s->kind == SRC_GZIP) {
        size_t n = 0;
        while (n < cap) {
            unsigned chunk = (cap - n) > (1u << 30) ? (1u << 30) : (unsigned)(cap - n);
            int got = gzread(s->gz, buf + n, chunk);
            if (got < 0) return (size_t)-1;
            if (got == 0) break;
            n += (size_t)got;
        }
        return n;
    }
    /* BGZF */
    if (cap == 0) return 0;
    if (!s->gpu_tried) (void)bgzf_gpu_stage(s);
    if (s->d_text) {                                                 /* the text is on the device: copy the next stretch out */
        pthread_mutex_lock(&s->g_mu);                                /* until the stager has decoded that far (or knows where the text ends) */
        for (;;) {
            const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
            const size_t want = limit - s->dev_pos < cap ? limit : s->dev_pos + cap;
            if (s->g_err || s->dev_ready >= want || s->g_finished) break;
            pthread_cond_wait(&s->g_cv, &s->g_mu);
        }
        const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
        const size_t end = limit - s->dev_pos < cap ? limit : s->dev_pos + cap;
        const int bad = s->g_err || s->dev_ready < end;
        pthread_mutex_unlock(&s->g_mu);
        if (bad) return (size_t)-1;
        const size_t n = end - s->dev_pos;
        if (n == 0) return 0;
        if (hpgv_memcpy_d2h(CTX, buf, (const char *)s->d_text + s->dev_pos, n, s->rstream) != HPGV_OK) return (size_t)-1;
        s->dev_pos += n;
        return n;
    }
    if (s->pend_pos < s->pend_len) {                                  /* rest of the block set aside last time */
        size_t n = s->pend_len - s->pend_pos < cap ? s->pend_len - s->pend_pos : cap;
        memcpy(buf, s->pend + s->pend_pos, n);
        s->pend_pos += n;
        return n;
    }
    size_t total = 0;
    int nb = 0;
    if (!s->blk && !(s->blk = (size_t *)malloc(sizeof(size_t) * 4 * MAXB))) return (size_t)-1;
    size_t *const b_in = s->blk, *const b_clen = s->blk + MAXB, *const b_out = s->blk + 2 * MAXB, *const b_isize = s->blk + 3 * MAXB;
    size_t pos = s->map_pos;
    while (pos < (size_t)s->size && nb < MAXB) {
        size_t bs, co, is;
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is) || is > 65536) return (size_t)-1;
        if (total + is > cap) break;
        b_in[nb] = pos + co; b_clen[nb] = bs - co - 8; b_out[nb] = total; b_isize[nb] = is;
        total += is; pos += bs; nb++;
    }
    if (total == 0) {
        if (pos >= (size_t)s->size) { s->map_pos = pos; return 0; }   /* only empty blocks (the EOF marker) were left */
        size_t bs, co, is;                                            /* the next block is larger than the room */
        if (!bgzf_block(s->map + pos, (size_t)s->size - pos, &bs, &co, &is)) return (size_t)-1;
        if (!s->pend && !(s->pend = (unsigned char *)malloc(65536))) return (size_t)-1;
        if (inflate_block(s->map + pos + co, bs - co - 8, s->pend, is)) return (size_t)-1;
        s->map_pos = pos + bs;
        s->pend_len = is; s->pend_pos = cap;
        memcpy(buf, s->pend, cap);
        return cap;
    }
    s->job_buf = buf; s->job_want = (size_t)nb; s->job_bad = 0;
    pool_run(s->pool, source_task_inflate, s, (nb + INFLATE_GROUP - 1) / INFLATE_GROUP);
    const int bad = s->job_bad;
    if (bad) return (size_t)-1;
    s->map_pos = pos;
    return total;
}

/* batch reader on top of a source: whole lines, about batch_bytes per batch; the unfinished last line is
 * carried over to the next batch */

/* bgzip decoded on the device: the next batch is a window of the device text that ends with a line.  Only the stretch
 * around the window's end comes back to the host, to find that line end; the engine fills the batch's host buffer with
 * the line heads (hpgv_text_alias).  Returns the window's bytes (0 at the end, (size_t)-1 on error), its device address
 * in r->last_dev and the context of its device in r->last_ctx.  `last`: the text's end is the file's end (the last line may
 * lack its newline); otherwise -- a part that another part follows -- every window ends with a newline and what is left
 * behind the text's last newline stays unread (the seam line's head). */
static size_t read_window(line_reader_t *r, source_t *s, size_t cap, int last) {
    const ctx_saved_t saved = SRC_CTX(s);
    const size_t start = s->dev_pos;
    size_t ret = (size_t)-1;
    r->last_dev = (const char *)s->d_text + start;
    r->last_ctx = s->ctx;
    r->last_base = (const char *)s->d_text; r->last_tiles = s->d_tiles; r->last_n_tiles = s->n_tiles;
    pthread_mutex_lock(&s->g_mu);                                    /* until the stager has decoded that far (or knows where the text ends) */
    for (;;) {
        const size_t lim = s->dev_len_known ? s->dev_len : (size_t)-1;
        const size_t want = lim - start <= cap ? lim : start + cap;
        if (s->g_err || s->dev_ready >= want || s->g_finished) break;
        pthread_cond_wait(&s->g_cv, &s->g_mu);
    }
    const size_t limit = s->dev_len_known ? s->dev_len : (size_t)-1;
    size_t end = limit - start <= cap ? limit : start + cap;
    const int bad = s->g_err || s->dev_ready < end;
    pthread_mutex_unlock(&s->g_mu);
    if (bad) goto out;
    if (start >= limit) { ret = 0; goto out; }
    if (end < limit || !last) {                                      /* cut at the last newline before `end` */
        size_t look = 1u << 18;
        for (;;) {
            if (look > end - start) look = end - start;
            if (r->tailcap < look) { free(r->tailbuf); r->tailbuf = (char *)malloc(look); r->tailcap = r->tailbuf ? look : 0; }
            if (!r->tailbuf) goto out;
            if (hpgv_memcpy_d2h(CTX, r->tailbuf, (const char *)s->d_text + end - look, look, s->rstream) != HPGV_OK) goto out;
            const char *nl = (const char *)memrchr(r->tailbuf, '\n', look);
            if (nl) { end = end - look + (size_t)(nl - r->tailbuf) + 1; break; }
            if (look == end - start) {
                if (end == limit && !last) { ret = 0; goto out; }    /* the part's rest holds no line end: all of it is the seam line's head */
                goto out;                                            /* a line longer than the batch */
            }
            look *= 4;
        }
    }
    s->dev_pos = end;
    ret = end - start;
out:
    ctx_back(saved);
    return ret;
}
/* a part's text length, known when its stager has seen its last block; (size_t)-1 on failure */
static size_t part_text_len(source_t *s) {
    pthread_mutex_lock(&s->g_mu);
    while (!s->dev_len_known && !s->g_err && !s->g_finished) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const size_t n = (s->g_err || !s->dev_len_known) ? (size_t)-1 : s->dev_len;
    pthread_mutex_unlock(&s->g_mu);
    return n;
}
/* bytes [from, to) of a part's text to the host, once they are decoded */
static int part_text_fetch(source_t *s, size_t from, size_t to, char *dst) {
    if (to <= from) return 1;
    pthread_mutex_lock(&s->g_mu);
    while (s->dev_ready < to && !s->g_err && !s->g_finished) pthread_cond_wait(&s->g_cv, &s->g_mu);
    const int ok = !s->g_err && s->dev_ready >= to;
    pthread_mutex_unlock(&s->g_mu);
    if (!ok) return 0;
    const ctx_saved_t saved = SRC_CTX(s);
    const int rc = hpgv_memcpy_d2h(CTX, dst, (const char *)s->d_text + from, to - from, s->rstream);
    ctx_back(saved);
    return rc == HPGV_OK;
}
static int seam_room(src_parts_t *mp, size_t more) {
    if (mp->seam_len + more <= mp->seam_cap) return 1;
    const size_t cap = (mp->seam_len + more) * 2 + 4096;
    char *q = (char *)realloc(mp->seam, cap);
    if (!q) return 0;
    mp->seam = q; mp->seam_cap = cap;
    return 1;
}
/* the next batch of a file on the device: a window of its text (r->last_dev set: the engine tokenizes it in place and writes
 * the lines' heads into the batch's host buffer), or -- a file staged in parts -- the line that straddles two parts, copied
 * into buf as ordinary host text (r->last_dev NULL). */
size_t read_lines_dev(line_reader_t *r, char *buf, size_t bufcap, size_t cap) {
    src_parts_t *mp = r->src.mp;
    if (!mp) return read_window(r, &r->src, cap, 1);
    for (;;) {
        if (mp->seam_len) {                                          /* the seam line in front of part `cur` */
            if (mp->seam_len > bufcap) return (size_t)-1;
            memcpy(buf, mp->seam, mp->seam_len);
            const size_t n = mp->seam_len;
            mp->seam_len = 0;
            r->last_dev = NULL; r->last_ctx = NULL; r->last_tiles = NULL;
            return n;
        }
        source_t *s = mp->p[mp->cur];
        const int last = mp->cur == mp->n - 1;
        const size_t n = read_window(r, s, cap, last);
        if (n != 0 || last) return n;
        /* part `cur` is through: what lies behind its last newline is the head of a line that goes on in the next part(s) */
        const size_t len = part_text_len(s);
        if (len == (size_t)-1) return (size_t)-1;
        const size_t head = len - s->dev_pos;
        if (!seam_room(mp, head) || !part_text_fetch(s, s->dev_pos, len, mp->seam)) return (size_t)-1;
        mp->seam_len = head;
        s->dev_pos = len;
        while (++mp->cur < mp->n) {
            source_t *t = mp->p[mp->cur];
            if (head == 0 && mp->seam_len == 0) break;               /* the part ended with a newline: the next one starts a line */
            /* the tail: the next part's text up to its first newline, looked for in growing stretches */
            size_t have = 0, look = 1u << 16, nl_at = (size_t)-1;
            for (;;) {
                size_t tlen = (size_t)-1;
                pthread_mutex_lock(&t->g_mu);
                while (t->dev_ready < have + look && !t->dev_len_known && !t->g_err && !t->g_finished) pthread_cond_wait(&t->g_cv, &t->g_mu);
                if (t->dev_len_known) tlen = t->dev_len;
                const int bad = t->g_err || (!t->dev_len_known && t->dev_ready < have + look);
                pthread_mutex_unlock(&t->g_mu);
                if (bad) return (size_t)-1;
                size_t to = have + look;
                if (tlen != (size_t)-1 && to > tlen) to = tlen;
                if (!seam_room(mp, to - have) || !part_text_fetch(t, have, to, mp->seam + mp->seam_len)) return (size_t)-1;
                const char *nl = (const char *)memchr(mp->seam + mp->seam_len, '\n', to - have);
                if (nl) { nl_at = have + (size_t)(nl - (mp->seam + mp->seam_len)); mp->seam_len += nl_at - have + 1; break; }
                mp->seam_len += to - have;
                have = to;
                if (tlen != (size_t)-1 && have >= tlen) break;       /* the whole part is one piece of this line: on to the next part */
                look *= 4;
            }
            if (nl_at != (size_t)-1) { t->dev_pos = nl_at + 1; break; }
            t->dev_pos = have;
        }
        if (mp->cur >= mp->n) mp->cur = mp->n - 1;                   /* the line ran to the file's end: hand it out, then the last part reports the end */
    }
}

/* fills buf (capacity cap) with whole lines; returns the byte count, 0 at the end, (size_t)-1 when a
 * single line does not fit or the source fails */
size_t read_lines(line_reader_t *r, char *buf, size_t cap) {
    size_t n = 0;
    /* the carry is the stretch of the source right before its read position: with the text on the device the batch is
     * the device bytes from (position - carry) on */
    if (!r->src.gpu_tried && r->src.kind == SRC_BGZF) (void)bgzf_gpu_stage(&r->src);
    r->last_dev = r->src.d_text ? (const char *)r->src.d_text + (r->src.dev_pos - r->carry_len) : NULL;
    r->last_base = (const char *)r->src.d_text; r->last_tiles = r->src.d_text ? r->src.d_tiles : NULL; r->last_n_tiles = r->src.n_tiles;
    if (r->carry_len) {                                 /* may exceed cap: what followed the header in its read buffer */
        n = r->carry_len < cap ? r->carry_len : cap;
        memcpy(buf, r->carry, n);
        r->carry_len -= n;
        if (r->carry_len) {
            memmove(r->carry, r->carry + n, r->carry_len);
            size_t end = n;
            while (end > 0 && buf[end - 1] != '\n') end--;
            if (end == 0) return (size_t)-1;
            const size_t tail = n - end;                /* put the cut line back in front of the rest */
            if (r->carry_len + tail > r->carry_cap) return (size_t)-1;     /* cannot happen: n bytes were just taken out */
            memmove(r->carry + tail, r->carry, r->carry_len);
            memcpy(r->carry, buf + end, tail);
            r->carry_len += tail;
            return end;
        }
    }
    while (!r->eof && n < cap) {
        size_t got = source_read(&r->src, buf + n, cap - n);
        if (got == (size_t)-1) return (size_t)-1;
        if (got == 0) { r->eof = 1; break; }
        n += got;
    }
    if (n == 0) return 0;
    if (r->eof) return n;                               /* last batch: may end without a newline */
    size_t end = n;
    while (end > 0 && buf[end - 1] != '\n') end--;
    if (end == 0) return (size_t)-1;
    size_t tail = n - end;
    if (tail > r->carry_cap) { free(r->carry); r->carry = (char *)malloc(tail); r->carry_cap = r->carry ? tail : 0; }
    if (tail && !r->carry) return (size_t)-1;
    memcpy(r->carry, buf + end, tail);
    r->carry_len = tail;
    return end;
}

/* VCF header: skips the '##' lines and takes the sample names from the '#CHROM' line; what follows that line
 * in the bytes already read becomes the reader's carry.  Returns the number of samples, -1 without a
 * '#CHROM' line; *hdr_out owns the text the names point into. */
int vcf_header_read(line_reader_t *rd, char **hdr_out, char ***names_out, size_t *chrom_off) {
    char *hdr = NULL;
    int n_samples = -1;
    char **names = NULL;
    size_t cap = 4u << 20, have = 0;
    hdr = (char *)malloc(cap + 1);
    int done = 0;
    while (hdr && !done) {
        const int starved = cap == have;             /* header longer than the buffer: grow first */
        size_t got = starved ? 0 : source_read(&rd->src, hdr + have, cap - have);
        if (got == (size_t)-1) break;
        have += got;
        hdr[have] = 0;
        char *p = hdr, *chrom = NULL;
        size_t data_start = 0;
        while (*p == '#') {
            char *eol = (char *)memchr(p, '\n', have - (size_t)(p - hdr));
            if (!eol) { p = NULL; break; }
            if (!strncmp(p, "#CHROM", 6)) { chrom = p; *eol = 0; data_start = (size_t)(eol + 1 - hdr); break; }
            p = eol + 1;
        }
        if (chrom) {
            if (chrom_off) *chrom_off = (size_t)(chrom - hdr);
            size_t len = strlen(chrom);
            while (len > 0 && chrom[len - 1] == '\r') chrom[--len] = 0;
            int tabs = 0;
            for (size_t i = 0; i < len; i++) if (chrom[i] == '\t') tabs++;
            n_samples = tabs >= 9 ? tabs - 8 : 0;
            names = (char **)malloc(sizeof(char *) * (size_t)(n_samples + 1));
            int k = 0, col = 0;
            for (char *q = chrom; names && *q; q++)
                if (*q == '\t') { *q = 0; col++; if (col >= 9 && k < n_samples) names[k++] = q + 1; }
            size_t rest = have - data_start;
            if (rest) {
                rd->carry = (char *)malloc(rest);
                rd->carry_cap = rd->carry ? rest : 0;
                if (rd->carry) { memcpy(rd->carry, hdr + data_start, rest); rd->carry_len = rest; } else n_samples = -1;
            }
            if (!names) n_samples = -1;
            done = 1;
        } else if (p != NULL || (!starved && got == 0)) {
            break;                                   /* a data line came first, or the file ended */
        } else if (starved) {
            cap *= 2;
            char *nh = (char *)realloc(hdr, cap + 1);
            if (!nh) break;
            hdr = nh;
        }
    }
    *hdr_out = hdr; *names_out = names;
    return n_samples;
}
