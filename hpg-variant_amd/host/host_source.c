/* host_source.c -- the file runners' inputs: the PED table, byte sources (plain file, gzip, bgzip).
 * Part of libhpgv_host.so (see hpgv_host_internal.h for the map of its units). */
#include "hpgv_host_internal.h"

/* ------------------------------------------------------------------------ */
/* file-level runners: VCF + PED in, sorted TSV out                           */
/* ------------------------------------------------------------------------ */


void ped_table_free(ped_table_t *p) {
    free(p->fid); free(p->iid); free(p->pat); free(p->mat); free(p->phe); free(p->sex); free(p->pheno); free(p->blob);
    memset(p, 0, sizeof *p);
}

static char *next_ws_token(char **p) {
    char *s = *p;
    while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') s++;
    if (!*s) return NULL;
    char *e = s;
    while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++;
    if (*e) { *e = 0; e++; }
    *p = e;
    return s;
}

int ped_table_read(const char *path, ped_table_t *ped) {
    memset(ped, 0, sizeof *ped);
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot open PED file %s", path); return HPGV_ERR_INVALID; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    ped->blob = (char *)malloc((size_t)sz + 1);
    if (!ped->blob || fread(ped->blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); ped_table_free(ped); return HPGV_ERR_NOMEM; }
    fclose(f);
    ped->blob[sz] = 0;
    int cap = 0;
    for (long i = 0; i < sz; i++) if (ped->blob[i] == '\n') cap++;
    cap += 2;
    ped->fid = (char **)malloc(sizeof(char *) * (size_t)cap); ped->iid = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->pat = (char **)malloc(sizeof(char *) * (size_t)cap); ped->mat = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->phe = (char **)malloc(sizeof(char *) * (size_t)cap);
    ped->sex = (int *)malloc(sizeof(int) * (size_t)cap); ped->pheno = (int *)malloc(sizeof(int) * (size_t)cap);
    if (!ped->fid || !ped->iid || !ped->pat || !ped->mat || !ped->phe || !ped->sex || !ped->pheno) {
        ped_table_free(ped);
        snprintf(g_err, sizeof g_err, "out of memory reading the PED file");
        return HPGV_ERR_NOMEM;
    }
    char *line = ped->blob;
    while (line && *line) {
        char *eol = strchr(line, '\n');
        if (eol) *eol = 0;
        if (*line && *line != '#') {
            char *p = line;
            char *a = next_ws_token(&p), *b = next_ws_token(&p), *c = next_ws_token(&p), *d = next_ws_token(&p);
            char *e = next_ws_token(&p), *g = next_ws_token(&p);
            if (!(a && b && c && d && e && g)) {         /* a row without FID IID PAT MAT SEX PHENO is damaged input, not a sample to skip */
                snprintf(g_err, sizeof g_err, "PED row %d has fewer than 6 columns", ped->n + 1);
                ped_table_free(ped);
                return HPGV_ERR_INVALID;
            }
            if (ped->n < cap) {
                ped->fid[ped->n] = a; ped->iid[ped->n] = b; ped->pat[ped->n] = c; ped->mat[ped->n] = d; ped->phe[ped->n] = g;
                /* SEX and PHENO are read as numbers (the reference keeps the phenotype as a float, individual->variable):
                 * "2" and "2.0" are the same label; anything that is not 1 / 2 -- 0, -9, text -- is unknown / other */
                char *end;
                const double sx = strtod(e, &end);
                const int sx_ok = end != e && *end == 0;
                ped->sex[ped->n] = (sx_ok && sx == 1.0) ? HPGV_SEX_MALE : (sx_ok && sx == 2.0) ? HPGV_SEX_FEMALE : HPGV_SEX_UNKNOWN;
                const double ph = strtod(g, &end);
                const int ph_ok = end != g && *end == 0;
                ped->pheno[ped->n] = (ph_ok && ph == 2.0) ? HPGV_COND_AFFECTED : (ph_ok && ph == 1.0) ? HPGV_COND_UNAFFECTED : HPGV_COND_OTHER;
                ped->n++;
            }
        }
        line = eol ? eol + 1 : NULL;
    }
    return HPGV_OK;
}

/* ---- byte sources: plain file (pread by a small thread team), BGZF (blocks inflated in parallel),
 *      generic gzip (one zlib stream) -- shared_options.c:60-61 `--compression gzip|bgzip` ------------ */
hpgv_run_filters_t g_filters = { -1.0, -1.0, -1, -1, -1.0 };     /* hpgv_run_set_filters; negative = off */
double g_run_times[6];                           /* last run: read, engine, write, sort, total seconds, batches */
double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

int bgzf_block(const unsigned char *p, size_t avail, size_t *bsize, size_t *cdata_off, size_t *isize) {
    if (avail < 18 || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
    size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8), off = 12, end = 12 + xlen;
    if (end > avail) return 0;
    size_t bs = 0;
    while (off + 4 <= end) {
        size_t slen = (size_t)p[off + 2] | ((size_t)p[off + 3] << 8);
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2 && off + 6 <= end) bs = ((size_t)p[off + 4] | ((size_t)p[off + 5] << 8)) + 1;
        off += 4 + slen;
    }
    if (bs < end + 8 || bs > avail) return 0;
    *bsize = bs; *cdata_off = end;
    *isize = (size_t)p[bs - 4] | ((size_t)p[bs - 3] << 8) | ((size_t)p[bs - 2] << 16) | ((size_t)p[bs - 1] << 24);
    return 1;
}

int source_open(source_t *s, const char *path) {
    memset(s, 0, sizeof *s);
    s->fd = open(path, O_RDONLY);
    struct stat st;
    if (s->fd < 0 || fstat(s->fd, &st) != 0) { if (s->fd >= 0) close(s->fd); return 1; }
    s->size = st.st_size;
    unsigned char head[18];
    ssize_t got = pread(s->fd, head, sizeof head, 0);
    if (got >= 2 && head[0] == 31 && head[1] == 139) {
        size_t bs, co, is;
        s->map = (const unsigned char *)mmap(NULL, (size_t)s->size, PROT_READ, MAP_PRIVATE, s->fd, 0);
        s->map_base = s->map; s->map_len = (size_t)s->size;
        if (s->map != MAP_FAILED && bgzf_block(s->map, (size_t)s->size, &bs, &co, &is)) { s->kind = SRC_BGZF; return 0; }
        if (s->map != MAP_FAILED) munmap((void *)s->map, (size_t)s->size);
        s->map = NULL; s->map_base = NULL; s->map_len = 0;
        s->gz = gzdopen(dup(s->fd), "rb");
        if (!s->gz) { close(s->fd); return 1; }
        gzbuffer(s->gz, 1u << 20);
        s->kind = SRC_GZIP;
    }
    return 0;
}

void source_close(source_t *s) {
    if (s->mp) {                                                     /* the other parts of the file (this one is part 0) */
        for (int k = 1; k < s->mp->n; k++) if (s->mp->p[k]) { source_close(s->mp->p[k]); free(s->mp->p[k]); }
        s->size = s->mp->whole_size;
        free(s->mp->seam); free(s->mp); s->mp = NULL;
    }
    const ctx_saved_t saved = SRC_CTX(s);
    if (s->g_started) pthread_join(s->g_thread, NULL);               /* the stager reads the mapping: it goes first */
    if (s->u_started) {
        pthread_mutex_lock(&s->g_mu); s->u_cancel = 1; pthread_mutex_unlock(&s->g_mu);
        pthread_join(s->u_thread, NULL); s->u_started = 0;
    }
    if (s->kind == SRC_BGZF && s->map_base && !s->is_part) munmap((void *)s->map_base, s->map_len);
    free(s->pend); free(s->blk);
    if (s->g_sync) { pthread_mutex_destroy(&s->g_mu); pthread_cond_destroy(&s->g_cv); }
    free(s->g_in_off); free(s->g_out_off); free(s->g_in_len); free(s->g_out_len);
    if (g_ctx) {
        if (s->d_comp) (void)hpgv_dev_free(CTX, s->d_comp);
        if (s->d_tab) (void)hpgv_dev_free(CTX, s->d_tab);
        if (s->d_status) (void)hpgv_dev_free(CTX, s->d_status);
        if (s->d_text) dev_text_put(s->d_text, s->d_text_cap, s->d_text_kind);
        if (s->d_tiles) dev_tiles_put(s->d_tiles, s->d_tiles_cap);
        if (s->d_scan) (void)hpgv_dev_free(CTX, s->d_scan);
        stream_put(0, s->rstream);
        stream_put(s->c_low, s->cstream);
    }
    if (s->kind == SRC_GZIP && s->gz) gzclose(s->gz);
    if (s->fd >= 0) close(s->fd);
    ctx_back(saved);
}
