/*
 * mprime_oracle.c — CPU restatement of the O(N) loops of multiPrime-core_V20.py ("V20")
 * behind the C ABI of include/mprime.h.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (multiprime_amd/) links, loads or calls
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the
 * checker.  It is deliberately written the way the reference is written — strings of
 * characters, one sequence at a time, set membership for Y_distance — so that it shares no
 * arithmetic with the bit-plane HIP kernels it checks.  Parity pin: tests/test_oracle_golden.py
 * compares every function here against traces recorded from the reference itself
 * (tests/golden/make_golden.py) on all fixtures.
 *
 * Each function cites the V20 lines it follows (V20 = /root/reference/scripts/multiPrime-core_V20.py).
 */
#include "../include/mprime.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct mp_ctx {
    char err[512];
    int32_t n_ranks;  /* 1 after mp_comm_init (the only world the checker knows) */
    int32_t n_rows, reserve_cols;
    char **rows;      /* mapped characters, NUL-terminated */
    int32_t *len;
    /* windows */
    int32_t p0, n_win, k, v;
    char *kmers;      /* [n_win][n_rows][k] characters; kmers[..][0]==0 marks "not stored" (exception) */
    uint64_t *words;  /* [n_win][3][n_rows]; 64-bit inside for every k, MP_WORD_BYTES(k) at the ABI (rd_word / wr_word) */
    int32_t n_ex, cap_ex;
    int32_t *ex_win, *ex_row;
    char *ex_kmer;    /* [n_ex][k] */
    /* extra rows */
    int32_t n_extra;
    int32_t *extra_win;
    uint64_t *extra_words;
    /* unique */
    int64_t n_ent;
    int64_t *win_off;
    uint64_t *u_b0, *u_b1, *u_g;
    int32_t *u_count, *u_first;
    int32_t *labels;  /* [n_win][n_rows] or NULL */
    /* staged candidates */
    int32_t n_cand;
    int32_t *cand_win;
    uint8_t *cand_codes;
    uint64_t sF, sR;
    double eval_ms;
    int32_t eval_n;
    /* "resident" masks of mp_eval_masks_resident: plain host arrays here */
    uint64_t *mask_f, *mask_r;
    int32_t n_masks;
    /* mp_seq_load: the checker keeps the characters; its "resident" scans are the byte scans on them */
    uint8_t *sq_bytes;
    int64_t *sq_off;
    int32_t sq_n;
};

static int fail(mp_ctx *c, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return code;
}

const char *mp_backend_name(void) { return "oracle"; }
const char *mp_last_error(const mp_ctx *c) { return c ? c->err : "null context"; }
int mp_set_stream(mp_ctx *c, void *s) { (void)c; (void)s; return MP_OK; }

int mp_create(int dev, mp_ctx **out) {
    (void)dev;
    if (!out) return MP_ERR_ARG;
    *out = (mp_ctx *)calloc(1, sizeof(mp_ctx));
    return *out ? MP_OK : MP_ERR_NOMEM;
}

static void free_windows(mp_ctx *c) {
    free(c->kmers); free(c->words); free(c->ex_win); free(c->ex_row); free(c->ex_kmer);
    free(c->extra_win); free(c->extra_words);
    free(c->win_off); free(c->u_b0); free(c->u_b1); free(c->u_g); free(c->u_count); free(c->u_first);
    free(c->labels);
    c->kmers = NULL; c->words = NULL; c->ex_win = c->ex_row = NULL; c->ex_kmer = NULL;
    c->extra_win = NULL; c->extra_words = NULL; c->win_off = NULL;
    c->u_b0 = c->u_b1 = c->u_g = NULL; c->u_count = c->u_first = NULL; c->labels = NULL;
    c->n_ex = c->cap_ex = c->n_extra = 0; c->n_ent = 0; c->n_win = 0;
}

static void free_rows(mp_ctx *c) {
    if (c->rows) for (int32_t r = 0; r < c->n_rows; r++) free(c->rows[r]);
    free(c->rows); free(c->len);
    c->rows = NULL; c->len = NULL; c->n_rows = 0;
}

void mp_destroy(mp_ctx *c) {
    if (!c) return;
    free_windows(c); free_rows(c);
    free(c->cand_win); free(c->cand_codes);
    free(c->mask_f); free(c->mask_r);
    free(c->sq_bytes); free(c->sq_off);
    free(c);
}

/* V20:453  sequence = re.sub("[^ACGTRYMKSWHBVD]", "-", i.strip().upper()) */
static char map_char(unsigned char ch) {
    int u = toupper(ch);
    return (u && strchr("ACGTRYMKSWHBVD", u)) ? (char)u : '-';
}

int mp_reserve_columns(mp_ctx *c, int32_t n_columns) {
    if (!c) return MP_ERR_ARG;
    if (n_columns < 0 || n_columns > 0x3fffffff) return fail(c, MP_ERR_ARG, "mp_reserve_columns: bad width %d", n_columns);
    c->reserve_cols = n_columns;
    return MP_OK;
}

int mp_load_msa(mp_ctx *c, const uint8_t *bytes, const int64_t *off, int32_t n_rows) {
    if (!c || !bytes || !off || n_rows <= 0) return c ? fail(c, MP_ERR_ARG, "mp_load_msa: bad arguments") : MP_ERR_ARG;
    free_windows(c); free_rows(c);
    c->rows = (char **)calloc((size_t)n_rows, sizeof(char *));
    c->len = (int32_t *)calloc((size_t)n_rows, sizeof(int32_t));
    if (!c->rows || !c->len) return fail(c, MP_ERR_NOMEM, "out of memory");
    c->n_rows = n_rows;
    for (int32_t r = 0; r < n_rows; r++) {
        int64_t n = off[r + 1] - off[r];
        if (n < 0 || n > 0x7fffffff) return fail(c, MP_ERR_ARG, "row %d has bad length", r);
        c->rows[r] = (char *)malloc((size_t)n + 1);
        if (!c->rows[r]) return fail(c, MP_ERR_NOMEM, "out of memory");
        for (int64_t i = 0; i < n; i++) c->rows[r][i] = map_char(bytes[off[r] + i]);
        c->rows[r][n] = 0;
        c->len[r] = (int32_t)n;
    }
    return MP_OK;
}

/* V20:625-627  start = len - len(lstrip("-")), stop = len(rstrip("-")) */
int mp_row_attributes(mp_ctx *c, int32_t *lead, int32_t *rstrip, int32_t *rowlen) {
    if (!c || !c->rows) return c ? fail(c, MP_ERR_ARG, "no alignment loaded") : MP_ERR_ARG;
    for (int32_t r = 0; r < c->n_rows; r++) {
        const char *s = c->rows[r];
        int32_t n = c->len[r], a = 0, b = n;
        while (a < n && s[a] == '-') a++;
        while (b > 0 && s[b - 1] == '-') b--;
        if (lead) lead[r] = a;
        if (rstrip) rstrip[r] = b;
        if (rowlen) rowlen[r] = n;
    }
    return MP_OK;
}

/* the same as histograms (np.quantile of seq_attribute takes one order statistic of each, V20:628-633) */
int mp_row_histograms(mp_ctx *c, int32_t n_bins, int64_t *lead_hist, int64_t *rstrip_hist) {
    if (!c || !c->rows) return c ? fail(c, MP_ERR_ARG, "no alignment loaded") : MP_ERR_ARG;
    if (n_bins <= 0 || !lead_hist || !rstrip_hist) return fail(c, MP_ERR_ARG, "mp_row_histograms: bad arguments");
    memset(lead_hist, 0, sizeof(int64_t) * (size_t)n_bins);
    memset(rstrip_hist, 0, sizeof(int64_t) * (size_t)n_bins);
    for (int32_t r = 0; r < c->n_rows; r++) {
        const char *s = c->rows[r];
        int32_t n = c->len[r], a = 0, b = n;
        while (a < n && s[a] == '-') a++;            /* len - len(lstrip("-")) */
        while (b > 0 && s[b - 1] == '-') b--;        /* len(rstrip("-")) */
        if (a >= n_bins || b >= n_bins) return fail(c, MP_ERR_CAPACITY, "mp_row_histograms: a row is longer than %d", n_bins - 1);
        lead_hist[a]++;
        rstrip_hist[b]++;
    }
    return MP_OK;
}

/* ungapped characters of s[a:b) into dst, returns their number (V20:674 / :680 .replace("-", "")) */
static int32_t ungapped(const char *s, int32_t a, int32_t b, char *dst) {
    int32_t n = 0;
    for (int32_t i = a; i < b; i++) if (s[i] != '-') dst[n++] = s[i];
    return n;
}

/* V20:666-687: the k-mer of one sequence at window position p, with edge-gap repair.
 * Returns the length of the result (== k unless the row is too short, V20:683-687). */
static int32_t window_kmer(const char *s, int32_t len, int32_t p, int32_t k, char *buf, char *scratch) {
    int32_t m = len - p;
    if (m < 0) m = 0;
    if (m > k) m = k;
    memcpy(buf, s + (p < len ? p : len), (size_t)m);   /* V20:666 slice (upper() is a no-op after :453) */
    int32_t n = m;
    int all_gap = (m == k);
    for (int32_t i = 0; i < m && all_gap; i++) if (buf[i] != '-') all_gap = 0;
    int32_t left_end = p < len ? p : len;              /* s[0:p] */
    if (!all_gap) {                                      /* V20:668-670 */
        if (n > 0 && buf[0] == '-') {                    /* V20:671 */
            int32_t run = 0;
            while (run < n && buf[run] == '-') run++;    /* :672-673 */
            int32_t nl = ungapped(s, 0, left_end, scratch);          /* :674 */
            if (nl >= run)                               /* :675 */
                memcpy(buf, scratch + nl - run, (size_t)run);        /* :676 left_seq[-run:] + sequence_narrow */
        }
        if (n > 0 && buf[n - 1] == '-') {                /* V20:677 */
            int32_t run = 0;
            while (run < n && buf[n - 1 - run] == '-') run++;        /* :678-679 */
            int32_t a = p + k < len ? p + k : len;
            int32_t nr = ungapped(s, a, len, scratch);   /* :680 */
            if (nr >= run) memcpy(buf + n - run, scratch, (size_t)run);   /* :681-682 */
        }
    }
    if (n < k) {                                         /* V20:683 */
        int32_t need = k - n;
        int32_t nl = ungapped(s, 0, left_end, scratch);  /* :685 */
        if (nl >= need) {                                /* :686-687 */
            memmove(buf + need, buf, (size_t)n);
            memcpy(buf, scratch + nl - need, (size_t)need);
            n = k;
        }
    }
    return n;
}

/* window words at the ABI: uint32 while k <= MP_NARROW_K, uint64 above (mprime.h); SKIP is the word's top bit either way */
static uint64_t rd_word(const void *p, size_t i, int32_t k) {
    if (k > MP_NARROW_K) return ((const uint64_t *)p)[i];
    uint32_t x = ((const uint32_t *)p)[i];
    return (uint64_t)(x & ~MP_WIN_SKIP) | ((x & MP_WIN_SKIP) ? MP_WIN_SKIP64 : 0);
}
static void wr_word(void *p, size_t i, int32_t k, uint64_t x) {
    if (k > MP_NARROW_K) ((uint64_t *)p)[i] = x;
    else ((uint32_t *)p)[i] = (uint32_t)(x & 0x7FFFFFFFu) | ((x & MP_WIN_SKIP64) ? MP_WIN_SKIP : 0u);
}

static int base_index(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; }

static uint8_t iupac_code(char ch) {
    switch (ch) {
    case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8;
    case 'R': return 1 | 4; case 'Y': return 2 | 8; case 'M': return 1 | 2; case 'K': return 4 | 8;
    case 'S': return 4 | 2; case 'W': return 1 | 8; case 'H': return 1 | 8 | 2; case 'B': return 4 | 8 | 2;
    case 'V': return 4 | 1 | 2; case 'D': return 4 | 1 | 8; case 'N': return 15;
    default: return 0;
    }
}

int mp_build_windows(mp_ctx *c, int32_t p0, int32_t n_win, int32_t k, int32_t v, int32_t *n_exc) {
    if (!c || !c->rows) return c ? fail(c, MP_ERR_ARG, "no alignment loaded") : MP_ERR_ARG;
    if (k < 2 || k > MP_MAX_K || n_win <= 0 || p0 < 0 || v < 0 || v >= k)
        return fail(c, MP_ERR_ARG, "bad window arguments (k=%d v=%d n_windows=%d)", k, v, n_win);
    int32_t N = c->n_rows, maxlen = 0;
    for (int32_t r = 0; r < N; r++) if (c->len[r] > maxlen) maxlen = c->len[r];
    if (p0 + n_win > (maxlen > c->reserve_cols ? maxlen : c->reserve_cols)) return fail(c, MP_ERR_ARG, "windows run past the longest row");
    free_windows(c);
    c->p0 = p0; c->n_win = n_win; c->k = k; c->v = v;
    c->kmers = (char *)calloc((size_t)n_win * N * k, 1);
    c->words = (uint64_t *)calloc((size_t)n_win * 3 * N, sizeof(uint64_t));
    char *scratch = (char *)malloc((size_t)maxlen + 1);
    char buf[2 * MP_MAX_K + 4];
    if (!c->kmers || !c->words || !scratch) { free(scratch); return fail(c, MP_ERR_NOMEM, "out of memory"); }
    for (int32_t w = 0; w < n_win; w++) {
        for (int32_t r = 0; r < N; r++) {
            int32_t n = window_kmer(c->rows[r], c->len[r], p0 + w, k, buf, scratch);
            if (n < k) { free(scratch); return fail(c, MP_ERR_SHORT_WINDOW, "row %d has fewer than %d residues at window %d", r, k, p0 + w); }
            uint64_t b0 = 0, b1 = 0, g = 0;
            int iupac = 0;
            for (int32_t j = 0; j < k; j++) {
                int bi = base_index(buf[j]);
                if (buf[j] == '-') g |= 1ull << j;
                else if (bi < 0) iupac = 1;
                else { b0 |= (uint64_t)(bi & 1) << j; b1 |= (uint64_t)(bi >> 1) << j; }
            }
            uint64_t *W = c->words + (size_t)w * 3 * N;
            if (iupac) {
                if (c->n_ex == c->cap_ex) {
                    c->cap_ex = c->cap_ex ? 2 * c->cap_ex : 1024;
                    c->ex_win = (int32_t *)realloc(c->ex_win, sizeof(int32_t) * c->cap_ex);
                    c->ex_row = (int32_t *)realloc(c->ex_row, sizeof(int32_t) * c->cap_ex);
                    c->ex_kmer = (char *)realloc(c->ex_kmer, (size_t)c->cap_ex * k);
                }
                c->ex_win[c->n_ex] = w; c->ex_row[c->n_ex] = r;
                memcpy(c->ex_kmer + (size_t)c->n_ex * k, buf, (size_t)k);
                c->n_ex++;
                W[r] = 0; W[N + r] = 0; W[2 * N + r] = MP_WIN_SKIP64 | ((1ull << k) - 1);
            } else {
                memcpy(c->kmers + ((size_t)w * N + r) * k, buf, (size_t)k);
                W[r] = b0; W[N + r] = b1; W[2 * N + r] = g;
            }
        }
    }
    free(scratch);
    if (n_exc) *n_exc = c->n_ex;
    return MP_OK;
}

int mp_get_exceptions(mp_ctx *c, int32_t cap, int32_t *ew, int32_t *er, uint8_t *codes) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    if (cap < c->n_ex) return fail(c, MP_ERR_CAPACITY, "exception buffer too small: need %d", c->n_ex);
    for (int32_t i = 0; i < c->n_ex; i++) {
        ew[i] = c->ex_win[i]; er[i] = c->ex_row[i];
        for (int32_t j = 0; j < c->k; j++) codes[(size_t)i * c->k + j] = iupac_code(c->ex_kmer[(size_t)i * c->k + j]);
    }
    return MP_OK;
}

int mp_set_extra_rows(mp_ctx *c, int32_t n, const int32_t *win, const void *words) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    free(c->extra_win); free(c->extra_words);
    c->extra_win = NULL; c->extra_words = NULL; c->n_extra = 0;
    if (n <= 0) return MP_OK;
    for (int32_t i = 0; i < n; i++) {
        if (win[i] < 0 || win[i] >= c->n_win || (i && win[i] < win[i - 1])) return fail(c, MP_ERR_ARG, "extra rows must be sorted by window");
    }
    c->extra_win = (int32_t *)malloc(sizeof(int32_t) * n);
    c->extra_words = (uint64_t *)malloc(sizeof(uint64_t) * 3 * n);
    memcpy(c->extra_win, win, sizeof(int32_t) * n);
    for (size_t i = 0; i < 3 * (size_t)n; i++) c->extra_words[i] = rd_word(words, i, c->k);
    c->n_extra = n;
    return MP_OK;
}

int mp_get_window_words(mp_ctx *c, int32_t w, int32_t row0, int32_t n, void *out) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    if (w < 0 || w >= c->n_win || row0 < 0 || n < 0 || row0 + n > c->n_rows) return fail(c, MP_ERR_ARG, "bad range");
    const uint64_t *W = c->words + (size_t)w * 3 * c->n_rows;
    for (int p = 0; p < 3; p++)
        for (int32_t i = 0; i < n; i++) wr_word(out, (size_t)p * n + i, c->k, W[(size_t)p * c->n_rows + row0 + i]);
    return MP_OK;
}

/* ---- per-window histogram: V20:689-711 (cover[i] += 1 / gap_sequence[sequence] += 1) -------- */
typedef struct { uint64_t b0, b1, g; int32_t row; } keyrow;

static int cmp_keyrow(const void *a, const void *b) {
    const keyrow *x = (const keyrow *)a, *y = (const keyrow *)b;
    if (x->g != y->g) return x->g < y->g ? -1 : 1;
    if (x->b1 != y->b1) return x->b1 < y->b1 ? -1 : 1;
    if (x->b0 != y->b0) return x->b0 < y->b0 ? -1 : 1;
    return x->row < y->row ? -1 : x->row > y->row;
}

typedef struct { uint64_t b0, b1, g; int32_t count, first; } uent;

static int cmp_first(const void *a, const void *b) {
    const uent *x = (const uent *)a, *y = (const uent *)b;
    return x->first < y->first ? -1 : x->first > y->first;
}

int mp_window_unique(mp_ctx *c, int64_t cap, int32_t want_labels, int64_t *n_entries) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    int32_t N = c->n_rows, W = c->n_win;
    free(c->win_off); free(c->u_b0); free(c->u_b1); free(c->u_g); free(c->u_count); free(c->u_first); free(c->labels);
    c->labels = NULL;
    keyrow *kr = (keyrow *)malloc(sizeof(keyrow) * (size_t)N);
    uent *ue = (uent *)malloc(sizeof(uent) * (size_t)N);
    int64_t capn = 1024, n = 0;
    uent *all = (uent *)malloc(sizeof(uent) * (size_t)capn);
    c->win_off = (int64_t *)calloc((size_t)W + 1, sizeof(int64_t));
    if (want_labels) c->labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)W * N);
    int32_t *slot_of_sorted = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int32_t w = 0; w < W; w++) {
        const uint64_t *Wd = c->words + (size_t)w * 3 * N;
        int32_t m = 0;
        for (int32_t r = 0; r < N; r++) {
            if (c->labels) c->labels[(size_t)w * N + r] = -1;
            if (Wd[2 * N + r] & MP_WIN_SKIP64) continue;
            kr[m].b0 = Wd[r]; kr[m].b1 = Wd[N + r]; kr[m].g = Wd[2 * N + r]; kr[m].row = r; m++;
        }
        qsort(kr, (size_t)m, sizeof(keyrow), cmp_keyrow);
        int32_t u = 0;
        for (int32_t i = 0; i < m; i++) {
            if (i == 0 || kr[i].b0 != kr[i - 1].b0 || kr[i].b1 != kr[i - 1].b1 || kr[i].g != kr[i - 1].g) {
                ue[u].b0 = kr[i].b0; ue[u].b1 = kr[i].b1; ue[u].g = kr[i].g; ue[u].count = 0; ue[u].first = kr[i].row; u++;
            }
            ue[u - 1].count++;
        }
        qsort(ue, (size_t)u, sizeof(uent), cmp_first);   /* first-seen order = dict insertion order */
        if (c->labels) {
            /* label = index of the row's entry in first-seen order */
            for (int32_t e = 0; e < u; e++) slot_of_sorted[e] = 0;
            for (int32_t i = 0; i < m; i++) {
                /* binary search entry by key among ue (sorted by first): linear is fine for an oracle */
                for (int32_t e = 0; e < u; e++)
                    if (ue[e].b0 == kr[i].b0 && ue[e].b1 == kr[i].b1 && ue[e].g == kr[i].g) { c->labels[(size_t)w * N + kr[i].row] = e; break; }
            }
        }
        if (n + u > capn) { while (n + u > capn) capn *= 2; all = (uent *)realloc(all, sizeof(uent) * (size_t)capn); }
        memcpy(all + n, ue, sizeof(uent) * (size_t)u);
        n += u;
        c->win_off[w + 1] = n;
    }
    free(kr); free(ue); free(slot_of_sorted);
    if (n_entries) *n_entries = n;
    if (n > cap) { free(all); c->n_ent = 0; return fail(c, MP_ERR_CAPACITY, "unique table needs %lld entries", (long long)n); }
    c->n_ent = n;
    c->u_b0 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1)); c->u_b1 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    c->u_g = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
    c->u_count = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1)); c->u_first = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < n; i++) {
        c->u_b0[i] = all[i].b0; c->u_b1[i] = all[i].b1; c->u_g[i] = all[i].g; c->u_count[i] = all[i].count; c->u_first[i] = all[i].first;
    }
    free(all);
    return MP_OK;
}

/* the device-side entropy gate (mprime.h): the checker never rejects a window itself */
int mp_set_entropy_gate(mp_ctx *c, double threshold) {
    if (!c) return MP_ERR_ARG;
    if (!(threshold >= 0)) return fail(c, MP_ERR_ARG, "mp_set_entropy_gate: the threshold must be >= 0");
    return MP_OK;
}
int mp_entropy_gate_result(mp_ctx *c, int32_t *n_rejected, uint8_t *rejected) {
    if (!c) return MP_ERR_ARG;
    if (n_rejected) *n_rejected = 0;
    if (rejected && c->n_win > 0) memset(rejected, 0, (size_t)c->n_win);
    return MP_OK;
}

int mp_get_unique(mp_ctx *c, int64_t *win_off, void *words, int32_t *count, int32_t *first_row) {
    if (!c || !c->win_off) return c ? fail(c, MP_ERR_ARG, "mp_window_unique has not run") : MP_ERR_ARG;
    int64_t n = c->n_ent;
    memcpy(win_off, c->win_off, sizeof(int64_t) * ((size_t)c->n_win + 1));
    for (int64_t i = 0; i < n; i++) {
        wr_word(words, (size_t)i, c->k, c->u_b0[i]);
        wr_word(words, (size_t)(n + i), c->k, c->u_b1[i]);
        wr_word(words, (size_t)(2 * n + i), c->k, c->u_g[i]);
    }
    memcpy(count, c->u_count, sizeof(int32_t) * (size_t)n);
    memcpy(first_row, c->u_first, sizeof(int32_t) * (size_t)n);
    return MP_OK;
}

int mp_get_labels(mp_ctx *c, int32_t w, int32_t *labels) {
    if (!c || !c->labels) return c ? fail(c, MP_ERR_ARG, "labels were not requested") : MP_ERR_ARG;
    if (w < 0 || w >= c->n_win) return fail(c, MP_ERR_ARG, "bad window");
    memcpy(labels, c->labels + (size_t)w * c->n_rows, sizeof(int32_t) * (size_t)c->n_rows);
    return MP_OK;
}

int mp_get_labels_many(mp_ctx *c, int32_t n, const int32_t *windows, int32_t *labels) {
    if (!c || !c->labels) return c ? fail(c, MP_ERR_ARG, "labels were not requested") : MP_ERR_ARG;
    if (n < 0 || (n && (!windows || !labels))) return fail(c, MP_ERR_ARG, "mp_get_labels_many: bad arguments");
    for (int32_t i = 0; i < n; i++) {
        int rc = mp_get_labels(c, windows[i], labels + (size_t)i * c->n_rows);
        if (rc) return rc;
    }
    return MP_OK;
}

/* ---- candidate x sequence evaluation: V20:1103-1130 + Y_distance V20:229-233 --------------- */
/* Y_distance: position j is a mismatch iff the concrete symbol is not in the IUPAC set of the
 * primer symbol; '-' is in no set (verified exhaustively against score_table). */
/* state_matrix (V20:541-554) + trans_matrix (V20:556-577) of every window: one row per universe k-mer */
static void stats_one(const char *kmer, int32_t k, int32_t v, int64_t *freq, int64_t *nn) {
    int32_t gaps = 0;
    for (int32_t j = 0; j < k; j++) gaps += kmer[j] == '-';
    if (gaps > v) return;                                  /* a gap row: not in `cover` (V20:689) */
    for (int32_t j = 0; j < k; j++) {
        int a = base_index(kmer[j]);
        if (a < 0) continue;                               /* the '-' row is dropped (V20:551-552) */
        freq[(size_t)a * k + j]++;
        if (j + 1 < k) {
            int b = base_index(kmer[j + 1]);
            if (b >= 0) nn[((size_t)j * 4 + a) * 4 + b]++;  /* pairs touching '-' do not count (V20:569-575) */
        }
    }
}

int mp_window_stats(mp_ctx *c, int64_t *freq, int64_t *nn) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    if (!freq || !nn) return fail(c, MP_ERR_ARG, "null output");
    int32_t N = c->n_rows, k = c->k;
    char km[MP_MAX_K + 1];
    memset(freq, 0, sizeof(int64_t) * (size_t)c->n_win * 4 * k);
    memset(nn, 0, sizeof(int64_t) * (size_t)c->n_win * (k - 1) * 16);
    int32_t e = 0;
    for (int32_t w = 0; w < c->n_win; w++) {
        int64_t *f = freq + (size_t)w * 4 * k, *t = nn + (size_t)w * (k - 1) * 16;
        for (int32_t r = 0; r < N; r++) {
            const char *kmer = c->kmers + ((size_t)w * N + r) * k;
            if (!kmer[0]) continue;                        /* exception slot: its expansions are extra rows */
            stats_one(kmer, k, c->v, f, t);
        }
        while (e < c->n_extra && c->extra_win[e] < w) e++;
        for (; e < c->n_extra && c->extra_win[e] == w; e++) {
            km[k] = 0;
            for (int32_t j = 0; j < k; j++)
                km[j] = (c->extra_words[3 * e + 2] >> j & 1) ? '-' : "ACGT"[(c->extra_words[3 * e] >> j & 1) | (c->extra_words[3 * e + 1] >> j & 1) << 1];
            stats_one(km, k, c->v, f, t);
        }
    }
    return MP_OK;
}

static void eval_one(const uint8_t *cand, int32_t k, int32_t v, uint64_t sF, uint64_t sR,
                     const char *kmer, int64_t *out) {
    int32_t nd = 0, gaps = 0;
    uint64_t D = 0;
    for (int32_t j = 0; j < k; j++) {
        if (kmer[j] == '-') gaps++;
        int bi = base_index(kmer[j]);
        int in_set = bi >= 0 && (cand[j] >> bi & 1);
        if (!in_set) { nd++; D |= 1ull << j; }            /* m_dist, V20:231 */
    }
    if (gaps > v) return;                                  /* not in `cover` (V20:689) */
    if (nd == 0) { out[0]++; return; }                     /* in optimal_primer_set, V20:1105-1106 */
    if (nd > v) return;                                    /* V20:1114 */
    if (!(D & sF)) out[1]++;                               /* V20:1120-1123 */
    if (!(D & sR)) out[2]++;                               /* V20:1124-1127 */
}

static void words_to_kmer(uint64_t b0, uint64_t b1, uint64_t g, int32_t k, char *kmer) {
    for (int32_t j = 0; j < k; j++)
        kmer[j] = (g >> j & 1) ? '-' : "ACGT"[(b0 >> j & 1) | (b1 >> j & 1) << 1];
}

static int eval_all(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR, int64_t *out) {
    int32_t N = c->n_rows, k = c->k;
    char km[MP_MAX_K + 1];
    int32_t e0 = 0;
    for (int32_t i = 0; i < n_cand; i++) {
        int32_t w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        int64_t *o = out + 3 * (size_t)i;
        o[0] = o[1] = o[2] = 0;
        for (int32_t r = 0; r < N; r++) {
            const char *kmer = c->kmers + ((size_t)w * N + r) * k;
            if (!kmer[0]) continue;                       /* exception slot */
            eval_one(codes + (size_t)i * k, k, c->v, sF, sR, kmer, o);
        }
        while (e0 < c->n_extra && c->extra_win[e0] < w) e0++;
        for (int32_t e = e0; e < c->n_extra && c->extra_win[e] == w; e++) {
            words_to_kmer(c->extra_words[3 * e], c->extra_words[3 * e + 1], c->extra_words[3 * e + 2], k, km);
            eval_one(codes + (size_t)i * k, k, c->v, sF, sR, km, o);
        }
    }
    return MP_OK;
}

int mp_eval_candidates(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes,
                       uint64_t sF, uint64_t sR, int64_t *out) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    if (n_cand < 0 || (n_cand && (!cw || !codes || !out))) return fail(c, MP_ERR_ARG, "bad arguments");
    return eval_all(c, n_cand, cw, codes, sF, sR, out);
}

int mp_eval_upload(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    free(c->cand_win); free(c->cand_codes);
    c->cand_win = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_cand + 1));
    c->cand_codes = (uint8_t *)malloc((size_t)(n_cand + 1) * c->k);
    memcpy(c->cand_win, cw, sizeof(int32_t) * (size_t)n_cand);
    memcpy(c->cand_codes, codes, (size_t)n_cand * c->k);
    c->n_cand = n_cand; c->sF = sF; c->sR = sR;
    return MP_OK;
}

int mp_eval_launch(mp_ctx *c, int64_t *out) {
    if (!c || !c->cand_win) return c ? fail(c, MP_ERR_ARG, "mp_eval_upload has not run") : MP_ERR_ARG;
    c->eval_n++;
    return eval_all(c, c->n_cand, c->cand_win, c->cand_codes, c->sF, c->sR, out);
}

int mp_window_stats_begin(mp_ctx *c) { return c ? MP_OK : MP_ERR_ARG; }                  /* (nothing runs beside anything here) */
int mp_window_stats_end(mp_ctx *c, int64_t *freq, int64_t *nn) { return mp_window_stats(c, freq, nn); }
int mp_eval_launch_alt(mp_ctx *c, int64_t *out) { return mp_eval_launch(c, out); }      /* (one "stream": the calling thread) */
int mp_eval_sync(mp_ctx *c) { return c ? MP_OK : MP_ERR_ARG; }

/* the rotating form (mprime.h): out is taken as zeroed and ADDED to, the other block is cleared */
int mp_eval_launch_rotating(mp_ctx *c, int64_t *out, int64_t *clear) {
    if (!c || !c->cand_win) return c ? fail(c, MP_ERR_ARG, "mp_eval_upload has not run") : MP_ERR_ARG;
    if (!out || clear == out) return fail(c, MP_ERR_ARG, "bad counter blocks");
    size_t n = 3 * (size_t)c->n_cand;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!tmp) return fail(c, MP_ERR_NOMEM, "out of memory");
    c->eval_n++;
    int rc = eval_all(c, c->n_cand, c->cand_win, c->cand_codes, c->sF, c->sR, tmp);
    if (rc == MP_OK) {
        for (size_t i = 0; i < n; i++) out[i] += tmp[i];
        if (clear) memset(clear, 0, sizeof(int64_t) * n);
    }
    free(tmp);
    return rc;
}

int mp_eval_timing(mp_ctx *c, int32_t reset, double *ms, int32_t *n) {
    if (!c) return MP_ERR_ARG;
    if (ms) *ms = c->eval_ms;
    if (n) *n = c->eval_n;
    if (reset) { c->eval_ms = 0; c->eval_n = 0; }
    return MP_OK;
}

int mp_eval_plan_info(mp_ctx *c, int32_t *info) {
    if (!c || !info) return MP_ERR_ARG;
    info[0] = info[1] = info[2] = info[3] = 0;
    return MP_OK;
}

int mp_eval_timing_samples(mp_ctx *c, int32_t cap, float *ms, int32_t *n) {
    (void)cap; (void)ms;
    if (!c || !n) return MP_ERR_ARG;
    *n = 0;                      /* the oracle keeps totals only */
    return MP_OK;
}

int mp_device_bytes(mp_ctx *c, int64_t *bytes) {
    if (!c || !bytes) return MP_ERR_ARG;
    *bytes = 0;
    return MP_OK;
}

/* (9) row shards: the checker is one process — a world of one rank, where every collective is the identity (mprime.h).  The
 * multi-rank tests run one checker per rank and let the Python host's gloo collectives stand in for RCCL (tests/test_multirank.py). */
static int one_rank(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    return c->n_ranks == 1 ? MP_OK : fail(c, MP_ERR_ARG, "no communicator (mp_comm_init has not run)");
}
int mp_comm_unique_id(uint8_t *id) {
    if (!id) return MP_ERR_ARG;
    memset(id, 0, MP_COMM_ID_BYTES);
    return MP_OK;
}
int mp_comm_init(mp_ctx *c, int32_t n_ranks, int32_t rank, const uint8_t *id) {
    (void)id;
    if (!c) return MP_ERR_ARG;
    if (n_ranks != 1 || rank != 0) return fail(c, MP_ERR_ARG, "the checker library has no collectives: n_ranks must be 1");
    c->n_ranks = 1;
    return MP_OK;
}
int mp_comm_destroy(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    c->n_ranks = 0;
    return MP_OK;
}
int mp_comm_describe(mp_ctx *c, int32_t *ranks_seen, char *library_path, int32_t path_bytes) {
    int rc = one_rank(c);
    if (rc) return rc;
    if (!ranks_seen || (path_bytes > 0 && !library_path)) return fail(c, MP_ERR_ARG, "mp_comm_describe: null output");
    ranks_seen[0] = 1; ranks_seen[1] = 0;
    if (path_bytes > 0) library_path[0] = 0;
    return MP_OK;
}
int mp_comm_allreduce_i64(mp_ctx *c, int64_t *buf, int64_t n) { (void)buf; (void)n; return one_rank(c); }
int mp_comm_allreduce_host_i64(mp_ctx *c, int64_t *buf, int64_t n) { (void)buf; (void)n; return one_rank(c); }
int mp_comm_allgather_i64(mp_ctx *c, int64_t value, int64_t *out) {
    int rc = one_rank(c);
    if (rc) return rc;
    if (!out) return fail(c, MP_ERR_ARG, "null output");
    out[0] = value;
    return MP_OK;
}
int mp_comm_allgatherv(mp_ctx *c, const void *send, int64_t n_bytes, const int64_t *counts, void *recv) {
    int rc = one_rank(c);
    if (rc) return rc;
    if (n_bytes < 0 || !counts || counts[0] != n_bytes || (n_bytes && (!send || !recv))) return fail(c, MP_ERR_ARG, "mp_comm_allgatherv: bad arguments");
    if (n_bytes) memcpy(recv, send, (size_t)n_bytes);
    return MP_OK;
}
int mp_comm_alltoall_counts(mp_ctx *c, const int64_t *send_counts, int64_t *recv_counts) {
    int rc = one_rank(c);
    if (rc) return rc;
    if (!send_counts || !recv_counts || send_counts[0] < 0) return fail(c, MP_ERR_ARG, "mp_comm_alltoall_counts: bad arguments");
    recv_counts[0] = send_counts[0];
    return MP_OK;
}
int mp_comm_alltoallv(mp_ctx *c, const void *send, const int64_t *send_counts, void *recv, const int64_t *recv_counts) {
    int rc = one_rank(c);
    if (rc) return rc;
    if (!send_counts || !recv_counts || send_counts[0] < 0 || recv_counts[0] != send_counts[0] || (send_counts[0] && (!send || !recv)))
        return fail(c, MP_ERR_ARG, "mp_comm_alltoallv: bad arguments");
    if (send_counts[0]) memcpy(recv, send, (size_t)send_counts[0]);
    return MP_OK;
}
int mp_eval_candidates_allreduce(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR, int64_t *out) {
    int rc = one_rank(c);
    if (rc) return rc;
    return mp_eval_candidates(c, n_cand, cw, codes, sF, sR, out);
}

/* =================================================================================================
 * (5) 3'-end dimer scan: restatement of Dimer.dimer_check (finDimer_V4.py:191-224, "FD" below) and
 * dimer_examination (get_Maxprimerset_V1.3.py:193-215, "MS"), on character strings.
 * ================================================================================================= */
static const char *iupac_members(uint8_t code) {
    /* FD:46-48 degenerate_base / MS:70-72 degenerate_pair: member order of each symbol */
    switch (code) {
    case 1: return "A"; case 2: return "C"; case 4: return "G"; case 8: return "T";
    case 5: return "AG"; case 10: return "CT"; case 3: return "AC"; case 12: return "GT";
    case 6: return "GC"; case 9: return "AT"; case 11: return "ATC"; case 14: return "GTC";
    case 7: return "GAC"; case 13: return "GAT"; case 15: return "ATGC";
    default: return NULL;
    }
}

/* expansion number `idx` of codes[0..n) in itertools.product order (last position fastest), FD:146-158 */
static void expand_at(const uint8_t *codes, int n, long long idx, char *out) {
    for (int p = n - 1; p >= 0; p--) {
        const char *m = iupac_members(codes[p]);
        int sz = (int)strlen(m);
        out[p] = m[idx % sz];
        idx /= sz;
    }
    out[n] = 0;
}

static long long n_expansions(const uint8_t *codes, int n) {
    long long d = 1;
    for (int p = 0; p < n; p++) {
        const char *m = iupac_members(codes[p]);
        if (!m) return -1;
        d *= (long long)strlen(m);
        if (d > (1LL << 24)) return -2;
    }
    return d;
}

static char comp_char(char c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C'; }

/* deltaG of a concrete end, FD:171-189, as the left-to-right sum of the caller's constants */
static double end_delta_g(const char *e, int l, const double *dg) {
    double g = 0;
    for (int n = 0; n + 1 < l; n++) g += dg[base_index(e[n + 1]) * 4 + base_index(e[n])];           /* FD:176-178 */
    int ta = l >= 2 && e[l - 2] == 'T' && e[l - 1] == 'A';                                            /* FD:179-180 */
    g += dg[16 + (base_index(e[0]) * 4 + base_index(e[l - 1])) * 2 + ta];                            /* FD:181-183 */
    g -= dg[48 + l];                                                                                   /* FD:185 */
    int sym = (l % 2 == 0);                                                                            /* FD:115-125 */
    for (int t = 0; sym && t < l / 2; t++) if (e[t] != comp_char(e[l / 2 + t])) sym = 0;
    if (sym) g += dg[48 + MP_DIMER_MAX_LEN + 1];                                                      /* FD:186-187 */
    return g;
}

/* one (x, y) pair; returns 1 and fills rec on the first passing combination */
static int dimer_pair(const uint8_t *cx, int lx, const uint8_t *cy, int ly, int l_hi, int l_lo,
                      const uint8_t *loss_hit, const double *dg, double dg_limit, int32_t *rec) {
    char e[MP_DIMER_MAX_LEN + 1], rc[MP_DIMER_MAX_LEN + 1], p[MP_DIMER_MAX_LEN + 1];
    long long dy = n_expansions(cy, ly);
    for (int l = l_hi; l >= l_lo; l--) {                       /* ends sorted by length, longest first (FD:193) */
        if (l <= 0 || l > lx) continue;
        const uint8_t *suffix = cx + (lx - l);                 /* primer[-l:] */
        long long de = n_expansions(suffix, l);
        for (long long ei = 0; ei < de; ei++) {
            expand_at(suffix, l, ei, e);
            for (int t = 0; t < l; t++) rc[t] = comp_char(e[l - 1 - t]);      /* reversecomplement(end) */
            rc[l] = 0;
            int gc = 0;
            for (int t = 0; t < l; t++) gc += e[t] == 'G' || e[t] == 'C';
            for (long long pi = 0; pi < dy; pi++) {            /* for p in degenerate_seq(ps), FD:197 */
                expand_at(cy, ly, pi, p);
                const char *f = strstr(p, rc);                 /* p.find(...): first occurrence, FD:198 */
                if (!f) continue;
                int idx = (int)(f - p);
                int d2 = ly - l - idx;                         /* FD:203 */
                int hit = loss_hit[((size_t)l * (MP_DIMER_MAX_LEN + 1) + gc) * 64 + d2];      /* Loss >= threshold */
                if (!hit && d2 == 0) hit = end_delta_g(e, l, dg) < dg_limit;                  /* delta_G < -5 and d1 == d2 */
                if (hit) {
                    rec[2] = l; rec[3] = (int32_t)ei; rec[4] = (int32_t)pi; rec[5] = idx;
                    return 1;
                }
            }
        }
    }
    return 0;
}

int mp_dimer_scan(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int32_t mode, int32_t n_new,
                  const uint8_t *loss_hit, const double *dg, double dg_limit, int64_t cap, int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || !n_hits || (mode != 0 && mode != 1)) return fail(c, MP_ERR_ARG, "mp_dimer_scan: bad arguments");
    for (int32_t i = 0; i < n; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > MP_DIMER_MAX_LEN) return fail(c, MP_ERR_ARG, "primer %d has length %d (1..%d supported)", i, len, MP_DIMER_MAX_LEN);
        long long d = n_expansions(codes + off[i], len);
        if (d < 0) return fail(c, MP_ERR_ARG, d == -1 ? "primer %d holds a gap / unknown symbol" : "primer %d has too many expansions", i);
    }
    int64_t nh = 0;
    for (int32_t x = 0; x < n; x++) {
        int lx = off[x + 1] - off[x];
        for (int32_t y = (mode == 0 ? x : 0); y < n; y++) {
            if (mode == 1 && x >= n_new && y >= n_new) continue;
            int ly = off[y + 1] - off[y];
            int32_t rec[6] = {x, y, 0, 0, 0, 0};
            int l_hi, l_lo;
            if (mode == 0) { l_hi = lx < 18 ? lx : 18; l_lo = lx < 5 ? lx : 5; }     /* FD:162-169: primer[-i:], i = 5..18 */
            else { l_hi = lx - 1; l_lo = 5; }                                          /* MS:149-154: range(5, len) */
            if (dimer_pair(codes + off[x], lx, codes + off[y], ly, l_hi, l_lo, loss_hit, dg, dg_limit, rec)) {
                if (nh < cap && hits) memcpy(hits + 6 * nh, rec, sizeof rec);
                nh++;
            }
        }
    }
    *n_hits = nh;
    return MP_OK;
}

/* Primers_filter.dimer_check (get_multiPrime_V8.py:419-438), one ordered pair x -> y at a time */
int mp_dimer_pairs(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int64_t n_pairs, const int32_t *pairs,
                   const uint8_t *loss_hit, const double *dg, double dg_limit, uint8_t *flags) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || n_pairs < 0 || (n_pairs && (!pairs || !flags)))
        return fail(c, MP_ERR_ARG, "mp_dimer_pairs: bad arguments");
    for (int32_t i = 0; i < n; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > MP_DIMER_MAX_LEN || n_expansions(codes + off[i], len) < 0)
            return fail(c, MP_ERR_ARG, "primer %d is not usable (length %d)", i, len);
    }
    for (int64_t p = 0; p < n_pairs; p++) {
        int32_t x = pairs[2 * p], y = pairs[2 * p + 1];
        if (x < 0 || x >= n || y < 0 || y >= n) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)p);
        int lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
        int32_t rec[6];
        flags[p] = (uint8_t)dimer_pair(codes + off[x], lx, codes + off[y], ly, lx < 18 ? lx : 18, lx < 5 ? lx : 5,
                                       loss_hit, dg, dg_limit, rec);
    }
    return MP_OK;
}

/* len(set(un_cover_list)) of get_multiPrime_V8.py:560-569 on bitsets */
int mp_pair_coverage(mp_ctx *c, int32_t n_sets, int32_t n_words, const uint64_t *a, const uint64_t *b, int64_t n_pairs,
                     const int32_t *pairs, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_sets < 0 || n_words < 0 || n_pairs < 0 || (n_pairs && (!a || !b || !pairs || !out)))
        return fail(c, MP_ERR_ARG, "mp_pair_coverage: bad arguments");
    for (int64_t p = 0; p < n_pairs; p++) {
        int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
        if (i < 0 || i >= n_sets || j < 0 || j >= n_sets) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)p);
        int32_t cnt = 0;
        for (int32_t w = 0; w < n_words; w++) {
            uint64_t x = a[(size_t)i * n_words + w] | b[(size_t)j * n_words + w];
            while (x) { x &= x - 1; cnt++; }
        }
        out[p] = cnt;
    }
    return MP_OK;
}

/* bitset form of gap_seq_id / non_coverage_seq_id for one candidate per entry (V20:689-698, 1107-1127) */
int mp_eval_masks(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR,
                  uint64_t *not_f, uint64_t *not_r) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    if (n_cand < 0 || (n_cand && (!cw || !codes || !not_f || !not_r))) return fail(c, MP_ERR_ARG, "bad arguments");
    int32_t N = c->n_rows, k = c->k;
    size_t nw = ((size_t)N + 63) / 64;
    for (int32_t i = 0; i < n_cand; i++) {
        int32_t w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        uint64_t *F = not_f + (size_t)i * nw, *R = not_r + (size_t)i * nw;
        memset(F, 0, nw * 8); memset(R, 0, nw * 8);
        for (int32_t r = 0; r < N; r++) {
            const char *kmer = c->kmers + ((size_t)w * N + r) * k;
            if (!kmer[0]) continue;                                   /* exception slot: the host owns it */
            int64_t o[3] = {0, 0, 0};
            int gaps = 0;
            for (int32_t j = 0; j < k; j++) gaps += kmer[j] == '-';
            int bf, br;
            if (gaps > c->v) bf = br = 1;                             /* gap row -> gap_seq_id */
            else {
                eval_one(codes + (size_t)i * k, k, c->v, sF, sR, kmer, o);
                bf = !(o[0] || o[1]);                                  /* not perfect and not F-admissible -> F_non_cover */
                br = !(o[0] || o[2]);
            }
            if (bf) F[r >> 6] |= 1ull << (r & 63);
            if (br) R[r >> 6] |= 1ull << (r & 63);
        }
    }
    return MP_OK;
}

/* (4d) the resident form: the same masks kept in the context, single-bit fix-ups, popcounts of unions */
int mp_eval_masks_resident(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR) {
    if (!c || !c->words) return c ? fail(c, MP_ERR_ARG, "no windows built") : MP_ERR_ARG;
    size_t nw = ((size_t)c->n_rows + 63) / 64;
    free(c->mask_f); free(c->mask_r);
    c->mask_f = c->mask_r = NULL; c->n_masks = 0;
    if (n_cand <= 0) return n_cand < 0 ? fail(c, MP_ERR_ARG, "bad arguments") : MP_OK;
    c->mask_f = (uint64_t *)calloc((size_t)n_cand * nw, 8);
    c->mask_r = (uint64_t *)calloc((size_t)n_cand * nw, 8);
    if (!c->mask_f || !c->mask_r) return fail(c, MP_ERR_NOMEM, "out of memory");
    int rc = mp_eval_masks(c, n_cand, cw, codes, sF, sR, c->mask_f, c->mask_r);
    if (rc == MP_OK) c->n_masks = n_cand;
    return rc;
}

int mp_masks_set_bits(mp_ctx *c, int64_t n, const int32_t *cand, const int32_t *row, const uint8_t *which, const uint8_t *value) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || (n && (!cand || !row || !which || !value))) return fail(c, MP_ERR_ARG, "mp_masks_set_bits: bad arguments");
    if (n == 0) return MP_OK;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    size_t nw = ((size_t)c->n_rows + 63) / 64;
    for (int64_t i = 0; i < n; i++) {
        if (cand[i] < 0 || cand[i] >= c->n_masks || row[i] < 0 || row[i] >= c->n_rows) return fail(c, MP_ERR_ARG, "assignment %lld out of range", (long long)i);
        uint64_t *m = (which[i] ? c->mask_r : c->mask_f) + (size_t)cand[i] * nw + (size_t)(row[i] >> 6);
        if (value[i]) *m |= 1ull << (row[i] & 63);
        else *m &= ~(1ull << (row[i] & 63));
    }
    return MP_OK;
}

int mp_masks_fetch(mp_ctx *c, uint64_t *not_f, uint64_t *not_r) {
    if (!c) return MP_ERR_ARG;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    if (!not_f || !not_r) return fail(c, MP_ERR_ARG, "null output");
    size_t n = (size_t)c->n_masks * (((size_t)c->n_rows + 63) / 64);
    memcpy(not_f, c->mask_f, n * 8); memcpy(not_r, c->mask_r, n * 8);
    return MP_OK;
}

int mp_pair_coverage_resident(mp_ctx *c, int64_t n_pairs, const int32_t *pairs, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_pairs < 0 || (n_pairs && (!pairs || !out))) return fail(c, MP_ERR_ARG, "mp_pair_coverage_resident: bad arguments");
    if (n_pairs == 0) return MP_OK;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    size_t nw = ((size_t)c->n_rows + 63) / 64;
    for (int64_t p = 0; p < n_pairs; p++) {
        int32_t i = pairs[2 * p], j = pairs[2 * p + 1];
        if (i < 0 || i >= c->n_masks || j < 0 || j >= c->n_masks) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)p);
        int cnt = 0;
        for (size_t w = 0; w < nw; w++) cnt += __builtin_popcountll(c->mask_f[(size_t)i * nw + w] | c->mask_r[(size_t)j * nw + w]);
        out[p] = cnt;
    }
    return MP_OK;
}

/* =================================================================================================
 * (7) exact in-silico PCR: Product.get_PCR_PRODUCT (extract_PCR_product_V1.py:189-216, "PCR"), on strings
 * ================================================================================================= */
static const char *find_str(const char *hay, int hay_len, const char *needle, int n_len) {
    for (int i = 0; i + n_len <= hay_len; i++)
        if (!memcmp(hay + i, needle, (size_t)n_len)) return hay + i;
    return NULL;
}

int mp_pcr_scan(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_pairs,
                const uint8_t *codes, const int32_t *off, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || n_pairs < 0 || (n_rows && (!bytes || !row_off)) || (n_pairs && (!codes || !off)) || (n_rows && n_pairs && !out))
        return fail(c, MP_ERR_ARG, "mp_pcr_scan: bad arguments");
    for (int32_t i = 0; i < 2 * n_pairs; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > MP_PATTERN_MAX_LEN || n_expansions(codes + off[i], len) < 0)
            return fail(c, MP_ERR_ARG, "primer %d is not usable (length %d)", i, len);
    }
    char f[MP_PATTERN_MAX_LEN + 1], r[MP_PATTERN_MAX_LEN + 1], rc[MP_PATTERN_MAX_LEN + 1];
    for (int32_t p = 0; p < n_pairs; p++) {
        const uint8_t *cf = codes + off[2 * p], *cr = codes + off[2 * p + 1];
        int lf = off[2 * p + 1] - off[2 * p], lr = off[2 * p + 2] - off[2 * p + 1];
        long long df = n_expansions(cf, lf), dr = n_expansions(cr, lr);
        for (int32_t row = 0; row < n_rows; row++) {
            const char *s = (const char *)bytes + row_off[row];
            int len = (int)(row_off[row + 1] - row_off[row]);
            int32_t *o = out + ((size_t)p * n_rows + row) * 4;
            o[0] = o[1] = o[2] = o[3] = -1;
            for (long long fi = 0; fi < df && o[0] < 0; fi++) {            /* for sequence in Fseq, PCR:198 */
                expand_at(cf, lf, fi, f);
                const char *a = find_str(s, len, f, lf);                   /* re.search(sequence, i), PCR:199 */
                if (!a) continue;
                int p1 = (int)(a - s);
                const char *b = find_str(a + lf, len - p1 - lf, f, lf);    /* i.split(sequence): next occurrence */
                int end = b ? (int)(b - s) : len;                          /* Product = sequence + line[1], PCR:200-201 */
                for (long long ri = 0; ri < dr; ri++) {                    /* for sequence2 in Rseq, PCR:202 */
                    expand_at(cr, lr, ri, r);
                    for (int t = 0; t < lr; t++) rc[t] = comp_char(r[lr - 1 - t]);
                    const char *q = find_str(a, end - p1, rc, lr);         /* re.search(RC(sequence2), Product) */
                    if (q) { o[0] = (int32_t)fi; o[1] = p1; o[2] = (int32_t)ri; o[3] = (int)(q - s); break; }
                }
            }
        }
    }
    return MP_OK;
}

/* (8) k-mismatch primer-site scan: the acceptance rule of mprime.h, character by character.  bowtie2 and
 * samtools are not in this image; the rule is pinned to the bowtie2 run the reference ships (tests/test_validate_bwt.py:
 * 1158 sequences decided identically). */
int mp_kmm_scan(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_pat, const uint8_t *pat_codes,
                const int32_t *pat_off, int32_t max_mm, int32_t term, int64_t cap, int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || n_pat < 0 || !n_hits || cap < 0 || (cap && !hits) || (n_rows && (!bytes || !row_off)) ||
        (n_pat && (!pat_codes || !pat_off)) || max_mm < 0 || term < 0)
        return fail(c, MP_ERR_ARG, "mp_kmm_scan: bad arguments");
    *n_hits = 0;
    static const char base_of[9] = {0, 'A', 'C', 0, 'G', 0, 0, 0, 'T'};
    for (int32_t i = 0; i < n_pat; i++) {
        int len = pat_off[i + 1] - pat_off[i];
        if (len < 4 || len > MP_PATTERN_MAX_LEN) return fail(c, MP_ERR_ARG, "pattern %d has length %d (4..%d supported)", i, len, MP_PATTERN_MAX_LEN);
        for (int j = 0; j < len; j++) {
            uint8_t m = pat_codes[pat_off[i] + j];
            if (m != 1 && m != 2 && m != 4 && m != 8) return fail(c, MP_ERR_ARG, "pattern %d is not a concrete A/C/G/T sequence", i);
        }
    }
    int64_t n = 0;
    for (int32_t r = 0; r < n_rows; r++) {
        const uint8_t *s = bytes + row_off[r];
        int64_t len = row_off[r + 1] - row_off[r];
        for (int64_t p = 0; p < len; p++) {
            for (int32_t i = 0; i < n_pat; i++) {
                int m = pat_off[i + 1] - pat_off[i];
                if (p + m > len || term > m) continue;
                for (int strand = 0; strand < 2; strand++) {
                    int mism = 0, run = 0;                /* run: matches since the last mismatch, left to right */
                    for (int j = 0; j < m; j++) {
                        char want = base_of[pat_codes[pat_off[i] + (strand ? m - 1 - j : j)]];
                        if (strand) want = want == 'A' ? 'T' : want == 'C' ? 'G' : want == 'G' ? 'C' : 'A';
                        char have = (char)toupper(s[p + j]);
                        if (have == want) run++;          /* a character outside ACGT equals no base */
                        else { mism++; run = 0; }
                    }
                    if (mism <= max_mm && run >= term) {  /* trailing number of the MD:Z string >= threshold (BWT:253-259) */
                        if (n < cap) { hits[4 * n] = r; hits[4 * n + 1] = (int32_t)p; hits[4 * n + 2] = i; hits[4 * n + 3] = strand; }
                        n++;
                    }
                }
            }
        }
    }
    *n_hits = n;
    return MP_OK;
}

/* (8b) the resident sequence store: the checker keeps the characters and runs the byte scans of (7) / (8) on them */
int mp_seq_free(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    free(c->sq_bytes); free(c->sq_off);
    c->sq_bytes = NULL; c->sq_off = NULL; c->sq_n = 0;
    return MP_OK;
}
int mp_seq_load(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || (n_rows && (!bytes || !row_off))) return fail(c, MP_ERR_ARG, "mp_seq_load: bad arguments");
    mp_seq_free(c);
    if (n_rows == 0) return MP_OK;
    int64_t total = row_off[n_rows] - row_off[0];
    c->sq_bytes = malloc((size_t)total + 1);
    c->sq_off = malloc(sizeof(int64_t) * ((size_t)n_rows + 1));
    if (!c->sq_bytes || !c->sq_off) { mp_seq_free(c); return fail(c, MP_ERR_NOMEM, "mp_seq_load: out of memory"); }
    memcpy(c->sq_bytes, bytes + row_off[0], (size_t)total);
    for (int32_t r = 0; r <= n_rows; r++) c->sq_off[r] = row_off[r] - row_off[0];
    c->sq_n = n_rows;
    return MP_OK;
}
int mp_seq_info(mp_ctx *c, int32_t *n_rows, int64_t *n_bases, int64_t *device_bytes) {
    if (!c) return MP_ERR_ARG;
    if (n_rows) *n_rows = c->sq_n;
    if (n_bases) *n_bases = c->sq_n ? c->sq_off[c->sq_n] : 0;
    if (device_bytes) *device_bytes = 0;
    return MP_OK;
}
int mp_pcr_scan_resident(mp_ctx *c, int32_t n_pairs, const uint8_t *codes, const int32_t *off, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (c->sq_n == 0 || n_pairs == 0) return MP_OK;
    return mp_pcr_scan(c, c->sq_bytes, c->sq_off, c->sq_n, n_pairs, codes, off, out);
}
int mp_kmm_scan_resident(mp_ctx *c, int32_t n_pat, const uint8_t *pat_codes, const int32_t *pat_off, int32_t max_mm, int32_t term, int64_t cap,
                         int32_t *hits, int64_t *n_hits) {
    if (!c || !n_hits) return MP_ERR_ARG;
    *n_hits = 0;
    if (c->sq_n == 0 || n_pat == 0) return MP_OK;
    return mp_kmm_scan(c, c->sq_bytes, c->sq_off, c->sq_n, n_pat, pat_codes, pat_off, max_mm, term, cap, hits, n_hits);
}
