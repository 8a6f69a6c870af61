/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the banded seed-extension mode (po_overlaps_ex).
 *
 * PARITY UNPINNED for max_diff > 0: the reference overlapper is exact (/root/reference/src/overlapper.cpp:28-150 has
 * no error tolerance anywhere), so nothing of the reference can check an inexact overlap.  This file is the build's
 * own statement of what that mode computes, written as the plainest possible row-by-row DP so that the HIP kernel
 * (phasm_amd/csrc/extend.hip.h: antidiagonal sweep, one lane per diagonal) has something independent to agree with.
 * With max_diff = 0 it reduces to the exact contract, and THAT is pinned: tests/test_oracle.py checks this file with
 * max_diff = 0 against every reference golden.
 *
 * Definition.  Anchor: b's K-byte prefix occurs at a[p..p+K), p <= la - m, lb >= m, a != b (index).  x = a[p:],
 * y = b, rem = la - p.  D[i][j] = unit-cost edit distance of x[:i], y[:j] over cells with |j - i| <= W only.
 *   A: rem <= lb + W and min_j D[rem][j] <= E  (1 <= j <= lb)    -> row (a, b, p, la, 0, j*)
 *   B: lb <= rem + W and min_i D[i][lb]  <= E  (1 <= i <= rem)   -> row (a, b, p, p + i*, 0, lb)
 *   ties: closest to the main diagonal, then the smaller coordinate.
 * A rows: only the smallest p per ordered pair (a, b) ("longest only", as in the exact contract); B rows: every one. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t a_idx, b_idx;
    int32_t astart, aend, bstart, bend;
} ext_row;

#define EXT_INF (1u << 24)

static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

/* band rows: cell (i, j) lives at row[j - i + W]; out-of-band / out-of-matrix = INF */
static void extend_one(const uint8_t* x, uint32_t rem, const uint8_t* y, uint32_t lb, uint32_t E, uint32_t W,
                       int* okA, uint32_t* jA, int* okB, uint32_t* iB) {
    const uint32_t nb = 2 * W + 1;
    uint32_t* base = (uint32_t*)malloc(sizeof(uint32_t) * nb * 2);
    uint32_t* prev = base;
    uint32_t* cur = base + nb;
    for (uint32_t k = 0; k < nb; ++k) prev[k] = cur[k] = EXT_INF;
    *okA = *okB = 0;
    uint32_t bestA = 0xFFFFFFFFu, bestB = 0xFFFFFFFFu;
    const int canA = rem <= lb + W, canB = lb <= rem + W;
    for (uint32_t i = 0; i <= rem; ++i) {
        for (uint32_t k = 0; k < nb; ++k) {
            const int64_t j = (int64_t)i + (int64_t)k - (int64_t)W;
            uint32_t v = EXT_INF;
            if (j >= 0 && j <= (int64_t)lb) {
                if (i == 0 && j == 0) {
                    v = 0;
                } else {
                    if (i > 0 && j > 0) v = umin(v, prev[k] + (x[i - 1] != y[j - 1] ? 1u : 0u)); /* (i-1, j-1): same diagonal   */
                    if (i > 0 && k + 1 < nb) v = umin(v, prev[k + 1] + 1u);                        /* (i-1, j): diagonal + 1     */
                    if (j > 0 && k > 0) v = umin(v, cur[k - 1] + 1u);                              /* (i, j-1): diagonal - 1     */
                    v = umin(v, EXT_INF);
                }
                /* ends */
                const uint32_t off = k >= W ? k - W : W - k;
                if (canA && i == rem && j >= 1 && v <= E) {
                    const uint32_t key = (v << 8) | (off << 1) | (k > W ? 1u : 0u);
                    if (key < bestA) { bestA = key; *jA = (uint32_t)j; *okA = 1; }
                }
                if (canB && j == (int64_t)lb && i >= 1 && v <= E) {
                    const uint32_t key = (v << 8) | (off << 1) | (k < W ? 1u : 0u);
                    if (key < bestB) { bestB = key; *iB = i; *okB = 1; }
                }
            }
            cur[k] = v;
        }
        uint32_t* t = prev; prev = cur; cur = t;
    }
    free(base);
}

typedef struct { ext_row* rows; uint64_t n, cap; } rowvec;
static int push(rowvec* v, uint32_t a, uint32_t b, int32_t s, int32_t e, int32_t be) {
    if (v->n == v->cap) {
        uint64_t nc = v->cap ? v->cap * 2 : 1024;
        ext_row* nr = (ext_row*)realloc(v->rows, nc * sizeof(ext_row));
        if (!nr) return -1;
        v->rows = nr; v->cap = nc;
    }
    ext_row* r = &v->rows[v->n++];
    r->a_idx = a; r->b_idx = b; r->astart = s; r->aend = e; r->bstart = 0; r->bend = be;
    return 0;
}

/* reads: cat[offs[i] .. offs[i]+lens[i]).  K = anchor length the library uses (min(bases per 64-bit word, m)). */
int oracle_overlaps_ex(const uint8_t* cat, const uint64_t* offs, const uint32_t* lens, uint32_t n, uint32_t min_length,
                       uint32_t max_diff, uint32_t band, uint32_t K, ext_row** rows_out, uint64_t* nrows_out) {
    rowvec out = {0, 0, 0};
    *rows_out = 0; *nrows_out = 0;
    const uint32_t m = min_length ? min_length : 1;
    if (K > m) K = m;
    const uint32_t W = max_diff ? band : 0;
    uint32_t* seenA = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (!seenA) return -1;
    memset(seenA, 0xff, sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t a = 0; a < n; ++a) {
        const uint32_t la = lens[a];
        if (la < m) continue;
        const uint8_t* sa = cat + offs[a];
        for (uint32_t p = 0; p + m <= la; ++p) {
            for (uint32_t b = 0; b < n; ++b) {
                if (b == a || lens[b] < m) continue;
                const uint8_t* sb = cat + offs[b];
                if (memcmp(sa + p, sb, K) != 0) continue;
                int okA, okB; uint32_t jA = 0, iB = 0;
                extend_one(sa + p, la - p, sb, lens[b], max_diff, W, &okA, &jA, &okB, &iB);
                if (okA && seenA[b] != a) {       /* ascending p: the first accepted A of the pair is the longest */
                    seenA[b] = a;
                    if (push(&out, a, b, (int32_t)p, (int32_t)la, (int32_t)jA)) return -1;
                }
                if (okB && push(&out, a, b, (int32_t)p, (int32_t)(p + iB), (int32_t)lens[b])) return -1;
            }
        }
    }
    free(seenA);
    *rows_out = out.rows; *nrows_out = out.n;
    return 0;
}

void oracle_ex_free(void* p) { free(p); }

/* ---- pair-level entry points for the FULL-SIZE tests (tests/test_gpu_fullsize.py): the candidates of a 100 k-read set
 * are known from the generator's truth, so the checker does not search anchors, it evaluates listed pairs.
 * cat = all reads back to back; candidate k compares x = cat[ax[k] .. ax[k]+rem[k]) with y = cat[ay[k] .. ay[k]+lb[k]). */

/* Hamming distance of the first n[k] bytes of both sides (substitution-only noise: an upper bound of the DP's cost on
 * the main diagonal, and equal to it unless two indels beat the substitutions between them). */
void oracle_pair_hamming(const uint8_t* cat, uint64_t npairs, const uint64_t* ax, const uint64_t* ay, const uint32_t* n, uint32_t* out) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t k = 0; k < (int64_t)npairs; ++k) {
        const uint8_t* x = cat + ax[k];
        const uint8_t* y = cat + ay[k];
        uint32_t h = 0;
        for (uint32_t i = 0; i < n[k]; ++i) h += x[i] != y[i];
        out[k] = h;
    }
}

/* extend_one on listed pairs: out[4k..4k+4) = {okA, jA, okB, iB}. */
void oracle_extend_pairs(const uint8_t* cat, uint64_t npairs, const uint64_t* ax, const uint32_t* rem, const uint64_t* ay,
                         const uint32_t* lb, uint32_t max_diff, uint32_t band, uint32_t* out) {
    const uint32_t W = max_diff ? band : 0;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t k = 0; k < (int64_t)npairs; ++k) {
        int okA, okB; uint32_t jA = 0, iB = 0;
        extend_one(cat + ax[k], rem[k], cat + ay[k], lb[k], max_diff, W, &okA, &jA, &okB, &iB);
        out[4 * k] = (uint32_t)okA; out[4 * k + 1] = jA; out[4 * k + 2] = (uint32_t)okB; out[4 * k + 3] = iB;
    }
}
