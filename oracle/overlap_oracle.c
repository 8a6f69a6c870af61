/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of what
 * ExactOverlapper::overlaps() computes (/root/reference/src/overlapper.cpp:28-150).
 *
 * PARITY PINNING: this restatement is pinned against outputs of the reference itself,
 * compiled from /root/reference by oracle/Makefile (target `ref`) and run in the build
 * container: the tests/golden/ fixtures were produced by tests/golden/make_golden.py from
 * oracle/_ref/ref_overlapper, and tests/test_oracle.py checks this file against every one
 * of them (plus live against oracle/_ref when the binary is present).
 *
 * The reference walks an enhanced suffix array of all reads (overlapper.cpp:33-36) and
 * emits two families of rows (SURVEY.md section 8a-2 has the derivation):
 *
 *   A rows (overlapper.cpp:64-91, pushes at :40-58, pops at :121-146)
 *      for every ordered pair of distinct read indices (a, b): the single LONGEST l with
 *      max(min_length,1) <= l <= min(la, lb) and a[la-l:] == b[:l]
 *      -> (a, b, la-l, la, 0, l)          ("stack top" = deepest pushed suffix, :77-82)
 *   B rows (overlapper.cpp:95-115)
 *      for every b with lb >= max(min_length,1) and EVERY occurrence p of the whole of b
 *      inside another read a (index != b) -> (a, b, p, p+lb, 0, lb)
 *
 * A and B are not de-duplicated against each other (b == suffix of a gives the row twice).
 * Row ORDER is not part of the contract (unordered_map iteration, :30,:68); callers compare
 * sorted multisets.  This file emits a-major, then by astart, then by b.
 *
 * Method here (no suffix tree): hash the first K = min(16, m) bytes of every read b with
 * lb >= m; roll the same hash over every position p <= la-m of every read a; on a hash hit
 * memcmp the full n = min(la-p, lb) bytes.  la-p <= lb is a suffix-prefix (A) candidate and the
 * first success in ascending p is the longest; la-p >= lb is a containment (B) occurrence.
 * Byte equality throughout, exactly like the reference's CharString compare.               */
#include "overlap_oracle.h"

#include <stdlib.h>
#include <string.h>

#define HASH_BASE 0x100000001b3ULL

typedef struct {
    oracle_row* rows;
    uint64_t n, cap;
} rowvec;

static int push_row(rowvec* v, uint32_t a, uint32_t b, int32_t as, int32_t ae, int32_t be) {
    if (v->n == v->cap) {
        uint64_t ncap = v->cap ? v->cap * 2 : 1024;
        oracle_row* nr = (oracle_row*)realloc(v->rows, ncap * sizeof(oracle_row));
        if (!nr) return -1;
        v->rows = nr;
        v->cap = ncap;
    }
    oracle_row* r = &v->rows[v->n++];
    r->a_idx = a; r->b_idx = b; r->astart = as; r->aend = ae; r->bstart = 0; r->bend = be;
    return 0;
}

static uint64_t hash_bytes(const uint8_t* s, uint32_t k) {
    uint64_t h = 0;
    for (uint32_t i = 0; i < k; ++i) h = h * HASH_BASE + (uint64_t)s[i] + 1;
    return h;
}

static uint32_t next_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

static int overlaps_impl(const uint8_t* const* seqs, const uint32_t* lens, uint32_t n,
                         uint32_t min_length, oracle_row** rows_out, uint64_t* nrows_out) {
    rowvec out = {0, 0, 0};
    *rows_out = 0;
    *nrows_out = 0;
    if (n == 0) return 0;
    /* m=0 behaves as m=1 in the reference: a suffix array holds no empty suffixes. */
    uint32_t m = min_length ? min_length : 1;
    uint32_t K = m < 16 ? m : 16;

    /* bucket table: heads[] + next[] chains of read indices keyed by prefix hash.
     * Chains are built back to front so each chain lists reads in ascending index. */
    uint32_t tsize = next_pow2(n * 2 + 1);
    uint32_t* head = (uint32_t*)malloc((size_t)tsize * sizeof(uint32_t));
    uint32_t* next = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint64_t* phash = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
    uint32_t* seen = (uint32_t*)malloc((size_t)n * sizeof(uint32_t)); /* A-row emitted for (a,b)? */
    if (!head || !next || !phash || !seen) { free(head); free(next); free(phash); free(seen); return -1; }
    memset(head, 0xff, (size_t)tsize * sizeof(uint32_t));
    memset(seen, 0xff, (size_t)n * sizeof(uint32_t));
    for (uint32_t i = n; i-- > 0;) {
        next[i] = 0xffffffffu;
        if (lens[i] < m) continue; /* shorter than min_length: can never be a `b` */
        phash[i] = hash_bytes(seqs[i], K);
        uint32_t slot = (uint32_t)((phash[i] * 0x9E3779B97F4A7C15ULL) >> 32) & (tsize - 1);
        next[i] = head[slot];
        head[slot] = i;
    }
    uint64_t top = 1; /* HASH_BASE^(K-1) */
    for (uint32_t i = 1; i < K; ++i) top *= HASH_BASE;

    int rc = 0;
    for (uint32_t a = 0; a < n && rc == 0; ++a) {
        uint32_t la = lens[a];
        if (la < m) continue; /* no suffix of length >= m, and nothing of length >= m fits inside */
        const uint8_t* sa = seqs[a];
        uint64_t h = hash_bytes(sa, K);
        uint32_t last = la - m;
        for (uint32_t p = 0;; ++p) {
            uint32_t slot = (uint32_t)((h * 0x9E3779B97F4A7C15ULL) >> 32) & (tsize - 1);
            for (uint32_t b = head[slot]; b != 0xffffffffu; b = next[b]) {
                if (b == a || phash[b] != h) continue;
                uint32_t lb = lens[b];
                uint32_t rem = la - p;
                uint32_t cmp = rem < lb ? rem : lb;
                if (memcmp(sa + p, seqs[b], cmp) != 0) continue;
                if (rem <= lb && seen[b] != a) { /* A row: first success in ascending p = longest */
                    seen[b] = a;
                    if (push_row(&out, a, b, (int32_t)p, (int32_t)la, (int32_t)rem)) { rc = -1; break; }
                }
                if (rem >= lb) { /* B row: every occurrence of the whole of b */
                    if (push_row(&out, a, b, (int32_t)p, (int32_t)(p + lb), (int32_t)lb)) { rc = -1; break; }
                }
            }
            if (p == last || rc) break;
            h = (h - ((uint64_t)sa[p] + 1) * top) * HASH_BASE + (uint64_t)sa[p + K] + 1;
        }
    }
    free(head); free(next); free(phash); free(seen);
    if (rc) { free(out.rows); return rc; }
    *rows_out = out.rows;
    *nrows_out = out.n;
    return 0;
}

int oracle_overlaps(const uint8_t* const* seqs, const uint32_t* lens, uint32_t n,
                    uint32_t min_length, oracle_row** rows_out, uint64_t* nrows_out) {
    return overlaps_impl(seqs, lens, n, min_length, rows_out, nrows_out);
}

int oracle_overlaps_cat(const uint8_t* cat, const uint64_t* offs, const uint32_t* lens, uint32_t n,
                        uint32_t min_length, oracle_row** rows_out, uint64_t* nrows_out) {
    const uint8_t** seqs = (const uint8_t**)malloc((size_t)(n ? n : 1) * sizeof(uint8_t*));
    if (!seqs) return -1;
    for (uint32_t i = 0; i < n; ++i) seqs[i] = cat + offs[i];
    int rc = overlaps_impl(seqs, lens, n, min_length, rows_out, nrows_out);
    free(seqs);
    return rc;
}

void oracle_free(void* p) { free(p); }
