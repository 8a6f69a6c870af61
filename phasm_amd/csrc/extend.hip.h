// Banded seed-extension DP (po_overlaps_ex): the verify step of kernels.hip.h with up to `max_diff` differences.
//
// The reference overlapper is exact (/root/reference/src/overlapper.cpp:28-150: suffix-tree matches only), so this
// mode has no reference to be checked against for max_diff > 0 -- PARITY UNPINNED there; its checker is the CPU
// restatement oracle/extend_oracle.c.  With max_diff = 0 it must (and does, tests/test_gpu_extend.py) give exactly
// the rows of po_overlaps, which the reference goldens pin.
//
// One candidate (a, p, b) = b's K-base prefix occurs at a[p..p+K) (the same anchors as the exact path).  Let
// x = a[p:] (length rem) and y = b (length lb).  D[i][j] = edit distance (unit costs) of x[:i] and y[:j] restricted to
// the band |j - i| <= W:
//   A (suffix-prefix)  costA = min over j of D[rem][j] <= max_diff   -> row (a, b, p, la, 0, j*)
//   B (containment)    costB = min over i of D[i][lb]  <= max_diff   -> row (a, b, p, p + i*, 0, lb)
// ties go to the cell closest to the main diagonal, then to the smaller coordinate.
//
// Mapping to the machine: ONE WAVE PER CANDIDATE, one lane per diagonal (lane k owns delta = j - i = k - W, so a
// band of up to 63 diagonals fills the wave), swept antidiagonal by antidiagonal: at step d = i + j the lanes with
// (d + W - k) even hold the cells of that antidiagonal and compute
//     D[i][j] = min(D[i-1][j-1] + (x[i-1] != y[j-1]),  D[i-1][j] + 1,  D[i][j-1] + 1)
// from their own value two steps back and their neighbours' values of the previous step -- lane k+1 (up) and lane
// k-1 (left), fetched with whole-wave DPP shifts (wave_shl:1 / wave_shr:1), no LDS, no ds_bpermute.  The bases flow
// through the lanes systolically: x enters at lane 0 and moves one lane up per step, y enters at lane 63 and moves
// one lane down, so every lane sees exactly the pair (x[i-1], y[j-1]) of its cell without any per-lane addressing.
// The sequences themselves are staged through LDS tiles of 64 dwords per wave and side (coalesced 256-byte loads
// of the packed reads; the feed words are picked up from there by the whole wave, uniformly).  Every 64 steps the
// band minimum is reduced across the wave (DPP row_shr / row_bcast) and a candidate whose whole band is above
// max_diff stops: an exact-mode mismatch (max_diff = 0) ends within 32 bases.
//
// Integer DP on 2-bit (or 8-bit) codes: no MFMA anywhere.
#pragma once
#include "kernels.hip.h"

namespace po {

struct ExtArgs {
    const uint64_t* words;
    const uint64_t* woff;
    const uint32_t* len;
    const uint32_t* cand_a;
    const uint32_t* cand_p;
    const uint32_t* cand_b;
    uint32_t n_cand;
    uint32_t max_diff;   // E
    uint32_t band;       // W: diagonals -W..W (W <= 31)
    uint32_t paired;     // strand-mirror mode (only with max_diff == 0): keep_bits decides what this candidate may give
    const uint32_t* exc_off;   // exception records (2-bit reads, max_diff == 0 only)
    const uint32_t* exc_pos;
    const uint8_t* exc_byte;
    uint8_t* type;       // out: bit0 A accepted, bit1 B accepted
    uint32_t* end_a;     // out: bend of the A row
    uint32_t* end_b;     // out: aend - p of the B row
    unsigned long long* counters;  // [0] DP steps executed (antidiagonals), [1] candidates stopped early
};

constexpr uint32_t EXT_INF = 1u << 24;
constexpr int EXT_TILE = 64;  // dwords per LDS tile

__device__ __forceinline__ uint32_t dpp_from_lower(uint32_t fill, uint32_t v) {   // lane k <- lane k-1; lane 0 <- fill
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t dpp_from_upper(uint32_t fill, uint32_t v) {   // lane k <- lane k+1; lane 63 <- fill
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

// wave minimum through the DPP scan network (same shape as wave_incl_scan)
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}

// One side's packed sequence, staged through an LDS tile of EXT_TILE dwords.  All state is wave-uniform.
template <int BITS>
struct SeqFeed {
    const uint32_t* g;     // the read's packed dwords in global memory
    uint32_t* tile;        // this wave's LDS tile
    uint32_t first;        // first base of the sequence inside the read (p for a, 0 for b)
    uint32_t n;            // bases in the sequence (rem / lb)
    uint32_t tile_dw;      // dword index (inside the read) of tile[0]
    uint32_t cur_dw;       // dword the feed word below was read from (~0u: none yet)
    uint32_t cur_w;
    uint32_t sentinel;

    __device__ __forceinline__ void stage(uint32_t dw0) {  // whole wave: tile <- dwords [dw0, dw0 + 64) of the read
        wave_lds_fence();
        tile[lane_id()] = g[dw0 + lane_id()];   // (reads past the read's end land in guard words / the next read: never used)
        tile_dw = dw0;
        wave_lds_fence();
    }
    // base q of the sequence (uniform q, non-decreasing from call to call); sentinel outside [0, n)
    __device__ __forceinline__ uint32_t base(int32_t q) {
        if (q < 0 || (uint32_t)q >= n) return sentinel;
        const uint32_t bitpos = (first + (uint32_t)q) * BITS;
        const uint32_t dw = bitpos >> 5;
        if (dw != cur_dw) {   // a new feed word every 32 / BITS bases: one LDS read for the whole wave
            while (dw - tile_dw >= (uint32_t)EXT_TILE) stage(tile_dw + EXT_TILE);
            cur_w = __builtin_amdgcn_readfirstlane(tile[dw - tile_dw]);
            cur_dw = dw;
        }
        return (cur_w >> (bitpos & 31u)) & ((1u << BITS) - 1u);
    }
    // the same for a per-lane index (initial register fill only; the index stays inside the first tile)
    __device__ __forceinline__ uint32_t base_lane(int32_t q) {
        if (q < 0 || (uint32_t)q >= n) return sentinel;
        const uint32_t bitpos = (first + (uint32_t)q) * BITS;
        const uint32_t dw = bitpos >> 5;
        return (tile[dw - tile_dw] >> (bitpos & 31u)) & ((1u << BITS) - 1u);
    }
};

template <int BITS>
__global__ __launch_bounds__(256) void k_extend_dp(const ExtArgs A) {
    __shared__ uint32_t s_tiles[256 / WAVE][2][EXT_TILE];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t c = __builtin_amdgcn_readfirstlane(blockIdx.x * (256 / WAVE) + wv);
    if (c >= A.n_cand) return;  // whole wave
    const uint32_t a = A.cand_a[c], p = A.cand_p[c], b = A.cand_b[c];
    const uint32_t la = A.len[a], lb = A.len[b];
    const uint32_t rem = la - p;
    const uint32_t E = A.max_diff, W = E ? A.band : 0u;
    // which rows can this candidate give?
    bool canA = rem <= lb + W, canB = lb <= rem + W;
    if (A.paired) {   // (exact mode only) the member of the strand-mirror pair this library computes
        const uint32_t k = keep_bits(a, b, rem, lb, A.paired);
        canA = canA && (k & 1u);
        canB = canB && (k & 2u);
    }
    if (a == b || (!canA && !canB)) {
        if (lane == 0) A.type[c] = 0;
        return;
    }
    SeqFeed<BITS> fx, fy;
    fx.g = reinterpret_cast<const uint32_t*>(A.words + A.woff[a]);
    fx.tile = s_tiles[wv][0];
    fx.first = p;
    fx.n = rem;
    fx.cur_dw = ~0u;
    fx.cur_w = 0;
    fx.sentinel = 0x100u;
    fy.g = reinterpret_cast<const uint32_t*>(A.words + A.woff[b]);
    fy.tile = s_tiles[wv][1];
    fy.first = 0;
    fy.n = lb;
    fy.cur_dw = ~0u;
    fy.cur_w = 0;
    fy.sentinel = 0x200u;
    fx.stage((p * BITS) >> 5);   // (tiles start at the dword that holds base p: the initial fill stays inside the first one)
    fy.stage(0);

    const int32_t iW = (int32_t)W;
    const int32_t k = (int32_t)lane;
    const bool in_band = lane <= 2u * W;
    // lane parity: lane k holds a cell of antidiagonal d iff (d + W - k) is even
    const bool even_lane = ((iW - k) & 1) == 0;   // commits on even d
    // registers at d = 0: lane k faces x[((W - k) >> 1) - 1] and y[((k - W) >> 1) - 1]  (floor shifts)
    uint32_t xr = fx.base_lane(((iW - k) >> 1) - 1);
    uint32_t yr = fy.base_lane(((k - iW) >> 1) - 1);
    uint32_t H = (k == iW) ? 0u : EXT_INF;       // D[0][0] = 0, everything else starts outside the matrix
    uint32_t endA = EXT_INF, endB = EXT_INF;
    const uint32_t one = 1u;

    // last antidiagonal that can hold a wanted cell
    uint32_t d_end = 0;
    if (canA) d_end = max(d_end, 2u * rem + W);
    if (canB) d_end = max(d_end, 2u * lb + W);
    d_end = min(d_end, rem + lb);
    const int32_t recA0 = canA ? (int32_t)(2u * rem) - iW : 0x7FFFFFFF;  // steps at which row rem is reached (lane k: recA0 + k)
    const int32_t recB1 = canB ? (int32_t)(2u * lb) + iW : -0x7FFFFFFF;  // ... and column lb (lane k: recB1 - k)
    uint32_t steps = 0;
    bool dead = false;
    for (uint32_t d = 1; d <= d_end; ++d) {
        // ---- feeds (uniform): lane 0 takes x[((d + W) >> 1) - 1], lane 63 takes y[((d - W + 63) >> 1) - 1]
        const uint32_t fa = fx.base((int32_t)((d + W) >> 1) - 1);
        const uint32_t fb = fy.base((int32_t)((d + 63u - W) >> 1) - 1);
        xr = dpp_from_lower(fa, xr);
        yr = dpp_from_upper(fb, yr);
        // ---- the cell
        const uint32_t up1 = dpp_from_upper(EXT_INF, H) + one;    // D[i-1][j] + 1   (diagonal delta + 1, previous step)
        const uint32_t left1 = dpp_from_lower(EXT_INF, H) + one;  // D[i][j-1] + 1   (diagonal delta - 1, previous step)
        const uint32_t diag = H + min(xr ^ yr, one);              // D[i-1][j-1] + mismatch (own value, two steps back)
        const uint32_t nw = min(min(diag, up1), min(left1, EXT_INF));
        const bool commit = in_band && (((d & 1u) == 0u) == even_lane);
        H = commit ? nw : H;
        // ---- ends: row rem (A) is reached by lane d - recA0, column lb (B) by lane recB1 - d
        const int32_t tA = (int32_t)d - recA0, tB = recB1 - (int32_t)d;
        if (tA >= 0 && tA <= 2 * iW && k == tA) endA = H;
        if (tB >= 0 && tB <= 2 * iW && k == tB) endB = H;
        ++steps;
        if ((d & 63u) == 0u) {
            // every future cell is reached through the last two antidiagonals, and costs never fall along a path
            if (wave_min(in_band ? H : EXT_INF) > E) {
                dead = true;
                break;
            }
        }
    }
    // ---- best end per family: smallest cost, then closest to the main diagonal, then the smaller coordinate
    const uint32_t off = (uint32_t)(k >= iW ? k - iW : iW - k);
    const int32_t jA = (int32_t)rem - iW + k;   // column reached on row rem
    const int32_t iB = (int32_t)lb + iW - k;    // row reached on column lb
    const bool okA = canA && in_band && endA <= E && jA >= 1 && jA <= (int32_t)lb;
    const bool okB = canB && in_band && endB <= E && iB >= 1 && iB <= (int32_t)rem;
    const uint32_t keyA = okA ? ((endA << 8) | (off << 1) | (k > iW ? 1u : 0u)) : 0xFFFFFFFFu;
    const uint32_t keyB = okB ? ((endB << 8) | (off << 1) | (k < iW ? 1u : 0u)) : 0xFFFFFFFFu;
    const uint32_t bestA = wave_min(keyA), bestB = wave_min(keyB);
    uint32_t t = 0;
    if (bestA != 0xFFFFFFFFu) {
        t |= 1u;
        if (keyA == bestA) A.end_a[c] = (uint32_t)jA;   // (exactly one lane holds the winning key)
    }
    if (bestB != 0xFFFFFFFFu) {
        t |= 2u;
        if (keyB == bestB) A.end_b[c] = (uint32_t)iB;
    }
    if (lane == 0) {
        // exact mode on 2-bit reads with exception records: the codes matched; the bytes are equal iff the records agree
        if (t && A.exc_off && E == 0 &&
            !exceptions_equal(A.exc_off, A.exc_pos, A.exc_byte, a, p, b, (t & 1u) ? rem : lb))
            t = 0;
        A.type[c] = (uint8_t)t;
        atomicAdd(&A.counters[0], (unsigned long long)steps);
        if (dead) atomicAdd(&A.counters[1], 1ull);
    }
}

// Rows of the inexact mode: ends come from the DP (no strand-mirror shortcut: every candidate was extended itself).
__global__ __launch_bounds__(256) void k_emit_ex(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                                                 const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                                 const uint32_t* __restrict__ end_a, const uint32_t* __restrict__ end_b,
                                                 const uint32_t* __restrict__ row_off, uint32_t n_cand,
                                                 const uint32_t* __restrict__ len, Row* __restrict__ rows, uint32_t bits,
                                                 unsigned long long* __restrict__ counters) {
    uint64_t nver = 0, suml = 0, sumb = 0, sume = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += gridDim.x * blockDim.x) {
        const uint32_t t = type[i];
        if (t == 0) continue;
        nver += 1;
        const uint32_t a = cand_a[i], p = cand_p[i], b = cand_b[i];
        const uint32_t la = len[a], lb = len[b];
        uint32_t off = row_off[i];
        if (t & 1u) {
            const uint32_t l = end_a[i];
            rows[off++] = Row{a, b, (int32_t)p, (int32_t)la, 0, (int32_t)l};
            suml += l;
            sumb += packed_bytes(l, bits) + packed_bytes(la - p, bits);
        }
        if (t & 2u) {
            const uint32_t n = end_b[i];
            rows[off++] = Row{a, b, (int32_t)p, (int32_t)(p + n), 0, (int32_t)lb};
            suml += lb;
            sumb += packed_bytes(lb, bits) + packed_bytes(n, bits);
        }
        sume += packed_bytes(la - p < lb ? la - p : lb, bits) * 2ull;
    }
    flush_counters(nver, suml, sumb, sume, counters);
}

}  // namespace po
