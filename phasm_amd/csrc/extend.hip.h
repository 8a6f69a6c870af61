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
// Mapping to the machine: ONE WAVE PER CANDIDATE, one lane per diagonal (lane k owns delta = j - i = k - 1 - W, so a
// band of up to 61 diagonals sits on lanes 1..61), swept antidiagonal by antidiagonal: at step d = i + j the lanes with
// (d - delta) even hold the cells of that antidiagonal and compute
//     D[i][j] = min(D[i-1][j-1] + (x[i-1] != y[j-1]),  D[i-1][j] + 1,  D[i][j-1] + 1)
// from their own value two steps back and their neighbours' values of the previous step -- lane k+1 (up) and lane
// k-1 (left), fetched with whole-wave DPP shifts (wave_shl:1 / wave_shr:1), no LDS, no ds_bpermute.  The bases flow
// through the lanes systolically: x enters at lane 0 and moves one lane up per step, y enters at lane 63 and moves
// one lane down, so every lane sees exactly the pair (x[i-1], y[j-1]) of its cell without any per-lane addressing.
// The steady state runs in blocks of 32 antidiagonals fed from two 32-bit scalar windows (16 bases of x, 16 of y):
// per antidiagonal ~11 vector instructions and 3 scalar ones; the first and last steps of a candidate (sentinels
// past the sequence ends, the cells where row rem / column lb are reached) take a slower general step.
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
    uint32_t band;       // W: diagonals -W..W (W <= 30)
    uint32_t paired;     // strand-mirror mode (only with max_diff == 0): keep_bits decides what this candidate may give
    const uint32_t* exc_off;   // exception records (2-bit reads, max_diff == 0 only)
    const uint32_t* exc_pos;
    const uint8_t* exc_byte;
    uint8_t* type;       // out: bit0 A accepted, bit1 B accepted
    uint32_t* end_a;     // out: bend of the A row
    uint32_t* end_b;     // out: aend - p of the B row
    unsigned long long* counters;  // [0] DP steps executed (antidiagonals), [1] candidates stopped early
};

constexpr uint32_t EXT_INF = 1u << 30;   // (never clamped: the host refuses sweeps of 2^30 antidiagonals or more)
constexpr int EXT_TILE = 64;  // dwords per LDS tile

__device__ __forceinline__ uint32_t dpp_from_lower(uint32_t fill, uint32_t v) {   // lane k <- lane k-1; lane 0 <- fill
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t dpp_from_upper(uint32_t fill, uint32_t v) {   // lane k <- lane k+1; lane 63 <- fill
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

// a uniform value enters the wave at one lane (v_writelane_b32: SGPR -> one lane of a VGPR)
#define door(vreg, sval, LANE) asm("v_writelane_b32 %0, %1, " #LANE : "+v"(vreg) : "s"(sval))

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

    // Lane layout: the band's 2W+1 diagonals sit on lanes 1 .. 2W+1 (lane kk owns delta = j - i = kk - 1 - W); lanes
    // 0 and 63 are the doors the bases come in through and never hold a cell, so a neighbour read that runs off the
    // wave (DPP bound_ctrl: reads 0) only ever lands in a lane whose value is not used.
    const int32_t iW = (int32_t)W;
    const int32_t dlt = (int32_t)lane - 1 - iW;
    const bool in_band = lane >= 1u && lane <= 2u * W + 1u;
    // lane kk holds a cell of antidiagonal d iff (d - delta) is even
    const bool commit_even = in_band && (dlt & 1) == 0;   // ... on even d
    const bool commit_odd = in_band && (dlt & 1) != 0;    // ... on odd d
    // registers at d = 0 (what the doors would have let in at steps -kk and kk - 63)
    uint32_t xr = fx.base_lane(((-dlt) >> 1) - 1);
    uint32_t yr = fy.base_lane((dlt >> 1) - 1);
    uint32_t H = (in_band && dlt == 0) ? 0u : EXT_INF;    // D[0][0] = 0, everything else starts outside the matrix
    uint32_t endA = EXT_INF, endB = EXT_INF;
    const uint32_t one = 1u;
    constexpr uint32_t BMASK = (1u << BITS) - 1u;
    constexpr int UNITS = 32 / BITS;   // bases per 32-bit feed window

    // one antidiagonal: neighbours of the previous step through whole-wave DPP shifts, own value from two steps back
    auto cell = [&](bool commit) __attribute__((always_inline)) {
#ifdef PO_DP_PLAIN_CELL
        const uint32_t up1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H, 0x130, 0xf, 0xf, true) + one;    // D[i-1][j] + 1: lane kk+1
        const uint32_t left1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H, 0x138, 0xf, 0xf, true) + one;  // D[i][j-1] + 1: lane kk-1
        const uint32_t diag = H + min(xr ^ yr, one);                                                          // D[i-1][j-1] + mismatch
        const uint32_t nw = min(min(diag, up1), min(left1, EXT_INF));
        H = commit ? nw : H;
#else
        // the same, spelled as the instructions it should be (tools/ubench.hip measured what each costs on this chip:
        // v_add/v_xor/v_and/v_or/v_mov/v_lshrrev/v_bitop3 issue in 2.1-2.4 cycles, every DPP form, v_min/v_max, v_cmp,
        // v_addc and the VOP3 forms in 4.1, and v_cndmask through VCC in 20): the neighbour reads are DPP operands of
        // the adds, the commit is ONE v_bitop3 select against a per-lane all-ones / all-zeros mask instead of a
        // v_cndmask, and the clamp at INF is gone (INF + the longest sweep stays below 2^32, checked on the host).
        // H was last written eight or more instructions ago: no DPP read-after-write wait states needed.
        const uint32_t m = commit ? 0xFFFFFFFFu : 0u;
        uint32_t t0, t1, t2;
        asm volatile(
            "v_add_u32_dpp %[up], %[H], %[one] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   // D[i-1][j] + 1: lane kk+1
            "v_add_u32_dpp %[lf], %[H], %[one] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   // D[i][j-1] + 1: lane kk-1
            "v_cmp_ne_u32 vcc, %[x], %[y]\n\t"
            "v_addc_co_u32 %[dg], vcc, 0, %[H], vcc\n\t"                                               // D[i-1][j-1] + mismatch
            "v_min_u32 %[up], %[up], %[lf]\n\t"
            "v_min_u32 %[up], %[up], %[dg]\n\t"
            "v_bitop3_b32 %[H], %[up], %[m], %[H] bitop3:0xe2"                                         // (m & new) | (~m & H)   [src0 = 0xf0, src1 = 0xcc, src2 = 0xaa]
            : [H] "+v"(H), [up] "=&v"(t0), [lf] "=&v"(t1), [dg] "=&v"(t2)
            : [one] "v"(one), [x] "v"(xr), [y] "v"(yr), [m] "v"(m)
            : "vcc");
#endif
    };
    auto shift_x = [&]() __attribute__((always_inline)) {   // lane kk <- lane kk-1; lane 0 keeps its base
        xr = (uint32_t)__builtin_amdgcn_update_dpp((int)xr, (int)xr, 0x138, 0xf, 0xf, false);
    };
    auto shift_y = [&]() __attribute__((always_inline)) {   // lane kk <- lane kk+1; lane 63 keeps its base
        yr = (uint32_t)__builtin_amdgcn_update_dpp((int)yr, (int)yr, 0x130, 0xf, 0xf, false);
    };

    // last antidiagonal that can hold a wanted cell, first one that reaches row rem (A) / column lb (B)
    uint32_t d_end = 0;
    if (canA) d_end = max(d_end, 2u * rem + W);
    if (canB) d_end = max(d_end, 2u * lb + W);
    d_end = min(d_end, rem + lb);
    const uint32_t first_rec = min(canA ? 2u * rem - min(W, 2u * rem) : 0xFFFFFFFFu, canB ? 2u * lb - min(W, 2u * lb) : 0xFFFFFFFFu);
    uint32_t steps = 0;
    bool dead = false;
    uint32_t d = 1;
    // slow step: feeds addressed from scratch (sentinels past the ends), ends recorded.  Door feeds at step d:
    // lane 0 takes x[((d + W + 1) >> 1) - 1], lane 63 takes y[((d + 62 - W) >> 1) - 1]
    auto slow_step = [&]() __attribute__((always_inline)) {
        const uint32_t fa = fx.base((int32_t)((d + W + 1u) >> 1) - 1);
        const uint32_t fb = fy.base((int32_t)((d + 62u - W) >> 1) - 1);
        shift_x();
        door(xr, fa, 0);
        shift_y();
        door(yr, fb, 63);
        cell((d & 1u) ? commit_odd : commit_even);
        // ends: row rem is reached on diagonal d - 2 rem, column lb on diagonal 2 lb - d
        if (canA && in_band && dlt == (int32_t)d - (int32_t)(2u * rem)) endA = H;
        if (canB && in_band && dlt == (int32_t)(2u * lb) - (int32_t)d) endB = H;
        ++d;
        ++steps;
    };
    // every future cell is reached through the last two antidiagonals, and costs never fall along a path
    auto band_dead = [&]() __attribute__((always_inline)) -> bool { return wave_min(in_band ? H : EXT_INF) > E; };

    // ---- head: up to the first step at which x advances ((d + W + 1) even)
    while (d <= d_end && ((d + W + 1u) & 1u)) slow_step();
    // ---- main phase: blocks of 2 * UNITS steps fed from two 32-bit windows (UNITS bases of x, UNITS of y), no
    // sentinels and no ends inside a block.  In a unit of two steps x advances first, then y.
    {
        const bool c_first = (d & 1u) ? commit_odd : commit_even, c_second = (d & 1u) ? commit_even : commit_odd;
        auto window = [&](SeqFeed<BITS>& f, uint32_t q0) __attribute__((always_inline)) -> uint32_t {
            const uint32_t bitpos = (f.first + q0) * BITS;
            const uint32_t dw = bitpos >> 5, sh = bitpos & 31u;
            if (dw + 1u - f.tile_dw >= (uint32_t)EXT_TILE) f.stage(dw);   // (tile must hold dw and dw + 1)
            const uint32_t lo = __builtin_amdgcn_readfirstlane(f.tile[dw - f.tile_dw]);
            const uint32_t hi = __builtin_amdgcn_readfirstlane(f.tile[dw + 1u - f.tile_dw]);
            return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
        };
        for (;;) {
            const uint32_t qx0 = ((d + W + 1u) >> 1) - 1u, qy0 = ((d + 1u + 62u - W) >> 1) - 1u;  // first bases this block lets in
            const uint32_t last = d + 2u * UNITS - 1u;
            if (last > d_end || last >= first_rec || qx0 + UNITS > rem || qy0 + UNITS > lb) break;
            uint32_t wx = window(fx, qx0), wy = window(fy, qy0);
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                const uint32_t sx = wx & BMASK;
                wx >>= BITS;
                shift_x();
                door(xr, sx, 0);
                shift_y();
                cell(c_first);
                const uint32_t sy = wy & BMASK;
                wy >>= BITS;
                shift_x();
                shift_y();
                door(yr, sy, 63);
                cell(c_second);
            }
            d += 2u * UNITS;
            steps += 2u * UNITS;
            if (band_dead()) {
                dead = true;
                break;
            }
        }
        fx.cur_dw = ~0u;   // (the slow feeds start again from whatever tile is staged)
        fy.cur_dw = ~0u;
    }
    // ---- tail: the last steps, with sentinels and ends
    while (!dead && d <= d_end) {
        slow_step();
        if ((d & 63u) == 0u && band_dead()) dead = true;
    }
    // ---- best end per family: smallest cost, then closest to the main diagonal, then the smaller coordinate
    const uint32_t off = (uint32_t)(dlt >= 0 ? dlt : -dlt);
    const int32_t jA = (int32_t)rem + dlt;   // column reached on row rem
    const int32_t iB = (int32_t)lb - dlt;    // row reached on column lb
    const bool okA = canA && in_band && endA <= E && jA >= 1 && jA <= (int32_t)lb;
    const bool okB = canB && in_band && endB <= E && iB >= 1 && iB <= (int32_t)rem;
    const uint32_t keyA = okA ? ((endA << 8) | (off << 1) | (dlt > 0 ? 1u : 0u)) : 0xFFFFFFFFu;
    const uint32_t keyB = okB ? ((endB << 8) | (off << 1) | (dlt < 0 ? 1u : 0u)) : 0xFFFFFFFFu;
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

// ----------------------------------------------------------------------------------------------------------------
// The same DP with the parallelism turned round: ONE LANE PER CANDIDATE, 64 candidates per wave.
//
// k_extend_dp spreads one candidate over a wave (a lane per diagonal).  That is the mapping for few, long, wide
// alignments; an overlap job has MILLIONS of candidates and narrow bands (config 4: 6.9 M candidates, W = 8, 17
// diagonals -> 45 of 64 lanes idle, and every step pays four DPP exchanges).  Here a lane keeps its candidate's whole
// band row D[i][i-W .. i+W] in registers (2W + 1 VGPRs, W a template bound) and walks the rows serially:
//     D[i][j] = min(D[i-1][j-1] + (x[i-1] != y[j-1]),  min(D[i-1][j], D[i][j-1]) + 1)
// no cross-lane traffic at all.  x is a 16-base register window; y is a sliding (2W+1)-base window kept packed in 64
// bits, and one row's 2W+1 base comparisons are three bitwise ops on that window (x replicated into every 2-bit
// field, XOR, fold the two bits of each field).  Per cell: one bit extract, two adds, two mins.  Candidates of a wave
// are neighbours in the a-major candidate order (similar lengths); a lane whose whole band is above max_diff stops,
// the wave ends with its longest candidate.  2-bit reads, W <= 15 (the y window is 64 bits).  Same results as
// k_extend_dp, bit for bit (tests run both); ~12 x its throughput at config 4.
// rows the lane kernel will walk for candidate c (its sort key: a wave's 64 lanes should finish together)
__global__ void k_dp_rows(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                          const uint32_t* __restrict__ cand_b, const uint32_t* __restrict__ len, uint32_t n_cand,
                          uint32_t band, uint32_t* __restrict__ rows) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cand) return;
    const uint32_t rem = len[cand_a[c]] - cand_p[c], lb = len[cand_b[c]];
    rows[c] = min(rem, lb + band);
}

template <int WB>
__global__ __launch_bounds__(256) void k_extend_lanes(const ExtArgs A, const uint32_t* __restrict__ perm) {
    constexpr int NB = 2 * WB + 1;   // band cells kept per lane (the call's band W <= WB uses the middle 2W + 1)
    const uint32_t c0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = c0 < A.n_cand;
    // Candidates come a-major with ascending p: the 64 consecutive candidates of a wave would span one read's whole
    // list, overlap lengths from 15 kb down to 1 kb, and the wave would run as long as its longest lane (53 % of the
    // lane-rows idle at config 4).  `perm` lists the candidates by the number of rows they need (counting sort,
    // k_dp_rows + k_read_sort): a wave's lanes then finish together.
    const uint32_t c = in_range ? (perm ? perm[c0] : c0) : 0u;   // (lanes past the end walk candidate 0 with nothing to
                                                                 //  do: no early return, the wave sums at the end need every lane)
    const uint32_t a = A.cand_a[c], p = A.cand_p[c], b = A.cand_b[c];
    const uint32_t la = A.len[a], lb = A.len[b];
    const uint32_t rem = la - p;
    const uint32_t E = A.max_diff;
    // (readfirstlane: the band tests below are then scalar compares and branches, not lane masks)
    const int32_t W = (int32_t)__builtin_amdgcn_readfirstlane(E ? A.band : 0u);
    bool canA = rem <= lb + (uint32_t)W, canB = lb <= rem + (uint32_t)W;
    if (A.paired) {
        const uint32_t kb = keep_bits(a, b, rem, lb, A.paired);
        canA = canA && (kb & 1u);
        canB = canB && (kb & 2u);
    }
    if (!in_range || a == b) canA = canB = false;
    const uint32_t* __restrict__ gx = reinterpret_cast<const uint32_t*>(A.words + A.woff[a]);
    const uint32_t* __restrict__ gy = reinterpret_cast<const uint32_t*>(A.words + A.woff[b]);
    // base q of y (0 outside [0, lb): never compared where it matters -- cells beyond column lb feed nothing wanted)
    auto ybase = [&](int32_t q) __attribute__((always_inline)) -> uint32_t {
        if (q < 0 || (uint32_t)q >= lb) return 0u;
        return (gy[(uint32_t)q >> 4] >> (((uint32_t)q & 15u) * 2u)) & 3u;
    };
    // D[k] = D[i][i - WB + k] for the current row i; cells outside the call's band or the matrix hold INF
    uint32_t D[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int32_t j = k - WB;   // row 0: D[0][j] = j inside the band
        D[k] = (j >= 0 && j <= W && (uint32_t)j <= lb) ? (uint32_t)j : EXT_INF;
    }
    // y window for row i: bases y[i - WB - 1 + k], k = 0 .. 2 WB, two bits each, in (ylo, yhi); row 1 first
    uint32_t ylo = 0, yhi = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t v = ybase(1 - WB - 1 + k);
        if (k < 16) ylo |= v << (2 * k); else yhi |= v << (2 * (k - 16));
    }
    uint32_t ob[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) ob[k] = __builtin_amdgcn_readfirstlane((k - WB < -W || k - WB > W) ? EXT_INF : 0u);
    const uint32_t rows = canA ? rem : (canB ? min(rem, lb + (uint32_t)W) : 0u);   // last row that holds a wanted cell
    uint32_t bestA = 0xFFFFFFFFu, bestB = 0xFFFFFFFFu, endA_j = 0, endB_i = 0;
    uint32_t xw = 0, yw = 0;
    uint32_t steps = 0;
    bool dead = false;
    for (uint32_t i = 1; i <= rows; ++i) {
        // ---- x[i-1] and the base that enters the y window after this row, y[i + WB]: 16-base register windows,
        // refilled from the packed reads every 16 rows (past a read's end come guard words / the next read: a base
        // beyond column lb only ever reaches cells beyond column lb, which feed nothing that is wanted)
        const uint32_t xpos = p + i - 1u, ypos = i + (uint32_t)WB;
        if (i == 1u || (xpos & 15u) == 0u) xw = gx[xpos >> 4] >> ((xpos & 15u) * 2u);
        if (i == 1u || (ypos & 15u) == 0u) yw = gy[ypos >> 4] >> ((ypos & 15u) * 2u);
        const uint32_t xb = xw & 3u;
        xw >>= 2;
        // ---- all 2 WB + 1 comparisons of the row at once: bit 2k of nm = (x[i-1] != y[i - WB - 1 + k])
        const uint32_t xrep = ((xb & 1u) ? 0x55555555u : 0u) | ((xb & 2u) ? 0xAAAAAAAAu : 0u);
        const uint32_t tl = ylo ^ xrep, th = yhi ^ xrep;
        const uint32_t nml = (tl | (tl >> 1)) & 0x55555555u, nmh = (th | (th >> 1)) & 0x55555555u;
        // ---- the row, left to right (the left neighbour is this row's previous cell).  Diagonals outside the call's
        // band are pinned at INF by OR-ing a per-diagonal scalar (branches on the band made hipcc rebuild lane masks in
        // every cell); cells left of the matrix (j < 0) need no test: everything they are computed from is INF.
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const uint32_t mism = k < 16 ? (nml >> (2 * k)) & 1u : (nmh >> (2 * (k - 16))) & 1u;
            const uint32_t up = k + 1 < NB ? D[k + 1] : EXT_INF;
            const uint32_t left = k > 0 ? D[k - 1] : EXT_INF;   // (this row's value already)
            D[k] = min(D[k] + mism, min(up, left) + 1u) | ob[k];   // ob: INF on the diagonals outside the call's band
        }
        ++steps;
        // ---- ends.  B: column lb is cell k = lb - i + WB of this row (rows lb - W .. lb + W)
        if (canB) {
            const int32_t kb = (int32_t)lb - (int32_t)i + WB;
            if (kb >= WB - W && kb <= WB + W) {
                uint32_t v = EXT_INF;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (k == kb) v = D[k];
                const int32_t dl = kb - WB;                 // delta = lb - i
                if (v <= E) {
                    const uint32_t key = (v << 8) | ((uint32_t)(dl >= 0 ? dl : -dl) << 1) | (dl < 0 ? 1u : 0u);
                    if (key < bestB) {
                        bestB = key;
                        endB_i = i;
                    }
                }
            }
        }
        if (canA && i == rem) {   // A: row rem, columns 1 .. lb
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int32_t dl = k - WB;
                const int64_t j = (int64_t)rem + dl;
                if (dl >= -W && dl <= W && j >= 1 && j <= (int64_t)lb && D[k] <= E) {
                    const uint32_t key = (D[k] << 8) | ((uint32_t)(dl >= 0 ? dl : -dl) << 1) | (dl > 0 ? 1u : 0u);
                    if (key < bestA) {
                        bestA = key;
                        endA_j = (uint32_t)j;
                    }
                }
            }
        }
        // ---- every 16 rows: nothing in the band at or below max_diff means no later cell can be (costs never fall
        // along a path, and every path to a later row crosses this one)
        if ((i & 15u) == 0u) {
            uint32_t rowmin = EXT_INF;
#pragma unroll
            for (int k = 0; k < NB; ++k) rowmin = min(rowmin, D[k]);
            if (rowmin > E) {
                dead = i < rows;
                break;
            }
        }
        // ---- slide the y window: drop y[i - WB - 1], take in y[i + WB]
        ylo = (ylo >> 2) | (yhi << 30);
        yhi >>= 2;
        const uint32_t nv = yw & 3u;
        yw >>= 2;
        if (2 * WB < 16) ylo |= nv << (2 * (2 * WB)); else yhi |= nv << (2 * (2 * WB - 16));
    }
    uint32_t t = 0;
    if (bestA != 0xFFFFFFFFu) {   // (only lanes with a candidate of their own ever get here with a best end)
        t |= 1u;
        A.end_a[c] = endA_j;
    }
    if (bestB != 0xFFFFFFFFu) {
        t |= 2u;
        A.end_b[c] = endB_i;
    }
    if (t && A.exc_off && E == 0 && !exceptions_equal(A.exc_off, A.exc_pos, A.exc_byte, a, p, b, (t & 1u) ? rem : lb)) t = 0;
    if (in_range) A.type[c] = (uint8_t)t;
    // (counters: rows walked, candidates stopped early -- one atomic per wave)
    const uint64_t ws = wave_sum64(2ull * steps);   // two antidiagonals per row: comparable with k_extend_dp's count
    const uint32_t wd = wave_sum(dead ? 1u : 0u);
    if (lane_id() == 0) {
        atomicAdd(&A.counters[0], (unsigned long long)ws);
        if (wd) atomicAdd(&A.counters[1], (unsigned long long)wd);
    }
}

// ----------------------------------------------------------------------------------------
// Mapping (3): one lane per candidate, the band row as a BIT VECTOR (Myers 1999 / Hyyro 2003, diagonal band).
//
// k_extend_lanes spends six instructions on each of the 2W + 1 cells of a row (102 per row at W = 8).  The cells of a
// row differ from their neighbours by -1, 0 or +1, so the whole row is two bit masks -- HP / HN: bit k set iff
// C[k] - C[k-1] = +1 / -1, C[k] = D[i][i - W + k] -- plus ONE number, the main-diagonal cell S = D[i][i].  One row of
// the DP is then ~20 vector instructions whatever the band (W <= 15: 31 bits):
//   the previous row seen through the window shifted by one diagonal step: P'[k] = C_prev[k + 1], so its deltas are the
//   previous row's, shifted right by one (the cell beyond the band's top edge counts as one higher: never the minimum);
//   Eq = match bits of x[i-1] against the 2W + 1 bases of y facing the band (two bit planes of y slide along);
//   D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN      -- bit k: D[i][j] == D[i-1][j-1]   (the carry runs along the row)
//   vertical deltas hp = VN | ~(D0 | VP), hn = D0 & VP; this row's HP = (hn << 1) | ~(D0 | (hp << 1 | 1)),
//   HN = D0 & (hp << 1 | 1)   (the cell below the band's bottom edge counts as one higher as well);  S += 1 - D0[W].
// Cells left of the matrix (j < 0, the first W rows) are the virtual cells D[i][j] = i + |j| with every comparison a
// mismatch: consistent with the recurrence and never on a best path.  A cell's value, where one is wanted -- the ends, the
// band minimum every 16 rows -- is S plus the deltas between: popcounts.  Same rows, bit for bit, as the other two
// mappings (tests/test_gpu_extend.py runs all three against oracle/extend_oracle.c; the prototype of this recurrence
// was held against the plain DP on 40 000 random cases first).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_extend_bits(const ExtArgs A, const uint32_t* __restrict__ perm) {
    const uint32_t c0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = c0 < A.n_cand;
    const uint32_t c = in_range ? (perm ? perm[c0] : c0) : 0u;
    const uint32_t a = A.cand_a[c], p = A.cand_p[c], b = A.cand_b[c];
    const uint32_t la = A.len[a], lb = A.len[b];
    const uint32_t rem = la - p;
    const uint32_t E = A.max_diff;
    const uint32_t W = __builtin_amdgcn_readfirstlane(E ? A.band : 0u);   // (wave-uniform: shifts by W are scalar operands)
    const uint32_t N = 2u * W + 1u;
    const uint32_t ALL = (1u << N) - 1u, TOP = 1u << (N - 1u);
    bool canA = rem <= lb + W, canB = lb <= rem + W;
    if (A.paired) {
        const uint32_t kb = keep_bits(a, b, rem, lb, A.paired);
        canA = canA && (kb & 1u);
        canB = canB && (kb & 2u);
    }
    if (!in_range || a == b) canA = canB = false;
    const uint32_t* __restrict__ gx = reinterpret_cast<const uint32_t*>(A.words + A.woff[a]);
    const uint32_t* __restrict__ gy = reinterpret_cast<const uint32_t*>(A.words + A.woff[b]);
    auto ybase = [&](int32_t q) __attribute__((always_inline)) -> uint32_t {
        if (q < 0 || (uint32_t)q >= lb) return 0u;
        return (gy[(uint32_t)q >> 4] >> (((uint32_t)q & 15u) * 2u)) & 3u;
    };
    // row 0: C[k] = |k - W|
    uint32_t HP = ALL & ~((2u << W) - 1u), HN = ((2u << W) - 1u) & ~1u;   // bits W+1 .. N-1 / bits 1 .. W
    uint32_t S = 0;
    // the two bit planes of the y window of row 1: bit k = base y[k - W]
    uint32_t plo = 0, phi = 0;
    for (uint32_t k = 0; k < N; ++k) {
        const uint32_t v = ybase((int32_t)k - (int32_t)W);
        plo |= (v & 1u) << k;
        phi |= (v >> 1) << k;
    }
    uint32_t vm = ALL & ~((1u << W) - 1u);   // row 1: columns j >= 1 are the cells k >= W
    const uint32_t rows = canA ? rem : (canB ? min(rem, lb + W) : 0u);
    uint32_t bestA = 0xFFFFFFFFu, bestB = 0xFFFFFFFFu, endA_j = 0, endB_i = 0;
    uint32_t xw = 0, yw = 0, steps = 0;
    bool dead = false, stop = false;
    const uint32_t above = ALL & ~((2u << W) - 1u), below = ((2u << W) - 1u) & ~1u;   // delta bits above / at-and-below the diagonal
    auto cell = [&](uint32_t k) __attribute__((always_inline)) -> uint32_t {   // C[k] from S and the deltas between
        if (k >= W) {
            const uint32_t m = ((2u << k) - 1u) & above;          // bits W+1 .. k
            return S + (uint32_t)__popc(HP & m) - (uint32_t)__popc(HN & m);
        }
        const uint32_t m = below & ~((2u << k) - 1u);             // bits k+1 .. W
        return S - (uint32_t)__popc(HP & m) + (uint32_t)__popc(HN & m);
    };
    auto band_min = [&]() __attribute__((always_inline)) -> uint32_t {   // walk the deltas up and down from the diagonal
        uint32_t rowmin = S, v = S;
        for (uint32_t k = W + 1u; k < N; ++k) {
            v += ((HP >> k) & 1u) - ((HN >> k) & 1u);
            rowmin = min(rowmin, v);
        }
        v = S;
        for (uint32_t k = W; k >= 1u; --k) {
            v -= ((HP >> k) & 1u) - ((HN >> k) & 1u);
            // (cells left of the matrix are virtual -- i + |j| -- and never the smallest of a row that has real cells)
            rowmin = min(rowmin, v);
        }
        return rowmin;
    };
    // One row of the recurrence, with everything a row can need: the end cells of B (rows lb - W .. lb + W) and of A (row
    // rem), the band minimum every 16 rows, validity of the cells left of the matrix (the first W rows).  Its bases come
    // from 4-byte loads as it crosses a dword -- this is the path of a candidate's FIRST rows (until the band is inside
    // the matrix and x stands at a 64-base boundary) and of its LAST ones; everything between runs in fast_block below.
    bool fresh = true;   // (re)load the two base dwords: at the start of a slow phase
    auto slow_row = [&](uint32_t i) __attribute__((always_inline)) {
        const uint32_t xpos = p + i - 1u, ypos = i + W;
        if (fresh || (xpos & 15u) == 0u) xw = gx[xpos >> 4] >> ((xpos & 15u) * 2u);
        if (fresh || (ypos & 15u) == 0u) yw = gy[ypos >> 4] >> ((ypos & 15u) * 2u);
        fresh = false;
        const uint32_t xb = xw & 3u;
        xw >>= 2;
        const uint32_t xl = 0u - (xb & 1u), xh = 0u - (xb >> 1);
        const uint32_t Eq = ~((plo ^ xl) | (phi ^ xh)) & vm;
        const uint32_t VP = (HP >> 1) | TOP, VN = HN >> 1;
        const uint32_t D0 = ((((Eq & VP) + VP) ^ VP) | Eq | VN) & ALL;
        const uint32_t hp = (VN | ~(D0 | VP)) & ALL, hn = D0 & VP;
        const uint32_t hps = ((hp << 1) | 1u) & ALL, hns = (hn << 1) & ALL;
        HP = (hns | ~(D0 | hps)) & ALL;
        HN = D0 & hps;
        S += 1u - ((D0 >> W) & 1u);
        ++steps;
        if (canB && i + W >= lb && i <= lb + W) {   // B: column lb is cell k = lb - i + W of this row
            const uint32_t kb = lb + W - i;
            const uint32_t v = cell(kb);
            const int32_t dl = (int32_t)kb - (int32_t)W;
            if (v <= E) {
                const uint32_t key = (v << 8) | ((uint32_t)(dl >= 0 ? dl : -dl) << 1) | (dl < 0 ? 1u : 0u);
                if (key < bestB) {
                    bestB = key;
                    endB_i = i;
                }
            }
        }
        if (canA && i == rem) {   // A: row rem, columns 1 .. lb
            for (uint32_t k = 0; k < N; ++k) {
                const int32_t dl = (int32_t)k - (int32_t)W;
                const int64_t j = (int64_t)rem + dl;
                if (j >= 1 && j <= (int64_t)lb) {
                    const uint32_t v = cell(k);
                    if (v <= E) {
                        const uint32_t key = (v << 8) | ((uint32_t)(dl >= 0 ? dl : -dl) << 1) | (dl > 0 ? 1u : 0u);
                        if (key < bestA) {
                            bestA = key;
                            endA_j = (uint32_t)j;
                        }
                    }
                }
            }
        }
        if ((i & 15u) == 0u && band_min() > E) {   // every 16 rows: the band minimum
            dead = i < rows;
            stop = true;
        }
        // slide the y window: drop y[i - W - 1], take in y[i + W]; one more column is inside the matrix
        const uint32_t nv = yw & 3u;
        yw >>= 2;
        plo = (plo >> 1) | ((nv & 1u) << (N - 1u));
        phi = (phi >> 1) | ((nv >> 1) << (N - 1u));
        vm |= vm >> 1;
    };
    // Sixteen rows without a branch: x stands at a dword boundary, every cell of the band is inside the matrix (vm = ALL),
    // no row of the block is an end row.  The sixteen bases of y that slide into the window come as two 16-bit planes
    // behind the window's own bits (P, Q: up to 47 bits), so the window of row r is ONE funnel shift of the block's start;
    // the diagonal cell's sixteen match bits are summed in place (bit W of D0) and folded into S once per block.
    const uint32_t DIAG = 1u << W;
    auto even_bits = [](uint32_t v) __attribute__((always_inline)) -> uint32_t {   // bits 0, 2, 4 .. 30 packed into 16
        v &= 0x55555555u;
        v = (v | (v >> 1)) & 0x33333333u;
        v = (v | (v >> 2)) & 0x0F0F0F0Fu;
        v = (v | (v >> 4)) & 0x00FF00FFu;
        return (v | (v >> 8)) & 0x0000FFFFu;
    };
    auto fast_block = [&](const uint32_t xd, const uint32_t yd) __attribute__((always_inline)) {
        const uint32_t ylo = even_bits(yd), yhi = even_bits(yd >> 1);
        const uint32_t Plo = plo | (ylo << N), Phi = ylo >> (32u - N);   // (N <= 31, ylo < 2^16: nothing is lost)
        const uint32_t Qlo = phi | (yhi << N), Qhi = yhi >> (32u - N);
        uint32_t sacc = 0;
        // Inside the block the state is the NEXT row's view of the deltas, VP = (HP >> 1) | TOP and VN = HN >> 1: with
        // hp / hn this row's vertical deltas, HP >> 1 = hn | ~((D0 >> 1) | hp) below the top bit and HN >> 1 =
        // (D0 >> 1) & hp -- one shift of D0 instead of the four shifts of HP, HN, hp and hn (bit 0 of HP / HN, the delta
        // towards the cell below the band, is never read).
        uint32_t VP = (HP >> 1) | TOP, VN = HN >> 1;
        const uint32_t ALLS = ALL >> 1;
#pragma unroll
        for (uint32_t r = 0; r < 16u; ++r) {
            const uint32_t xl = (uint32_t)((int32_t)(xd << (31u - 2u * r)) >> 31);
            const uint32_t xh = (uint32_t)((int32_t)(xd << (30u - 2u * r)) >> 31);
            const uint32_t wl = r ? __builtin_amdgcn_alignbit(Phi, Plo, r) : Plo;
            const uint32_t wh = r ? __builtin_amdgcn_alignbit(Qhi, Qlo, r) : Qlo;
            // (three-input boolean functions spelled as v_bitop3 truth tables -- inputs 0xF0, 0xCC, 0xAA -- because the
            // compiler's own factoring of these expressions came out four instructions per row longer)
            const uint32_t e1 = __builtin_amdgcn_bitop3_b32(wl, xl, ALL, 0x82);        // ~(wl ^ xl) & ALL
            const uint32_t Eq = __builtin_amdgcn_bitop3_b32(e1, wh, xh, 0x90);         // e1 & ~(wh ^ xh)
            const uint32_t t = (Eq & VP) + VP;
            const uint32_t u = __builtin_amdgcn_bitop3_b32(t, VP, Eq, 0xBE);           // (t ^ VP) | Eq
            const uint32_t D0 = __builtin_amdgcn_bitop3_b32(u, VN, ALL, 0xA8);         // (u | VN) & ALL
            const uint32_t hp = __builtin_amdgcn_bitop3_b32(VN, D0, VP, 0xF1);         // VN | ~(D0 | VP)
            const uint32_t hn = D0 & VP;
            const uint32_t D0s = D0 >> 1;
            const uint32_t x = __builtin_amdgcn_bitop3_b32(hn, D0s, hp, 0xF1);         // hn | ~(D0s | hp)
            VP = __builtin_amdgcn_bitop3_b32(x, ALLS, TOP, 0xEA);                      // (x & ALLS) | TOP
            VN = D0s & hp;
            sacc += D0 & DIAG;
        }
        HP = (VP << 1) & ALL;
        HN = (VN << 1) & ALL;
        S += 16u - (sacc >> W);
        plo = __builtin_amdgcn_alignbit(Phi, Plo, 16u) & ALL;
        phi = __builtin_amdgcn_alignbit(Qhi, Qlo, 16u) & ALL;
    };
    uint32_t i = 1;
    // the last row a fast block may contain: before the end rows of B and of A
    uint32_t lim = rows;
    if (canA) lim = min(lim, rem - 1u);
    if (canB) lim = min(lim, lb > W + 1u ? lb - W - 1u : 0u);
    // ---- the first rows: until the band has left the matrix's left edge and x stands at a dword boundary ...
    while (i <= rows && !stop && (i <= W + 1u || ((p + i - 1u) & 15u) != 0u)) {
        slow_row(i);
        ++i;
    }
    // ... then single blocks (their three dwords loaded on the spot) up to a 64-base boundary of x
    auto single_blocks = [&](bool to_boundary) __attribute__((always_inline)) {
        while (!stop && i + 15u <= lim && (!to_boundary || ((p + i - 1u) & 63u) != 0u)) {
            const uint32_t* yq = gy + ((i + W) >> 4);
            fast_block(gx[(p + i - 1u) >> 4], __builtin_amdgcn_alignbit(yq[1], yq[0], ((i + W) & 15u) * 2u));
            i += 16u;
            steps += 16u;
        }
    };
    single_blocks(true);
    // ---- 64 rows at a time: x in one aligned 16-byte piece, the 64 bases of y that slide in as five dwords from wherever
    // they start (4-byte aligned loads), both requested one round ahead
    if (!stop && i + 63u <= lim) {
        const uint32_t* yp = gy + ((i + W) >> 4);
        const uint32_t ysh = ((i + W) & 15u) * 2u;
        const u32x4* xp = reinterpret_cast<const u32x4*>(gx) + ((p + i - 1u) >> 6);
        u32x4 xa = xp[0];
        u32x4 ya = {yp[0], yp[1], yp[2], yp[3]};
        u32x4 yb = {yp[4], yp[5], yp[6], yp[7]};
        do {
            xp += 1;
            yp += 4;
            const u32x4 xn = xp[0];                             // (past the end of a read: the next read, or the buffer's 72 zero words)
            const u32x4 yc = {yp[4], yp[5], yp[6], yp[7]};
            fast_block(xa.x, __builtin_amdgcn_alignbit(ya.y, ya.x, ysh));
            fast_block(xa.y, __builtin_amdgcn_alignbit(ya.z, ya.y, ysh));
            fast_block(xa.z, __builtin_amdgcn_alignbit(ya.w, ya.z, ysh));
            fast_block(xa.w, __builtin_amdgcn_alignbit(yb.x, ya.w, ysh));
            xa = xn;
            ya = yb;
            yb = yc;
            i += 64u;
            steps += 64u;
            if (band_min() > E) {   // (the band minimum never falls from one row to the next: any row may test it)
                dead = i - 1u < rows;
                stop = true;
            }
        } while (!stop && i + 63u <= lim);
    }
    single_blocks(false);
    fresh = true;
    // ---- the last rows, end rows included
    while (i <= rows && !stop) {
        slow_row(i);
        ++i;
    }
    uint32_t t = 0;
    if (bestA != 0xFFFFFFFFu) {
        t |= 1u;
        A.end_a[c] = endA_j;
    }
    if (bestB != 0xFFFFFFFFu) {
        t |= 2u;
        A.end_b[c] = endB_i;
    }
    if (t && A.exc_off && E == 0 && !exceptions_equal(A.exc_off, A.exc_pos, A.exc_byte, a, p, b, (t & 1u) ? rem : lb)) t = 0;
    if (in_range) A.type[c] = (uint8_t)t;
    const uint64_t ws = wave_sum64(2ull * steps);
    const uint32_t wd = wave_sum(dead ? 1u : 0u);
    if (lane_id() == 0) {
        atomicAdd(&A.counters[0], (unsigned long long)ws);
        if (wd) atomicAdd(&A.counters[1], (unsigned long long)wd);
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
