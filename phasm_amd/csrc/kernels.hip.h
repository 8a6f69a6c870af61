// Device code of libphasm_overlap.so -- hand-written HIP for gfx950 (MI355X), wave64.
//
// Computes what ExactOverlapper::overlaps() computes (/root/reference/src/overlapper.cpp:28-150)
// without a suffix tree.  Every reportable row of read b starts with b's first K bases
// (K = min(bases per 64-bit word, min_length)), so:
//
//   index   : open-addressed table {K-mer -> chain of reads whose prefix it is} + Bloom filter
//   scan    : every position p <= la - min_length of every a-side read probes the filter (LDS)
//             and then the table (L2); hits become candidates (a, p, b), a-major, p ascending
//   verify  : packed word compare of a[p : p+n) with b[0 : n), n = min(la-p, lb)
//   select  : "longest only" for suffix-prefix (A) rows; every occurrence for containment (B)
//   emit    : 24-byte rows
//
// Integer/bit work, HBM/L2-bound: no MFMA anywhere.  BITS = 2 (pure ACGT) or 8 (raw bytes; the
// reference compares bytes, so N / lower case must stay distinct).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace po {

// native 4 x u32 vector (one dwordx4 load; stays in VGPRs, unlike HIP's struct-based uint4 when it is
// carried across loop iterations)
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr uint32_t NO_SELFREP = 0xFFFFFFFFu;
constexpr uint64_t KEY_EMPTY = ~0ull;          // slot-claim sentinel; a real all-ones K-mer lives
                                               // in the dedicated extra slot at index 1<<tbits
constexpr int WAVE = 64;

// A piece of a streamed step whose candidate count is PREDICTED (the previous call's) rather than waited for: the
// kernels behind the counting pass are launched without a host round trip, sized for `cap` candidates, and read the real
// number from device memory (the counting pass's prefix-sum total).  Should the count exceed what the buffers hold,
// every one of them does nothing -- the host sees the count at the end of the piece and takes the safe form of the call.
// -DPO_VER_PAD=n / -DPO_SCAN_PAD=n (measurement builds, tools/pad_probe.sh): n extra full-rate VALU instructions per
// 16-byte compare of the verify kernel / per position of the scan filter.  How much a kernel's time grows per added
// VALU cycle says how far it is bound by VALU issue (slope 1) rather than by latency (slope 0).
// -DPO_VER_LDS_SWZ=1 (measurement build, tools/lds_swz_probe.sh): the words of a sit in LDS with ONE PAD DWORD PER 32 --
// physical dword = x + (x >> 5).  A 16-lane group reads a at a stride of four dwords (16 bytes of b per lane), which
// touches 8 of the 32 banks twice; with the pad the upper eight lanes move one bank on and the group's 16 reads hit 16
// banks.  Costs a shift and an add per LDS dword read.
#ifndef PO_VER_LDS_SWZ
#define PO_VER_LDS_SWZ 0
#endif
__host__ __device__ inline uint32_t ver_swz(uint32_t x) { return PO_VER_LDS_SWZ ? x + (x >> 5) : x; }
// 64-bit words of LDS that hold lds_words words of a
__host__ __device__ inline uint32_t ver_a_words(uint32_t lds_words) {
    return PO_VER_LDS_SWZ ? (((2u * lds_words + ((2u * lds_words) >> 5) + 2u) / 2u + 1u) & ~1u) : lds_words;
}
#ifndef PO_VER_PAD
#define PO_VER_PAD 0
#endif
#ifndef PO_SCAN_PAD
#define PO_SCAN_PAD 0
#endif
template <int N>
__device__ __forceinline__ void valu_pad(uint32_t& r, uint32_t c) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c));
}

struct CandGuard {
    const unsigned long long* n_dev;   // nullptr: the host knows the count, no check
    uint32_t cap;
    __device__ __forceinline__ bool overflow() const { return n_dev && *n_dev > (unsigned long long)cap; }
};
#ifndef PO_FILTER_DEPTH
#define PO_FILTER_DEPTH 6
#endif
constexpr int SCAN_BLOCK = 1024;               // 16 waves: one persistent workgroup per CU (<= 128 VGPRs)
constexpr int TILE_WORDS = 64;                 // one 64-bit word per lane

struct __attribute__((aligned(16))) Slot {
    uint64_t key;
    uint32_t start;   // first entry in chain[]  -- or, for a one-read chain, the read itself
    uint32_t count;   // 0 = empty slot; bit 31 set = one-read chain, low 31 bits = that read's length
};
constexpr uint32_t SLOT_SINGLE = 0x80000000u;

struct Row {
    uint32_t a_idx, b_idx;
    int32_t astart, aend, bstart, bend;
};

// One scan tile = TILE_WORDS consecutive words of one read.  32 bytes, built at upload, read with
// scalar loads (the tile index is wave-uniform) so a tile's whole geometry costs one round trip.
struct __attribute__((aligned(32))) TileRec {
    uint64_t wabs;    // index into words[] of the tile's first word
    uint64_t wread;   // index into words[] of the read's first word
    uint32_t read;    // read index
    uint32_t la;      // read length in bases
    uint32_t word0;   // tile's first word relative to the read
    uint32_t pad;
};

// The tile records, written at upload from what the host sends anyway (woff, len, first tile of every read):
// one thread per tile finds its read by binary search over read_tile0 (n_reads + 1 entries, L2-resident).
// Reads without tiles (length 0) share their successor's first tile index; the LAST read whose first tile is
// <= t is the one that owns tile t.
__global__ void k_build_tiles(const uint64_t* __restrict__ woff, const uint32_t* __restrict__ len,
                              const uint32_t* __restrict__ read_tile0, uint32_t n_reads, uint32_t n_tiles,
                              TileRec* __restrict__ tiles) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    uint32_t lo = 0, hi = n_reads;  // read_tile0[lo] <= t < read_tile0[hi]  (read_tile0[n_reads] = n_tiles)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (read_tile0[mid] <= t) lo = mid; else hi = mid;
    }
    TileRec rec;
    rec.word0 = (t - read_tile0[lo]) * (uint32_t)TILE_WORDS;
    rec.wread = woff[lo];
    rec.wabs = rec.wread + rec.word0;
    rec.read = lo;
    rec.la = len[lo];
    rec.pad = 0;
    tiles[t] = rec;
}

// woff arrives relative to each read's host store (even reads: store 0, odd reads: store 1); on the device the
// stores sit one behind the other
__global__ void k_abs_woff(uint64_t* __restrict__ woff, uint32_t n_reads, uint64_t base1) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads && (r & 1u)) woff[r] += base1;
}

// word-by-word difference count of two buffers (PHASM_VERIFY_GENERATED: generated store 1 against the host's)
__global__ void k_count_diff(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t n,
                             unsigned long long* __restrict__ n_diff) {
    unsigned long long d = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        d += a[i] != b[i];
    if (d) atomicAdd(n_diff, d);
}

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
__host__ __device__ inline void kmer_hash(uint64_t k, uint32_t& h1, uint32_t& h2) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = lo ^ ((hi << 13) | (hi >> 19));
    h1 = x * 0x9E3779B1u;
    h2 = (x ^ (x >> 15) ^ (hi * 5u)) * 0x85EBCA77u;
}

// Blocked Bloom filter, 32-bit blocks, 3 bits per key, one multiply and ONE LDS read per probe.
__host__ __device__ inline void bloom_slot(uint64_t k, uint32_t bloom_log2, uint32_t& word, uint32_t& mask) {
    uint32_t h1, h2;
    kmer_hash(k, h1, h2);
    (void)h2;
    word = h1 >> (32 - (bloom_log2 - 5));
    mask = (1u << ((h1 >> 7) & 31)) | (1u << ((h1 >> 12) & 31)) | (1u << ((h1 >> 2) & 31));
}

// selfrep[a] = smallest p > 0 at which a's own prefix K-mer recurs; *n_marked counts the reads that
// have one (the host skips the duplicate-A machinery of k_select when there are none)
__device__ inline void note_selfrep(uint32_t* __restrict__ selfrep, uint32_t a, uint32_t p, uint32_t* __restrict__ n_marked) {
    if (atomicMin(&selfrep[a], p) == NO_SELFREP) atomicAdd(n_marked, 1u);
}

__device__ inline uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

// Wave64 inclusive prefix sum in 6 DPP adds (gfx9 row_shr within rows of 16, then row_bcast:15 /
// row_bcast:31 to carry across rows): no LDS, no ds_bpermute round trips.
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ inline uint32_t read_last_lane(uint32_t v) { return __builtin_amdgcn_readlane(v, WAVE - 1); }

// Wave total, the same value in every lane: the DPP scan and one readlane.  (A butterfly of six __shfl_xor is six
// ds_bpermute round trips through the LDS crossbar, each with its own lgkmcnt(0): over 600 cycles per tile in
// k_scan_probe's retire step.)
__device__ inline uint32_t wave_sum(uint32_t v) { return read_last_lane(wave_incl_scan(v)); }

// 64-bit wave total from four 16-bit slices (each slice's total fits 22 bits), all through the DPP scan: exact
// modulo 2^64 and no ds_bpermute (a 64-bit butterfly is twelve of them, one after the other)
__device__ inline uint64_t wave_sum64(uint64_t v) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint64_t s0 = wave_sum(lo & 0xFFFFu), s1 = wave_sum(lo >> 16), s2 = wave_sum(hi & 0xFFFFu), s3 = wave_sum(hi >> 16);
    return s0 + (s1 << 16) + (s2 << 32) + (s3 << 48);
}

// LDS hand-off between lanes of ONE wave: DS ops of a wave execute in issue order; this only has
// to stop the compiler from moving accesses across and to drain lgkmcnt.
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ inline uint64_t funnel(uint64_t lo, uint64_t hi, uint32_t sh) {
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// Strand-mirror pairs: candidate (a, b) and its mirror (flip(b), flip(a)) describe the same overlap.  With
// c = flip(b) the two are the ordered pairs (a, c) and (c, a); exactly one of them is "canonical" (a == c
// is its own mirror): the one whose a-side read comes first in a fixed order of the reads.  `paired` names
// the order (both are bijections of the index, so exactly one member of a pair wins):
//   1  plain index order, a <= c.  Whole-set calls: workgroups run in index order and read only b's of
//      higher index, so the set of reads still in use shrinks as the kernel proceeds (verify 1.32 ms at
//      config 2 against 1.39 ms for any scrambled order).
//   2  blocks of 256 reads in bit-reversed block order (index order inside a block).  Sharded calls: ranks are
//      equidistributed over any contiguous index range, so a-side shards balanced by bases are also balanced
//      in verify work (plain order: the first of 8 shards had 15x the candidates of the last).
// Measured and rejected: a per-pair coin flip (every tile half full instead of full and empty tiles: fill
// and consume lose lane occupancy, 3.03 -> 3.29 ms); a multiplicative hash (its multiplies pushed
// k_scan_fill from 54 to 121 VGPRs and k_scan_probe into scratch); bit-reversing from bit 0 or bit 3 on
// (rank decided by the low index bits = by the XCD / CU a verify workgroup lands on: verify 1.33 -> 2.2-2.4 ms).
__device__ inline uint32_t mirror_rank(uint32_t x, uint32_t paired) {
    const uint32_t scrambled = __brev(x >> 8) | (x & 0xFFu);
    return paired == 2u ? scrambled : x;
}
__device__ inline bool canonical_pair(uint32_t a, uint32_t c, uint32_t paired) {
    return mirror_rank(a, paired) <= mirror_rank(c, paired);
}
//   3  REVERSED index order, a >= c (streamed calls, po_overlaps_to_host on reads that are still travelling to the
//      device piece by piece in index order): the kept member of a pair has its b-side read at or below a | 1, i.e.
//      in a piece that has arrived when a's piece is scanned.  Containments (B) can name any b: the bits above the
//      mode carry b_limit = (paired >> 2), the first read NOT on the device yet; a B candidate at or beyond it is
//      kept by the scan (limit = all ones there), dropped by the verify kernel (limit = end of the piece) and
//      settled from the deferred list once every piece has arrived (k_defer_split, k_verify_flat).
//      A compile-time choice (STREAM) in every kernel that evaluates it: the hand-scheduled scan kernel must keep the
//      code it has for orders 1 and 2 (tools/check_scan_isa.py).
constexpr uint32_t PAIRED_STREAM_ALL = 0xFFFFFFFFu;  // mode 3, no limit
__device__ __host__ inline uint32_t paired_stream(uint32_t b_limit) { return 3u | (b_limit << 2); }

// Which rows can candidate (a, p, b) give, and is it the member of its strand-mirror pair that this
// library computes?  bit0: A (suffix of a = prefix of b; needs la-p <= lb), bit1: B (b inside a;
// needs la-p >= lb).  Paired mode keeps A only for the canonical member and B only for a on the + strand;
// k_emit writes the mirrored rows.  A read never pairs with itself (overlapper.cpp:72,:103).
template <bool STREAM = false>
__device__ inline uint32_t keep_bits(uint32_t a, uint32_t b, uint32_t rem, uint32_t lb, uint32_t paired) {
    if (a == b) return 0;
    if constexpr (STREAM) {
        return ((rem <= lb && a >= (b ^ 1u)) ? 1u : 0u) | ((rem >= lb && (a & 1u) == 0u && b < (paired >> 2)) ? 2u : 0u);
    } else {
        return ((rem <= lb && (!paired || canonical_pair(a, b ^ 1u, paired))) ? 1u : 0u) |
               ((rem >= lb && (!paired || (a & 1u) == 0u)) ? 2u : 0u);
    }
}

// The narrow anchor table is probed in aligned groups of PROBE_GROUP slots (64 bytes: one fetch settles a probe
// unless the whole group is taken by other keys): a key's home is the first slot of its group, collisions go on
// linearly from there.  No deletions, so an empty slot anywhere from the home on means "key absent".  At 2.5-5
// slots per key the table of config 2 is 4 MB and stays in the XCDs' L2s under the streamed reads (16 slots per
// key and pairs of slots, the earlier layout: 32 MB, every probe a 128-byte line from the fabric, 6 GB per launch).
constexpr uint32_t PROBE_GROUP = 4;
__host__ __device__ inline uint32_t narrow_home(uint32_t h1, uint32_t tbits) {
    return (h1 >> (32 - tbits)) & ~(PROBE_GROUP - 1u);
}

__device__ inline void table_probe(const Slot* __restrict__ tab, uint32_t tbits, uint64_t kmer,
                                   uint32_t& start, uint32_t& cnt) {
    start = 0;
    cnt = 0;
    const uint32_t tmask = (1u << tbits) - 1u;
    if (kmer == KEY_EMPTY) {
        const u32x4 s = *reinterpret_cast<const u32x4*>(&tab[tmask + 1u]);
        start = s.z;
        cnt = s.w;
        return;
    }
    uint32_t h1, h2;
    kmer_hash(kmer, h1, h2);
    uint32_t i = narrow_home(h1, tbits);
    {
        // the home group in one go (four independent loads): the first slot that is empty or holds the key settles it
        const u32x4* g = reinterpret_cast<const u32x4*>(&tab[i]);
        const u32x4 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3];
        const uint32_t klo = (uint32_t)kmer, khi = (uint32_t)(kmer >> 32);
        const bool h0 = q0.w == 0 || (q0.x == klo && q0.y == khi), h1s = q1.w == 0 || (q1.x == klo && q1.y == khi);
        const bool h2 = q2.w == 0 || (q2.x == klo && q2.y == khi), h3 = q3.w == 0 || (q3.x == klo && q3.y == khi);
        if (h0 | h1s | h2 | h3) {
            const u32x4 s = h0 ? q0 : h1s ? q1 : h2 ? q2 : q3;
            if (s.w != 0) {
                start = s.z;
                cnt = s.w;
            }
            return;
        }
        i = (i + PROBE_GROUP) & tmask;
    }
    for (;;) {
        const u32x4 s = *reinterpret_cast<const u32x4*>(&tab[i]);
        if (s.w == 0) return;
        if ((((uint64_t)s.y << 32) | s.x) == kmer) {
            start = s.z;
            cnt = s.w;
            return;
        }
        i = (i + 1u) & tmask;
    }
}

// ----------------------------------------------------------------------------------------
// index build
// ----------------------------------------------------------------------------------------
// Everything a call starts from, in one launch: empty table, zero slot counters, "no self-repeat" for every
// read, an empty filter and zero scalars (four small launches and memsets cost ~5 us each in a 2.6 ms step).
__global__ void k_call_init(Slot* tab, uint32_t nslots, uint32_t* slot_cnt, uint32_t* slot_cur, uint32_t* selfrep,
                            uint32_t n_reads, uint32_t* bloom, uint32_t bloom_words, unsigned long long* scalars) {   // scalars: both blocks (16)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        tab[i].key = KEY_EMPTY;
        tab[i].start = 0;
        tab[i].count = 0;
        slot_cnt[i] = 0;
        slot_cur[i] = 0;
    }
    if (i < n_reads) selfrep[i] = NO_SELFREP;
    if (i < bloom_words) bloom[i] = 0;
    if (i < 16) scalars[i] = 0;
}

// the per-call part of k_call_init, for a call that reuses the index of the previous one
__global__ void k_call_reset(uint32_t* selfrep, uint32_t n_reads, unsigned long long* scalars, uint32_t n_scalars, uint32_t* clear_a, uint32_t n_a,
                             uint32_t* clear_b, uint32_t n_b) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads) selfrep[i] = NO_SELFREP;
    if (i < n_scalars) scalars[i] = 0;   // (8: this piece's block; 16: both blocks, from the base)
    // (two small per-call arrays of the narrow scan, cleared here instead of by two fill commands)
    if (i < n_a) clear_a[i] = 0;
    if (i < n_b) clear_b[i] = 0;
}

__global__ void k_fill_u32(uint32_t* p, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// One thread per read: claim the slot of its K-base prefix, count it, set its Bloom bits.
// Reads shorter than min_length can never be a `b` (they are never at a pushed node or a
// contained leaf, overlapper.cpp:40,:95) and are left out of the index.
__global__ void k_table_insert(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                               const uint32_t* __restrict__ len, uint32_t n_reads, uint32_t m,
                               uint64_t kmask, Slot* tab, uint32_t tbits, uint32_t* slot_cnt,
                               uint32_t* read_slot, uint32_t* bloom, uint32_t bloom_log2, uint32_t bits) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (len[r] < m) {
        read_slot[r] = 0xFFFFFFFFu;
        return;
    }
    const uint64_t key = words[woff[r]] & kmask;
    const uint32_t tmask = (1u << tbits) - 1u;
    uint32_t h1, h2;
    kmer_hash(key, h1, h2);
    uint32_t i;
    if (key == KEY_EMPTY) {
        i = tmask + 1u;
    } else {
        i = narrow_home(h1, tbits);
        for (;;) {
            unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[i].key),
                                                (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (prev == KEY_EMPTY || prev == key) break;
            i = (i + 1u) & tmask;
        }
    }
    atomicAdd(&slot_cnt[i], 1u);
    read_slot[r] = i;
    if (bits == 2) {
        // 2-bit reads: 64-bit blocks addressed by sequence bits themselves (uniform for DNA, no
        // multiply): block = bits 5.. of the first 16 bases, two bits in the block's low word (from
        // bases 0-2 and 10-12), one in its high word (from bases 16-18).  See filter_tile<2>.
        const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        const uint32_t blk = (lo >> 5) & ((1u << (bloom_log2 - 6)) - 1u);
        atomicOr(&bloom[2 * blk], (1u << (lo & 31)) | (1u << ((lo >> 20) & 31)));
        atomicOr(&bloom[2 * blk + 1], 1u << (hi & 31));
    } else {
        uint32_t bword, bmask;
        bloom_slot(key, bloom_log2, bword, bmask);
        atomicOr(&bloom[bword], bmask);
    }
}

// Runs after the chains are filled and sorted.  Most prefixes belong to one read only: that read
// and its length are stored in the slot itself, so a probe that hits needs no second round trip.
__global__ void k_table_finalize(Slot* tab, uint32_t nslots, const uint32_t* __restrict__ slot_cnt,
                                 const uint32_t* __restrict__ slot_start, const uint32_t* __restrict__ chain,
                                 const uint32_t* __restrict__ len) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        const uint32_t c = slot_cnt[i];
        if (c == 1) {
            const uint32_t b = chain[slot_start[i]];
            tab[i].start = b;
            tab[i].count = SLOT_SINGLE | len[b];
        } else {
            tab[i].start = slot_start[i];
            tab[i].count = c;
        }
    }
}

__global__ void k_chain_fill(const uint32_t* __restrict__ read_slot, uint32_t n_reads,
                             const uint32_t* __restrict__ slot_start, uint32_t* slot_cur, uint32_t* chain) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t s = read_slot[r];
    if (s == 0xFFFFFFFFu) return;
    chain[slot_start[s] + atomicAdd(&slot_cur[s], 1u)] = r;
}

// Chains are filled in arrival order; put each one into ascending read index so that candidate
// (and so row) order is the same on every run.  Short chains: one thread.  Long chains (many
// reads with one prefix) are queued for k_chain_sort_long.
constexpr uint32_t CHAIN_SHORT = 16;

template <typename E>
__global__ void k_chain_sort_short(const uint32_t* __restrict__ slot_cnt, const uint32_t* __restrict__ slot_start,
                                   uint32_t nslots, E* chain, uint32_t* long_list, uint32_t* n_long) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const uint32_t c = slot_cnt[i];
    if (c < 2) return;
    if (c > CHAIN_SHORT) {
        long_list[atomicAdd(n_long, 1u)] = i;
        return;
    }
    E* v = chain + slot_start[i];
    for (uint32_t x = 1; x < c; ++x) {
        E key = v[x];
        uint32_t y = x;
        while (y > 0 && v[y - 1] > key) {
            v[y] = v[y - 1];
            --y;
        }
        v[y] = key;
    }
}

// One workgroup per long chain: rank sort through a scratch copy (read indices are distinct).
template <typename E>
__global__ void k_chain_sort_long(const uint32_t* __restrict__ slot_cnt, const uint32_t* __restrict__ slot_start,
                                  const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ n_long,
                                  E* chain, E* scratch) {
    for (uint32_t li = blockIdx.x; li < *n_long; li += gridDim.x) {
        const uint32_t s = long_list[li];
        const uint32_t c = slot_cnt[s], st = slot_start[s];
        for (uint32_t x = threadIdx.x; x < c; x += blockDim.x) scratch[st + x] = chain[st + x];
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < c; x += blockDim.x) {
            const E v = scratch[st + x];
            uint32_t rank = 0;
            for (uint32_t y = 0; y < c; ++y) rank += scratch[st + y] < v;
            chain[st + rank] = v;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------
// wide index (large read sets)
// ----------------------------------------------------------------------------------------
// The LDS filter of the position scan holds ~10 bits per read prefix at up to ~150 k reads; beyond
// that it stops filtering and every position would reach the table.  For large read sets the roles
// are swapped: index the K-mers at the first W offsets j = 0..W-1 of every read b (W = bases per
// word, K = W) and probe only the WORD-ALIGNED K-mers of a -- the packed words themselves, no
// shifts, no filter.  An overlap (a, p, b) of length >= 2W-1 is found exactly once: at the word of a
// that starts at p + j with j = (-p) mod W.  Work is one table probe per word of a (1/W of the
// positions) against a table of W entries per read: linear in the input, at the price of random
// access into a table that lives in HBM.  Needs min_length >= 2W-1 (63 bases at 2 bit).
//
// Slot: {key, start, count} as in the narrow index.  Chain entries and the one-read chain embedded in its slot: WideEnc.
// Window minimisers (WW > 1; round 4).  Probing EVERY word of a is one random line fetch per word -- 207 M of them at
// config 3, at the 54 G lines/s this memory system gives random lines beyond an XCD's L2 (tools/ubench.hip, measured) --
// and 3 of 4 probes miss.  Instead of b's K-mer at offset j, the index holds, per read b and phase j, the MINIMISER of
// the first WW word-spaced K-mers of that phase -- the K-mer at offset o = j + W t*, t* = argmin over t < WW of
// (hash(K-mer at j + W t), t) -- and the scan probes only those words of a that are the minimiser of SOME window of WW
// consecutive words of a (about 2 / (WW + 1) of them).  No overlap is lost: an overlap (a, p, b) of l >= m bases, j = (-p)
// mod W, covers the words of a at p + j + W t for t < floor((l - j) / W), which are b's K-mers at offsets j + W t; with
// m >= W WW + W - 1 the first WW of them lie inside every overlap, they form a complete window of a's words and the
// same sequence as b's phase-j window, so both sides pick the same element, the word at q = p + o is probed, finds
// (b, o) and gives p = q - o.  Found exactly once: for given (a, p, b) the phase, hence the entry, hence q is fixed.
// The streamed step keeps WW = 1: it builds the index from the first TWO words of every read, sent ahead of the pieces.
template <int WW>
struct WideEnc {
    static constexpr uint32_t OBITS = WW == 1 ? 5u : 9u;           // bits of the offset o of an entry (o < W WW <= 512)
    static constexpr uint32_t OMAX = (1u << OBITS) - 1u;
    // a one-read chain lives in its slot: start = b, count = SLOT_SINGLE | o << SLOT_LEN_BITS | len[b]
    static constexpr uint32_t SLOT_LEN_BITS = 31u - OBITS;         // 26 / 22
    // A chain entry, 64 bits: [OMAX - o : OBITS | read : 32 | length of the read : 32 - OBITS] -- sorted as a number it is
    // ordered by descending offset = ascending candidate position p = q - o, then by read; the read's length rides along
    // so that walking a chain is ONE load per entry instead of two dependent ones (chain, then len[read]).  A length that
    // does not fit (all ones) sends the reader to len[].
    static constexpr uint32_t CHAIN_LEN_BITS = 32u - OBITS;        // 27 / 23
    static constexpr uint32_t CHAIN_LEN_ESC = (1u << CHAIN_LEN_BITS) - 1u;
    __host__ __device__ static inline uint64_t entry(uint32_t o, uint32_t read, uint32_t len) {
        return ((uint64_t)(OMAX - o) << (32u + CHAIN_LEN_BITS)) | ((uint64_t)read << CHAIN_LEN_BITS) | (len < CHAIN_LEN_ESC ? len : CHAIN_LEN_ESC);
    }
    __host__ __device__ static inline uint32_t read(uint64_t e) { return (uint32_t)(e >> CHAIN_LEN_BITS); }
    __host__ __device__ static inline uint32_t off(uint64_t e) { return OMAX - (uint32_t)(e >> (32u + CHAIN_LEN_BITS)); }
    __device__ static inline uint32_t len(uint64_t e, const uint32_t* __restrict__ lens) {
        const uint32_t l = (uint32_t)e & CHAIN_LEN_ESC;
        return l != CHAIN_LEN_ESC ? l : lens[read(e)];
    }
    __device__ static inline uint32_t slot_off(uint32_t count) { return (count >> SLOT_LEN_BITS) & OMAX; }
    __device__ static inline uint32_t slot_len(uint32_t count) { return count & ((1u << SLOT_LEN_BITS) - 1u); }
};
// the order of a window's elements: hash first (its low 7 bits make room for the position), position second
__host__ __device__ inline uint32_t window_key(uint32_t h2, uint32_t pos) { return (h2 & ~127u) | pos; }

// Sliced wide index (multi-GPU): the table is N sub-tables, a key lives in sub-table wide_slice(key); rank g builds
// sub-table g only and the sub-tables travel in one all-gather (phasm_amd/dist.py: IndexExchange).  The all-ones key
// (the CAS sentinel, kept in an extra slot) belongs to sub-table 0.
__device__ inline uint32_t wide_slice(uint64_t key, uint32_t h2, uint32_t n_slices) {
    return key == KEY_EMPTY ? 0u : __umulhi(h2, n_slices);
}

template <int BITS, int WW>
__global__ void k_wide_insert(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                              const uint32_t* __restrict__ len, uint32_t n_reads, uint32_t m, Slot* tab,
                              uint32_t tbits, uint32_t* slot_cnt, uint32_t* entry_slot, uint16_t* __restrict__ entry_off,
                              uint32_t n_slices, uint32_t my_slice) {
    constexpr uint32_t W = 64 / BITS;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)n_reads * W) return;
    const uint32_t r = (uint32_t)(e / W), j = (uint32_t)(e % W);
    if (len[r] < m) {
        entry_slot[e] = 0xFFFFFFFFu;
        return;
    }
    const uint64_t* __restrict__ rw = words + woff[r];
    uint64_t key = funnel(rw[0], rw[1], j * BITS);  // m >= 2W-1: the K-mer at offset j is inside the read
    uint32_t h1, h2;
    kmer_hash(key, h1, h2);
    uint32_t o = j;
    if constexpr (WW > 1) {
        // the minimiser of this phase's first WW K-mers (m >= W WW + W - 1: all of them inside the read)
        uint32_t best = window_key(h2, 0u);
        for (uint32_t t = 1; t < (uint32_t)WW; ++t) {
            const uint64_t k = funnel(rw[t], rw[t + 1], j * BITS);
            uint32_t g1, g2;
            kmer_hash(k, g1, g2);
            const uint32_t c = window_key(g2, t);
            if (c < best) {
                best = c;
                key = k;
                h1 = g1;
                h2 = g2;
                o = j + W * t;
            }
        }
    }
    entry_off[e] = (uint16_t)o;
    const uint32_t tmask = (1u << tbits) - 1u;
    if (n_slices > 1u && wide_slice(key, h2, n_slices) != my_slice) {   // another rank's sub-table
        entry_slot[e] = 0xFFFFFFFFu;
        return;
    }
    uint32_t i;
    if (key == KEY_EMPTY) {
        i = tmask + 1u;
    } else {
        i = h1 >> (32 - tbits);
        for (;;) {
            unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[i].key),
                                                (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (prev == KEY_EMPTY || prev == key) break;
            i = (i + 1u) & tmask;
        }
    }
    atomicAdd(&slot_cnt[i], 1u);
    entry_slot[e] = i;
}

template <int BITS, int WW>
__global__ void k_wide_chain_fill(const uint32_t* __restrict__ entry_slot, const uint16_t* __restrict__ entry_off, uint64_t n_entries,
                                  const uint32_t* __restrict__ slot_start, uint32_t* slot_cur, uint64_t* chain,
                                  const uint32_t* __restrict__ len) {
    constexpr uint32_t W = 64 / BITS;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t s = entry_slot[e];
    if (s == 0xFFFFFFFFu) return;
    const uint32_t r = (uint32_t)(e / W);
    chain[slot_start[s] + atomicAdd(&slot_cur[s], 1u)] = WideEnc<WW>::entry(entry_off[e], r, len[r]);
}

template <int BITS, int WW>
__global__ void k_wide_finalize(Slot* tab, uint32_t nslots, const uint32_t* __restrict__ slot_cnt,
                                const uint32_t* __restrict__ slot_start, const uint64_t* __restrict__ chain,
                                const uint32_t* __restrict__ len) {
    using E = WideEnc<WW>;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const uint32_t c = slot_cnt[i];
    tab[i].start = slot_start[i];
    tab[i].count = c;
    if (c == 1) {
        const uint64_t e = chain[slot_start[i]];
        const uint32_t b = E::read(e), o = E::off(e);
        const uint32_t lb = E::len(e, len);
        if (lb < (1u << E::SLOT_LEN_BITS)) {
            tab[i].start = b;
            tab[i].count = SLOT_SINGLE | (o << E::SLOT_LEN_BITS) | lb;
        }
    }
}

// candidates of the word-aligned K-mer at base offset pw of read a, given the slot it found
template <int BITS, bool STREAM = false, int WW = 1, typename F>
__device__ inline void for_each_candidate_wide(const uint64_t* __restrict__ chain, const uint32_t* __restrict__ len,
                                               uint32_t paired, uint32_t z, uint32_t w, uint32_t a, uint32_t la,
                                               uint32_t pw, uint32_t m, F&& f) {
    using E = WideEnc<WW>;
    auto one = [&](uint32_t b, uint32_t j, uint32_t lb) __attribute__((always_inline)) {
        if (j > pw) return;            // p = pw - j would be negative
        const uint32_t p = pw - j;
        if (p + m > la) return;        // too close to the end of a to reach min_length
        const uint32_t k = keep_bits<STREAM>(a, b, la - p, lb, paired);
        if (k) f(b, p, k);
    };
    if (w & SLOT_SINGLE) {
        one(z, E::slot_off(w), E::slot_len(w));
    } else {
        for (uint32_t i = 0; i < w; ++i) {
            const uint64_t e = chain[z + i];
            one(E::read(e), E::off(e), E::len(e, len));
        }
    }
}

struct WideArgs {
    const uint64_t* words;
    const TileRec* tiles;
    uint32_t tile_begin, tile_end;
    uint32_t m;
    const Slot* table;
    uint32_t tbits;
    const uint64_t* chain;
    const uint32_t* len;
    uint32_t paired;
    uint32_t n_slices;        // > 1: `table` is a gathered sliced index -- N chunks of chunk_slots 16-byte units, each
    uint32_t chunk_slots;     //      [2^tbits + 1 slots | padding | chain entries from unit chain_off_slots on]
    uint32_t chain_off_slots;
    uint32_t* tile_count;
    uint32_t* lane_slot;   // per (tile, lane): slot index + 1 of the word's K-mer, 0 = no candidates
    const uint32_t* tile_off;
    uint32_t* cand_a;
    uint32_t* cand_p;
    uint32_t* cand_b;
};

// One wave per tile, one word per lane: probe the table with the word itself.  FILL = false: count
// candidates per tile and remember the slot per lane; FILL = true: write the candidates.
//
// The chain entries of a wave's hits are walked by ALL its lanes: about a quarter of the words hit, a quarter of those
// hits have a chain of several reads, and a lane that walks its own chain does one dependent load after the other while
// the other 63 wait (the counting pass of config 5 spent its 23 ms there, not on the misses: DESIGN.md 5.1).  Every hit
// lane contributes its entries (1 for a slot with its one read embedded, w for a chain) to a task list in lane order;
// the tasks are handed out 64 at a time, one per lane, so that a round of chain loads is ONE round trip for the whole
// wave.  Task order = lane, then entry = the order the per-lane walk emits, so FILL writes the same array.  A wave with
// more than WIDE_TASK_CAP entries (tandem repeats) walks per lane as before.
constexpr uint32_t WIDE_TASK_CAP = 256;

// Which words of a tile are window minimisers (WW > 1)?  V = the window keys of the PAD = WW - 1 words before the tile,
// the tile's 64 words and the PAD words behind it (0xFFFFFFFF for positions outside the read's complete words: a window
// that holds one is not a window of the read, and whatever it selects beyond the true minimisers only costs a probe),
// held as R0[l] = V[l] and R1[l] = V[64 + l] (l < 2 PAD).  Word i is selected iff it is the minimum of one of the WW
// windows that contain it: sliding minimum over WW (doubling, forwards), then the sliding MAXIMUM of those minima over
// the windows that contain i (doubling, backwards) -- every one of them is <= V[i], so the maximum equals V[i] iff one
// of them does.  All 64 lanes take part (ds_bpermute).
// own: window_key(hash, PAD + lane) of the lane's own word; side: lanes < PAD the word PAD - lane before the tile
// (position lane), lanes PAD .. 2 PAD - 1 the words behind it (position 64 + lane).
template <int WW>
__device__ __forceinline__ bool window_selected(uint32_t own, uint32_t side, uint32_t lane) {
    constexpr uint32_t PAD = WW - 1, INF = 0xFFFFFFFFu;
    const uint32_t up = (uint32_t)__shfl((int)own, (int)((lane - PAD) & 63u));        // own[lane - PAD]
    const uint32_t dn = (uint32_t)__shfl((int)own, (int)((lane + 64u - PAD) & 63u));  // own[lane + 64 - PAD]
    uint32_t r0 = lane < PAD ? side : up;
    uint32_t r1 = lane < PAD ? dn : (lane < 2u * PAD ? side : INF);
#pragma unroll
    for (uint32_t k = 1; k < (uint32_t)WW; k <<= 1) {   // r[i] = min V[i .. i + 2k - 1]
        const uint32_t x0 = (uint32_t)__shfl((int)r0, (int)((lane + k) & 63u));
        const uint32_t x1 = (uint32_t)__shfl((int)r1, (int)((lane + k) & 63u));
        const uint32_t n0 = lane + k < 64u ? x0 : x1;
        const uint32_t n1 = lane + k < 2u * PAD ? x1 : INF;
        r0 = r0 < n0 ? r0 : n0;
        r1 = r1 < n1 ? r1 : n1;
    }
#pragma unroll
    for (uint32_t k = 1; k < (uint32_t)WW; k <<= 1) {   // r[i] = max Wmin[i - 2k + 1 .. i]
        const uint32_t y0 = (uint32_t)__shfl((int)r0, (int)((lane - k) & 63u));
        const uint32_t y1 = (uint32_t)__shfl((int)r1, (int)((lane - k) & 63u));
        const uint32_t m0 = lane >= k ? y0 : 0u;        // (no window starts before the loaded words)
        const uint32_t m1 = lane >= k ? y1 : y0;        // V-index 64 + lane - k < 64: in r0, at lane 64 + lane - k
        r0 = r0 > m0 ? r0 : m0;
        r1 = r1 > m1 ? r1 : m1;
    }
    // the lane's own word sits at V-index PAD + lane
    const uint32_t a0 = (uint32_t)__shfl((int)r0, (int)((lane + PAD) & 63u));
    const uint32_t a1 = (uint32_t)__shfl((int)r1, (int)((lane + PAD) & 63u));
    const uint32_t y = lane + PAD < 64u ? a0 : a1;
    return own != INF && y == own;
}

template <int BITS, bool FILL, bool STREAM = false, int WW = 1>
__global__ __launch_bounds__(256) void k_wide_scan(const WideArgs A, const CandGuard G) {
    constexpr uint32_t W = 64 / BITS;
    using E = WideEnc<WW>;
    __shared__ uint8_t s_owner[256 / WAVE][WIDE_TASK_CAP];
    __shared__ uint8_t s_kept[256 / WAVE][WAVE];
    const uint32_t lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    if (FILL && G.overflow()) return;
    const uint32_t t = __builtin_amdgcn_readfirstlane(A.tile_begin + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (t >= A.tile_end) return;
    const TileRec rec = A.tiles[t];
    const uint32_t a = rec.read, la = rec.la;
    const uint32_t wi = rec.word0 + lane;
    const uint32_t pw = wi * W;
    const size_t li = (size_t)t * WAVE + lane;
    uint32_t n = 0, z = 0, w = 0, slot1 = 0, base = 0;
    const bool live = la >= A.m && pw + W <= la;  // the word lies wholly inside the read
    if (!FILL) {
        uint64_t kmer = 0;
        uint32_t h1 = 0, h2 = 0;
        if (live) {
            kmer = A.words[rec.wabs + lane];
            kmer_hash(kmer, h1, h2);
        }
        bool probe = live;
        if constexpr (WW > 1) {
            // only the window minimisers among a's words are probed (see WideEnc): the lane's own key, and -- lanes below
            // 2 (WW - 1) -- the key of one of the words just before / just behind the tile
            constexpr uint32_t PAD = WW - 1;
            const uint32_t n_words = la >= A.m ? la / W : 0u;   // complete words of the read
            const int64_t sw = lane < PAD ? (int64_t)rec.word0 - (int64_t)PAD + (int64_t)lane : (int64_t)rec.word0 + 64 + (int64_t)(lane - PAD);
            uint32_t side = 0xFFFFFFFFu;
            if (lane < 2u * PAD && sw >= 0 && sw < (int64_t)n_words) {
                uint32_t g1, g2;
                kmer_hash(A.words[rec.wread + (uint64_t)sw], g1, g2);
                side = window_key(g2, lane < PAD ? lane : 64u + lane);
            }
            const uint32_t own = live ? window_key(h2, PAD + lane) : 0xFFFFFFFFu;
            probe = window_selected<WW>(own, side, lane);
        }
        if (probe) {
            const uint32_t tmask = (1u << A.tbits) - 1u;
            // (sliced index: the key's sub-table; its slots and its chain segment sit in one chunk)
            base = A.n_slices > 1u ? wide_slice(kmer, h2, A.n_slices) * A.chunk_slots : 0u;
            const Slot* __restrict__ tab = A.table + base;
            uint32_t i;
            if (kmer == KEY_EMPTY) {
                i = tmask + 1u;
                const u32x4 s = *reinterpret_cast<const u32x4*>(&tab[i]);
                z = s.z;
                w = s.w;
            } else {
                i = h1 >> (32 - A.tbits);
                for (;;) {
                    const u32x4 s = *reinterpret_cast<const u32x4*>(&tab[i]);
                    if (s.w == 0) break;
                    if ((((uint64_t)s.y << 32) | s.x) == kmer) {
                        z = s.z;
                        w = s.w;
                        break;
                    }
                    i = (i + 1u) & tmask;
                }
            }
            if (w) slot1 = base + i + 1u;
        }
    } else {
        slot1 = A.lane_slot[li];
        if (slot1) {
            const u32x4 s = *reinterpret_cast<const u32x4*>(&A.table[slot1 - 1u]);
            z = s.z;
            w = s.w;
            if (A.n_slices > 1u) base = ((slot1 - 1u) / A.chunk_slots) * A.chunk_slots;
        }
    }
    auto chain_of = [&](uint32_t b0) __attribute__((always_inline)) -> const uint64_t* {
        return A.n_slices > 1u ? reinterpret_cast<const uint64_t*>(A.table + b0 + A.chain_off_slots) : A.chain;
    };
    // ---- the wave's task list
    const uint32_t cnt = !w ? 0u : (w & SLOT_SINGLE) ? 1u : w;
    const uint32_t incl = wave_incl_scan(cnt), excl = incl - cnt;
    const uint32_t T = read_last_lane(incl);
    if (T <= WIDE_TASK_CAP) {
        for (uint32_t i = 0; i < cnt; ++i) s_owner[wv][excl + i] = (uint8_t)lane;
        if (!FILL) s_kept[wv][lane] = 0;
        wave_lds_fence();
        uint32_t out = FILL ? A.tile_off[t] : 0u;   // FILL: where the next kept candidate goes; COUNT: kept so far
        for (uint32_t r0 = 0; r0 < T; r0 += WAVE) {
            const uint32_t task = r0 + lane;
            bool kept = false;
            // (the shuffles with every lane active: a task's owner must be, and lanes beyond the list ask for the last task's)
            const uint32_t o = s_owner[wv][task < T ? task : T - 1u];
            const uint32_t zo = (uint32_t)__shfl((int)z, (int)o), wo = (uint32_t)__shfl((int)w, (int)o);
            const uint32_t eo = task - (uint32_t)__shfl((int)excl, (int)o);
            const uint32_t bo = A.n_slices > 1u ? (uint32_t)__shfl((int)base, (int)o) : 0u;
            uint32_t cb = 0, cp = 0;
            if (task < T) {
                const uint32_t pwo = (rec.word0 + o) * W;
                uint32_t j, lb;
                if (wo & SLOT_SINGLE) {
                    cb = zo;
                    j = E::slot_off(wo);
                    lb = E::slot_len(wo);
                } else {
                    const uint64_t e = chain_of(bo)[zo + eo];
                    cb = E::read(e);
                    j = E::off(e);
                    lb = E::len(e, A.len);
                }
                if (j <= pwo) {
                    cp = pwo - j;
                    kept = cp + A.m <= la && keep_bits<STREAM>(a, cb, la - cp, lb, A.paired) != 0u;
                }
            }
            const uint64_t bal = __ballot(kept);
            if (FILL) {
                if (kept) {
                    const uint32_t at = out + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    A.cand_a[at] = a;
                    A.cand_p[at] = cp;
                    A.cand_b[at] = cb;
                }
            } else if (kept) {
                s_kept[wv][o] = 1;
            }
            out += (uint32_t)__popcll(bal);
        }
        if (!FILL) {
            wave_lds_fence();
            if (!s_kept[wv][lane]) slot1 = 0;   // (a hit none of whose entries is kept: nothing for the fill pass)
            A.lane_slot[li] = slot1;
            if (lane == 0) A.tile_count[t] = out;
        }
        return;
    }
    // ---- more entries than the task list holds: every lane walks its own
    const uint64_t* __restrict__ chain = chain_of(base);
    if (!FILL) {
        if (w) for_each_candidate_wide<BITS, STREAM, WW>(chain, A.len, A.paired, z, w, a, la, pw, A.m,
                                                     [&](uint32_t, uint32_t, uint32_t) { ++n; });
        if (!n) slot1 = 0;
        A.lane_slot[li] = slot1;
        const uint32_t tot = wave_sum(n);
        if (lane == 0) A.tile_count[t] = tot;
    } else {
        if (slot1)
            for_each_candidate_wide<BITS, STREAM, WW>(chain, A.len, A.paired, z, w, a, la, pw, A.m,
                                          [&](uint32_t, uint32_t, uint32_t) { ++n; });
        const uint32_t inc = wave_incl_scan(n);
        uint32_t off = A.tile_off[t] + inc - n;
        if (n) {
            for_each_candidate_wide<BITS, STREAM, WW>(chain, A.len, A.paired, z, w, a, la, pw, A.m,
                                          [&](uint32_t b, uint32_t p, uint32_t) {
                                              A.cand_a[off] = a;
                                              A.cand_p[off] = p;
                                              A.cand_b[off] = b;
                                              ++off;
                                          });
        }
    }
}

// ----------------------------------------------------------------------------------------
// position scan
// ----------------------------------------------------------------------------------------
struct ScanArgs {
    const uint64_t* words;
    const TileRec* tiles;
    uint32_t tile_begin, tile_end;
    uint32_t m;       // effective min_length (>= 1)
    uint64_t kmask;   // low K*BITS bits
    const uint32_t* bloom;
    uint32_t bloom_log2;   // filter bits = 1 << bloom_log2 (32-bit blocks)
    const Slot* table;
    uint32_t tbits;
    const uint32_t* chain;
    const uint32_t* len;
    uint32_t paired;
    uint32_t* selfrep;     // COUNT: min p>0 at which a read's own prefix K-mer recurs
    uint32_t* n_selfrep;   //        number of reads that have one
    unsigned long long* dbg;  // diagnostic builds (-DPO_STAMPS): per wave, cycles per pipeline stage
    uint2* left;           // COUNT out: per wave LEFT_CAP deferred positions {tile, lane << 8 | s}
    uint32_t* left_cnt;    //            and how many each wave deferred
    uint32_t* tile_extra;  //            candidates resolved in place when a leftover list was full (zeroed; rare)
    uint32_t* tile_count;  // COUNT out: candidates per tile
    uint32_t* truemask;    // COUNT out / FILL in: per (tile, lane) bit s set = position p0+s has candidates
    const uint32_t* tile_off;  // FILL in
    uint32_t* cand_a;
    uint32_t* cand_p;
    uint32_t* cand_b;
};

template <int BITS>
__device__ inline uint64_t window(uint64_t w0, uint64_t w1, int s) {
    return s == 0 ? w0 : ((w0 >> (s * BITS)) | (w1 << ((64 - s * BITS) & 63)));
}

// Filter one lane's W positions (K-mers starting in word w0, spilling into w1) against the LDS
// filter; bit s of the result = position s passes.
//  BITS == 8: hashed blocked Bloom filter, 32-bit blocks, 3 bits per key (bloom_slot).
//  BITS == 2: the packed bases are already uniform bits, so no hash: T[s] = the 32 bits of the read
//  starting at base s; a K-mer's block is T[s] bits 5..18, its three bits are T[s] & 31 and
//  (T[s] >> 20) & 31 (bases 10-12, clear of the block index) in the block's low word and
//  T[s+16] & 31 (bases 16-18) in its high word.  One ds_read_b64 and ~12 VALU ops per position;
//  T[s+16] is shared between positions s and s+16.  Three bits keep the survivors near 2.6 % of the
//  positions (1 % are real), so that a lane rarely has more than NPEND of them.
// FULLK: the anchor is a whole word (min_length >= W, the normal case): no K-mer mask to apply.
template <int BITS, bool FULLK>
__device__ inline uint32_t filter_tile(const uint32_t* __restrict__ s_bloom, uint32_t bloom_log2, uint64_t w0,
                                       uint64_t w1, uint64_t kmask) {
    constexpr int W = 64 / BITS;
    uint32_t hitmask = 0;
    if constexpr (BITS == 2) {
        const uint32_t x[4] = {(uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32)};
        const uint32_t klo = FULLK ? ~0u : (uint32_t)kmask, khi = FULLK ? ~0u : (uint32_t)(kmask >> 32);
        const uint32_t bmask = (1u << (bloom_log2 - 6)) - 1u;
        const uint32_t amask = bmask << 3;
        // T[s] = the 32 bits of the read that start at base s of this lane's word (s < 48), made when first needed
        uint32_t T[48];
        auto make_T = [&](int s) __attribute__((always_inline)) {
            const int j = (2 * s) >> 5, sh = (2 * s) & 31;
            T[s] = sh ? __builtin_amdgcn_alignbit(x[j + 1], x[j], sh) : x[j];
        };
        typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
        typedef const u32x2 __attribute__((address_space(3))) lds_u32x2;
        // The filter is the first thing in the kernel's LDS (offset 0, no static LDS in k_scan_probe): the byte offset
        // of a block IS its LDS address -- spelled as an address-space-3 pointer so that the compiler does not add
        // the (relocatable, zero) base to it.  T[s + 1] is T[s] >> 2 and T[s + 10] is T[s] >> 20 (with younger bits on
        // top): with a whole-word anchor the block address and the second bit index cost no shift of their own.
        auto block_of = [&](int s) __attribute__((always_inline)) -> u32x2 {
            const uint32_t addr = FULLK ? (T[s + 1] & amask) : (((T[s] & klo) >> 2) & amask);
            return *reinterpret_cast<lds_u32x2*>((uintptr_t)addr);
        };
        // PO_FILTER_DEPTH block reads in flight per lane, kept in this order by a scheduling barrier per position:
        // the reads are random (about 3.5 lanes of a 32-lane group on one bank pair), and four waves per SIMD do
        // not cover that latency with two in flight (the order hipcc picks on its own)
        static_assert(PO_FILTER_DEPTH >= 1 && PO_FILTER_DEPTH <= 16, "the T window below reaches 17 bases ahead");
        u32x2 blk[PO_FILTER_DEPTH];
#pragma unroll
        for (int s = 0; s <= 16; ++s) make_T(s);
#pragma unroll
        for (int s = 0; s < PO_FILTER_DEPTH; ++s) blk[s] = block_of(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s + 17 < 48) make_T(s + 17);
            const uint32_t t1 = T[s] & klo, t2 = T[s + 16] & khi;
            const uint32_t sel2 = FULLK ? T[s + 10] : (t1 >> 20);
            const u32x2 b = blk[s % PO_FILTER_DEPTH];
            // only bit 0 of the product enters: v_alignbit shifts it in at the top, after 32 rounds position s is bit s
            hitmask = __builtin_amdgcn_alignbit((b.x >> (t1 & 31)) & (b.x >> (sel2 & 31)) & (b.y >> (t2 & 31)), hitmask, 1);
            if (s + PO_FILTER_DEPTH < 32) blk[s % PO_FILTER_DEPTH] = block_of(s + PO_FILTER_DEPTH);
            if constexpr (PO_SCAN_PAD > 0) valu_pad<PO_SCAN_PAD>(hitmask, 0u);   // (measurement builds only; x ^ 0)
            asm volatile("" : "+v"(hitmask));   // pins this position's arithmetic in front of the barrier
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const uint32_t wshift = 32 - (bloom_log2 - 5);
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const uint64_t kmer = FULLK ? window<BITS>(w0, w1, s) : (window<BITS>(w0, w1, s) & kmask);
            const uint32_t lo = (uint32_t)kmer, hi = (uint32_t)(kmer >> 32);
            const uint32_t h1 = (lo ^ ((hi << 13) | (hi >> 19))) * 0x9E3779B1u;
            const uint32_t bm = (1u << ((h1 >> 7) & 31)) | (1u << ((h1 >> 12) & 31)) | (1u << ((h1 >> 2) & 31));
            const uint32_t bw = s_bloom[h1 >> wshift];
            hitmask |= ((bw & bm) == bm ? 1u : 0u) << s;
        }
    }
    return hitmask;
}

// Candidates of one position p of read a, given the slot its K-mer found (z = start/read, w = count
// word).  Calls f(b, lb, keep) for every chain entry that survives keep_bits().
// (pointers by value: a reference to the kernel-argument struct would force it into scratch)
template <bool STREAM = false, typename F>
__device__ inline void for_each_candidate(const uint32_t* __restrict__ chain, const uint32_t* __restrict__ len,
                                          uint32_t paired, uint32_t z, uint32_t w, uint32_t a, uint32_t rem, F&& f) {
    if (w & SLOT_SINGLE) {
        const uint32_t lb = w & ~SLOT_SINGLE;
        const uint32_t k = keep_bits<STREAM>(a, z, rem, lb, paired);
        if (k) f(z, lb, k);
    } else {
        for (uint32_t j = 0; j < w; ++j) {
            const uint32_t b = chain[z + j];
            const uint32_t lb = len[b];
            const uint32_t k = keep_bits<STREAM>(a, b, rem, lb, paired);
            if (k) f(b, lb, k);
        }
    }
}

// Scan pass 1 (filter + count), ONE kernel.  Persistent workgroups (grid <= #CUs); the filter
// lives in LDS for the whole launch.  One wave per tile = 64 words = 64*W positions; lane l owns
// word l (+ the next one for windows that straddle) and tests its W positions against the filter
// (filter_tile: VALU + one LDS read per position).  The wave's survivors (about 40 of 2048 positions)
// are compacted through a 64-entry wave-private LDS queue, so that every lane settles at most ONE
// survivor per tile: it requests the two candidate table slots from L2 and the wave goes straight on
// to filter the next tile -- the replies are consumed one tile later, under the filter arithmetic.
// Positions that cannot be settled from registers (a third table slot, a chain of several reads, a
// 65th survivor) go to a per-wave leftover list and are settled afterwards by k_scan_fixup.
// Out: truemask[tile][lane] bit s = position has candidates; tile_count[tile] = their number.
#ifndef PO_LEFT_CAP
#define PO_LEFT_CAP 2048
#endif
constexpr uint32_t LEFT_CAP = PO_LEFT_CAP;   // deferred positions per scan wave before it falls back to resolving them in place
constexpr int SCAN_LDS_PER_WAVE = 2 * WAVE * 4 + WAVE * 4;  // queue of positions (two rounds), result masks
#ifdef PO_STAMPS
#define PO_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PO_STAMP(var) do { } while (0)
#endif

// COMPILER CONTRACT: the landing registers of the asm loads below are "defined" at issue as far as hipcc knows; it must
// not copy or spill them before the asm wait that retires them.  tools/check_scan_isa.py walks the generated code for
// that; it runs inside phasm_amd/build.py (a violating library is deleted, not shipped) and in the CPU tests, and was
// validated against the code of hipcc 7.2.26015 (ROCm 7.2.0, AMD clang 22.0.0git) -- build.py:VALIDATED_HIPCC; the
// test fails on any other compiler until the walk has been checked against its output.
template <int BITS, bool FULLK, bool STREAM = false>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_probe(const ScanArgs A) {  // A stays in SGPRs: never take its address
    constexpr int W = 64 / BITS;
    extern __shared__ uint64_t smem[];
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t bloom_words = (1u << A.bloom_log2) >> 5;
    uint32_t* s_bloom = reinterpret_cast<uint32_t*>(smem);                      // filter first: its block offsets are LDS addresses
    uint32_t* qs = s_bloom + bloom_words + wave * (2 * WAVE);                   // queue: lane << 8 | s
    uint32_t* tm = s_bloom + bloom_words + (2 * nwaves + wave) * WAVE;          // result mask per owner lane
    for (uint32_t i = threadIdx.x; i < bloom_words; i += blockDim.x) s_bloom[i] = A.bloom[i];
    tm[lane] = 0;
    __syncthreads();

    const uint32_t stride = gridDim.x * nwaves;
    const uint32_t tbits = A.tbits, tmask = (1u << A.tbits) - 1u;
    const Slot* __restrict__ table = A.table;
    const uint32_t* __restrict__ chain = A.chain;
    const uint32_t* __restrict__ len = A.len;
    const uint64_t* __restrict__ words = A.words;
    const TileRec* __restrict__ tiles = A.tiles;
    uint32_t* __restrict__ selfrep = A.selfrep;
    uint32_t* __restrict__ n_selfrep = A.n_selfrep;
    uint32_t* __restrict__ truemask = A.truemask;
    uint32_t* __restrict__ tile_count = A.tile_count;
    const uint32_t paired = A.paired, m = A.m, tile_end = A.tile_end;
    const uint64_t kmask = A.kmask;

    // the probe in flight: it belongs to the PREVIOUS tile of this wave
    uint64_t pk = 0;
    u32x4 ps0 = {0, 0, 0, 0}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0}, ps3 = {0, 0, 0, 0};
    uint32_t pidx = 0, psrc = 0;
    bool pon = false;
    // second round of the same tile, only issued when it has more than 64 survivors (about every third tile at
    // config 2; survivors 129.. go to the leftover list)
    uint64_t pk2 = 0;
    u32x4 pt0 = {0, 0, 0, 0}, pt1 = {0, 0, 0, 0}, pt2 = {0, 0, 0, 0}, pt3 = {0, 0, 0, 0};
    uint32_t pidx2 = 0, psrc2 = 0;
    bool pon2 = false, extra = false;  // extra is wave-uniform
    uint32_t prev_t = 0xFFFFFFFFu, prev_a = 0, prev_la = 0, prev_word0 = 0;

    // a found slot -> number of candidates of position p of read a (and the selfrep side effect)
    auto count_slot = [&](u32x4 s, uint32_t a, uint32_t la, uint32_t p) __attribute__((always_inline)) -> uint32_t {
        uint32_t n = 0;
        if (p > 0) {
            if (s.w & SLOT_SINGLE) {
                if (s.z == a) note_selfrep(selfrep, a, p, n_selfrep);
            } else {
                for (uint32_t j = 0; j < s.w; ++j)
                    if (chain[s.z + j] == a) note_selfrep(selfrep, a, p, n_selfrep);
            }
        }
        for_each_candidate<STREAM>(chain, len, paired, s.z, s.w, a, la - p, [&](uint32_t, uint32_t, uint32_t) { ++n; });
        return n;
    };
    // Leftover list: a dependent load in the steady state would force `s_waitcnt vmcnt(0)` and drain the
    // whole look-ahead (the tile words come from HBM), so hard positions are appended here with plain
    // stores.  Only when the list is full does the wave resolve them in place.
    uint2* __restrict__ my_left = A.left + (size_t)(blockIdx.x * nwaves + wave) * LEFT_CAP;
    uint32_t left_n = 0;  // wave-uniform
    auto defer = [&](bool want, uint32_t tile, uint32_t lane_s) __attribute__((always_inline)) -> bool {
        // returns true when the position was deferred; false = list full, caller resolves it in place
        // (every lane of the wave must call this together: left_n has to stay wave-uniform)
        const uint64_t bal = __ballot(want);
        if (bal == 0) return false;
        const uint32_t cnt = (uint32_t)__popcll(bal);
        if (left_n + cnt > LEFT_CAP) return false;
        if (want) my_left[left_n + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = uint2{tile, lane_s};
        left_n = __builtin_amdgcn_readfirstlane(left_n + cnt);
        return want;
    };

    // Hand-scheduled software pipeline over this wave's tiles k = t, t+stride, ...  The three load
    // streams of the steady state are issued with inline asm so that the waits can be COUNTED
    // (hipcc falls back to vmcnt(0) at every use here, which would serialise the probes):
    //   R(k+2)  record of tile k+2     2 x dwordx4   issued in pass k, used from pass k+1
    //   W(k+1)  words of tile k+1      1 x dwordx4   issued in pass k, used in pass k+1
    //   P(k)    table slots of tile k  4 x dwordx4   issued at the end of pass k, used in pass k+1
    // vm ops complete in issue order, and vmcnt(N) waits until at most N are outstanding, so a wait
    // needs N <= (ops issued after the one wanted).  Stores and the rare compiler-tracked loads only
    // add younger ops, so the counts below are lower bounds: safe.  Every steady-state load is
    // unconditional (clamped tile index / slot 0 for "no survivor"), which keeps the counts exact.
    const uint32_t t_last = tile_end - 1;  // launch guarantees tile_end > tile_begin
    uint32_t t = __builtin_amdgcn_readfirstlane(A.tile_begin + blockIdx.x * nwaves + wave);
    if (t >= tile_end) return;
    auto ld16 = [](const void* ptr) __attribute__((always_inline)) -> u32x4 {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    TileRec rec = tiles[t];  // tile k (ordinary load)
    // in flight: R = record of the next tile; W = words of the tile to filter next.  W has two
    // landing register sets used alternately (the pass body is instantiated twice): the words of
    // tile k stay live in their landing registers through the whole pass while W(k+1) lands in the
    // other set, so no register whose load is still in flight ever has to be copied.
    u32x4 r_lo, r_hi, wa, wb = {0, 0, 0, 0};
    {
        const TileRec* rp = &tiles[__builtin_amdgcn_readfirstlane(min(t + stride, t_last))];
        r_lo = ld16(rp);
        r_hi = ld16(reinterpret_cast<const char*>(rp) + 16);
        wa = ld16(&words[rec.wabs + lane]);  // (tail padding keeps every lane in bounds)
        ps0 = ld16(&table[0]);               // dummy probe: same queue shape as the steady state
        ps1 = ld16(&table[1]);
        ps2 = ld16(&table[2]);
        ps3 = ld16(&table[3]);
    }
#ifdef PO_STAMPS
    unsigned long long acc_s[6] = {0, 0, 0, 0, 0, 0};
#endif
    // one pass; returns false after the last tile has been retired
    auto pass = [&](u32x4& wcur, u32x4& wnext) __attribute__((always_inline)) -> bool {
        const bool have_tile = t < tile_end;  // wave-uniform
        const uint32_t tn = t + stride;
        unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;
        (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)st5;
        PO_STAMP(st0);
        // ---- R and W of the previous pass have the 4 (8 with a second round) probe loads behind them
        // (ONE asm statement, the choice is made inside it: two statements in the arms of an `if` make hipcc copy
        // the landing registers in front of the branch -- a read of registers whose loads are still in flight)
        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lpo_w4_%=\n\ts_waitcnt vmcnt(8)\n\ts_branch .Lpo_wd_%=\n"
                     ".Lpo_w4_%=:\n\ts_waitcnt vmcnt(4)\n.Lpo_wd_%=:"
                     : "+v"(r_lo), "+v"(r_hi), "+v"(wcur) : "s"(extra ? 1u : 0u) : "memory", "scc");
        PO_STAMP(st1);
        TileRec rec1;  // tile k+1
        rec1.wabs = ((uint64_t)__builtin_amdgcn_readfirstlane(r_lo.y) << 32) | __builtin_amdgcn_readfirstlane(r_lo.x);
        rec1.wread = ((uint64_t)__builtin_amdgcn_readfirstlane(r_lo.w) << 32) | __builtin_amdgcn_readfirstlane(r_lo.z);
        rec1.read = __builtin_amdgcn_readfirstlane(r_hi.x);
        rec1.la = __builtin_amdgcn_readfirstlane(r_hi.y);
        rec1.word0 = __builtin_amdgcn_readfirstlane(r_hi.z);
        rec1.pad = 0;
        {
            const TileRec* rp = &tiles[__builtin_amdgcn_readfirstlane(min(tn + stride, t_last))];
            r_lo = ld16(rp);
            r_hi = ld16(reinterpret_cast<const char*>(rp) + 16);
            wnext = ld16(&words[rec1.wabs + lane]);
        }
        const uint64_t w0 = ((uint64_t)wcur.y << 32) | wcur.x, w1 = ((uint64_t)wcur.w << 32) | wcur.z;
        // ---- filter this tile
        uint32_t hitmask = 0;
        const uint32_t p0 = (rec.word0 + lane) * W;
        if (have_tile && rec.la >= m && p0 <= rec.la - m) {
            hitmask = filter_tile<BITS, FULLK>(s_bloom, A.bloom_log2, w0, w1, kmask);
            const uint32_t nvalid = rec.la - m - p0 + 1;  // positions whose suffix/containment can reach min_length
            if (nvalid < (uint32_t)W) hitmask &= (1u << nvalid) - 1u;
        }
        PO_STAMP(st2);
        // ---- the previous tile's probes have only this pass's R and W (3 loads) behind them
        // (both rounds at once: the second is consumed right after the first)
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(ps0), "+v"(ps1), "+v"(ps2), "+v"(ps3), "+v"(pt0), "+v"(pt1), "+v"(pt2), "+v"(pt3)
                     : : "memory");
        PO_STAMP(st3);
        if (prev_t != 0xFFFFFFFFu) {
            // one survivor of the previous tile: linear probing over its group -- the first slot that is empty (key
            // absent) or holds the key settles it (the all-ones key sits alone in the extra slot: empty or equal,
            // its first slot always settles).  Every lane of the wave calls this (defer() is a wave operation).
            auto settle = [&](uint64_t key, uint32_t home, uint32_t src, bool on, const u32x4& q0, const u32x4& q1,
                              const u32x4& q2, const u32x4& q3) __attribute__((always_inline)) -> uint32_t {
                const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
                auto stops = [&](const u32x4& q) __attribute__((always_inline)) -> bool {
                    return q.w == 0 || (q.x == klo && q.y == khi);
                };
                const bool h0 = stops(q0), h1 = stops(q1), h2 = stops(q2), h3 = stops(q3);
                u32x4 s = q3;
                if (h2) s = q2;
                if (h1) s = q1;
                if (h0) s = q0;
                const bool settled = h0 | h1 | h2 | h3;
                // needs a slot beyond the group, or heads a chain of several reads -> leftover list
                const bool hard = on && (!settled || (s.w != 0 && !(s.w & SLOT_SINGLE)));
                const bool deferred = defer(hard, prev_t, src);
                uint32_t n = 0;
                if (on && !deferred) {
                    if (!settled) {  // (leftover list full) resolve in place
                        uint32_t j = (home + PROBE_GROUP - 1u) & tmask;
                        while (s.w != 0 && (((uint64_t)s.y << 32) | s.x) != key) {
                            j = (j + 1u) & tmask;
                            s = *reinterpret_cast<const u32x4*>(&table[j]);
                        }
                    }
                    if (s.w != 0) n = count_slot(s, prev_a, prev_la, (prev_word0 + (src >> 8)) * W + (src & 255u));
                    if (n) atomicOr(&tm[src >> 8], 1u << (src & 255u));
                }
                return n;
            };
            uint32_t n = settle(pk, pidx, psrc, pon, ps0, ps1, ps2, ps3);
            if (extra) n += settle(pk2, pidx2, psrc2, pon2, pt0, pt1, pt2, pt3);
            wave_lds_fence();
            truemask[(size_t)prev_t * WAVE + lane] = tm[lane];  // each lane owns the mask of its word
            tm[lane] = 0;
            // a lane settles one survivor with one candidate, unless it resolved a chain in place (leftover list full):
            // the tile's count is a popcount of a ballot almost always
            n = __any(n > 1u) ? wave_sum(n) : (uint32_t)__popcll(__ballot(n != 0u));
            if (lane == 0) tile_count[prev_t] = n;
        }
        PO_STAMP(st4);
        if (!have_tile) return false;
        // ---- compact this tile's survivors: survivor number r goes to lane r
        const uint32_t nh = __popc(hitmask);
        const uint32_t incl = wave_incl_scan(nh);
        const uint32_t total = read_last_lane(incl);
        uint32_t rank = incl - nh;
        while (hitmask && rank < 2u * WAVE) {  // the queue only names the position: (lane, s)
            const uint32_t sft = __ffs(hitmask) - 1;
            hitmask &= hitmask - 1;
            qs[rank] = (lane << 8) | sft;
            ++rank;
        }
        while (__any(hitmask != 0)) {  // survivors beyond the 128th: defer them, one per lane per round
            const bool want = hitmask != 0;
            const uint32_t sft = want ? __ffs(hitmask) - 1 : 0;
            const bool ok = defer(want, t, (lane << 8) | sft);  // all lanes take part, every round
            if (!__any(ok)) break;                              // leftover list full
            if (ok) hitmask &= hitmask - 1;
        }
        uint32_t ovf_n = 0;
        while (hitmask) {  // leftover list full: resolve them now
            const uint32_t sft = __ffs(hitmask) - 1;
            hitmask &= hitmask - 1;
            const uint64_t kmer = funnel(w0, w1, sft * BITS) & kmask;
            uint32_t z = 0, w = 0;
            table_probe(table, tbits, kmer, z, w);
            if (w) {
                const uint32_t nn = count_slot(u32x4{0, 0, z, w}, rec.read, rec.la, p0 + sft);
                if (nn) {
                    atomicOr(&tm[lane], 1u << sft);  // picked up when this tile is retired
                    ovf_n += nn;
                }
            }
        }
        wave_lds_fence();
        // ---- request the group of table slots of this lane's survivor: always 4 loads per lane (lanes without one
        // read group 0), and 4 more for survivors 65..128 when the tile has that many
        auto request = [&](uint32_t src, bool on, uint64_t& key, uint32_t& home, u32x4& q0, u32x4& q1, u32x4& q2,
                           u32x4& q3) __attribute__((always_inline)) {
            // the K-mer is cut out of the owner lane's two words, fetched across the wave (ds_bpermute)
            const uint32_t sl = src >> 8;
            const uint32_t x0 = __shfl((uint32_t)w0, sl, WAVE), x1 = __shfl((uint32_t)(w0 >> 32), sl, WAVE);
            const uint32_t x2 = __shfl((uint32_t)w1, sl, WAVE), x3 = __shfl((uint32_t)(w1 >> 32), sl, WAVE);
            // 64 bits at bit offset sh of the four dwords, in 32-bit pieces: three selects and two v_alignbit (which
            // takes its shift mod 32) instead of three 64-bit shifts, which issue at a quarter of the rate
            const uint32_t sh = (src & 255u) * BITS;
            const bool up = sh >= 32u;
            uint32_t lo = __builtin_amdgcn_alignbit(up ? x2 : x1, up ? x1 : x0, sh);
            uint32_t hi = __builtin_amdgcn_alignbit(up ? x3 : x2, up ? x2 : x1, sh);
            if constexpr (!FULLK) {
                lo &= (uint32_t)kmask;
                hi &= (uint32_t)(kmask >> 32);
            }
            const uint64_t kmer = ((uint64_t)hi << 32) | lo;
            uint32_t idx = narrow_home((lo ^ ((hi << 13) | (hi >> 19))) * 0x9E3779B1u, tbits);
            if (kmer == KEY_EMPTY) idx = tmask + 1u;  // the all-ones key lives in the extra slot (three padding slots follow it)
            if (!on) idx = 0;
            key = kmer;
            home = idx;
            const Slot* g = &table[idx];
            q0 = ld16(g);
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(q1) : "v"(g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(q2) : "v"(g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(q3) : "v"(g) : "memory");
        };
        pon = lane < total;
        psrc = pon ? qs[lane] : 0u;
        request(psrc, pon, pk, pidx, ps0, ps1, ps2, ps3);
        extra = total > (uint32_t)WAVE;
        if (extra) {
            pon2 = lane + (uint32_t)WAVE < total;
            psrc2 = pon2 ? qs[WAVE + lane] : 0u;
            request(psrc2, pon2, pk2, pidx2, pt0, pt1, pt2, pt3);
        }
        wave_lds_fence();  // queue reads done before the next tile overwrites it
        if (__any(ovf_n != 0)) {  // (leftover list was full) counts resolved in place join the tile later
            const uint32_t e = wave_sum(ovf_n);
            if (lane == 0) atomicAdd(&A.tile_extra[t], e);
        }
        prev_t = t;
        prev_a = rec.read;
        prev_la = rec.la;
        prev_word0 = rec.word0;
        t = tn;
        rec = rec1;
#ifdef PO_STAMPS
        PO_STAMP(st5);
        acc_s[0] += st1 - st0;  // wait for record + words
        acc_s[1] += st2 - st1;  // look-ahead issue + filter
        acc_s[2] += st3 - st2;  // wait for probes
        acc_s[3] += st4 - st3;  // consume + retire
        acc_s[4] += st5 - st4;  // compaction + probe issue
        acc_s[5] += 1;
#endif
        return true;
    };
    while (pass(wa, wb) && pass(wb, wa)) {
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads of the last pass
    if (lane == 0) A.left_cnt[blockIdx.x * nwaves + wave] = left_n;
#ifdef PO_STAMPS
    if (A.dbg && lane == 0) {
        unsigned long long* o = A.dbg + (size_t)(blockIdx.x * nwaves + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = acc_s[k];
    }
#endif
}

// Settle the positions the scan waves deferred (k_scan_probe: third table slot needed, chain of several
// reads, more than NPEND survivors in a lane): one thread each, ordinary dependent loads, results
// added to the tile's truemask bit and candidate count.  Runs after k_scan_probe has retired every tile.
// clean != 0: the list's counter is zero again when the kernel ends (the pieces of a streamed step: no reset launch per piece)
template <int BITS, bool STREAM = false>
__global__ void k_scan_fixup(const ScanArgs A, uint32_t n_waves, uint32_t clean) {
    constexpr int W = 64 / BITS;
    // one workgroup of 256 threads per scan wave's list: every position is a chain of dependent loads, so the
    // kernel's time is a latency times the number of rounds (64 threads per list: 16 rounds for a full list;
    // one thread per slot of every list's capacity: the launch of 32 k mostly empty workgroups cost more than that)
    const uint32_t wv = blockIdx.x;
    if (wv >= n_waves) return;
    const uint32_t cnt = min(A.left_cnt[wv], LEFT_CAP);
    const uint2* __restrict__ list = A.left + (size_t)wv * LEFT_CAP;
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
        const uint2 e = list[k];
        const uint32_t t = e.x, ln = e.y >> 8, sft = e.y & 255u;
        if (t < A.tile_begin || t >= A.tile_end || ln >= (uint32_t)WAVE || sft >= (uint32_t)W) continue;  // (defensive)
        const TileRec rec = A.tiles[t];
        const uint64_t w0 = A.words[rec.wabs + ln], w1 = A.words[rec.wabs + ln + 1];
        const uint64_t kmer = funnel(w0, w1, sft * BITS) & A.kmask;
        uint32_t z = 0, w = 0;
        table_probe(A.table, A.tbits, kmer, z, w);
        if (!w) continue;
        const uint32_t a = rec.read, p = (rec.word0 + ln) * W + sft;
        if (p > 0) {
            if (w & SLOT_SINGLE) {
                if (z == a) note_selfrep(A.selfrep, a, p, A.n_selfrep);
            } else {
                for (uint32_t j = 0; j < w; ++j)
                    if (A.chain[z + j] == a) note_selfrep(A.selfrep, a, p, A.n_selfrep);
            }
        }
        uint32_t n = 0;
        for_each_candidate<STREAM>(A.chain, A.len, A.paired, z, w, a, rec.la - p, [&](uint32_t, uint32_t, uint32_t) { ++n; });
        if (n) {
            atomicOr(&A.truemask[(size_t)t * WAVE + ln], 1u << sft);
            atomicAdd(&A.tile_count[t], n);
        }
    }
    if (clean) {
        __syncthreads();   // (every thread has read its entries' count)
        if (threadIdx.x == 0) A.left_cnt[wv] = 0;
    }
}

__global__ void k_add_extra(uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_extra, uint32_t t0, uint32_t t1) {
    const uint32_t t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t < t1 && tile_extra[t]) tile_count[t] += tile_extra[t];
}

// Scan pass 2 (fill): one wave per FILL_TILES consecutive tiles, ordinary grid.  Positions with
// candidates come from truemask; each is probed again (L2-resident table) and its (a, p, b) triples
// are written in ascending (tile, p), chain order (ascending b).  Consecutive tiles own consecutive
// candidate ranges (tile_off is a running sum), so the wave writes one contiguous range starting at
// tile_off[first tile]; pooling the hits of several tiles fills the 64-entry probe rounds and turns
// four short latency chains into one.
#ifndef PO_FILL_TILES
#define PO_FILL_TILES 8
#endif
constexpr int FILL_TILES = PO_FILL_TILES;

template <int BITS, bool STREAM = false>
__global__ __launch_bounds__(256) void k_scan_fill(const ScanArgs A, const CandGuard G) {
    constexpr int W = 64 / BITS;
    __shared__ uint32_t q_src[4 * WAVE];
    __shared__ TileRec q_rec[4 * FILL_TILES];
    if (G.overflow()) return;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t* qs = q_src + wave * WAVE;
    TileRec* qr = q_rec + wave * FILL_TILES;
    const uint32_t t0 = __builtin_amdgcn_readfirstlane(A.tile_begin + (blockIdx.x * 4 + wave) * FILL_TILES);
    if (t0 >= A.tile_end) return;
    // The kernel is a chain of dependent loads per wave (SQ_WAIT_ANY: 73 % of its wave-cycles), so its time is the
    // number of waves per wave slot times the links of that chain.  Masks and tile records do not depend on each
    // other and are requested together; a hit fetches its own two words (the lanes used to hold the words of
    // every tile: 4 registers per tile, which capped the tiles per wave at 4).
    uint32_t hm[FILL_TILES], rk[FILL_TILES];
#pragma unroll
    for (int i = 0; i < FILL_TILES; ++i) hm[i] = t0 + i < A.tile_end ? A.truemask[(size_t)(t0 + i) * WAVE + lane] : 0u;
    if (lane < (uint32_t)FILL_TILES) qr[lane] = A.tiles[min(t0 + lane, A.tile_end - 1)];
    uint32_t out = A.tile_off[t0];
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < FILL_TILES; ++i) {
        const uint32_t nh = __popc(hm[i]);
        const uint32_t incl = wave_incl_scan(nh);
        rk[i] = total + incl - nh;  // rank of this lane's first hit of tile i among the wave's hits
        total += read_last_lane(incl);
    }
    if (total == 0) return;
    for (uint32_t r0 = 0; r0 < total; r0 += WAVE) {
#pragma unroll
        for (int i = 0; i < FILL_TILES; ++i) {
            while (hm[i] && rk[i] < r0 + WAVE) {
                const uint32_t sft = __ffs(hm[i]) - 1;
                hm[i] &= hm[i] - 1;
                qs[rk[i] - r0] = ((uint32_t)i << 16) | (lane << 8) | sft;
                ++rk[i];
            }
        }
        wave_lds_fence();  // (also orders the tile records written above before their first use)
        const bool has = r0 + lane < total;
        const uint32_t src = has ? qs[lane] : 0u;
        const TileRec rec = qr[src >> 16];
        wave_lds_fence();
        const uint32_t ln = (src >> 8) & 255u, sft = src & 255u;
        uint32_t z = 0, w = 0;
        if (has) {
            const uint64_t w0 = A.words[rec.wabs + ln], w1 = A.words[rec.wabs + ln + 1];
            table_probe(A.table, A.tbits, funnel(w0, w1, sft * BITS) & A.kmask, z, w);
        }
        const uint32_t a = rec.read, la = rec.la;
        const uint32_t p = (rec.word0 + ln) * W + sft;
        uint32_t n = 0;
        if (w) for_each_candidate<STREAM>(A.chain, A.len, A.paired, z, w, a, la - p, [&](uint32_t, uint32_t, uint32_t) { ++n; });
        const uint32_t inc = wave_incl_scan(n);
        uint32_t off = out + inc - n;
        if (n) {
            for_each_candidate<STREAM>(A.chain, A.len, A.paired, z, w, a, la - p, [&](uint32_t b, uint32_t, uint32_t) {
                A.cand_a[off] = a;
                A.cand_p[off] = p;
                A.cand_b[off] = b;
                ++off;
            });
        }
        out += read_last_lane(inc);
    }
}

// ----------------------------------------------------------------------------------------
// verify order.  Each b is read by ~35 different a's (the reads that overlap it from the left), in an order
// unrelated to where the reads come from, so every XCD's L2 keeps missing (hit rate 25 %: 9.5 GB of fabric
// traffic for 375 MB of reads).  Reads that cover the same stretch of the genome have nearly the same
// candidate list, hence the same top-ranked read in it: that rank is a locality label for free.
// Sorting the a's by label (counting sort) puts neighbours next to each other; k_verify_a hands XCD x the
// x-th eighth of the sorted list, so neighbours meet in one L2.
// ----------------------------------------------------------------------------------------
// 16 lanes per read: label[i] = the highest rank among read r_begin + i and the mirrors b ^ 1 of its candidates,
// in the order that picked the canonical candidates (mirror_rank: the index for whole-set calls, the scrambled
// rank for sharded calls).  A read keeps exactly the candidates that rank above it, so the top-ranked read of a
// neighbourhood is on the list of every one of its neighbours: they all get the same label.
// The walk over every candidate's b also does k_defer_split's job in a streamed step (defer != nullptr): a candidate
// whose b has not arrived (b >= b_limit: a containment, the only kind the scan keeps of those) goes on the deferred list.
struct DeferOut {
    const uint32_t* cand_p;
    uint32_t b_limit;
    uint4* list;          // Cand entries
    uint32_t cap;
    uint32_t* counter;
};
__global__ __launch_bounds__(256) void k_read_label(const uint32_t* __restrict__ read_tile0, const uint32_t* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ cand_b, uint32_t r_begin, uint32_t n_reads,
                                                    uint32_t paired, uint32_t* __restrict__ label, const DeferOut defer,
                                                    const CandGuard G) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 4, sub = t & 15u;
    if (i >= n_reads) return;  // (whole 16-lane groups leave together)
    if (G.overflow()) {        // (the sort behind this kernel still wants a label per read)
        if (sub == 0) label[i] = r_begin + i;
        return;
    }
    const uint32_t a = r_begin + i;
    const uint32_t seg0 = tile_off[read_tile0[a]], seg1 = tile_off[read_tile0[a + 1]];
    const bool reversed = (paired & 3u) == 3u;   // (a streamed step's order: ~index)
    auto rank = [&](uint32_t x) { return reversed ? ~x : mirror_rank(x, paired); };
    uint32_t best = rank(a);
    for (uint32_t c = seg0 + sub; c < seg1; c += 16) {
        const uint32_t b = cand_b[c];
        best = max(best, rank(b ^ 1u));
        if (defer.list && b >= defer.b_limit) {
            const uint32_t k = atomicAdd(defer.counter, 1u);
            if (k < defer.cap) defer.list[k] = uint4{a, defer.cand_p[c], b, 0u};
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 16));
    // the label is the top-ranked READ, whatever order ranked it: a scrambled rank (sharded calls) is turned back
    // into the read index, so that the sort's bins hold a few neighbouring indices in both modes -- binning the
    // scrambled rank's top bits put the 256 reads of an index block into one bin and lost the locality
    if (paired == 2u) best = (__brev(best & 0xFFFFFF00u) << 8) | (best & 0xFFu);
    if (reversed) best = ~best;
    if (sub == 0) label[i] = best;
}

// Counting sort of the reads by label >> shift, in ONE workgroup with the bins in LDS: 100 k device-scope atomics
// on a few thousand hot addresses cost 35-70 us (histogram and scatter each), the same on LDS a few.  The host
// picks shift so that the bins fit (neighbouring labels then share a bin: their clusters end up side by side).
// The workgroup only produces rank[i] = position of read i in bin order (coalesced stores; 100 k scattered
// stores from a single CU took 48 us); k_read_invert turns that into perm[rank] = read on the whole chip.
// Order inside a bin is whatever the atomics give: the verify result does not depend on workgroup order.
constexpr int SORT_BLOCK = 1024;
__global__ __launch_bounds__(SORT_BLOCK) void k_read_sort(const uint32_t* __restrict__ label, uint32_t n_reads, uint32_t shift,
                                                          uint32_t n_bins, uint32_t* __restrict__ rank) {
    extern __shared__ uint32_t s_bin[];          // n_bins counters, then SORT_BLOCK / 64 wave totals
    uint32_t* s_wave = s_bin + n_bins;
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < n_bins; b += SORT_BLOCK) s_bin[b] = 0;
    __syncthreads();
    // (one workgroup has little memory-level parallelism: keep 16 label loads in flight per thread)
    constexpr uint32_t UN = 16;
    for (uint32_t base = tid; base < n_reads; base += UN * SORT_BLOCK) {
        uint32_t v[UN];
#pragma unroll
        for (uint32_t k = 0; k < UN; ++k) {
            const uint32_t i = base + k * SORT_BLOCK;
            v[k] = i < n_reads ? label[i] : ~0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < UN; ++k)
            if (v[k] != ~0u) atomicAdd(&s_bin[v[k] >> shift], 1u);
    }
    __syncthreads();
    // exclusive scan of the bins: each thread owns a contiguous chunk
    const uint32_t per = (n_bins + SORT_BLOCK - 1) / SORT_BLOCK;
    const uint32_t lo = min(tid * per, n_bins), hi = min(lo + per, n_bins);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += s_bin[b];
    const uint32_t incl = wave_incl_scan(sum);
    if (lane_id() == WAVE - 1) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < (tid >> 6); ++w) run += s_wave[w];
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t v = s_bin[b];
        s_bin[b] = run;
        run += v;
    }
    __syncthreads();
    for (uint32_t base = tid; base < n_reads; base += UN * SORT_BLOCK) {
        uint32_t v[UN];
#pragma unroll
        for (uint32_t k = 0; k < UN; ++k) {
            const uint32_t i = base + k * SORT_BLOCK;
            v[k] = i < n_reads ? label[i] : ~0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < UN; ++k)
            if (v[k] != ~0u) rank[base + k * SORT_BLOCK] = atomicAdd(&s_bin[v[k] >> shift], 1u);
    }
}

__global__ void k_read_invert(const uint32_t* __restrict__ rank, uint32_t n_reads, uint32_t r_begin, uint32_t* __restrict__ perm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads && rank[i] < n_reads) perm[rank[i]] = r_begin + i;
}

// ----------------------------------------------------------------------------------------
// verify: packed exact compare of a[p : p+n) against b[0 : n), n = min(la-p, lb)
// ----------------------------------------------------------------------------------------
// One workgroup per a-side read: a's packed words are staged in LDS once -- together with one 16-byte record per
// candidate (position, compare length, row mask, where b starts) -- and every candidate of a streams only its b
// side from global memory, 16 bytes per lane per block (global_load_dwordx4), three 256-byte blocks per step.
// 16 lanes form a group that walks one candidate; the four groups of a wave advance independently through a
// flattened loop, and a group whose candidate has finished or mismatched draws the next one from a counter
// shared by the workgroup.  The window of a that faces b is cut out of adjacent LDS dwords with v_alignbit_b32.
// type: bit0 = suffix-prefix (A) candidate holds, bit1 = b wholly contained at p (B); 0 = mismatch.
// Do the exception records of a inside [p, p+n) equal those of b inside [0, n) (same offsets, same
// bytes)?  Records are sorted by position; they are rare, one lane walks them.
__device__ inline bool exceptions_equal(const uint32_t* __restrict__ exc_off, const uint32_t* __restrict__ exc_pos,
                                        const uint8_t* __restrict__ exc_byte, uint32_t a, uint32_t p, uint32_t b,
                                        uint32_t n) {
    uint32_t ia = exc_off[a];
    const uint32_t ea = exc_off[a + 1];
    uint32_t ib = exc_off[b];
    const uint32_t eb = exc_off[b + 1];
    if (ia == ea && ib == eb) return true;
    {  // first record of a at or after p
        uint32_t lo = ia, hi = ea;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (exc_pos[mid] < p) lo = mid + 1; else hi = mid;
        }
        ia = lo;
    }
    for (;;) {
        const bool ha = ia < ea && exc_pos[ia] - p < n;
        const bool hb = ib < eb && exc_pos[ib] < n;
        if (!ha && !hb) return true;
        if (ha != hb) return false;
        if (exc_pos[ia] - p != exc_pos[ib] || exc_byte[ia] != exc_byte[ib]) return false;
        ++ia;
        ++ib;
    }
}

#ifndef PO_VER_GROUP
#define PO_VER_GROUP 16
#endif
constexpr int VER_GROUP = PO_VER_GROUP;  // lanes per candidate (16 bytes of b each per block); 8 lanes x 4 blocks measures 3 % faster, 32 lanes 15 % slower
#ifndef PO_VER_BLOCK
#define PO_VER_BLOCK 256
#endif
constexpr int VER_BLOCK = PO_VER_BLOCK;  // threads per verify workgroup (one a-side read); 128 and 256 measure the same, 64 and 512 worse
#ifndef PO_VER_BLOCKS
#define PO_VER_BLOCKS 3
#endif
#ifndef PO_VER_FIRST
#define PO_VER_FIRST PO_VER_BLOCKS  // blocks in a candidate's first step
#endif
constexpr int VER_BLOCKS = PO_VER_BLOCKS;  // 256-byte blocks per group per step after the first step

// SCRAMBLED: the canonical-pair order of sharded calls (keep_bits); a compile-time choice here because the
// extra rank arithmetic sits on the per-candidate setup path (index order: 1.32 ms, run-time select: 1.37 ms).
#ifdef PO_VSTAMPS
// diagnostic build (-DPO_VSTAMPS): where a verify wave spends its cycles (s_memtime = shader clock).
// [0] workgroup start -> first compare iteration (read geometry, staging of a, first metadata), [1] waiting
// for the b loads of an iteration, [2] LDS reads + compare, [3] ballot + bookkeeping + start of the next
// candidate, [4] wave lifetime, [5] iterations, [6] waves, [7] group-steps with work
__device__ unsigned long long g_vstamps[64 * 8];
#define VST(...) __VA_ARGS__
#else
#define VST(...)
#endif

// The compare loop of k_verify_a, compiled once for a in LDS and once for a in global memory (reads too long
// for LDS).  With one body the compiler folds the two a-pointers into a generic pointer: five flat_load_dword
// and ten 64-bit address instructions per 256-byte block instead of five ds_read_b32 off one 32-bit address.
// STAGED: the records {p, b, len[b], first word of b} of a read's candidates are put into LDS by the whole
// workgroup while a is staged, so that a group starting a candidate reads one 16-byte record instead of running
// a two-deep prefetch of four dependent global loads (the start/finish code is half of this kernel's
// instructions, and the kernel is issue bound).
constexpr int VREC_CAP = 512;  // records per staging batch (8 KB of LDS)
// (longest-only selection inside a read's candidate list, defined with k_select_local below: k_verify_a can run it in its
// epilogue -- the workgroup that verified a read's candidates owns the list)
constexpr uint32_t SEL_CAP = 512;  // LDS table entries per wave (load <= 1/2)
__device__ __forceinline__ void select_read_list(uint32_t seg0, uint32_t seg1, const uint32_t* __restrict__ cand_p,
                                                 const uint32_t* __restrict__ cand_b, uint8_t* __restrict__ type,
                                                 uint32_t* __restrict__ selfrep, uint32_t* __restrict__ n_deferred,
                                                 uint32_t* key, uint32_t* mn, uint32_t lane);
struct __attribute__((aligned(16))) VRec {
    uint32_t pk;  // p | (keep & 1) << 31        p, n < 2^31: the top bits carry `keep` (which rows the candidate can
    uint32_t b;   //                             give, keep_bits) -- worked out once by the staging thread instead of by
    uint32_t nk;  // n | (keep >> 1) << 31       every wave whose group starts the candidate; n = min(la - p, len[b])
    uint32_t wo;  // first 64-bit word of b in words[] (the host checks that it fits 32 bits)
};

// fetch-and-increment of an LDS counter by the lanes that call it.  Spelled as the instruction: for a plain
// atomicAdd hipcc builds the wave-aggregated form (two v_mbcnt, a compare, one ds_add_rtn of the lane count,
// v_readfirstlane, an add) -- right for 64 callers, six extra instructions in the verify kernel's completion path
// for the one to four that call here
__device__ __forceinline__ uint32_t lds_draw(uint32_t* counter) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds_u32*)counter;
    uint32_t old, one = 1u;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(one) : "memory");
    return old;
}

// value of the group's first lane in every lane of the group: a 16-lane group is a DPP row (row_newbcast:0, one
// VALU op) -- no ds_bpermute round trip behind the LDS atomic that drew the value
__device__ __forceinline__ uint32_t group_bcast0(uint32_t v, uint32_t gshift) {
    if constexpr (VER_GROUP == 16) {
        (void)gshift;
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150, 0xf, 0xf, false);
    } else {
        return (uint32_t)__shfl((int)v, (int)gshift);
    }
}

template <int BITS, bool SCRAMBLED, bool IN_LDS, bool STAGED, bool STREAM = false>
__device__ __forceinline__ void verify_run(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                           const uint32_t* __restrict__ len, const uint32_t* __restrict__ cand_p,
                                           const uint32_t* __restrict__ cand_b, uint32_t paired,
                                           const uint32_t* __restrict__ exc_off, const uint32_t* __restrict__ exc_pos,
                                           const uint8_t* __restrict__ exc_byte, uint8_t* __restrict__ type,
                                           const uint32_t* __restrict__ s_a, const uint32_t* __restrict__ ga32,
                                           const VRec* __restrict__ s_rec, uint32_t* s_next, uint32_t a, uint32_t la,
                                           uint32_t seg0, uint32_t seg1
                                           VST(, unsigned long long vt_start, unsigned long long (&vt)[8])) {
    constexpr int W = 64 / BITS;
    (void)W;
    // a is addressed as 32-bit words from here on: the window of a that faces b starts at an
    // arbitrary bit, and v_alignbit_b32 extracts 32 bits at any offset from two adjacent dwords
    const uint32_t sub = threadIdx.x & (VER_GROUP - 1);
    const uint32_t gshift = (lane_id() / VER_GROUP) * VER_GROUP;
    constexpr uint32_t NGROUPS = VER_BLOCK / VER_GROUP;

    // A group works on candidate c while the metadata of its next two (cA, then cB) is on the way.  Which
    // candidates those are is decided when they are requested -- cB comes from a workgroup-wide counter -- so a
    // group that drew short candidates (46 % die in their first block) simply draws more often: with a fixed
    // round-robin share the workgroup waited for the group that happened to get the long ones.
    uint32_t c = seg0;
    uint32_t cA = seg0 + threadIdx.x / VER_GROUP, cB = cA + NGROUPS;
    bool have = cA < seg1;
    // per candidate: nbits = compared bits, q = first dword of a's window, sh = its bit offset,
    // d = this lane's first dword of b in the current step, keep = rows it can give
    uint32_t nbits = 0, sh = 0, q = 0, d = 0, keep = 0, nblk_var = 1, cur_p = 0, cur_b = 0;
    uint32_t nd = 0;  // dwords of b to compare = ceil(nbits / 32): block j of a step is this lane's iff d + j * BLK < nd
    // every step covers VER_BLOCKS blocks unless a shorter first step is configured (PO_VER_FIRST): since the
    // locality order keeps b in L2, reading 768 bytes of a candidate that dies in its first 256 costs less than the
    // extra iteration the survivors would need (1.09 -> 1.06 ms)
    constexpr bool UNIFORM_STEPS = PO_VER_FIRST == PO_VER_BLOCKS;
    const uint32_t* B = reinterpret_cast<const uint32_t*>(words);
    // (!STAGED only) Candidate metadata runs two candidates ahead of the compare loop, so that a group starting a
    // new candidate has (p, b, len[b], woff[b]) in registers already: m0 = the next candidate to
    // start (complete), m1 = the one after (p, b only; its len/woff are requested when it moves up).
    const uint32_t c_last = seg1 - 1;
    uint32_t m0p = 0, m0b = 0, m0l = 0, m1p = 0, m1b = 0;
    uint64_t m0w = 0;
    if constexpr (!STAGED) {
        const uint32_t c0 = min(cA, c_last), c1 = min(cB, c_last);
        m0p = cand_p[c0];
        m0b = cand_b[c0];
        m1p = cand_p[c1];
        m1b = cand_b[c1];
        m0l = len[m0b];
        m0w = woff[m0b];
    }
    auto init = [&]() __attribute__((always_inline)) {
        uint32_t p, b, lb;
        uint64_t wo;
        if constexpr (STAGED) {
            // (seg0 .. seg1 is the staged batch; s_next hands out absolute candidate indices)
            c = cA;
            const VRec r = s_rec[c - seg0];
            p = r.pk;   // (packed: see VRec)
            b = r.b;
            lb = r.nk;
            wo = r.wo;
            uint32_t drawn = 0;
            if (sub == 0) drawn = lds_draw(s_next);
            cA = group_bcast0(drawn, gshift);  // lane 0 of the group drew for all 16
        } else {
            p = m0p;
            b = m0b;
            lb = m0l;
            wo = m0w;
            // advance the look-ahead: requests only, nothing here is needed before the next candidate
            c = cA;
            cA = cB;
            m0p = m1p;
            m0b = m1b;
            m0l = len[m1b];
            m0w = woff[m1b];
            uint32_t drawn = 0;
            if (sub == 0) drawn = lds_draw(s_next);
            cB = group_bcast0(drawn, gshift);  // lane 0 of the group drew for all 16
            const uint32_t c2 = min(cB, c_last);
            m1p = cand_p[c2];
            m1b = cand_b[c2];
        }
        if constexpr (STAGED) {
            keep = (p >> 31) | ((lb >> 31) << 1);
            p &= 0x7FFFFFFFu;
            nbits = (lb & 0x7FFFFFFFu) * BITS;
        } else {
            const uint32_t rem = la - p;
            keep = STREAM ? keep_bits<true>(a, b, rem, lb, paired) : keep_bits(a, b, rem, lb, SCRAMBLED ? 2u : (paired ? 1u : 0u));
            const uint32_t n = rem < lb ? rem : lb;
            nbits = keep ? n * BITS : 0;
        }
        cur_p = p;
        cur_b = b;
        nd = (nbits + 31u) >> 5;
        const uint64_t bitpos = (uint64_t)p * BITS;
        q = (uint32_t)(bitpos >> 5);
        sh = (uint32_t)(bitpos & 31);
        B = reinterpret_cast<const uint32_t*>(words + wo);
        d = 4 * sub;
        nblk_var = PO_VER_FIRST;  // first step (a wrong-haplotype candidate dies in its first 256 bytes);
                   // all loads of a step are issued before its first compare
    };
    // compare this lane's 16 bytes of b at dword dd with the facing window of a
    auto cmp16 = [&](uint32_t dd, u32x4 bv) __attribute__((always_inline)) -> uint32_t {
        uint32_t a0, a1, a2, a3, a4;
        if constexpr (IN_LDS) {
            a0 = s_a[ver_swz(q + dd)];
            a1 = s_a[ver_swz(q + dd + 1)];
            a2 = s_a[ver_swz(q + dd + 2)];
            a3 = s_a[ver_swz(q + dd + 3)];
            a4 = s_a[ver_swz(q + dd + 4)];
        } else {
            a0 = ga32[q + dd];
            a1 = ga32[q + dd + 1];
            a2 = ga32[q + dd + 2];
            a3 = ga32[q + dd + 3];
            a4 = ga32[q + dd + 4];
        }
        if constexpr (PO_VER_PAD > 0) valu_pad<PO_VER_PAD>(a0, sh);   // (measurement builds only)
        uint32_t x0 = __builtin_amdgcn_alignbit(a1, a0, sh) ^ bv.x;
        uint32_t x1 = __builtin_amdgcn_alignbit(a2, a1, sh) ^ bv.y;
        uint32_t x2 = __builtin_amdgcn_alignbit(a3, a2, sh) ^ bv.z;
        uint32_t x3 = __builtin_amdgcn_alignbit(a4, a3, sh) ^ bv.w;
        const uint32_t left = nbits - dd * 32;  // compared bits from this lane's first dword on (>= 1)
        if (left < 128) {                       // the range ends inside these 16 bytes: mask the tail
            x0 &= left >= 32 ? ~0u : ((1u << left) - 1u);
            x1 &= left >= 64 ? ~0u : (left > 32 ? ((1u << (left - 32)) - 1u) : 0u);
            x2 &= left >= 96 ? ~0u : (left > 64 ? ((1u << (left - 64)) - 1u) : 0u);
            x3 &= left > 96 ? ((1u << (left - 96)) - 1u) : 0u;
        }
        return x0 | x1 | x2 | x3;
    };
    constexpr uint32_t BLK = 4 * VER_GROUP;  // dwords per 256-byte block
    if (have) init();
    VST(unsigned long long vt_prev = __builtin_amdgcn_s_memtime(); vt[0] = vt_prev - vt_start;)
    while (__any(have)) {
        uint32_t diff = 0;
        VST(vt[5] += 1; vt[7] += __popcll(__ballot(have && sub == 0));)
        if (have) {
            u32x4 bv[VER_BLOCKS];
            const int32_t rem = (int32_t)(nd - d);  // dwords from this lane's first one to the end of the range (<= 0: none)
#pragma unroll
            for (int j = 0; j < VER_BLOCKS; ++j)  // b starts 16-byte aligned
                if ((uint32_t)j < (UNIFORM_STEPS ? (uint32_t)VER_BLOCKS : nblk_var) && rem > (int32_t)(j * BLK)) bv[j] = *reinterpret_cast<const u32x4*>(B + d + j * BLK);
            VST(__builtin_amdgcn_s_waitcnt(0); { const unsigned long long t = __builtin_amdgcn_s_memtime(); vt[1] += t - vt_prev; vt_prev = t; })
#pragma unroll
            for (int j = 0; j < VER_BLOCKS; ++j)
                if ((uint32_t)j < (UNIFORM_STEPS ? (uint32_t)VER_BLOCKS : nblk_var) && rem > (int32_t)(j * BLK)) diff |= cmp16(d + j * BLK, bv[j]);
        }
        const uint64_t bal = __ballot(diff != 0);
        VST({ const unsigned long long t = __builtin_amdgcn_s_memtime(); vt[2] += t - vt_prev; vt_prev = t; })
        if (have) {
            const bool mismatch = ((bal >> gshift) & (VER_GROUP >= 64 ? ~0ull : ((1ull << (VER_GROUP & 63)) - 1ull))) != 0;
            d += (UNIFORM_STEPS ? (uint32_t)VER_BLOCKS : nblk_var) * BLK;
            nblk_var = VER_BLOCKS;
            if (mismatch || d - 4 * sub >= nd) {  // group-uniform: candidate finished
                if (sub == 0) {
                    uint32_t t = mismatch ? 0u : keep;
                    // 2-bit reads with exception records (non-ACGT bytes, stored as code 0): the packed
                    // compare only proved the codes equal; the bytes are equal iff the exception
                    // records inside the compared range also agree
                    if (t && exc_off && !exceptions_equal(exc_off, exc_pos, exc_byte, a, cur_p, cur_b, nbits / BITS)) t = 0;
                    type[c] = (uint8_t)t;
                }
                have = cA < seg1;
                if (have) init();
            }
        }
        VST({ const unsigned long long t = __builtin_amdgcn_s_memtime(); vt[3] += t - vt_prev; vt_prev = t; })
    }
    VST(if (lane_id() == 0) {
        vt[4] = __builtin_amdgcn_s_memtime() - vt_start;
        vt[6] = 1;
        for (int k = 0; k < 8; ++k) atomicAdd(&g_vstamps[(blockIdx.x & 63) * 8 + k], vt[k]);
    })
}

template <int BITS, bool SCRAMBLED, bool STAGED, bool STREAM = false>
__global__ __launch_bounds__(VER_BLOCK) void k_verify_a(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                         const uint32_t* __restrict__ len,
                                                         const uint32_t* __restrict__ read_tile0,
                                                         const uint32_t* __restrict__ tile_off,
                                                         const uint32_t* __restrict__ cand_p,
                                                         const uint32_t* __restrict__ cand_b, uint32_t r_begin,
                                                         uint32_t lds_words, uint32_t paired,
                                                         const uint32_t* __restrict__ exc_off,
                                                         const uint32_t* __restrict__ exc_pos,
                                                         const uint8_t* __restrict__ exc_byte,
                                                         uint8_t* __restrict__ type,
                                                         const uint32_t* __restrict__ perm, uint32_t n_a, const CandGuard G,
                                                         uint32_t* __restrict__ sel_selfrep, uint32_t* __restrict__ sel_deferred) {
    constexpr int W = 64 / BITS;
    extern __shared__ __attribute__((aligned(16))) uint64_t s_a64[];
    if (G.overflow()) return;
    VST(const unsigned long long vt_start = __builtin_amdgcn_s_memtime(); unsigned long long vt[8] = {};)
    uint32_t a = r_begin + blockIdx.x;
    if (perm) {
        // workgroup i runs on XCD i mod 8: give XCD x the x-th eighth of the locality-sorted read list
        const uint32_t per = (n_a + 7u) >> 3;
        const uint32_t k = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if (k >= n_a || (blockIdx.x >> 3) >= per) return;
        a = perm[k];
        if (a - r_begin >= n_a) return;  // (cannot happen: perm is a permutation of the shard's reads)
    } else if (blockIdx.x >= n_a) {
        return;
    }
    // (the words of a are requested before the candidate range is known: the range hangs on two dependent loads,
    // read_tile0 -> tile_off, and a read without candidates is rare)
    const uint32_t t0 = read_tile0[a], t1 = read_tile0[a + 1];
    const uint32_t la = len[a];
    const uint32_t nwa = (la + W - 1) / W;
    const uint64_t* __restrict__ ga = words + woff[a];
    const bool in_lds = nwa + 3 <= lds_words;  // workgroup-uniform
    // dynamic LDS only (a static variable would move its base off the 16-byte boundary the records need):
    // lds_words (even) words of a, VREC_CAP candidate records, the draw counter
    VRec* s_rec = reinterpret_cast<VRec*>(s_a64 + ver_a_words(lds_words));
    uint32_t* s_next = reinterpret_cast<uint32_t*>(s_rec + VREC_CAP);
    constexpr uint32_t NGROUPS = VER_BLOCK / VER_GROUP;
    if (in_lds) {
        if constexpr (PO_VER_LDS_SWZ) {
            uint32_t* s_w = reinterpret_cast<uint32_t*>(s_a64);
            for (uint32_t i = threadIdx.x; i < nwa + 3; i += VER_BLOCK) {
                const uint64_t v = ga[i];
                s_w[ver_swz(2u * i)] = (uint32_t)v;
                s_w[ver_swz(2u * i + 1u)] = (uint32_t)(v >> 32);
            }
        } else {
            for (uint32_t i = threadIdx.x; i < nwa + 3; i += VER_BLOCK) s_a64[i] = ga[i];
        }
    }
    const uint32_t seg0 = tile_off[t0], seg1 = tile_off[t1];
    if (seg0 == seg1) return;  // (workgroup-uniform)
    const uint32_t* __restrict__ s_a = reinterpret_cast<const uint32_t*>(s_a64);
    const uint32_t* __restrict__ ga32 = reinterpret_cast<const uint32_t*>(ga);
    if constexpr (STAGED) {
        for (uint32_t batch0 = seg0; batch0 < seg1; batch0 += VREC_CAP) {
            const uint32_t nb = min((uint32_t)VREC_CAP, seg1 - batch0);
            if (batch0 != seg0) __syncthreads();  // every group is done with the previous batch's records
            for (uint32_t i = threadIdx.x; i < nb; i += VER_BLOCK) {
                const uint32_t p = cand_p[batch0 + i], b = cand_b[batch0 + i];
                const uint32_t lb = len[b], rem = la - p;
                const uint32_t keep = STREAM ? keep_bits<true>(a, b, rem, lb, paired)
                                             : keep_bits(a, b, rem, lb, SCRAMBLED ? 2u : (paired ? 1u : 0u));
                const uint32_t n = rem < lb ? rem : lb;
                s_rec[i] = VRec{p | ((keep & 1u) << 31), b, keep ? (n | ((keep >> 1) << 31)) : 0u, (uint32_t)woff[b]};
            }
            if (threadIdx.x == 0) *s_next = batch0 + NGROUPS;  // the first candidate of every group is its position
            __syncthreads();
            if (in_lds)
                verify_run<BITS, SCRAMBLED, true, true, STREAM>(words, woff, len, cand_p, cand_b, paired, exc_off, exc_pos, exc_byte, type,
                                                        s_a, ga32, s_rec, s_next, a, la, batch0, batch0 + nb VST(, vt_start, vt));
            else
                verify_run<BITS, SCRAMBLED, false, true, STREAM>(words, woff, len, cand_p, cand_b, paired, exc_off, exc_pos, exc_byte, type,
                                                         s_a, ga32, s_rec, s_next, a, la, batch0, batch0 + nb VST(, vt_start, vt));
        }
    } else {
        // candidates are handed out dynamically: the first two per group by position, the rest from this counter
        if (threadIdx.x == 0) *s_next = seg0 + 2 * NGROUPS;
        __syncthreads();
        if (in_lds)
            verify_run<BITS, SCRAMBLED, true, false, STREAM>(words, woff, len, cand_p, cand_b, paired, exc_off, exc_pos, exc_byte, type,
                                                     s_a, ga32, s_rec, s_next, a, la, seg0, seg1 VST(, vt_start, vt));
        else
            verify_run<BITS, SCRAMBLED, false, false, STREAM>(words, woff, len, cand_p, cand_b, paired, exc_off, exc_pos, exc_byte, type,
                                                      s_a, ga32, s_rec, s_next, a, la, seg0, seg1 VST(, vt_start, vt));
    }
    // sel_deferred != nullptr: the longest-only selection inside this read's list (k_select_local's work) right here -- the
    // workgroup has just written every type of the list; one launch less per piece of a streamed step.  The records' LDS is
    // free now: 2 x SEL_CAP words of it hold the wave's table.
    if (sel_deferred) {
        static_assert((size_t)VREC_CAP * sizeof(VRec) >= 2 * SEL_CAP * sizeof(uint32_t), "the record area holds the selection table");
        __syncthreads();   // (every group is done: all types of [seg0, seg1) are written and visible to the workgroup)
        if (threadIdx.x < WAVE) {
            uint32_t* key = reinterpret_cast<uint32_t*>(s_rec);
            select_read_list(seg0, seg1, cand_p, cand_b, type, sel_selfrep, sel_deferred, key, key + SEL_CAP, threadIdx.x);
        }
    }
}

// ----------------------------------------------------------------------------------------
// select: rows per candidate.  B rows: every occurrence.  A rows: only the longest per (a,b),
// i.e. the smallest p = the smallest candidate index (candidates of one a come in ascending p).
// Two A hits for one (a,b) force b's prefix K-mer to recur inside b (period p-p'), so only
// candidates whose b has selfrep[b] set can be duplicates -- none at all in non-repetitive data,
// where the host skips this machinery.  Where they exist, k_select_mark records the smallest
// verified-A candidate index per (a,b) in an open-addressed table (size 2x the number of such
// candidates: O(1) per candidate, also for tandem repeats with thousands of hits per pair) and
// k_select drops every other one.
// ----------------------------------------------------------------------------------------
// one slot of the (a, b) -> smallest verified-A candidate table: key and minimum share 16 bytes, so the claim, the
// atomicMin and k_select's lookup touch one cache line (memset 0xFF = empty key, "infinite" minimum)
struct __attribute__((aligned(16))) PairSlot {
    unsigned long long key;
    uint32_t min_idx;
    uint32_t pad;
};

__device__ inline uint32_t pair_slot(uint32_t a, uint32_t b, uint32_t tbits) {
    uint32_t h1, h2;
    kmer_hash(((uint64_t)a << 32) | b, h1, h2);
    return (h1 ^ (h2 >> 3)) >> (32 - tbits);
}

// Longest-only selection inside one read's candidate list, in LDS (sharded calls and the wide index, which do
// not know which reads repeat their prefix): two verified A candidates of one (a, b) pair are necessarily in a's
// own list, so a wave takes one read, enters its verified A candidates into a small LDS table b -> smallest
// candidate POSITION p (the longest overlap; a list need not come in ascending p: the window-minimiser index emits a
// read's candidates in the order its probed words find them), and clears the A bit of every candidate of that b at
// another position -- (a, p, b) is found once, so the position names the candidate.  No global table, no
// device-scope atomics (9.3 M of them cost 1.3 ms at config 3).  A read with more verified A candidates than the
// table takes (tandem repeats) marks its b's as suspects instead and leaves them to the global table below.
// one wave, one read's candidate list [seg0, seg1): key / mn = SEL_CAP words of LDS each
__device__ __forceinline__ void select_read_list(uint32_t seg0, uint32_t seg1, const uint32_t* __restrict__ cand_p,
                                                 const uint32_t* __restrict__ cand_b, uint8_t* __restrict__ type,
                                                 uint32_t* __restrict__ selfrep, uint32_t* __restrict__ n_deferred,
                                                 uint32_t* key, uint32_t* mn, uint32_t lane) {
    if (seg1 - seg0 < 2) return;
    uint32_t n_a = 0;
    for (uint32_t c0 = seg0; c0 < seg1; c0 += WAVE) {
        const uint32_t c = c0 + lane;
        n_a += (uint32_t)__popcll(__ballot(c < seg1 && (type[c] & 1u)));
    }
    if (n_a < 2) return;  // (wave-uniform)
    if (n_a > SEL_CAP / 2) {
        for (uint32_t c = seg0 + lane; c < seg1; c += WAVE)
            if (type[c] & 1u) selfrep[cand_b[c]] = 0;  // "may repeat": the global selection handles this read
        if (lane == 0) atomicAdd(n_deferred, 1u);   // (the global table is only filled and looked at when this is non-zero)
        return;
    }
    for (uint32_t k = lane; k < SEL_CAP; k += WAVE) {
        key[k] = 0;
        mn[k] = ~0u;
    }
    wave_lds_fence();
    for (uint32_t c = seg0 + lane; c < seg1; c += WAVE) {
        if (!(type[c] & 1u)) continue;
        const uint32_t b = cand_b[c];
        uint32_t s = (b * 0x9E3779B1u) >> (32 - 9);
        for (;;) {
            const uint32_t prev = atomicCAS(&key[s], 0u, b + 1u);
            if (prev == 0u || prev == b + 1u) break;
            s = (s + 1u) & (SEL_CAP - 1u);
        }
        atomicMin(&mn[s], cand_p[c]);
    }
    wave_lds_fence();
    for (uint32_t c = seg0 + lane; c < seg1; c += WAVE) {
        const uint32_t t = type[c];
        if (!(t & 1u)) continue;
        const uint32_t b = cand_b[c];
        uint32_t s = (b * 0x9E3779B1u) >> (32 - 9);
        while (key[s] != b + 1u) s = (s + 1u) & (SEL_CAP - 1u);
        if (mn[s] != cand_p[c]) type[c] = (uint8_t)(t & ~1u);  // a longer overlap of the same pair exists
    }
}

__global__ __launch_bounds__(256) void k_select_local(const uint32_t* __restrict__ read_tile0, const uint32_t* __restrict__ tile_off,
                                                      const uint32_t* __restrict__ cand_p,
                                                      const uint32_t* __restrict__ cand_b, uint8_t* __restrict__ type,
                                                      uint32_t r_begin, uint32_t n_reads, uint32_t* __restrict__ selfrep,
                                                      uint32_t* __restrict__ n_deferred, const CandGuard G) {
    __shared__ uint32_t s_key[256 / WAVE][SEL_CAP];  // b + 1, 0 = empty
    if (G.overflow()) return;
    __shared__ uint32_t s_min[256 / WAVE][SEL_CAP];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * (256 / WAVE) + wave;
    if (i >= n_reads) return;  // whole wave
    const uint32_t a = r_begin + i;
    select_read_list(tile_off[read_tile0[a]], tile_off[read_tile0[a + 1]], cand_p, cand_b, type, selfrep, n_deferred, s_key[wave], s_min[wave], lane);
}

__global__ __launch_bounds__(256) void k_count_suspects(const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                                        uint32_t n_cand, const uint32_t* __restrict__ selfrep,
                                                        uint32_t* __restrict__ n_suspect) {
    __shared__ uint32_t s_part[256 / WAVE];
    uint32_t n = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += gridDim.x * blockDim.x)
        n += ((type[i] & 1u) && selfrep[cand_b[i]] != NO_SELFREP) ? 1u : 0u;
    n = (uint32_t)wave_sum64(n);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 256 / WAVE; ++w) t += s_part[w];
        if (t) atomicAdd(n_suspect, t);  // one atomic per workgroup
    }
}

// `gate` (may be null = always): the number of reads k_select_local handed over.  Sharded calls size the global
// (a, b) table for the worst case without asking the host; when no read was handed over -- every data set without
// tandem repeats -- the table is neither initialised nor filled nor looked at (28 MB of memset per shard otherwise).
__global__ void k_fill_gated(uint4* __restrict__ p, uint64_t n16, uint32_t v, const uint32_t* __restrict__ gate) {
    if (gate && *gate == 0u) return;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = uint4{v, v, v, v};
}

__global__ void k_select_mark(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p, const uint32_t* __restrict__ cand_b,
                              const uint8_t* __restrict__ type, uint32_t n_cand, const uint32_t* __restrict__ selfrep,
                              PairSlot* __restrict__ ptab, uint32_t tbits, const uint32_t* __restrict__ gate) {
    if (gate && *gate == 0u) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand || !(type[i] & 1u)) return;
    const uint32_t a = cand_a[i], b = cand_b[i];
    if (selfrep[b] == NO_SELFREP) return;
    const unsigned long long key = ((unsigned long long)a << 32) | b;  // a != b, so never ~0
    const uint32_t tmask = (1u << tbits) - 1u;
    uint32_t s = pair_slot(a, b, tbits);
    for (;;) {
        unsigned long long prev = ptab[s].key;
        if (prev == ~0ull) prev = atomicCAS(&ptab[s].key, ~0ull, key);
        if (prev == ~0ull || prev == key) break;
        s = (s + 1u) & tmask;
    }
    atomicMin(&ptab[s].min_idx, cand_p[i]);   // (the smallest POSITION = the longest overlap of the pair)
}

// Rows per verified candidate in emission order: A row, [its mirror], B row, [its mirror].
__device__ inline uint32_t rows_of(uint32_t t, uint32_t a, uint32_t b, uint32_t paired) {
    if (!paired) return (t & 1u) + ((t >> 1) & 1u);
    return ((t & 1u) ? (a == (b ^ 1u) ? 1u : 2u) : 0u) + ((t & 2u) ? 2u : 0u);
}

// ptab == nullptr: no read has a self-repeating prefix, every verified A candidate is the longest
__global__ void k_select(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p, const uint32_t* __restrict__ cand_b,
                         uint8_t* __restrict__ type, uint32_t n_cand, const uint32_t* __restrict__ selfrep,
                         const PairSlot* __restrict__ ptab, uint32_t tbits,
                         uint32_t paired, uint8_t* __restrict__ rowcnt, uint8_t* __restrict__ flag,
                         const uint32_t* __restrict__ gate) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    if (gate && *gate == 0u) ptab = nullptr;   // nothing was handed to the global table: it was never filled
    uint32_t t = type[i];
    if (t == 0) {
        rowcnt[i] = 0;
        if (flag) flag[i] = 0;  // candidates-only callers: flag = "survives" (rows > 0 <=> type != 0)
        return;
    }
    const uint32_t a = cand_a[i], b = cand_b[i];
    if (ptab && (t & 1u) && selfrep[b] != NO_SELFREP) {
        const unsigned long long key = ((unsigned long long)a << 32) | b;
        const uint32_t tmask = (1u << tbits) - 1u;
        uint32_t s = pair_slot(a, b, tbits);
        uint32_t first;
        for (;;) {  // k_select_mark inserted the key
            const uint4 q = *reinterpret_cast<const uint4*>(&ptab[s]);
            if ((((unsigned long long)q.y << 32) | q.x) == key) {
                first = q.z;
                break;
            }
            s = (s + 1u) & tmask;
        }
        if (first != cand_p[i]) {  // a longer overlap of the same pair exists
            t &= ~1u;
            type[i] = (uint8_t)t;
        }
    }
    rowcnt[i] = (uint8_t)rows_of(t, a, b, paired);
    if (flag) flag[i] = t != 0;
}

// Write the rows of one verified candidate at rows[off...].  Mirrors (exact
// when every read 2i+1 is the reverse complement of read 2i):
//   A (a, b, la-l, la, 0, l)   <->  (b^1, a^1, lb-l, lb, 0, l)
//   B (a, b, p, p+lb, 0, lb)   <->  (a^1, b^1, la-p-lb, la-p, 0, lb)
// ceil(l * bits / 8) for bits in {2, 8} without 64-bit arithmetic (v_mad_u64_u32 issues at a quarter of the rate)
__device__ inline uint32_t packed_bytes(uint32_t l, uint32_t bits) { return bits == 8u ? l : (l >> 2) + ((l & 3u) != 0u); }

__device__ inline void write_rows(Row* __restrict__ rows, uint32_t off, uint32_t t, uint32_t a, uint32_t p, uint32_t b,
                                  uint32_t la, uint32_t lb, uint32_t bits, uint32_t paired, uint64_t& suml,
                                  uint64_t& sumb, uint64_t& sume) {
    // bytes the verify kernel really compared for this candidate (both sides, once -- the mirrored rows cost nothing)
    sume += 2ull * packed_bytes((t & 1u) ? la - p : lb, bits);
    // (the counters grow by one row's worth next to each row store: sums of k * l as 64-bit products or shifts
    // issue at a quarter of the rate)
    if (t & 1u) {
        const uint32_t l = la - p;
        const uint64_t two_pb = 2ull * packed_bytes(l, bits);
        rows[off++] = Row{a, b, (int32_t)p, (int32_t)la, 0, (int32_t)l};
        suml += l;
        sumb += two_pb;
        if (paired && a != (b ^ 1u)) {
            rows[off++] = Row{b ^ 1u, a ^ 1u, (int32_t)(lb - l), (int32_t)lb, 0, (int32_t)l};
            suml += l;
            sumb += two_pb;
        }
    }
    if (t & 2u) {
        const uint64_t two_pb = 2ull * packed_bytes(lb, bits);
        rows[off++] = Row{a, b, (int32_t)p, (int32_t)(p + lb), 0, (int32_t)lb};
        suml += lb;
        sumb += two_pb;
        if (paired) {
            rows[off++] = Row{a ^ 1u, b ^ 1u, (int32_t)(la - p - lb), (int32_t)(la - p), 0, (int32_t)lb};
            suml += lb;
            sumb += two_pb;
        }
    }
}

// one atomic per counter per workgroup: [0]=verified candidates [1]=sum l [2]=sum 2*ceil(l*bits/8) over the rows
// [3]=the same over the verified candidates (what the verify kernel compared)
__device__ inline void flush_counters(uint64_t nver, uint64_t suml, uint64_t sumb, uint64_t sume,
                                      unsigned long long* __restrict__ counters) {
    __shared__ uint64_t s_red[4][1024 / WAVE];   // (workgroups of up to 1024 threads)
    nver = wave_sum64(nver);
    suml = wave_sum64(suml);
    sumb = wave_sum64(sumb);
    sume = wave_sum64(sume);
    if (lane_id() == 0) {
        s_red[0][threadIdx.x >> 6] = nver;
        s_red[1][threadIdx.x >> 6] = suml;
        s_red[2][threadIdx.x >> 6] = sumb;
        s_red[3][threadIdx.x >> 6] = sume;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t v = 0;
        for (uint32_t w = 0; w < blockDim.x / WAVE; ++w) v += s_red[threadIdx.x][w];
        if (v) atomicAdd(&counters[threadIdx.x], (unsigned long long)v);
    }
}

// Rows of candidate i start at row_off[i].
__global__ __launch_bounds__(256) void k_emit(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                                              const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                              const uint32_t* __restrict__ row_off, uint32_t n_cand,
                                              const uint32_t* __restrict__ len, Row* __restrict__ rows, uint32_t bits,
                                              uint32_t paired, unsigned long long* __restrict__ counters) {
    uint64_t nver = 0, suml = 0, sumb = 0, sume = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += gridDim.x * blockDim.x) {
        const uint32_t t = type[i];
        if (t == 0) continue;
        nver += 1;
        const uint32_t a = cand_a[i], p = cand_p[i], b = cand_b[i];
        write_rows(rows, row_off[i], t, a, p, b, len[a], len[b], bits, paired, suml, sumb, sume);
    }
    flush_counters(nver, suml, sumb, sume, counters);
}

// ---- multi-GPU form: verified candidates travel (16 B each, one per strand-mirror pair) instead of
// rows (24 B each, both members): compact on the producer, all-gather, expand on every rank.
struct Cand {
    uint32_t a, p, b, type;
};

// A verified-candidate record in 8 bytes, when the read set allows it: a and b below 2^sh_b, p below 2^(62 - sh_p),
// sh_p = 2 sh_b.  (rows home in compact form)
__host__ __device__ inline uint64_t pack_record(uint32_t a, uint32_t p, uint32_t b, uint32_t type, uint32_t sh_b, uint32_t sh_p) {
    return (uint64_t)a | ((uint64_t)b << sh_b) | ((uint64_t)p << sh_p) | ((uint64_t)type << 62);
}
__host__ __device__ inline Cand unpack_record(uint64_t v, uint32_t sh_b, uint32_t sh_p) {
    const uint64_t mr = (1ull << sh_b) - 1ull;
    return Cand{(uint32_t)(v & mr), (uint32_t)((v >> sh_p) & ((1ull << (62u - sh_p)) - 1ull)), (uint32_t)((v >> sh_b) & mr), (uint32_t)(v >> 62)};
}

__global__ void k_compact(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                          const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                          const uint8_t* __restrict__ flag, const uint32_t* __restrict__ flag_off, uint32_t n_cand,
                          Cand* __restrict__ out, uint32_t cap) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand || flag[i] == 0) return;
    // (cap: `out` may have been chosen before the number of kept candidates reached the host -- what does not fit is
    // not written, the host sees the number afterwards and compacts again into a buffer that holds them all)
    const uint32_t o = flag_off[i];
    if (o < cap) out[o] = Cand{cand_a[i], cand_p[i], cand_b[i], type[i]};
}

__global__ void k_flag(const uint8_t* __restrict__ rowcnt, uint32_t n, uint8_t* __restrict__ flag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = rowcnt[i] != 0;
}

__global__ void k_cand_rowcnt(const Cand* __restrict__ cands, uint32_t n, uint32_t n_reads, uint32_t paired,
                              uint8_t* __restrict__ rowcnt, uint32_t* __restrict__ n_bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cand c = cands[i];
    if ((c.a | c.p | c.b | c.type) == 0) {  // all-zero entry: padding of a fixed-size exchange buffer
        rowcnt[i] = 0;
        return;
    }
    if (c.a >= n_reads || c.b >= n_reads || c.a == c.b || c.type == 0 || c.type > 3) {  // not produced by this library
        rowcnt[i] = 0;
        atomicAdd(n_bad, 1u);
        return;
    }
    rowcnt[i] = (uint8_t)rows_of(c.type, c.a, c.b, paired);
}

__global__ __launch_bounds__(256) void k_emit_cands(const Cand* __restrict__ cands, const uint8_t* __restrict__ rowcnt,
                                                    const uint32_t* __restrict__ row_off, uint32_t n,
                                                    const uint32_t* __restrict__ len, Row* __restrict__ rows,
                                                    uint32_t bits, uint32_t paired,
                                                    unsigned long long* __restrict__ counters) {
    uint64_t nver = 0, suml = 0, sumb = 0, sume = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (rowcnt[i] == 0) continue;
        const Cand c = cands[i];
        nver += 1;
        write_rows(rows, row_off[i], c.type, c.a, c.p, c.b, len[c.a], len[c.b], bits, paired, suml, sumb, sume);
    }
    flush_counters(nver, suml, sumb, sume, counters);
}

// Paired-strand detection (2-bit reads only): is read 2i+1 exactly the reverse complement of read
// 2i, for every i?  One wave per pair; lane l checks words l, l+64, ... of the odd read against the
// matching window of the even read, reversed (bit reverse + swap the two bits of every base) and
// complemented (~).  Any failure bumps *n_bad.
__global__ __launch_bounds__(256) void k_paired_check(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                      const uint32_t* __restrict__ len, uint32_t n_pairs,
                                                      const uint8_t* __restrict__ pair_state,
                                                      uint32_t* __restrict__ n_bad) {
    const uint32_t pair = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pair >= n_pairs) return;
    const uint32_t lane = lane_id();
    const uint32_t L = len[2 * pair];
    bool bad = false;
    // pairs with exception records (non-ACGT bytes) were compared byte-wise on the host:
    // 1 = reverse complements, 2 = not; 0 = compare the 2-bit codes here
    const uint32_t hs = pair_state ? pair_state[pair] : 0u;
    if (hs != 0) {
        bad = hs == 2;
    } else if (len[2 * pair + 1] != L) {
        bad = true;
    } else {
        const uint64_t* __restrict__ R = words + woff[2 * pair];
        const uint64_t* __restrict__ S = words + woff[2 * pair + 1];
        const uint32_t nw = (L + 31) / 32;
        for (uint32_t w = lane; w < nw && !bad; w += WAVE) {
            const int64_t o = (int64_t)L - 32 * (int64_t)w - 32;  // first base of R facing this word
            uint64_t x;
            uint32_t valid = 32;
            if (o >= 0) {
                x = funnel(R[o >> 5], R[(o >> 5) + 1], (uint32_t)(o & 31) * 2);
            } else {
                valid = (uint32_t)(32 + o);
                x = R[0] << ((uint32_t)(-o) * 2);
            }
            uint64_t y = __brevll(x);
            y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
            y = ~y;
            const uint64_t mask = valid >= 32 ? ~0ull : ((1ull << (valid * 2)) - 1ull);
            bad = ((y ^ S[w]) & mask) != 0;
        }
    }
    if (__any(bad) && lane == 0) atomicAdd(n_bad, 1u);
}

// Store 1 rebuilt on the device: when every odd read is the reverse complement of its even partner (checked on the
// host as the reads were added) only the even reads cross PCIe.  One wave per pair; lane l writes words l, l+64, ...
// of read 2i+1 = the matching window of read 2i reversed (bit reverse + swap the two bits of every base) and
// complemented (~), bits beyond the read's end cleared, plus the zero guard word -- bit for bit what the host packs.
// exc_off / exc_pos (may be null): reads with exception records (non-ACGT bytes) carry code 0 at those positions in
// BOTH strands -- the complement of code 0 is code 3, so the positions of the odd read's own records are cleared here.
__global__ __launch_bounds__(256) void k_revcomp_store(uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                       const uint32_t* __restrict__ len, uint32_t pair0, uint32_t n_pairs,
                                                       const uint32_t* __restrict__ exc_off, const uint32_t* __restrict__ exc_pos) {
    const uint32_t pair = pair0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);   // pairs [pair0, n_pairs)
    if (pair >= n_pairs) return;
    const uint32_t e0 = exc_off ? exc_off[2 * pair + 1] : 0u, e1 = exc_off ? exc_off[2 * pair + 2] : 0u;
    const uint32_t lane = lane_id();
    const uint32_t L = len[2 * pair];
    const uint64_t* __restrict__ R = words + woff[2 * pair];
    uint64_t* __restrict__ S = words + woff[2 * pair + 1];
    const uint32_t nw = (L + 31) / 32;
    // words nw (the guard word) and, when the next read's 16-byte alignment leaves one, the padding word behind it
    const uint32_t last = nw + (uint32_t)((woff[2 * pair + 1] + nw + 1) & 1ull);
    for (uint32_t w = lane; w <= last; w += WAVE) {
        uint64_t out = 0;  // w >= nw: guard / padding
        if (w < nw) {
            const int64_t o = (int64_t)L - 32 * (int64_t)w - 32;  // first base of R facing this word
            uint64_t x;
            uint32_t valid = 32;
            if (o >= 0) {
                x = funnel(R[o >> 5], R[(o >> 5) + 1], (uint32_t)(o & 31) * 2);
            } else {
                valid = (uint32_t)(32 + o);
                x = R[0] << ((uint32_t)(-o) * 2);
            }
            uint64_t y = __brevll(x);
            y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
            y = ~y;
            out = valid >= 32 ? y : (y & ((1ull << (valid * 2)) - 1ull));
            for (uint32_t e = e0; e < e1; ++e) {   // (rare: a read with non-ACGT bytes)
                const uint32_t pos = exc_pos[e];
                if ((pos >> 5) == w) out &= ~(3ull << ((pos & 31u) * 2u));
            }
        }
        S[w] = out;
    }
}

// ---- streamed step (po_overlaps_to_host while the reads are still crossing PCIe piece by piece) ----------------
// The index only needs every read's first word (its prefix K-mer, K <= 32 bases of a word-aligned read; the wide index:
// the first two): those travel first, 16 bytes per read, and are put where the reads of the LATER pieces will land (reads [r0, n): the first
// piece has landed before the index is built).  The pieces later bring the same values; their copies are ordered
// behind this kernel.
__global__ void k_scatter_first(uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                const ulonglong2* __restrict__ first, uint32_t r0, uint32_t n) {
    const uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    // two words (the wide index reads the K-mers at the first W offsets: bases 0 .. 2W-2); a read starts on a 16-byte
    // boundary, so both words are its own -- data, guard or padding
    const ulonglong2 f = first[r];
    *reinterpret_cast<ulonglong2*>(words + woff[r]) = f;
}

// ... and the first `nw` words of every read (the wide index with windows of 4 reads the K-mers at the first 4 W offsets: bases
// 0 .. 5W-2).  Clipped to the words the read owns (its data words and the guard word behind them): a short read's neighbour
// is not touched -- the last read of store 0 is followed by store 1's first read on the device, by nothing on the host.
__global__ void k_scatter_lead(uint64_t* __restrict__ words, const uint64_t* __restrict__ woff, const uint32_t* __restrict__ len,
                               const uint64_t* __restrict__ lead, uint32_t nw, uint32_t r0, uint32_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = r0 + t / nw;
    const uint32_t k = (uint32_t)(t % nw);
    if (r >= n) return;
    const uint32_t own = (len[r] + 31u) / 32u + 1u;
    if (k < own) words[woff[r] + k] = lead[r * nw + k];
}

// Containment candidates (B) whose b-side read has not arrived yet: the verify kernel leaves them alone
// (keep_bits with the piece's b_limit); they are copied to one list and settled after the last piece.
// counter[0] keeps counting past `cap`, so that the host learns how much room the list needed.
__global__ void k_defer_split(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                              const uint32_t* __restrict__ cand_b, uint32_t n_cand, uint32_t b_limit, Cand* __restrict__ out,
                              uint32_t cap, uint32_t* __restrict__ counter, const CandGuard G) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (G.n_dev) {   // (predicted count: the grid covers G.cap candidates, the real number is on the device)
        if (G.overflow()) return;
        n_cand = (uint32_t)*G.n_dev;
    }
    if (i >= n_cand) return;
    const uint32_t b = cand_b[i];
    if (b < b_limit) return;
    const uint32_t k = atomicAdd(counter, 1u);
    if (k < cap) out[k] = Cand{cand_a[i], cand_p[i], b, 0u};
}

// The deferred list, once every read is on the device: is b (all of it) equal to a[p, p + len b)?  One wave per
// candidate, lane l compares words l, l + 64, ... of b with the window of a behind p.  2-bit reads without exception
// records only (the streamed step's precondition).  type = 2 (B row) or 0.
__global__ __launch_bounds__(256) void k_verify_flat(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                     const uint32_t* __restrict__ len, Cand* __restrict__ cands, uint32_t n,
                                                     const uint32_t* __restrict__ exc_off, const uint32_t* __restrict__ exc_pos,
                                                     const uint8_t* __restrict__ exc_byte) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;   // (whole waves leave together)
    const uint32_t lane = lane_id();
    const Cand c = cands[i];
    const uint32_t la = len[c.a], lb = len[c.b];
    bool bad = c.p > la || la - c.p < lb || c.a == c.b || (c.a & 1u);   // (not a containment of this mode: no row)
    if (!bad) {
        const uint64_t* __restrict__ A = words + woff[c.a] + (c.p >> 5);
        const uint64_t* __restrict__ B = words + woff[c.b];
        const uint32_t sh = (c.p & 31u) * 2u, nw = (lb + 31u) >> 5;
        for (uint32_t w = lane; w < nw; w += WAVE) {
            uint64_t d = funnel(A[w], A[w + 1], sh) ^ B[w];   // (a's guard word keeps A[w + 1] in bounds)
            if (w == nw - 1 && (lb & 31u)) d &= (1ull << ((lb & 31u) * 2u)) - 1ull;
            bad |= d != 0;
        }
    }
    bool any_bad = __any(bad);
    // (reads with exception records -- non-ACGT bytes, stored as code 0: the codes agree, do the bytes?)
    if (lane == 0 && !any_bad && exc_off && !exceptions_equal(exc_off, exc_pos, exc_byte, c.a, c.p, c.b, lb)) any_bad = true;
    if (lane == 0) cands[i].type = any_bad ? 0u : 2u;
}

// rows of the settled deferred list (k_emit_cands writes them)
__global__ void k_deferred_rowcnt(const Cand* __restrict__ cands, uint32_t n, uint32_t paired, uint8_t* __restrict__ rowcnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cand c = cands[i];
    rowcnt[i] = c.type ? (uint8_t)rows_of(c.type, c.a, c.b, paired) : (uint8_t)0;
}

// ----------------------------------------------------------------------------------------
// exclusive prefix sum (u8 / u32 in, u32 out, u64 total): reduce, spine, down-sweep
// ----------------------------------------------------------------------------------------
constexpr int PS_BLOCK = 256;
constexpr int PS_ITEMS = 16;
constexpr int PS_TILE = PS_BLOCK * PS_ITEMS;

template <typename T>
__global__ __launch_bounds__(PS_BLOCK) void k_ps_reduce(const T* __restrict__ in, uint64_t n, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t s_part[PS_BLOCK / WAVE];
    const uint64_t base = (uint64_t)blockIdx.x * PS_TILE;
    uint64_t acc = 0;
    const uint64_t first = base + (uint64_t)threadIdx.x * PS_ITEMS;   // this thread's PS_ITEMS consecutive items
    if (first + PS_ITEMS <= n && (reinterpret_cast<uintptr_t>(in + first) & 15u) == 0) {
        // whole, 16-byte aligned chunk: 16-byte loads (a u8 input is one load per thread instead of sixteen)
        const u32x4* src = reinterpret_cast<const u32x4*>(in + first);
#pragma unroll
        for (int k = 0; k < (int)(PS_ITEMS * sizeof(T) / 16); ++k) {
            const u32x4 q = src[k];
            if constexpr (sizeof(T) == 1) {
                const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += (d[j] & 0xFFu) + ((d[j] >> 8) & 0xFFu) + ((d[j] >> 16) & 0xFFu) + (d[j] >> 24);
            } else {
                acc += (uint64_t)q.x + q.y + q.z + q.w;
            }
        }
    } else {
        for (int k = 0; k < PS_ITEMS; ++k)
            if (first + k < n) acc += in[first + k];
    }
    acc = wave_sum64(acc);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (int w = 0; w < PS_BLOCK / WAVE; ++w) s += s_part[w];
        block_sums[blockIdx.x] = s;
    }
}

// single workgroup: exclusive scan of the block sums in place, grand total to *total.  Each thread owns a
// contiguous chunk; the 1024 chunk sums are scanned with wave scans (a serial loop in thread 0 took 13 us,
// 3-4 times per step).
__global__ __launch_bounds__(1024) void k_ps_spine(uint64_t* __restrict__ block_sums, uint32_t nblocks, uint64_t* __restrict__ total,
                                                  uint64_t* __restrict__ total_host, const uint64_t* __restrict__ also_src = nullptr,
                                                  uint64_t* __restrict__ also_host = nullptr) {
    __shared__ uint64_t s_wave[1024 / WAVE];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, nblocks);
    const uint32_t hi = min(lo + per, nblocks);
    uint64_t acc = 0;
    for (uint32_t i = lo; i < hi; ++i) acc += block_sums[i];
    // inclusive scan of acc across the wave (64-bit, shuffle based), then across the 16 waves
    uint64_t incl = acc;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d, WAVE);
        if (lane_id() >= (uint32_t)d) incl += up;
    }
    if (lane_id() == WAVE - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t run = incl - acc;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += s_wave[w];
    if (threadIdx.x == 1023) {
        *total = run + acc;
        *total_host = run + acc;   // (page-locked host memory mapped into the device: no copy command behind the kernel)
        if (also_src) *also_host = *also_src;   // (a device counter the host wants with the total)
    }
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t v = block_sums[i];
        block_sums[i] = run;
        run += v;
    }
}

// The same scan for inputs of a few thousand items (the tile counts of one piece of a streamed step, of one shard)
// in ONE workgroup: three dependent launches and a copy command cost more than the arithmetic there.  `extra`
// (optional) is added to the input first and the sum written back (k_add_extra folded in); `also_src` (optional) is a
// device counter the host wants next to the total.
constexpr uint32_t PS_SMALL_MAX = 1u << 13;   // (8 items per thread; at 128 k items the one workgroup took 300 us)
__global__ __launch_bounds__(1024) void k_ps_small(uint32_t* __restrict__ in, const uint32_t* __restrict__ extra, uint32_t n,
                                                  uint32_t* __restrict__ out, uint64_t* __restrict__ total,
                                                  uint64_t* __restrict__ total_host, const uint64_t* __restrict__ also_src,
                                                  uint64_t* __restrict__ also_host) {
    __shared__ uint64_t s_wave[1024 / WAVE];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, n);
    const uint32_t hi = min(lo + per, n);
    uint64_t acc = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t v = in[i];
        if (extra && extra[i]) {
            v += extra[i];
            in[i] = v;
        }
        acc += v;
    }
    uint64_t incl = acc;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d, WAVE);
        if (lane_id() >= (uint32_t)d) incl += up;
    }
    if (lane_id() == WAVE - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t run = incl - acc;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += s_wave[w];
    if (threadIdx.x == 1023) {
        *total = run + acc;
        *total_host = run + acc;
        out[n] = (uint32_t)(run + acc);   // closing sentinel, as k_ps_down writes it: out has n + 1 entries
        if (also_src) *also_host = *also_src;
    }
    for (uint32_t i = lo; i < hi; ++i) {
        out[i] = (uint32_t)run;
        run += in[i];
    }
}

template <typename T>
__global__ __launch_bounds__(PS_BLOCK) void k_ps_down(const T* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_sums,
                                                      uint32_t* __restrict__ out) {
    // thread owns PS_ITEMS consecutive items; block-wide scan of the per-thread sums.  Whole, 16-byte aligned
    // runs are read and written as 16-byte vectors (a u8 input is one load per thread instead of sixteen, the
    // offsets four stores instead of sixteen 64-byte-strided ones: 38 -> 13 us for 6.5 M items).
    static_assert(PS_ITEMS == 16, "vector paths assume 16 items per thread");
    __shared__ uint32_t s_wave[PS_BLOCK / WAVE];
    const uint64_t base = (uint64_t)blockIdx.x * PS_TILE + (uint64_t)threadIdx.x * PS_ITEMS;
    const bool full = base + PS_ITEMS <= n;
    uint32_t v[PS_ITEMS];
    if (full && (reinterpret_cast<uintptr_t>(in + base) & 15u) == 0) {
        if constexpr (sizeof(T) == 1) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(in + base);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < PS_ITEMS; ++k) v[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 q = *reinterpret_cast<const u32x4*>(in + base + 4 * g);
                v[4 * g] = q.x;
                v[4 * g + 1] = q.y;
                v[4 * g + 2] = q.z;
                v[4 * g + 3] = q.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PS_ITEMS; ++k) {
            const uint64_t i = base + k;
            v[k] = i < n ? (uint32_t)in[i] : 0u;
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) acc += v[k];
    const uint32_t incl = wave_incl_scan(acc);
    if (lane_id() == WAVE - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wave_base += s_wave[w];
    uint32_t run = (uint32_t)block_sums[blockIdx.x] + wave_base + incl - acc;
    uint32_t o[PS_ITEMS];
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) {
        o[k] = run;
        run += v[k];
    }
    if (full && (reinterpret_cast<uintptr_t>(out + base) & 15u) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x4 q;
            q.x = o[4 * g];
            q.y = o[4 * g + 1];
            q.z = o[4 * g + 2];
            q.w = o[4 * g + 3];
            *reinterpret_cast<u32x4*>(out + base + 4 * g) = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PS_ITEMS; ++k)
            if (base + k < n) out[base + k] = o[k];
    }
    if (base < n && n <= base + PS_ITEMS) out[n] = run;  // closing sentinel (this thread holds item n-1): out has n+1 entries
}


// ----------------------------------------------------------------------------------------
// Single-pass ("chained") scan: ONE launch instead of reduce + spine + down-sweep.
//
// A streamed piece is ~24 dependent launches of which most are a few microseconds of work; every dependent launch
// costs ~4.5 us on this chip whatever it does (profiles/r03_piece_timeline.txt), so the three-kernel scans (run twice
// per piece) and the select -> scan -> emit chain are folded into single kernels here.
//
// Decoupled look-back: tiles are handed out in ticket order (a tile's predecessors have all started), a tile
// publishes its aggregate, then looks back over its predecessors' status words -- flag and value packed into one
// 64-bit word, so a relaxed load sees them together -- until it meets an inclusive prefix.  The state (ticket, done
// counter, status words) is all-zero between launches: the last tile to finish clears it.
// ----------------------------------------------------------------------------------------
constexpr unsigned long long CHAIN_AGG = 1ull << 62, CHAIN_PFX = 2ull << 62, CHAIN_VAL = (1ull << 62) - 1ull;
struct ChainState {
    uint32_t ticket, done, pad0, pad1;
    unsigned long long status[1];   // [n_tiles]
};
__host__ __device__ inline size_t chain_state_bytes(uint32_t n_tiles) { return 16 + (size_t)n_tiles * 8; }

__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the tile this workgroup works on (ticket order); call with all threads, one __syncthreads inside
__device__ __forceinline__ uint32_t chain_take_tile(ChainState* st, uint32_t* s_tile) {
    if (threadIdx.x == 0) *s_tile = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    return *s_tile;
}

// exclusive prefix of tile `tile` given its aggregate; executed by ONE whole wave (all 64 lanes call it, same arguments)
__device__ __forceinline__ unsigned long long chain_lookback(ChainState* st, uint32_t tile, unsigned long long aggregate) {
    const uint32_t lane = lane_id();
    if (lane == 0) chain_store(&st->status[tile], (tile == 0 ? CHAIN_PFX : CHAIN_AGG) | aggregate);
    unsigned long long prefix = 0;
    if (tile == 0) return 0;
    int64_t j = (int64_t)tile - 1;
    for (;;) {
        const int64_t idx = j - (int64_t)lane;
        unsigned long long v;
        for (;;) {   // every predecessor holds an earlier ticket: it is running or done, its word will come
            v = idx >= 0 ? chain_load(&st->status[idx]) : CHAIN_PFX;
            if (!__any((v >> 62) == 0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const uint64_t pm = __ballot((v >> 62) == 2);
        const uint32_t first = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;
        unsigned long long part = lane <= first ? (v & CHAIN_VAL) : 0ull;
        part = wave_sum64(part);
        prefix += part;
        if (pm) break;
        j -= 64;
    }
    if (lane == 0) chain_store(&st->status[tile], CHAIN_PFX | (prefix + aggregate));
    return prefix;
}

// a tile is finished with the state; the last one of the launch clears it for the next launch.  All threads call it.
__device__ __forceinline__ bool chain_finish(ChainState* st, uint32_t n_tiles, uint32_t* s_last) {
    // (no __threadfence: on this chip a device-scope release writes back the whole L2 of the XCD.  What must be ordered
    // before the done count are this workgroup's READS of status words, and those have returned)
    __syncthreads();
    if (threadIdx.x == 0) *s_last = atomicAdd(&st->done, 1u) == n_tiles - 1u ? 1u : 0u;
    __syncthreads();
    const bool last = *s_last != 0u;
    if (last) {
        for (uint32_t i = threadIdx.x; i < n_tiles; i += blockDim.x) st->status[i] = 0;
        if (threadIdx.x == 0) {
            st->ticket = 0;
            st->done = 0;
        }
    }
    return last;
}

// exclusive scan of n items (u8 / u32) -> u32 offsets (+ closing sentinel out[n]), total to *total and the pinned
// *total_host; `extra` (u32 inputs only) is added to the input first and the sum written back (k_add_extra folded in)
template <typename T>
__global__ __launch_bounds__(PS_BLOCK) void k_ps_chain(T* __restrict__ in, const uint32_t* __restrict__ extra, uint64_t n,
                                                       uint32_t* __restrict__ out, ChainState* __restrict__ st, uint32_t n_tiles,
                                                       uint64_t* __restrict__ total, uint64_t* __restrict__ total_host,
                                                       const uint64_t* __restrict__ also_src, uint64_t* __restrict__ also_host) {
    static_assert(PS_ITEMS == 16, "vector paths assume 16 items per thread");
    __shared__ uint32_t s_wave[PS_BLOCK / WAVE];
    __shared__ uint32_t s_tile, s_last;
    __shared__ unsigned long long s_prefix;
    const uint32_t tile = chain_take_tile(st, &s_tile);
    const uint64_t base = (uint64_t)tile * PS_TILE + (uint64_t)threadIdx.x * PS_ITEMS;
    const bool full = base + PS_ITEMS <= n;
    uint32_t v[PS_ITEMS];
    if (full && (reinterpret_cast<uintptr_t>(in + base) & 15u) == 0) {
        if constexpr (sizeof(T) == 1) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(in + base);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < PS_ITEMS; ++k) v[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 q = *reinterpret_cast<const u32x4*>(in + base + 4 * g);
                v[4 * g] = q.x;
                v[4 * g + 1] = q.y;
                v[4 * g + 2] = q.z;
                v[4 * g + 3] = q.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PS_ITEMS; ++k) {
            const uint64_t i = base + k;
            v[k] = i < n ? (uint32_t)in[i] : 0u;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (extra) {
#pragma unroll
            for (int k = 0; k < PS_ITEMS; ++k) {
                const uint64_t i = base + k;
                const uint32_t e = i < n ? extra[i] : 0u;
                if (e) {
                    v[k] += e;
                    in[i] = v[k];
                }
            }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) acc += v[k];
    const uint32_t incl = wave_incl_scan(acc);
    if (lane_id() == WAVE - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0, block_total = 0;
    for (uint32_t w = 0; w < PS_BLOCK / WAVE; ++w) {
        if (w < (threadIdx.x >> 6)) wave_base += s_wave[w];
        block_total += s_wave[w];
    }
    if (threadIdx.x < WAVE) {
        const unsigned long long pfx = chain_lookback(st, tile, block_total);
        if (threadIdx.x == 0) s_prefix = pfx;
    }
    __syncthreads();
    uint32_t run = (uint32_t)s_prefix + wave_base + incl - acc;
    uint32_t o[PS_ITEMS];
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) {
        o[k] = run;
        run += v[k];
    }
    if (full && (reinterpret_cast<uintptr_t>(out + base) & 15u) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x4 q;
            q.x = o[4 * g];
            q.y = o[4 * g + 1];
            q.z = o[4 * g + 2];
            q.w = o[4 * g + 3];
            *reinterpret_cast<u32x4*>(out + base + 4 * g) = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PS_ITEMS; ++k)
            if (base + k < n) out[base + k] = o[k];
    }
    if (base < n && n <= base + PS_ITEMS) {   // this thread holds item n-1: closing sentinel and the grand total
        const unsigned long long tot = s_prefix + wave_base + incl;
        out[n] = (uint32_t)tot;
        *total = tot;
        *total_host = tot;   // (page-locked host memory mapped into the device)
        if (also_src) *also_host = *also_src;
    }
    chain_finish(st, n_tiles, &s_last);
}

// ----------------------------------------------------------------------------------------
// The tail of a call in TWO launches: rows per tile of candidates (k_tile_rows), then -- every workgroup adds up the
// tiles before its own, a few KB from L2 -- offsets, the rows themselves and the counters, written straight into the
// pinned landing zone (k_tail).  Instead of k_select, three scan kernels, k_emit and a copy command; for calls whose
// row buffer is known to hold the worst case (the steady state).  No workgroup waits for another: a first version
// chained the tiles with a decoupled look-back inside ONE kernel, and 600 spinning waves reading status words across
// the eight XCDs took 130-340 us for what these two kernels do in 60.
// `gate` (may be null): reads k_select_local handed to the global (a, b) table; if there are any, the two kernels write
// nothing but host_out[7] = 1 and the host takes the classic path.
// Thread t of a tile takes candidates t, t + 256, ...: neighbours write neighbouring rows.
// host_out: [0] rows, [1] verified candidates, [2] sum l, [3] algorithmic bytes, [4] compared bytes, [7] fallback flag
// ----------------------------------------------------------------------------------------
constexpr int TAIL_BLOCK = 256;
constexpr int TAIL_ITEMS = 2;
constexpr int TAIL_TILE = TAIL_BLOCK * TAIL_ITEMS;
constexpr uint32_t TAIL_MAX_TILES = 1u << 15;   // (a workgroup sums the tiles before its own: 16 M candidates at most -- a piece
                                                //  of a streamed step, a shard; a whole-set call of 6.5 M candidates is faster classic)

// COMPACT: the tail hands out verified candidates instead of rows (k_tail_cands): tile_rows[] = verified candidates per tile
template <bool COMPACT = false>
__global__ __launch_bounds__(TAIL_BLOCK) void k_tile_rows(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_b,
                                                          const uint8_t* __restrict__ type, uint32_t n_cand, uint32_t paired,
                                                          const uint32_t* __restrict__ gate, uint32_t* __restrict__ tile_rows,
                                                          const CandGuard G) {
    __shared__ uint32_t s_part[TAIL_BLOCK / WAVE];
    if (gate && *gate != 0u) return;
    if (G.n_dev) {   // (predicted count: the grid covers G.cap candidates, the real number is on the device)
        if (G.overflow()) return;
        n_cand = (uint32_t)*G.n_dev;
    }
    const uint32_t base = blockIdx.x * TAIL_TILE;
    if (base >= n_cand) return;
    uint32_t n = 0;
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        const uint32_t i = base + r * TAIL_BLOCK + threadIdx.x;
        const uint32_t t = i < n_cand ? type[i] : 0u;
        if (COMPACT) n += t != 0u;
        else if (t) n += rows_of(t, cand_a[i], cand_b[i], paired);
    }
    n = wave_sum(n);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) s += s_part[w];
        tile_rows[blockIdx.x] = s;
    }
}

// done: a device counter, zero between launches (the last workgroup resets it)
__global__ __launch_bounds__(TAIL_BLOCK) void k_tail(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                                                     const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                                     uint32_t n_cand, const uint32_t* __restrict__ len, Row* __restrict__ rows,
                                                     uint32_t bits, uint32_t paired, const uint32_t* __restrict__ gate,
                                                     const uint32_t* __restrict__ tile_rows, uint32_t n_tiles, uint32_t* __restrict__ done,
                                                     unsigned long long* __restrict__ counters, uint64_t* __restrict__ host_out,
                                                     const CandGuard G) {
    __shared__ uint32_t s_cnt[TAIL_ITEMS][TAIL_BLOCK / WAVE];
    __shared__ uint32_t s_pre[TAIL_BLOCK / WAVE];
    __shared__ uint32_t s_last;
    if ((gate && *gate != 0u) || G.overflow()) {   // (grid-uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) host_out[7] = G.overflow() ? 2 : 1;
        return;
    }
    if (G.n_dev) {   // (predicted count: the grid covers G.cap candidates)
        n_cand = (uint32_t)*G.n_dev;
        n_tiles = (n_cand + TAIL_TILE - 1) / TAIL_TILE;
        if (n_tiles == 0) {   // (no candidates after all: nothing to write but the zeros)
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                for (int k = 0; k < 5; ++k) host_out[k] = 0;
                host_out[7] = 0;
            }
            return;
        }
        if (blockIdx.x >= n_tiles) return;
    }
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * TAIL_TILE;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    // rows of the tiles before this one (a few KB, L2-resident: every workgroup reads the same array)
    uint32_t pre = 0;
    {
        uint32_t k = threadIdx.x;
        for (; k + 3 * TAIL_BLOCK < tile; k += 4 * TAIL_BLOCK)   // (four loads in flight)
            pre += tile_rows[k] + tile_rows[k + TAIL_BLOCK] + tile_rows[k + 2 * TAIL_BLOCK] + tile_rows[k + 3 * TAIL_BLOCK];
        for (; k < tile; k += TAIL_BLOCK) pre += tile_rows[k];
    }
    uint32_t t[TAIL_ITEMS], a[TAIL_ITEMS], b[TAIL_ITEMS], cnt[TAIL_ITEMS], excl[TAIL_ITEMS];
    uint32_t pp[TAIL_ITEMS], la[TAIL_ITEMS], lb[TAIL_ITEMS];
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        const uint32_t i = base + r * TAIL_BLOCK + threadIdx.x;
        t[r] = i < n_cand ? type[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {   // (all loads before the first row store)
        const uint32_t i = base + r * TAIL_BLOCK + threadIdx.x;
        a[r] = b[r] = pp[r] = la[r] = lb[r] = 0;
        if (t[r]) {
            a[r] = cand_a[i];
            b[r] = cand_b[i];
            pp[r] = cand_p[i];
        }
    }
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        if (t[r]) {
            la[r] = len[a[r]];
            lb[r] = len[b[r]];
        }
        cnt[r] = t[r] ? rows_of(t[r], a[r], b[r], paired) : 0u;
        const uint32_t inc = wave_incl_scan(cnt[r]);
        excl[r] = inc - cnt[r];
        if (lane == WAVE - 1) s_cnt[r][wave] = inc;
    }
    pre = wave_sum(pre);
    if (lane == 0) s_pre[wave] = pre;
    __syncthreads();
    uint32_t pfx = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) pfx += s_pre[w];
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        uint32_t before = block_total;   // rows of the earlier rounds
#pragma unroll
        for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) {
            if ((uint32_t)w < wave) before += s_cnt[r][w];
            block_total += s_cnt[r][w];
        }
        excl[r] += before;
    }
    uint64_t nver = 0, suml = 0, sumb = 0, sume = 0;
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        if (!t[r]) continue;
        nver += 1;
        write_rows(rows, pfx + excl[r], t[r], a[r], pp[r], b[r], la[r], lb[r], bits, paired, suml, sumb, sume);
    }
    // counters: one RETURNING atomic per counter per workgroup -- the value coming back means the add has been performed
    // at the coherence point, so the done count below is ordered behind it without a fence (a device-scope release
    // fence writes back the XCD's whole L2, rows and all: 2.3 ms per call with one per workgroup)
    {
        __shared__ uint64_t s_red[4][TAIL_BLOCK / WAVE];
        nver = wave_sum64(nver);
        suml = wave_sum64(suml);
        sumb = wave_sum64(sumb);
        sume = wave_sum64(sume);
        if (lane == 0) {
            s_red[0][wave] = nver;
            s_red[1][wave] = suml;
            s_red[2][wave] = sumb;
            s_red[3][wave] = sume;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            uint64_t v = 0;
            for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) v += s_red[threadIdx.x][w];
            unsigned long long old = 0;
            if (v) old = atomicAdd(&counters[threadIdx.x], (unsigned long long)v);
            asm volatile("" :: "v"((uint32_t)old));   // (keeps the return value, hence the wait for it)
        }
    }
    if (tile == n_tiles - 1u && threadIdx.x == 0) host_out[0] = (uint64_t)pfx + block_total;   // (the last tile: all rows)
    __syncthreads();
    if (threadIdx.x == 0) {
        s_last = atomicAdd(done, 1u) == n_tiles - 1u ? 1u : 0u;
        if (s_last) {
            *done = 0;
            for (int k = 0; k < 4; ++k) {
                host_out[1 + k] = __hip_atomic_load(&counters[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                counters[k] = 0;   // (zero again for the next piece: every workgroup's add has been performed, see above)
            }
            host_out[7] = 0;
        }
    }
}

// ----------------------------------------------------------------------------------------
// The tail for rows that go home in COMPACT form (po_overlaps_to_host; c_api.hip "rows home"): the verified candidates of
// the call, compacted in candidate order as 16-byte records {a, p, b, type} -- one per strand-mirror pair in paired mode --
// instead of their rows (24 bytes each, two to four per record).  Host threads expand them into the page-locked row array
// as the pieces land (expand_records), bit for bit what write_rows writes: PCIe carries a third of the bytes.
// tile_recs[] = verified candidates per tile (k_tile_rows<true>).  The row total is summed here; the byte counters of
// po_stats (sum l, algorithmic bytes, compared bytes) are taken on the host from the same records.
// host_out: [0] rows, [1] verified candidates = records written, [2..4] zero, [7] fallback flag.
// rows_ctr: a device counter, zero at launch (scalars[3]).
// ----------------------------------------------------------------------------------------
__device__ inline uint32_t row_sums(uint32_t t, uint32_t a, uint32_t p, uint32_t b, uint32_t la, uint32_t lb, uint32_t bits,
                                    uint32_t paired, uint64_t& suml, uint64_t& sumb, uint64_t& sume) {
    // (the counter arithmetic of write_rows without the stores)
    uint32_t n = 0;
    sume += 2ull * packed_bytes((t & 1u) ? la - p : lb, bits);
    if (t & 1u) {
        const uint32_t l = la - p;
        const uint32_t k = (paired && a != (b ^ 1u)) ? 2u : 1u;
        suml += (uint64_t)k * l;
        sumb += (uint64_t)k * 2ull * packed_bytes(l, bits);
        n += k;
    }
    if (t & 2u) {
        const uint32_t k = paired ? 2u : 1u;
        suml += (uint64_t)k * lb;
        sumb += (uint64_t)k * 2ull * packed_bytes(lb, bits);
        n += k;
    }
    return n;
}

__global__ __launch_bounds__(TAIL_BLOCK) void k_tail_cands(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                                                           const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                                           uint32_t n_cand, Cand* __restrict__ out, uint32_t paired,
                                                           const uint32_t* __restrict__ gate, const uint32_t* __restrict__ tile_recs,
                                                           uint32_t n_tiles, uint32_t* __restrict__ done, unsigned long long* __restrict__ rows_ctr,
                                                           uint64_t* __restrict__ host_out, const CandGuard G, uint32_t sh_b, uint32_t sh_p) {
    __shared__ uint32_t s_cnt[TAIL_ITEMS][TAIL_BLOCK / WAVE];
    __shared__ uint32_t s_pre[TAIL_BLOCK / WAVE];
    __shared__ uint32_t s_rows[TAIL_BLOCK / WAVE];
    __shared__ uint32_t s_last;
    if ((gate && *gate != 0u) || G.overflow()) {   // (grid-uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) host_out[7] = G.overflow() ? 2 : 1;
        return;
    }
    if (G.n_dev) {   // (predicted count: the grid covers G.cap candidates)
        n_cand = (uint32_t)*G.n_dev;
        n_tiles = (n_cand + TAIL_TILE - 1) / TAIL_TILE;
        if (n_tiles == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                for (int k = 0; k < 5; ++k) host_out[k] = 0;
                host_out[7] = 0;
            }
            return;
        }
        if (blockIdx.x >= n_tiles) return;
    }
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * TAIL_TILE;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t pre = 0;   // records of the tiles before this one
    {
        uint32_t k = threadIdx.x;
        for (; k + 3 * TAIL_BLOCK < tile; k += 4 * TAIL_BLOCK)
            pre += tile_recs[k] + tile_recs[k + TAIL_BLOCK] + tile_recs[k + 2 * TAIL_BLOCK] + tile_recs[k + 3 * TAIL_BLOCK];
        for (; k < tile; k += TAIL_BLOCK) pre += tile_recs[k];
    }
    uint32_t t[TAIL_ITEMS], a[TAIL_ITEMS], b[TAIL_ITEMS], pp[TAIL_ITEMS], excl[TAIL_ITEMS];
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        const uint32_t i = base + r * TAIL_BLOCK + threadIdx.x;
        t[r] = i < n_cand ? type[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        const uint32_t i = base + r * TAIL_BLOCK + threadIdx.x;
        a[r] = b[r] = pp[r] = 0;
        if (t[r]) {
            a[r] = cand_a[i];
            b[r] = cand_b[i];
            pp[r] = cand_p[i];
        }
    }
    uint32_t nrows = 0;
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        const uint32_t cnt = t[r] ? 1u : 0u;
        const uint32_t inc = wave_incl_scan(cnt);
        excl[r] = inc - cnt;
        if (lane == WAVE - 1) s_cnt[r][wave] = inc;
        if (t[r]) nrows += rows_of(t[r], a[r], b[r], paired);
    }
    pre = wave_sum(pre);
    nrows = wave_sum(nrows);
    if (lane == 0) {
        s_pre[wave] = pre;
        s_rows[wave] = nrows;
    }
    __syncthreads();
    uint32_t pfx = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) pfx += s_pre[w];
#pragma unroll
    for (int r = 0; r < TAIL_ITEMS; ++r) {
        uint32_t before = block_total;
#pragma unroll
        for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) {
            if ((uint32_t)w < wave) before += s_cnt[r][w];
            block_total += s_cnt[r][w];
        }
        excl[r] += before;
    }
    // sh_b != 0: the record in EIGHT bytes -- a | b << sh_b | p << sh_p | type << 62 (pack_record; the host chose the widths
    // from the number of reads and the longest read, c_api.hip: home_pack_shifts) -- half the bytes on the wire again
    if (sh_b) {
        uint64_t* out8 = reinterpret_cast<uint64_t*>(out);
#pragma unroll
        for (int r = 0; r < TAIL_ITEMS; ++r)
            if (t[r]) out8[pfx + excl[r]] = pack_record(a[r], pp[r], b[r], t[r], sh_b, sh_p);
    } else {
#pragma unroll
        for (int r = 0; r < TAIL_ITEMS; ++r)
            if (t[r]) out[pfx + excl[r]] = Cand{a[r], pp[r], b[r], t[r]};
    }
    if (tile == n_tiles - 1u && threadIdx.x == 0) host_out[1] = (uint64_t)pfx + block_total;   // (the last tile: all records)
    if (threadIdx.x == 0) {
        // ONE returning atomic per workgroup for the row total (the byte sums of po_stats are taken on the host while the
        // records are expanded: five same-address atomics per workgroup were most of this kernel's 30 us per piece),
        // performed at the coherence point before the done count below (see k_tail)
        uint32_t rows = 0;
        for (int w = 0; w < TAIL_BLOCK / WAVE; ++w) rows += s_rows[w];
        unsigned long long old = 0;
        if (rows) old = atomicAdd(rows_ctr, (unsigned long long)rows);
        asm volatile("" :: "v"((uint32_t)old));
        s_last = atomicAdd(done, 1u) == n_tiles - 1u ? 1u : 0u;
        if (s_last) {
            *done = 0;
            host_out[0] = __hip_atomic_load(rows_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *rows_ctr = 0;   // (zero again for the next piece)
            host_out[2] = host_out[3] = host_out[4] = 0;
            host_out[7] = 0;
        }
    }
}

}  // namespace po
