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

constexpr uint64_t KEY_EMPTY = ~0ull;          // slot-claim sentinel; a real all-ones K-mer lives
                                               // in the dedicated extra slot at index 1<<tbits
constexpr uint32_t NO_SELFREP = 0xFFFFFFFFu;
constexpr int WAVE = 64;
constexpr int SCAN_BLOCK = 1024;               // 16 waves: one persistent workgroup per CU
constexpr int TILE_WORDS = 64;                 // one 64-bit word per lane

struct __attribute__((aligned(16))) Slot {
    uint64_t key;
    uint32_t start;   // first entry in chain[]
    uint32_t count;   // 0 = empty slot
};

struct Row {
    uint32_t a_idx, b_idx;
    int32_t astart, aend, bstart, bend;
};

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
__host__ __device__ inline void kmer_hash(uint64_t k, uint32_t& h1, uint32_t& h2) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = lo ^ ((hi << 13) | (hi >> 19));
    h1 = x * 0x9E3779B1u;
    h2 = (x ^ (x >> 15) ^ (hi * 5u)) * 0x85EBCA77u;
}

__device__ inline uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

__device__ inline uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

__device__ inline uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

__device__ inline uint32_t read_last_lane(uint32_t v) { return __builtin_amdgcn_readlane(v, WAVE - 1); }

// LDS hand-off between lanes of ONE wave: DS ops of a wave execute in issue order; this only has
// to stop the compiler from moving accesses across and to drain lgkmcnt.
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ inline uint64_t funnel(uint64_t lo, uint64_t hi, uint32_t sh) {
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

__device__ inline void table_probe(const Slot* __restrict__ tab, uint32_t tbits, uint64_t kmer,
                                   uint32_t& start, uint32_t& cnt) {
    start = 0;
    cnt = 0;
    const uint32_t tmask = (1u << tbits) - 1u;
    if (kmer == KEY_EMPTY) {
        const uint4 s = *reinterpret_cast<const uint4*>(&tab[tmask + 1u]);
        start = s.z;
        cnt = s.w;
        return;
    }
    uint32_t h1, h2;
    kmer_hash(kmer, h1, h2);
    uint32_t i = h1 >> (32 - tbits);
    for (;;) {
        const uint4 s = *reinterpret_cast<const uint4*>(&tab[i]);
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
__global__ void k_table_init(Slot* tab, uint32_t nslots, uint32_t* slot_cnt, uint32_t* slot_cur) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        tab[i].key = KEY_EMPTY;
        tab[i].start = 0;
        tab[i].count = 0;
        slot_cnt[i] = 0;
        slot_cur[i] = 0;
    }
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
                               uint32_t* read_slot, uint32_t* bloom, uint32_t bloom_log2) {
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
        i = h1 >> (32 - tbits);
        for (;;) {
            unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[i].key),
                                                (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (prev == KEY_EMPTY || prev == key) break;
            i = (i + 1u) & tmask;
        }
    }
    atomicAdd(&slot_cnt[i], 1u);
    read_slot[r] = i;
    const uint32_t i1 = h1 >> (32 - bloom_log2), i2 = h2 >> (32 - bloom_log2);
    atomicOr(&bloom[i1 >> 5], 1u << (i1 & 31));
    atomicOr(&bloom[i2 >> 5], 1u << (i2 & 31));
}

__global__ void k_table_finalize(Slot* tab, uint32_t nslots, const uint32_t* __restrict__ slot_cnt,
                                 const uint32_t* __restrict__ slot_start) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        tab[i].start = slot_start[i];
        tab[i].count = slot_cnt[i];
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

__global__ void k_chain_sort_short(const uint32_t* __restrict__ slot_cnt, const uint32_t* __restrict__ slot_start,
                                   uint32_t nslots, uint32_t* chain, uint32_t* long_list, uint32_t* n_long) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const uint32_t c = slot_cnt[i];
    if (c < 2) return;
    if (c > CHAIN_SHORT) {
        long_list[atomicAdd(n_long, 1u)] = i;
        return;
    }
    uint32_t* v = chain + slot_start[i];
    for (uint32_t x = 1; x < c; ++x) {
        uint32_t key = v[x];
        uint32_t y = x;
        while (y > 0 && v[y - 1] > key) {
            v[y] = v[y - 1];
            --y;
        }
        v[y] = key;
    }
}

// One workgroup per long chain: rank sort through a scratch copy (read indices are distinct).
__global__ void k_chain_sort_long(const uint32_t* __restrict__ slot_cnt, const uint32_t* __restrict__ slot_start,
                                  const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ n_long,
                                  uint32_t* chain, uint32_t* scratch) {
    for (uint32_t li = blockIdx.x; li < *n_long; li += gridDim.x) {
        const uint32_t s = long_list[li];
        const uint32_t c = slot_cnt[s], st = slot_start[s];
        for (uint32_t x = threadIdx.x; x < c; x += blockDim.x) scratch[st + x] = chain[st + x];
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < c; x += blockDim.x) {
            const uint32_t v = scratch[st + x];
            uint32_t rank = 0;
            for (uint32_t y = 0; y < c; ++y) rank += scratch[st + y] < v;
            chain[st + rank] = v;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------
// position scan
// ----------------------------------------------------------------------------------------
struct ScanArgs {
    const uint64_t* words;
    const uint64_t* woff;
    const uint32_t* len;
    const uint32_t* tile_read;
    const uint32_t* tile_word0;
    uint32_t tile_begin, tile_end;
    uint32_t m;       // effective min_length (>= 1)
    uint64_t kmask;   // low K*BITS bits
    const uint32_t* bloom;
    uint32_t bloom_log2;
    const Slot* table;
    uint32_t tbits;
    const uint32_t* chain;
    uint32_t* selfrep;     // COUNT: min p>0 at which a read's own prefix K-mer recurs
    uint32_t* tile_count;  // COUNT out
    const uint32_t* tile_off;  // FILL in
    uint32_t* cand_a;
    uint32_t* cand_p;
    uint32_t* cand_b;
};

enum { SCAN_COUNT = 0, SCAN_FILL = 1 };

// Persistent workgroups (grid <= #CUs), 16 waves each; the Bloom filter lives in LDS for the whole
// launch.  One wave per tile = 64 consecutive words of one read = 64*W positions.  Lane l owns word
// l (+ the next one for windows that straddle) and tests its W positions against the filter; the
// survivors of the whole wave are compacted through a wave-private LDS queue so that the L2 table
// probes run with all lanes busy, in ascending p.
template <int BITS, int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(const ScanArgs A) {
    constexpr int W = 64 / BITS;
    extern __shared__ uint64_t smem[];
    const uint32_t nwaves = blockDim.x >> 6;
    uint64_t* q_kmer = smem;                                            // nwaves * 64 * 8 B
    uint32_t* q_p = reinterpret_cast<uint32_t*>(smem + nwaves * WAVE);  // nwaves * 64 * 4 B
    uint32_t* s_bloom = q_p + nwaves * WAVE;
    const uint32_t bloom_words = (1u << A.bloom_log2) >> 5;
    for (uint32_t i = threadIdx.x; i < bloom_words; i += blockDim.x) s_bloom[i] = A.bloom[i];
    __syncthreads();

    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint64_t* qk = q_kmer + wave * WAVE;
    uint32_t* qp = q_p + wave * WAVE;
    const uint32_t bshift = 32 - A.bloom_log2;

    for (uint32_t t = A.tile_begin + blockIdx.x * nwaves + wave; t < A.tile_end; t += gridDim.x * nwaves) {
        const uint32_t a = A.tile_read[t];
        const uint32_t la = A.len[a];
        uint32_t tile_total = 0;
        if (la >= A.m) {  // wave-uniform
            const uint64_t* __restrict__ rw = A.words + A.woff[a];
            const uint32_t nw = (la + W - 1) / W;
            const uint32_t pmax = la - A.m;  // last position whose suffix/containment can reach min_length
            const uint32_t wi = A.tile_word0[t] + lane;
            const uint32_t p0 = wi * W;
            const uint64_t key_a = rw[0] & A.kmask;
            uint32_t hitmask = 0;
            uint64_t w0 = 0, w1 = 0;
            if (wi < nw && p0 <= pmax) {
                w0 = rw[wi];
                w1 = rw[wi + 1];  // guard word after every read keeps this in bounds
#pragma unroll
                for (int s = 0; s < W; ++s) {
                    const uint64_t kmer = (s == 0 ? w0 : ((w0 >> (s * BITS)) | (w1 << ((64 - s * BITS) & 63)))) & A.kmask;
                    uint32_t h1, h2;
                    kmer_hash(kmer, h1, h2);
                    const uint32_t i1 = h1 >> bshift, i2 = h2 >> bshift;
                    const uint32_t b1 = s_bloom[i1 >> 5] >> (i1 & 31);
                    const uint32_t b2 = s_bloom[i2 >> 5] >> (i2 & 31);
                    hitmask |= (b1 & b2 & 1u) << s;
                }
                const uint32_t nvalid = pmax - p0 + 1;
                if (nvalid < (uint32_t)W) hitmask &= (1u << nvalid) - 1u;
            }
            const uint32_t nh = __popc(hitmask);
            const uint32_t incl = wave_incl_scan(nh);
            const uint32_t total = read_last_lane(incl);
            uint32_t rank = incl - nh;
            uint32_t base = 0;
            if (MODE == SCAN_FILL) base = A.tile_off[t];
            for (uint32_t r0 = 0; r0 < total; r0 += WAVE) {
                while (hitmask && rank < r0 + WAVE) {
                    const uint32_t s = __ffs(hitmask) - 1;
                    hitmask &= hitmask - 1;
                    qk[rank - r0] = funnel(w0, w1, s * BITS) & A.kmask;
                    qp[rank - r0] = p0 + s;
                    ++rank;
                }
                wave_lds_fence();
                const bool has = r0 + lane < total;
                const uint64_t kmer = qk[lane];
                const uint32_t p = qp[lane];
                wave_lds_fence();
                uint32_t start = 0, cnt = 0;
                if (has) table_probe(A.table, A.tbits, kmer, start, cnt);
                const bool self = cnt != 0 && kmer == key_a;  // a's own chain entry is not a candidate
                const uint32_t ceff = cnt - (self ? 1u : 0u);
                if (MODE == SCAN_COUNT) {
                    if (self && p > 0) atomicMin(&A.selfrep[a], p);
                    tile_total += ceff;
                } else {
                    const uint32_t inc = wave_incl_scan(ceff);
                    uint32_t off = base + inc - ceff;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const uint32_t b = A.chain[start + j];
                        if (b != a) {
                            A.cand_a[off] = a;
                            A.cand_p[off] = p;
                            A.cand_b[off] = b;
                            ++off;
                        }
                    }
                    base += read_last_lane(inc);
                }
            }
            if (MODE == SCAN_COUNT) tile_total = wave_sum(tile_total);
        }
        if (MODE == SCAN_COUNT && lane == 0) A.tile_count[t] = tile_total;
    }
}

// selfrep for reads OUTSIDE the a-side shard of this call (multi-GPU / sharded calls): the COUNT
// pass only visits the shard's reads, but the select step needs selfrep[b] for every b.
template <int BITS>
__global__ __launch_bounds__(256) void k_selfrep(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                 const uint32_t* __restrict__ len, const uint32_t* __restrict__ tile_read,
                                                 const uint32_t* __restrict__ tile_word0, uint32_t n_tiles,
                                                 uint32_t skip_begin, uint32_t skip_end, uint32_t m,
                                                 uint64_t kmask, uint32_t* selfrep) {
    constexpr int W = 64 / BITS;
    const uint32_t lane = lane_id();
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < n_tiles; t += nwaves) {
        if (t >= skip_begin && t < skip_end) continue;
        const uint32_t a = tile_read[t];
        const uint32_t la = len[a];
        if (la < m) continue;
        const uint64_t* __restrict__ rw = words + woff[a];
        const uint32_t nw = (la + W - 1) / W, pmax = la - m;
        const uint32_t wi = tile_word0[t] + lane, p0 = wi * W;
        const uint64_t key_a = rw[0] & kmask;
        uint32_t best = NO_SELFREP;
        if (wi < nw && p0 <= pmax) {
            const uint64_t w0 = rw[wi], w1 = rw[wi + 1];
#pragma unroll
            for (int s = W - 1; s >= 0; --s) {
                const uint64_t kmer = (s == 0 ? w0 : ((w0 >> (s * BITS)) | (w1 << ((64 - s * BITS) & 63)))) & kmask;
                const uint32_t p = p0 + s;
                if (kmer == key_a && p > 0 && p <= pmax) best = p;
            }
        }
        if (best != NO_SELFREP) atomicMin(&selfrep[a], best);
    }
}

// ----------------------------------------------------------------------------------------
// verify: packed exact compare of a[p : p+n) against b[0 : n)
// ----------------------------------------------------------------------------------------
// 16 lanes per candidate, one 64-bit word of b per lane per step (the matching window of a is
// funnel-shifted out of two words).  type: bit0 = suffix-prefix (A) candidate holds, bit1 = b is
// wholly contained at p (B).  0 = mismatch.
constexpr int VER_GROUP = 16;

template <int BITS>
__global__ __launch_bounds__(256) void k_verify(const uint64_t* __restrict__ words, const uint64_t* __restrict__ woff,
                                                const uint32_t* __restrict__ len, const uint32_t* __restrict__ cand_a,
                                                const uint32_t* __restrict__ cand_p, const uint32_t* __restrict__ cand_b,
                                                uint32_t n_cand, uint8_t* __restrict__ type) {
    constexpr int W = 64 / BITS;
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / VER_GROUP;
    const uint32_t sub = threadIdx.x & (VER_GROUP - 1);
    const uint32_t gshift = (lane_id() / VER_GROUP) * VER_GROUP;
    bool live = gid < n_cand;
    uint32_t n = 0, nwords = 0, sh = 0, rem = 0, lb = 0;
    const uint64_t* A = words;
    const uint64_t* B = words;
    if (live) {
        const uint32_t a = cand_a[gid], p = cand_p[gid], b = cand_b[gid];
        rem = len[a] - p;
        lb = len[b];
        n = rem < lb ? rem : lb;
        nwords = (n + W - 1) / W;
        const uint64_t bitpos = (uint64_t)p * BITS;
        A = words + woff[a] + (bitpos >> 6);
        sh = (uint32_t)(bitpos & 63);
        B = words + woff[b];
    }
    bool ok = true;
    uint32_t c0 = 0;
    bool active = live && c0 < nwords;
    while (__any(active)) {
        uint64_t diff = 0;
        const uint32_t c = c0 + sub;
        if (active && c < nwords) {
            const uint64_t bw = B[c];
            const uint64_t av = funnel(A[c], A[c + 1], sh);
            const uint32_t valid = n - c * W;  // bases of this word inside the compared range
            const uint64_t mask = valid >= (uint32_t)W ? ~0ull : ((1ull << (valid * BITS)) - 1ull);
            diff = (av ^ bw) & mask;
        }
        const uint64_t bal = __ballot(diff != 0);
        if (active) {
            if ((bal >> gshift) & ((1ull << VER_GROUP) - 1ull)) {
                ok = false;
                active = false;
            } else {
                c0 += VER_GROUP;
                active = c0 < nwords;
            }
        }
    }
    if (live && sub == 0) {
        uint8_t t = 0;
        if (ok) t = (uint8_t)((rem <= lb ? 1u : 0u) | (rem >= lb ? 2u : 0u));
        type[gid] = t;
    }
}

// ----------------------------------------------------------------------------------------
// select: rows per candidate.  B rows: every occurrence.  A rows: only the longest per (a,b),
// i.e. the smallest p.  Candidates of one `a` are contiguous and in ascending p, so an A
// candidate loses iff an earlier candidate of the same (a,b) also verified as A.  Two A hits
// for one (a,b) force b's prefix K-mer to recur inside b (period p-p'), so only reads with
// selfrep[b] set need the look-back.
// ----------------------------------------------------------------------------------------
__global__ void k_select(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_b,
                         const uint8_t* __restrict__ type, uint32_t n_cand, const uint32_t* __restrict__ selfrep,
                         const uint32_t* __restrict__ read_tile0, const uint32_t* __restrict__ tile_off,
                         uint8_t* __restrict__ rowcnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const uint32_t t = type[i];
    uint32_t cnt = (t & 1u) + ((t >> 1) & 1u);
    if (t & 1u) {
        const uint32_t b = cand_b[i];
        if (selfrep[b] != NO_SELFREP) {
            const uint32_t seg0 = tile_off[read_tile0[cand_a[i]]];
            for (uint32_t j = seg0; j < i; ++j) {
                if (cand_b[j] == b && (type[j] & 1u)) {
                    --cnt;
                    break;
                }
            }
        }
    }
    rowcnt[i] = (uint8_t)cnt;
}

__global__ __launch_bounds__(256) void k_emit(const uint32_t* __restrict__ cand_a, const uint32_t* __restrict__ cand_p,
                                              const uint32_t* __restrict__ cand_b, const uint8_t* __restrict__ type,
                                              const uint8_t* __restrict__ rowcnt, const uint32_t* __restrict__ row_off,
                                              uint32_t n_cand, const uint32_t* __restrict__ len, Row* __restrict__ rows,
                                              uint32_t bits,
                                              unsigned long long* __restrict__ counters /* [0]=verified [1]=sum l [2]=sum 2*ceil(l*bits/8) */) {
    __shared__ uint64_t s_red[3][256 / WAVE];
    uint64_t nver = 0, suml = 0, sumb = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += gridDim.x * blockDim.x) {
        const uint32_t t = type[i], rc = rowcnt[i];
        nver += t != 0;
        if (rc) {
            const uint32_t a = cand_a[i], p = cand_p[i], b = cand_b[i];
            const uint32_t la = len[a], lb = len[b];
            uint32_t off = row_off[i];
            const bool emit_a = (t & 1u) && rc == (t & 1u) + ((t >> 1) & 1u);
            if (emit_a) {
                Row r = {a, b, (int32_t)p, (int32_t)la, 0, (int32_t)(la - p)};
                rows[off++] = r;
                suml += la - p;
                sumb += 2ull * (((uint64_t)(la - p) * bits + 7) / 8);
            }
            if (t & 2u) {
                Row r = {a, b, (int32_t)p, (int32_t)(p + lb), 0, (int32_t)lb};
                rows[off] = r;
                suml += lb;
                sumb += 2ull * (((uint64_t)lb * bits + 7) / 8);
            }
        }
    }
    nver = wave_sum64(nver);
    suml = wave_sum64(suml);
    sumb = wave_sum64(sumb);
    if (lane_id() == 0) {
        s_red[0][threadIdx.x >> 6] = nver;
        s_red[1][threadIdx.x >> 6] = suml;
        s_red[2][threadIdx.x >> 6] = sumb;
    }
    __syncthreads();
    if (threadIdx.x < 3) {  // one atomic per counter per workgroup
        uint64_t v = 0;
        for (int w = 0; w < 256 / WAVE; ++w) v += s_red[threadIdx.x][w];
        if (v) atomicAdd(&counters[threadIdx.x], (unsigned long long)v);
    }
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
    for (int k = 0; k < PS_ITEMS; ++k) {
        const uint64_t i = base + (uint64_t)k * PS_BLOCK + threadIdx.x;
        if (i < n) acc += in[i];
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

// single workgroup: exclusive scan of the block sums in place, grand total to *total
__global__ __launch_bounds__(1024) void k_ps_spine(uint64_t* __restrict__ block_sums, uint32_t nblocks, uint64_t* __restrict__ total) {
    __shared__ uint64_t s_part[1024];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per;
    const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
    uint64_t acc = 0;
    for (uint32_t i = lo; i < hi; ++i) acc += block_sums[i];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const uint64_t v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    uint64_t run = s_part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t v = block_sums[i];
        block_sums[i] = run;
        run += v;
    }
}

template <typename T>
__global__ __launch_bounds__(PS_BLOCK) void k_ps_down(const T* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_sums,
                                                      uint32_t* __restrict__ out) {
    // thread owns PS_ITEMS consecutive items; block-wide scan of the per-thread sums
    __shared__ uint32_t s_wave[PS_BLOCK / WAVE];
    const uint64_t base = (uint64_t)blockIdx.x * PS_TILE + (uint64_t)threadIdx.x * PS_ITEMS;
    uint32_t v[PS_ITEMS];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) {
        const uint64_t i = base + k;
        v[k] = i < n ? (uint32_t)in[i] : 0u;
        acc += v[k];
    }
    const uint32_t incl = wave_incl_scan(acc);
    if (lane_id() == WAVE - 1) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wave_base += s_wave[w];
    uint32_t run = (uint32_t)block_sums[blockIdx.x] + wave_base + incl - acc;
#pragma unroll
    for (int k = 0; k < PS_ITEMS; ++k) {
        const uint64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}

}  // namespace po
