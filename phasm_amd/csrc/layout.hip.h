// Device side of `phasm layout` stage 1: what the reference does to
// every E line before graph cleaning starts --
//   LocalAlignment.classify / get_overlap_length / get_overhang   phasm/alignments.py:239-258
//   ContainedReads, MinReadLength, MinOverlapLength, MaxOverhang   phasm/filter.py:37-122
//   build_assembly_graph                                           phasm/assembly_graph.py:136-179
//   removal of every filtered read in both orientations            phasm/cli/assembler.py:113-126
// restated as order-independent data-parallel passes over the 24-byte row array.
//
// The reference walks the E lines once, in file order, with stateful filters.  Its FINAL edge set
// does not depend on that order (every contained read loses both of its nodes at the end,
// assembler.py:113-126, whenever it was discovered), so the passes below compute
//   C      = reads that are the contained side of some row                       (k_layout_classify)
//   pass   = rows of an OVERLAP type that satisfy the three per-row predicates    (k_layout_classify)
//   edges  = the two edges of every pass row with neither read in C, one record per distinct
//            (u, v): networkx's add_edge overwrites the attributes of an existing edge, so the
//            row that comes LAST in the array owns the edge    (k_layout_insert / _winner / _emit)
//            -- a hash table keyed by the twin pair {(u, v), (v^1, u^1)} with an atomic max of the row
// Nodes are oriented-read indices; the reverse node of x is x ^ 1 (reads are added as x+ / x-).
#pragma once

namespace po {

struct Edge {
    uint32_t u, v;
    int32_t weight, overlap_len;  // g[u][v]['weight'], g[u][v]['overlap_len']  (assembly_graph.py:147-150)
};

struct LayoutParams {
    uint32_t min_read_length;     // 0: MinReadLength not installed (assembler.py:80)
    uint32_t min_overlap_length;  // 0: MinOverlapLength not installed (assembler.py:83)
    uint32_t max_overhang_abs;
    uint32_t pad;
    double max_overhang_rel;
};

// AlignmentType, phasm/alignments.py:16-20
enum : uint32_t { LT_OVERLAP_AB = 0, LT_OVERLAP_BA = 1, LT_A_CONTAINED = 2, LT_B_CONTAINED = 3 };
// rflag byte per row
enum : uint32_t { RF_TYPE = 3u, RF_PASS = 4u, RF_SHADOW = 8u, RF_INVALID = 128u };
// counters (u64 each)
enum { LC_TYPE0 = 0, LC_SHORT = 4, LC_MINOVL = 5, LC_OVERHANG = 6, LC_PASS = 7, LC_INVALID = 8, LC_N = 9 };

constexpr unsigned long long EDGE_EMPTY = ~0ull;

// one slot of the (u, v) -> last-writer table: key and writer share a 16-byte slot so that the claim,
// the atomicMax and the later lookup touch one cache line
struct __attribute__((aligned(16))) EdgeSlot {
    unsigned long long key;  // u << 32 | v; EDGE_EMPTY = free (memset 0xFF)
    uint32_t writer;         // ~row: memset 0xFF = "no writer"; the smallest complement = the last row
    uint32_t pad;
};

template <int N>
__device__ inline void block_add(const uint64_t (&v)[N], unsigned long long* __restrict__ counters) {
    __shared__ uint64_t s_red[N][256 / WAVE];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint64_t s = wave_sum64(v[k]);
        if (lane_id() == 0) s_red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        uint64_t s = 0;
        for (int w = 0; w < 256 / WAVE; ++w) s += s_red[threadIdx.x][w];
        if (s) atomicAdd(&counters[threadIdx.x], (unsigned long long)s);
    }
}

// classify() of one row, phasm/alignments.py:248-258, on 64-bit integers
__device__ inline uint32_t classify_row(int64_t as, int64_t ae, int64_t bs, int64_t be, int64_t la, int64_t lb) {
    const int64_t ra = la - ae, rb = lb - be;
    if (as <= bs && ra <= rb) return LT_A_CONTAINED;
    if (as >= bs && ra >= rb) return LT_B_CONTAINED;
    return as >= bs ? LT_OVERLAP_AB : LT_OVERLAP_BA;
}

// The two edges build_assembly_graph adds for an OVERLAP row (assembly_graph.py:146-176).
__device__ inline void row_edges(const Row& r, uint32_t type, int64_t la, int64_t lb, Edge& e1, Edge& e2) {
    const int64_t as = r.astart, ae = r.aend, bs = r.bstart, be = r.bend;
    const int64_t ovl = (ae - as) > (be - bs) ? (ae - as) : (be - bs);  // get_overlap_length, alignments.py:239-241
    if (type == LT_OVERLAP_AB) {
        e1 = Edge{r.a_idx, r.b_idx, (int32_t)(as - bs), (int32_t)ovl};
        e2 = Edge{r.b_idx ^ 1u, r.a_idx ^ 1u, (int32_t)((lb - be) - (la - ae)), (int32_t)ovl};
    } else {
        e1 = Edge{r.b_idx, r.a_idx, (int32_t)(bs - as), (int32_t)ovl};
        e2 = Edge{r.a_idx ^ 1u, r.b_idx ^ 1u, (int32_t)((la - ae) - (lb - be)), (int32_t)ovl};
    }
}

// slot in [0, n_slots): multiply-shift range reduction, no power-of-two table needed
__device__ inline uint32_t edge_slot(uint32_t u, uint32_t v, uint32_t n_slots) {
    const unsigned long long k = (((unsigned long long)u << 32) | v) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(((k >> 32) * (unsigned long long)n_slots) >> 32);
}

// Pass 1: type + per-row predicates -> rflag; contained reads -> removed[read] (read = node >> 1).
__global__ __launch_bounds__(256) void k_layout_classify(const Row* __restrict__ rows, uint32_t n_rows,
                                                         const uint32_t* __restrict__ len, uint32_t n_nodes,
                                                         LayoutParams prm, uint8_t* __restrict__ rflag,
                                                         uint8_t* __restrict__ removed,
                                                         unsigned long long* __restrict__ counters) {
    uint64_t c[LC_N] = {};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        const Row r = rows[i];
        if (r.a_idx >= n_nodes || r.b_idx >= n_nodes) {
            rflag[i] = (uint8_t)RF_INVALID;
            c[LC_INVALID] += 1;
            continue;
        }
        const int64_t la = len[r.a_idx], lb = len[r.b_idx];
        const int64_t as = r.astart, ae = r.aend, bs = r.bstart, be = r.bend;
        const uint32_t type = classify_row(as, ae, bs, be, la, lb);
        c[LC_TYPE0 + type] += 1;
        uint32_t f = type;
        if (type == LT_A_CONTAINED) {
            removed[r.a_idx >> 1] = 1;  // ContainedReads.nodes_to_remove.add(la.a), filter.py:93-95
        } else if (type == LT_B_CONTAINED) {
            removed[r.b_idx >> 1] = 1;  // filter.py:96-98
        } else {
            const int64_t ovl = (ae - as) > (be - bs) ? (ae - as) : (be - bs);
            const int64_t ra = la - ae, rb = lb - be;
            const int64_t overhang = (as < bs ? as : bs) + (ra < rb ? ra : rb);  // alignments.py:243-246
            // MaxOverhang: overhang <= min(max_overhang, ratio * overlap_length), filter.py:119-122
            // (Python compares int with float exactly; every operand here is exact in a double)
            double thr = prm.max_overhang_rel * (double)ovl;
            if ((double)prm.max_overhang_abs < thr) thr = (double)prm.max_overhang_abs;
            if (prm.min_read_length && (la < (int64_t)prm.min_read_length || lb < (int64_t)prm.min_read_length)) {
                c[LC_SHORT] += 1;  // MinReadLength, filter.py:45-52
            } else if (prm.min_overlap_length && ovl < (int64_t)prm.min_overlap_length) {
                c[LC_MINOVL] += 1;  // MinOverlapLength, filter.py:73-74
            } else if (!((double)overhang <= thr)) {
                c[LC_OVERHANG] += 1;
            } else {
                f |= RF_PASS;
                c[LC_PASS] += 1;
            }
        }
        rflag[i] = (uint8_t)f;
    }
    block_add<LC_N>(c, counters);
}

// The two edges of a row are twins: e2 = (v1 ^ 1, u1 ^ 1).  Every row that writes (u, v) also writes its
// twin, so both edges have the same set of writers and the same LAST writer row: one table entry per twin
// pair -- keyed by the smaller of the two 64-bit keys -- is enough.  (u == v ^ 1 makes the two edges one
// and the same; the reference then applies edge 2 after edge 1, so edge 2's attributes stay.)
__device__ inline unsigned long long pair_key(const Edge& e1, const Edge& e2, bool& self_twin) {
    const unsigned long long k1 = ((unsigned long long)e1.u << 32) | e1.v;
    const unsigned long long k2 = ((unsigned long long)e2.u << 32) | e2.v;
    self_twin = k1 == k2;
    return k1 < k2 ? k1 : k2;
}

// Pass 2: every surviving row claims the slot of its twin pair; the largest row index stays (stored
// complemented, so one memset(0xFF) initialises keys and writers) -- the row whose add_edge calls the
// reference would have applied last.
__global__ __launch_bounds__(256) void k_layout_insert(const Row* __restrict__ rows, uint32_t n_rows,
                                                       const uint32_t* __restrict__ len,
                                                       uint8_t* rflag,
                                                       const uint8_t* __restrict__ removed,
                                                       EdgeSlot* __restrict__ table, uint32_t n_slots) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        const uint32_t f = rflag[i];
        if (!(f & RF_PASS)) continue;
        const Row r = rows[i];
        if (removed[r.a_idx >> 1] | removed[r.b_idx >> 1]) continue;
        Edge e1, e2;
        row_edges(r, f & RF_TYPE, len[r.a_idx], len[r.b_idx], e1, e2);
        bool self_twin;
        const unsigned long long key = pair_key(e1, e2, self_twin);
        // A row whose successor writes the same twin pair can never be the last writer: it stays out of the
        // table (and k_layout_winner skips it).  The rows of po_overlaps come as (row, strand mirror) pairs, so
        // this halves the atomics -- device-scope atomics are what this pass costs.
        if (i + 1 < n_rows) {
            const uint32_t f2 = rflag[i + 1];
            if (f2 & RF_PASS) {  // (same twin pair => same two reads => same contained-read status)
                const Row r2 = rows[i + 1];
                Edge g1, g2;
                row_edges(r2, f2 & RF_TYPE, len[r2.a_idx], len[r2.b_idx], g1, g2);
                bool st2;
                if (pair_key(g1, g2, st2) == key) {
                    rflag[i] = (uint8_t)(f | RF_SHADOW);
                    continue;
                }
            }
        }
        uint32_t s = edge_slot((uint32_t)(key >> 32), (uint32_t)key, n_slots);
        for (;;) {
            unsigned long long cur = table[s].key;
            if (cur == EDGE_EMPTY) cur = atomicCAS(&table[s].key, EDGE_EMPTY, key);
            if (cur == EDGE_EMPTY || cur == key) break;
            if (++s == n_slots) s = 0;
        }
        atomicMin(&table[s].writer, ~i);
    }
}

// Pass 3: does this row own its twin pair?  ewin bit k = edge k is emitted; ecnt = how many.
__global__ __launch_bounds__(256) void k_layout_winner(const Row* __restrict__ rows, uint32_t n_rows,
                                                       const uint32_t* __restrict__ len,
                                                       const uint8_t* __restrict__ rflag,
                                                       const uint8_t* __restrict__ removed,
                                                       const EdgeSlot* __restrict__ table, uint32_t n_slots,
                                                       uint8_t* __restrict__ ecnt, uint8_t* __restrict__ ewin) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t f = rflag[i];
    uint32_t win = 0;
    if ((f & RF_PASS) && !(f & RF_SHADOW)) {
        const Row r = rows[i];
        if (!(removed[r.a_idx >> 1] | removed[r.b_idx >> 1])) {
            Edge e1, e2;
            row_edges(r, f & RF_TYPE, len[r.a_idx], len[r.b_idx], e1, e2);
            bool self_twin;
            const unsigned long long key = pair_key(e1, e2, self_twin);
            uint32_t s = edge_slot((uint32_t)(key >> 32), (uint32_t)key, n_slots);
            for (;;) {  // k_layout_insert put the key there
                const uint4 q = *reinterpret_cast<const uint4*>(&table[s]);
                if ((((unsigned long long)q.y << 32) | q.x) == key) {
                    if (q.z == ~i) win = self_twin ? 2u : 3u;
                    break;
                }
                if (++s == n_slots) s = 0;
            }
        }
    }
    ewin[i] = (uint8_t)win;
    ecnt[i] = (uint8_t)__popc(win);
}

// Passes 2 + 3 in one, without the table, for rows that come straight from this library's paired-strand emission
// (k_emit / k_emit_cands with `paired`): a twin pair {(u, v), (v^1, u^1)} is then written by exactly ONE verified
// candidate -- A rows are unique per ordered pair, the candidate's strand mirror IS the twin, and k_emit puts the
// mirror row right behind the row -- so the last writer of a pair is simply the last row of its adjacent group.  No
// CAS, no atomicMin, no 224 MB table to initialise (0.54 -> 0.1 ms at config 2).
__global__ __launch_bounds__(256) void k_layout_winner_adjacent(const Row* __restrict__ rows, uint32_t n_rows,
                                                                const uint32_t* __restrict__ len,
                                                                const uint8_t* __restrict__ rflag,
                                                                const uint8_t* __restrict__ removed,
                                                                uint8_t* __restrict__ ecnt, uint8_t* __restrict__ ewin) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t f = rflag[i];
    uint32_t win = 0;
    if (f & RF_PASS) {
        const Row r = rows[i];
        if (!(removed[r.a_idx >> 1] | removed[r.b_idx >> 1])) {
            Edge e1, e2;
            row_edges(r, f & RF_TYPE, len[r.a_idx], len[r.b_idx], e1, e2);
            bool self_twin;
            const unsigned long long key = pair_key(e1, e2, self_twin);
            bool shadowed = false;   // the next row writes the same pair: it is the later writer
            if (i + 1 < n_rows) {
                const uint32_t f2 = rflag[i + 1];
                if (f2 & RF_PASS) {
                    const Row r2 = rows[i + 1];
                    Edge g1, g2;
                    row_edges(r2, f2 & RF_TYPE, len[r2.a_idx], len[r2.b_idx], g1, g2);
                    bool st2;
                    shadowed = pair_key(g1, g2, st2) == key;
                }
            }
            if (!shadowed) win = self_twin ? 2u : 3u;
        }
    }
    ewin[i] = (uint8_t)win;
    ecnt[i] = (uint8_t)__popc(win);
}

// Pass 4: edges of row i at edges[eoff[i]...], edge 1 before edge 2.
__global__ __launch_bounds__(256) void k_layout_emit(const Row* __restrict__ rows, uint32_t n_rows,
                                                     const uint32_t* __restrict__ len,
                                                     const uint8_t* __restrict__ rflag,
                                                     const uint8_t* __restrict__ ewin,
                                                     const uint32_t* __restrict__ eoff, Edge* __restrict__ edges) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t win = ewin[i];
    if (!win) return;
    const Row r = rows[i];
    Edge e1, e2;
    row_edges(r, rflag[i] & RF_TYPE, len[r.a_idx], len[r.b_idx], e1, e2);
    uint32_t o = eoff[i];
    if (win & 1u) edges[o++] = e1;
    if (win & 2u) edges[o] = e2;
}

// number of removed reads (bytes set in removed[])
__global__ __launch_bounds__(256) void k_count_bytes(const uint8_t* __restrict__ v, uint32_t n,
                                                     unsigned long long* __restrict__ out) {
    uint64_t c[1] = {0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c[0] += v[i] != 0;
    block_add<1>(c, out);
}

}  // namespace po
