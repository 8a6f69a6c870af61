// Host side of libphasm_overlap.so: the C ABI declared in include/phasm_overlap.h.
// Holds the read set (2-bit or 8-bit packed 64-bit words), keeps a device-resident copy, and
// drives the kernels of kernels.hip.h on the handle's own HIP stream.  No CPU compute path.
#include "../../include/phasm_overlap.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "kernels.hip.h"
#include "extend.hip.h"
#include "layout.hip.h"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool arena = false;   // carved out of the handle's arena (po_handle::arena_*): lives until the handle dies
    template <typename T> T* as() const { return static_cast<T*>(p); }
    void release() {
        if (p && !arena) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        arena = false;
    }
};

// Every range of HOST memory this library makes visible to the GPU (hipHostMalloc, hipHostRegister) is noted here, for
// the whole life of the process: po_debug_host_ranges hands the list out, and the tests assert that no page of the reads
// they handed to po_add_sequence ever lay inside one (the library's device->host copies and
// host-mapped stores can only reach memory that is on this list).
struct PinNote {
    uint64_t base, bytes;
    uint32_t kind;   // 1 hipHostMalloc, 2 hipHostRegister (packed read store), bit 8: still live
};
std::mutex g_pin_mu;
std::vector<PinNote> g_pins;
uint64_t g_pins_ever = 0;
constexpr size_t PIN_LOG_MAX = 1u << 16;
void pin_note(const void* p, size_t bytes, uint32_t kind) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pin_mu);
    ++g_pins_ever;
    for (PinNote& n : g_pins)   // (the same range again: one entry)
        if (n.base == (uint64_t)(uintptr_t)p && n.bytes == bytes && (n.kind & 0xFFu) == kind) {
            n.kind |= 0x100u;
            return;
        }
    if (g_pins.size() < PIN_LOG_MAX) g_pins.push_back(PinNote{(uint64_t)(uintptr_t)p, (uint64_t)bytes, kind | 0x100u});
}
void pin_drop(const void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pin_mu);
    for (PinNote& n : g_pins)
        if (n.base == (uint64_t)(uintptr_t)p && (n.kind & 0x100u)) n.kind &= ~0x100u;
}

// host-pinned buffer (hipHostMalloc): every device->host copy of this library lands in one of these, never in
// pageable heap memory -- the HIP runtime then neither stages the copy nor pins (and caches the pin of) a range
// of the caller's malloc heap
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    void release() {
        if (p) {
            pin_drop(p);
            (void)hipHostFree(p);
        }
        p = nullptr;
        cap = 0;
    }
};

// Stage-boundary timing events (po_stats.ms_index .. ms_emit).  A recorded event is a marker the command processor has to
// retire between two kernels (~5 us each, measured): a whole-set call records all of them, the pieces of a streamed step only
// the pairs around the two big kernels (ms_scan_probe, ms_verify_kernel) unless PHASM_PHASE_EVENTS=1 / PHASM_STREAM_TRACE ask.
inline int phase_events_env() {
    // -1: default (whole-set calls record every stage boundary, the pieces of a streamed step NOTHING but their end -- round 4:
    // with the rows going home as records the step is paced by the device, and seven markers per piece were 35 us of it);
    // 0: nothing but the end anywhere; 1: every boundary everywhere; 2: pieces record the pairs around their two big kernels
    // (ms_scan_probe, ms_verify_kernel) and their start / end (ms_total) -- what bench.py's extra steps ask for
    if (const char* e = getenv("PHASM_PHASE_EVENTS")) {
        const int v = atoi(e);
        return v == 1 ? 1 : v == 2 ? 2 : 0;
    }
    return getenv("PHASM_STREAM_TRACE") ? 1 : -1;
}
enum { EV_START = 0, EV_INDEX, EV_COUNT, EV_FILL, EV_VERIFY, EV_SELECT, EV_EMIT, EV_PROBE0, EV_PROBE1, EV_VER0, EV_VER1, EV_DONE, EV_N };

}  // namespace

constexpr int PO_MAX_PIECES = 16;

// The packed host stores live in memory that is REGISTERED with the HIP runtime from the moment it is allocated
// (page-aligned malloc + hipHostRegister, portable across devices): po_add_sequence grows the stores, so the cost of
// pinning is paid there, piece by piece as the vector doubles -- not inside the first po_overlaps call, whose H2D then
// runs at the DMA rate straight away (a cold call used to start with 2-4 ms of hipHostRegister).  Without a GPU (or
// with PHASM_NO_PIN) the registration fails or is skipped and the memory is plain heap: the copy works either way.
struct RegHeader {
    size_t bytes;
    int registered;
};
constexpr size_t REG_PAGE = 4096;
thread_local bool g_reg_defer = false;   // (see RegAlloc::reg_now)
template <class T>
struct RegAlloc {
    using value_type = T;
    RegAlloc() = default;
    template <class U> RegAlloc(const RegAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = ((n * sizeof(T) + REG_PAGE - 1) / REG_PAGE) * REG_PAGE;
        void* base = nullptr;
        if (posix_memalign(&base, REG_PAGE, bytes + REG_PAGE) != 0 || !base) throw std::bad_alloc();
        RegHeader* hd = static_cast<RegHeader*>(base);
        hd->bytes = bytes;
        hd->registered = 0;
        char* data = static_cast<char*>(base) + REG_PAGE;
        if (getenv("PHASM_POISON_HOST")) std::memset(data, 0xA5, bytes);   // (tests: a word nobody wrote shows)
        if (!g_reg_defer) reg_now(data);
        return reinterpret_cast<T*>(data);
    }
    // page-lock the block behind `data` (allocate() does it at once, unless the caller has asked to do it later: po_add_fasta
    // registers the stores on a thread of its own WHILE the packing threads fill them -- the call blocks until the runtime is
    // up and pins 4 KB pages one by one, 40-60 ms per 190 MB store, which used to sit in front of the packing)
    static void reg_now(void* data) {
        RegHeader* hd = reinterpret_cast<RegHeader*>(static_cast<char*>(data) - REG_PAGE);
        if (hd->registered || hd->bytes < (1u << 20) || getenv("PHASM_NO_PIN")) return;
        if (hipHostRegister(data, hd->bytes, hipHostRegisterPortable) == hipSuccess) {
            hd->registered = 1;
            pin_note(data, hd->bytes, 2u);
        } else {
            (void)hipGetLastError();
        }
    }
    // vector::resize(n) (no value) leaves the new words as they are: the caller writes every one of them
    template <class U> void construct(U*) noexcept {}
    template <class U, class A0, class... Args> void construct(U* q, A0&& a0, Args&&... args) {
        ::new (static_cast<void*>(q)) U(std::forward<A0>(a0), std::forward<Args>(args)...);
    }
    void deallocate(T* p, size_t) {
        char* base = reinterpret_cast<char*>(p) - REG_PAGE;
        RegHeader* hd = reinterpret_cast<RegHeader*>(base);
        if (hd->registered) {
            pin_drop(p);
            (void)hipHostUnregister(p);
        }
        free(base);
    }
    template <class U> bool operator==(const RegAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const RegAlloc<U>&) const { return false; }
};
using WordStore = std::vector<uint64_t, RegAlloc<uint64_t>>;
inline bool store_is_pinned(const WordStore& w) {
    if (w.capacity() == 0) return false;
    const RegHeader* hd = reinterpret_cast<const RegHeader*>(reinterpret_cast<const char*>(w.data()) - REG_PAGE);
    return hd->registered != 0;
}
constexpr size_t STAGE_MAX = 4u << 20;   // larger sources are registered stores (or, failing that, go as they are)

struct po_handle {
    int device = 0;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev_sets[2][EV_N] = {};   // (two sets: the pieces of a streamed step alternate, a piece's stage times are read one piece later)
    hipEvent_t* ev = ev_sets[0];
    hipEvent_t ev_up0 = nullptr, ev_up1 = nullptr;
    uint64_t* pinned = nullptr;  // host-pinned landing zone for device totals/counters (128 x u64; [64..95]: the two zones of a streamed step's pieces)
    uint64_t* pinned_dev = nullptr;  // the same memory as the device sees it (kernels write totals there directly)
    int n_cu = 256;
    size_t lds_max = 64 * 1024;
    std::string err;

    // host read store
    int bits = 2;
    std::vector<std::string> ids;
    std::vector<uint32_t> len;
    std::vector<uint64_t> cum_len;   // running sum of len (shard_range), extended as reads are added
    // Packed reads live in TWO host stores: words[0] holds the reads with an even index, words[1] those with an odd
    // index, woff[r] is read r's first word inside ITS store.  `phasm overlap` adds every read as (x, revcomp x):
    // when every odd read is exactly the reverse complement of its even partner (all_pairs_rc, checked word by word
    // as the reads arrive) only store 0 crosses PCIe and store 1 is rebuilt on the device -- half the H2D bytes.
    // On the device the two stores are one buffer [store 0 | store 1 | padding]; d_woff holds absolute offsets.
    std::vector<uint64_t> woff;
    WordStore words[2];
    bool all_pairs_rc = true;   // every complete (even, odd) pair so far: same length, odd == revcomp(even), no exception records
    bool all_pairs_rcx = true;  // the same with exception records allowed: such a pair was compared byte by byte on the host (pair_state 1)
    uint64_t base1 = 0;         // first word of store 1 in the device buffer (set at upload)
    uint64_t dev_words = 0;     // words in the device buffer, padding included
    uint64_t upload_bytes = 0;  // bytes the last upload moved host->device
    // 2-bit mode: bytes other than upper-case A/C/G/T are stored as code 0 plus an exception record
    // (position, byte), sorted by read then position; exc_off has one entry per read + 1
    std::vector<uint32_t> exc_off{0};
    std::vector<uint32_t> exc_pos;
    std::vector<uint8_t> exc_byte;
    std::vector<uint8_t> pair_state;  // per read pair: 0 = check codes on device, 1 = verified on host, 2 = not a pair
    uint64_t total_bases = 0;
    bool dirty = true;

    // device read set + tiling (built at upload)
    DevBuf d_words, d_woff, d_len, d_tiles, d_read_tile0, d_exc_off, d_exc_pos, d_exc_byte, d_pair_state;
    std::vector<uint32_t> h_read_tile0;  // n_reads + 1
    uint32_t n_tiles = 0;
    uint32_t max_len = 0;
    size_t n_exc_uploaded = 0;
    bool paired = false;  // every read 2i+1 is the reverse complement of read 2i (checked on device at upload)

    // per-call workspace (grow-only)
    DevBuf d_table, d_slot_cnt, d_slot_cur, d_slot_start, d_read_slot, d_chain, d_chain_tmp, d_long_list;
    DevBuf d_entry_off;   // wide index: offset o of every (read, phase) entry inside its read (k_wide_insert -> k_wide_chain_fill)
    DevBuf d_bloom, d_selfrep, d_tile_count, d_tile_off, d_truemask, d_ps_blocks, d_scalars, d_left, d_left_cnt, d_tile_extra;
    DevBuf d_cand_a, d_cand_p, d_cand_b, d_type, d_rowcnt, d_row_off, d_flag, d_pair_key, d_pair_min;
    DevBuf d_vlabel, d_vrank, d_vperm;  // verify order (k_read_label, k_read_sort, k_read_invert)
    DevBuf d_chain_state;               // ticket / done / status words of the single-pass scans (all-zero between launches)
    DevBuf d_tail_state;                // k_tile_rows / k_tail: row sums per tile of candidates + the done counter
    DevBuf spare_rows;   // device buffers of freed results, kept for the next call (hipFree / hipMalloc of a
    DevBuf spare_cands;  // 50-170 MB buffer costs ~0.2 ms each and synchronises the device)
    DevBuf spare_edges;
    HostBuf spare_host;    // pinned row buffer of a freed result, kept for the next po_result_rows
    HostBuf scratch_host;  // pinned landing zone for small device->host copies into caller memory
    // pinned result pool sized while the reads are added (result_pool_grow): bytes of total_bases it was sized for
    uint64_t pool_bases = 0;
    // streamed step: per-piece workspaces are sized for the LARGEST piece when the first one asks (ws_scale > 1), so
    // that no later piece has to free and re-allocate (hipFree synchronises the device)
    double ws_scale = 1.0;
    // small device workspaces (<= 8 MB each, some forty of them) are carved out of 64 MB chunks instead of being
    // allocated one by one: on a process whose allocator has handed memory back, every small hipMalloc is a trip to
    // the kernel driver (0.2 ms each, 4 ms per first call of a handle)
    std::vector<void*> arena_chunks;
    char* arena_cur = nullptr;
    size_t arena_left = 0;
    int poison = -1;       // PHASM_POISON=<byte>: per-call workspaces are filled with it before every call
    // po_overlaps_ex: the verify step is the banded DP of extend.hip.h (set around the call by po_overlaps_ex)
    // po_overlaps_to_host: a second stream copies chunk k's rows to the host while chunk k + 1 is computed
    hipStream_t copy_stream = nullptr;
    DevBuf chunk_rows[PO_MAX_PIECES + 1];
    // rows home in compact form (namespace home): chunk_compact[k] = chunk_rows[k] holds 16-byte records, not rows; the
    // records land in home_stage (page-locked; a bump allocator that starts over whenever the helper threads have caught up)
    bool chunk_compact[PO_MAX_PIECES + 1] = {};
    bool home_on = false;          // this po_overlaps_to_host call hands its rows home as records where the fused tail runs
    HostBuf home_stage;
    size_t home_used = 0;
    uint64_t home_seq = 0;         // pieces submitted to the helper threads by this call
    uint32_t home_gen = 0;         // number written behind a piece's copy (never repeats on a handle's landing zone)
    // records of 8 bytes instead of 16 where the read set allows it (po::pack_record): the shifts of this call (0 = 16-byte
    // records), and the state of the read set they were worked out for
    uint32_t home_sh_b = 0, home_sh_p = 0;
    uint64_t home_pack_n = ~0ull, home_pack_bases = ~0ull;
    uint32_t home_pack_b = 0, home_pack_p = 0;
    uint64_t home_last_bytes = 0;  // record bytes of the previous call (sizes home_stage)

    // streamed step (po_overlaps_to_host on a changed read set): the packed reads cross PCIe piece by piece on
    // up_stream while the pieces that have arrived go through the kernels and their rows travel back
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_piece[PO_MAX_PIECES] = {};
    // the odd reads (reverse complements) of piece k are written on a stream of their own the moment the piece has landed,
    // beside whatever the handle's stream is doing for the piece before: ev_rc[k] = piece k is complete, both strands
    hipStream_t rc_stream = nullptr;
    hipEvent_t ev_rc[PO_MAX_PIECES] = {};
    // Two-stream pieces (round 4): with the rows going home as records the streamed step is paced by the DEVICE, and a piece
    // is two halves that need different things -- the counting pass (k_scan_probe, k_scan_fixup, the tile prefix sum) needs
    // the piece's own reads and the index, everything behind it (fill, locality order, verify, select, tail) the candidate
    // buffers.  The counting pass of piece k + 1 runs on scan_stream beside the second half of piece k on the handle's
    // stream: ev_s1[k & 1] = piece k's counting pass is done; the small per-piece state both halves touch (scalars,
    // tile offsets) exists twice, by the piece's parity.
    hipStream_t scan_stream = nullptr;
    hipEvent_t ev_s1[2] = {};
    hipEvent_t ev_idx = nullptr;     // the step's index (and everything queued before it on the handle's stream) is complete
    int two_stream = 0;              // this streamed step runs its pieces that way: 1 = on scan_stream; 2 = on rc_stream, and the
                                     // counting pass of piece k + 1 does not start before the verify kernel of piece k has
    hipEvent_t ev_gate[2] = {};      // (ev_gate[k & 1]: recorded on the handle's stream right in front of piece k's verify kernel)
    int want_two = 0;                // what the step asked for (stream_begin queues no reverse complements up front for 2)
    uint32_t st_k = 0;               // the piece run_overlaps is working on (its event in ev_rc)
    hipEvent_t ev_meta = nullptr;
    hipEvent_t ev_first = nullptr;   // the first words of the later pieces are in place
    // first two packed words of every read, appended as the reads are added (registered memory: the streamed step sends
    // them ahead of the pieces straight from here); first_n = reads it covers (a bulk ingest fills it in afterwards)
    WordStore first_words;
    uint32_t first_n = 0;
    // ... and the first LEAD_WORDS words of every read, made when a streamed step first wants them (large read sets: the wide
    // index with windows of 4, whose K-mers reach into the fifth word); st_lead = words per read travelling ahead this step
    WordStore lead_words;
    uint32_t lead_n = 0;
    uint32_t st_lead = 2;
    // per-read metadata of the upload, kept page-locked while the read set is unchanged (reads are only ever appended):
    // [woff x n | len x n | first tile x (n + 1)]; meta_n = reads it covers, meta_bits = encoding it was counted for
    HostBuf meta_host;
    // Small host->device copies out of ordinary heap memory (a packed store below the registration size, the exception
    // records, the pair states) go through this page-locked block: handed a pageable source, the HIP runtime pins -- and
    // keeps a cache of the pin of -- whatever range of the PROCESS heap it lies in, next to the caller's own objects.
    HostBuf stage_host;
    size_t stage_used = 0;
    uint32_t meta_n = 0xFFFFFFFFu;
    int meta_bits = 0;
    uint64_t elig_n = ~0ull, elig_val = 0;   // reads of length >= elig_m among the first elig_n (cache of a 100 k-iteration loop)
    uint32_t elig_m = 0;
    DevBuf d_first, d_defer;
    uint32_t defer_need = 0;       // deferred containment candidates the last streamed call produced
    // a piece whose kernels are queued but whose row count has not been read yet (run_overlaps returned without the
    // closing synchronisation: the next piece's first host wait covers it, st_harvest then sends its rows home)
    struct Pending {
        bool valid = false;
        uint32_t k = 0;
        po_stats S = {};
        bool ver_timed = false;
        bool full_events = true;
        bool pair_events = true;
        bool tail = false;         // the piece's tail ran as k_tail: counts in pinned[zone..], fallback flag in pinned[zone + 7]
        bool compact = false;      // ... as k_tail_cands: the piece's buffer holds records
        int zone = 48;
        uint32_t cap_c = 0;        // > 0: the candidate count was predicted (real count in pinned[zone + 8])
        hipEvent_t* ev = nullptr;
    } st_pend;
    std::function<po_status()> st_harvest;
    // candidates per piece of the last streamed call, valid for the same reads, cuts and min_length (st_pred_sig)
    uint64_t st_pred_cand[PO_MAX_PIECES] = {};
    std::vector<uint32_t> st_pred_sig;
    bool st_pred_valid = false;
    bool phase_events = true;   // this call records the stage-boundary events (phase_events_env)
    bool pair_events = true;    // ... at least the pairs around the two big kernels and the call's start / end
    bool st_selfclean = false;     // the previous piece of this streamed step left the per-call counters zero (no reset launch needed)
    bool st_early_index = false;   // this streamed step builds its index before piece 0 has landed
    bool st_tail_gave_up = false;  // the last streamed step was abandoned because of tandem-repeat reads (statistics / tests)
    bool idx_only = false;         // run_overlaps stops behind the index build (the streamed step builds it ahead of piece 0)
    bool st_on = false;            // run_overlaps works on piece [st_r_begin, st_r_end) of a streamed step
    uint32_t st_r_begin = 0, st_r_end = 0, st_defer_cap = 0;
    uint64_t last_host_rows = 0;   // rows of the previous po_overlaps_to_host call (sizes the pinned buffer up front)
    // the anchor index of the last call, reusable while the device copy of the reads and the parameters it was built
    // for are unchanged (the chunks of po_overlaps_to_host, the shards of a multi-GPU step, repeated calls)
    uint64_t upload_gen = 0;
    bool idx_valid = false;
    uint64_t idx_gen = 0;
    uint32_t idx_m = 0, idx_tbits = 0, idx_bits = 0, idx_ww = 0;
    bool idx_wide = false;
    // sliced wide index (multi-GPU, phasm_amd/dist.py IndexExchange): sl_build_n > 1 makes run_overlaps stop after it has
    // built sub-table sl_build_slice; ext_index makes it probe a gathered sliced index instead of building one
    uint32_t sl_build_slice = 0, sl_build_n = 0;
    bool sl_is_wide = false;
    uint32_t sl_tbits = 0;
    uint64_t sl_entries = 0;
    const void* ext_index = nullptr;
    uint32_t ext_slices = 0, ext_tbits = 0, ext_chunk_slots = 0, ext_chain_off = 0;
    // sharded upload (multi-GPU): store 0 arrives as nshards pieces that other ranks uploaded and xGMI carried here
    const uint64_t* asm_pieces = nullptr;
    uint64_t asm_slot_words = 0;
    uint32_t asm_n = 0, asm_parts = 1;
    bool ex_on = false;
    uint32_t ex_E = 0, ex_W = 0;
    DevBuf d_end_a, d_end_b, d_dpcnt;
    int live_results = 0;

    po_stats stats = {};

    // layout stage 1 (po_layout_edges)
    bool segments_only = false;  // reads were added by po_add_segment: lengths and names, no sequence
    int ids_paired = -1;         // -1 unknown, 0/1: ids come in (name+"+", name+"-") pairs
    hipEvent_t ev_lay[4] = {};
    DevBuf d_lay_len, d_lay_cnt, d_rflag, d_removed, d_ekey, d_ecnt, d_ewin, d_eoff;
    po_layout_stats lstats = {};
};

struct po_result {
    po_handle* h = nullptr;
    DevBuf d_rows;          // po_row[count] (elem 24) or po_cand[count] (elem 16)
    uint64_t count = 0;
    size_t elem = sizeof(po_row);
    bool kind_edges = false;  // po_edge entries (same size as po_cand)
    void* host = nullptr;     // host copy of the entries: pinned (hipHostMalloc, host_cap bytes) unless host_malloced
    size_t host_cap = 0;
    bool host_malloced = false;  // po_result_from_rows: plain malloc (works without a GPU)
    // rows written by this library's paired-strand emission: every (row, strand mirror) group is the only writer of
    // its twin edge pair -- po_layout_edges needs no dedupe table for them (layout.hip.h, k_layout_winner_adjacent)
    bool unique_twins = false;
    // po_candidates_shard_into: the caller's buffer the candidates go to when they fit
    void* ext_dst = nullptr;
    uint64_t ext_cap = 0;
    bool wrote_ext = false;
    bool compact = false;   // (inside po_overlaps_to_host only) d_rows holds the verified-candidate records of `count` ROWS
};

namespace {

po_status fail(po_handle* h, po_status st, const std::string& msg) {
    if (h) h->err = msg;
    return st;
}

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            return fail((h), e_ == hipErrorOutOfMemory ? PO_ERR_NOMEM : PO_ERR_HIP,             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
        }                                                                                       \
    } while (0)

// PHASM_ALLOC_TRACE=1: every allocation / registration a call makes, with its wall time, on stderr (developer aid:
// what a COLD call -- the first one on a handle -- pays that the steady state does not)
struct AllocTrace {
    const char* what;
    size_t bytes;
    std::chrono::steady_clock::time_point t0;
    bool on;
    AllocTrace(const char* w, size_t b) : what(w), bytes(b), on(getenv("PHASM_ALLOC_TRACE") != nullptr) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~AllocTrace() {
        if (on)
            std::fprintf(stderr, "[alloc] %-16s %10.3f MB  %8.3f ms\n", what, bytes / 1e6,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

po_status ensure(po_handle* h, DevBuf& b, size_t bytes, double scale = 1.0, bool arena_ok = true) {
    if (bytes <= b.cap) return PO_OK;
    AllocTrace tr("hipMalloc", bytes);
    const bool first = b.p == nullptr;
    b.release();
    size_t want = bytes + bytes / 8 + 256;
    if (scale > 1.0) want = (size_t)((double)bytes * scale) + 256;   // (a streamed piece: room for the largest piece)
    if (h && h->st_on) want += 8192;   // (... and for the next call's predicted count plus its slack, small inputs included)
    constexpr size_t ARENA_CHUNK = 64u << 20, ARENA_MAX = 8u << 20;
    // (arena_ok = false: buffers that leave the handle with a result -- rows, candidates, edges -- are freed or kept as
    // spares one by one; carved out of a chunk they would stay behind until the handle dies)
    if (h && first && arena_ok && want <= ARENA_MAX && !getenv("PHASM_NO_ARENA")) {
        // (only a buffer's FIRST allocation: one that has to grow moves out, so a chunk never fills up with dead pieces)
        want = (want + 255) & ~size_t(255);
        if (h->arena_left < want) {
            void* c = nullptr;
            if (hipMalloc(&c, ARENA_CHUNK) == hipSuccess) {
                h->arena_chunks.push_back(c);
                h->arena_cur = static_cast<char*>(c);
                h->arena_left = ARENA_CHUNK;
            } else {
                (void)hipGetLastError();
            }
        }
        if (h->arena_left >= want) {
            b.p = h->arena_cur;
            b.cap = want;
            b.arena = true;
            h->arena_cur += want;
            h->arena_left -= want;
            if (h->poison >= 0 && h->stream) (void)hipMemsetAsync(b.p, h->poison, want, h->stream);
            return PO_OK;
        }
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(h, PO_ERR_NOMEM, "hipMalloc(" + std::to_string(want) + " bytes): " + hipGetErrorString(e));
    }
    b.cap = want;
    // PHASM_POISON: a fresh allocation starts out as garbage of the caller's choosing (hipMalloc usually hands out
    // zeros or whatever an earlier handle of the process left there): a kernel that reads what this call has not
    // written shows up as a parity failure instead of depending on the history of the process
    if (h && h->poison >= 0 && h->stream) (void)hipMemsetAsync(b.p, h->poison, want, h->stream);
    return PO_OK;
}

// a workspace whose size follows the candidate count of the a-side range at hand
po_status ensure_piece(po_handle* h, DevBuf& b, size_t bytes) { return ensure(h, b, bytes, h->ws_scale); }

po_status ensure_host(po_handle* h, HostBuf& b, size_t bytes) {
    if (bytes <= b.cap) return PO_OK;
    AllocTrace tr("hipHostMalloc", bytes);
    b.release();
    const size_t want = bytes + bytes / 16 + 4096;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(h, PO_ERR_NOMEM, "hipHostMalloc(" + std::to_string(want) + " bytes): " + hipGetErrorString(e));
    }
    b.cap = want;
    pin_note(b.p, want, 1u);
    return PO_OK;
}

// the packed host store is about to change (it may move, and the old block is unregistered and freed): no copy out of
// it may be in flight
void quiesce_store(po_handle* h) {
    if (!h->dev_ready) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    if (h->rc_stream) (void)hipStreamSynchronize(h->rc_stream);
    if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
}

// Pinned result pool, sized WHILE THE READS ARE ADDED.  A fresh 170 MB page-locked array costs 8 ms (hipHostMalloc maps
// and pins every page), more than the whole steady-state call: a cold call that allocates it -- let alone grows it
// twice, as the first call on a handle used to -- pays several times the step.  How many rows a call will return is
// not known before the scan, so the pool is a guess made from what IS known, the bases added so far: one row per
// 160 oriented bases (BASELINE config 2 gives one per 214, config 3 one per 320, config 5 one per 436), re-made
// whenever the read set has grown by half, capped at 4 GB.  A call that needs more grows the array as before.
po_status init_device(po_handle* h);

inline bool home_enabled() {
    const char* e = getenv("PHASM_HOME");
    return !(e && atoi(e) == 0);
}

void result_pool_grow(po_handle* h, uint64_t bases = 0) {
    if (!bases) bases = h->total_bases;   // (bases > 0: an estimate made before the reads are in -- po_add_fasta's warm-up thread)
    h->pool_bases = bases + bases / 2;   // next look when the set has grown by half
    if (getenv("PHASM_NO_POOL")) return;
    // a read set this size is headed for the GPU: bring the device up now (runtime start, stream, events: 130 ms in a
    // fresh process) rather than inside the first call.  No GPU: the call itself will say so.
    if (!h->dev_ready && init_device(h) != PO_OK) h->err.clear();
    const uint64_t want = std::min<uint64_t>(h->pool_bases / 160 * sizeof(po_row), 4ull << 30);
    if (h->spare_host.cap >= want || h->live_results) return;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        h->pool_bases = ~0ull;   // no GPU here: never ask again
        return;
    }
    AllocTrace tr("result pool", want);
    h->spare_host.release();
    if (hipHostMalloc(&h->spare_host.p, want, hipHostMallocPortable) == hipSuccess) {
        h->spare_host.cap = want;
        pin_note(h->spare_host.p, want, 1u);
        // the landing block of the compact records (a third of the row bytes), sized with it -- not inside the first call
        if (home_enabled() && h->home_stage.cap < want / 3) (void)ensure_host(h, h->home_stage, want / 3);
        // and the first LARGE copy in each direction (the copy engines' queues are made by it: the first piece of the first
        // call of a process took 7.5 ms instead of 0.5) moves 8 MB of the fresh pool to the device and back
        if (h->dev_ready && h->up_stream && h->copy_stream && want >= (16u << 20) && !getenv("PHASM_NO_WARM")) {
            void* w = nullptr;
            const size_t nb = 8u << 20;
            if (hipMalloc(&w, nb) == hipSuccess) {
                (void)hipMemcpyAsync(w, h->spare_host.p, nb, hipMemcpyHostToDevice, h->up_stream);
                (void)hipStreamSynchronize(h->up_stream);
                (void)hipMemcpyAsync(static_cast<char*>(h->spare_host.p) + nb, w, nb, hipMemcpyDeviceToHost, h->copy_stream);
                (void)hipStreamSynchronize(h->copy_stream);
                (void)hipFree(w);
            }
            (void)hipGetLastError();
        }
    } else {
        (void)hipGetLastError();
        h->spare_host.p = nullptr;
        h->pool_bases = ~0ull;
    }
}

#define PO_TRY(expr)                    \
    do {                                \
        po_status s_ = (expr);          \
        if (s_ != PO_OK) return s_;     \
    } while (0)

// Streams, events and the small page-locked landing zone of a handle are kept for the NEXT handle of the process instead
// of being destroyed with this one: a test process (and a server) creates and destroys handles by the hundred, each with
// four streams and ~70 events -- churn inside the HIP runtime (its completion handlers run on threads of their own) that
// buys nothing, and ~1 ms of a fresh handle's first call.  A kit is idle when it is put back (every stream synchronised).
struct DevKit {
    int device = -1;
    hipStream_t stream = nullptr, copy_stream = nullptr, up_stream = nullptr, rc_stream = nullptr, scan_stream = nullptr;
    hipEvent_t ev_s1[2] = {}, ev_idx = nullptr;
    hipEvent_t ev_sets[2][EV_N] = {};
    hipEvent_t ev_up0 = nullptr, ev_up1 = nullptr, ev_meta = nullptr, ev_first = nullptr;
    hipEvent_t ev_piece[PO_MAX_PIECES] = {}, ev_rc[PO_MAX_PIECES] = {}, ev_lay[4] = {};

    uint64_t* pinned = nullptr;
    uint64_t* pinned_dev = nullptr;
};
std::mutex g_kit_mu;
std::vector<DevKit> g_kits;
constexpr size_t KIT_POOL_MAX = 8;

bool kit_take(po_handle* h) {
    std::lock_guard<std::mutex> lock(g_kit_mu);
    for (size_t i = 0; i < g_kits.size(); ++i) {
        if (g_kits[i].device != h->device) continue;
        const DevKit k = g_kits[i];
        g_kits.erase(g_kits.begin() + (long)i);
        h->stream = k.stream;
        h->copy_stream = k.copy_stream;
        h->up_stream = k.up_stream;
        h->rc_stream = k.rc_stream;
        h->scan_stream = k.scan_stream;
        h->ev_s1[0] = k.ev_s1[0];
        h->ev_s1[1] = k.ev_s1[1];
        h->ev_idx = k.ev_idx;
        std::memcpy(h->ev_sets, k.ev_sets, sizeof(k.ev_sets));
        h->ev_up0 = k.ev_up0;
        h->ev_up1 = k.ev_up1;
        h->ev_meta = k.ev_meta;
        h->ev_first = k.ev_first;
        std::memcpy(h->ev_piece, k.ev_piece, sizeof(k.ev_piece));
        std::memcpy(h->ev_rc, k.ev_rc, sizeof(k.ev_rc));
        std::memcpy(h->ev_lay, k.ev_lay, sizeof(k.ev_lay));
        h->pinned = k.pinned;
        h->pinned_dev = k.pinned_dev;
        return true;
    }
    return false;
}

// true: the handle's streams / events / landing zone went back to the pool (the caller must not destroy them)
bool kit_give(po_handle* h) {
    if (getenv("PHASM_NO_KIT_POOL") || !h->stream || !h->copy_stream || !h->up_stream || !h->rc_stream || !h->pinned) return false;
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->copy_stream) != hipSuccess ||
        hipStreamSynchronize(h->up_stream) != hipSuccess || hipStreamSynchronize(h->rc_stream) != hipSuccess ||
        (h->scan_stream && hipStreamSynchronize(h->scan_stream) != hipSuccess)) {
        (void)hipGetLastError();
        return false;
    }
    DevKit k;
    k.device = h->device;
    k.stream = h->stream;
    k.copy_stream = h->copy_stream;
    k.up_stream = h->up_stream;
    k.rc_stream = h->rc_stream;
    k.scan_stream = h->scan_stream;
    k.ev_s1[0] = h->ev_s1[0];
    k.ev_s1[1] = h->ev_s1[1];
    k.ev_idx = h->ev_idx;
    std::memcpy(k.ev_sets, h->ev_sets, sizeof(k.ev_sets));
    k.ev_up0 = h->ev_up0;
    k.ev_up1 = h->ev_up1;
    k.ev_meta = h->ev_meta;
    k.ev_first = h->ev_first;
    std::memcpy(k.ev_piece, h->ev_piece, sizeof(k.ev_piece));
    std::memcpy(k.ev_rc, h->ev_rc, sizeof(k.ev_rc));
    std::memcpy(k.ev_lay, h->ev_lay, sizeof(k.ev_lay));
    k.pinned = h->pinned;
    k.pinned_dev = h->pinned_dev;
    std::lock_guard<std::mutex> lock(g_kit_mu);
    if (g_kits.size() >= KIT_POOL_MAX) return false;
    g_kits.push_back(k);
    return true;
}


// CPUs this process may use: the hardware's count, cut to the container's CPU quota where there is one (cgroup v2 cpu.max
// "quota period", v1 cpu.cfs_quota_us / _period_us) -- hardware_concurrency() says 256 on a box whose process may use 16, and
// threads beyond the share only get the whole process throttled
unsigned cpu_share() {
    static const unsigned share = [] {
        unsigned n = std::thread::hardware_concurrency();
        n = n ? n : 4u;
        long long q = -1, per = 100000;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char buf[64] = {0};
            if (std::fscanf(f, "%63s %lld", buf, &per) >= 1 && std::strcmp(buf, "max") != 0) q = atoll(buf);
            std::fclose(f);
        } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(g, "%lld", &q) != 1) q = -1;
            std::fclose(g);
            if (FILE* g2 = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g2, "%lld", &per) != 1) per = 100000;
                std::fclose(g2);
            }
        }
        if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / per));
        return n;
    }();
    return share;
}

// ---- rows home in compact form -------------------------------------------------------------------------------------------
// po_overlaps_to_host used to bring every 24-byte row across PCIe: 168 MB at BASELINE config 2, and the step ended when that
// copy ended.  Half of those rows are strand mirrors of the other half, and every exact row is a function of {a, p, b, type}
// and the two read lengths -- the verified-candidate record the multi-GPU exchange already uses (po_cand, 16 bytes), or ONE
// 64-bit word where the read set allows it (po::pack_record: 23 MB at config 2).  So the device hands out one record per
// strand-mirror pair (k_tail_cands), the records cross PCIe, and host threads write the rows into the page-locked result
// array while later pieces are still on the device: the same array, byte for byte, that k_tail / k_emit write on the device
// (rows per record in write_rows' order: A row, [its mirror], B row, [its mirror]; records in candidate order).  SURVEY.md
// section 8c has the mirror rules; the row fields are those of src/overlapper.cpp:77-82,104-110.
//
// One pool per process, threads started at the first use (three quarters of the process's CPU share, 12 at most;
// PHASM_HOME_THREADS).  Helpers sleep between pieces and spin inside one (a piece is 50-400 us of work).  Thread 0 takes the
// pieces in order: it polls the page-locked word a one-thread kernel writes behind the piece's device->host copy, then all
// threads count the rows of the piece's chunks (CHUNK records each), thread 0 turns the counts into offsets -- and checks the
// total against what the device counted --, and all threads write.
namespace home {

constexpr uint32_t CHUNK = 4096;

struct Job {
    const void* rec = nullptr;       // page-locked staging memory: po::Cand records, or (sh_b != 0) packed 8-byte records
    uint32_t sh_b = 0, sh_p = 0;     // po::pack_record's shifts
    uint64_t n_rec = 0;
    po_row* out = nullptr;           // where this piece's rows start in the result array
    uint64_t n_rows = 0;             // what the device counted for the piece
    // the piece's records are home when *flag == want: a one-thread kernel queued behind their device->host copy on the
    // copy stream writes `want` into page-locked memory (polling a word costs the caller's launches nothing; polling
    // hipEventQuery takes the runtime's locks on every call).  flag == nullptr: the records are there already.
    const volatile uint32_t* flag = nullptr;
    uint32_t want = 0;
    // A piece may arrive in several copies (the LAST piece of a step does: its rows are what the call ends with, and its
    // first part is expanded while the rest is still crossing PCIe).  A part knows its rows only when the pool has counted
    // them: cont = this part's rows start where the previous part's ended; n_rows = 0 for every part but the last, which
    // carries the piece's total for the check.
    bool cont = false;
    bool more = false;               // further parts of the same piece follow
};

struct Pool {
    std::vector<std::thread> thr;
    std::mutex mu;                   // queue, job hand-over, sleep / wake
    std::condition_variable cv_queue;   // thread 0: a piece was submitted
    std::condition_variable cv_job;     // helpers: a piece was published
    std::mutex call_mu;              // one call at a time uses the pool
    std::deque<Job> queue;
    std::atomic<uint64_t> submitted{0}, finished{0};
    std::atomic<int> error{0};       // 1: counts disagree, 2: a record names an unknown read
    // the call's read set
    const uint32_t* len = nullptr;
    uint32_t n_reads = 0, paired = 0, bits = 2;
    // byte counters of po_stats over the call's records (k_emit / k_tail sum them on the device for rows emitted there)
    std::atomic<uint64_t> sum_l{0}, sum_b{0}, sum_e{0};
    // the piece being expanded
    Job cur;
    po_row* part_next = nullptr;     // where the next part of the current piece starts
    uint64_t part_rows = 0;          // rows of the piece's parts so far
    uint64_t job_seq = 0;            // (under mu) pieces published so far
    uint32_t n_chunks = 0;
    std::vector<uint64_t> chunk_off;
    std::atomic<uint32_t> next_count{0}, done_count{0}, next_write{0}, done_write{0};
    std::atomic<int> phase{0};       // 0: none, 1: count, 2: write
    std::atomic<int> inside{0};      // helper threads inside the current piece's loops
    // PHASM_STREAM_TRACE: per piece, microseconds since the call began -- records home / rows written
    std::chrono::steady_clock::time_point t0;
    std::vector<std::array<double, 3>> trace;   // {ready, done, rows}
    bool tracing = false;
};

Pool* g_pool = nullptr;
std::once_flag g_pool_once;

inline void cpu_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// Spin until pred() holds.  The waits of a piece are microseconds long when every thread has a core; on a host whose cores
// are busy with other people's work a thread can spin through the time slice of the very thread it waits for (a box like
// that gave 5.2-5.5 ms per step instead of 4.7), so after a few thousand pauses the waiter offers its core.
std::atomic<uint32_t> g_spin_limit{4096};   // pauses before a waiter starts yielding (PHASM_HOME_SPIN; 0 = never yield)
template <class Pred>
inline void spin_until(Pred pred) {
    const uint32_t limit = g_spin_limit.load(std::memory_order_relaxed);
    for (uint32_t n = 0; !pred(); ++n) {
        if (limit == 0 || n < limit) cpu_pause();
        else std::this_thread::yield();
    }
}

inline po::Cand load_rec(const Job& j, uint64_t i) {
    if (!j.sh_b) return static_cast<const po::Cand*>(j.rec)[i];
    return po::unpack_record(static_cast<const uint64_t*>(j.rec)[i], j.sh_b, j.sh_p);
}

inline uint32_t rows_of_rec(const po::Cand& c, uint32_t paired) {
    if (!paired) return (c.type & 1u) + ((c.type >> 1) & 1u);
    return ((c.type & 1u) ? (c.a == (c.b ^ 1u) ? 1u : 2u) : 0u) + ((c.type & 2u) ? 2u : 0u);
}

inline void put_row(uint64_t*& o, uint32_t a, uint32_t b, uint32_t astart, uint32_t aend, uint32_t bend) {
    // (a row is three 8-byte words; non-temporal stores: the array is written once and read by the caller much later)
    __builtin_nontemporal_store((uint64_t)a | ((uint64_t)b << 32), o);
    __builtin_nontemporal_store((uint64_t)astart | ((uint64_t)aend << 32), o + 1);
    __builtin_nontemporal_store((uint64_t)bend << 32, o + 2);
    o += 3;
}

// rows of records [lo, hi) -> out; returns false when a record names a read the handle does not hold
// (the byte counters of po_stats -- write_rows' counters, kernels.hip.h -- are taken here, where the two lengths are at hand:
// the counting pass in front of this one then needs nothing but the records)
bool expand_records(Pool& P, uint64_t lo, uint64_t hi, po_row* out) {
    uint64_t* o = reinterpret_cast<uint64_t*>(out);
    const Job& job = P.cur;
    const uint32_t* len = P.len;
    const uint32_t paired = P.paired, n_reads = P.n_reads, bits = P.bits;
    auto pb = [bits](uint32_t l) -> uint64_t { return bits == 8u ? l : (l >> 2) + ((l & 3u) != 0u); };
    uint64_t sl = 0, sb = 0, se = 0;
    for (uint64_t i = lo; i < hi; ++i) {
        const po::Cand c = load_rec(job, i);
        if (c.a >= n_reads || c.b >= n_reads) return false;
        const uint32_t la = len[c.a], lb = len[c.b];
        se += 2 * pb((c.type & 1u) ? la - c.p : lb);
        if (c.type & 1u) {
            const uint32_t l = la - c.p;
            const bool twin = paired && c.a != (c.b ^ 1u);
            put_row(o, c.a, c.b, c.p, la, l);
            if (twin) put_row(o, c.b ^ 1u, c.a ^ 1u, lb - l, lb, l);
            const uint64_t k = twin ? 2 : 1;
            sl += k * l;
            sb += k * 2 * pb(l);
        }
        if (c.type & 2u) {
            put_row(o, c.a, c.b, c.p, c.p + lb, lb);
            if (paired) put_row(o, c.a ^ 1u, c.b ^ 1u, la - c.p - lb, la - c.p, lb);
            const uint64_t k = paired ? 2 : 1;
            sl += k * lb;
            sb += k * 2 * pb(lb);
        }
    }
    P.sum_l.fetch_add(sl, std::memory_order_relaxed);
    P.sum_b.fetch_add(sb, std::memory_order_relaxed);
    P.sum_e.fetch_add(se, std::memory_order_relaxed);
    return true;
}

// the two passes over the current piece; `lead` (thread 0) turns the counts into offsets between them.  A piece is
// 50-400 us of work for the pool: inside it the threads spin, between pieces they sleep.
void run_phases(Pool& P, bool lead) {
    const uint32_t nc = P.n_chunks;
    for (;;) {   // count
        const uint32_t c = P.next_count.fetch_add(1, std::memory_order_relaxed);
        if (c >= nc) break;
        const uint64_t lo = (uint64_t)c * CHUNK, hi = std::min<uint64_t>(P.cur.n_rec, lo + CHUNK);
        uint64_t n = 0;
        const uint32_t paired = P.paired;
        for (uint64_t i = lo; i < hi; ++i) n += rows_of_rec(load_rec(P.cur, i), paired);
        P.chunk_off[c + 1] = n;
        P.done_count.fetch_add(1, std::memory_order_release);
    }
    if (lead) {
        spin_until([&] { return P.done_count.load(std::memory_order_acquire) >= nc; });
        P.chunk_off[0] = 0;
        for (uint32_t c = 0; c < nc; ++c) P.chunk_off[c + 1] += P.chunk_off[c];
        P.part_rows += P.chunk_off[nc];
        if (!P.cur.more && P.part_rows != P.cur.n_rows) P.error.store(1);   // (the device counted differently: nothing is written)
        P.part_next = P.cur.out + P.chunk_off[nc];
        P.phase.store(P.error.load() ? 0 : 2, std::memory_order_release);
    } else {
        spin_until([&] { return P.phase.load(std::memory_order_acquire) != 1; });
    }
    if (P.phase.load(std::memory_order_acquire) != 2) return;
    for (;;) {   // write
        const uint32_t c = P.next_write.fetch_add(1, std::memory_order_relaxed);
        if (c >= nc) break;
        const uint64_t lo = (uint64_t)c * CHUNK, hi = std::min<uint64_t>(P.cur.n_rec, lo + CHUNK);
        if (!expand_records(P, lo, hi, P.cur.out + P.chunk_off[c])) P.error.store(2);
#if defined(__x86_64__)
        __builtin_ia32_sfence();   // (non-temporal stores are ordered by sfence only: before the chunk counts as written)
#endif
        P.done_write.fetch_add(1, std::memory_order_release);
    }
}

void helper(Pool* Pp) {
    Pool& P = *Pp;
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(P.mu);
            P.cv_job.wait(lock, [&] { return P.job_seq != seen; });
            seen = P.job_seq;
        }
        P.inside.fetch_add(1, std::memory_order_acq_rel);
        if (P.phase.load(std::memory_order_acquire) != 0) run_phases(P, false);
        P.inside.fetch_sub(1, std::memory_order_release);
    }
}

void leader(Pool* Pp) {
    Pool& P = *Pp;
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lock(P.mu);
            P.cv_queue.wait(lock, [&] { return !P.queue.empty(); });
            j = P.queue.front();
            P.queue.pop_front();
        }
        if (j.flag)   // the records are home when the word behind their copy has arrived
            spin_until([&] { return *j.flag == j.want; });
        std::atomic_thread_fence(std::memory_order_acquire);
        const double t_ready = P.tracing ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - P.t0).count() : 0.0;
        if (P.error.load() == 0 && j.n_rec) {
            // (no helper is inside the previous piece's loops any more: the counters may be reset)
            spin_until([&] { return P.inside.load(std::memory_order_acquire) == 0; });
            if (j.cont) j.out = P.part_next;
            else P.part_rows = 0;
            P.cur = j;
            P.n_chunks = (uint32_t)((j.n_rec + CHUNK - 1) / CHUNK);
            if (P.chunk_off.size() < (size_t)P.n_chunks + 1) P.chunk_off.resize((size_t)P.n_chunks + 1);
            P.next_count.store(0);
            P.done_count.store(0);
            P.next_write.store(0);
            P.done_write.store(0);
            P.phase.store(1, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lock(P.mu);
                ++P.job_seq;
            }
            P.cv_job.notify_all();
            run_phases(P, true);
            if (P.phase.load() == 2)
                spin_until([&] { return P.done_write.load(std::memory_order_acquire) >= P.n_chunks; });
            P.phase.store(0, std::memory_order_release);
            std::atomic_thread_fence(std::memory_order_seq_cst);
        }
        if (P.tracing)
            P.trace.push_back({t_ready, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - P.t0).count(), (double)j.n_rows});
        P.finished.fetch_add(1, std::memory_order_release);
    }
}

Pool* pool() {
    std::call_once(g_pool_once, [] {
        Pool* P = new Pool();
        // (a GPU box gives one GPU's process a share of the host's cores: the pool stays well inside it -- the caller's
        // thread and the HIP runtime's own threads need cores too, and threads beyond the share would stall everyone)
        unsigned n = cpu_share();
        // three quarters of what the process may use, 12 at most (measured at config 2 on a 16-CPU share: 4 threads 5.57 ms per
        // step, 7: 5.09, 10: 4.68, 12: 4.64; 14 leave the caller's thread and the runtime's threads without a CPU: 5.75)
        n = std::max(1u, std::min(n * 3 / 4, 12u));
        if (const char* e = getenv("PHASM_HOME_THREADS")) n = (unsigned)std::max(1, std::min(64, atoi(e)));
        try {
            P->thr.emplace_back(leader, P);
            P->thr.back().detach();   // (they sleep until the process ends)
            for (unsigned i = 1; i < n; ++i) {
                P->thr.emplace_back(helper, P);
                P->thr.back().detach();
            }
        } catch (const std::system_error&) {
        }
        if (P->thr.empty()) {
            delete P;
            P = nullptr;
        }
        g_pool = P;
    });
    return g_pool;
}

void begin(Pool* P, const uint32_t* len, uint32_t n_reads, bool tracing) {
    if (const char* e = getenv("PHASM_HOME_SPIN")) g_spin_limit.store((uint32_t)std::max(0, atoi(e)));
    else g_spin_limit.store(4096);
    P->len = len;
    P->n_reads = n_reads;
    P->paired = 0;
    P->error.store(0);
    P->submitted.store(0);
    P->finished.store(0);
    P->sum_l.store(0);
    P->sum_b.store(0);
    P->sum_e.store(0);
    P->tracing = tracing;
    P->trace.clear();
    P->t0 = std::chrono::steady_clock::now();
}

void submit(Pool* P, const Job& j) {
    {
        std::lock_guard<std::mutex> lock(P->mu);
        P->queue.push_back(j);
    }
    P->submitted.fetch_add(1, std::memory_order_release);
    P->cv_queue.notify_one();
}

void wait_all(Pool* P) {
    // (the caller has nothing else to do: it spins -- the last piece's rows are what the call is waiting for)
    spin_until([&] { return P->finished.load(std::memory_order_acquire) >= P->submitted.load(std::memory_order_acquire); });
}

}  // namespace home

po_status init_device(po_handle* h) {
    if (h->dev_ready) {
        HIP_TRY(h, hipSetDevice(h->device));
        return PO_OK;
    }
    AllocTrace tr("init_device", 0);
    const auto ti0 = std::chrono::steady_clock::now();
    auto imark = [&](const char* what) {
        if (tr.on) std::fprintf(stderr, "[init] %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count());
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    imark("runtime up (hipGetDeviceCount)");
    if (e != hipSuccess || n == 0)
        return fail(h, PO_ERR_HIP, "no HIP device available (libphasm_overlap has no CPU fallback)");
    if (h->device < 0 || h->device >= n) return fail(h, PO_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->lds_max = prop.sharedMemPerBlock;
    if (const char* e = getenv("PHASM_POISON")) h->poison = (int)(strtol(e, nullptr, 0) & 0xFF);
    if (!getenv("PHASM_NO_KIT_POOL") && kit_take(h)) {
        h->ev = h->ev_sets[0];
        std::memset(h->pinned, 0, 1024);
        hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->stream, reinterpret_cast<uint32_t*>(h->pinned_dev + 63), (uint64_t)1, 0u);
        (void)hipGetLastError();
        h->dev_ready = true;
        return PO_OK;
    }
    imark("device selected, properties read");
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    imark("first stream made");
    for (int i = 0; i < 2 * EV_N; ++i) HIP_TRY(h, hipEventCreate(&h->ev_sets[i / EV_N][i % EV_N]));
    HIP_TRY(h, hipEventCreate(&h->ev_up0));
    HIP_TRY(h, hipEventCreate(&h->ev_up1));
    imark("first events made");
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->pinned), 1024, hipHostMallocDefault));   // 128 slots
    pin_note(h->pinned, 1024, 1u);
    HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pinned_dev), h->pinned, 0));
    imark("landing zone made");
    if (const char* e = getenv("PHASM_POISON")) h->poison = (int)(strtol(e, nullptr, 0) & 0xFF);
    // the streams and events of the host-to-host call: created here (a few tenths of a millisecond each), not in its first call
    HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    imark("copy stream made");
    HIP_TRY(h, hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->rc_stream, hipStreamNonBlocking));
    // (four streams = the runtime's four hardware queues, one each.  A fifth stream shares a queue with one of these, and
    // whatever it launches waits behind everything queued there -- rc_stream's kernels are all queued up front behind the
    // pieces' arrival events: profiles/r04_two_stream.txt.  The opt-in two-stream pieces make theirs when first asked for.)
    imark("other streams made");
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_s1[0], hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_s1[1], hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_idx, hipEventDisableTiming));
    for (int k = 0; k < PO_MAX_PIECES; ++k) {
        HIP_TRY(h, hipEventCreate(&h->ev_piece[k]));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_rc[k], hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_first, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_meta, hipEventDisableTiming));
    imark("streams and events made");
    // the library's code object is loaded by the first launch out of it (15 ms in a fresh process): here, not in a call
    hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->stream, reinterpret_cast<uint32_t*>(h->pinned_dev + 63), (uint64_t)1, 0u);
    (void)hipGetLastError();
    imark("first launch queued (code object loaded)");
    // ... and so are the hardware queues behind the three other streams and the runtime's copy paths: a stream's queue is made
    // by the first command it gets, the host->device / device->host machinery by the first copy in each direction -- 15 ms
    // of the first call of a process (`8 pieces queued at 15.056 ms`, tools/cold_probe.py) when left to the call.  One
    // word each way on the streams that will carry the copies, one launch on every stream; the device comes up while the
    // reads are still being added (result_pool_grow, po_add_fasta's warm-up thread), so none of this is waited for there.
    if (!getenv("PHASM_NO_WARM")) {
        void* w = nullptr;
        if (hipMalloc(&w, 256) == hipSuccess) {
            uint32_t* wd = static_cast<uint32_t*>(w);
            (void)hipMemcpyAsync(wd, h->pinned + 62, 8, hipMemcpyHostToDevice, h->up_stream);
            hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->up_stream, wd + 8, (uint64_t)1, 0u);
            (void)hipMemcpyAsync(wd + 16, h->pinned + 62, 8, hipMemcpyHostToDevice, h->stream);
            hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->rc_stream, wd + 24, (uint64_t)1, 0u);
            if (h->scan_stream) hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->scan_stream, wd + 40, (uint64_t)1, 0u);
            hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->copy_stream, wd + 32, (uint64_t)1, 0u);
            (void)hipMemcpyAsync(h->pinned + 61, wd + 32, 8, hipMemcpyDeviceToHost, h->copy_stream);
            (void)hipMemcpyAsync(h->pinned + 60, wd + 16, 8, hipMemcpyDeviceToHost, h->stream);
            (void)hipStreamSynchronize(h->up_stream);
            (void)hipStreamSynchronize(h->rc_stream);
            if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
            (void)hipStreamSynchronize(h->copy_stream);
            (void)hipStreamSynchronize(h->stream);
            (void)hipFree(w);
        }
        (void)hipGetLastError();
        imark("queues and copy paths warmed");
        // every kernel of a 2-bit call is looked up in the code object now (the runtime resolves a kernel at its first launch:
        // ~35 of them at 0.1-0.3 ms each inside the first call otherwise)
        static const void* const warm_kernels[] = {
            (const void*)po::k_abs_woff, (const void*)po::k_build_tiles, (const void*)po::k_revcomp_store, (const void*)po::k_scatter_first,
            (const void*)po::k_paired_check, (const void*)po::k_call_init, (const void*)po::k_call_reset, (const void*)po::k_table_insert,
            (const void*)po::k_chain_fill, (const void*)po::k_chain_sort_short<uint32_t>, (const void*)po::k_chain_sort_long<uint32_t>,
            (const void*)po::k_table_finalize, (const void*)po::k_ps_reduce<uint32_t>, (const void*)po::k_ps_spine, (const void*)po::k_ps_down<uint32_t>,
            (const void*)po::k_ps_reduce<uint8_t>, (const void*)po::k_ps_down<uint8_t>, (const void*)po::k_ps_chain<uint32_t>,
            (const void*)po::k_ps_chain<uint8_t>, (const void*)po::k_ps_small,
            (const void*)po::k_scan_probe<2, true, true>, (const void*)po::k_scan_probe<2, true, false>, (const void*)po::k_scan_fixup<2, true>,
            (const void*)po::k_scan_fixup<2, false>, (const void*)po::k_scan_fill<2, true>, (const void*)po::k_scan_fill<2, false>,
            (const void*)po::k_read_label, (const void*)po::k_read_sort, (const void*)po::k_read_invert, (const void*)po::k_defer_split,
            (const void*)po::k_verify_a<2, false, true, true>, (const void*)po::k_verify_a<2, false, true, false>,
            (const void*)po::k_verify_a<2, true, true, false>, (const void*)po::k_select_local, (const void*)po::k_tile_rows<true>,
            (const void*)po::k_tile_rows<false>, (const void*)po::k_tail, (const void*)po::k_tail_cands, (const void*)po::k_select,
            (const void*)po::k_emit, (const void*)po::k_verify_flat, (const void*)po::k_deferred_rowcnt, (const void*)po::k_emit_cands,
            (const void*)po::k_wide_insert<2, 1>, (const void*)po::k_wide_insert<2, 16>, (const void*)po::k_wide_chain_fill<2, 1>,
            (const void*)po::k_wide_chain_fill<2, 16>, (const void*)po::k_wide_finalize<2, 1>, (const void*)po::k_wide_finalize<2, 16>,
            (const void*)po::k_chain_sort_short<uint64_t>, (const void*)po::k_chain_sort_long<uint64_t>,
            (const void*)po::k_wide_scan<2, false, true, 1>, (const void*)po::k_wide_scan<2, true, true, 1>,
            (const void*)po::k_wide_scan<2, false, false, 16>, (const void*)po::k_wide_scan<2, true, false, 16>};
        for (const void* k : warm_kernels) {
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, k);
        }
        (void)hipGetLastError();
        imark("kernels looked up");
    }
    h->dev_ready = true;
    return PO_OK;
}

// 2-bit code per byte; 0x80 = not one of upper-case A/C/G/T
struct BaseLut {
    uint8_t v[256];
    BaseLut() {
        std::memset(v, 0x80, sizeof(v));
        v[(unsigned char)'A'] = 0;
        v[(unsigned char)'C'] = 1;
        v[(unsigned char)'G'] = 2;
        v[(unsigned char)'T'] = 3;
    }
};
const BaseLut g_lut;

// Append one read to the packed store.  Layout: every read starts on a 16-byte boundary and is
// followed by at least one zero guard word (kernels read one word past the last data word).
// bits == 2: bytes other than upper-case A/C/G/T become exception records (code 0 in the packed
// words); returns false (and appends nothing) when they are too dense for that (> n/64 + 16), in
// which case the caller moves the whole handle to 8 bits per base.
bool append_packed(po_handle* h, const unsigned char* s, size_t n, int bits) {
    const size_t per = 64 / bits;
    WordStore& store = h->words[h->woff.size() & 1];  // (woff has one entry per read appended so far)
    const size_t old_size = store.size();
    const size_t off = (old_size + 1) & ~size_t(1);
    const size_t nw = (n + per - 1) / per;
    store.resize(off + nw + 1, 0);
    uint64_t* w = store.data() + off;
    if (bits == 2) {
        const uint8_t* lut = g_lut.v;
        uint8_t bad = 0;
        const size_t full = n / 32;
        for (size_t k = 0; k < full; ++k) {
            const unsigned char* q = s + k * 32;
            uint64_t acc = 0;
            for (int j = 0; j < 32; ++j) {
                const uint8_t c = lut[q[j]];
                bad |= c;
                acc |= (uint64_t)(c & 3) << (2 * j);
            }
            w[k] = acc;
        }
        if (full * 32 < n) {
            uint64_t acc = 0;
            for (size_t i = full * 32; i < n; ++i) {
                const uint8_t c = lut[s[i]];
                bad |= c;
                acc |= (uint64_t)(c & 3) << (2 * (i & 31));
            }
            w[full] = acc;
        }
        if (bad & 0x80) {
            size_t cnt = 0;
            for (size_t i = 0; i < n; ++i) cnt += lut[s[i]] >> 7;
            if (cnt > n / 64 + 16) {
                store.resize(old_size);
                return false;
            }
            for (size_t i = 0; i < n; ++i) {
                if (lut[s[i]] & 0x80) {
                    h->exc_pos.push_back((uint32_t)i);
                    h->exc_byte.push_back(s[i]);
                }
            }
        }
        h->exc_off.push_back((uint32_t)h->exc_pos.size());
    } else {
        for (size_t i = 0; i < n; ++i) w[i >> 3] |= (uint64_t)s[i] << ((i & 7) * 8);
    }
    h->woff.push_back(off);
    return true;
}

// The bytes of read r as they were added (2-bit mode: codes + exception records).
void materialize(const po_handle* h, size_t r, const WordStore* stores, const uint64_t* woff, std::vector<unsigned char>& out) {
    const uint32_t n = h->len[r];
    out.resize(n);
    const uint64_t* w = stores[r & 1].data() + woff[r];
    for (uint32_t i = 0; i < n; ++i) out[i] = "ACGT"[(w[i >> 5] >> ((i & 31) * 2)) & 3];
    for (uint32_t e = h->exc_off[r]; e < h->exc_off[r + 1]; ++e) out[h->exc_pos[e]] = h->exc_byte[e];
}

// Non-ACGT bytes too dense for exception records: re-encode everything held so far at 8 bits per
// base (lossless).
void widen_to_bytes(po_handle* h) {
    WordStore old_words[2];
    std::vector<uint64_t> old_off;
    old_words[0].swap(h->words[0]);
    old_words[1].swap(h->words[1]);
    old_off.swap(h->woff);
    std::vector<unsigned char> tmp;
    for (size_t r = 0; r < h->len.size(); ++r) {
        materialize(h, r, old_words, old_off.data(), tmp);
        append_packed(h, tmp.data(), h->len[r], 8);
    }
    h->bits = 8;
    h->all_pairs_rc = false;
    h->all_pairs_rcx = false;
    h->first_n = 0;
    h->first_words.clear();
    h->lead_n = 0;
    h->lead_words.clear();
    h->exc_off.assign(1, 0);
    h->exc_pos.clear();
    h->exc_byte.clear();
    h->pair_state.clear();
}

// IUPAC-aware complement, identity on everything else (the table of phasm_amd/io/fasta.py)
const unsigned char* comp_table() {
    static unsigned char comp[256];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) comp[i] = (unsigned char)i;
        const char* from = "ACGTURYKMBVDHSWNacgturykmbvdhswn";
        const char* to = "TGCAAYRMKVBHDSWNtgcaayrmkvbhdswn";
        for (int i = 0; from[i]; ++i) comp[(unsigned char)from[i]] = (unsigned char)to[i];
        init = true;
    }
    return comp;
}

// 2-bit mode, after read r (odd index) was appended: if it or its partner r-1 carries exception
// records, decide on the host whether the two are exact reverse complements of each other under a
// complement map that is an involution on the bytes present (the device check only sees 2-bit codes).
void host_pair_check(po_handle* h, size_t r, const unsigned char* s) {
    const size_t pair = r / 2;
    if (h->pair_state.size() <= pair) h->pair_state.resize(pair + 1, 0);
    const bool has_exc = h->exc_off[r - 1] != h->exc_off[r] || h->exc_off[r] != h->exc_off[r + 1];
    if (!has_exc) return;  // state 0: the device compares the codes
    const uint32_t n = h->len[r];
    uint8_t state = 1;
    if (h->len[r - 1] != n) {
        state = 2;
    } else {
        std::vector<unsigned char> prev;
        materialize(h, r - 1, h->words, h->woff.data(), prev);
        const unsigned char* c = comp_table();
        for (uint32_t i = 0; i < n; ++i) {
            const unsigned char x = prev[n - 1 - i], y = s[i];
            if (c[x] != y || c[y] != x) {
                state = 2;
                break;
            }
        }
    }
    h->pair_state[pair] = state;
}

// 2-bit mode, read r (odd index) just appended: are its packed words exactly the reverse complement of read r-1's?
// (word arithmetic of k_paired_check / k_revcomp_store on the host: ~2 us per 15 kb read)
bool packed_is_revcomp(const po_handle* h, size_t r) {
    const uint32_t L = h->len[r];
    if (h->len[r - 1] != L) return false;
    if (h->exc_off[r - 1] != h->exc_off[r] || h->exc_off[r] != h->exc_off[r + 1]) return false;  // exception records
    const uint64_t* R = h->words[0].data() + h->woff[r - 1];
    const uint64_t* S = h->words[1].data() + h->woff[r];
    const uint32_t nw = (L + 31) / 32;
    for (uint32_t w = 0; w < nw; ++w) {
        const int64_t o = (int64_t)L - 32 * (int64_t)w - 32;  // first base of R facing word w of S
        uint64_t x;
        uint32_t valid = 32;
        if (o >= 0) {
            const uint32_t sh = (uint32_t)(o & 31) * 2;
            x = sh ? (R[o >> 5] >> sh) | (R[(o >> 5) + 1] << (64 - sh)) : R[o >> 5];
        } else {
            valid = (uint32_t)(32 + o);
            x = R[0] << ((uint32_t)(-o) * 2);
        }
        uint64_t y = __builtin_bitreverse64(x);
        y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
        y = ~y;
        const uint64_t mask = valid >= 32 ? ~0ull : ((1ull << (valid * 2)) - 1ull);
        if (((y ^ S[w]) & mask) != 0) return false;
    }
    return true;
}

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// first_words covers every read: the first two packed words of read r (a read owns its data words and a zero guard
// word; the word behind an EMPTY read is alignment padding, or -- for the last read of a store -- not part of the host
// store at all: zero on the device either way)
void note_first_words(po_handle* h) {
    const uint32_t n = (uint32_t)h->len.size();
    if (h->first_n == n && h->first_words.size() == 2 * (size_t)n) return;
    if (h->first_n > n || h->first_words.size() != 2 * (size_t)h->first_n) {
        h->first_n = 0;
        h->first_words.clear();
    }
    if (n - h->first_n > 1) h->first_words.reserve(2 * (size_t)n);   // (a bulk fill; one read at a time grows geometrically)
    for (uint32_t r = h->first_n; r < n; ++r) {
        const WordStore& store = h->words[r & 1];
        const uint64_t o = h->woff[r];
        h->first_words.push_back(o < store.size() ? store[o] : 0);
        h->first_words.push_back(o + 1 < store.size() ? store[o + 1] : 0);
    }
    h->first_n = n;
}

constexpr uint32_t LEAD_WORDS = 5;   // windows of 4 word-spaced K-mers at W phases: bases 0 .. 5 W - 2 of a read
void note_lead_words(po_handle* h) {
    const uint32_t n = (uint32_t)h->len.size();
    if (h->lead_n == n && h->lead_words.size() == (size_t)LEAD_WORDS * n) return;
    if (h->lead_n > n || h->lead_words.size() != (size_t)LEAD_WORDS * h->lead_n) {
        h->lead_n = 0;
        h->lead_words.clear();
    }
    h->lead_words.resize((size_t)LEAD_WORDS * n, 0);
    uint64_t* out = h->lead_words.data();
    for (uint32_t r = h->lead_n; r < n; ++r) {
        const WordStore& store = h->words[r & 1];
        const uint64_t o = h->woff[r];
        for (uint32_t k = 0; k < LEAD_WORDS; ++k) out[(size_t)r * LEAD_WORDS + k] = o + k < store.size() ? store[o + k] : 0;
    }
    h->lead_n = n;
}

// reads that can take part in an overlap of m bases or more
uint64_t count_eligible(po_handle* h, uint32_t m) {
    const uint64_t n = h->len.size();
    if (h->elig_n != n || h->elig_m != m) {
        uint64_t c = 0;
        for (uint64_t r = 0; r < n; ++r) c += h->len[r] >= m;
        h->elig_n = n;
        h->elig_m = m;
        h->elig_val = c;
    }
    return h->elig_val;
}

void shard_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint32_t* r_begin, uint32_t* r_end, uint64_t* bases);

// words of host store 0 (the even reads) that belong to shard `shard` of `nshards`: [*begin, *begin + *count)
void store0_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint64_t* begin, uint64_t* count) {
    uint32_t rb = 0, re = 0;
    shard_range(h, shard, nshards, &rb, &re, nullptr);
    const uint32_t n = (uint32_t)h->len.size();
    const uint32_t e0 = (rb + 1u) & ~1u, e1 = (re + 1u) & ~1u;   // first even read of this shard / of the next one
    const uint64_t w0 = e0 < n ? h->woff[e0] : h->words[0].size();
    const uint64_t w1 = e1 < n ? h->woff[e1] : h->words[0].size();
    *begin = w0;
    *count = w1 > w0 ? w1 - w0 : 0;
}

// part `part` of `nparts` of that range (equal chunks of an even number of words, the last one shorter or empty): the
// sharded upload is cut this way so that the host->device copy of one part runs under the all-gather of the part before
void store0_part_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint32_t part, uint32_t nparts, uint64_t* begin, uint64_t* count) {
    uint64_t wb = 0, wc = 0;
    store0_range(h, shard, nshards, &wb, &wc);
    const uint64_t cs = (((wc + nparts - 1) / nparts) + 1) & ~uint64_t(1);
    const uint64_t lo = std::min(wc, (uint64_t)part * cs), hi = std::min(wc, (uint64_t)(part + 1) * cs);
    *begin = wb + lo;
    *count = hi - lo;
}

// Everything of an upload but the packed words themselves: scan tiles counted, device buffers sized, per-read
// offsets / lengths / tile numbers copied, tile records built on the device (they need no read data).
// a page-locked copy of `bytes` bytes at `src` (valid until the next upload starts), or `src` itself when the block is
// too small for it / the source is large
const void* staged(po_handle* h, const void* src, size_t bytes) {
    if (!bytes || bytes > STAGE_MAX || !h->stage_host.p) return src;
    const size_t off = (h->stage_used + 63) & ~size_t(63);
    if (off + bytes > h->stage_host.cap) return src;
    std::memcpy(static_cast<char*>(h->stage_host.p) + off, src, bytes);
    h->stage_used = off + bytes;
    return static_cast<const char*>(h->stage_host.p) + off;
}

po_status upload_meta(po_handle* h, bool* generate_out) {
    const uint32_t n = (uint32_t)h->len.size();
    const size_t per = 64 / h->bits;
    {
        // (every copy of the previous upload has completed: upload() and the streamed step end with a synchronisation)
        size_t need = 0;
        for (const WordStore& w : h->words)
            if (!store_is_pinned(w) && w.size() * 8 <= STAGE_MAX) need += w.size() * 8 + 64;
        if (h->bits == 2) need += ((size_t)n + 1) * 4 + h->exc_pos.size() * 5 + 192;
        need += n / 2 + 64;
        h->stage_used = 0;
        if (need > 256) PO_TRY(ensure_host(h, h->stage_host, need));
    }
    // tiles: 64 words each, never spanning reads.  The host only counts them (first tile of every read); the
    // 32-byte records are written on the device (k_build_tiles) instead of travelling over PCIe (25 MB at config 2).
    // Counted once per state of the read set, together with a page-locked copy of the per-read arrays (an async copy
    // out of a std::vector goes through the runtime's staging buffer, synchronously).
    if (h->meta_n != n || h->meta_bits != h->bits || !h->meta_host.p) {
        h->h_read_tile0.assign((size_t)n + 1, 0);
        h->max_len = 0;
        uint64_t nt = 0;
        for (uint32_t r = 0; r < n; ++r) {
            h->h_read_tile0[r] = (uint32_t)nt;
            h->max_len = std::max(h->max_len, h->len[r]);
            const size_t nw = (h->len[r] + per - 1) / per;
            nt += (nw + po::TILE_WORDS - 1) / po::TILE_WORDS;
            if (nt > 0x7FFFFF00ull / po::WAVE) return fail(h, PO_ERR_CAPACITY, "too many scan tiles");
        }
        h->h_read_tile0[n] = (uint32_t)nt;
        h->n_tiles = (uint32_t)nt;
        PO_TRY(ensure_host(h, h->meta_host, (size_t)n * 16 + 64));
        char* mh = static_cast<char*>(h->meta_host.p);
        if (n) {
            std::memcpy(mh, h->woff.data(), (size_t)n * 8);
            std::memcpy(mh + (size_t)n * 8, h->len.data(), (size_t)n * 4);
        }
        std::memcpy(mh + (size_t)n * 12, h->h_read_tile0.data(), ((size_t)n + 1) * 4);
        h->meta_n = n;
        h->meta_bits = h->bits;
    }
    const char* mh = static_cast<const char*>(h->meta_host.p);

    // Device buffer: [store 0 | store 1 | 72 zero words] (a scan tile may read 64+1 words past a read's start).
    // Only store 0 travels when every odd read is the reverse complement of its even partner: store 1 is then
    // rebuilt on the device, bit for bit what the host packed (k_revcomp_store).
    const uint64_t base1 = (h->words[0].size() + 1) & ~uint64_t(1);
    const uint64_t nwords = base1 + h->words[1].size() + 72;
    const bool generate = h->bits == 2 && n >= 2 && (n % 2) == 0 && h->all_pairs_rcx && !getenv("PHASM_FULL_UPLOAD");
    *generate_out = generate;
    h->base1 = base1;
    h->dev_words = nwords;
    PO_TRY(ensure(h, h->d_words, nwords * 8));
    PO_TRY(ensure(h, h->d_woff, ((size_t)n + 1) * 8));
    PO_TRY(ensure(h, h->d_len, ((size_t)n + 1) * 4));
    PO_TRY(ensure(h, h->d_tiles, ((size_t)h->n_tiles + 1) * sizeof(po::TileRec)));
    PO_TRY(ensure(h, h->d_read_tile0, ((size_t)n + 1) * 4));
    HIP_TRY(h, hipEventRecord(h->ev_up0, h->stream));
    uint64_t* dw = h->d_words.as<uint64_t>();
    // (the guard words between the reads are zeros in the host stores already: only the seam and the tail need clearing)
    HIP_TRY(h, hipMemsetAsync(dw + base1 + h->words[1].size(), 0, 72 * 8, h->stream));
    if (base1 != h->words[0].size()) HIP_TRY(h, hipMemsetAsync(dw + h->words[0].size(), 0, 8, h->stream));
    h->upload_bytes = 0;
    if (n) {
        HIP_TRY(h, hipMemcpyAsync(h->d_woff.p, mh, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_len.p, mh + (size_t)n * 8, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        h->upload_bytes += (size_t)n * 12;
        // store-relative offsets -> offsets into the device buffer
        hipLaunchKernelGGL(po::k_abs_woff, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, h->d_woff.as<uint64_t>(), n, base1);
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_read_tile0.p, mh + (size_t)n * 12, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, h->stream));
    h->upload_bytes += ((size_t)n + 1) * 4;
    if (h->n_tiles) {
        hipLaunchKernelGGL(po::k_build_tiles, dim3(cdiv(h->n_tiles, 256)), dim3(256), 0, h->stream, h->d_woff.as<uint64_t>(),
                           h->d_len.as<uint32_t>(), h->d_read_tile0.as<uint32_t>(), n, h->n_tiles, h->d_tiles.as<po::TileRec>());
    }
    // exception records (2-bit mode only; usually none).  Up before anything that looks at the reads: the kernel that
    // rebuilds the odd reads on the device needs them, and so does every verify
    const size_t n_exc = h->bits == 2 ? h->exc_pos.size() : 0;
    if (n_exc) {
        PO_TRY(ensure(h, h->d_exc_off, ((size_t)n + 1) * 4));
        PO_TRY(ensure(h, h->d_exc_pos, n_exc * 4));
        PO_TRY(ensure(h, h->d_exc_byte, n_exc));
        HIP_TRY(h, hipMemcpyAsync(h->d_exc_off.p, staged(h, h->exc_off.data(), ((size_t)n + 1) * 4), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_exc_pos.p, staged(h, h->exc_pos.data(), n_exc * 4), n_exc * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_exc_byte.p, staged(h, h->exc_byte.data(), n_exc), n_exc, hipMemcpyHostToDevice, h->stream));
        h->upload_bytes += ((size_t)n + 1) * 4 + n_exc * 5;
    }
    h->n_exc_uploaded = n_exc;
    HIP_TRY(h, hipGetLastError());
    return PO_OK;
}

po_status upload(po_handle* h) {
    PO_TRY(init_device(h));
    if (!h->dirty) return PO_OK;
    const uint32_t n = (uint32_t)h->len.size();
    bool generate = false;
    PO_TRY(upload_meta(h, &generate));
    uint64_t* dw = h->d_words.as<uint64_t>();
    const uint64_t base1 = h->base1;
    if (h->asm_pieces) {
        // sharded upload: piece k = the store-0 words of shard k's reads, uploaded by rank k, gathered over xGMI
        if (!generate) return fail(h, PO_ERR_INVALID, "po_upload_assemble needs reads added as (x, reverse complement of x) pairs");
        // (layout of the gathered buffer: [part][shard][slot])
        for (uint32_t part = 0; part < h->asm_parts; ++part) {
            for (uint32_t k = 0; k < h->asm_n; ++k) {
                uint64_t wb = 0, wc = 0;
                store0_part_range(h, k, h->asm_n, part, h->asm_parts, &wb, &wc);
                if (wc > h->asm_slot_words) return fail(h, PO_ERR_INVALID, "po_upload_assemble: a piece is longer than the slot");
                if (wc) HIP_TRY(h, hipMemcpyAsync(dw + wb, h->asm_pieces + ((size_t)part * h->asm_n + k) * h->asm_slot_words, wc * 8,
                                                  hipMemcpyDeviceToDevice, h->stream));
            }
        }
    } else if (!h->words[0].empty()) {
        const void* src0 = store_is_pinned(h->words[0]) ? h->words[0].data() : staged(h, h->words[0].data(), h->words[0].size() * 8);
        HIP_TRY(h, hipMemcpyAsync(dw, src0, h->words[0].size() * 8, hipMemcpyHostToDevice, h->stream));
        h->upload_bytes += h->words[0].size() * 8;
    }
    if (!generate && !h->words[1].empty()) {
        const void* src1 = store_is_pinned(h->words[1]) ? h->words[1].data() : staged(h, h->words[1].data(), h->words[1].size() * 8);
        HIP_TRY(h, hipMemcpyAsync(dw + base1, src1, h->words[1].size() * 8, hipMemcpyHostToDevice, h->stream));
        h->upload_bytes += h->words[1].size() * 8;
    }
    if (generate) {
        hipLaunchKernelGGL(po::k_revcomp_store, dim3(cdiv((uint64_t)(n / 2) * 64, 256)), dim3(256), 0, h->stream, dw,
                           h->d_woff.as<uint64_t>(), h->d_len.as<uint32_t>(), 0u, n / 2,
                           h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr, h->d_exc_pos.as<uint32_t>());
        if (getenv("PHASM_VERIFY_GENERATED") && !h->words[1].empty()) {
            // test mode: the host's own store 1 is uploaded next to the generated one and compared word by word
            DevBuf tmp;
            PO_TRY(ensure(h, tmp, h->words[1].size() * 8));
            PO_TRY(ensure(h, h->d_scalars, 128));
            HIP_TRY(h, hipMemsetAsync(h->d_scalars.p, 0, 64, h->stream));
            HIP_TRY(h, hipMemcpyAsync(tmp.p, h->words[1].data(), h->words[1].size() * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(po::k_count_diff, dim3(1024), dim3(256), 0, h->stream, dw + base1, tmp.as<uint64_t>(),
                               (uint64_t)h->words[1].size(), h->d_scalars.as<unsigned long long>());
            HIP_TRY(h, hipMemcpyAsync(h->pinned, h->d_scalars.p, 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            tmp.release();
            if (h->pinned[0] != 0)
                return fail(h, PO_ERR_HIP, "generated reverse-complement store differs from the host's in " +
                                               std::to_string(h->pinned[0]) + " words");
        }
    }
    HIP_TRY(h, hipGetLastError());
    const size_t n_exc = h->n_exc_uploaded;   // (exception records went up with the per-read tables, upload_meta)
    // strand pairing: decides whether po_overlaps may compute one member of each mirror pair
    h->paired = false;
    const bool mirror_ok = !getenv("PHASM_NO_MIRROR");
    const bool try_paired = h->bits == 2 && n >= 2 && (n % 2) == 0 && mirror_ok && !generate;
    if (generate && mirror_ok) h->paired = true;  // (checked word by word on the host as the reads arrived)
    if (try_paired) {
        PO_TRY(ensure(h, h->d_scalars, 128));
        HIP_TRY(h, hipMemsetAsync(h->d_scalars.p, 0, 64, h->stream));
        const uint8_t* pair_state = nullptr;
        if (n_exc) {  // pairs with exception records were compared byte-wise on the host
            h->pair_state.resize(n / 2, 0);
            PO_TRY(ensure(h, h->d_pair_state, n / 2));
            HIP_TRY(h, hipMemcpyAsync(h->d_pair_state.p, staged(h, h->pair_state.data(), n / 2), n / 2, hipMemcpyHostToDevice, h->stream));
            pair_state = h->d_pair_state.as<uint8_t>();
        }
        hipLaunchKernelGGL(po::k_paired_check, dim3(cdiv((uint64_t)(n / 2) * 64, 256)), dim3(256), 0, h->stream,
                           h->d_words.as<uint64_t>(), h->d_woff.as<uint64_t>(), h->d_len.as<uint32_t>(), n / 2, pair_state,
                           h->d_scalars.as<uint32_t>());
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(h->pinned, h->d_scalars.p, 8, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipEventRecord(h->ev_up1, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (try_paired) h->paired = (uint32_t)h->pinned[0] == 0;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h->ev_up0, h->ev_up1);
    h->stats.ms_upload = ms;
    h->stats.upload_bytes = h->upload_bytes;
    ++h->upload_gen;
    h->dirty = false;
    return PO_OK;
}

// a-side shard: contiguous read ranges balanced by bases
void shard_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint32_t* r_begin, uint32_t* r_end, uint64_t* bases) {
    const uint32_t n = (uint32_t)h->len.size();
    // (the running sum of the read lengths is kept on the handle: a step of an N-rank job asks for dozens of shard and
    // piece ranges, and 100 k additions each time were milliseconds of host time per step)
    std::vector<uint64_t>& cum = const_cast<po_handle*>(h)->cum_len;
    if (cum.size() != (size_t)n + 1) {
        size_t have = cum.empty() ? 0 : std::min(cum.size() - 1, (size_t)n);   // (reads are only ever appended)
        cum.resize((size_t)n + 1);
        if (have == 0) cum[0] = 0;
        for (size_t r = have; r < n; ++r) cum[r + 1] = cum[r] + h->len[r];
    }
    auto cut = [&](uint32_t s) -> uint32_t {
        if (s == 0) return 0;
        if (s >= nshards) return n;
        const uint64_t target = (uint64_t)((__uint128_t)cum[n] * s / nshards);
        return (uint32_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
    };
    *r_begin = cut(shard);
    *r_end = cut(shard + 1);
    if (*r_end < *r_begin) *r_end = *r_begin;
    if (bases) *bases = cum[*r_end] - cum[*r_begin];
}

// exclusive scan of n items (u8 or u32) -> u32 offsets; *total_host gets the grand total
// (a pinned slot: valid after the next hipStreamSynchronize)
// the state of the single-pass scans (kernels.hip.h, ChainState): zero when allocated, left zero by every launch
po_status chain_state(po_handle* h, uint32_t n_tiles, po::ChainState** out, hipStream_t on) {
    const size_t need = po::chain_state_bytes(n_tiles);
    if (need > h->d_chain_state.cap) {
        PO_TRY(ensure(h, h->d_chain_state, std::max<size_t>(need * 2, 1u << 16)));
        HIP_TRY(h, hipMemsetAsync(h->d_chain_state.p, 0, h->d_chain_state.cap, on));
    }
    *out = h->d_chain_state.as<po::ChainState>();
    return PO_OK;
}

template <typename T>
po_status prefix_sum(po_handle* h, const T* in, uint64_t n, uint32_t* out, volatile uint64_t* total_host,
                     const uint64_t* also_src = nullptr, volatile uint64_t* also_host = nullptr, const uint32_t* extra = nullptr,
                     hipStream_t on = nullptr, uint64_t* total_dev_at = nullptr) {
    *total_host = 0;
    if (n == 0) return PO_OK;
    const hipStream_t ps = on ? on : h->stream;
    const uint32_t nblocks = cdiv(n, po::PS_TILE);
    uint64_t* total_dev = total_dev_at ? total_dev_at : h->d_scalars.as<uint64_t>();
    uint64_t* total_mapped = h->pinned_dev + (const_cast<uint64_t*>(total_host) - h->pinned);   // the slot as the device sees it
    uint64_t* also_mapped = also_host ? h->pinned_dev + (const_cast<uint64_t*>(also_host) - h->pinned) : nullptr;
    if (nblocks <= 64 && !getenv("PHASM_PS_CLASSIC")) {
        // ONE launch: decoupled look-back (k_ps_chain); `extra` is added to the input on the way (u32 inputs).  Short
        // inputs only -- the tile counts of a piece or a shard: hundreds of tiles spinning on status words across the
        // XCDs are slower than three launches

        po::ChainState* cs = nullptr;
        PO_TRY(chain_state(h, nblocks, &cs, ps));
        hipLaunchKernelGGL(po::k_ps_chain<T>, dim3(nblocks), dim3(po::PS_BLOCK), 0, ps, const_cast<T*>(in), extra, n, out, cs,
                           nblocks, total_dev, total_mapped, also_src, also_mapped);
        HIP_TRY(h, hipGetLastError());
        return PO_OK;
    }
    if (extra) {   // (u32 inputs: add in place first, as a launch of its own)
        hipLaunchKernelGGL(po::k_add_extra, dim3(cdiv(n, 256)), dim3(256), 0, ps, const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(in)),
                           extra, 0u, (uint32_t)n);
    }
    PO_TRY(ensure_piece(h, h->d_ps_blocks, (size_t)nblocks * 8));
    uint64_t* blocks = h->d_ps_blocks.as<uint64_t>();
    hipLaunchKernelGGL(po::k_ps_reduce<T>, dim3(nblocks), dim3(po::PS_BLOCK), 0, ps, in, n, blocks);
    hipLaunchKernelGGL(po::k_ps_spine, dim3(1), dim3(1024), 0, ps, blocks, nblocks, total_dev, total_mapped, also_src, also_mapped);
    hipLaunchKernelGGL(po::k_ps_down<T>, dim3(nblocks), dim3(po::PS_BLOCK), 0, ps, in, n, blocks, out);
    HIP_TRY(h, hipGetLastError());
    return PO_OK;
}

// one-workgroup form for short u32 inputs (kernels.hip.h, k_ps_small); `extra` is added into `in` first, `also_src`
// (a device counter) lands in the pinned slot `also_host`
po_status prefix_sum_small(po_handle* h, uint32_t* in, const uint32_t* extra, uint32_t n, uint32_t* out, volatile uint64_t* total_host,
                           const uint64_t* also_src, volatile uint64_t* also_host, hipStream_t on = nullptr, uint64_t* total_dev_at = nullptr) {
    *total_host = 0;
    if (n == 0) return PO_OK;
    const hipStream_t ps = on ? on : h->stream;
    uint64_t* total_dev = total_dev_at ? total_dev_at : h->d_scalars.as<uint64_t>();
    uint64_t* total_mapped = h->pinned_dev + (const_cast<uint64_t*>(total_host) - h->pinned);
    uint64_t* also_mapped = also_host ? h->pinned_dev + (const_cast<uint64_t*>(also_host) - h->pinned) : nullptr;
    hipLaunchKernelGGL(po::k_ps_small, dim3(1), dim3(1024), 0, ps, in, extra, n, out, total_dev, total_mapped,
                       also_src, also_mapped);
    HIP_TRY(h, hipGetLastError());
    return PO_OK;
}

void stage_times(po_stats& S, hipEvent_t* ev, bool ver_timed, bool full, bool pairs = true) {
    if (!full && !pairs) {   // (a piece of a streamed step that recorded nothing but its end)
        S.ms_index = S.ms_scan_count = S.ms_scan_fill = S.ms_verify = S.ms_select = S.ms_emit = 0.f;
        S.ms_total = S.ms_scan_probe = S.ms_verify_kernel = 0.f;
        return;
    }
    if (!full) {
        S.ms_index = S.ms_scan_count = S.ms_scan_fill = S.ms_verify = S.ms_select = S.ms_emit = 0.f;
        (void)hipEventElapsedTime(&S.ms_total, ev[EV_START], ev[EV_EMIT]);
        (void)hipEventElapsedTime(&S.ms_scan_probe, ev[EV_PROBE0], ev[EV_PROBE1]);
        S.ms_verify_kernel = 0.f;
        if (ver_timed) (void)hipEventElapsedTime(&S.ms_verify_kernel, ev[EV_VER0], ev[EV_VER1]);
        return;
    }
    (void)hipEventElapsedTime(&S.ms_index, ev[EV_START], ev[EV_INDEX]);
    (void)hipEventElapsedTime(&S.ms_scan_count, ev[EV_INDEX], ev[EV_COUNT]);
    (void)hipEventElapsedTime(&S.ms_scan_fill, ev[EV_COUNT], ev[EV_FILL]);
    (void)hipEventElapsedTime(&S.ms_verify, ev[EV_FILL], ev[EV_VERIFY]);
    (void)hipEventElapsedTime(&S.ms_select, ev[EV_VERIFY], ev[EV_SELECT]);
    (void)hipEventElapsedTime(&S.ms_emit, ev[EV_SELECT], ev[EV_EMIT]);
    (void)hipEventElapsedTime(&S.ms_total, ev[EV_START], ev[EV_EMIT]);
    (void)hipEventElapsedTime(&S.ms_scan_probe, ev[EV_PROBE0], ev[EV_PROBE1]);
    S.ms_verify_kernel = 0.f;
    if (ver_timed) (void)hipEventElapsedTime(&S.ms_verify_kernel, ev[EV_VER0], ev[EV_VER1]);
}

// the closing part of run_overlaps for a pending piece of a streamed step; the handle's stream has been synchronised
// since the piece was queued.  Returns the piece's row count.
// *needs_classic: k_tail found reads handed to the global (a, b) table and wrote nothing (the piece's workspaces have
// been reused by now: the streamed step gives up and the call takes the chunked form).
uint64_t finish_piece(po_handle* h, bool* needs_classic) {
    po_handle::Pending& P = h->st_pend;
    const int o = P.tail ? P.zone : 2, c = P.tail ? P.zone + 1 : 4;
    *needs_classic = P.tail && h->pinned[P.zone + 7] != 0;
    if (P.cap_c) {   // the count that was predicted: more than the buffers held -> nothing was computed
        P.S.n_candidates = h->pinned[P.zone + 8];
        if (P.S.n_candidates > P.cap_c) *needs_classic = true;
    }
    const uint64_t n_rows = h->pinned[o];
    P.S.n_rows = n_rows;
    P.S.n_verified = h->pinned[c];
    P.S.sum_overlap_bases = h->pinned[c + 1];
    P.S.verify_bytes_algo = h->pinned[c + 2];
    P.S.verify_bytes_exec = h->pinned[c + 3];
    stage_times(P.S, P.ev, P.ver_timed, P.full_events, P.pair_events);
    P.valid = false;
    return n_rows;
}

template <int BITS>
po_status run_overlaps(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, bool want_cands,
                       po_result* res) {
    constexpr uint32_t W = 64 / BITS;
    const uint32_t n = (uint32_t)h->len.size();
    const uint32_t m = min_length ? min_length : 1;  // a suffix array has no empty suffix
    const uint32_t K = m < W ? m : W;
    const uint64_t kmask = K * BITS >= 64 ? ~0ull : ((1ull << (K * BITS)) - 1ull);
    hipStream_t st = h->stream;
    po_stats& S = h->stats;
    const float keep_upload = S.ms_upload;
    S = po_stats();
    S.ms_upload = keep_upload;
    S.upload_bytes = h->upload_bytes;
    S.bits_per_base = BITS;
    S.kmer = K;
    S.n_reads = n;
    S.total_bases = h->total_bases;
    // strand-mirror mode + the read order that picks the canonical member of a mirror pair (keep_bits):
    // 1 = index order (whole-set calls), 2 = scrambled block order (sharded calls, balances verify work)
    const bool dp = h->ex_on;
    const uint32_t dpE = dp ? h->ex_E : 0u, dpW = dp && h->ex_E ? h->ex_W : 0u;
    // (inexact extension: every candidate is extended itself, no strand-mirror shortcut)
    // 3 = reversed index order, a piece of a streamed step (kernels.hip.h, mirror_rank): the scan keeps containment
    // candidates of any b, the verify kernel only those whose b has arrived (b < st_r_end)
    const bool streamed = h->st_on;
    const uint32_t paired = (BITS == 2 && h->paired && dpE == 0) ? (streamed ? po::PAIRED_STREAM_ALL : nshards > 1 ? 2u : 1u) : 0u;
    const uint32_t paired_ver = streamed ? po::paired_stream(h->st_r_end) : paired;
    if (streamed && (!paired || want_cands)) return fail(h, PO_ERR_INVALID, "streamed step without strand pairs");
    // two-stream pieces (po_handle::scan_stream): the counting pass of this piece runs on s1, beside the second half of the
    // piece before it on st
    const bool two = streamed && h->two_stream && !h->idx_only;
    const bool gated = two && h->two_stream == 2;
    bool gate_recorded = false;
    const hipStream_t s1 = two ? (gated ? h->rc_stream : h->scan_stream) : st;
    S.paired = paired ? 1u : 0u;
    S.max_diff = dpE;
    S.band = dpW;
    if (dp && dpE && want_cands) return fail(h, PO_ERR_INVALID, "the candidate (multi-GPU) form has no inexact mode");
    if (dp && dpE && BITS == 2 && !h->exc_pos.empty())
        return fail(h, PO_ERR_INVALID, "po_overlaps_ex with max_diff > 0 needs pure upper-case ACGT reads (or the 8-bit representation)");

    uint32_t r_begin = 0, r_end = n;
    S.shard_bases = h->total_bases;
    if (streamed) {
        r_begin = h->st_r_begin;
        r_end = h->st_r_end;
        S.shard_bases = 0;
        for (uint32_t r = r_begin; r < r_end; ++r) S.shard_bases += h->len[r];
    } else if (nshards > 1) {
        shard_range(h, shard, nshards, &r_begin, &r_end, &S.shard_bases);
    }
    const uint32_t tile_begin = h->h_read_tile0[r_begin], tile_end = h->h_read_tile0[r_end];
    const uint32_t ntiles = tile_end - tile_begin;
    S.n_tiles = ntiles;

    const uint64_t n_elig = count_eligible(h, m);
    S.n_eligible = n_elig;
    res->count = 0;
    if (n == 0 || n_elig == 0 || ntiles == 0) return PO_OK;

    // ---- index flavour.  narrow: prefix K-mer per read + LDS filter + every position probed (best up
    // to ~150 k reads: the filter needs ~10 bits per read in 128 KB of LDS).  wide: W K-mers per read,
    // only word-aligned K-mers of a probed, no filter (linear in the input; needs min_length >= 2W-1).
    bool wide = n_elig > 160000 && m >= 2 * W - 1;
    if (const char* e = getenv("PHASM_INDEX")) {
        if (!strcmp(e, "wide")) wide = m >= 2 * W - 1;
        if (!strcmp(e, "narrow")) wide = false;
    }
    S.wide_index = wide ? 1u : 0u;
    // wide index: window of the minimiser scheme (kernels.hip.h WideEnc) -- the largest of 16 / 4 / 1 words with
    // W ww + W - 1 <= min_length.  The streamed step keeps 1: its index is built from the first two words of every read,
    // which travel ahead of the pieces.  PHASM_WIDE_WINDOW=1|4|16 forces a smaller one (tests, A/B).
    uint32_t ww = 1;
    if (wide && BITS == 2) {
        ww = m >= W * 16 + W - 1 ? 16u : m >= W * 4 + W - 1 ? 4u : 1u;
        // (a streamed step's index is built from the words that travel ahead of the pieces: two per read, or five)
        if (streamed) ww = (ww >= 4 && h->st_lead >= LEAD_WORDS) ? 4u : 1u;
        if (const char* e = getenv("PHASM_WIDE_WINDOW")) {
            const uint32_t v = (uint32_t)atoi(e);
            if ((v == 1 || v == 4 || v == 16) && v <= ww) ww = v;
        }
    }
    bool WA_ext = false;
    const bool slice_build = h->sl_build_n > 1;     // build one sub-table of the sliced wide index, then stop
    const bool ext_idx = h->ext_index != nullptr;   // probe a gathered sliced index
    if (slice_build) {
        h->sl_is_wide = wide;
        if (!wide) return PO_OK;                    // (the narrow index is 0.06 ms: every rank builds its own)
    }
    if (ext_idx && !wide) return fail(h, PO_ERR_INVALID, "a sliced index was supplied, but this call uses the narrow index");
    bool rows_late = false;  // rows emitted into a kept buffer before their number reached the host
    bool cands_late = false, cands_ext = false;  // the same for compacted candidates (want_cands): destination chosen before the number was known
    uint64_t cands_cap = 0;
    bool self_clean = false;  // a piece of a streamed step that leaves the per-call counters zero for the next piece
    bool used_tail = false;  // the call's tail ran as k_tail (counts in pinned[tail_zone..], fallback flag in pinned[tail_zone + 7])
    int tail_zone = 48;
    std::function<po_status()> tail_fallback;
    bool ver_timed = false;  // the verify kernel ran (there were candidates): its own events are valid
    uint64_t n_keys = wide ? n_elig * W : n_elig;
    if (slice_build) n_keys = n_keys / h->sl_build_n + n_keys / (16ull * h->sl_build_n) + 4096;  // (a slice's share + slack)
    // ---- sizes
    uint32_t tbits = 10;
    // narrow: 5-10 slots per key, probed in aligned groups of four (kernels.hip.h PROBE_GROUP).  A probe for an
    // absent key needs more than its group when all four slots are taken: 0.75 % of the groups at 5.2 slots per
    // key (7 % at 2.6), and those positions go to the leftover list; 8 MB at config 2
    double table_mult = wide ? 2.0 : 5.0;
    if (const char* e = getenv("PHASM_TABLE_MULT")) table_mult = std::max(wide ? 2.0 : 1.5, atof(e));
    if (slice_build) table_mult = 1.5;   // (a sub-table travels over xGMI: the smallest power of two that keeps the load <= 2/3)
    while ((double)(1ull << tbits) < table_mult * (double)n_keys) ++tbits;
    if (ext_idx) tbits = h->ext_tbits;
    if (tbits > 30) return fail(h, PO_ERR_CAPACITY, "too many reads for the anchor table");
    // (+1: the slot of the all-ones key; narrow: three more so that a group fetch of that slot stays in bounds)
    const uint32_t nslots = (1u << tbits) + (wide ? 1u : po::PROBE_GROUP);
    uint32_t bloom_log2 = 13;
    while (bloom_log2 < 20 && (1ull << bloom_log2) < 16 * n_elig) ++bloom_log2;
    const size_t bloom_bytes = (size_t)1 << (bloom_log2 - 3);

    PO_TRY(ensure(h, h->d_scalars, 128));
    const size_t n_entries = wide ? (size_t)n * W : (size_t)n;     // index entries (slots of read_slot / chain)
    const size_t chain_elem = wide ? 8 : 4;
    if (!ext_idx) {   // (a supplied index needs none of the build's workspaces)
        PO_TRY(ensure(h, h->d_table, (size_t)nslots * sizeof(po::Slot)));
        PO_TRY(ensure(h, h->d_slot_cnt, (size_t)nslots * 4));
        PO_TRY(ensure(h, h->d_slot_cur, (size_t)nslots * 4));
        PO_TRY(ensure(h, h->d_slot_start, ((size_t)nslots + 1) * 4));
        PO_TRY(ensure(h, h->d_read_slot, n_entries * 4));
        // a slice's chain holds its share of the entries (+ slack; the exact number comes back from the prefix sum)
        const size_t chain_entries = slice_build ? n_entries / h->sl_build_n + n_entries / (4ull * h->sl_build_n) + 65536 : n_entries;
        PO_TRY(ensure(h, h->d_chain, std::min(chain_entries, n_entries) * chain_elem));
        PO_TRY(ensure(h, h->d_chain_tmp, std::min(chain_entries, n_entries) * chain_elem));
        PO_TRY(ensure(h, h->d_long_list, (size_t)nslots * 4));
        if (wide) PO_TRY(ensure(h, h->d_entry_off, n_entries * 2));
    }
    PO_TRY(ensure(h, h->d_bloom, bloom_bytes));
    PO_TRY(ensure(h, h->d_selfrep, (size_t)n * 4));
    PO_TRY(ensure(h, h->d_tile_count, ((size_t)h->n_tiles + 1) * 4));
    PO_TRY(ensure(h, h->d_tile_off, ((size_t)h->n_tiles + 2) * 4 * 2));   // (twice: the pieces of a streamed step alternate)
    PO_TRY(ensure(h, h->d_truemask, ((size_t)h->n_tiles + 1) * po::WAVE * 4));

    const uint64_t* words = h->d_words.as<uint64_t>();
    const uint64_t* woff = h->d_woff.as<uint64_t>();
    const uint32_t* len = h->d_len.as<uint32_t>();
    po::Slot* table = h->d_table.as<po::Slot>();
    uint32_t* slot_cnt = h->d_slot_cnt.as<uint32_t>();
    uint32_t* slot_cur = h->d_slot_cur.as<uint32_t>();
    uint32_t* slot_start = h->d_slot_start.as<uint32_t>();
    uint32_t* read_slot = h->d_read_slot.as<uint32_t>();
    uint32_t* chain = h->d_chain.as<uint32_t>();
    uint32_t* bloom = h->d_bloom.as<uint32_t>();
    uint32_t* selfrep = h->d_selfrep.as<uint32_t>();
    // The small state both halves of a piece touch exists twice, by the piece's parity: while the second half of piece k reads
    // its candidate count and tile offsets, the counting pass of piece k + 1 writes its own (two-stream pieces, see po_handle)
    const uint32_t parity = streamed ? (shard & 1u) : 0u;
    unsigned long long* scalars = h->d_scalars.as<unsigned long long>() + 8 * parity;  // [0] scan total, [1] n_long, [3] rows, [4..7] emit counters
    uint32_t* const tile_off_p = h->d_tile_off.as<uint32_t>() + (size_t)parity * ((size_t)h->n_tiles + 2);
    uint32_t* n_long = reinterpret_cast<uint32_t*>(scalars + 1);

    if (h->poison >= 0) {
        // PHASM_POISON: whatever earlier calls left in the per-call workspaces is replaced by the caller's byte
        DevBuf* ws[] = {&h->d_table, &h->d_slot_cnt, &h->d_slot_cur, &h->d_slot_start, &h->d_read_slot, &h->d_chain,
                        &h->d_chain_tmp, &h->d_long_list, &h->d_bloom, &h->d_selfrep, &h->d_tile_count, &h->d_tile_off,
                        &h->d_truemask, &h->d_ps_blocks, &h->d_left, &h->d_left_cnt, &h->d_tile_extra, &h->d_cand_a,
                        &h->d_cand_p, &h->d_cand_b, &h->d_type, &h->d_rowcnt, &h->d_row_off, &h->d_flag, &h->d_pair_key,
                        &h->d_vlabel, &h->d_vrank, &h->d_vperm, &h->spare_rows, &h->spare_cands};
        for (DevBuf* b : ws)
            if (b->p) HIP_TRY(h, hipMemsetAsync(b->p, h->poison, b->cap, st));
    }
    // ---- index: anchor table, chains, Bloom filter.  Built per call like the reference builds its suffix array per
    // call (overlapper.cpp:33-36) -- unless THIS handle built exactly this index for exactly this device copy of the
    // reads already (same upload, min_length, flavour, size): the chunks of po_overlaps_to_host and the shards of a
    // multi-GPU step then share one build instead of repeating it (the replicated part of a sharded step).
    {
        const int pe = phase_events_env();
        h->phase_events = pe >= 0 ? pe == 1 : !streamed;
        h->pair_events = h->phase_events || pe == 2 || (pe < 0 && !streamed);
    }
    if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_START], st));
    const bool reuse_index = !slice_build && !ext_idx && h->idx_valid && h->idx_gen == h->upload_gen && h->idx_m == m &&
                             h->idx_wide == wide && h->idx_tbits == tbits && h->idx_bits == (uint32_t)BITS && h->idx_ww == ww && h->poison < 0 &&
                             !getenv("PHASM_NO_INDEX_REUSE");
    S.index_reused = reuse_index ? 1u : 0u;
    h->idx_valid = false;
    // (narrow scan on a reused index: the reset kernel also clears the scan's two small per-call arrays, below)
    const bool fold_clear = reuse_index && !wide;
    if (fold_clear) {
    } else if (reuse_index || ext_idx) {
        hipLaunchKernelGGL(po::k_call_reset, dim3(cdiv(std::max(n, 16u), 256)), dim3(256), 0, s1, selfrep, n, scalars, 8u,
                           (uint32_t*)nullptr, 0u, (uint32_t*)nullptr, 0u);
    } else {
        const uint32_t bloom_words = (uint32_t)(bloom_bytes / 4);
        const uint32_t init_n = std::max(std::max(nslots, n), std::max(bloom_words, 8u));
        hipLaunchKernelGGL(po::k_call_init, dim3(cdiv(init_n, 256)), dim3(256), 0, st, table, nslots, slot_cnt, slot_cur, selfrep, n,
                           bloom, bloom_words, h->d_scalars.as<unsigned long long>());
    }
    if (reuse_index || ext_idx) {
        // (nothing to build)
    } else if (!wide) {
        hipLaunchKernelGGL(po::k_table_insert, dim3(cdiv(n, 256)), dim3(256), 0, st, words, woff, len, n, m, kmask, table,
                           tbits, slot_cnt, read_slot, bloom, bloom_log2, (uint32_t)BITS);
        PO_TRY(prefix_sum<uint32_t>(h, slot_cnt, nslots, slot_start, &h->pinned[0]));
        hipLaunchKernelGGL(po::k_chain_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, read_slot, n, slot_start, slot_cur, chain);
        hipLaunchKernelGGL(po::k_chain_sort_short<uint32_t>, dim3(cdiv(nslots, 256)), dim3(256), 0, st, slot_cnt, slot_start,
                           nslots, chain, h->d_long_list.as<uint32_t>(), n_long);
        hipLaunchKernelGGL(po::k_chain_sort_long<uint32_t>, dim3(64), dim3(256), 0, st, slot_cnt, slot_start,
                           h->d_long_list.as<uint32_t>(), n_long, chain, h->d_chain_tmp.as<uint32_t>());
        hipLaunchKernelGGL(po::k_table_finalize, dim3(cdiv(nslots, 256)), dim3(256), 0, st, table, nslots, slot_cnt,
                           slot_start, chain, len);
    } else {
        uint16_t* entry_off = h->d_entry_off.as<uint16_t>();
        auto k_ins = ww == 16 ? po::k_wide_insert<BITS, 16> : ww == 4 ? po::k_wide_insert<BITS, 4> : po::k_wide_insert<BITS, 1>;
        auto k_cfill = ww == 16 ? po::k_wide_chain_fill<BITS, 16> : ww == 4 ? po::k_wide_chain_fill<BITS, 4> : po::k_wide_chain_fill<BITS, 1>;
        auto k_fin = ww == 16 ? po::k_wide_finalize<BITS, 16> : ww == 4 ? po::k_wide_finalize<BITS, 4> : po::k_wide_finalize<BITS, 1>;
        hipLaunchKernelGGL(k_ins, dim3(cdiv(n_entries, 256)), dim3(256), 0, st, words, woff, len, n, m,
                           table, tbits, slot_cnt, read_slot, entry_off, slice_build ? h->sl_build_n : 1u, h->sl_build_slice);
        PO_TRY(prefix_sum<uint32_t>(h, slot_cnt, nslots, slot_start, &h->pinned[0]));
        if (slice_build) {
            // a sub-table's chain segment was sized for its expected share: learn the real number before anything is
            // written (repetitive reads put all their entries into one sub-table)
            HIP_TRY(h, hipStreamSynchronize(st));
            PO_TRY(ensure(h, h->d_chain, (size_t)h->pinned[0] * chain_elem));
            PO_TRY(ensure(h, h->d_chain_tmp, (size_t)h->pinned[0] * chain_elem));
        }
        uint64_t* chain64 = h->d_chain.as<uint64_t>();
        hipLaunchKernelGGL(k_cfill, dim3(cdiv(n_entries, 256)), dim3(256), 0, st, read_slot, entry_off,
                           (uint64_t)n_entries, slot_start, slot_cur, chain64, len);
        hipLaunchKernelGGL(po::k_chain_sort_short<uint64_t>, dim3(cdiv(nslots, 256)), dim3(256), 0, st, slot_cnt, slot_start,
                           nslots, chain64, h->d_long_list.as<uint32_t>(), n_long);
        hipLaunchKernelGGL(po::k_chain_sort_long<uint64_t>, dim3(64), dim3(256), 0, st, slot_cnt, slot_start,
                           h->d_long_list.as<uint32_t>(), n_long, chain64, h->d_chain_tmp.as<uint64_t>());
        hipLaunchKernelGGL(k_fin, dim3(cdiv(nslots, 256)), dim3(256), 0, st, table, nslots, slot_cnt,
                           slot_start, chain64, len);
    }
    // Which reads repeat their own prefix K-mer (selfrep: only their A candidates can be non-longest duplicates)?
    // The narrow whole-set scan finds that as a side effect.  The other cases do not scan every read for it -- a
    // pass over all positions cost 2.2 ms at config 3 with the wide index, and more than the shard's own scan at 8
    // shards -- and settle duplicates inside each read's own candidate list instead (k_select_local, below).
    HIP_TRY(h, hipGetLastError());
    if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_INDEX], st));
    if (slice_build) {
        // the sub-table and its chain segment are complete in the workspaces: po_index_slice_export copies them out
        HIP_TRY(h, hipStreamSynchronize(st));
        h->sl_tbits = tbits;
        h->sl_entries = h->pinned[0];   // (the prefix sum's total = chain entries of this sub-table)
        if (h->phase_events) (void)hipEventElapsedTime(&S.ms_index, h->ev[EV_START], h->ev[EV_INDEX]);
        return PO_OK;
    }
    if (ext_idx) {
        WA_ext = true;
    }
    if (h->idx_only) {
        // (streamed step: the index is built from the first words of every read while piece 0 is still on the wire; the
        // pieces find it valid -- same upload, min_length, flavour -- and reuse it)
        if (h->ev_idx) HIP_TRY(h, hipEventRecord(h->ev_idx, st));
        h->idx_valid = true;
        h->idx_gen = h->upload_gen;
        h->idx_m = m;
        h->idx_wide = wide;
        h->idx_tbits = tbits;
        h->idx_bits = (uint32_t)BITS;
        h->idx_ww = ww;
        return PO_OK;
    }
    h->idx_valid = !ext_idx;
    h->idx_gen = h->upload_gen;
    h->idx_m = m;
    h->idx_wide = wide;
    h->idx_tbits = tbits;
    h->idx_bits = (uint32_t)BITS;
    h->idx_ww = ww;

    if (two) {
        // the counting pass needs the index (and the per-read tables and tiles queued before it), this piece's reads with their
        // reverse complements, and the parity's small state free again: the second half of the piece two before this one
        HIP_TRY(h, hipStreamWaitEvent(s1, h->ev_idx, 0));
        if (gated) {
            // (this stream IS the one that writes the odd reads: piece k's reverse complements are queued here, in front of
            // its counting pass, not all up front -- a stream's commands run in order, and the pass of piece k must not sit
            // behind the reverse complements of pieces that have not landed)
            const uint32_t p0 = r_begin / 2, p1 = r_end / 2;
            HIP_TRY(h, hipStreamWaitEvent(s1, h->ev_piece[h->st_k], 0));
            if (p1 > p0)
                hipLaunchKernelGGL(po::k_revcomp_store, dim3(cdiv((uint64_t)(p1 - p0) * 64, 256)), dim3(256), 0, s1, h->d_words.as<uint64_t>(),
                                   h->d_woff.as<uint64_t>(), h->d_len.as<uint32_t>(), p0, p1,
                                   h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr, h->d_exc_pos.as<uint32_t>());
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipEventRecord(h->ev_rc[h->st_k], s1));
            // ... and the pass starts when the verify kernel of the piece before does: its one persistent workgroup per CU
            // needs a whole CU, so it moves in as the verify workgroups drain -- the tail of one big kernel under the start
            // of the next -- instead of taking the chip away from the small kernels in front of that verify kernel
            if (h->st_k >= 1) HIP_TRY(h, hipStreamWaitEvent(s1, h->ev_gate[(h->st_k - 1) & 1], 0));
        } else {
            HIP_TRY(h, hipStreamWaitEvent(s1, h->ev_rc[h->st_k], 0));
        }
        if (h->st_k >= 2) HIP_TRY(h, hipStreamWaitEvent(s1, h->ev[EV_DONE], 0));
    }
    // ---- scan, counting pass
    po::ScanArgs A = {};
    A.words = words;
    A.tiles = h->d_tiles.as<po::TileRec>();
    A.tile_begin = tile_begin;
    A.tile_end = tile_end;
    A.m = m;
    A.kmask = kmask;
    A.bloom = bloom;
    A.bloom_log2 = bloom_log2;
    A.table = table;
    A.tbits = tbits;
    A.chain = chain;
    A.len = len;
    A.paired = paired;
    A.selfrep = selfrep;
    A.n_selfrep = reinterpret_cast<uint32_t*>(scalars + 2);  // scalars[2] lo: reads with a self-repeating prefix
    A.tile_count = h->d_tile_count.as<uint32_t>();
    A.tile_off = tile_off_p;
    A.truemask = h->d_truemask.as<uint32_t>();
#ifdef PO_STAMPS
    PO_TRY(ensure(h, h->d_flag, (size_t)4096 * 64));
    HIP_TRY(h, hipMemsetAsync(h->d_flag.p, 0, (size_t)4096 * 64, st));
    A.dbg = h->d_flag.as<unsigned long long>();
#endif
    uint32_t scan_waves = po::SCAN_BLOCK / 64;
    if (const char* e = getenv("PHASM_SCAN_WAVES")) scan_waves = std::max(1, std::min(16, atoi(e)));
    const uint32_t scan_grid = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)h->n_cu, cdiv(ntiles, scan_waves)));
    po::WideArgs WA = {};
    WA.words = words;
    WA.tiles = A.tiles;
    WA.tile_begin = tile_begin;
    WA.tile_end = tile_end;
    WA.m = m;
    WA.table = table;
    WA.tbits = tbits;
    WA.chain = h->d_chain.as<uint64_t>();
    WA.n_slices = 1;
    if (WA_ext) {   // the gathered sliced index: N chunks of [sub-table | chain segment]
        WA.table = static_cast<const po::Slot*>(h->ext_index);
        WA.chain = nullptr;
        WA.n_slices = h->ext_slices;
        WA.chunk_slots = h->ext_chunk_slots;
        WA.chain_off_slots = h->ext_chain_off;
    }
    WA.len = len;
    WA.paired = paired;
    WA.tile_count = A.tile_count;
    WA.lane_slot = A.truemask;
    WA.tile_off = A.tile_off;
    if (wide) {
        if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_PROBE0], s1));
        auto wscan = streamed ? (ww == 4 ? po::k_wide_scan<BITS, false, BITS == 2, 4> : po::k_wide_scan<BITS, false, BITS == 2, 1>)
                     : ww == 16 ? po::k_wide_scan<BITS, false, false, 16> : ww == 4 ? po::k_wide_scan<BITS, false, false, 4> : po::k_wide_scan<BITS, false, false, 1>;
        hipLaunchKernelGGL(wscan, dim3(cdiv(ntiles, 4)), dim3(256), 0, s1, WA, po::CandGuard{nullptr, 0u});
        if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_PROBE1], s1));
    } else {
        const size_t scan_lds = (size_t)scan_waves * po::SCAN_LDS_PER_WAVE + bloom_bytes;
        if (scan_lds > h->lds_max) return fail(h, PO_ERR_HIP, "device LDS too small for the scan kernel");
        // (a streamed step -- 2-bit reads only -- has its own instantiations: the reversed pair order is a compile-time choice)
        constexpr bool CAN_STREAM = BITS == 2;
        auto probe_full = streamed ? po::k_scan_probe<BITS, true, CAN_STREAM> : po::k_scan_probe<BITS, true, false>;
        auto probe_part = streamed ? po::k_scan_probe<BITS, false, CAN_STREAM> : po::k_scan_probe<BITS, false, false>;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(probe_full), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(probe_part), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds));
        const uint32_t n_scan_waves = scan_grid * scan_waves;
        PO_TRY(ensure(h, h->d_left, (size_t)n_scan_waves * po::LEFT_CAP * sizeof(uint2)));
        PO_TRY(ensure(h, h->d_left_cnt, (size_t)n_scan_waves * 4));
        PO_TRY(ensure(h, h->d_tile_extra, ((size_t)h->n_tiles + 1) * 4));
        // (the pieces of a streamed step clean up after themselves -- k_scan_fixup zeroes its list counters, the fused tail its
        // counters, the first piece's reset clears tile_extra for every tile of the read set --: no reset launch per piece)
        self_clean = streamed && !getenv("PHASM_DEBUG_LEFT") && !getenv("PHASM_PIECE_RESET");
        if (fold_clear && self_clean && h->st_selfclean) {
            // nothing to launch
        } else if (fold_clear) {
            const uint32_t te0 = self_clean ? 0u : tile_begin, ten = self_clean ? h->n_tiles : ntiles;
            // (a self-cleaning step's first piece zeroes both parities' scalars; any other reset only its own block)
            hipLaunchKernelGGL(po::k_call_reset, dim3(cdiv(std::max(std::max(n, 16u), std::max(n_scan_waves, ten)), 256)), dim3(256), 0, s1,
                               selfrep, n, self_clean ? h->d_scalars.as<unsigned long long>() : scalars, self_clean ? 16u : 8u,
                               h->d_left_cnt.as<uint32_t>(), n_scan_waves, h->d_tile_extra.as<uint32_t>() + te0, ten);
        } else {
            HIP_TRY(h, hipMemsetAsync(h->d_left_cnt.p, 0, (size_t)n_scan_waves * 4, s1));
            HIP_TRY(h, hipMemsetAsync(h->d_tile_extra.as<uint32_t>() + tile_begin, 0, (size_t)ntiles * 4, s1));
        }
        A.left = h->d_left.as<uint2>();
        A.left_cnt = h->d_left_cnt.as<uint32_t>();
        A.tile_extra = h->d_tile_extra.as<uint32_t>();
        if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_PROBE0], s1));
        hipLaunchKernelGGL(K == W ? probe_full : probe_part, dim3(scan_grid), dim3(scan_waves * 64), scan_lds, s1, A);
        if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_PROBE1], s1));
        auto fixup = streamed ? po::k_scan_fixup<BITS, CAN_STREAM> : po::k_scan_fixup<BITS, false>;
        hipLaunchKernelGGL(fixup, dim3(n_scan_waves), dim3(256), 0, s1, A, n_scan_waves, self_clean ? 1u : 0u);
        // (the leftover counts, tile_extra, are added to the tile counts by the prefix sum below)
        if (getenv("PHASM_DEBUG_LEFT")) {  // how many positions did the scan waves defer to k_scan_fixup?
            PO_TRY(ensure_host(h, h->scratch_host, (size_t)n_scan_waves * 4));
            HIP_TRY(h, hipMemcpyAsync(h->scratch_host.p, h->d_left_cnt.p, (size_t)n_scan_waves * 4, hipMemcpyDeviceToHost, s1));
            HIP_TRY(h, hipStreamSynchronize(s1));
            const uint32_t* lc = static_cast<const uint32_t*>(h->scratch_host.p);
            uint64_t sum = 0;
            uint32_t mx = 0;
            for (uint32_t k = 0; k < n_scan_waves; ++k) sum += lc[k], mx = std::max(mx, lc[k]);
            std::fprintf(stderr, "[left] %u scan waves deferred %llu positions (max %u per wave, cap %d)\n", n_scan_waves,
                         (unsigned long long)sum, mx, (int)po::LEFT_CAP);
        }
    }
    HIP_TRY(h, hipGetLastError());
#ifdef PO_STAMPS
    if (!wide) {
        std::vector<unsigned long long> dbg(4096 * 8);
        HIP_TRY(h, hipMemcpyAsync(dbg.data(), h->d_flag.p, dbg.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        double sum[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < 4096; ++w) for (int k = 0; k < 6; ++k) sum[k] += (double)dbg[w * 8 + k];
        if (sum[5] > 0)
            std::fprintf(stderr, "[stamps] per pass (cycles): wait_rw %.0f  filter %.0f  wait_probe %.0f  consume %.0f  issue %.0f  (passes %.0f)\n",
                         sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5], sum[3] / sum[5], sum[4] / sum[5], sum[5]);
    }
#endif
    // ---- a piece of a streamed step whose candidate count is PREDICTED (the same piece of the previous call): nothing
    // waits for the counting pass -- the kernels behind it are launched at once, sized for `cap_c` candidates, and read
    // the real count on the device (CandGuard); the host learns it with the piece's other numbers when it collects the
    // piece.  Needs every per-piece buffer to hold cap_c already (a re-allocation would synchronise the device).
    const int zone = 64 + 16 * (int)(shard & 1u);   // this piece's slots of the pinned landing area
    uint32_t cap_c = 0;
    bool async_count = false, pred_order = true;
    if (streamed && h->st_harvest && h->st_pred_valid && !dp && !want_cands && !getenv("PHASM_SYNC_COUNT") && !getenv("PHASM_TAIL_CLASSIC")) {
        uint64_t pred = h->st_pred_cand[shard];
        uint64_t cap = pred + pred / 50 + 256;
        if (const char* e = getenv("PHASM_PRED_SCALE")) cap = pred = (uint64_t)((double)pred * atof(e));   // (tests: a prediction that is too small)
        const uint64_t worst = cap * 4u;
        bool order = pred >= 400000 && (r_end - r_begin) >= 4096;
        if (const char* e = getenv("PHASM_VERIFY_ORDER")) order = atoi(e) != 0;
        if (const char* e = getenv("PHASM_PIECE_ORDER")) order = order && atoi(e) != 0;   // (A/B: pieces without the locality order)
        pred_order = order;
        auto fits = [](const DevBuf& b, uint64_t bytes) { return b.p && b.cap >= bytes; };
        async_count = pred > 0 && cap < (16u << 20) && cdiv(cap, po::TAIL_TILE) <= po::TAIL_MAX_TILES &&
                      fits(h->d_cand_a, cap * 4) && fits(h->d_cand_p, cap * 4) && fits(h->d_cand_b, cap * 4) && fits(h->d_type, cap) &&
                      fits(h->d_rowcnt, cap) && fits(h->d_row_off, (cap + 1) * 4) &&
                      fits(h->spare_rows, h->home_on ? cap * sizeof(po::Cand) : worst * sizeof(po_row)) &&
                      (!order || (fits(h->d_vlabel, (uint64_t)(r_end - r_begin) * 4) && fits(h->d_vperm, (uint64_t)(r_end - r_begin) * 4) &&
                                  fits(h->d_vrank, (uint64_t)(r_end - r_begin) * 4))) && h->dev_words < 0xFFFFFFF0ull;
        cap_c = (uint32_t)cap;
    }
    volatile uint64_t* count_slot = async_count ? &h->pinned[zone + 8] : &h->pinned[1];
    volatile uint64_t* also_slot = async_count ? &h->pinned[zone + 9] : &h->pinned[8];
    if (ntiles <= po::PS_SMALL_MAX) {
        PO_TRY(prefix_sum_small(h, A.tile_count + tile_begin, wide ? nullptr : A.tile_extra + tile_begin, ntiles,
                                tile_off_p + tile_begin, count_slot,
                                reinterpret_cast<const uint64_t*>(scalars + 2), also_slot, s1, reinterpret_cast<uint64_t*>(scalars)));
    } else {
        PO_TRY(prefix_sum<uint32_t>(h, A.tile_count + tile_begin, ntiles, tile_off_p + tile_begin, count_slot,
                                    reinterpret_cast<const uint64_t*>(scalars + 2), also_slot,
                                    wide ? nullptr : A.tile_extra + tile_begin, s1, reinterpret_cast<uint64_t*>(scalars)));
    }
    if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_COUNT], s1));
    if (two) {   // everything behind the counting pass runs on the handle's stream, after it
        HIP_TRY(h, hipEventRecord(h->ev_s1[parity], s1));
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_s1[parity], 0));
    }
    po::CandGuard G = {nullptr, 0u};
    uint64_t n_cand64;
    uint32_t n_selfrep_reads;
    if (async_count) {
        // (sizes and grids below are for cap_c candidates; the kernels read the real number from scalars[0])
        n_cand64 = cap_c;
        n_selfrep_reads = 0;
        S.n_predicted = 1;
        G.n_dev = scalars;
        G.cap = cap_c;
    } else {
        if (h->st_pend.valid && h->st_harvest) {
            // the previous piece of a streamed step returned with its kernels queued; this piece's counting pass is queued
            // behind them now.  Wait for the previous piece alone, send its rows home, THEN wait for this piece's count:
            // the device is never idle while the host does that, and the rows leave the moment they exist
            HIP_TRY(h, hipEventSynchronize(h->st_pend.ev[EV_DONE]));
            PO_TRY(h->st_harvest());
        }
        HIP_TRY(h, hipStreamSynchronize(s1));
        n_cand64 = h->pinned[1];
        n_selfrep_reads = (uint32_t)h->pinned[8];
    }
    if (nshards > 1 || streamed || wide || dpE) n_selfrep_reads |= 1u;  // k_select_local may hand repetitive reads to the global selection
    S.n_candidates = n_cand64;
    if (n_cand64 >= 0xFFFFFF00ull)
        return fail(h, PO_ERR_CAPACITY, "candidate count " + std::to_string(n_cand64) + " exceeds one call's capacity (2^32)");
    const uint32_t n_cand = (uint32_t)n_cand64;

    uint64_t n_rows64 = 0;
    // (the tail's variables live at function scope: classic_tail may run after the block below, as k_tail's fallback)
    po::PairSlot* ptab = nullptr;
    uint32_t pbits = 0;
    uint32_t* n_deferred = reinterpret_cast<uint32_t*>(scalars + 1) + 1;  // reads k_select_local hands to the global table
    const uint32_t* gate = nullptr;
    const uint64_t worst_rows = (uint64_t)n_cand * (paired ? 4u : 2u);
    std::function<po_status()> classic_tail;
    // the longest-only selection inside each read's list runs in the verify kernel's epilogue where the exact verify runs
    // (not the DP kernels, which have no per-read workgroup); PHASM_SELECT_KERNEL=1 keeps the separate launch (tests compare)
    const bool sel_in_verify = (nshards > 1 || streamed || wide) && !dp && !getenv("PHASM_SELECT_KERNEL");
    if (n_cand) {
        PO_TRY(ensure_piece(h, h->d_cand_a, (size_t)n_cand * 4));
        PO_TRY(ensure_piece(h, h->d_cand_p, (size_t)n_cand * 4));
        PO_TRY(ensure_piece(h, h->d_cand_b, (size_t)n_cand * 4));
        PO_TRY(ensure_piece(h, h->d_type, (size_t)n_cand));
        PO_TRY(ensure_piece(h, h->d_rowcnt, (size_t)n_cand));
        PO_TRY(ensure_piece(h, h->d_row_off, ((size_t)n_cand + 1) * 4));
        A.cand_a = h->d_cand_a.as<uint32_t>();
        A.cand_p = h->d_cand_p.as<uint32_t>();
        A.cand_b = h->d_cand_b.as<uint32_t>();
        // ---- scan, fill pass
        if (wide) {
            WA.cand_a = A.cand_a;
            WA.cand_p = A.cand_p;
            WA.cand_b = A.cand_b;
            auto wfill = streamed ? (ww == 4 ? po::k_wide_scan<BITS, true, BITS == 2, 4> : po::k_wide_scan<BITS, true, BITS == 2, 1>)
                         : ww == 16 ? po::k_wide_scan<BITS, true, false, 16> : ww == 4 ? po::k_wide_scan<BITS, true, false, 4> : po::k_wide_scan<BITS, true, false, 1>;
            hipLaunchKernelGGL(wfill, dim3(cdiv(ntiles, 4)), dim3(256), 0, st, WA, G);
        } else {
            auto fill = streamed ? po::k_scan_fill<BITS, BITS == 2> : po::k_scan_fill<BITS, false>;
            hipLaunchKernelGGL(fill, dim3(cdiv(ntiles, 4 * po::FILL_TILES)), dim3(256), 0, st, A, G);
        }
        // locality order of the a-side reads (k_read_label): worth its ~50 us only when the verify is long
        const bool use_order_early = [&]() {
            bool u = n_cand >= 400000 && (r_end - r_begin) >= 4096;
            if (const char* e = getenv("PHASM_VERIFY_ORDER")) u = atoi(e) != 0;
            if (streamed)
                if (const char* e = getenv("PHASM_PIECE_ORDER")) u = u && atoi(e) != 0;
            if (async_count) u = pred_order;   // (decided from the predicted count, before anything was launched)
            return u && !dp;
        }();
        const bool defer_needed = streamed && r_end < n;
        if (defer_needed && !use_order_early) {
            // containment candidates whose b has not arrived: onto the deferred list (the verify kernel skips them);
            // when the locality order is computed, k_read_label's walk over the candidates does this on the way
            hipLaunchKernelGGL(po::k_defer_split, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b, n_cand,
                               r_end, h->d_defer.as<po::Cand>(), h->st_defer_cap, h->d_defer.as<uint32_t>() + (size_t)h->st_defer_cap * 4, G);
        }
        HIP_TRY(h, hipGetLastError());
        if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_FILL], st));
        // ---- verify
        if (dp) {
            // banded seed-extension DP, one wave per candidate (extend.hip.h); max_diff = 0 gives the packed compare's answer
            if (2ull * h->max_len + 64ull >= (1ull << 30))   // (the kernel's "infinity" plus the longest sweep must fit 32 bits)
                return fail(h, PO_ERR_CAPACITY, "po_overlaps_ex: reads of 2^29 bases or more are beyond the DP kernel");
            PO_TRY(ensure(h, h->d_end_a, (size_t)n_cand * 4));
            PO_TRY(ensure(h, h->d_end_b, (size_t)n_cand * 4));
            PO_TRY(ensure(h, h->d_dpcnt, 64));
            HIP_TRY(h, hipMemsetAsync(h->d_dpcnt.p, 0, 64, st));
            po::ExtArgs X = {};
            X.words = words;
            X.woff = woff;
            X.len = len;
            X.cand_a = A.cand_a;
            X.cand_p = A.cand_p;
            X.cand_b = A.cand_b;
            X.n_cand = n_cand;
            X.max_diff = dpE;
            X.band = dpW;
            X.paired = paired;
            X.exc_off = h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr;
            X.exc_pos = h->d_exc_pos.as<uint32_t>();
            X.exc_byte = h->d_exc_byte.as<uint8_t>();
            X.type = h->d_type.as<uint8_t>();
            X.end_a = h->d_end_a.as<uint32_t>();
            X.end_b = h->d_end_b.as<uint32_t>();
            X.counters = h->d_dpcnt.as<unsigned long long>();
            // two mappings of the same DP (extend.hip.h): a LANE per candidate (2-bit reads, band <= 15: the one an
            // overlap job wants -- millions of candidates, narrow bands), or a WAVE per candidate with a lane per
            // diagonal (any encoding, band <= 30); PHASM_DP_KERNEL=wave|lanes forces one (tests run both)
            // three mappings of the same DP (extend.hip.h), same rows bit for bit: a LANE per candidate with the band row as a
            // BIT VECTOR (k_extend_bits: 2-bit reads, band <= 15 -- the default where it applies), a lane per candidate with
            // the band row in registers (k_extend_lanes), a WAVE per candidate with a lane per diagonal (k_extend_dp: any
            // encoding, band <= 30).  PHASM_DP_KERNEL=bits|lanes|wave forces one (the tests run all three)
            bool lanes = BITS == 2 && dpW <= 15;
            bool bitvec = lanes;
            if (const char* e = getenv("PHASM_DP_KERNEL")) {
                if (!strcmp(e, "wave")) lanes = bitvec = false;
                if (!strcmp(e, "lanes")) bitvec = false;
                if ((!strcmp(e, "lanes") || !strcmp(e, "bits")) && !(BITS == 2 && dpW <= 15))
                    return fail(h, PO_ERR_INVALID, "PHASM_DP_KERNEL=lanes|bits needs 2-bit reads and band <= 15");
            }
            if (const char* e = getenv("PHASM_DP_KERNEL_SOFT"))   // (tests: "lanes" where a lane mapping applies at all, no error elsewhere)
                if (!strcmp(e, "lanes")) bitvec = false;
            S.dp_lanes = bitvec ? 2u : lanes ? 1u : 0u;
            if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VER0], st));
            if (lanes) {
                // candidates ordered by the rows they need, so that the 64 lanes of a wave finish together
                const uint32_t* perm = nullptr;
                bool sorted = n_cand >= 4096;
                if (const char* e = getenv("PHASM_DP_SORT")) sorted = atoi(e) != 0;   // (tests force it on small inputs)
                if (sorted) {
                    const uint32_t max_bins = (uint32_t)((std::min<size_t>(h->lds_max, 160 * 1024) - 1024) / 4);
                    const uint32_t max_rows = h->max_len + dpW;
                    uint32_t shift = 0;
                    while ((max_rows >> shift) + 1 > max_bins) ++shift;
                    const uint32_t n_bins = (max_rows >> shift) + 1;
                    const size_t sort_lds = ((size_t)n_bins + po::SORT_BLOCK / 64) * 4;
                    PO_TRY(ensure(h, h->d_vlabel, (size_t)n_cand * 4));
                    PO_TRY(ensure(h, h->d_vrank, (size_t)n_cand * 4));
                    PO_TRY(ensure(h, h->d_vperm, (size_t)n_cand * 4));
                    hipLaunchKernelGGL(po::k_dp_rows, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b, len, n_cand,
                                       dpW, h->d_vlabel.as<uint32_t>());
                    if (sort_lds > 48 * 1024)
                        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(po::k_read_sort),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
                    hipLaunchKernelGGL(po::k_read_sort, dim3(1), dim3(po::SORT_BLOCK), sort_lds, st, h->d_vlabel.as<uint32_t>(), n_cand,
                                       shift, n_bins, h->d_vrank.as<uint32_t>());
                    hipLaunchKernelGGL(po::k_read_invert, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, h->d_vrank.as<uint32_t>(), n_cand, 0u,
                                       h->d_vperm.as<uint32_t>());
                    perm = h->d_vperm.as<uint32_t>();
                }
                auto kern = bitvec ? po::k_extend_bits : dpW <= 4 ? po::k_extend_lanes<4> : dpW <= 8 ? po::k_extend_lanes<8> : po::k_extend_lanes<15>;
                hipLaunchKernelGGL(kern, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, X, perm);
            } else {
                hipLaunchKernelGGL((po::k_extend_dp<BITS>), dim3(cdiv(n_cand, 256 / po::WAVE)), dim3(256), 0, st, X);
            }
            if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VER1], st));
            HIP_TRY(h, hipMemcpyAsync(h->pinned + 32, h->d_dpcnt.p, 16, hipMemcpyDeviceToHost, st));
            ver_timed = true;
        } else {
            // a's words live in LDS (read length + 3 guard words); reads too long for 64 KB use the global path
            const uint64_t need_words = ((uint64_t)h->max_len + W - 1) / W + 3;
            const uint32_t lds_words_raw = (uint32_t)std::min<uint64_t>(need_words, 8192 - 1100 - (PO_VER_LDS_SWZ ? 256 : 0));  // (room for the records in 64 KB)
            const uint32_t n_a = r_end - r_begin;
            const uint32_t* perm = nullptr;
            // (sharded calls too: 0.65 -> 0.51 ms at 2 shards, 0.19 -> 0.17 at 8 -- once the label of a read ranked by
            // the scrambled order was turned back into a read index, see k_read_label)
            const bool use_order = use_order_early;
            if (use_order) {
                // bins of label >> shift, as many as fit into one workgroup's LDS.  Labels are read indices
                // (whole-set calls) or 32-bit scrambled ranks (sharded calls)
                const uint32_t max_bins = (uint32_t)((std::min<size_t>(h->lds_max, 160 * 1024) - 1024) / 4);
                uint32_t shift = 0, n_bins;
                while ((n >> shift) + 1 > max_bins) ++shift;   // labels are read indices in both pairing modes
                n_bins = (n >> shift) + 1;
                const size_t sort_lds = ((size_t)n_bins + po::SORT_BLOCK / 64) * 4;
                PO_TRY(ensure(h, h->d_vlabel, (size_t)n_a * 4));
                PO_TRY(ensure(h, h->d_vperm, (size_t)n_a * 4));
                po::DeferOut dfo = {};
                if (defer_needed) {
                    dfo.cand_p = A.cand_p;
                    dfo.b_limit = r_end;
                    dfo.list = h->d_defer.as<uint4>();
                    dfo.cap = h->st_defer_cap;
                    dfo.counter = h->d_defer.as<uint32_t>() + (size_t)h->st_defer_cap * 4;
                }
                hipLaunchKernelGGL(po::k_read_label, dim3(cdiv((uint64_t)n_a * 16, 256)), dim3(256), 0, st,
                                   h->d_read_tile0.as<uint32_t>(), tile_off_p, A.cand_b, r_begin, n_a, paired,
                                   h->d_vlabel.as<uint32_t>(), dfo, G);
                if (sort_lds > 48 * 1024)
                    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(po::k_read_sort),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
                PO_TRY(ensure(h, h->d_vrank, (size_t)n_a * 4));
                hipLaunchKernelGGL(po::k_read_sort, dim3(1), dim3(po::SORT_BLOCK), sort_lds, st, h->d_vlabel.as<uint32_t>(), n_a,
                                   shift, n_bins, h->d_vrank.as<uint32_t>());
                hipLaunchKernelGGL(po::k_read_invert, dim3(cdiv(n_a, 256)), dim3(256), 0, st, h->d_vrank.as<uint32_t>(), n_a, r_begin,
                                   h->d_vperm.as<uint32_t>());
                perm = h->d_vperm.as<uint32_t>();
            }
            const uint32_t ver_grid = perm ? 8 * ((n_a + 7) / 8) : n_a;
            // candidate records staged in LDS (word offsets must fit 32 bits); PHASM_VERIFY_STAGED=0 switches back
            bool staged = h->dev_words < 0xFFFFFFF0ull;
            if (const char* e = getenv("PHASM_VERIFY_STAGED")) staged = staged && atoi(e) != 0;
            auto verify = paired == 2u ? (staged ? po::k_verify_a<BITS, true, true> : po::k_verify_a<BITS, true, false>)
                                       : (staged ? po::k_verify_a<BITS, false, true> : po::k_verify_a<BITS, false, false>);
            if (streamed) verify = staged ? po::k_verify_a<BITS, false, true, BITS == 2> : po::k_verify_a<BITS, false, false, BITS == 2>;
            const uint32_t lds_words = (lds_words_raw + 1u) & ~1u;  // even: the records behind a sit on a 16-byte boundary
            const size_t ver_lds = (size_t)po::ver_a_words(lds_words) * 8 + (size_t)po::VREC_CAP * sizeof(po::VRec) + 16;
            if (ver_lds > 48 * 1024)
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(verify), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ver_lds));
            if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VER0], st));
            if (gated) {
                HIP_TRY(h, hipEventRecord(h->ev_gate[h->st_k & 1], st));
                gate_recorded = true;
            }
            hipLaunchKernelGGL(verify, dim3(ver_grid), dim3(po::VER_BLOCK), ver_lds, st,
                               words, woff, len, h->d_read_tile0.as<uint32_t>(), tile_off_p, A.cand_p,
                               A.cand_b, r_begin, lds_words, paired_ver,
                               h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr, h->d_exc_pos.as<uint32_t>(),
                               h->d_exc_byte.as<uint8_t>(), h->d_type.as<uint8_t>(), perm, n_a, G,
                               sel_in_verify ? selfrep : nullptr, sel_in_verify ? n_deferred : nullptr);
            if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VER1], st));
            ver_timed = true;
        }
        HIP_TRY(h, hipGetLastError());
#ifdef PO_VSTAMPS
        {
            unsigned long long v[64 * 8], z[64 * 8] = {};
            HIP_TRY(h, hipStreamSynchronize(st));
            HIP_TRY(h, hipMemcpyFromSymbol(v, HIP_SYMBOL(po::g_vstamps), sizeof(v)));
            HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(po::g_vstamps), z, sizeof(z)));
            double sum[8] = {};
            for (int i = 0; i < 64; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)v[i * 8 + k];
            if (sum[6] > 0)
                std::fprintf(stderr, "[vstamps] per wave (shader clocks): setup %.0f  wait_b %.0f  lds+cmp %.0f  book+init %.0f  life %.0f | iterations %.1f  group-steps/iteration %.2f  waves %.0f\n",
                             sum[0] / sum[6], sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6], sum[4] / sum[6], sum[5] / sum[6], sum[7] / sum[5], sum[6]);
        }
#endif
        if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VERIFY], st));
        // ---- select + row offsets
        if ((nshards > 1 || streamed || wide || dpE) && !sel_in_verify) {
            static_assert(po::SEL_CAP == 512, "k_select_local hashes to 9 bits");
            hipLaunchKernelGGL(po::k_select_local, dim3(cdiv(r_end - r_begin, 256 / po::WAVE)), dim3(256), 0, st,
                               h->d_read_tile0.as<uint32_t>(), tile_off_p, A.cand_p, A.cand_b, h->d_type.as<uint8_t>(),
                               r_begin, r_end - r_begin, selfrep, n_deferred, G);
        }
        // A candidate gives at most 2 rows (4 with their mirrors: worst_rows).  When the row buffer kept from an earlier
        // call holds that many, the rows are written without asking the host for their number first.
        // ---- the tail in ONE launch (k_tail: rows per candidate, their prefix sum, the rows, the counters) when the kept
        // row buffer holds the worst case and no global (a, b) table is needed -- or only a gated one: if k_select_local
        // hands a read over after all, k_tail writes nothing but a flag and the classic tail below runs instead
        // (rows home in compact form -- h->home_on, po_overlaps_to_host -- : the tail writes one 16-byte record per verified
        // candidate, so its buffer needs n_cand records, not the worst case of rows; a call whose kept buffer is too small
        // makes one here -- never a piece with a predicted count, whose buffers were checked before anything was launched)
        const bool home_tail_ok = h->home_on && !want_cands && !dpE && !getenv("PHASM_TAIL_CLASSIC") &&
                                  (!n_selfrep_reads || ((nshards > 1 || streamed || wide) && n_cand < (16u << 20))) &&
                                  cdiv(n_cand, po::TAIL_TILE) <= po::TAIL_MAX_TILES;
        if (home_tail_ok && !async_count && h->spare_rows.cap < (size_t)n_cand * sizeof(po::Cand))
            PO_TRY(ensure(h, h->spare_rows, (size_t)n_cand * sizeof(po::Cand) + (streamed ? 65536 : 0), 1.0, false));
        const bool compact_tail = home_tail_ok && h->spare_rows.p && h->spare_rows.cap >= (size_t)n_cand * sizeof(po::Cand);
        const bool can_tail = compact_tail ||
                              (!want_cands && !dpE && h->spare_rows.p && h->spare_rows.cap >= worst_rows * sizeof(po_row) &&
                               (!n_selfrep_reads || ((nshards > 1 || streamed || wide) && n_cand < (16u << 20))) &&
                               // (rows left in HBM, whole set: 6.5 M candidates are 0.02 ms faster through the classic kernels)
                               cdiv(n_cand, po::TAIL_TILE) <= (streamed || h->home_on ? po::TAIL_MAX_TILES : 4096u) && !getenv("PHASM_TAIL_CLASSIC"));
        // (a piece with a predicted count has nothing but the fused tail: the classic kernels take the real count from the host)
        if (async_count && !can_tail) return fail(h, PO_ERR_HIP, "internal: a piece with a predicted candidate count needs the fused tail");
        classic_tail = [&]() -> po_status {
        if (n_selfrep_reads) {
            // some read's prefix recurs inside it (or a read was too repetitive for k_select_local): A candidates
            // of such b may be non-longest duplicates
            uint32_t n_sus;
            if ((nshards > 1 || streamed || wide) && n_cand < (4u << 20)) {
                n_sus = n_cand;  // upper bound: no counting pass, no host round trip (a big call sizes its table exactly)
                gate = n_deferred;  // ... and nothing of it is touched unless k_select_local handed a read over
            } else {
                uint32_t* n_suspect = reinterpret_cast<uint32_t*>(scalars + 2) + 1;
                hipLaunchKernelGGL(po::k_count_suspects, dim3(std::min<uint32_t>(cdiv(n_cand, 256), (uint32_t)h->n_cu * 8)), dim3(256), 0,
                                   st, A.cand_b, h->d_type.as<uint8_t>(), n_cand, selfrep, n_suspect);
                HIP_TRY(h, hipMemcpyAsync(h->pinned + 8, scalars + 2, 8, hipMemcpyDeviceToHost, st));
                HIP_TRY(h, hipStreamSynchronize(st));
                n_sus = (uint32_t)(h->pinned[8] >> 32);
            }
            if (n_sus) {
                pbits = 4;
                while ((1ull << pbits) < 2ull * n_sus) ++pbits;
                PO_TRY(ensure_piece(h, h->d_pair_key, ((size_t)1 << pbits) * sizeof(po::PairSlot)));
                hipLaunchKernelGGL(po::k_fill_gated, dim3((uint32_t)h->n_cu * 8), dim3(256), 0, st, h->d_pair_key.as<uint4>(),
                                   (uint64_t)((size_t)1 << pbits) * sizeof(po::PairSlot) / 16, 0xFFFFFFFFu, gate);
                ptab = h->d_pair_key.as<po::PairSlot>();
                hipLaunchKernelGGL(po::k_select_mark, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b,
                                   h->d_type.as<uint8_t>(), n_cand, selfrep, ptab, pbits, gate);
            }
        }
        if (want_cands) PO_TRY(ensure(h, h->d_flag, (size_t)n_cand));
        hipLaunchKernelGGL(po::k_select, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b,
                           h->d_type.as<uint8_t>(), n_cand, selfrep, ptab, pbits, paired, h->d_rowcnt.as<uint8_t>(),
                           want_cands ? h->d_flag.as<uint8_t>() : nullptr, gate);
        HIP_TRY(h, hipGetLastError());
        // A candidate gives at most 2 rows (4 with their mirrors).  When the row buffer kept from an earlier call
        // holds that many, the rows are emitted without asking the host for their number first (one host round
        // trip less per step); the number arrives with the counters at the end.
        if (!want_cands) {
            PO_TRY(prefix_sum<uint8_t>(h, h->d_rowcnt.as<uint8_t>(), n_cand, h->d_row_off.as<uint32_t>(), &h->pinned[2]));
            if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_SELECT], st));
            rows_late = h->spare_rows.p && h->spare_rows.cap >= worst_rows * sizeof(po_row);
            if (!rows_late) {
                HIP_TRY(h, hipStreamSynchronize(st));
                n_rows64 = h->pinned[2];
                if (n_rows64 >= 0xFFFFFF00ull) return fail(h, PO_ERR_CAPACITY, "row count exceeds one call's capacity (2^32)");
            }
        }
        if (want_cands) {
            // ---- multi-GPU form: hand out the verified candidates (one per strand-mirror pair), compacted.  Where the
            // destination is known to be large enough for what the call is expected to keep -- the caller's exchange slot,
            // or a buffer kept from an earlier call that holds even the worst case -- the compaction is queued without
            // asking the host for the number first (one host round trip less per shard call: ~40 us of ~0.5 ms at 8
            // shards); the number arrives with the closing synchronisation, and a slot that turns out too small is
            // handled there (cands_late below)
            PO_TRY(prefix_sum<uint8_t>(h, h->d_flag.as<uint8_t>(), n_cand, h->d_row_off.as<uint32_t>(), &h->pinned[3]));
            if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_SELECT], st));
            po::Cand* dst;
            uint64_t dst_cap;
            if (res->ext_dst && res->ext_cap && !getenv("PHASM_COMPACT_SYNC")) {
                dst = static_cast<po::Cand*>(res->ext_dst);  // straight into the caller's exchange buffer
                dst_cap = res->ext_cap;
                cands_late = true;
                cands_ext = true;
            } else if (h->spare_cands.p && h->spare_cands.cap >= (size_t)n_cand * sizeof(po::Cand) && !getenv("PHASM_COMPACT_SYNC")) {
                res->d_rows = h->spare_cands;
                h->spare_cands = DevBuf();
                dst = res->d_rows.as<po::Cand>();
                dst_cap = n_cand;
                cands_late = true;
            } else {
                HIP_TRY(h, hipStreamSynchronize(st));
                const uint64_t n_ver = h->pinned[3];
                if (res->ext_dst && n_ver <= res->ext_cap) {
                    dst = static_cast<po::Cand*>(res->ext_dst);
                    res->wrote_ext = true;
                } else {
                    if (h->spare_cands.p && h->spare_cands.cap >= n_ver * sizeof(po::Cand)) {
                        res->d_rows = h->spare_cands;
                        h->spare_cands = DevBuf();
                    }
                    PO_TRY(ensure(h, res->d_rows, std::max<size_t>(n_ver * sizeof(po::Cand), 256), 1.0, false));
                    dst = res->d_rows.as<po::Cand>();
                }
                dst_cap = n_ver;
                n_rows64 = n_ver;
            }
            cands_cap = dst_cap;
            hipLaunchKernelGGL(po::k_compact, dim3(cdiv(n_cand, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b,
                               h->d_type.as<uint8_t>(), h->d_flag.as<uint8_t>(), h->d_row_off.as<uint32_t>(), n_cand, dst,
                               (uint32_t)std::min<uint64_t>(dst_cap, 0xFFFFFFFFull));
            HIP_TRY(h, hipGetLastError());
            res->elem = sizeof(po::Cand);
        } else {
            // ---- emit
            if (rows_late || (h->spare_rows.cap >= n_rows64 * sizeof(po_row) && h->spare_rows.p)) {
                res->d_rows = h->spare_rows;
                h->spare_rows = DevBuf();
            } else {
                h->spare_rows.release();
            }
            if (!rows_late) {
                // (room for the worst case up to 2 GiB, so that the next call of this size need not ask)
                const size_t exact = n_rows64 * sizeof(po_row);
                size_t roomy = worst_rows * sizeof(po_row) <= (2ull << 30) ? (size_t)(worst_rows * sizeof(po_row)) : 0;
                if (roomy && streamed) roomy += 4096 * sizeof(po_row);   // (room for the next call's predicted count and its slack)
                PO_TRY(ensure(h, res->d_rows, std::max<size_t>(std::max(exact, roomy), 256), 1.0, false));
            }
            if (dpE)
                hipLaunchKernelGGL(po::k_emit_ex, dim3(std::min<uint32_t>(cdiv(n_cand, 256), (uint32_t)h->n_cu * 16)), dim3(256), 0,
                                   st, A.cand_a, A.cand_p, A.cand_b, h->d_type.as<uint8_t>(), h->d_end_a.as<uint32_t>(),
                                   h->d_end_b.as<uint32_t>(), h->d_row_off.as<uint32_t>(), n_cand, len,
                                   res->d_rows.as<po::Row>(), (uint32_t)BITS, scalars + 4);
            else
            hipLaunchKernelGGL(po::k_emit, dim3(std::min<uint32_t>(cdiv(n_cand, 256), (uint32_t)h->n_cu * 16)), dim3(256), 0,
                               st, A.cand_a, A.cand_p, A.cand_b, h->d_type.as<uint8_t>(), h->d_row_off.as<uint32_t>(),
                               n_cand, len, res->d_rows.as<po::Row>(), (uint32_t)BITS, paired, scalars + 4);
            HIP_TRY(h, hipGetLastError());
        }
        return PO_OK;
        };
        if (can_tail) {
            const uint32_t n_tt = cdiv(n_cand, po::TAIL_TILE);
            // [tile row sums x n_tt | done counter]; the counter is zero between launches (allocated zero, reset by k_tail)
            if (((size_t)po::TAIL_MAX_TILES + 4) * 4 > h->d_tail_state.cap) {
                PO_TRY(ensure(h, h->d_tail_state, ((size_t)po::TAIL_MAX_TILES + 4) * 4));
                HIP_TRY(h, hipMemsetAsync(h->d_tail_state.p, 0, h->d_tail_state.cap, st));
            }
            uint32_t* tile_rows = h->d_tail_state.as<uint32_t>();
            uint32_t* tail_done = tile_rows + po::TAIL_MAX_TILES;
            const uint32_t* tgate = n_selfrep_reads ? n_deferred : nullptr;
            hipLaunchKernelGGL(compact_tail ? po::k_tile_rows<true> : po::k_tile_rows<false>, dim3(n_tt), dim3(po::TAIL_BLOCK), 0, st, A.cand_a,
                               A.cand_b, h->d_type.as<uint8_t>(), n_cand, paired, tgate, tile_rows, G);
            if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_SELECT], st));
            rows_late = true;
            used_tail = true;
            S.fused_tail = 1;
            res->d_rows = h->spare_rows;
            h->spare_rows = DevBuf();
            tail_zone = async_count ? zone : 48;
            h->pinned[tail_zone] = 0;
            h->pinned[tail_zone + 7] = 0;
            if (compact_tail) {
                res->compact = true;
                hipLaunchKernelGGL(po::k_tail_cands, dim3(n_tt), dim3(po::TAIL_BLOCK), 0, st, A.cand_a, A.cand_p, A.cand_b, h->d_type.as<uint8_t>(),
                                   n_cand, res->d_rows.as<po::Cand>(), paired, tgate, tile_rows, n_tt, tail_done,
                                   scalars + 3, h->pinned_dev + tail_zone, G, h->home_sh_b, h->home_sh_p);
            } else
            hipLaunchKernelGGL(po::k_tail, dim3(n_tt), dim3(po::TAIL_BLOCK), 0, st, A.cand_a, A.cand_p, A.cand_b, h->d_type.as<uint8_t>(),
                               n_cand, len, res->d_rows.as<po::Row>(), (uint32_t)BITS, paired, tgate, tile_rows, n_tt, tail_done,
                               scalars + 4, h->pinned_dev + tail_zone, G);
            HIP_TRY(h, hipGetLastError());
        } else {
            PO_TRY(classic_tail());
        }
        tail_fallback = [&]() -> po_status {
            // k_tail found reads handed to the global table: nothing was written -- the classic tail, now
            h->spare_rows = res->d_rows;
            res->d_rows = DevBuf();
            res->compact = false;
            rows_late = false;
            used_tail = false;
            S.fused_tail = 0;
            S.tail_fallback = 1;
            PO_TRY(classic_tail());
            return PO_OK;
        };
    } else {
        if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_FILL], st));
        if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_VERIFY], st));
        if (h->phase_events) HIP_TRY(h, hipEventRecord(h->ev[EV_SELECT], st));
    }
    if (h->pair_events) HIP_TRY(h, hipEventRecord(h->ev[EV_EMIT], st));
    uint64_t* counters = h->pinned + 4;
    if (!used_tail) HIP_TRY(h, hipMemcpyAsync(counters, scalars + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    res->unique_twins = !want_cands && paired != 0 && !dpE;
    if (gated && !gate_recorded) HIP_TRY(h, hipEventRecord(h->ev_gate[h->st_k & 1], st));   // (a piece without candidates: nothing to wait behind)
    HIP_TRY(h, hipEventRecord(h->ev[EV_DONE], st));
    h->st_selfclean = streamed && self_clean && used_tail && !wide;   // (the next piece of this step may skip its reset)
    if (async_count && h->st_pend.valid) {
        // everything of this piece is queued; now collect the piece before it (its numbers sit in the other zone)
        HIP_TRY(h, hipEventSynchronize(h->st_pend.ev[EV_DONE]));
        PO_TRY(h->st_harvest());
    }
    if (streamed && rows_late && h->st_harvest && !h->st_pend.valid) {
        h->st_pend.tail = used_tail;
        h->st_pend.compact = res->compact;
        h->st_pend.zone = tail_zone;
        h->st_pend.cap_c = async_count ? cap_c : 0u;
        // a piece of a streamed step with its rows in a buffer known to be large enough: nothing here needs the host
        // to wait -- the counts are read when the next piece waits for ITS candidate count (finish_piece)
        h->st_pend.valid = true;
        h->st_pend.k = shard;
        h->st_pend.S = S;
        h->st_pend.ver_timed = ver_timed;
        h->st_pend.full_events = h->phase_events;
        h->st_pend.pair_events = h->pair_events;
        h->st_pend.ev = h->ev;
        res->count = 0;
        return PO_OK;
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    if (async_count) return fail(h, PO_ERR_HIP, "internal: a piece with a predicted count must stay pending");
    if (used_tail && h->pinned[tail_zone + 7]) {
        PO_TRY(tail_fallback());
        HIP_TRY(h, hipEventRecord(h->ev[EV_EMIT], st));
        HIP_TRY(h, hipMemcpyAsync(counters, scalars + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipEventRecord(h->ev[EV_DONE], st));
        HIP_TRY(h, hipStreamSynchronize(st));
    }
    if (cands_late) {
        const uint64_t n_ver = h->pinned[3];
        if (n_ver > cands_cap) {
            // the exchange slot was too small for this shard (the caller sizes it from the previous step): once more, into
            // a buffer of the library's own -- the candidate arrays, flags and offsets are all still in place
            PO_TRY(ensure(h, res->d_rows, std::max<size_t>(n_ver * sizeof(po::Cand), 256), 1.0, false));
            hipLaunchKernelGGL(po::k_compact, dim3(cdiv((uint32_t)S.n_candidates, 256)), dim3(256), 0, st, A.cand_a, A.cand_p, A.cand_b,
                               h->d_type.as<uint8_t>(), h->d_flag.as<uint8_t>(), h->d_row_off.as<uint32_t>(), (uint32_t)S.n_candidates,
                               res->d_rows.as<po::Cand>(), (uint32_t)n_ver);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(st));
        } else if (cands_ext) {
            res->wrote_ext = true;
        }
        n_rows64 = n_ver;
    }
    if (used_tail) {
        n_rows64 = h->pinned[tail_zone];
        for (int k = 0; k < 4; ++k) counters[k] = h->pinned[tail_zone + 1 + k];
    } else if (rows_late) {
        n_rows64 = h->pinned[2];  // (<= worst_rows < 2^32 by construction of the fast path)
    }
    res->count = n_rows64;
    S.n_rows = want_cands ? 0 : n_rows64;
    S.n_verified = want_cands ? n_rows64 : counters[0];
    S.sum_overlap_bases = counters[1];
    S.verify_bytes_algo = counters[2];
    S.verify_bytes_exec = counters[3];
    if (dp && n_cand64) {
        S.dp_steps = h->pinned[32];
        S.dp_stopped = h->pinned[33];
    }
    stage_times(S, h->ev, ver_timed, h->phase_events, h->pair_events);
    return PO_OK;
}

// rows from a (merged) verified-candidate array that lives on this device
po_status run_expand(po_handle* h, const void* d_cands, uint64_t n, po_result* res) {
    hipStream_t st = h->stream;
    po_stats& S = h->stats;
    res->count = 0;
    if (n == 0) return PO_OK;
    if (n >= 0xFFFFFF00ull) return fail(h, PO_ERR_CAPACITY, "candidate count exceeds one call's capacity (2^32)");
    const uint32_t nc = (uint32_t)n;
    const uint32_t paired = (h->bits == 2 && h->paired) ? 1u : 0u;
    PO_TRY(ensure(h, h->d_scalars, 128));
    PO_TRY(ensure(h, h->d_rowcnt, (size_t)nc));
    PO_TRY(ensure(h, h->d_row_off, ((size_t)nc + 1) * 4));
    unsigned long long* scalars = h->d_scalars.as<unsigned long long>();
    const po::Cand* cands = static_cast<const po::Cand*>(d_cands);
    HIP_TRY(h, hipEventRecord(h->ev[EV_SELECT], st));
    HIP_TRY(h, hipMemsetAsync(h->d_scalars.p, 0, 64, st));
    hipLaunchKernelGGL(po::k_cand_rowcnt, dim3(cdiv(nc, 256)), dim3(256), 0, st, cands, nc, (uint32_t)h->len.size(), paired,
                       h->d_rowcnt.as<uint8_t>(), reinterpret_cast<uint32_t*>(scalars + 3));  // [0] is the prefix-sum total
    HIP_TRY(h, hipGetLastError());
    PO_TRY(prefix_sum<uint8_t>(h, h->d_rowcnt.as<uint8_t>(), nc, h->d_row_off.as<uint32_t>(), &h->pinned[2]));
    HIP_TRY(h, hipMemcpyAsync(h->pinned + 3, scalars + 3, 8, hipMemcpyDeviceToHost, st));
    // (as in po_overlaps: a kept row buffer that holds the worst case -- 4 rows per candidate -- is written without
    // waiting for the count; invalid entries have no rows, so emitting before the check writes nothing for them)
    const uint64_t worst_rows = (uint64_t)nc * (paired ? 4u : 2u);
    const bool rows_late = h->spare_rows.p && h->spare_rows.cap >= worst_rows * sizeof(po_row);
    uint64_t n_rows = 0;
    if (!rows_late) {
        HIP_TRY(h, hipStreamSynchronize(st));
        if ((uint32_t)h->pinned[3] != 0) return fail(h, PO_ERR_INVALID, "po_expand: candidate array holds invalid entries");
        n_rows = h->pinned[2];
        if (n_rows >= 0xFFFFFF00ull) return fail(h, PO_ERR_CAPACITY, "row count exceeds one call's capacity (2^32)");
    }
    if (rows_late || (h->spare_rows.cap >= n_rows * sizeof(po_row) && h->spare_rows.p)) {
        res->d_rows = h->spare_rows;
        h->spare_rows = DevBuf();
    } else {
        h->spare_rows.release();
    }
    if (!rows_late) {
        const size_t exact = n_rows * sizeof(po_row);
        const size_t roomy = worst_rows * sizeof(po_row) <= (2ull << 30) ? (size_t)(worst_rows * sizeof(po_row)) : 0;
        PO_TRY(ensure(h, res->d_rows, std::max<size_t>(std::max(exact, roomy), 256), 1.0, false));
    }
    HIP_TRY(h, hipMemsetAsync(h->d_scalars.p, 0, 64, st));
    hipLaunchKernelGGL(po::k_emit_cands, dim3(std::min<uint32_t>(cdiv(nc, 256), (uint32_t)h->n_cu * 16)), dim3(256), 0, st,
                       cands, h->d_rowcnt.as<uint8_t>(), h->d_row_off.as<uint32_t>(), nc, h->d_len.as<uint32_t>(),
                       res->d_rows.as<po::Row>(), (uint32_t)h->bits, paired, scalars + 4);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->ev[EV_EMIT], st));
    HIP_TRY(h, hipMemcpyAsync(h->pinned + 4, scalars + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    if (rows_late) {
        if ((uint32_t)h->pinned[3] != 0) return fail(h, PO_ERR_INVALID, "po_expand: candidate array holds invalid entries");
        n_rows = h->pinned[2];
    }
    res->count = n_rows;
    res->unique_twins = paired != 0;
    S.n_rows = n_rows;
    S.sum_overlap_bases = h->pinned[5];
    S.verify_bytes_algo = h->pinned[6];
    S.verify_bytes_exec = h->pinned[7];
    (void)hipEventElapsedTime(&S.ms_emit, h->ev[EV_SELECT], h->ev[EV_EMIT]);
    return PO_OK;
}


// ---- layout stage 1 (po_layout_edges): rows -> contained reads + assembly-graph edges ----------

// ids come in strand pairs: ids[2i] = name + "+", ids[2i+1] = name + "-"  (assembler.py:39-40)
bool ids_are_strand_pairs(po_handle* h) {
    if (h->ids_paired >= 0) return h->ids_paired == 1;
    bool ok = (h->ids.size() % 2) == 0;
    for (size_t i = 0; ok && i < h->ids.size(); i += 2) {
        const std::string &p = h->ids[i], &m = h->ids[i + 1];
        ok = !p.empty() && p.size() == m.size() && p.back() == '+' && m.back() == '-' &&
             std::memcmp(p.data(), m.data(), p.size() - 1) == 0;
    }
    h->ids_paired = ok ? 1 : 0;
    return ok;
}

po_status rows_to_device(po_handle* h, po_result* r) {
    if (r->count == 0 || r->d_rows.p) return PO_OK;
    if (!r->host) return fail(h, PO_ERR_INVALID, "result holds no rows");
    PO_TRY(ensure(h, r->d_rows, r->count * r->elem, 1.0, false));
    HIP_TRY(h, hipMemcpyAsync(r->d_rows.p, r->host, r->count * r->elem, hipMemcpyHostToDevice, h->stream));
    return PO_OK;
}

po_status run_layout(po_handle* h, po_result* rows, const po_layout_params& prm, uint8_t* removed_out, po_result* res) {
    PO_TRY(init_device(h));
    hipStream_t st = h->stream;
    po_layout_stats& L = h->lstats;
    L = po_layout_stats();
    res->count = 0;
    res->elem = sizeof(po_edge);
    res->kind_edges = true;
    const uint32_t n_nodes = (uint32_t)h->len.size();
    const uint64_t n_rows64 = rows->count;
    if (n_rows64 >= 0x7FFFFFF0ull) return fail(h, PO_ERR_CAPACITY, "po_layout_edges: more than 2^31 rows in one call");
    const uint32_t n_rows = (uint32_t)n_rows64;
    const uint32_t n_names = n_nodes / 2;
    L.n_rows = n_rows;
    for (hipEvent_t& e : h->ev_lay)
        if (!e) HIP_TRY(h, hipEventCreate(&e));
    PO_TRY(rows_to_device(h, rows));
    PO_TRY(ensure(h, h->d_lay_len, ((size_t)n_nodes + 1) * 4));
    PO_TRY(ensure(h, h->d_lay_cnt, 128));
    PO_TRY(ensure(h, h->d_scalars, 128));
    PO_TRY(ensure(h, h->d_rflag, (size_t)n_rows + 1));
    PO_TRY(ensure(h, h->d_removed, (size_t)n_names + 1));
    if (n_nodes) HIP_TRY(h, hipMemcpyAsync(h->d_lay_len.p, h->len.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemsetAsync(h->d_lay_cnt.p, 0, 128, st));
    HIP_TRY(h, hipMemsetAsync(h->d_removed.p, 0, (size_t)n_names + 1, st));
    unsigned long long* cnt = h->d_lay_cnt.as<unsigned long long>();
    const po::Row* d_rows = rows->d_rows.as<po::Row>();
    const uint32_t* d_len = h->d_lay_len.as<uint32_t>();
    po::LayoutParams P;
    P.min_read_length = prm.min_read_length;
    P.min_overlap_length = prm.min_overlap_length;
    P.max_overhang_abs = prm.max_overhang_abs;
    P.pad = 0;
    P.max_overhang_rel = prm.max_overhang_rel;
    const uint32_t stride_grid = std::max<uint32_t>(1u, std::min<uint32_t>(cdiv(n_rows, 256), (uint32_t)h->n_cu * 8));
    HIP_TRY(h, hipEventRecord(h->ev_lay[0], st));
    if (n_rows) {
        hipLaunchKernelGGL(po::k_layout_classify, dim3(stride_grid), dim3(256), 0, st, d_rows, n_rows, d_len, n_nodes, P,
                           h->d_rflag.as<uint8_t>(), h->d_removed.as<uint8_t>(), cnt);
        hipLaunchKernelGGL(po::k_count_bytes, dim3(std::max<uint32_t>(1u, std::min<uint32_t>(cdiv(n_names, 256), 256u))), dim3(256), 0,
                           st, h->d_removed.as<uint8_t>(), n_names, cnt + po::LC_N);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipEventRecord(h->ev_lay[1], st));
    HIP_TRY(h, hipMemcpyAsync(h->pinned + 16, cnt, (po::LC_N + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    const uint64_t* c = h->pinned + 16;
    if (c[po::LC_INVALID]) return fail(h, PO_ERR_INVALID, "po_layout_edges: a row names a read the handle does not hold");
    for (int t = 0; t < 4; ++t) L.n_type[t] = c[po::LC_TYPE0 + t];
    L.n_short = c[po::LC_SHORT];
    L.n_min_overlap = c[po::LC_MINOVL];
    L.n_overhang = c[po::LC_OVERHANG];
    L.n_pass = c[po::LC_PASS];
    L.n_contained_reads = c[po::LC_N];
    uint64_t n_edges = 0;
    if (L.n_pass && rows->unique_twins && !getenv("PHASM_LAYOUT_TABLE")) {
        // rows straight from the paired-strand emission: the last row of an adjacent (row, mirror) group owns the pair
        PO_TRY(ensure(h, h->d_ecnt, (size_t)n_rows));
        PO_TRY(ensure(h, h->d_ewin, (size_t)n_rows));
        PO_TRY(ensure(h, h->d_eoff, ((size_t)n_rows + 1) * 4));
        hipLaunchKernelGGL(po::k_layout_winner_adjacent, dim3(cdiv(n_rows, 256)), dim3(256), 0, st, d_rows, n_rows, d_len,
                           h->d_rflag.as<uint8_t>(), h->d_removed.as<uint8_t>(), h->d_ecnt.as<uint8_t>(), h->d_ewin.as<uint8_t>());
        HIP_TRY(h, hipGetLastError());
        PO_TRY(prefix_sum<uint8_t>(h, h->d_ecnt.as<uint8_t>(), n_rows, h->d_eoff.as<uint32_t>(), &h->pinned[2]));
        HIP_TRY(h, hipEventRecord(h->ev_lay[2], st));
        HIP_TRY(h, hipStreamSynchronize(st));
        n_edges = h->pinned[2];
        if (h->spare_edges.p && h->spare_edges.cap >= n_edges * sizeof(po_edge)) {
            res->d_rows = h->spare_edges;
            h->spare_edges = DevBuf();
        }
        PO_TRY(ensure(h, res->d_rows, std::max<size_t>(n_edges * sizeof(po_edge), 256), 1.0, false));
        if (n_edges) {
            hipLaunchKernelGGL(po::k_layout_emit, dim3(cdiv(n_rows, 256)), dim3(256), 0, st, d_rows, n_rows, d_len,
                               h->d_rflag.as<uint8_t>(), h->d_ewin.as<uint8_t>(), h->d_eoff.as<uint32_t>(),
                               res->d_rows.as<po::Edge>());
            HIP_TRY(h, hipGetLastError());
        }
    } else if (L.n_pass) {
        // twin pair -> last writer row: 2 slots per surviving row (load <= 1/2; 1/4 when every row has its
        // strand-mirror twin), 16-byte slots
        if (2 * L.n_pass + 64 >= 0xFFFFFF00ull) return fail(h, PO_ERR_CAPACITY, "po_layout_edges: too many edges for one call");
        const uint32_t n_slots = (uint32_t)(2 * L.n_pass + 64);
        PO_TRY(ensure(h, h->d_ekey, (size_t)n_slots * sizeof(po::EdgeSlot)));
        PO_TRY(ensure(h, h->d_ecnt, (size_t)n_rows));
        PO_TRY(ensure(h, h->d_ewin, (size_t)n_rows));
        PO_TRY(ensure(h, h->d_eoff, ((size_t)n_rows + 1) * 4));
        HIP_TRY(h, hipMemsetAsync(h->d_ekey.p, 0xFF, (size_t)n_slots * sizeof(po::EdgeSlot), st));
        hipLaunchKernelGGL(po::k_layout_insert, dim3(stride_grid), dim3(256), 0, st, d_rows, n_rows, d_len, h->d_rflag.as<uint8_t>(),
                           h->d_removed.as<uint8_t>(), h->d_ekey.as<po::EdgeSlot>(), n_slots);
        hipLaunchKernelGGL(po::k_layout_winner, dim3(cdiv(n_rows, 256)), dim3(256), 0, st, d_rows, n_rows, d_len,
                           h->d_rflag.as<uint8_t>(), h->d_removed.as<uint8_t>(), h->d_ekey.as<po::EdgeSlot>(), n_slots,
                           h->d_ecnt.as<uint8_t>(), h->d_ewin.as<uint8_t>());
        HIP_TRY(h, hipGetLastError());
        PO_TRY(prefix_sum<uint8_t>(h, h->d_ecnt.as<uint8_t>(), n_rows, h->d_eoff.as<uint32_t>(), &h->pinned[2]));
        HIP_TRY(h, hipEventRecord(h->ev_lay[2], st));
        HIP_TRY(h, hipStreamSynchronize(st));
        n_edges = h->pinned[2];
        if (h->spare_edges.p && h->spare_edges.cap >= n_edges * sizeof(po_edge)) {
            res->d_rows = h->spare_edges;
            h->spare_edges = DevBuf();
        }
        PO_TRY(ensure(h, res->d_rows, std::max<size_t>(n_edges * sizeof(po_edge), 256), 1.0, false));
        if (n_edges) {
            hipLaunchKernelGGL(po::k_layout_emit, dim3(cdiv(n_rows, 256)), dim3(256), 0, st, d_rows, n_rows, d_len,
                               h->d_rflag.as<uint8_t>(), h->d_ewin.as<uint8_t>(), h->d_eoff.as<uint32_t>(),
                               res->d_rows.as<po::Edge>());
            HIP_TRY(h, hipGetLastError());
        }
    } else {
        HIP_TRY(h, hipEventRecord(h->ev_lay[2], st));
    }
    HIP_TRY(h, hipEventRecord(h->ev_lay[3], st));
    if (removed_out && n_names) {
        PO_TRY(ensure_host(h, h->scratch_host, n_names));
        HIP_TRY(h, hipMemcpyAsync(h->scratch_host.p, h->d_removed.p, n_names, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    if (removed_out && n_names) std::memcpy(removed_out, h->scratch_host.p, n_names);
    res->count = n_edges;
    L.n_edges = n_edges;
    (void)hipEventElapsedTime(&L.ms_classify, h->ev_lay[0], h->ev_lay[1]);
    (void)hipEventElapsedTime(&L.ms_dedupe, h->ev_lay[1], h->ev_lay[2]);
    (void)hipEventElapsedTime(&L.ms_emit, h->ev_lay[2], h->ev_lay[3]);
    (void)hipEventElapsedTime(&L.ms_total, h->ev_lay[0], h->ev_lay[3]);
    return PO_OK;
}

// ---- GFA2 reader for `phasm layout` (S and E lines) ---------------------------------------------

struct Field {
    const char* p;
    size_t n;
};

// Python's line.strip().split('\t'), then .strip() of a field
inline void strip(const char*& p, size_t& n) {
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
    while (n && ws(p[0])) ++p, --n;
    while (n && ws(p[n - 1])) --n;
}

size_t split_tabs(const char* p, size_t n, Field* out, size_t max_fields) {
    strip(p, n);
    size_t k = 0;
    const char* f0 = p;
    for (size_t i = 0; i <= n; ++i) {
        if (i == n || p[i] == '\t') {
            if (k < max_fields) out[k] = Field{f0, (size_t)(p + i - f0)};
            ++k;
            f0 = p + i + 1;
        }
    }
    return k;
}

// int(text) for a decimal field (optional sign, surrounding blanks); _gfa_pos_to_int (gfa.py:65-69) drops one
// trailing '$' first
bool parse_int(Field f, bool allow_dollar, long long& out) {
    const char* p = f.p;
    size_t n = f.n;
    if (allow_dollar && n && p[n - 1] == '$') --n;
    strip(p, n);
    if (!n) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') {
        neg = *p == '-';
        ++p, --n;
    }
    if (!n || n > 18) return false;
    long long v = 0;
    for (size_t i = 0; i < n; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (p[i] - '0');
    }
    out = neg ? -v : v;
    return true;
}

// name -> read index (the reference keeps a dict, gfa.py:107-109)
struct NameIndex {
    std::vector<uint32_t> slot;  // read index + 1, 0 = empty
    uint32_t mask = 0;
    static uint64_t hash(const char* p, size_t n) {
        uint64_t x = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) x = (x ^ (unsigned char)p[i]) * 1099511628211ull;
        return x ^ (x >> 29);
    }
    void build(const po_handle* h) {
        const size_t n_names = h->ids.size() / 2;
        size_t cap = 16;
        while (cap < 2 * n_names + 2) cap <<= 1;
        slot.assign(cap, 0);
        mask = (uint32_t)(cap - 1);
        for (size_t i = 0; i < n_names; ++i) {
            const std::string& id = h->ids[2 * i];
            uint32_t s = (uint32_t)hash(id.data(), id.size() - 1) & mask;
            while (slot[s]) s = (s + 1) & mask;
            slot[s] = (uint32_t)i + 1;
        }
    }
    long find(const po_handle* h, const char* p, size_t n) const {
        uint32_t s = (uint32_t)hash(p, n) & mask;
        while (slot[s]) {
            const std::string& id = h->ids[2 * (size_t)(slot[s] - 1)];
            if (id.size() - 1 == n && std::memcmp(id.data(), p, n) == 0) return (long)slot[s] - 1;
            s = (s + 1) & mask;
        }
        return -1;
    }
};

po_status add_segment(po_handle* h, const char* name, size_t name_len, uint32_t length) {
    if (!h->len.empty() && !h->segments_only)
        return fail(h, PO_ERR_INVALID, "this handle already holds sequences; segments need a handle of their own");
    if (h->len.size() >= 0xFFFFFFF0ull - 2) return fail(h, PO_ERR_CAPACITY, "too many reads");
    h->segments_only = true;
    std::string id(name ? name : "", name_len);
    id.push_back('+');
    h->ids.push_back(id);
    id.back() = '-';
    h->ids.push_back(id);
    h->len.push_back(length);
    h->len.push_back(length);
    h->total_bases += 2ull * length;
    h->ids_paired = -1;
    return PO_OK;
}


// ---- parallel FASTA ingest (pure-ACGT files, the common case) -------------------------------------
// po_add_fasta's sequential path parses, reverse-complements and packs one record after the other: 0.9 s for the
// 715 MB of config 2, the longest part of the `overlap` command.  Here one pass over the mapped file finds the
// records and their lengths, the packed store is sized once, and a few threads gather / reverse-complement / pack
// the records into their final places.  Any byte outside upper-case ACGT (exception records, pairing checks), an
// 8-bit handle or a handle that already holds exception records makes it step aside for the sequential path:
// returns false with the handle untouched.
struct FastaRec {
    const char* name;
    size_t name_len;
    const char* seq_begin;  // first byte after the header line
    const char* seq_end;    // start of the next header (or end of file)
    size_t seq_len;         // bases once lines are joined and blanks trimmed
};

// what handle_line does to a sequence line: cut \r\n, then blanks at both ends
inline void trim_seq_line(const char*& p, size_t& n) {
    while (n && (p[n - 1] == '\r' || p[n - 1] == '\n')) --n;
    while (n && (p[n - 1] == ' ' || p[n - 1] == '\t')) --n;
    while (n && (p[0] == ' ' || p[0] == '\t')) ++p, --n;
}

bool add_fasta_parallel(po_handle* h, const char* data, size_t size, uint64_t* n_records) {
    if (h->bits != 2 || !h->exc_pos.empty()) return false;
    std::vector<FastaRec> recs;
    {   // pass A: records, in file order.  The file is cut into ranges; a range's thread takes the records whose header
        // line STARTS inside the range and follows its last record to the next header (or the end of the file) -- one
        // thread over the 715 MB of config 2 was 0.15 s of the command's 0.19 s of ingest
        // (threads: the process's CPU share less the two that bring the device up and register the stores beside this)
        const unsigned hw = std::max(1u, cpu_share() > 3 ? cpu_share() - 2 : cpu_share());
        unsigned n_rng = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(hw, 16u), size >> 22));
        if (const char* e = getenv("PHASM_FASTA_RANGES")) n_rng = (unsigned)std::max(1, std::min(64, atoi(e)));   // (tests: many ranges on small files)
        if ((size_t)n_rng > size) n_rng = 1;
        std::vector<std::vector<FastaRec>> part(n_rng);
        auto scan = [&](unsigned k) {
            const size_t lo = size / n_rng * k, hi = k + 1 == n_rng ? size : size / n_rng * (k + 1);
            size_t pos = lo;
            if (lo) {   // first line start at or after lo
                const char* nl = static_cast<const char*>(std::memchr(data + lo - 1, '\n', size - (lo - 1)));
                pos = nl ? (size_t)(nl - data) + 1 : size;
            }
            std::vector<FastaRec>& out = part[k];
            bool have = false;
            while (pos < size) {
                const char* nl = static_cast<const char*>(std::memchr(data + pos, '\n', size - pos));
                const size_t len = nl ? (size_t)(nl - (data + pos)) + 1 : size - pos;
                const char* p = data + pos;
                size_t n = len;
                while (n && (p[n - 1] == '\r' || p[n - 1] == '\n')) --n;
                if (n && p[0] == '>') {
                    if (have) out.back().seq_end = data + pos;
                    if (pos >= hi) return;   // the next range's first record
                    out.push_back(FastaRec{p + 1, n - 1, data + pos + len, data + size, 0});
                    have = true;
                } else if (n && have) {
                    size_t m = len;
                    trim_seq_line(p, m);
                    out.back().seq_len += m;
                } else if (!have && pos >= hi) {
                    return;                  // (no header started in this range)
                }
                pos += len;
            }
        };
        std::vector<std::thread> thr;
        try {
            for (unsigned k = 1; k < n_rng; ++k) thr.emplace_back(scan, k);
        } catch (const std::system_error&) {
        }
        const unsigned started = (unsigned)thr.size() + 1;
        scan(0);
        for (auto& th : thr) th.join();
        for (unsigned k = started; k < n_rng; ++k) scan(k);   // (threads that could not be started)
        size_t total = 0;
        for (const auto& v : part) total += v.size();
        recs.reserve(total);
        for (auto& v : part) recs.insert(recs.end(), v.begin(), v.end());
    }
    if (getenv("PHASM_FASTA_TRACE")) std::fprintf(stderr, "[fasta] %zu records found\n", recs.size());
    if (recs.empty()) {
        if (n_records) *n_records = 0;
        return true;
    }
    for (const FastaRec& r : recs)
        if (r.seq_len > 0x7FFFFFF0ull) return false;  // (the sequential path reports it)
    if (h->len.size() + 2 * recs.size() >= 0xFFFFFFF0ull) return false;
    // final place of every oriented read in its packed store (append_packed's layout): forward reads go to the
    // store of the next read's parity, reverse complements to the other one
    if (h->len.size() & 1) return false;  // (an odd number of reads so far: pairs would straddle the stores; sequential path)
    const size_t old_words[2] = {h->words[0].size(), h->words[1].size()};
    std::vector<size_t> off(2 * recs.size());
    size_t cur[2] = {old_words[0], old_words[1]};
    for (size_t i = 0; i < recs.size(); ++i) {
        const size_t nw = (recs[i].seq_len + 31) / 32;
        for (int k = 0; k < 2; ++k) {
            const size_t o = (cur[k] + 1) & ~size_t(1);
            off[2 * i + k] = o;
            cur[k] = o + nw + 1;
        }
    }
    const bool ftrace = getenv("PHASM_FASTA_TRACE") != nullptr;
    const auto ft0 = std::chrono::steady_clock::now();
    auto fmark = [&](const char* what) {
        if (ftrace) std::fprintf(stderr, "[fasta] %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ft0).count());
    };
    // the stores grow WITHOUT being zero-filled and without being page-locked here: the packing threads below write every
    // word (first touch in parallel), and the registration runs beside them on a thread of its own (RegAlloc::reg_now)
    g_reg_defer = true;
    try {
        h->words[0].resize(cur[0]);
        h->words[1].resize(cur[1]);
    } catch (...) {
        g_reg_defer = false;
        throw;
    }
    g_reg_defer = false;
    fmark("stores sized");
    uint64_t* words[2] = {h->words[0].data(), h->words[1].data()};
    std::thread reg_thread;
    try {
        uint64_t* blocks[2] = {h->words[0].capacity() ? words[0] : nullptr, h->words[1].capacity() ? words[1] : nullptr};
        reg_thread = std::thread([blocks, &fmark] {
            for (int k = 0; k < 2; ++k)
                if (blocks[k]) RegAlloc<uint64_t>::reg_now(blocks[k]);
            fmark("stores registered");
        });
    } catch (const std::system_error&) {
    }
    struct RegJoin {
        std::thread& t;
        po_handle* h;
        ~RegJoin() {
            if (t.joinable()) t.join();
            else   // (no thread: register here)
                for (int k = 0; k < 2; ++k)
                    if (h->words[k].capacity()) RegAlloc<uint64_t>::reg_now(h->words[k].data());
        }
    } reg_join{reg_thread, h};
    const uint8_t* lut = g_lut.v;
    const unsigned char* comp = comp_table();
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto pack = [&](const unsigned char* sq, size_t n, uint64_t* w) -> uint8_t {
        uint8_t b = 0;
        const size_t full = n / 32;
        for (size_t k = 0; k < full; ++k) {
            const unsigned char* q = sq + k * 32;
            uint64_t acc = 0;
            for (int j = 0; j < 32; ++j) {
                const uint8_t c = lut[q[j]];
                b |= c;
                acc |= (uint64_t)(c & 3) << (2 * j);
            }
            w[k] = acc;
        }
        if (full * 32 < n) {
            uint64_t acc = 0;
            for (size_t i = full * 32; i < n; ++i) {
                const uint8_t c = lut[sq[i]];
                b |= c;
                acc |= (uint64_t)(c & 3) << (2 * (i & 31));
            }
            w[full] = acc;
        }
        return b;
    };
    auto worker = [&]() {
      try {
        std::string seq, rc;
        for (;;) {
            const size_t lo = next.fetch_add(64);
            if (lo >= recs.size() || bad.load(std::memory_order_relaxed)) return;
            const size_t hi = std::min(recs.size(), lo + 64);
            for (size_t i = lo; i < hi; ++i) {
                const FastaRec& r = recs[i];
                seq.clear();
                seq.reserve(r.seq_len);
                for (const char* p = r.seq_begin; p < r.seq_end;) {
                    const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(r.seq_end - p)));
                    size_t n = nl ? (size_t)(nl - p) + 1 : (size_t)(r.seq_end - p);
                    const char* q = p;
                    size_t m = n;
                    trim_seq_line(q, m);
                    seq.append(q, m);
                    p += n;
                }
                const size_t n = seq.size();
                if (n != r.seq_len) {  // (pass A sized the stores with seq_len: never pack anything else into them)
                    bad.store(1);
                    return;
                }
                rc.resize(n);
                for (size_t k = 0; k < n; ++k) rc[k] = (char)comp[(unsigned char)seq[n - 1 - k]];
                // (the words between the reads: the alignment gap in front, if there is one, and the spare word behind)
                const size_t nw = (n + 31) / 32;
                for (int k = 0; k < 2; ++k) {
                    const size_t prev_end = i ? off[2 * (i - 1) + k] + (recs[i - 1].seq_len + 31) / 32 + 1 : old_words[k];
                    if (off[2 * i + k] > prev_end) words[k][off[2 * i + k] - 1] = 0;
                    words[k][off[2 * i + k] + nw] = 0;
                }
                uint8_t b = pack(reinterpret_cast<const unsigned char*>(seq.data()), n, words[0] + off[2 * i]);
                b |= pack(reinterpret_cast<const unsigned char*>(rc.data()), n, words[1] + off[2 * i + 1]);
                if (n != r.seq_len || (b & 0x80)) {
                    bad.store(1);
                    return;
                }
            }
        }
      } catch (...) {
        bad.store(1);  // (out of memory in a worker: the sequential path will report it)
      }
    };
    {
        const unsigned hw = std::max(1u, cpu_share() > 3 ? cpu_share() - 2 : cpu_share());
        const unsigned n_thr = std::max(1u, std::min(hw, 16u));
        std::vector<std::thread> thr;
        try {
            for (unsigned t = 1; t < n_thr; ++t) thr.emplace_back(worker);
        } catch (const std::system_error&) {
            // fewer threads than asked for: carry on with those that started
        }
        worker();
        for (auto& th : thr) th.join();
    }
    fmark("packed");
    if (bad.load()) {  // something other than upper-case ACGT: let the sequential path deal with it
        h->words[0].resize(old_words[0]);
        h->words[1].resize(old_words[1]);
        return false;
    }
    h->ids.reserve(h->ids.size() + 2 * recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        std::string id(recs[i].name, recs[i].name_len);
        id.push_back('+');
        h->ids.push_back(id);
        id.back() = '-';
        h->ids.push_back(std::move(id));
        for (int k = 0; k < 2; ++k) {
            h->len.push_back((uint32_t)recs[i].seq_len);
            h->woff.push_back(off[2 * i + k]);
            h->exc_off.push_back((uint32_t)h->exc_pos.size());
        }
        h->total_bases += 2 * recs[i].seq_len;
    }
    h->dirty = true;
    h->ids_paired = -1;
    if (n_records) *n_records = recs.size();
    fmark("ids and tables appended");
    return true;
}

}  // namespace

extern "C" {

int po_abi_version(void) { return PO_ABI_VERSION; }

po_status po_create(po_handle** out) {
    if (!out) return PO_ERR_INVALID;
    po_handle* h = new (std::nothrow) po_handle();
    if (!h) return PO_ERR_NOMEM;
    *out = h;
    return PO_OK;
}

void po_destroy(po_handle* h) {
    if (!h) return;
    if (h->dev_ready) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        DevBuf* bufs[] = {&h->d_words, &h->d_woff, &h->d_len, &h->d_tiles, &h->d_read_tile0, &h->d_left, &h->d_left_cnt, &h->d_tile_extra, &h->d_exc_off, &h->d_exc_pos, &h->d_exc_byte, &h->d_pair_state, &h->d_truemask,
                          &h->d_table, &h->d_slot_cnt, &h->d_slot_cur, &h->d_slot_start, &h->d_read_slot, &h->d_chain,
                          &h->d_chain_tmp, &h->d_long_list, &h->d_entry_off, &h->d_bloom, &h->d_selfrep, &h->d_tile_count, &h->d_tile_off,
                          &h->d_ps_blocks, &h->d_scalars, &h->d_cand_a, &h->d_cand_p, &h->d_cand_b, &h->d_type,
                          &h->d_rowcnt, &h->d_row_off, &h->d_flag, &h->d_pair_key, &h->d_pair_min, &h->spare_rows, &h->spare_cands, &h->spare_edges,
                          &h->d_vlabel, &h->d_vrank, &h->d_vperm, &h->d_end_a, &h->d_end_b, &h->d_dpcnt, &h->d_lay_len, &h->d_lay_cnt, &h->d_rflag, &h->d_removed, &h->d_ekey, &h->d_ecnt,
                          &h->d_ewin, &h->d_eoff, &h->d_chain_state, &h->d_tail_state};
        // every stream idle before anything the device (or a copy) may still touch is given back
        if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
        if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
        if (h->rc_stream) (void)hipStreamSynchronize(h->rc_stream);
        if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
        for (DevBuf* b : bufs) b->release();
        const bool pooled = kit_give(h);
        if (!pooled) {
        for (int i = 0; i < 2 * EV_N; ++i) (void)hipEventDestroy(h->ev_sets[i / EV_N][i % EV_N]);
        for (hipEvent_t e : h->ev_lay)
            if (e) (void)hipEventDestroy(e);
        (void)hipEventDestroy(h->ev_up0);
        (void)hipEventDestroy(h->ev_up1);
        if (h->pinned) {
            pin_drop(h->pinned);
            (void)hipHostFree(h->pinned);
        }
        }
        h->spare_host.release();
        h->scratch_host.release();
        h->home_stage.release();
        for (DevBuf& b : h->chunk_rows) b.release();
        h->d_first.release();
        h->d_defer.release();
        h->meta_host.release();
        h->stage_host.release();
        for (void* c : h->arena_chunks) (void)hipFree(c);
        h->arena_chunks.clear();
        for (hipEvent_t& e : h->ev_gate)   // (the handle's own, never part of the pooled kit)
            if (e) {
                (void)hipEventDestroy(e);
                e = nullptr;
            }
        if (!pooled) {
        if (h->up_stream) {
            (void)hipStreamSynchronize(h->up_stream);
            (void)hipStreamDestroy(h->up_stream);
        }
        for (hipEvent_t e : h->ev_piece)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->ev_rc)
            if (e) (void)hipEventDestroy(e);
        if (h->ev_meta) (void)hipEventDestroy(h->ev_meta);
        if (h->rc_stream) {
            (void)hipStreamSynchronize(h->rc_stream);
            if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
            (void)hipStreamDestroy(h->rc_stream);
        }
        if (h->ev_first) (void)hipEventDestroy(h->ev_first);
        if (h->scan_stream) {
            (void)hipStreamSynchronize(h->scan_stream);
            (void)hipStreamDestroy(h->scan_stream);
        }
        for (hipEvent_t e : h->ev_s1)
            if (e) (void)hipEventDestroy(e);
        if (h->ev_idx) (void)hipEventDestroy(h->ev_idx);
        if (h->copy_stream) {
            (void)hipStreamSynchronize(h->copy_stream);
            (void)hipStreamDestroy(h->copy_stream);
        }
        (void)hipStreamDestroy(h->stream);
        }
    }
    h->spare_host.release();   // (the result pool may exist without the handle ever having made a call)
    h->home_stage.release();
    delete h;
}

po_status po_set_device(po_handle* h, int device) {
    if (!h) return PO_ERR_INVALID;
    if (h->dev_ready && device != h->device) return fail(h, PO_ERR_INVALID, "device already initialised");
    h->device = device;
    return PO_OK;
}

po_status po_add_sequence(po_handle* h, const char* id, size_t id_len, const char* seq, size_t seq_len) {
    if (!h || (!id && id_len) || (!seq && seq_len)) return PO_ERR_INVALID;
    if (h->segments_only) return fail(h, PO_ERR_INVALID, "this handle holds GFA segments (no sequences); use a new handle");
    if (seq_len > 0x7FFFFFF0ull) return fail(h, PO_ERR_CAPACITY, "read longer than 2^31 bases");
    if (h->len.size() >= 0xFFFFFFF0ull) return fail(h, PO_ERR_CAPACITY, "too many reads");
    quiesce_store(h);  // (the store may move)
    try {
        const unsigned char* s = reinterpret_cast<const unsigned char*>(seq);
        if (h->bits != 2 || !append_packed(h, s, seq_len, 2)) {
            if (h->bits == 2) widen_to_bytes(h);  // first non-ACGT byte: everything moves to 8 bits/base
            append_packed(h, s, seq_len, 8);
        }
        h->ids.emplace_back(id ? id : "", id_len);
        h->len.push_back((uint32_t)seq_len);
        h->total_bases += seq_len;
        h->dirty = true;
        h->ids_paired = -1;
        if (h->total_bases >= h->pool_bases && h->total_bases >= (64ull << 20)) result_pool_grow(h);
        if (h->bits == 2 && h->first_n + 1 == h->len.size()) note_first_words(h);   // (O(1): this read's two words)
        if (h->bits == 2 && (h->len.size() & 1) == 0) {
            const size_t r = h->len.size() - 1;
            if (!h->exc_pos.empty()) host_pair_check(h, r, s);
            const bool has_exc = h->exc_off[r - 1] != h->exc_off[r] || h->exc_off[r] != h->exc_off[r + 1];
            if (h->all_pairs_rcx) {
                const bool ok = has_exc ? (h->pair_state.size() > r / 2 && h->pair_state[r / 2] == 1) : packed_is_revcomp(h, r);
                h->all_pairs_rcx = ok;
                if (has_exc || !ok) h->all_pairs_rc = false;
            } else {
                h->all_pairs_rc = false;
            }
        }
    } catch (const std::bad_alloc&) {
        return fail(h, PO_ERR_NOMEM, "out of host memory in po_add_sequence");
    }
    return PO_OK;
}

// FASTA ingest for the overlap command (the reference delegates this to dinopy.FastaReader and
// dinopy.reverse_complement, assembler.py:32-40): header = the whole line after '>', sequence lines
// concatenated, blank lines skipped, bytes kept as they are; with both_strands every record is added
// as name+"+" / sequence and name+"-" / reverse complement (IUPAC-aware, case preserving).
po_status po_add_fasta(po_handle* h, const char* path, int both_strands, uint64_t* n_records) {
    if (!h || !path) return PO_ERR_INVALID;
    if (n_records) *n_records = 0;
    quiesce_store(h);  // (the store may move)
    if (both_strands && !h->segments_only && !getenv("PHASM_FASTA_SEQUENTIAL")) {
        // fast path: map the file, pack with a few threads (pure upper-case ACGT only; see add_fasta_parallel)
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(h, PO_ERR_INVALID, std::string("cannot open ") + path);
        struct stat sb;
        bool done = false;
        if (::fstat(fd, &sb) == 0 && sb.st_size > 0) {
            void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                // A file this size is headed for the GPU: the device comes up (runtime start, streams, events, the first code
                // object: ~0.15 s in a fresh process) and the page-locked result pool is made WHILE the file is parsed and
                // packed, not after it -- the `overlap` command is one process and one call (assembler.py:42).  The pool is
                // sized from the file's size (both strands: at most two oriented bases per byte).
                std::thread warm;
                if ((uint64_t)sb.st_size >= (32ull << 20) && !h->dev_ready && !getenv("PHASM_NO_POOL") && !getenv("PHASM_NO_WARM")) {
                    const uint64_t est_bases = 2ull * (uint64_t)sb.st_size;
                    try {
                        warm = std::thread([h, est_bases] {
                            if (h->total_bases + est_bases >= h->pool_bases) result_pool_grow(h, h->total_bases + est_bases);
                        });
                    } catch (const std::system_error&) {
                    }
                }
                struct Joiner {
                    std::thread& t;
                    ~Joiner() {
                        if (t.joinable()) t.join();
                    }
                } joiner{warm};
                try {
                    const auto tf0 = std::chrono::steady_clock::now();
                    done = add_fasta_parallel(h, static_cast<const char*>(m), (size_t)sb.st_size, n_records);
                    if (getenv("PHASM_FASTA_TRACE")) {
                        const double t_parse = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count();
                        if (warm.joinable()) warm.join();
                        std::fprintf(stderr, "[fasta] parsed and packed in %.1f ms; the device and the result pool were ready %.1f ms after the start\n", t_parse,
                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
                    }
                } catch (const std::bad_alloc&) {
                    ::munmap(m, (size_t)sb.st_size);
                    ::close(fd);
                    return fail(h, PO_ERR_NOMEM, "out of host memory in po_add_fasta");
                } catch (const std::system_error&) {
                    done = false;
                }
                ::munmap(m, (size_t)sb.st_size);
            }
        }
        ::close(fd);
        if (done) {
            if (h->total_bases >= h->pool_bases && h->total_bases >= (64ull << 20)) result_pool_grow(h);
            if (h->bits == 2) note_first_words(h);
            return PO_OK;
        }
    }
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(h, PO_ERR_INVALID, std::string("cannot open ") + path);
    const unsigned char* comp = comp_table();
    po_status st = PO_OK;
    uint64_t nrec = 0;
    try {
        std::string name, seq, rc, line;
        bool have = false;
        std::vector<char> buf(1 << 22);
        auto flush_record = [&]() -> po_status {
            if (!have) return PO_OK;
            ++nrec;
            if (!both_strands) return po_add_sequence(h, name.data(), name.size(), seq.data(), seq.size());
            std::string id = name + "+";
            po_status s1 = po_add_sequence(h, id.data(), id.size(), seq.data(), seq.size());
            if (s1 != PO_OK) return s1;
            rc.resize(seq.size());
            for (size_t i = 0, n = seq.size(); i < n; ++i) rc[i] = (char)comp[(unsigned char)seq[n - 1 - i]];
            id.back() = '-';
            return po_add_sequence(h, id.data(), id.size(), rc.data(), rc.size());
        };
        auto handle_line = [&](const char* p, size_t n) -> po_status {
            while (n && (p[n - 1] == '\r' || p[n - 1] == '\n')) --n;
            if (n == 0) return PO_OK;
            if (p[0] == '>') {
                po_status s1 = flush_record();
                if (s1 != PO_OK) return s1;
                name.assign(p + 1, n - 1);
                seq.clear();
                have = true;
            } else if (have) {
                while (n && (p[n - 1] == ' ' || p[n - 1] == '\t')) --n;
                size_t b = 0;
                while (b < n && (p[b] == ' ' || p[b] == '\t')) ++b;
                seq.append(p + b, n - b);
            }
            return PO_OK;
        };
        size_t got;
        while (st == PO_OK && (got = std::fread(buf.data(), 1, buf.size(), f)) > 0) {
            size_t pos = 0;
            while (pos < got && st == PO_OK) {
                const char* nl = static_cast<const char*>(std::memchr(buf.data() + pos, '\n', got - pos));
                if (!nl) {
                    line.append(buf.data() + pos, got - pos);
                    pos = got;
                } else {
                    const size_t len = (size_t)(nl - (buf.data() + pos)) + 1;
                    if (line.empty()) {
                        st = handle_line(buf.data() + pos, len);
                    } else {
                        line.append(buf.data() + pos, len);
                        st = handle_line(line.data(), line.size());
                        line.clear();
                    }
                    pos += len;
                }
            }
        }
        if (st == PO_OK && !line.empty()) st = handle_line(line.data(), line.size());
        if (st == PO_OK) st = flush_record();
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_add_fasta");
    }
    std::fclose(f);
    if (n_records) *n_records = nrec;
    return st;
}

uint32_t po_num_sequences(const po_handle* h) { return h ? (uint32_t)h->len.size() : 0; }

po_status po_get_id(const po_handle* h, uint32_t idx, const char** id, size_t* id_len) {
    if (!h || idx >= h->ids.size() || !id || !id_len) return PO_ERR_INVALID;
    *id = h->ids[idx].data();
    *id_len = h->ids[idx].size();
    return PO_OK;
}

uint32_t po_get_length(const po_handle* h, uint32_t idx) { return (h && idx < h->len.size()) ? h->len[idx] : 0; }

po_status po_upload(po_handle* h) {
    if (!h) return PO_ERR_INVALID;
    po_status st;
    try {
        st = upload(h);
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_upload");
    }
    // (a failed upload may have copies from the host store in flight: nothing returns before they are done)
    if (st != PO_OK && h->dev_ready) (void)hipStreamSynchronize(h->stream);
    return st;
}

// ---- sharded upload (multi-GPU): each rank brings 1/N of the packed reads over ITS PCIe link, xGMI does the rest ----
po_status po_upload_piece_part(po_handle* h, uint32_t shard, uint32_t nshards, uint32_t part, uint32_t nparts, void* dst_device,
                               uint64_t capacity_words, uint64_t* word_count, int* ok) {
    if (!h || !word_count || !ok || nshards == 0 || shard >= nshards || nparts == 0 || part >= nparts || nparts > 64) return PO_ERR_INVALID;
    *ok = 0;
    *word_count = 0;
    const uint32_t n = (uint32_t)h->len.size();
    // only when store 1 can be rebuilt on the device (every odd read the reverse complement of its even partner)
    if (!(h->bits == 2 && n >= 2 && (n % 2) == 0 && h->all_pairs_rc && h->exc_pos.empty()) || h->segments_only) return PO_OK;
    PO_TRY(init_device(h));
    uint64_t wb = 0, wc = 0;
    try {
        store0_part_range(h, shard, nshards, part, nparts, &wb, &wc);
    } catch (const std::bad_alloc&) {
        return fail(h, PO_ERR_NOMEM, "out of host memory");
    }
    *word_count = wc;
    *ok = 1;
    if (!dst_device) return PO_OK;   // (a query: how many words is this part of the shard's piece?)
    if (wc > capacity_words) return fail(h, PO_ERR_INVALID, "po_upload_piece: the piece does not fit the destination");
    if (wc) {
        // (a store below the registration size is ordinary heap memory: it goes through the page-locked staging block like
        // every other small source -- handed a pageable pointer the runtime would pin the heap range around it)
        const void* src = h->words[0].data() + wb;
        if (!store_is_pinned(h->words[0]) && wc * 8 <= STAGE_MAX) {
            h->stage_used = 0;   // (no copy is in flight: every path that queued one has synchronised)
            PO_TRY(ensure_host(h, h->stage_host, wc * 8 + 64));
            src = staged(h, src, wc * 8);
        }
        HIP_TRY(h, hipMemcpyAsync(dst_device, src, wc * 8, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->upload_bytes = (part == 0 ? 0 : h->upload_bytes) + wc * 8;
    return PO_OK;
}

po_status po_upload_piece(po_handle* h, uint32_t shard, uint32_t nshards, void* dst_device, uint64_t capacity_words,
                          uint64_t* word_count, int* ok) {
    return po_upload_piece_part(h, shard, nshards, 0, 1, dst_device, capacity_words, word_count, ok);
}

po_status po_upload_assemble_parts(po_handle* h, const void* pieces_device, uint64_t slot_words, uint32_t nshards, uint32_t nparts) {
    if (!h || !pieces_device || nshards == 0 || nparts == 0 || nparts > 64) return PO_ERR_INVALID;
    h->asm_pieces = static_cast<const uint64_t*>(pieces_device);
    h->asm_slot_words = slot_words;
    h->asm_n = nshards;
    h->asm_parts = nparts;
    h->dirty = true;
    const uint64_t piece_bytes = h->upload_bytes;
    po_status st;
    try {
        st = upload(h);
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_upload_assemble");
    }
    h->asm_pieces = nullptr;
    h->asm_parts = 1;
    if (st != PO_OK && h->dev_ready) (void)hipStreamSynchronize(h->stream);
    if (st == PO_OK) {
        h->upload_bytes += piece_bytes;   // (what THIS rank moved over PCIe: its piece + the per-read tables)
        h->stats.upload_bytes = h->upload_bytes;
    }
    return st;
}

po_status po_upload_assemble(po_handle* h, const void* pieces_device, uint64_t slot_words, uint32_t nshards) {
    return po_upload_assemble_parts(h, pieces_device, slot_words, nshards, 1);
}

po_status po_invalidate(po_handle* h) {
    if (!h) return PO_ERR_INVALID;
    h->dirty = true;
    return PO_OK;
}

static po_status overlaps_common(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, bool want_cands,
                                 po_result** out, void* ext_dst = nullptr, uint64_t ext_cap = 0) {
    if (!h || !out) return PO_ERR_INVALID;
    *out = nullptr;
    if (nshards == 0 || shard >= nshards) return fail(h, PO_ERR_INVALID, "shard must be < nshards");
    if (h->segments_only) return fail(h, PO_ERR_INVALID, "this handle holds GFA segments without sequences: nothing to overlap");
    po_result* r = new (std::nothrow) po_result();
    if (!r) return fail(h, PO_ERR_NOMEM, "out of host memory");
    r->h = h;
    r->ext_dst = ext_dst;
    r->ext_cap = ext_cap;
    po_status st;
    try {
        st = upload(h);
        if (st == PO_OK) st = h->bits == 2 ? run_overlaps<2>(h, min_length, shard, nshards, want_cands, r)
                                           : run_overlaps<8>(h, min_length, shard, nshards, want_cands, r);
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_overlaps");
    }
    if (st != PO_OK) {
        if (h->dev_ready) (void)hipStreamSynchronize(h->stream);
        r->d_rows.release();
        delete r;
        return st;
    }
    ++h->live_results;
    *out = r;
    return PO_OK;
}

po_status po_overlaps_shard(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, po_result** out) {
    return overlaps_common(h, min_length, shard, nshards, false, out);
}

po_status po_candidates_shard(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, po_result** out) {
    return overlaps_common(h, min_length, shard, nshards, true, out);
}

po_status po_candidates_shard_into(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, void* dst_device,
                                   uint64_t capacity, int* written, po_result** out) {
    if (written) *written = 0;
    if (!dst_device && capacity) return PO_ERR_INVALID;
    const po_status st = overlaps_common(h, min_length, shard, nshards, true, out, dst_device, capacity);
    if (st == PO_OK && written) *written = (*out)->wrote_ext ? 1 : 0;
    return st;
}

po_status po_expand(po_handle* h, const void* candidates_device, uint64_t n_candidates, po_result** out) {
    if (!h || !out || (!candidates_device && n_candidates)) return PO_ERR_INVALID;
    *out = nullptr;
    if (h->segments_only) return fail(h, PO_ERR_INVALID, "this handle holds GFA segments without sequences");
    po_result* r = new (std::nothrow) po_result();
    if (!r) return fail(h, PO_ERR_NOMEM, "out of host memory");
    r->h = h;
    po_status st;
    try {
        st = upload(h);
        if (st == PO_OK) st = run_expand(h, candidates_device, n_candidates, r);
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_expand");
    }
    if (st != PO_OK) {
        if (h->dev_ready) (void)hipStreamSynchronize(h->stream);
        r->d_rows.release();
        delete r;
        return st;
    }
    ++h->live_results;
    *out = r;
    return PO_OK;
}

po_status po_overlaps(po_handle* h, uint32_t min_length, po_result** out) {
    return po_overlaps_shard(h, min_length, 0, 1, out);
}

// po_overlaps + po_result_rows in one call, pipelined.  Two forms:
//
// * the reads are on the device already: the a-side reads are cut into chunks (the shards of po_overlaps_shard), and
//   while chunk k + 1 goes through the kernels on the handle's stream, chunk k's rows travel device -> host on a
//   second stream into ONE page-locked array;
// * the read set changed since the last upload (what the reference's overlaps() faces every time: the reads are host
//   memory, overlapper.cpp:22-36): the STREAMED step.  The packed reads cross PCIe in pieces, in index order, on a
//   third stream; as soon as piece k is there, its reads are scanned against the whole index (built from every read's
//   first word, which travels ahead: 8 bytes per read), and the pairs whose b-side read has arrived are verified
//   and emitted -- the reversed index order of kernels.hip.h (mirror_rank, mode 3) picks the member of every
//   strand-mirror pair whose b lies at or below its a, so every suffix-prefix candidate of piece k is of that kind --
//   and piece k's rows travel back while piece k + 1 is still coming in: PCIe carries both directions at once.
//   Containments of a read that has not arrived are the only candidates that must wait (the deferred list).
//
// Same rows as po_overlaps as a multiset (which member of a strand-mirror pair is computed differs, the emitted
// pair of rows does not), a-major chunk by chunk.  The result holds the host array only.
extern "C++" {
namespace {

struct HostRows {
    HostBuf hb;
    uint64_t total = 0;
};

// ---- the helper threads of a po_overlaps_to_host call (namespace home) ----
bool home_begin(po_handle* h) {
    h->home_on = false;
    if (!home_enabled()) return false;
    home::Pool* P = home::pool();
    if (!P) return false;
    P->call_mu.lock();
    home::begin(P, h->len.data(), (uint32_t)h->len.size(), getenv("PHASM_STREAM_TRACE") != nullptr || getenv("PHASM_HOME_TRACE") != nullptr);
    h->home_on = true;
    h->home_used = 0;
    h->home_seq = 0;
    // a and b in ceil(log2(reads)) bits each, p in ceil(log2(longest read + 1)), type in the top two: one 64-bit word when
    // that fits (100 k reads of 15 kb: 17 + 17 + 14; 2 M reads of 12 kb: 21 + 21 + 14), the 16-byte record otherwise
    if (h->home_pack_n != h->len.size() || h->home_pack_bases != h->total_bases) {
        uint32_t longest = 0;
        for (uint32_t l : h->len) longest = std::max(longest, l);
        uint32_t br = 1, bp = 1;
        while ((1ull << br) < (uint64_t)h->len.size()) ++br;
        while ((1ull << bp) < (uint64_t)longest + 1ull) ++bp;
        const bool fits = 2 * br + bp <= 62;
        h->home_pack_b = fits ? br : 0u;
        h->home_pack_p = fits ? 2 * br : 0u;
        h->home_pack_n = h->len.size();
        h->home_pack_bases = h->total_bases;
    }
    const char* pk = getenv("PHASM_HOME_PACK");
    const bool pack = !(pk && atoi(pk) == 0);
    h->home_sh_b = pack ? h->home_pack_b : 0u;
    h->home_sh_p = pack ? h->home_pack_p : 0u;
    return true;
}

// every piece submitted so far has been expanded (or the pool has stopped on an error)
void home_wait(po_handle* h) {
    if (h->home_on) home::wait_all(home::g_pool);
}

// returns the pool's error code (0 = fine)
int home_end(po_handle* h, po_stats* sum = nullptr) {
    if (!h->home_on) return 0;
    home::Pool* P = home::g_pool;
    home::wait_all(P);
    const int err = P->error.load();
    if (sum) {   // (the byte counters of the pieces that went home as records)
        sum->sum_overlap_bases += P->sum_l.load();
        sum->verify_bytes_algo += P->sum_b.load();
        sum->verify_bytes_exec += P->sum_e.load();
    }
    if (P->tracing)
        for (size_t i = 0; i < P->trace.size(); ++i)
            std::fprintf(stderr, "[home] piece %zu: records home at %.3f ms, %.0f rows written at %.3f ms (since the call set the pool up)\n", i,
                         P->trace[i][0] * 1e-3, P->trace[i][2], P->trace[i][1] * 1e-3);
    h->home_on = false;
    P->call_mu.unlock();
    return err;
}

// room for `nk` more rows in the page-locked array (first call, or more rows than last time: guess the whole from what has
// been seen, move what is there).  seen_share (streamed step): the share of all (a, b) index pairs the pieces so far
// cover -- rows grow with the SQUARE of the reads that have arrived, a linear guess from the first piece is 6 x short
po_status rows_room(po_handle* h, HostRows& R, uint64_t nk, uint32_t k, uint32_t n_chunks, double seen_share) {
    const size_t need = (size_t)(R.total + nk) * sizeof(po_row);
    if (need <= R.hb.cap) return PO_OK;
    uint64_t guess = std::max<uint64_t>(h->last_host_rows, (R.total + nk) * n_chunks / (k + 1));
    if (seen_share > 0.0) guess = std::max<uint64_t>(guess, (uint64_t)((double)(R.total + nk) / seen_share * 1.05));
    HostBuf bigger;
    PO_TRY(ensure_host(h, bigger, std::max<size_t>(need, (size_t)(guess + guess / 8) * sizeof(po_row))));
    home_wait(h);   // (helper threads may be writing rows into the old array)
    if (hipStreamSynchronize(h->copy_stream) != hipSuccess) {
        bigger.release();
        return fail(h, PO_ERR_HIP, "copy stream");
    }
    if (R.total) std::memcpy(bigger.p, R.hb.p, (size_t)R.total * sizeof(po_row));
    R.hb.release();
    R.hb = bigger;
    return PO_OK;
}

// the rows of one chunk as RECORDS: their copy on the copy stream into the staging block, an event behind it, and the piece
// goes to the helper threads, which write its nk rows at R.total
po_status append_home(po_handle* h, HostRows& R, const DevBuf& dev, uint64_t n_rec, uint64_t nk, uint32_t k, uint32_t n_chunks,
                      double seen_share = 0.0) {
    if (nk == 0) return PO_OK;
    home::Pool* P = home::g_pool;
    PO_TRY(rows_room(h, R, nk, k, n_chunks, seen_share));
    const size_t elem = h->home_sh_b ? sizeof(uint64_t) : sizeof(po::Cand);
    const size_t bytes = (size_t)n_rec * elem;
    constexpr uint64_t N_EV = 32;   // flag words: slots 96 .. 127 of the landing zone
    if (h->home_used + bytes > h->home_stage.cap || (h->home_seq && h->home_seq % N_EV == 0)) {
        // the block is full (or every flag word has been used once): wait for the helper threads, start over at its beginning
        home_wait(h);
        if (hipStreamSynchronize(h->copy_stream) != hipSuccess) return fail(h, PO_ERR_HIP, "copy stream");
        h->home_used = 0;
        if (bytes > h->home_stage.cap)
            PO_TRY(ensure_host(h, h->home_stage, std::max<size_t>({bytes * 2, (size_t)(h->home_last_bytes + h->home_last_bytes / 8), (size_t)8 << 20})));
    }
    // behind a copy, on the same stream: a one-thread kernel writes a number into a page-locked word of the handle's landing
    // zone (slots 96 ..) -- what the pool's first thread polls.  The LAST piece of a call can travel in parts, so that its
    // first rows are being written while its last records are still on the wire.
    uint32_t n_parts = 1;
    // (measured, round 4: interleaved A/B at config 2, 4.68 ms with the split against 4.64 without -- the three extra copies
    // and their words cost what the overlap gains; PHASM_HOME_SPLIT=n asks for n parts)
    if (const char* e = getenv("PHASM_HOME_SPLIT"))
        if (k + 1 == n_chunks && n_rec >= 65536) n_parts = (uint32_t)std::max(1, std::min(8, atoi(e)));
    if (h->home_seq % N_EV + n_parts > N_EV) n_parts = 1;   // (not enough unused flag words left in this round)
    P->paired = (h->bits == 2 && h->paired) ? 1u : 0u;
    P->bits = (uint32_t)h->bits;
    char* dst0 = static_cast<char*>(h->home_stage.p) + h->home_used;
    for (uint32_t part = 0; part < n_parts; ++part) {
        const uint64_t lo = n_rec * part / n_parts, hi = n_rec * (part + 1) / n_parts;
        const uint32_t slot = (uint32_t)(h->home_seq % N_EV);
        const uint32_t want = ++h->home_gen;
        char* dst = dst0 + lo * elem;
        // (Measured and not kept, round 4 -- profiles/r04_copy_kernels.txt.  Under the tracer the counting pass of the NEXT
        // piece shows 200-230 us instead of 90 while this copy is in flight (the runtime copies with a kernel of its own);
        // the HIP events of an untraced step do not (91 us per piece), and tools/copy_beside_kernel.py finds x 1.04-1.08.
        // A copy kernel of ours writing the page-locked block with 1 .. 256 workgroups: 5.1, 5.0, 5.1, 5.3, 5.4, 5.5 ms per step
        // against 4.7 -- the fewer waves the better, and the runtime's copy better than all of them; the same for the upload,
        // 6.8-7.8 ms.  Copying 50 % / 10 % of the bytes (timing only, the host reading the previous step's identical records):
        // 4.57 / 4.49 ms -- all of the interference is worth 0.2 ms, 8-byte records would buy 0.13.)
        HIP_TRY(h, hipMemcpyAsync(dst, static_cast<const char*>(dev.p) + lo * elem, (hi - lo) * elem, hipMemcpyDeviceToHost, h->copy_stream));
        hipLaunchKernelGGL(po::k_fill_u32, dim3(1), dim3(64), 0, h->copy_stream, reinterpret_cast<uint32_t*>(h->pinned_dev + 96 + slot), (uint64_t)1, want);
        HIP_TRY(h, hipGetLastError());
        home::Job j;
        j.rec = dst;
        j.sh_b = h->home_sh_b;
        j.sh_p = h->home_sh_p;
        j.n_rec = hi - lo;
        j.out = static_cast<po_row*>(R.hb.p) + R.total;
        j.n_rows = nk;
        j.flag = reinterpret_cast<const volatile uint32_t*>(h->pinned + 96 + slot);
        j.want = want;
        j.cont = part > 0;
        j.more = part + 1 < n_parts;
        home::submit(P, j);
        ++h->home_seq;
    }
    h->home_used += (bytes + 255) & ~size_t(255);
    h->home_last_bytes += bytes;
    R.total += nk;
    return PO_OK;
}

// rows of one chunk: room in the page-locked array, then the copy on the copy stream (dev must stay untouched
// until that stream has been synchronised)
po_status append_rows(po_handle* h, HostRows& R, const DevBuf& dev, uint64_t nk, uint32_t k, uint32_t n_chunks, double seen_share = 0.0) {
    if (nk == 0) return PO_OK;
    PO_TRY(rows_room(h, R, nk, k, n_chunks, seen_share));
    HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(R.hb.p) + (size_t)R.total * sizeof(po_row), dev.p, (size_t)nk * sizeof(po_row),
                              hipMemcpyDeviceToHost, h->copy_stream));
    R.total += nk;
    return PO_OK;
}

void add_stats(po_stats& sum, const po_stats& S) {
    sum.n_candidates += S.n_candidates;
    sum.n_verified += S.n_verified;
    sum.n_rows += S.n_rows;
    sum.sum_overlap_bases += S.sum_overlap_bases;
    sum.verify_bytes_algo += S.verify_bytes_algo;
    sum.verify_bytes_exec += S.verify_bytes_exec;
    sum.n_tiles += S.n_tiles;
    sum.shard_bases += S.shard_bases;
    sum.fused_tail += S.fused_tail;
    sum.tail_fallback += S.tail_fallback;
    sum.n_predicted += S.n_predicted;
    sum.ms_index += S.ms_index;
    sum.ms_scan_count += S.ms_scan_count;
    sum.ms_scan_fill += S.ms_scan_fill;
    sum.ms_verify += S.ms_verify;
    sum.ms_select += S.ms_select;
    sum.ms_emit += S.ms_emit;
    sum.ms_total += S.ms_total;
    sum.ms_scan_probe += S.ms_scan_probe;
    sum.ms_verify_kernel += S.ms_verify_kernel;
}

// chunk k of a call emits into its own device buffer (kept on the handle): it must outlive its copy
po_status run_chunk(po_handle* h, uint32_t min_length, uint32_t k, uint32_t n_chunks, uint64_t* nk) {
    po_result part;
    part.h = h;
    if (h->chunk_rows[k].p) {
        h->spare_rows.release();
        h->spare_rows = h->chunk_rows[k];
        h->chunk_rows[k] = DevBuf();
    }
    const po_status st = h->bits == 2 ? run_overlaps<2>(h, min_length, k, n_chunks, false, &part)
                                      : run_overlaps<8>(h, min_length, k, n_chunks, false, &part);
    h->chunk_rows[k] = part.d_rows;   // (run_overlaps returned: this chunk's rows are complete on the device)
    h->chunk_compact[k] = part.compact;
    part.d_rows = DevBuf();
    *nk = part.count;
    return st;
}

// May this call take the streamed form?  (the index flavour is decided as run_overlaps decides it)
bool stream_eligible(const po_handle* h, uint32_t min_length) {
    const uint32_t n = (uint32_t)h->len.size();
    if (!h->dirty || h->bits != 2 || n < 4 || (n % 2) != 0 || !h->all_pairs_rcx) return false;
    if (h->asm_pieces || h->ex_on || h->sl_build_n > 1 || h->ext_index) return false;
    if (getenv("PHASM_FULL_UPLOAD") || getenv("PHASM_NO_MIRROR")) return false;
    // (worth it from ~16 MB of packed even reads on: below that a piece's fixed cost, ~0.2 ms of small launches, is more
    // than the transfer time it hides)
    bool want = n >= 4096 && h->words[0].size() * 8 >= (16ull << 20);
    if (const char* e = getenv("PHASM_STREAM")) want = atoi(e) != 0;
    if (!want) return false;
    const uint32_t m = min_length ? min_length : 1;
    const uint64_t n_elig = count_eligible(const_cast<po_handle*>(h), m);
    if (n_elig == 0) return false;
    return true;   // (either index flavour: the narrow one needs every read's first word ahead of the pieces, the wide one two)
}

// piece boundaries (even read indices, first 0, last n): cut points in thousandths of the packed store.  Pair (a, b)
// can be computed once both reads are there, so the work released by a piece grows with its position -- the last
// pieces are made small, their rows are what is left to send home after the upload has ended.
std::vector<uint32_t> stream_bounds(const po_handle* h) {
    const uint32_t n = (uint32_t)h->len.size();
    // one piece per ~16 MB, 2 to 12 of them (8 until a piece's fixed device cost fell in round 3: config 2 5.18 -> 5.11 ms,
    // config 3 19.6 -> 19.0), cut at 1 - (1 - i/P)^1.25: 154/302/444/580/707/823/926 thousandths at P = 8
    // (config 2, 190 MB: 5.3 ms per step; 5 to 10 pieces measure within 3 % of it.  Config 2 scaled to 12 k reads, 45 MB:
    // 8 pieces 2.24 ms, 3 pieces 1.83, none 2.38; to 6 k reads, 22 MB: 8 pieces 1.74, 2 pieces 1.15, none 1.52)
    const uint64_t bytes0 = (uint64_t)h->words[0].size() * 8;
    // (the FIRST streamed call on a handle has no kept row buffers yet: every piece then costs two host round trips, and
    // eight pieces are the better cut -- 7.9 against 10.5 ms for the cold call at config 2)
    uint64_t max_pieces = h->chunk_rows[0].p ? 12 : 8;
    // (stores of half a gigabyte and more -- configs 3 and 5: a piece is milliseconds of device work there, its fixed cost is
    // nothing, and what counts is how little is left to do when the last byte has landed: 15 pieces, later ones smaller.
    // Config 3: 17.04 -> 16.59 ms; config 5, whose kernels take as long as its upload: 68.99 -> 68.77)
    const bool big = bytes0 >= (512ull << 20) && h->chunk_rows[0].p;
    if (big) max_pieces = 15;
    if (const char* e = getenv("PHASM_STREAM_MAX_PIECES")) max_pieces = (uint64_t)std::max(2, std::min(PO_MAX_PIECES - 1, atoi(e)));
    const uint32_t n_pieces = (uint32_t)std::min<uint64_t>(max_pieces, std::max<uint64_t>(2, (bytes0 + (8ull << 20)) / (16ull << 20)));
    const double skew = big ? 1.6 : 1.25;
    std::vector<uint32_t> cuts;
    for (uint32_t i = 1; i < n_pieces; ++i) cuts.push_back((uint32_t)(1000.0 * (1.0 - std::pow(1.0 - (double)i / n_pieces, skew))));
    if (const char* e = getenv("PHASM_STREAM_CUTS")) {
        cuts.clear();
        for (const char* q = e; *q;) {
            char* endp = nullptr;
            const long v = strtol(q, &endp, 10);
            if (endp == q) break;
            if (v > 0 && v < 1000 && (cuts.empty() || (uint32_t)v > cuts.back()) && cuts.size() + 1 < (size_t)PO_MAX_PIECES) cuts.push_back((uint32_t)v);
            q = *endp ? endp + 1 : endp;
        }
    }
    std::vector<uint32_t> b{0};
    const uint64_t total = h->words[0].size();
    for (uint32_t c : cuts) {
        const uint64_t target = total * c / 1000;
        uint32_t lo = 0, hi = n / 2;   // smallest pair i with woff[2 i] >= target
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (h->woff[2 * (size_t)mid] >= target) hi = mid; else lo = mid + 1;
        }
        if (2 * lo > b.back() && 2 * lo < n) b.push_back(2 * lo);
    }
    b.push_back(n);
    return b;
}

// start of a streamed step: everything but the packed words goes up, the pieces are queued on up_stream (one event
// each), every read's first word is put in place for the index
po_status stream_begin(po_handle* h, const std::vector<uint32_t>& bounds) {
    const uint32_t n = (uint32_t)h->len.size();
    const uint32_t P = (uint32_t)bounds.size() - 1;
    bool generate = false;
    if (!h->up_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    for (uint32_t k = 0; k < P; ++k)
        if (!h->ev_piece[k]) HIP_TRY(h, hipEventCreate(&h->ev_piece[k]));
    uint64_t piece0_bytes = 0;
    auto queue_piece = [&](uint32_t k) -> po_status {
        uint64_t* dw = h->d_words.as<uint64_t>();
        const uint64_t wb = bounds[k] < n ? h->woff[bounds[k]] : h->words[0].size();
        const uint64_t we = bounds[k + 1] < n ? h->woff[bounds[k + 1]] : h->words[0].size();
        if (we > wb) {
            HIP_TRY(h, hipMemcpyAsync(dw + wb, h->words[0].data() + wb, (we - wb) * 8, hipMemcpyHostToDevice, h->up_stream));
            h->upload_bytes += (we - wb) * 8;
            if (k == 0) piece0_bytes = (we - wb) * 8;
        }
        HIP_TRY(h, hipEventRecord(h->ev_piece[k], h->up_stream));
        return PO_OK;
    };
    // (Measured and not kept, round 4: the first piece on the wire BEFORE the step's preparation.  The per-read tables and the
    // first words are host->device copies too and queue behind the piece's 20 MB on the same engine, so the index -- built
    // from them while piece 0 is on the wire -- is late by what the piece takes to land.  PHASM_EARLY_PIECE0=1 brings it back.)
    bool early0 = false;
    if (getenv("PHASM_EARLY_PIECE0")) {
        const uint64_t base1 = (h->words[0].size() + 1) & ~uint64_t(1);
        const uint64_t nwords = base1 + h->words[1].size() + 72;
        if (h->poison < 0 && h->d_words.p && h->d_words.cap >= nwords * 8 && h->bits == 2 && h->all_pairs_rcx && !getenv("PHASM_FULL_UPLOAD")) {
            HIP_TRY(h, hipEventRecord(h->ev_up0, h->up_stream));
            PO_TRY(queue_piece(0));
            early0 = true;
        }
    }
    PO_TRY(upload_meta(h, &generate));   // (sets upload_bytes to what the tables weigh)
    if (!generate) {
        if (early0) (void)hipStreamSynchronize(h->up_stream);
        return fail(h, PO_ERR_INVALID, "streamed step on reads that are not (x, reverse complement of x) pairs");
    }
    uint64_t* dw = h->d_words.as<uint64_t>();
    if (h->poison >= 0) {
        // (PHASM_POISON fills a fresh device buffer on the handle's stream: the pieces must not land under that fill)
        HIP_TRY(h, hipEventRecord(h->ev_up1, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->up_stream, h->ev_up1, 0));
    }
    if (early0) {
        h->upload_bytes += piece0_bytes;
    } else {
        HIP_TRY(h, hipEventRecord(h->ev_up0, h->up_stream));
        PO_TRY(queue_piece(0));
    }
    if (P > 1) {
        // first words of the reads of the later pieces (both strands: the host packed the odd store too, it just does
        // not travel), put in place on the handle's stream while piece 0 is crossing; the later pieces' copies are
        // ordered behind that kernel -- they bring the same values, but two writers of one word want an order
        // With the index built ahead of piece 0 (overlaps_streamed) the first words of piece 0's reads go up too: the index
        // build behind this kernel is then their only reader before the piece itself has landed, and the piece brings
        // the same values.
        const uint32_t r0 = h->st_early_index ? 0u : bounds[1];
        if (h->st_lead > 2) {
            // (large read sets: five words per read, so that the index can hold window minimisers -- 40 B per read ahead of
            // the pieces instead of 16, 2 % of the upload, for a counting pass of 0.55 x the time)
            note_lead_words(h);
            PO_TRY(ensure(h, h->d_first, (size_t)n * LEAD_WORDS * 8));
            HIP_TRY(h, hipMemcpyAsync(h->d_first.as<uint64_t>() + (size_t)LEAD_WORDS * r0, h->lead_words.data() + (size_t)LEAD_WORDS * r0,
                                      (size_t)(n - r0) * LEAD_WORDS * 8, hipMemcpyHostToDevice, h->stream));
            h->upload_bytes += (size_t)(n - r0) * LEAD_WORDS * 8;
            hipLaunchKernelGGL(po::k_scatter_lead, dim3(cdiv((uint64_t)(n - r0) * LEAD_WORDS, 256)), dim3(256), 0, h->stream, dw,
                               h->d_woff.as<uint64_t>(), h->d_len.as<uint32_t>(), h->d_first.as<uint64_t>(), LEAD_WORDS, r0, n);
        } else {
            note_first_words(h);   // (nothing to do when po_add_sequence kept them up to date)
            PO_TRY(ensure(h, h->d_first, (size_t)n * 16));
            HIP_TRY(h, hipMemcpyAsync(h->d_first.as<uint64_t>() + 2 * (size_t)r0, h->first_words.data() + 2 * (size_t)r0,
                                      (size_t)(n - r0) * 16, hipMemcpyHostToDevice, h->stream));
            h->upload_bytes += (size_t)(n - r0) * 16;
            hipLaunchKernelGGL(po::k_scatter_first, dim3(cdiv(n - r0, 256)), dim3(256), 0, h->stream, dw, h->d_woff.as<uint64_t>(),
                               h->d_first.as<ulonglong2>(), r0, n);
        }
        HIP_TRY(h, hipGetLastError());
        if (!h->ev_first) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_first, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->ev_first, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->up_stream, h->ev_first, 0));
    }
    for (uint32_t k = 1; k < P; ++k) PO_TRY(queue_piece(k));
    HIP_TRY(h, hipEventRecord(h->ev_up1, h->up_stream));
    // reverse complements: piece k's odd reads as soon as piece k is there (needs the per-read tables of upload_meta and,
    // for the later pieces, the first words in place: both are on the handle's stream up to here)
    if (!h->rc_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->rc_stream, hipStreamNonBlocking));
    if (!h->ev_meta) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_meta, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->ev_meta, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->rc_stream, h->ev_meta, 0));
    for (uint32_t k = 0; k < P; ++k) {
        if (!h->ev_rc[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_rc[k], hipEventDisableTiming));
        if (h->want_two == 2) continue;   // (gated two-stream pieces queue their reverse complements themselves: run_overlaps)
        HIP_TRY(h, hipStreamWaitEvent(h->rc_stream, h->ev_piece[k], 0));
        const uint32_t p0 = bounds[k] / 2, p1 = bounds[k + 1] / 2;
        if (p1 > p0)
            hipLaunchKernelGGL(po::k_revcomp_store, dim3(cdiv((uint64_t)(p1 - p0) * 64, 256)), dim3(256), 0, h->rc_stream, dw,
                               h->d_woff.as<uint64_t>(), h->d_len.as<uint32_t>(), p0, p1,
                               h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr, h->d_exc_pos.as<uint32_t>());
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(h->ev_rc[k], h->rc_stream));
    }
    // deferred containment list: [Cand x cap | counter]
    h->st_defer_cap = std::max<uint32_t>(1u << 16, h->defer_need + h->defer_need / 2);
    if (const char* e = getenv("PHASM_DEFER_CAP")) h->st_defer_cap = (uint32_t)std::max(1, atoi(e));   // (tests: force the overflow path)
    PO_TRY(ensure(h, h->d_defer, (size_t)h->st_defer_cap * sizeof(po::Cand) + 16));
    if (!getenv("PHASM_DEFER_CAP")) h->st_defer_cap = (uint32_t)std::min<size_t>((h->d_defer.cap - 16) / sizeof(po::Cand), 0xFFFFFF00u);
    HIP_TRY(h, hipMemsetAsync(h->d_defer.as<char>() + (size_t)h->st_defer_cap * sizeof(po::Cand), 0, 16, h->stream));
    h->paired = true;   // (checked on the host as the reads arrived: all_pairs_rcx)
    ++h->upload_gen;
    h->dirty = false;
    return PO_OK;
}

// The streamed step.  *overflow: the deferred list was too small (nothing was handed out; the reads are resident now,
// the caller takes the chunked form, and the next streamed call sizes the list by what this one counted).
po_status overlaps_streamed(po_handle* h, uint32_t min_length, HostRows& R, po_stats& sum, bool* overflow) {
    *overflow = false;
    const uint32_t n = (uint32_t)h->len.size();
    const std::vector<uint32_t> bounds = stream_bounds(h);
    const uint32_t P = (uint32_t)bounds.size() - 1;
    const bool trace = getenv("PHASM_STREAM_TRACE") != nullptr;   // developer aid: host-clock marks of the pipeline on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    // the previous streamed call on the same reads, cuts and min_length tells how many candidates each piece will have
    std::vector<uint32_t> sig(bounds);
    sig.push_back(min_length);
    sig.push_back((uint32_t)(h->total_bases & 0xFFFFFFFFu));
    h->st_pred_valid = h->st_pred_valid && sig == h->st_pred_sig;
    uint64_t new_pred[PO_MAX_PIECES] = {};
    // the index ahead of piece 0: it needs every read's first word(s) only, and at 400 k reads (wide index, 2.9 ms; 16 ms at
    // 2 M reads) building it inside piece 0 -- after the piece has landed -- kept every later piece 2-3 ms behind its data
    h->st_selfclean = false;
    h->two_stream = 0;
    h->want_two = 0;
    h->st_early_index = P > 1 && h->poison < 0 && !getenv("PHASM_NO_INDEX_REUSE") && !getenv("PHASM_LATE_INDEX");
    {
        // words per read that travel ahead of the pieces: two, or five where the step will take the wide index and min_length
        // allows windows of 4 (run_overlaps decides the same way: more than 160 k eligible reads, min_length >= 5 W - 1)
        const uint32_t m = min_length ? min_length : 1;
        const char* idx = getenv("PHASM_INDEX");
        const char* win = getenv("PHASM_WIDE_WINDOW");
        const bool wide = (idx && !strcmp(idx, "wide")) || (!(idx && !strcmp(idx, "narrow")) && count_eligible(h, m) > 160000);
        h->st_lead = (h->bits == 2 && wide && m >= 32u * 4u + 31u && !(win && atoi(win) < 4) && !getenv("PHASM_STREAM_LEAD2")) ? LEAD_WORDS : 2u;
    }
    {
        // PHASM_TWO_STREAM=1: counting passes on a stream of their own (scan_stream); 2: on rc_stream, gated by the verify kernel
        // of the piece before.  Both need the index built ahead of piece 0 (its event is what the first pass waits for).
        const char* e2 = getenv("PHASM_TWO_STREAM");
        const int asked = e2 ? atoi(e2) : 0;
        h->want_two = (h->st_early_index && h->ev_idx && (asked == 1 || asked == 2)) ? asked : 0;
        for (int g = 0; g < 2 && h->want_two == 2; ++g)
            if (!h->ev_gate[g] && hipEventCreateWithFlags(&h->ev_gate[g], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                h->want_two = 0;
            }
    }
    PO_TRY(stream_begin(h, bounds));
    if (trace) std::fprintf(stderr, "[stream] %u pieces queued at %.3f ms\n", P, since());
    // (two-stream pieces need the index built ahead -- its event is what the first counting pass waits for -- and no poison
    // fills; PHASM_TWO_STREAM=0 keeps every piece on the handle's stream)
    // Measured, round 4 (config 2): 7.1-7.2 ms per step against 4.5 on one stream -- the persistent scan kernel takes every CU's
    // LDS and registers, the verify workgroups of the piece before wait behind it, and both run slower side by side than one
    // after the other.  Off unless PHASM_TWO_STREAM=1 asks for it.
    if (h->want_two == 1 && !h->scan_stream && hipStreamCreateWithFlags(&h->scan_stream, hipStreamNonBlocking) != hipSuccess) {
        h->scan_stream = nullptr;
        (void)hipGetLastError();
    }
    h->two_stream = h->want_two == 2 ? 2 : (h->want_two == 1 && h->scan_stream) ? 1 : 0;
    if (h->st_early_index) {
        po_result part;
        part.h = h;
        h->st_on = true;
        h->st_r_begin = bounds[0];
        h->st_r_end = bounds[1];
        h->idx_only = true;
        const po_status ist = run_overlaps<2>(h, min_length, 0, P, false, &part);
        h->idx_only = false;
        h->st_on = false;
        part.d_rows.release();
        if (ist != PO_OK) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->up_stream);
            (void)hipStreamSynchronize(h->rc_stream);
            if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
            h->dirty = true;
            return ist;
        }
    }
    uint64_t* dw = h->d_words.as<uint64_t>();
    po_status st = PO_OK;
    // a finished piece: statistics, rows queued for home.  Called by run_overlaps at the next piece's first host wait
    // (a piece that emitted into a buffer known to be large enough returns with its kernels still queued), or below.
    auto completed = [&](uint32_t k, const po_stats& S, uint64_t nk) -> po_status {
        add_stats(sum, S);
        new_pred[k] = S.n_candidates;
        const double seen = (double)bounds[k + 1] / (double)n;
        if (h->chunk_compact[k]) PO_TRY(append_home(h, R, h->chunk_rows[k], S.n_verified, nk, k, P, seen * seen));
        else PO_TRY(append_rows(h, R, h->chunk_rows[k], nk, k, P, seen * seen));
        if (trace) {
            float up = 0;
            (void)hipEventElapsedTime(&up, h->ev_up0, h->ev_piece[k]);
            std::fprintf(stderr, "[stream] piece %u reads [%u, %u): landed %.3f ms after the first copy started, rows counted at %.3f ms (device %.3f ms: scan %.3f verify %.3f), %llu rows (%.1f MB) queued for home\n",
                         k, bounds[k], bounds[k + 1], up, since(), S.ms_total, S.ms_scan_count + S.ms_scan_fill, S.ms_verify,
                         (unsigned long long)nk, nk * 24e-6);
        }
        return PO_OK;
    };
    h->st_pend.valid = false;
    bool tail_gave_up = false;   // a piece's k_tail met reads handed to the global (a, b) table (tandem repeats): chunked form
    h->st_harvest = [&]() -> po_status {
        const uint32_t k = h->st_pend.k;
        bool needs_classic = false;
        const uint64_t nk = finish_piece(h, &needs_classic);
        if (needs_classic) {
            tail_gave_up = true;
            return PO_OK;
        }
        return completed(k, h->st_pend.S, nk);
    };
    if (getenv("PHASM_STREAM_SYNC")) h->st_harvest = nullptr;   // (developer switch: every piece waits for its own end)
    for (uint32_t k = 0; k < P && st == PO_OK; ++k) {
        // piece k has landed and its odd reads (reverse complements) have been written next to it (rc_stream, stream_begin)
        // (gated two-stream pieces: the reverse complements are queued inside run_overlaps, in front of the counting pass the
        // handle's stream then waits for -- ev_rc[k] has not been recorded yet at this point)
        if (h->two_stream != 2 && hipStreamWaitEvent(h->stream, h->ev_rc[k], 0) != hipSuccess) { st = fail(h, PO_ERR_HIP, "hipStreamWaitEvent"); break; }
        h->st_on = true;
        h->st_k = k;
        h->st_r_begin = bounds[k];
        h->st_r_end = bounds[k + 1];
        // per-piece workspaces: sized for the largest piece when they are first needed.  A piece keeps the candidates
        // whose b lies below its a, so piece j has about (b[j+1]^2 - b[j]^2) / (b[k+1]^2 - b[k]^2) times piece k's
        h->ws_scale = 1.0;
        {
            const double mine = (double)bounds[k + 1] * bounds[k + 1] - (double)bounds[k] * bounds[k];
            for (uint32_t j = k + 1; j < P; ++j)
                h->ws_scale = std::max(h->ws_scale, ((double)bounds[j + 1] * bounds[j + 1] - (double)bounds[j] * bounds[j]) / mine * 1.1);
        }
        h->ev = h->ev_sets[k & 1];
        uint64_t nk = 0;
        st = run_chunk(h, min_length, k, P, &nk);
        h->st_on = false;
        h->ws_scale = 1.0;
        if (st != PO_OK || tail_gave_up) break;
        if (h->st_pend.valid && h->st_pend.k == k) continue;   // piece k is queued, its counts are read later (an older one was collected inside)
        if (h->st_pend.valid) {
            // (piece k never waited for the device -- it had nothing to scan: the older piece is still to be collected)
            if (hipStreamSynchronize(h->stream) != hipSuccess) { st = fail(h, PO_ERR_HIP, "stream"); break; }
            st = h->st_harvest();
            if (st != PO_OK) break;
        }
        st = completed(k, h->stats, nk);   // piece k waited for its own end
    }
    if (st == PO_OK && h->st_pend.valid && !tail_gave_up) {
        if (hipStreamSynchronize(h->stream) != hipSuccess) st = fail(h, PO_ERR_HIP, "stream");
        else st = h->st_harvest();
    }
    h->st_harvest = nullptr;
    h->st_pend.valid = false;
    h->ev = h->ev_sets[0];
    if (st == PO_OK && tail_gave_up) {
        // (rare: a read with hundreds of verified suffix-prefix hits.  Pieces may still be in flight and their odd reads
        // unwritten: everything is uploaded again by the chunked form)
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->up_stream);
        (void)hipStreamSynchronize(h->rc_stream);
        if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
        h->dirty = true;
        h->st_tail_gave_up = true;
        h->st_pred_valid = false;
        *overflow = true;
        return PO_OK;
    }
    if (st != PO_OK) {
        (void)hipStreamSynchronize(h->up_stream);
        (void)hipStreamSynchronize(h->rc_stream);
        if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
        h->dirty = true;   // (a piece may be missing on the device)
        h->st_pred_valid = false;
        return st;
    }
    h->st_pred_sig = sig;
    std::memcpy(h->st_pred_cand, new_pred, sizeof(new_pred));
    h->st_pred_valid = true;
    float ms = 0;
    (void)hipEventSynchronize(h->ev_up1);   // (recorded right behind the last piece's event, which the kernels waited for)
    (void)hipEventElapsedTime(&ms, h->ev_up0, h->ev_up1);
    h->stats.ms_upload = ms;
    // ---- the deferred containments: every read is there now
    HIP_TRY(h, hipMemcpyAsync(h->pinned + 40, h->d_defer.as<char>() + (size_t)h->st_defer_cap * sizeof(po::Cand), 8,
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const uint32_t n_def = (uint32_t)h->pinned[40];
    h->defer_need = n_def;
    if (n_def > h->st_defer_cap) {
        *overflow = true;
        return PO_OK;
    }
    if (n_def) {
        hipStream_t s = h->stream;
        po::Cand* list = h->d_defer.as<po::Cand>();
        PO_TRY(ensure(h, h->d_rowcnt, (size_t)n_def));
        PO_TRY(ensure(h, h->d_row_off, ((size_t)n_def + 1) * 4));
        HIP_TRY(h, hipEventRecord(h->ev[EV_START], s));
        hipLaunchKernelGGL(po::k_verify_flat, dim3(cdiv((uint64_t)n_def * 64, 256)), dim3(256), 0, s, dw, h->d_woff.as<uint64_t>(),
                           h->d_len.as<uint32_t>(), list, n_def, h->n_exc_uploaded ? h->d_exc_off.as<uint32_t>() : nullptr,
                           h->d_exc_pos.as<uint32_t>(), h->d_exc_byte.as<uint8_t>());
        hipLaunchKernelGGL(po::k_deferred_rowcnt, dim3(cdiv(n_def, 256)), dim3(256), 0, s, list, n_def, 1u, h->d_rowcnt.as<uint8_t>());
        HIP_TRY(h, hipGetLastError());
        PO_TRY(prefix_sum<uint8_t>(h, h->d_rowcnt.as<uint8_t>(), n_def, h->d_row_off.as<uint32_t>(), &h->pinned[2]));
        HIP_TRY(h, hipStreamSynchronize(s));
        const uint64_t n_rows = h->pinned[2];
        if (n_rows) {
            PO_TRY(ensure(h, h->chunk_rows[P], n_rows * sizeof(po_row)));
            unsigned long long* scalars = h->d_scalars.as<unsigned long long>();
            HIP_TRY(h, hipMemsetAsync(scalars + 4, 0, 32, s));
            hipLaunchKernelGGL(po::k_emit_cands, dim3(std::min<uint32_t>(cdiv(n_def, 256), (uint32_t)h->n_cu * 16)), dim3(256), 0, s,
                               list, h->d_rowcnt.as<uint8_t>(), h->d_row_off.as<uint32_t>(), n_def, h->d_len.as<uint32_t>(),
                               h->chunk_rows[P].as<po::Row>(), 2u, 1u, scalars + 4);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipMemcpyAsync(h->pinned + 4, scalars + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipEventRecord(h->ev[EV_EMIT], s));
            HIP_TRY(h, hipStreamSynchronize(s));
            float ms_def = 0;
            (void)hipEventElapsedTime(&ms_def, h->ev[EV_START], h->ev[EV_EMIT]);
            sum.n_verified += h->pinned[4];
            sum.sum_overlap_bases += h->pinned[5];
            sum.verify_bytes_algo += h->pinned[6];
            sum.verify_bytes_exec += h->pinned[7];
            sum.n_rows += n_rows;
            sum.ms_verify += ms_def;
            sum.ms_total += ms_def;
            PO_TRY(append_rows(h, R, h->chunk_rows[P], n_rows, P, P + 1));
        }
    }
    (void)n;
    if (trace) {
        std::fprintf(stderr, "[stream] deferred list settled at %.3f ms (%u candidates)\n", since(), n_def);
        (void)hipStreamSynchronize(h->copy_stream);
        std::fprintf(stderr, "[stream] last row home at %.3f ms\n", since());
    }
    return PO_OK;
}

}  // namespace
}  // extern "C++"

po_status po_overlaps_to_host(po_handle* h, uint32_t min_length, po_result** out) {
    if (!h || !out) return PO_ERR_INVALID;
    *out = nullptr;
    if (h->segments_only) return fail(h, PO_ERR_INVALID, "this handle holds GFA segments without sequences: nothing to overlap");
    // small jobs: nothing to overlap (a chunk costs ~40 kernel launches)
    uint32_t n_chunks = (h->len.size() >= 8192 && h->total_bases >= (64ull << 20)) ? 4u : 1u;
    if (const char* e = getenv("PHASM_HOST_CHUNKS")) n_chunks = (uint32_t)std::max(1, std::min(4, atoi(e)));
    po_result* r = new (std::nothrow) po_result();
    if (!r) return fail(h, PO_ERR_NOMEM, "out of host memory");
    r->h = h;
    po_status st = PO_OK;
    HostRows R;
    po_stats sum = {};
    bool streamed = false, overflowed = false;
    float ms_upload = 0;
    try {
        st = init_device(h);
        if (st == PO_OK && !h->copy_stream && hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess)
            st = fail(h, PO_ERR_HIP, "cannot create the copy stream");
        if (st == PO_OK && h->spare_host.p) {   // the pinned array of an earlier result, if there is one
            R.hb = h->spare_host;
            h->spare_host = HostBuf();
        }
        if (st == PO_OK) {
            (void)home_begin(h);   // rows home as records + helper threads (PHASM_HOME=0: every row crosses PCIe as before)
            h->home_last_bytes = 0;
        }
        if (st == PO_OK && stream_eligible(h, min_length)) {
            bool overflow = false;
            st = overlaps_streamed(h, min_length, R, sum, &overflow);
            ms_upload = h->stats.ms_upload;
            if (st == PO_OK && !overflow) {
                streamed = true;
            } else if (st == PO_OK) {
                // (rare: more containments of later reads than the list holds -- the reads are resident now)
                overflowed = true;
                home_wait(h);
                if (h->home_on) {   // (the abandoned step's records are not part of the result)
                    home::g_pool->sum_l.store(0);
                    home::g_pool->sum_b.store(0);
                    home::g_pool->sum_e.store(0);
                }
                if (hipStreamSynchronize(h->copy_stream) != hipSuccess) st = fail(h, PO_ERR_HIP, "copy stream");
                R.total = 0;
                sum = po_stats();
            }
        }
        if (st == PO_OK && !streamed) st = upload(h);
        for (uint32_t k = 0; k < n_chunks && st == PO_OK && !streamed; ++k) {
            uint64_t nk = 0;
            st = run_chunk(h, min_length, k, n_chunks, &nk);
            if (st != PO_OK) break;
            add_stats(sum, h->stats);
            st = h->chunk_compact[k] ? append_home(h, R, h->chunk_rows[k], h->stats.n_verified, nk, k, n_chunks)
                                     : append_rows(h, R, h->chunk_rows[k], nk, k, n_chunks);
        }
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_overlaps_to_host");
    }
    h->st_on = false;
    h->st_harvest = nullptr;   // (an exception may have left the streamed step half way: its callback captures dead locals)
    h->st_pend.valid = false;
    h->ev = h->ev_sets[0];
    if (h->dev_ready) (void)hipStreamSynchronize(h->stream);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    if (h->rc_stream) (void)hipStreamSynchronize(h->rc_stream);
    if (h->scan_stream) (void)hipStreamSynchronize(h->scan_stream);
    if (h->copy_stream && hipStreamSynchronize(h->copy_stream) != hipSuccess && st == PO_OK) st = fail(h, PO_ERR_HIP, "row copy device->host");
    {
        // the helper threads have written every piece's rows before the array is handed out (or released)
        const int herr = home_end(h, &sum);
        if (herr && st == PO_OK)
            st = fail(h, PO_ERR_HIP, herr == 1 ? "internal: the host's row count of a piece differs from the device's"
                                     : herr == 2 ? "internal: a verified-candidate record names a read the handle does not hold"
                                                 : "the event behind a device->host copy of records failed");
    }
    if (st != PO_OK) {
        R.hb.release();
        delete r;
        return st;
    }
    // the call's statistics: sums over the chunks (per-call fields from the last chunk)
    po_stats& S = h->stats;
    S.n_candidates = sum.n_candidates;
    S.n_verified = sum.n_verified;
    S.n_rows = sum.n_rows;
    S.sum_overlap_bases = sum.sum_overlap_bases;
    S.verify_bytes_algo = sum.verify_bytes_algo;
    S.verify_bytes_exec = sum.verify_bytes_exec;
    S.n_tiles = sum.n_tiles;
    S.shard_bases = sum.shard_bases;
    S.ms_index = sum.ms_index;
    S.ms_scan_count = sum.ms_scan_count;
    S.ms_scan_fill = sum.ms_scan_fill;
    S.ms_verify = sum.ms_verify;
    S.ms_select = sum.ms_select;
    S.ms_emit = sum.ms_emit;
    S.ms_total = sum.ms_total;
    S.ms_scan_probe = sum.ms_scan_probe;
    S.ms_verify_kernel = sum.ms_verify_kernel;
    S.fused_tail = sum.fused_tail;
    S.n_predicted = sum.n_predicted;
    S.home_record_bytes = h->home_last_bytes ? (h->home_sh_b ? 8u : 16u) : 0u;
    S.tail_fallback = sum.tail_fallback + (h->st_tail_gave_up ? 1u : 0u);
    h->st_tail_gave_up = false;
    S.streamed = streamed ? 1u : 0u;
    S.n_deferred = (streamed || overflowed) ? h->defer_need : 0u;   // (streamed == 0 with n_deferred > 0: the list overflowed, chunked form taken)
    if (streamed) {
        S.ms_upload = ms_upload;
        S.upload_bytes = h->upload_bytes;
    }
    h->last_host_rows = R.total;
    r->count = R.total;
    r->unique_twins = h->bits == 2 && h->paired;   // (chunks are a-major and disjoint in a: groups stay adjacent)
    if (R.total) {
        r->host = R.hb.p;
        r->host_cap = R.hb.cap;
    } else if (R.hb.p) {
        h->spare_host = R.hb;   // nothing to hand out: keep the buffer
    }
    ++h->live_results;
    *out = r;
    return PO_OK;
}

// ---- sliced wide index for multi-GPU steps (phasm_amd/dist.py: IndexExchange) ------------------------------------
po_status po_index_slice_build(po_handle* h, uint32_t min_length, uint32_t slice, uint32_t n_slices, uint32_t* is_wide,
                               uint32_t* slice_bits, uint64_t* chain_entries) {
    if (!h || !is_wide || !slice_bits || !chain_entries) return PO_ERR_INVALID;
    if (n_slices < 2 || slice >= n_slices) return fail(h, PO_ERR_INVALID, "po_index_slice_build: need 2 or more slices and slice < n_slices");
    *is_wide = 0;
    *slice_bits = 0;
    *chain_entries = 0;
    h->sl_build_slice = slice;
    h->sl_build_n = n_slices;
    h->sl_is_wide = false;
    po_result* r = nullptr;
    const po_status st = overlaps_common(h, min_length, 0, 1, false, &r);
    h->sl_build_n = 0;
    if (r) po_result_free(r);
    if (st != PO_OK) return st;
    *is_wide = h->sl_is_wide ? 1u : 0u;
    if (h->sl_is_wide) {
        *slice_bits = h->sl_tbits;
        *chain_entries = h->sl_entries;
    }
    return PO_OK;
}

// the triple a caller hands to the *_indexed entry points describes a buffer the library cannot see: refuse what cannot
// be a sliced index at all (a shift by slice_bits >= 64 is undefined, the scan addresses 2^slice_bits slots per chunk)
static bool slice_args_ok(uint32_t n_slices, uint32_t slice_bits, uint64_t chain_capacity) {
    return n_slices >= 2 && n_slices <= 4096 && slice_bits >= 1 && slice_bits <= 30 && chain_capacity < (1ull << 36);
}

uint64_t po_index_chunk_bytes(uint32_t slice_bits, uint64_t chain_capacity, uint64_t* chain_offset_bytes) {
    if (chain_offset_bytes) *chain_offset_bytes = 0;
    if (slice_bits < 1 || slice_bits > 30 || chain_capacity >= (1ull << 36)) return 0;   // (not a sliced index)
    // [2^bits + 1 slots of 16 bytes | padding to 256 | chain_capacity entries of 8 bytes | padding to 256]
    const uint64_t off = ((((1ull << slice_bits) + 1ull) * sizeof(po::Slot)) + 255ull) & ~255ull;
    if (chain_offset_bytes) *chain_offset_bytes = off;
    return (off + chain_capacity * 8ull + 255ull) & ~255ull;
}

po_status po_index_slice_export(po_handle* h, void* dst_device, uint64_t chain_capacity) {
    if (!h || !dst_device) return PO_ERR_INVALID;
    if (!h->sl_is_wide || !h->dev_ready) return fail(h, PO_ERR_INVALID, "po_index_slice_export: no sub-table has been built on this handle");
    if (chain_capacity < h->sl_entries) return fail(h, PO_ERR_INVALID, "po_index_slice_export: chain capacity below this sub-table's entries");
    uint64_t off = 0;
    (void)po_index_chunk_bytes(h->sl_tbits, chain_capacity, &off);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(dst_device, h->d_table.p, ((size_t)(1ull << h->sl_tbits) + 1) * sizeof(po::Slot), hipMemcpyDeviceToDevice, h->stream));
    if (h->sl_entries)
        HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(dst_device) + off, h->d_chain.p, (size_t)h->sl_entries * 8, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PO_OK;
}

po_status po_candidates_shard_indexed(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards,
                                      const void* index_device, uint32_t n_slices, uint32_t slice_bits, uint64_t chain_capacity,
                                      void* dst_device, uint64_t capacity, int* written, po_result** out) {
    if (written) *written = 0;
    if (!h || !out || !index_device) return PO_ERR_INVALID;
    if (!slice_args_ok(n_slices, slice_bits, chain_capacity))
        return fail(h, PO_ERR_INVALID, "sliced index: need 2..4096 slices of 2^1..2^30 slots (slice_bits as po_index_slice_build reported it)");
    if (!dst_device && capacity) return PO_ERR_INVALID;
    uint64_t off = 0;
    const uint64_t chunk = po_index_chunk_bytes(slice_bits, chain_capacity, &off);
    if ((chunk / 16) * n_slices >= 0xFFFFFFF0ull) return fail(h, PO_ERR_CAPACITY, "sliced index larger than 64 GB");
    h->ext_index = index_device;
    h->ext_slices = n_slices;
    h->ext_tbits = slice_bits;
    h->ext_chunk_slots = (uint32_t)(chunk / 16);
    h->ext_chain_off = (uint32_t)(off / 16);
    const po_status st = overlaps_common(h, min_length, shard, nshards, true, out, dst_device, capacity);
    h->ext_index = nullptr;
    if (st == PO_OK && written) *written = (*out)->wrote_ext ? 1 : 0;
    return st;
}

// (test hook: the row form of the same call)
po_status po_overlaps_shard_indexed(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, const void* index_device,
                                    uint32_t n_slices, uint32_t slice_bits, uint64_t chain_capacity, po_result** out) {
    if (!h || !out || !index_device) return PO_ERR_INVALID;
    if (!slice_args_ok(n_slices, slice_bits, chain_capacity))
        return fail(h, PO_ERR_INVALID, "sliced index: need 2..4096 slices of 2^1..2^30 slots (slice_bits as po_index_slice_build reported it)");
    uint64_t off = 0;
    const uint64_t chunk = po_index_chunk_bytes(slice_bits, chain_capacity, &off);
    if ((chunk / 16) * n_slices >= 0xFFFFFFF0ull) return fail(h, PO_ERR_CAPACITY, "sliced index larger than 64 GB");
    h->ext_index = index_device;
    h->ext_slices = n_slices;
    h->ext_tbits = slice_bits;
    h->ext_chunk_slots = (uint32_t)(chunk / 16);
    h->ext_chain_off = (uint32_t)(off / 16);
    const po_status st = overlaps_common(h, min_length, shard, nshards, false, out);
    h->ext_index = nullptr;
    return st;
}

po_status po_overlaps_ex(po_handle* h, uint32_t min_length, uint32_t max_diff, uint32_t band, po_result** out) {
    if (!h || !out) return PO_ERR_INVALID;
    *out = nullptr;
    if (band > 30) return fail(h, PO_ERR_INVALID, "po_overlaps_ex: band must be <= 30 (2*band+1 diagonals on lanes 1..61, one lane each; lanes 0 and 63 let the bases in)");
    if (max_diff >= (1u << 16)) return fail(h, PO_ERR_INVALID, "po_overlaps_ex: max_diff must be < 65536");
    h->ex_on = true;
    h->ex_E = max_diff;
    h->ex_W = band;
    const po_status st = overlaps_common(h, min_length, 0, 1, false, out);
    h->ex_on = false;
    return st;
}

po_status po_shard_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint32_t* r_begin, uint32_t* r_end) {
    if (!h || !r_begin || !r_end || nshards == 0 || shard >= nshards) return PO_ERR_INVALID;
    try {
        shard_range(h, shard, nshards, r_begin, r_end, nullptr);
    } catch (const std::bad_alloc&) {
        return PO_ERR_NOMEM;
    }
    return PO_OK;
}

uint64_t po_result_count(const po_result* r) { return r ? r->count : 0; }

const po_row* po_result_rows(po_result* r) {
    if (!r) return nullptr;
    if (r->host || r->count == 0) return static_cast<const po_row*>(r->host);
    po_handle* h = r->h;
    const void* src = r->wrote_ext ? r->ext_dst : r->d_rows.p;  // (wrote_ext: the entries live in the caller's buffer)
    if (!src) return nullptr;
    if (hipSetDevice(h->device) != hipSuccess) return nullptr;
    // the destination is pinned memory owned by the handle's pool: one DMA at the PCIe rate, and no pageable
    // heap range is ever handed to the runtime as a copy target
    const size_t bytes = r->count * r->elem;
    HostBuf hb;
    if (h->spare_host.cap >= bytes) {
        hb = h->spare_host;
        h->spare_host = HostBuf();
    } else if (ensure_host(h, hb, bytes) != PO_OK) {
        return nullptr;
    }
    hipError_t e = hipMemcpyAsync(hb.p, src, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        fail(h, PO_ERR_HIP, std::string("row copy device->host: ") + hipGetErrorString(e));
        hb.release();
        return nullptr;
    }
    r->host = hb.p;
    r->host_cap = hb.cap;
    return static_cast<const po_row*>(r->host);
}

// Rows [first, first + count) only: what ONE rank of a multi-GPU job brings home -- every rank holds the merged rows
// on its device, the ranks' hosts (one node) together receive them once.
const po_row* po_result_rows_range(po_result* r, uint64_t first, uint64_t count) {
    if (!r || first > r->count || count > r->count - first) return nullptr;
    if (count == 0) return nullptr;
    if (r->host) return reinterpret_cast<const po_row*>(static_cast<const char*>(r->host) + first * r->elem);
    po_handle* h = r->h;
    const char* src = static_cast<const char*>(r->wrote_ext ? r->ext_dst : r->d_rows.p);
    if (!src || hipSetDevice(h->device) != hipSuccess) return nullptr;
    const size_t bytes = count * r->elem;
    if (ensure_host(h, h->scratch_host, bytes) != PO_OK) return nullptr;
    hipError_t e = hipMemcpyAsync(h->scratch_host.p, src + first * r->elem, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        fail(h, PO_ERR_HIP, std::string("row copy device->host: ") + hipGetErrorString(e));
        return nullptr;
    }
    return static_cast<const po_row*>(h->scratch_host.p);   // (valid until the next call that uses the handle's scratch)
}

const void* po_result_device_rows(const po_result* r) { return r ? (r->wrote_ext ? r->ext_dst : r->d_rows.p) : nullptr; }

po_status po_result_copy_to_device(po_result* r, void* dst_device) {
    if (!r || (!dst_device && r->count)) return PO_ERR_INVALID;
    if (r->count == 0) return PO_OK;
    po_handle* h = r->h;
    PO_TRY(init_device(h));
    PO_TRY(rows_to_device(h, r));
    HIP_TRY(h, hipMemcpyAsync(dst_device, r->d_rows.p, r->count * r->elem, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PO_OK;
}

po_status po_result_copy_prefix_to_device(po_result* r, void* dst_device, uint64_t count) {
    if (!r || (!dst_device && count)) return PO_ERR_INVALID;
    if (count > r->count) return fail(r->h, PO_ERR_INVALID, "po_result_copy_prefix_to_device: count exceeds the result");
    if (count == 0) return PO_OK;
    po_handle* h = r->h;
    PO_TRY(init_device(h));
    PO_TRY(rows_to_device(h, r));
    HIP_TRY(h, hipMemcpyAsync(dst_device, r->d_rows.p, count * r->elem, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PO_OK;
}

void po_result_free(po_result* r) {
    if (!r) return;
    po_handle* h = r->h;
    if (r->host_malloced) {
        std::free(r->host);
    } else if (r->host) {
        HostBuf hb;
        hb.p = r->host;
        hb.cap = r->host_cap;
        if (h && hb.cap > h->spare_host.cap) {  // keep the larger pinned buffer for the next result
            h->spare_host.release();
            h->spare_host = hb;
        } else {
            hb.release();
        }
    }
    r->host = nullptr;
    if (h) {
        --h->live_results;
        // keep the larger buffer for the next call
        DevBuf* spare = r->elem == sizeof(po_row) ? &h->spare_rows : (r->kind_edges ? &h->spare_edges : &h->spare_cands);
        if (r->d_rows.p && r->d_rows.cap > spare->cap) {
            spare->release();
            *spare = r->d_rows;
            r->d_rows = DevBuf();
        }
    }
    r->d_rows.release();
    delete r;
}

// GFA2 edge lines for every row, byte-identical to the reference's
// gfa_line("E", "*", a_id, b_id, astart, aend, bstart, bend, "*")  (assembler.py:46-48, gfa.py:230-231).
po_status po_write_gfa_edges(po_result* r, int fd, uint64_t* lines_out) {
    if (!r || fd < 0) return PO_ERR_INVALID;
    po_handle* h = r->h;
    const bool graph_edges = r->elem == sizeof(po_edge);  // a po_layout_edges result: gfa2_write_graph's E lines
    if (r->elem != sizeof(po_row) && !graph_edges) return fail(h, PO_ERR_INVALID, "po_write_gfa_edges needs a row or edge result");
    if (lines_out) *lines_out = 0;
    if (r->count == 0) return PO_OK;
    const po_row* rows = po_result_rows(r);
    if (!rows) return PO_ERR_HIP;
    // Formatting 7 M lines and copying 324 MB into the file are the long parts of the `overlap` command after the call
    // itself (0.3-0.4 s at config 2 with a dozen threads appending to std::string and ONE thread writing).  Now: chunks of
    // 32 k rows are handed out dynamically to up to 16 threads; a thread formats its chunk straight into a buffer sized for
    // the worst case (two ids + four integers per line, digits from a two-digit table), learns its file offset from the
    // chunk before it (offsets are published along the chain as the sizes become known) and writes its own bytes with
    // pwrite -- formatting and the copy into the page cache both run in parallel, the file's bytes are the same.  A
    // descriptor that cannot seek (a pipe) keeps the ordered write() of whole chunks.
    size_t max_id = 0;
    for (const std::string& id : h->ids) max_id = std::max(max_id, id.size());
    const uint64_t chunk = 1u << 15;
    const uint64_t n_chunks = (r->count + chunk - 1) / chunk;
    const size_t line_max = 2 * max_id + 4 * 11 + 16;
    const off_t base_off = ::lseek(fd, 0, SEEK_CUR);
    const bool seekable = base_off >= 0;
    static const char digits2[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    auto put_int = [&](char* q, long long v) -> char* {
        if (v < 0) {
            *q++ = '-';
            v = -v;
        }
        unsigned long long u = (unsigned long long)v;
        char tmp[24];
        int n = 0;
        while (u >= 100) {
            const unsigned d = (unsigned)(u % 100) * 2;
            u /= 100;
            tmp[n++] = digits2[d + 1];
            tmp[n++] = digits2[d];
        }
        if (u >= 10) {
            tmp[n++] = digits2[u * 2 + 1];
            tmp[n++] = digits2[u * 2];
        } else {
            tmp[n++] = (char)('0' + u);
        }
        while (n) *q++ = tmp[--n];
        return q;
    };
    // one chunk -> buf; returns the bytes written, or -1 when a row names an unknown read
    auto format_chunk = [&](uint64_t lo, uint64_t hi, char* buf) -> long long {
        char* q = buf;
        const size_t n_ids = h->ids.size();
        for (uint64_t i = lo; i < hi; ++i) {
            po_row x;
            if (graph_edges) {
                // E * u v weight len(u) 0 overlap_len *   (gfa2_write_graph, phasm/io/gfa.py:315-327)
                const po_edge& e = reinterpret_cast<const po_edge*>(rows)[i];
                if (e.u >= n_ids || e.v >= n_ids) return -1;
                x = po_row{e.u, e.v, e.weight, (int32_t)h->len[e.u], 0, e.overlap_len};
            } else {
                x = rows[i];
            }
            if (x.a_idx >= n_ids || x.b_idx >= n_ids) return -1;
            const std::string &ia = h->ids[x.a_idx], &ib = h->ids[x.b_idx];
            std::memcpy(q, "E\t*\t", 4);
            q += 4;
            std::memcpy(q, ia.data(), ia.size());
            q += ia.size();
            *q++ = '\t';
            std::memcpy(q, ib.data(), ib.size());
            q += ib.size();
            *q++ = '\t';
            q = put_int(q, x.astart);
            *q++ = '\t';
            q = put_int(q, x.aend);
            *q++ = '\t';
            q = put_int(q, x.bstart);
            *q++ = '\t';
            q = put_int(q, x.bend);
            std::memcpy(q, "\t*\n", 3);
            q += 3;
        }
        return (long long)(q - buf);
    };
    try {
        const unsigned hw = cpu_share();
        const unsigned n_thr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw, 16), n_chunks));
        // off[c] = file offset of chunk c (-1 until chunk c - 1 has been formatted); off[n_chunks] = the end
        std::unique_ptr<std::atomic<long long>[]> off(new std::atomic<long long>[n_chunks + 1]);
        for (uint64_t c = 0; c <= n_chunks; ++c) off[c].store(-1, std::memory_order_relaxed);
        off[0].store(seekable ? (long long)base_off : 0);
        std::atomic<uint64_t> next{0};
        std::atomic<int> failed{0};   // 1 unknown read, 2 write error, 3 out of memory
        // a pipe: chunk c may only be written once chunk c - 1 has been (written[c] = done)
        std::unique_ptr<std::atomic<int>[]> written(new std::atomic<int>[n_chunks + 1]);
        for (uint64_t c = 0; c <= n_chunks; ++c) written[c].store(c == 0 ? 1 : 0, std::memory_order_relaxed);
        auto work = [&]() {
            std::unique_ptr<char[]> buf;
            try {
                buf.reset(new char[(size_t)chunk * line_max]);
            } catch (const std::bad_alloc&) {
                failed.store(3);
            }
            for (;;) {
                const uint64_t c = next.fetch_add(1);
                if (c >= n_chunks) return;
                long long n = -1;
                if (!failed.load()) {
                    n = format_chunk(c * chunk, std::min<uint64_t>(r->count, (c + 1) * chunk), buf.get());
                    if (n < 0) failed.store(1);
                }
                // (the chain of offsets goes on even after a failure: nobody may wait for ever)
                long long at;
                while ((at = off[c].load(std::memory_order_acquire)) < 0) home::cpu_pause();
                off[c + 1].store(at + std::max<long long>(n, 0), std::memory_order_release);
                if (!seekable)
                    while (!written[c].load(std::memory_order_acquire)) home::cpu_pause();
                if (n > 0 && !failed.load()) {
                    const char* p = buf.get();
                    long long left = n, pos = at;
                    while (left) {
                        const ssize_t w = seekable ? ::pwrite(fd, p, (size_t)left, (off_t)pos) : ::write(fd, p, (size_t)left);
                        if (w < 0) {
                            failed.store(2);
                            break;
                        }
                        p += w;
                        pos += w;
                        left -= w;
                    }
                }
                if (!seekable) written[c + 1].store(1, std::memory_order_release);
            }
        };
        std::vector<std::thread> thr;
        try {
            for (unsigned t = 1; t < n_thr; ++t) thr.emplace_back(work);
        } catch (const std::system_error&) {
            // fewer threads than asked for: carry on with those that started
        }
        work();
        for (auto& th : thr) th.join();
        if (failed.load() == 1) return fail(h, PO_ERR_INVALID, "a row names an unknown read");
        if (failed.load() == 2) return fail(h, PO_ERR_INVALID, "write failed");
        if (failed.load() == 3) return fail(h, PO_ERR_NOMEM, "out of host memory");
        if (seekable && ::lseek(fd, (off_t)off[n_chunks].load(), SEEK_SET) < 0) return fail(h, PO_ERR_INVALID, "lseek failed");
    } catch (const std::bad_alloc&) {
        return fail(h, PO_ERR_NOMEM, "out of host memory");
    }
    if (lines_out) *lines_out = r->count;
    return PO_OK;
}


// `S <name> <length> *` for every read pair (x+, x-) of the handle, the way the overlap command writes them
// before the E lines (assembler.py:38)
po_status po_write_gfa_segments(po_handle* h, int fd, uint64_t* lines_out) {
    if (!h || fd < 0) return PO_ERR_INVALID;
    if (lines_out) *lines_out = 0;
    if (!ids_are_strand_pairs(h)) return fail(h, PO_ERR_INVALID, "po_write_gfa_segments needs reads added as name+\"+\" / name+\"-\" pairs");
    try {
        std::string buf;
        buf.reserve(1 << 20);
        for (size_t i = 0; i < h->ids.size(); i += 2) {
            buf.append("S\t");
            buf.append(h->ids[i], 0, h->ids[i].size() - 1);
            buf.push_back('\t');
            buf.append(std::to_string(h->len[i]));
            buf.append("\t*\n");
        }
        const char* p = buf.data();
        size_t left = buf.size();
        while (left) {
            ssize_t w = ::write(fd, p, left);
            if (w < 0) return fail(h, PO_ERR_INVALID, "write failed");
            p += w;
            left -= (size_t)w;
        }
    } catch (const std::bad_alloc&) {
        return fail(h, PO_ERR_NOMEM, "out of host memory");
    }
    if (lines_out) *lines_out = h->ids.size() / 2;
    return PO_OK;
}

po_status po_add_segment(po_handle* h, const char* name, size_t name_len, uint32_t length) {
    if (!h || (!name && name_len)) return PO_ERR_INVALID;
    try {
        return add_segment(h, name, name_len, length);
    } catch (const std::bad_alloc&) {
        return fail(h, PO_ERR_NOMEM, "out of host memory in po_add_segment");
    }
}

po_status po_result_from_rows(po_handle* h, const po_row* rows, uint64_t n, po_result** out) {
    if (!h || !out || (!rows && n)) return PO_ERR_INVALID;
    *out = nullptr;
    po_result* r = new (std::nothrow) po_result();
    if (!r) return fail(h, PO_ERR_NOMEM, "out of host memory");
    r->h = h;
    r->count = n;
    if (n) {
        r->host = std::malloc(n * sizeof(po_row));
        if (!r->host) {
            delete r;
            return fail(h, PO_ERR_NOMEM, "out of host memory for the row array");
        }
        std::memcpy(r->host, rows, n * sizeof(po_row));
        r->host_malloced = true;
    }
    ++h->live_results;
    *out = r;
    return PO_OK;
}

po_status po_add_gfa(po_handle* h, const char* path, uint64_t* n_segments, po_result** rows_out) {
    if (!h || !path || !rows_out) return PO_ERR_INVALID;
    *rows_out = nullptr;
    if (n_segments) *n_segments = 0;
    if (!h->len.empty()) return fail(h, PO_ERR_INVALID, "po_add_gfa needs an empty handle");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(h, PO_ERR_INVALID, std::string("cannot open ") + path);
    struct stat sb;
    if (::fstat(fd, &sb) != 0) {
        ::close(fd);
        return fail(h, PO_ERR_INVALID, std::string("cannot stat ") + path);
    }
    const size_t size = (size_t)sb.st_size;
    const char* data = nullptr;
    if (size) {
        void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            return fail(h, PO_ERR_INVALID, std::string("cannot map ") + path);
        }
        data = static_cast<const char*>(m);
    }
    po_status st = PO_OK;
    std::vector<po_row> rows;
    try {
        // every line of the file, '\n'-terminated (a '\r' before it is white space to strip())
        auto for_lines = [&](char tag, auto&& fn) {
            size_t pos = 0;
            while (pos < size && st == PO_OK) {
                const char* nl = static_cast<const char*>(std::memchr(data + pos, '\n', size - pos));
                const size_t len = nl ? (size_t)(nl - (data + pos)) : size - pos;
                if (len && data[pos] == tag) fn(data + pos, len);
                pos += len + 1;
            }
        };
        // pass 1: segments (gfa2_segment_to_read, gfa.py:33-46)
        NameIndex idx;
        std::vector<std::pair<std::string, uint32_t>> segs;
        std::unordered_map<std::string, size_t> seen;
        for_lines('S', [&](const char* p, size_t n) {
            Field f[4];
            long long length = 0;
            if (split_tabs(p, n, f, 4) < 4 || !parse_int(f[2], false, length) || length < 0 || length > 0x7FFFFFF0ll) {
                st = fail(h, PO_ERR_INVALID, "malformed GFA2 segment line: " + std::string(p, std::min<size_t>(n, 80)));
                return;
            }
            strip(f[1].p, f[1].n);
            std::string name(f[1].p, f[1].n);
            auto it = seen.find(name);
            if (it != seen.end()) {
                segs[it->second].second = (uint32_t)length;  // dict semantics: one entry per name, the last length wins (gfa.py:109)
            } else {
                seen.emplace(name, segs.size());
                segs.emplace_back(std::move(name), (uint32_t)length);
            }
        });
        seen.clear();
        if (st == PO_OK) {
            for (auto& sgm : segs) {
                if ((st = add_segment(h, sgm.first.data(), sgm.first.size(), sgm.second)) != PO_OK) break;
            }
        }
        if (st == PO_OK) idx.build(h);
        if (n_segments) *n_segments = segs.size();
        segs.clear();
        segs.shrink_to_fit();
        // pass 2: edges (gfa2_parse_edge, gfa.py:72-87; gfa2_line_to_la, :90-104).  The file is cut into byte ranges
        // at line starts and a few threads parse one range each (7 M lines: 0.4 s on one core at config 2); the
        // per-range row vectors are joined in file order, the first error in file order wins.
        struct Part {
            std::vector<po_row> rows;
            std::string err;
        };
        auto parse_range = [&](size_t lo, size_t hi, Part& out) {
            const char* last_a = nullptr;
            size_t last_a_n = 0;
            long last_a_idx = -1;
            auto node_of = [&](Field f, bool cache) -> long {
                strip(f.p, f.n);
                if (f.n < 1) return -1;
                const char strand = f.p[f.n - 1];
                if (strand != '+' && strand != '-') return -1;
                long r;
                if (cache && last_a && last_a_n == f.n - 1 && std::memcmp(last_a, f.p, f.n - 1) == 0) {
                    r = last_a_idx;
                } else {
                    r = idx.find(h, f.p, f.n - 1);
                    if (cache) last_a = f.p, last_a_n = f.n - 1, last_a_idx = r;
                }
                return r < 0 ? -1 : 2 * r + (strand == '-');
            };
            size_t pos = lo;
            while (pos < hi) {
                const char* nl = static_cast<const char*>(std::memchr(data + pos, '\n', size - pos));
                const size_t n = nl ? (size_t)(nl - (data + pos)) : size - pos;
                const char* p = data + pos;
                pos += n + 1;
                if (!n || p[0] != 'E') continue;
                Field f[9];
                long long v[4];
                if (split_tabs(p, n, f, 9) < 9 || !parse_int(f[4], true, v[0]) || !parse_int(f[5], true, v[1]) ||
                    !parse_int(f[6], true, v[2]) || !parse_int(f[7], true, v[3])) {
                    out.err = "malformed GFA2 edge line: " + std::string(p, std::min<size_t>(n, 80));
                    return;
                }
                const long a = node_of(f[2], true), b = node_of(f[3], false);
                if (a < 0 || b < 0) {
                    out.err = "GFA2 edge names an unknown segment or strand: " + std::string(p, std::min<size_t>(n, 80));
                    return;
                }
                for (long long x : v)
                    if (x < -0x7FFFFFFFll || x > 0x7FFFFFFFll) {
                        out.err = "GFA2 edge position out of range";
                        return;
                    }
                out.rows.push_back(po_row{(uint32_t)a, (uint32_t)b, (int32_t)v[0], (int32_t)v[1], (int32_t)v[2], (int32_t)v[3]});
            }
        };
        if (st == PO_OK && size) {
            const unsigned hw = cpu_share();
            const unsigned n_parts = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(hw, 12u), size >> 22));
            std::vector<size_t> cut(n_parts + 1, size);
            cut[0] = 0;
            for (unsigned k = 1; k < n_parts; ++k) {  // a range starts right after a newline
                size_t c = size / n_parts * k;
                const char* nl = static_cast<const char*>(std::memchr(data + c, '\n', size - c));
                cut[k] = nl ? (size_t)(nl - data) + 1 : size;
            }
            std::vector<Part> parts(n_parts);
            std::vector<std::thread> thr;
            std::atomic<int> oom{0};
            auto run = [&](unsigned k) {
                try {
                    parse_range(cut[k], cut[k + 1], parts[k]);
                } catch (...) {
                    oom.store(1);
                }
            };
            try {
                for (unsigned k = 1; k < n_parts; ++k) thr.emplace_back(run, k);
            } catch (const std::system_error&) {
            }
            const unsigned started = (unsigned)thr.size() + 1;
            run(0);
            for (auto& th : thr) th.join();
            for (unsigned k = started; k < n_parts; ++k) run(k);  // (threads that could not be started)
            if (oom.load()) throw std::bad_alloc();
            size_t total = 0;
            for (const Part& pt : parts) {
                if (!pt.err.empty()) {
                    st = fail(h, PO_ERR_INVALID, pt.err);
                    break;
                }
                total += pt.rows.size();
            }
            if (st == PO_OK) {
                rows.reserve(total);
                for (Part& pt : parts) {
                    rows.insert(rows.end(), pt.rows.begin(), pt.rows.end());
                    std::vector<po_row>().swap(pt.rows);
                }
            }
        }
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_add_gfa");
    }
    if (data) ::munmap(const_cast<char*>(data), size);
    ::close(fd);
    if (st != PO_OK) return st;
    return po_result_from_rows(h, rows.data(), rows.size(), rows_out);
}

po_status po_layout_edges(po_handle* h, po_result* rows, const po_layout_params* params, uint8_t* removed_reads_out,
                          po_result** edges_out) {
    if (!h || !rows || !params || !edges_out) return PO_ERR_INVALID;
    *edges_out = nullptr;
    if (rows->h != h) return fail(h, PO_ERR_INVALID, "po_layout_edges: the rows belong to another handle");
    if (rows->elem != sizeof(po_row)) return fail(h, PO_ERR_INVALID, "po_layout_edges needs a row result");
    if (params->reserved != 0 || !(params->max_overhang_rel == params->max_overhang_rel))
        return fail(h, PO_ERR_INVALID, "po_layout_edges: bad parameters");
    if (!ids_are_strand_pairs(h))
        return fail(h, PO_ERR_INVALID, "po_layout_edges needs reads added as name+\"+\" / name+\"-\" pairs");
    po_result* r = new (std::nothrow) po_result();
    if (!r) return fail(h, PO_ERR_NOMEM, "out of host memory");
    r->h = h;
    po_status st;
    try {
        st = run_layout(h, rows, *params, removed_reads_out, r);
    } catch (const std::bad_alloc&) {
        st = fail(h, PO_ERR_NOMEM, "out of host memory in po_layout_edges");
    }
    if (st != PO_OK) {
        if (h->dev_ready) (void)hipStreamSynchronize(h->stream);
        r->d_rows.release();
        delete r;
        return st;
    }
    ++h->live_results;
    *edges_out = r;
    return PO_OK;
}

po_status po_get_layout_stats(const po_handle* h, po_layout_stats* out) {
    if (!h || !out) return PO_ERR_INVALID;
    *out = h->lstats;
    return PO_OK;
}

// ---- diagnostics (tests/checker.py: GuardedReads) ------------------------------------------------------------------
uint64_t po_debug_host_ranges(uint64_t* out, uint64_t cap_entries) {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    const uint64_t n = std::min<uint64_t>(cap_entries, g_pins.size());
    for (uint64_t i = 0; out && i < n; ++i) {
        out[3 * i] = g_pins[i].base;
        out[3 * i + 1] = g_pins[i].bytes;
        out[3 * i + 2] = g_pins[i].kind;
    }
    return g_pins.size();
}

uint64_t po_debug_store_words(const po_handle* h, int store, const uint64_t** words) {
    if (!h || store < 0 || store > 1) return 0;
    if (words) *words = h->words[store].data();
    return h->words[store].size();
}

// The host half of "rows home in compact form" on its own (no GPU): `n` records -> rows through the helper threads, exactly
// as po_overlaps_to_host runs them.  Returns 0, or the pool's error code (1 = the rows counted differ from n_rows_expected,
// 2 = a record names a read >= n_reads), -1 = no helper thread could be started.
int po_debug_expand_records(const po_cand* records, uint64_t n, const uint32_t* lengths, uint32_t n_reads, uint32_t paired,
                            po_row* rows_out, uint64_t n_rows_expected) {
    home::Pool* P = home::pool();
    if (!P) return -1;
    std::lock_guard<std::mutex> call(P->call_mu);
    home::begin(P, lengths, n_reads, false);
    P->paired = paired ? 1u : 0u;
    // (as the last piece of a streamed step travels: up to five parts of ONE piece, each part's rows behind the part before)
    const uint64_t n_jobs = n ? std::min<uint64_t>(5, (n + 2999) / 3000) : 0;
    for (uint64_t jn = 0; jn < n_jobs; ++jn) {
        const uint64_t lo = n * jn / n_jobs, hi = n * (jn + 1) / n_jobs;
        home::Job j;
        j.rec = reinterpret_cast<const po::Cand*>(records) + lo;
        j.n_rec = hi - lo;
        j.out = rows_out;
        j.n_rows = n_rows_expected;
        j.cont = jn > 0;
        j.more = jn + 1 < n_jobs;
        home::submit(P, j);
    }
    home::wait_all(P);
    return P->error.load();
}

// ... the same for records of 8 bytes (po::pack_record with the shifts sh_b, sh_p: a | b << sh_b | p << sh_p | type << 62)
int po_debug_expand_packed(const uint64_t* records, uint64_t n, uint32_t sh_b, uint32_t sh_p, const uint32_t* lengths, uint32_t n_reads,
                           uint32_t paired, po_row* rows_out, uint64_t n_rows_expected) {
    if (!sh_b || sh_p < sh_b || sh_p >= 62) return -2;
    home::Pool* P = home::pool();
    if (!P) return -1;
    std::lock_guard<std::mutex> call(P->call_mu);
    home::begin(P, lengths, n_reads, false);
    P->paired = paired ? 1u : 0u;
    const uint64_t n_jobs = n ? std::min<uint64_t>(5, (n + 2999) / 3000) : 0;
    for (uint64_t jn = 0; jn < n_jobs; ++jn) {
        const uint64_t lo = n * jn / n_jobs, hi = n * (jn + 1) / n_jobs;
        home::Job j;
        j.rec = records + lo;
        j.sh_b = sh_b;
        j.sh_p = sh_p;
        j.n_rec = hi - lo;
        j.out = rows_out;
        j.n_rows = n_rows_expected;
        j.cont = jn > 0;
        j.more = jn + 1 < n_jobs;
        home::submit(P, j);
    }
    home::wait_all(P);
    return P->error.load();
}

// SIGSEGV / SIGBUS: the faulting address and the NATIVE stack of the faulting thread to `fd`, then the handler that was
// installed before (Python's faulthandler prints Python frames only: a store by a runtime thread has none).
static int g_fault_fd = 2;
static struct sigaction g_fault_prev[2];
static void fault_handler(int sig, siginfo_t* si, void* ctx) {
    char line[128];
    const int n = std::snprintf(line, sizeof(line), "[phasm] signal %d at address %p (code %d); native stack of the faulting thread:\n", sig,
                                si ? si->si_addr : nullptr, si ? si->si_code : 0);
    if (n > 0) (void)!::write(g_fault_fd, line, (size_t)n);
    void* frames[64];
    const int k = backtrace(frames, 64);
    backtrace_symbols_fd(frames, k, g_fault_fd);
    const struct sigaction& prev = g_fault_prev[sig == SIGBUS ? 1 : 0];
    if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) {
        prev.sa_sigaction(sig, si, ctx);
        return;
    }
    if (prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler) {
        prev.sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

int po_debug_fault_backtrace(int fd) {
    g_fault_fd = fd;
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = fault_handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    void* warm[4];
    (void)backtrace(warm, 4);   // (loads libgcc now, not inside the handler)
    return (sigaction(SIGSEGV, &sa, &g_fault_prev[0]) == 0 && sigaction(SIGBUS, &sa, &g_fault_prev[1]) == 0) ? 0 : -1;
}

int po_debug_pointer_info(const void* p, int32_t* hip_type, int32_t* hsa_type, uint64_t* base, uint64_t* bytes) {
    if (hip_type) *hip_type = -1;
    if (hsa_type) *hsa_type = -1;
    if (base) *base = 0;
    if (bytes) *bytes = 0;
    if (!p) return 0;
    int known = 0;
    {
        hipPointerAttribute_t at;
        std::memset(&at, 0, sizeof(at));
        const hipError_t e = hipPointerGetAttributes(&at, p);
        if (e == hipSuccess) {
            if (hip_type) *hip_type = (int32_t)at.type;
            if (at.type != hipMemoryTypeUnregistered) known = 1;
        } else {
            (void)hipGetLastError();   // (an address the runtime has never heard of is an "invalid value" on older runtimes)
        }
    }
    {
        // the ROCr view: HIP's own pins of pageable copy sources / destinations (hsa_amd_memory_lock) show here too.
        // Resolved at run time from the HSA runtime the HIP runtime has loaded (this library links to HIP only).
        struct PtrInfo {   // hsa_amd_pointer_info_t (hsa_ext_amd.h)
            uint32_t size;
            uint32_t type;   // 0 unknown, 1 HSA allocation, 2 locked (registered host memory), 3 graphics, 4 IPC
            void* agentBaseAddress;
            void* hostBaseAddress;
            size_t sizeInBytes;
            void* userData;
            uint64_t agentOwner;
            uint32_t global_flags;
        };
        using Fn = int (*)(const void*, PtrInfo*, void* (*)(size_t), uint32_t*, void**);
        static Fn fn = reinterpret_cast<Fn>(dlsym(RTLD_DEFAULT, "hsa_amd_pointer_info"));
        if (fn) {
            PtrInfo info;
            std::memset(&info, 0, sizeof(info));
            info.size = sizeof(info);
            if (fn(p, &info, nullptr, nullptr, nullptr) == 0) {
                if (hsa_type) *hsa_type = (int32_t)info.type;
                if (info.type != 0) {
                    known = 1;
                    if (base) *base = (uint64_t)(uintptr_t)(info.hostBaseAddress ? info.hostBaseAddress : info.agentBaseAddress);
                    if (bytes) *bytes = info.sizeInBytes;
                }
            }
        }
    }
    return known;
}

po_status po_get_stats(const po_handle* h, po_stats* out) {
    if (!h || !out) return PO_ERR_INVALID;
    *out = h->stats;
    return PO_OK;
}

const char* po_last_error(const po_handle* h) { return h ? h->err.c_str() : "null handle"; }

}  // extern "C"
