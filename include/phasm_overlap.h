/* phasm_overlap.h -- C ABI of libphasm_overlap.so, the MI355X (gfx950) replacement for PHASM's
 * all-pairs exact read-overlap finder.
 *
 * What this boundary replaces.  The reference has no C ABI; its overlapper is a C++ class
 * exposed to Python through pybind11:
 *
 *   class ExactOverlapper                   /root/reference/src/overlapper.h:19-29
 *     ExactOverlapper()                     src/overlapper.cpp:19      (py::init,  src/phasm.cpp:13)
 *     addSequence(id, seq)                  src/overlapper.cpp:22-26   ("add_sequence", phasm.cpp:14)
 *     overlaps(min_length) -> vector<OverlapT>  src/overlapper.cpp:28-150 ("overlaps", phasm.cpp:15)
 *   OverlapT = tuple<string,string,int,int,int,int>                    src/overlapper.h:17
 *
 * Each entry point below names the reference interface it stands in for.  Only plain
 * pointers and sizes cross the boundary: no C++ types, no exceptions, no torch types.
 * Every function that can fail returns a po_status; po_last_error() has the message.
 *
 * There is no CPU fallback behind this ABI: every overlap is computed by the HIP kernels
 * in phasm_amd/csrc/.  Without a usable GPU, po_overlaps* fails with PO_ERR_HIP.
 *
 * Threading: one handle = one caller at a time (the reference object is not thread-safe
 * either).  Each handle owns one HIP stream and its device buffers.
 */
#ifndef PHASM_OVERLAP_H
#define PHASM_OVERLAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_ABI_VERSION 4

typedef enum {
    PO_OK = 0,
    PO_ERR_INVALID = 1,  /* bad argument (NULL handle, shard >= nshards, ...)            */
    PO_ERR_NOMEM = 2,    /* host or device allocation failed  -> MemoryError in the shim */
    PO_ERR_HIP = 3,      /* HIP runtime error / no device     -> RuntimeError            */
    PO_ERR_CAPACITY = 4  /* candidate or row count exceeds what one call can hold        */
} po_status;

typedef struct po_handle po_handle;
typedef struct po_result po_result;

/* One overlap edge.  Same six fields as OverlapT (src/overlapper.h:17) with the two id
 * strings replaced by read indices (insertion order of po_add_sequence); ids are resolved
 * with po_get_id.  Coordinates are 0-based half-open on the oriented string as added;
 * bstart is always 0 (src/overlapper.cpp:81,110).                                         */
typedef struct {
    uint32_t a_idx, b_idx;
    int32_t astart, aend, bstart, bend;
} po_row;

/* Per-call device timings (HIP events on the handle's stream) and counters of the last
 * po_overlaps* call on this handle; feeds bench.py's roofline object.                     */
typedef struct {
    uint32_t bits_per_base;      /* 2 (pure ACGT) or 8 (raw bytes)                          */
    uint32_t kmer;               /* anchor length K = min(64/bits, max(min_length,1))       */
    uint32_t paired;             /* 1: reads are (x, revcomp x) pairs -> one member of each  */
    uint32_t wide_index;         /*    strand-mirror pair verified, the other row mirrored.  */
                                 /* wide_index 1: W K-mers per read indexed, word-aligned      */
                                 /*    probes only (large read sets); 0: prefix + LDS filter  */
    uint64_t n_reads;            /* reads in the handle                                     */
    uint64_t n_eligible;         /* reads with length >= min_length (can be a `b`)          */
    uint64_t total_bases;        /* sum of read lengths (oriented bases B)                  */
    uint64_t shard_bases;        /* bases of the a-side reads scanned by this call          */
    uint64_t n_tiles;            /* 64-word scan tiles in this call's shard                 */
    uint64_t n_candidates;       /* anchor hits (a, p, b) handed to the verify kernel       */
    uint64_t n_verified;         /* candidates that verified                                */
    uint64_t n_rows;             /* rows emitted (A + B, duplicates included)               */
    uint64_t sum_overlap_bases;  /* sum over emitted rows of the overlap length l           */
    uint64_t verify_bytes_algo;  /* sum over emitted rows of 2*ceil(l*bits/8) (both sides)  */
                                 /* Stage times come from events recorded between the kernels (~5 us of device time     */
                                 /* each).  A streamed step (streamed == 1) records only the pairs around its two big   */
                                 /* kernels: ms_index .. ms_emit are 0 there, ms_total, ms_scan_probe and               */
                                 /* ms_verify_kernel are sums over the pieces; PHASM_PHASE_EVENTS=1 records them all.   */
    float ms_index;              /* anchor table + chains + Bloom filter build              */
    float ms_scan_count;         /* position scan, counting pass                            */
    float ms_scan_fill;          /* position scan, candidate fill pass                      */
    float ms_verify;             /* packed exact verify of all candidates                   */
    float ms_select;             /* longest-only selection + row counting                   */
    float ms_emit;               /* row emission                                            */
    float ms_total;              /* first kernel start -> last kernel end                   */
    float ms_upload;             /* H2D of the packed read set if this call uploaded it     */
    float ms_scan_probe;         /* the scan kernel alone (k_scan_probe / k_wide_scan count pass), inside ms_scan_count */
    float ms_verify_kernel;      /* the verify kernel alone (k_verify_a), inside ms_verify; 0 when there was nothing to verify */
    uint64_t verify_bytes_exec;  /* sum over the VERIFIED CANDIDATES of 2*ceil(n*bits/8): the bytes the verify kernel  */
                                 /* really compared (a strand-mirror pair is compared once and emitted twice)        */
    uint64_t dp_steps;           /* po_overlaps_ex: antidiagonals swept by the DP kernel, summed over the candidates */
    uint64_t dp_stopped;         /*                 candidates whose whole band rose above max_diff before the end   */
    uint32_t max_diff, band;     /*                 the parameters of the call (0, 0 for the exact entry points)     */
    uint32_t index_reused;       /* 1: the anchor index of the previous call on this handle was reused (same upload,   */
    uint32_t dp_lanes;           /*    same min_length and flavour); 0: built in this call.  dp_lanes (po_overlaps_ex):  */
                                 /*    2 = lane per candidate, band row as a bit vector; 1 = lane per candidate, band row */
                                 /*    in registers; 0 = wave per candidate (a lane per diagonal)                         */
    uint64_t upload_bytes;       /* bytes the last po_upload moved host->device (half the packed set when every   */
                                 /* odd read is the reverse complement of its even partner: the device rebuilds them) */
    uint32_t streamed;           /* po_overlaps_to_host: 1 = streamed step (reads uploaded piece by piece under the      */
    uint32_t n_deferred;         /*    kernels); n_deferred = containment candidates that waited for a later piece       */
                                 /*    (streamed == 0 with n_deferred > 0: their list overflowed, the chunked form ran)  */
    uint32_t fused_tail;         /* chunks / pieces of the call whose select + row offsets + emission ran as ONE kernel   */
    uint32_t tail_fallback;      /*    (k_tail: needs a kept row buffer that holds the worst case); tail_fallback = how   */
                                 /*    often that kernel met tandem-repeat reads and the classic kernels ran instead      */
    uint32_t n_predicted;        /* streamed step: pieces whose candidate count was predicted from the previous call on   */
    uint32_t home_record_bytes;  /*    the same reads (no host round trip between the counting pass and the rest).        */
                                 /* home_record_bytes (po_overlaps_to_host): what crossed PCIe per strand-mirror pair of  */
                                 /*    rows -- 8 or 16 (a verified-candidate record, the host wrote the rows), 0 = the rows */
} po_stats;

/* ExactOverlapper()  -- src/overlapper.cpp:19, py::init at src/phasm.cpp:13. */
po_status po_create(po_handle** out);
void po_destroy(po_handle* h);

/* Choose the HIP device (default 0).  Call it BEFORE adding reads: once the read set is large (64 M bases) the library
 * brings the device up while reads are still being added -- runtime start, stream, a pinned result pool sized from
 * the bases seen so far -- so that the first po_overlaps* call does not pay for it; after that a different device is
 * refused (PO_ERR_INVALID). */
po_status po_set_device(po_handle* h, int device);

/* addSequence(id, seq)  -- src/overlapper.cpp:22-26.  Copies id and seq (the caller may
 * free them at once).  seq is compared byte-wise, like the reference's CharString.  Reads are
 * stored at 2 bits per base; bytes other than upper-case A/C/G/T (N, IUPAC codes, lower case) are
 * kept as sparse exception records beside the 2-bit codes and compared exactly after the packed
 * compare.  Only when such bytes are dense (more than len/64 + 16 in one read) does the whole handle
 * move to the slower 8-bits-per-base representation.                                        */
po_status po_add_sequence(po_handle* h, const char* id, size_t id_len, const char* seq, size_t seq_len);

/* FASTA ingest for `phasm overlap` (the reference uses dinopy.FastaReader / dinopy.reverse_complement,
 * phasm/cli/assembler.py:32-40): the record name is the whole header line, sequence lines are joined,
 * blank lines skipped.  both_strands != 0 adds every record as name+"+" / sequence and name+"-" /
 * reverse complement, exactly what the CLI feeds the overlapper (:38-40). */
po_status po_add_fasta(po_handle* h, const char* path, int both_strands, uint64_t* n_records);

uint32_t po_num_sequences(const po_handle* h);
po_status po_get_id(const po_handle* h, uint32_t idx, const char** id, size_t* id_len);
uint32_t po_get_length(const po_handle* h, uint32_t idx);

/* Pack (if needed) and copy the read set to the device now instead of at the first
 * po_overlaps* call.  Idempotent until the next po_add_sequence.                          */
po_status po_upload(po_handle* h);

/* Sharded upload for multi-GPU jobs.  Every rank needs the whole read set in its HBM (any read can be a `b`), but not
 * over its own PCIe link: rank g copies only the words of shard g's even reads host->device, into its slot of an
 * exchange buffer (po_upload_piece; dst_device == NULL just asks for the piece's length), one all-gather over xGMI hands
 * every rank all pieces (phasm_amd/dist.py: ReadExchange), and po_upload_assemble puts them in place and finishes the
 * upload (odd reads rebuilt on the device, tiles, tables).  *ok = 0: the reads are not (x, reverse complement of x)
 * pairs -- use po_upload.  The reference has no counterpart (one process, reads already in host memory). */
po_status po_upload_piece(po_handle* h, uint32_t shard, uint32_t nshards, void* dst_device, uint64_t capacity_words,
                          uint64_t* word_count, int* ok);
po_status po_upload_assemble(po_handle* h, const void* pieces_device, uint64_t slot_words, uint32_t nshards);
/* The same in `nparts` parts per shard (equal chunks of the shard's piece): the host->device copy of part k + 1 runs while
 * the all-gather of part k is in flight.  The gathered buffer handed to po_upload_assemble_parts is laid out
 * [part][shard][slot_words]. */
po_status po_upload_piece_part(po_handle* h, uint32_t shard, uint32_t nshards, uint32_t part, uint32_t nparts, void* dst_device,
                               uint64_t capacity_words, uint64_t* word_count, int* ok);
po_status po_upload_assemble_parts(po_handle* h, const void* pieces_device, uint64_t slot_words, uint32_t nshards, uint32_t nparts);

/* Forget the device copy of the read set: the next po_upload / po_overlaps* copies the packed reads host->device
 * again, as the first call of a fresh process does.  The reference's overlaps() starts from the host-side string
 * set on every call (index built from `readset`, src/overlapper.cpp:33-36), so ONE reference call corresponds to
 * po_invalidate + po_overlaps_to_host + po_result_rows: the region bench.py times. */
po_status po_invalidate(po_handle* h);

/* overlaps(min_length)  -- src/overlapper.cpp:28-150.  Rows stay on the device until po_result_rows() is called.
 * Same rows on every call, like the reference.  The reference rebuilds its index inside every call (:33-36); this
 * library does so whenever the device copy of the reads has changed (po_add_*, po_invalidate: what a reference call
 * always faces) and otherwise REUSES the anchor index it built for the same upload, min_length and flavour
 * (po_stats.index_reused = 1; PHASM_NO_INDEX_REUSE=1 rebuilds per call).  min_length 0 behaves as 1 (a suffix array
 * has no empty suffix).  A handle also keeps its device workspaces, the pinned result buffers and -- for the streamed
 * form of po_overlaps_to_host -- the candidate counts per piece of the previous call on the same reads.              */
po_status po_overlaps(po_handle* h, uint32_t min_length, po_result** out);

/* overlaps(min_length) the way the reference returns it -- the whole vector<OverlapT> in host memory
 * (src/overlapper.cpp:149) -- as ONE pipelined call: po_overlaps + po_result_rows, with the rows of one chunk of a-side
 * reads travelling device->host (second stream, one page-locked array) while the next chunk is in the kernels.  The
 * result holds the host array only (po_result_rows returns it at once; po_result_device_rows is NULL; po_layout_edges
 * would copy the rows back).  Same multiset of rows as po_overlaps, a-major chunk by chunk.
 * When the read set changed since the last upload -- what the reference faces on every call, its reads are host memory
 * (:22-36) -- the call also does the upload, STREAMED: the packed reads cross PCIe in pieces on a third stream, a piece
 * that has landed is scanned against the whole index (built from every read's first two words, sent ahead), its candidates
 * whose b-side read has arrived are verified and emitted (every suffix-prefix candidate, by the choice of which member
 * of a strand-mirror pair is computed; containments of a read still on its way wait on a list), and its rows travel
 * home while the next piece is still coming in.  Needs reads added as (x, reverse complement of x) pairs of pure
 * upper-case ACGT; otherwise the call uploads first (po_upload) and runs the chunked form.
 * po_stats.streamed tells which form ran. */
po_status po_overlaps_to_host(po_handle* h, uint32_t min_length, po_result** out);

/* Banded seed-extension mode -- an EXTENSION BEYOND THE REFERENCE, which is exact (src/overlapper.cpp:28-150; CLI help
 * "exact overlaps", phasm/cli/assembler.py:436-439).  Same anchors as po_overlaps (b's K-base prefix found in a); every
 * candidate is extended by a banded edit-distance DP (unit costs, diagonals -band..band, band <= 30, one wavefront per
 * candidate: phasm_amd/csrc/extend.hip.h) and accepted with at most max_diff differences:
 *   A  all of a[p:] against a prefix of b  -> row (a, b, p, len(a), 0, bend)
 *   B  all of b against a prefix of a[p:]  -> row (a, b, p, aend, 0, len(b))
 * longest-only for A per ordered pair, every B occurrence, as in the exact contract.  max_diff = 0 returns exactly the
 * rows of po_overlaps (checked against the reference goldens); max_diff > 0 has no reference counterpart -- its checker
 * is the build's own CPU restatement, oracle/extend_oracle.c ("parity unpinned").  Needs pure ACGT (2-bit) or 8-bit reads
 * when max_diff > 0. */
po_status po_overlaps_ex(po_handle* h, uint32_t min_length, uint32_t max_diff, uint32_t band, po_result** out);

/* Multi-GPU form: only the rows whose `a` read lies in shard `shard` of `nshards` (contiguous
 * read-index ranges balanced by base count) are produced.  The union over all shards is
 * exactly the po_overlaps() result; the caller merges (RCCL all-gather in phasm_amd/dist.py). */
po_status po_overlaps_shard(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards,
                            po_result** out);

/* Multi-GPU exchange in compact form.  po_candidates_shard: like po_overlaps_shard, but the result
 * holds the shard's VERIFIED CANDIDATES as po_cand[po_result_count] (16 B each; in paired-strand mode
 * one per strand-mirror pair) instead of rows (24 B each, both members): 3-4x fewer bytes over xGMI.
 * Read it with po_result_device_rows / po_result_copy_to_device (or po_result_rows cast to po_cand*).
 * po_expand: turn a candidate array on this handle's device -- normally the rank-order concatenation
 * of every shard's candidates -- into rows: exactly the po_overlaps() rows (as a multiset; the emission order
 * follows the candidate array).  All-zero entries are padding and skipped (RCCL has no all-gatherv: the
 * shards travel in equal-sized slots); any other entry this library could not have produced is an error. */
typedef struct {
    uint32_t a_idx, p, b_idx, type; /* type bit0: A row (suffix of a = prefix of b), bit1: B row (b inside a) */
} po_cand;
po_status po_candidates_shard(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, po_result** out);
po_status po_expand(po_handle* h, const void* candidates_device, uint64_t n_candidates, po_result** out);
/* po_candidates_shard with the destination supplied: when the shard's candidates fit `capacity` entries they are
 * written straight to dst_device (e.g. this rank's slot of the exchange buffer; *written = 1, the result carries
 * the count and points at dst_device); otherwise *written = 0 and the result holds them as usual. */
po_status po_candidates_shard_into(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards,
                                   void* dst_device, uint64_t capacity, int* written, po_result** out);

/* Sliced wide index: the part of a multi-GPU step that used to be replicated.  Large read sets (> 160 k reads of
 * length >= min_length) use the wide index (W K-mers per read, GBs of table at config 5), and every rank used to build
 * all of it.  Here the table is n_slices sub-tables by key hash; rank g builds sub-table g only (po_index_slice_build),
 * copies it and its chain segment into its slot of an exchange buffer (po_index_slice_export: chunk layout and size
 * from po_index_chunk_bytes), the chunks travel in ONE all-gather (phasm_amd/dist.py: IndexExchange), and the shard
 * call probes the gathered index (po_candidates_shard_indexed) instead of building one.  *is_wide = 0 means this read
 * set uses the narrow index (0.06 ms to build: not worth exchanging) -- call po_candidates_shard as before.  The
 * reference has no counterpart (one process: overlapper.cpp:33-36 builds one suffix array). */
po_status po_index_slice_build(po_handle* h, uint32_t min_length, uint32_t slice, uint32_t n_slices, uint32_t* is_wide,
                               uint32_t* slice_bits, uint64_t* chain_entries);
uint64_t po_index_chunk_bytes(uint32_t slice_bits, uint64_t chain_capacity, uint64_t* chain_offset_bytes);
po_status po_index_slice_export(po_handle* h, void* dst_device, uint64_t chain_capacity);
po_status po_candidates_shard_indexed(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards,
                                      const void* index_device, uint32_t n_slices, uint32_t slice_bits, uint64_t chain_capacity,
                                      void* dst_device, uint64_t capacity, int* written, po_result** out);
po_status po_overlaps_shard_indexed(po_handle* h, uint32_t min_length, uint32_t shard, uint32_t nshards, const void* index_device,
                                    uint32_t n_slices, uint32_t slice_bits, uint64_t chain_capacity, po_result** out);

/* The read-index range [*r_begin, *r_end) that po_overlaps_shard(shard, nshards) scans on the
 * a-side.  Pure host logic (no GPU needed). */
po_status po_shard_range(const po_handle* h, uint32_t shard, uint32_t nshards, uint32_t* r_begin, uint32_t* r_end);

uint64_t po_result_count(const po_result* r);
/* Host pointer to po_result_count() rows (copied device->host on first use, into page-locked memory owned by the
 * library: one DMA, valid until po_result_free); NULL on error.  The counterpart of the reference returning its
 * vector<OverlapT> to the host (src/overlapper.cpp:149). */
const po_row* po_result_rows(po_result* r);
/* Host pointer to rows [first, first + count) only (one rank's share of a merged multi-GPU result: the ranks of a node
 * bring the rows home once between them).  Valid until the next call on the handle; NULL on error or count == 0. */
const po_row* po_result_rows_range(po_result* r, uint64_t first, uint64_t count);
/* Device pointer to the same rows (valid until po_result_free). */
const void* po_result_device_rows(const po_result* r);
/* Copy the rows device->device into dst (>= count*sizeof(po_row) bytes), e.g. a torch tensor. */
po_status po_result_copy_to_device(po_result* r, void* dst_device);
/* The same for the first `count` entries only (count <= po_result_count). */
po_status po_result_copy_prefix_to_device(po_result* r, void* dst_device, uint64_t count);
void po_result_free(po_result* r);

/* Write one GFA2 edge line per row to the file descriptor, byte-identical to the reference's
 * gfa_line("E", "*", a_id, b_id, astart, aend, bstart, bend, "*")  (assembler.py:46-48, gfa.py:230-231). */
po_status po_write_gfa_edges(po_result* r, int fd, uint64_t* lines_out);
/* The segment lines that go before them: `S <name> <length> *` per read pair (assembler.py:38; the handle's reads
 * must have been added as name+"+" / name+"-" pairs). */
po_status po_write_gfa_segments(po_handle* h, int fd, uint64_t* lines_out);
/* (Given a po_layout_edges result instead, the same call writes the graph's edges the way the reference's
 * graph writer does: `E * <u> <v> <weight> <len(u)> 0 <overlap_len> *`, gfa2_write_graph, gfa.py:315-327.) */

/* ---------------------------------------------------------------------------------------------
 * Next row of the path: the consumer of the E lines, stage 1 of
 * `phasm layout` (phasm/cli/assembler.py:52-139) -- classify every alignment, drop contained reads,
 * apply the alignment filters, build the assembly-graph edge list -- on the row array, on the device.
 * --------------------------------------------------------------------------------------------- */

/* The filter settings of `phasm layout` (assembler.py:78-87; CLI defaults :469-489). */
typedef struct {
    uint32_t min_read_length;    /* MinReadLength(n), phasm/filter.py:37-58; 0 = filter not installed  */
    uint32_t min_overlap_length; /* MinOverlapLength(n), filter.py:61-74;   0 = filter not installed  */
    uint32_t max_overhang_abs;   /* MaxOverhang(max_overhang, ratio), filter.py:104-122; default 1000 */
    uint32_t reserved;           /* must be 0                                                         */
    double max_overhang_rel;     /* default 0.8                                                       */
} po_layout_params;

/* One assembly-graph edge: g.add_edge(u, v, {weight, overlap_len}), phasm/assembly_graph.py:146-176.
 * u, v are oriented-read indices of the handle (x+ = 2i, x- = 2i+1; reverse node = index ^ 1). */
typedef struct {
    uint32_t u, v;
    int32_t weight, overlap_len;
} po_edge;

typedef struct {
    uint64_t n_rows;             /* alignments looked at                                              */
    uint64_t n_type[4];          /* rows per AlignmentType (phasm/alignments.py:16-20):               */
                                 /*   0 OVERLAP_AB, 1 OVERLAP_BA, 2 A_CONTAINED, 3 B_CONTAINED         */
    uint64_t n_short;            /* overlap rows with a read shorter than min_read_length             */
    uint64_t n_min_overlap;      /* ... then: shorter than min_overlap_length                         */
    uint64_t n_overhang;         /* ... then: overhang above the MaxOverhang threshold                */
    uint64_t n_pass;             /* overlap rows that satisfy all three predicates                    */
    uint64_t n_contained_reads;  /* reads (both strands count once) that are contained in another     */
    uint64_t n_edges;            /* distinct (u, v) edges of the graph after the contained reads left */
    float ms_classify, ms_dedupe, ms_emit, ms_total;
} po_layout_stats;

/* A read without sequence: one GFA2 segment line `S <name> <length> *` (gfa2_segment_to_read,
 * phasm/io/gfa.py:33-46).  Adds the two oriented nodes name+"+" and name+"-" of that length.  A handle
 * holds either sequences or segments, never both; po_overlaps* on a segment handle fails. */
po_status po_add_segment(po_handle* h, const char* name, size_t name_len, uint32_t length);

/* Wrap rows supplied by the caller (copied) into a result of this handle, e.g. alignments from
 * another producer of the same wire format (phasm/cli/convert.py:65-133). */
po_status po_result_from_rows(po_handle* h, const po_row* rows, uint64_t n, po_result** out);

/* Read a GFA2 file the way `phasm layout` does (assembler.py:56-60, :96-98): pass 1 takes every S line
 * (gfa2_parse_segments, phasm/io/gfa.py:107-109) into an EMPTY handle via po_add_segment, pass 2 turns
 * every E line into a row (gfa2_parse_edge + gfa2_line_to_la, gfa.py:72-104: ids end in the strand
 * character, positions may carry a trailing `$`).  An E line naming an unknown segment fails, as the
 * reference's dict lookup does.  After a failure the handle may already hold some of the segments: destroy it. */
po_status po_add_gfa(po_handle* h, const char* path, uint64_t* n_segments, po_result** rows_out);

/* Stage 1 of `phasm layout` on the rows of `rows` (a result of this handle):
 *   1. classify each row (LocalAlignment.classify, phasm/alignments.py:248-258);
 *   2. ContainedReads (filter.py:77-101): the contained read of every *_CONTAINED row is marked;
 *   3. MinReadLength / MinOverlapLength / MaxOverhang on the remaining rows (filter.py:37-74, 104-122);
 *   4. build_assembly_graph (assembly_graph.py:136-179): two edges per surviving row; a later row
 *      overwrites the attributes of an edge an earlier row added (networkx add_edge);
 *   5. every marked read loses both of its nodes and their edges (assembler.py:113-126).
 * `edges_out` holds po_edge[po_result_count] (read with po_result_rows cast to const po_edge*, or the
 * device pointer), ordered by producing row.  The edge SET equals the reference's `g.edges(data=True)`
 * at "Final graph" (assembler.py:136) for any order of the input lines; the reference's per-filter
 * `filtered` log counters depend on line order and are not reproduced.
 * removed_reads_out (may be NULL): po_num_sequences()/2 bytes, 1 = read i (nodes 2i, 2i+1) was contained.
 * Needs the handle's ids in strand pairs (2i = name+"+", 2i+1 = name+"-", what po_add_fasta with
 * both_strands, the CLI and po_add_segment produce); otherwise PO_ERR_INVALID. */
po_status po_layout_edges(po_handle* h, po_result* rows, const po_layout_params* params,
                          uint8_t* removed_reads_out, po_result** edges_out);
po_status po_get_layout_stats(const po_handle* h, po_layout_stats* out);

/* Diagnostics for the test suite (no reference counterpart: addSequence copies its argument and
 * never touches it again, src/overlapper.cpp:22-26 -- these two calls let a test PROVE that of this library).
 * po_debug_host_ranges: every range of host memory the library has made visible to the GPU in this process, ever
 * (hipHostMalloc'ed landing zones and result arrays, the hipHostRegister'ed packed read stores): `out` receives up to
 * cap_entries triplets {base, bytes, kind} (kind & 0xFF: 1 = page-locked allocation, 2 = registered store; bit 8: still
 * live); returns the number of entries on the list.  Device->host copies and host-mapped stores of the library can
 * only land inside these ranges.
 * po_debug_pointer_info: what the HIP runtime (hipPointerGetAttributes -> *hip_type, -1 = unknown to it) and the ROCr
 * runtime underneath (hsa_amd_pointer_info -> *hsa_type: 0 unknown, 1 runtime allocation, 2 locked / registered host
 * memory; -1 = not queried) know about address p, and the range they know it as.  Returns 1 if either knows it -- i.e.
 * the GPU can address that page -- else 0.
 * po_debug_fault_backtrace: install a SIGSEGV / SIGBUS handler that writes the faulting address and the native stack of
 * the faulting thread to `fd`, then runs whatever handler was installed before it (a store into a read-only input mapping
 * by a thread that has no Python frames -- a runtime thread -- is named this way).  Returns 0 on success. */
uint64_t po_debug_host_ranges(uint64_t* out, uint64_t cap_entries);
/* po_debug_store_words: the packed host store `store` (0: reads of even index, 1: odd) as it stands -- *words points into the
 * handle (valid until the next call that adds reads), returns the number of 8-byte words.  Tests compare the stores the
 * parallel FASTA ingest writes with the ones po_add_sequence builds. */
uint64_t po_debug_store_words(const po_handle* h, int store, const uint64_t** words);
/* The host half of po_overlaps_to_host's compact row transfer on its own, for tests without a GPU: the rows of `n`
 * verified-candidate records (po_cand; one per strand-mirror pair when paired != 0), written by the library's helper
 * threads in the order po_overlaps emits them -- A row, [its mirror], B row, [its mirror] per record (row fields:
 * src/overlapper.cpp:77-82,104-110; mirror rules: SURVEY.md section 8c).  0 = ok; 1 = the rows differ in number from
 * n_rows_expected (nothing reliable was written); 2 = a record names a read >= n_reads; -1 = no helper thread. */
int po_debug_expand_records(const po_cand* records, uint64_t n, const uint32_t* lengths, uint32_t n_reads, uint32_t paired,
                            po_row* rows_out, uint64_t n_rows_expected);
/* ... and of `n` records of EIGHT bytes, a | b << sh_b | p << sh_p | type << 62 (what po_overlaps_to_host sends when the
 * read set fits: 2 ceil(log2 reads) + ceil(log2 (longest read + 1)) <= 62); -2 = shifts out of range. */
int po_debug_expand_packed(const uint64_t* records, uint64_t n, uint32_t sh_b, uint32_t sh_p, const uint32_t* lengths, uint32_t n_reads,
                           uint32_t paired, po_row* rows_out, uint64_t n_rows_expected);
int po_debug_fault_backtrace(int fd);
int po_debug_pointer_info(const void* p, int32_t* hip_type, int32_t* hsa_type, uint64_t* base, uint64_t* bytes);

po_status po_get_stats(const po_handle* h, po_stats* out);
const char* po_last_error(const po_handle* h);
int po_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PHASM_OVERLAP_H */
