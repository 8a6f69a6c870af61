/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference overlapper's output contract.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (phasm_amd/) never links, imports or calls anything in oracle/.          */
#ifndef OVERLAP_ORACLE_H
#define OVERLAP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 24-byte row as include/phasm_overlap.h (po_row); mirrors OverlapT,
 * /root/reference/src/overlapper.h:17, with the two id strings replaced by read indices. */
typedef struct {
    uint32_t a_idx, b_idx;
    int32_t astart, aend, bstart, bend;
} oracle_row;

/* reads: n byte strings seqs[i] of length lens[i] (insertion order = read index).
 * Returns 0 and a malloc'd row array in *rows_out (caller frees with oracle_free), or -1 on OOM. */
int oracle_overlaps(const uint8_t* const* seqs, const uint32_t* lens, uint32_t n,
                    uint32_t min_length, oracle_row** rows_out, uint64_t* nrows_out);

/* Same, but over one concatenated buffer: read i = cat[offs[i] .. offs[i]+lens[i]). */
int oracle_overlaps_cat(const uint8_t* cat, const uint64_t* offs, const uint32_t* lens, uint32_t n,
                        uint32_t min_length, oracle_row** rows_out, uint64_t* nrows_out);

void oracle_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
