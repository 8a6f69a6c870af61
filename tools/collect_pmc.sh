#!/bin/bash
# Hardware-counter passes behind profiles/traffic.json (run on the GPU box):
#
#     tools/collect_pmc.sh gpurun_out/pmc_r02 && python3 tools/pmc_to_traffic.py gpurun_out/pmc_r02 profiles/r02_pmc
#
# Every pass is its own `rocprofv3 --pmc ...` run of the SAME workload (config 2, three calls), counters only -- never
# combined with a trace domain -- with the program itself behind `--`.  FETCH_SIZE and WRITE_SIZE do not fit one pass
# (MI355X_MICROARCH.md, rocprofv3 PMC slots); SQ counters go four at a time.
set -uo pipefail
OUT="${1:?output directory}"
CFG="${2:-cfg2}"     # cfg3: the wide index's kernels (k_wide_scan both passes, k_wide_insert, k_wide_chain_fill); the DP and
                     # layout passes are config-2 / config-4 only
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
export TMPDIR=/tmp
run_pass() {   # name, program args..., then counters after '--pmc--'
    local name="$1"; shift
    local prog=(); while [ "$1" != "--pmc--" ]; do prog+=("$1"); shift; done; shift
    echo "== pass $name: $*" >&2
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "${prog[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)" >&2
}
P="$ROOT/tools/perf_probe.py"
L="$ROOT/tools/layout_probe.py"
run_pass fetch  "$P" --config "$CFG" --iters 3 --pmc-- FETCH_SIZE
run_pass write  "$P" --config "$CFG" --iters 3 --pmc-- WRITE_SIZE
run_pass sq1    "$P" --config "$CFG" --iters 3 --pmc-- SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
run_pass sq2    "$P" --config "$CFG" --iters 3 --pmc-- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES
run_pass sq3    "$P" --config "$CFG" --iters 3 --pmc-- SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY
run_pass sq4    "$P" --config "$CFG" --iters 3 --pmc-- SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
run_pass tcc    "$P" --config "$CFG" --iters 3 --pmc-- TCC_HIT_sum TCC_MISS_sum
if [ "$CFG" != cfg2 ]; then echo done >&2; exit 0; fi
# config 4 through the banded DP (po_overlaps_ex): where does k_extend_dp spend its cycles?
# (both mappings: the lane-per-candidate kernel is the default, PHASM_DP_KERNEL=wave selects the wave-per-candidate one)
run_pass dp_sq1 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 --pmc-- SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
run_pass dp_sq2 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 --pmc-- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES
export PHASM_DP_KERNEL=wave
run_pass dp_wave_sq1 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 --pmc-- SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
run_pass dp_wave_sq2 "$P" --config cfg4 --iters 1 --max-diff 400 --band 8 --pmc-- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES
unset PHASM_DP_KERNEL
run_pass lfetch "$L" --config cfg2 --iters 3 --pmc-- FETCH_SIZE
run_pass lwrite "$L" --config cfg2 --iters 3 --pmc-- WRITE_SIZE
echo done >&2
