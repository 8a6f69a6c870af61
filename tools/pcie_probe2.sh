#!/bin/bash
# Developer probe: which engine moves a device->host copy -- under /opt/rocm's HIP runtime and under the one PyTorch
# bundles (the one a Python process of this package runs on, phasm_amd/_lib.py)?  A blit kernel shows up in the kernel trace.
# The runtime is chosen INSIDE the probed program (two binaries, one linked with an rpath to torch's libamdhip64):
# rocprofv3 execs the program directly with a clean environment -- nothing is preloaded into the launcher, which would
# initialise the GPU there and make the launch an exec out of a GPU-initialised process.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="${1:-/tmp/pcie_probe_out}"
mkdir -p "$OUT"
T="$(python3 -c "import importlib.util,os; print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")"
hipcc --offload-arch=gfx950 -O2 "$ROOT/tools/pcie_probe.hip" -o "$OUT/pcie_probe_rocm" || exit 1
hipcc --offload-arch=gfx950 -O2 "$ROOT/tools/pcie_probe.hip" -o "$OUT/pcie_probe_torch" -L"$T" -Wl,-rpath,"$T" || exit 1
export TMPDIR=/tmp
for prog in pcie_probe_rocm pcie_probe_torch; do
  for knob in "" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2"; do
    rm -rf "$OUT/pp"
    if [ -n "$knob" ]; then export "$knob"; fi
    (cd "$OUT" && rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/pp" -o run --output-format csv -- "$OUT/$prog" > "$OUT/pp.log" 2>&1) || true
    if [ -n "$knob" ]; then unset "${knob%%=*}"; fi
    echo "== $prog $knob"
    cut -d, -f2 "$OUT"/pp/*memory_copy_trace.csv 2>/dev/null | sort | uniq -c | grep -v Direction
    grep "rep 2\|pieces" "$OUT/pp.log" | head -3
    grep -h copyBuffer "$OUT"/pp/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | head -2
    [ "$prog" = "pcie_probe_rocm" ] && break
  done
done
