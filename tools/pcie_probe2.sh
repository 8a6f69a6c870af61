#!/bin/bash
# Developer probe: which engine moves a device->host copy -- under /opt/rocm's HIP runtime and under the one PyTorch
# bundles (the one a Python process of this package runs on, phasm_amd/_lib.py)?  A blit kernel shows up in the kernel trace.
set -e
hipcc --offload-arch=gfx950 -O2 $GRAFT_REPO_ROOT/tools/pcie_probe.hip -o /tmp/pcie_probe
T=$(python3 -c "import importlib.util,os; print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
cd /tmp && export TMPDIR=/tmp
for setting in "X=1" "LD_PRELOAD=$T/libamdhip64.so" ; do
  for knob in "Y=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2" "GPU_BLIT_ENGINE_TYPE=0"; do
  rm -rf /tmp/pp
  env $setting $knob rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/pp -o run --output-format csv -- /tmp/pcie_probe > /tmp/pp.log 2>&1 || true
  echo "== $setting $knob"
  cut -d, -f2 /tmp/pp/*memory_copy_trace.csv 2>/dev/null | sort | uniq -c | grep -v Direction
  grep "rep 2\|pieces" /tmp/pp.log | head -3
  grep -h copyBuffer /tmp/pp/*kernel_stats.csv | cut -d, -f1-4 | head -2
  [ "$setting" = "X=1" ] && break
  done
done
