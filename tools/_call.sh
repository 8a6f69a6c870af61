mkdir -p gpurun_out/r04q
PHASM_HOME_COPY_WGS=16 PHASM_UP_COPY_WGS=32 python -m pytest tests/test_gpu_streamed.py -x -q > gpurun_out/r04q/tests_copy.log 2>&1; rc=$?; tail -3 gpurun_out/r04q/tests_copy.log
[ $rc -eq 0 ] || exit $rc
python tools/ab_probe.py "" "PHASM_HOME_COPY_WGS=4" "PHASM_HOME_COPY_WGS=16" "PHASM_HOME_COPY_WGS=64" "PHASM_HOME_COPY_WGS=256" --rounds 3 > gpurun_out/r04q/ab_home_copy.log 2>&1; grep median gpurun_out/r04q/ab_home_copy.log
python tools/ab_probe.py "" "PHASM_UP_COPY_WGS=16" "PHASM_UP_COPY_WGS=64" "PHASM_UP_COPY_WGS=256" "PHASM_HOME_COPY_WGS=16 PHASM_UP_COPY_WGS=64" --rounds 3 > gpurun_out/r04q/ab_up_copy.log 2>&1; grep median gpurun_out/r04q/ab_up_copy.log
tools/collect_pmc.sh gpurun_out/r04q/pmc_cfg3 cfg3 > gpurun_out/r04q/pmc_cfg3.log 2>&1; echo pmc done
python tools/perf_probe.py --config cfg5 --iters 2 --windows 1,16 > gpurun_out/r04q/cfg5_windows.log 2>&1; grep window gpurun_out/r04q/cfg5_windows.log
