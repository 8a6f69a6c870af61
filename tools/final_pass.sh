set -o pipefail
mkdir -p gpurun_out/r2s
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2s/gpu_tests.log 2>&1 || { tail -20 gpurun_out/r2s/gpu_tests.log; exit 1; }
PHASM_POISON=0xA5 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streamed.py tests/test_gpu_extend.py tests/test_gpu_layout.py -x -q -m gpu > gpurun_out/r2s/gpu_tests_poison.log 2>&1 || { tail -20 gpurun_out/r2s/gpu_tests_poison.log; exit 1; }
timeout -k 10 300 python tools/ubench.py > gpurun_out/r2s/ubench.json 2> gpurun_out/r2s/ubench.err || exit 1
timeout -k 10 900 tools/collect_pmc.sh gpurun_out/r2s/pmc 2> gpurun_out/r2s/collect.err
python3 tools/pmc_to_traffic.py gpurun_out/r2s/pmc gpurun_out/r2s/pmc_profiles > gpurun_out/r2s/to_traffic.log 2>&1 || exit 1
cp profiles/traffic.json gpurun_out/r2s/traffic.json
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r2s/bench.json 2> gpurun_out/r2s/bench.err || { tail -5 gpurun_out/r2s/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2s/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cfg4 --no-tuples --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2s/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2s/prof.err
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/r2s/gpu_tests.log; tail -2 gpurun_out/r2s/gpu_tests_poison.log
ls gpurun_out/r2s/prof/* | head
