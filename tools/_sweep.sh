timeout -k 5 500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/t_par.log 2>&1; echo rc=$?; tail -2 gpurun_out/t_par.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err; python -c "
import json; d=json.load(open('gpurun_out/b_x.json')); print(d['ms_per_step'], d['rows_per_step'], d.get('stage_ms'))"
