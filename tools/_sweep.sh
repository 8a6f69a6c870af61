timeout -k 5 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo rc=$?; tail -2 gpurun_out/t_all.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b_x.json 2> gpurun_out/b_x.err; python -c "
import json; d=json.load(open('gpurun_out/b_x.json')); print(d['ms_per_step'], d['rows_per_step'], d.get('stage_ms')); print(d['layout_stage1']['ms_per_call'], d['layout_stage1']['n_edges'])"
