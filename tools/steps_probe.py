import os, sys, time
sys.path.insert(0, ".")
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper
cfg = synth.CONFIGS[sys.argv[1]]
ov = ExactOverlapper(device=0)
for name, seq in synth.oriented(synth.generate_reads(cfg)):
    ov.add_sequence(name, seq)
for it in range(int(sys.argv[2])):
    t = time.perf_counter()
    ov.invalidate()
    res = ov.overlaps_to_host_result(1000)
    n = len(res.rows_view())
    dt = (time.perf_counter() - t) * 1e3
    st = ov.stats()
    res.free()
    print("step %d: %.2f ms rows %d" % (it, dt, n), {k: st[k] for k in ("streamed", "n_predicted", "fused_tail", "tail_fallback", "home_record_bytes", "n_candidates", "wide_index", "ms_upload")}, flush=True)
ov.close()
