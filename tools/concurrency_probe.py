#!/usr/bin/env python3
"""Developer probe: do two half-jobs on two handles (two HIP streams) overlap on one GPU?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper
cfg = synth.CONFIGS["cfg2"]
reads = synth.oriented(synth.generate_reads(cfg))
hs = []
for k in range(2):
    ov = ExactOverlapper()
    for n, s in reads:
        ov.add_sequence(n, s)
    ov.upload()
    hs.append(ov)
def run(ov, shard, ns, reps, out):
    t = time.perf_counter()
    for _ in range(reps):
        r = ov.overlaps_result(1000, shard, ns); r.free()
    out.append(time.perf_counter() - t)
# warmup
for ov in hs:
    run(ov, 0, 1, 2, [])
reps = 10
o = []; run(hs[0], 0, 1, reps, o); print("whole job, one handle: %.2f ms/step" % (o[0] / reps * 1e3))
o = []; run(hs[0], 0, 2, reps, o); run(hs[0], 1, 2, reps, o); print("two half shards sequentially: %.2f ms per pair" % (sum(o) / reps * 1e3))
o = []
ts = [threading.Thread(target=run, args=(hs[k], k, 2, reps, o)) for k in range(2)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print("two half shards concurrently (2 streams): %.2f ms per pair" % ((time.perf_counter() - t0) / reps * 1e3))
o = []
ts = [threading.Thread(target=run, args=(hs[k], 0, 1, reps, o)) for k in range(2)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print("two WHOLE jobs concurrently: %.2f ms per pair (%.2f per job)" % ((time.perf_counter() - t0) / reps * 1e3, (time.perf_counter() - t0) / reps * 1e3 / 2))
