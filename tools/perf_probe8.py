#!/usr/bin/env python3
"""Developer probe: config-2-shaped reads with a few N bases -> 8-bit path timings."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phasm_amd import synth
from phasm_amd.overlapper import ExactOverlapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = synth.scaled(synth.CONFIGS["cfg2"], n)
reads = synth.oriented(synth.generate_reads(cfg))
ov = ExactOverlapper()
for k, (name, seq) in enumerate(reads):
    if k == 10:
        seq = seq[:100] + b"N" + seq[101:]
    ov.add_sequence(name, seq)
for it in range(3):
    res = ov.overlaps_result(1000); st = ov.stats(); res.free()
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}))
