"""phasm_amd -- MI355X-native replacement for PHASM's all-pairs exact read-overlap step.

Only the overlap hot path lives here: the ``ExactOverlapper`` drop-in
(:mod:`phasm_amd.overlapper`), the ``overlap`` command (:mod:`phasm_amd.cli`), the GFA2 line
emitter (:mod:`phasm_amd.io.gfa`) and the multi-GPU shard/merge helper (:mod:`phasm_amd.dist`).
The compute is hand-written HIP behind the C ABI in ``include/phasm_overlap.h``.
"""
__version__ = "0.1.0"
