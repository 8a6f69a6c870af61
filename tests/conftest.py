import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The checker process (oracle/sidecar.py: C oracle, layout restatement, reference binary, every child command)
    # starts HERE, before anything in this process initialises the GPU: the code under test never shares a heap
    # with what checks it, and a process that holds the GPU never forks.
    if "not gpu" not in (config.getoption("-m", default="") or ""):
        import checker
        checker.start()
        # a mismatch of a GPU run leaves its evidence behind (rows of both sides and the reads: tests/checker.py;
        # a read that changed under a call: tests/test_gpu_parity.py) -- gpurun_out/ travels back from the GPU box
        os.environ.setdefault("PHASM_MISMATCH_DIR", os.path.join(ROOT, "gpurun_out", "mismatch"))
        # a CPU store into a read-only input mapping (tests/checker.py GuardedReads) -- or any other fault -- ends the run with
        # every thread's Python stack: on stderr, and in a file that travels back from the GPU box
        import faulthandler
        global _fault_log
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out", "faults"), exist_ok=True)
            _fault_log = open(os.path.join(ROOT, "gpurun_out", "faults", "fault_%d.log" % os.getpid()), "w")
            faulthandler.enable(file=_fault_log, all_threads=True)
        except OSError:
            faulthandler.enable(all_threads=True)
        try:   # ... and the NATIVE stack of the faulting thread (a runtime thread has no Python frames), same file, first
            from phasm_amd import _lib
            _lib.load().po_debug_fault_backtrace(_fault_log.fileno() if _fault_log is not None else 2)
        except Exception:  # noqa: BLE001 -- the library may not be built in a CPU-only session that never loads it
            pass


_fault_log = None


def pytest_unconfigure(config):
    import checker
    checker.stop()
    if checker.GUARD_TALLY["guards"]:
        import json
        try:
            ranges = checker.host_ranges()
            checker.GUARD_TALLY["host_ranges_ever_pinned_by_the_library"] = len(ranges)
            checker.GUARD_TALLY["host_ranges_live_at_the_end"] = sum(1 for r in ranges if r[3])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "guard_summary.json"), "w") as fh:
                json.dump(checker.GUARD_TALLY, fh, indent=1)
        except Exception:  # noqa: BLE001
            pass
    global _fault_log
    if _fault_log is not None:   # (no fault: no file)
        import faulthandler
        faulthandler.disable()
        name = _fault_log.name
        _fault_log.close()
        _fault_log = None
        try:
            if os.path.getsize(name) == 0:
                os.remove(name)
        except OSError:
            pass


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a hung kernel or collective must end the run with a traceback, not sit there: per-test wall-clock limit
    # (pytest-timeout, "thread" method: it can end a process that is stuck inside a native call)
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600, method="thread"))
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
