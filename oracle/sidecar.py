"""TEST INFRASTRUCTURE ONLY -- the checkers in a process of their own.

The GPU parity tests run the product (libphasm_overlap.so + the HIP runtime + torch) in the pytest
process.  The things they are checked against -- ``oracle/overlap_oracle.c`` through ctypes, the numpy
layout restatement, the compiled reference binary, hipcc builds of library variants -- run HERE, in a
child that is started before the pytest process has touched the GPU:

* the checker shares no heap, no threads and no HIP runtime with the thing it checks (a checker that
  lives in the address space of the code under test can be damaged by it, and then proves nothing);
* the GPU process never forks: every ``subprocess`` the tests need is started by this child.

Protocol on stdin/stdout: 8-byte little-endian length + pickle.  Request ``(module, function, args, kwargs)``
-> reply ``("ok", value)`` or ``("err", traceback text)``.  ``module == "__run__"`` runs a command
(``subprocess.run``) and returns ``(returncode, stdout, stderr)``.
"""
from __future__ import annotations

import importlib
import os
import pickle
import struct
import subprocess
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_exact(f, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = f.read(n - len(buf))
        if not chunk:
            raise EOFError
        buf += chunk
    return bytes(buf)


def _send(f, obj) -> None:
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(data)))
    f.write(data)
    f.flush()


def _recv(f):
    (n,) = struct.unpack("<Q", _read_exact(f, 8))
    return pickle.loads(_read_exact(f, n))


def serve() -> int:
    """Child side.  fd 0 / fd 1 carry the protocol; anything the checkers print goes to stderr."""
    fin = os.fdopen(os.dup(0), "rb")
    fout = os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    while True:
        try:
            module, func, args, kwargs = _recv(fin)
        except EOFError:
            return 0
        try:
            if module == "__run__":
                p = subprocess.run(*args, **kwargs)
                out = (p.returncode, p.stdout, p.stderr)
            elif module == "__ping__":
                out = os.getpid()
            else:
                out = getattr(importlib.import_module(module), func)(*args, **kwargs)
            _send(fout, ("ok", out))
        except BaseException:  # noqa: BLE001 -- the parent gets the traceback, the server carries on
            _send(fout, ("err", traceback.format_exc()))


class Sidecar:
    """Parent side: ``call(module, function, *args)`` runs in the child."""

    def __init__(self):
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        # the child never uses the GPU
        env["HIP_VISIBLE_DEVICES"] = ""
        env["ROCR_VISIBLE_DEVICES"] = ""
        self.proc = subprocess.Popen([sys.executable, "-u", "-m", "oracle.sidecar"], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, cwd=ROOT, env=env)
        self.pid = self.call("__ping__", "")

    def call(self, module: str, func: str, *args, **kwargs):
        _send(self.proc.stdin, (module, func, args, kwargs))
        status, value = _recv(self.proc.stdout)
        if status != "ok":
            raise RuntimeError("checker process failed:\n" + value)
        return value

    def run(self, argv, **kwargs):
        """``subprocess.run(argv, **kwargs)`` in the child; returns (returncode, stdout, stderr)."""
        kwargs.setdefault("env", dict(os.environ))  # the caller's environment, not the child's (which hides the GPU)
        return self.call("__run__", "", argv, **kwargs)

    def close(self) -> None:
        if self.proc and self.proc.poll() is None:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=10)
            except Exception:  # noqa: BLE001
                self.proc.kill()
        self.proc = None


if __name__ == "__main__":
    sys.exit(serve())
