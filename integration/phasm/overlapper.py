"""``phasm.overlapper`` -- the module name the reference imports (phasm/cli/assembler.py:15), backed
by libphasm_overlap.so instead of the pybind11 extension of setup.py:44-70."""
from phasm_amd.overlapper import ExactOverlapper  # noqa: F401

__all__ = ["ExactOverlapper"]
