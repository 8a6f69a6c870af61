"""Overlay package: put this directory on ``sys.path`` *before* an installed PHASM to make
``from phasm.overlapper import ExactOverlapper`` (phasm/cli/assembler.py:15) resolve to the MI355X
library.  In a real PHASM checkout only ``overlapper.py`` is copied into ``phasm/``."""
