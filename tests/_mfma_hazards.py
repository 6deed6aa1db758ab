"""The MFMA-result hazard scan lives next to build.py (flash-attention-turing_amd/mfma_hazards.py: the build itself runs it); re-exported for the tests."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention-turing_amd"))
from mfma_hazards import NEED, scan_file, scan_kernel  # noqa: E402,F401
