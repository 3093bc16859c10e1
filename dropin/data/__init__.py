"""`from data.datasets_nii import ...` / `from data.transforms import *` (reference code/train.py:15-18) -> passion_amd.data.*
(GPU input pipeline); data.data_utils stays the reference's own file"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _alias import alias  # noqa: E402

alias(__name__, __path__, "passion_amd.data", ["datasets_nii", "transforms"])
