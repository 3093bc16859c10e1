"""`from options import args_parser` (reference code/train.py:22): the same flags and derived fields, passion_amd.options"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passion_amd.options import args_parser, build_parser  # noqa: E402,F401
