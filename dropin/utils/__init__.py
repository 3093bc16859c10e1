"""`from utils import criterions` / `from utils.predict import ...` (reference code/train.py:23-24) -> passion_amd.utils.*;
utils.lr_scheduler / utils.parser / utils.str2bool stay the reference's own files"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _alias import alias  # noqa: E402

alias(__name__, __path__, "passion_amd.utils", ["criterions", "predict"])
