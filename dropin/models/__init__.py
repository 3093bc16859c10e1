"""`from models import rfnet, mmformer, m2ftrans` (reference code/train.py:19) -> passion_amd.models.*"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _alias import alias  # noqa: E402

alias(__name__, __path__, "passion_amd.models", ["mmformer", "rfnet", "m2ftrans", "blocks"])
