#!/usr/bin/env python3
"""Builds hpg-variant_amd/lib/ablation/libhpgv.so: the engine with -DHPGV_ABLATION, i.e. with every kernel form that lost an
A/B comparison compiled in and selectable through hpgv_set_option (include/hpgv.h "Options": the unpipelined / persistent /
other-unroll association scans, Fisher passes of 8 / 32 / 64 lanes per variant, the lane-per-block and one-symbol-per-round
bgzip decoders, the one-sweep and the line-by-line tokenizers) and the occupancy experiments' environment switches.  The shipped
library holds one form of each kernel.  Use: HPGV_LIB=hpg-variant_amd/lib/ablation/libhpgv.so python tools/<tool>.py, or the
GPU suite over those forms: HPGV_LIB=... python -m pytest tests -m gpu."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
b = importlib.import_module("hpg-variant_amd._build")
print(b.build_device_lib(force="--force" in sys.argv, verbose=True, ablation=True))
