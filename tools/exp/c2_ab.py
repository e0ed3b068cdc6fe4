#!/usr/bin/env python3
"""C2 (100 M rows x 8 columns) step time, many steps: for A/B runs of an environment switch.

    TGX_FORM_MEMORY=1 python tools/exp/c2_ab.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import term_amd as T
from term_amd import synth
from term_amd._lib import spec
from secondary_bench import run_c2

T.init(distinct_capacity_hint=1 << 20)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
r = run_c2(T, torch, synth, spec, steps, 5, 0x7E570004)
print(json.dumps({k: r[k] for k in ("ms_per_step", "ms_min", "frac_of_8TBs", "verified") if k in r}))
