#!/usr/bin/env python3
"""Runs one seed of the differential tester several times and prints what the case is and what it finds."""
import os
import sys

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from fuzz_plans import Case, run_seed

seed = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = Case(seed, 2_600_000)
print(c.describe() if hasattr(c, "describe") else c.__dict__.keys())
for k in range(reps):
    try:
        run_seed(seed, 2_600_000)
        print("rep", k, "ok")
    except AssertionError as e:
        print("rep", k, "FAIL", str(e)[:300])
