#!/usr/bin/env python3
"""BASELINE config 2 (8x8, 4-in-row, n_playout 200, simple 6-conv net, 64 concurrent games) for a fixed number of
scheduler steps -- the program to put behind `rocprofv3 --kernel-trace --stats -- python3 tools/config2_run.py`."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import config_table as ct  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500      # [steps [lanes [pipeline groups [trunk_arith]]]]
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else 2
arith = sys.argv[4] if len(sys.argv) > 4 else "auto"
print(json.dumps(ct.gpu_config("simple", 8, 4, 200, 64, steps, lanes=lanes, pipeline=pipeline, trunk_arith=arith)))
