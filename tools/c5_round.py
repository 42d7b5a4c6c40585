#!/usr/bin/env python3
"""The QCQP round of bench.py's secondary.c5 on its own (profiling): tools/c5_round.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

print(json.dumps(bench.bench_c5(0, int(sys.argv[1]) if len(sys.argv) > 1 else 50), indent=1))
