#!/usr/bin/env python3
"""bench.py's secondary.mixed_cover on its own, optionally after torch has initialised the device: tools/mixed_in_bench.py [torch]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if len(sys.argv) > 1:
    import torch
    x = torch.zeros(1024, device="cuda:0")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
r = bench.bench_mixed_cover(0, 200)
for name, rec in r.items():
    print(name, {k: round(v["round_us"], 1) for k, v in rec.items() if isinstance(v, dict)})
