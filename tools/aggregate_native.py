"""bench.py's `aggregate` block on its own: N depth-20 signals -> gl355_aggregate_units -> BN254 wrap, cold (artifacts from disk) and warm"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
import bench  # noqa: E402

gl = importlib.import_module("stark-verifier_amd")
print(json.dumps(bench.aggregate_figure(gl, 0, n_ctx=int(sys.argv[1]) if len(sys.argv) > 1 else 8)))
