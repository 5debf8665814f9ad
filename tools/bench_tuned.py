#!/usr/bin/env python3
"""bench.py with library tuning keys set first.  usage: tools/bench_tuned.py key=value [key=value ...] -- <bench.py arguments>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (the HIP runtime is initialised through torch first, as in bench.py)
from maskflownet_amd import _lib
args = sys.argv[1:]
rest = args[args.index("--") + 1:] if "--" in args else []
for kv in (args[:args.index("--")] if "--" in args else args):
    k, v = kv.split("=")
    _lib.set_tuning(**{k: int(v)})
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
