#!/usr/bin/env python3
"""bench.py's module end-to-end entry alone (reduce_hip from the libvips module on a host image)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(gpus=1)
ctx = bench.Ctx(args)
print(json.dumps(bench.run_module_e2e(ctx), indent=1))
