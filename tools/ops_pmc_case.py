"""A few calls of one ops-table entry ($OPS_ENTRY) for tools/pmc.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
sys.argv = ["bench.py", "--config", "ops", "--ops", os.environ.get("OPS_ENTRY", "convi_3x3_u8"), "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-verify"]
import bench
bench.main()
