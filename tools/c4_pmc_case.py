"""One C4 step of 64 images (one launch of the fused kernel) for tools/pmc.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
sys.argv = ["bench.py", "--config", "c4", "--images", os.environ.get("C4_IMAGES", "64"), "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-verify"]
import bench
bench.main()
