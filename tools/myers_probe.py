"""bench.py's Myers section alone (batch of short pairs, the ccheck-sized pair, the reference beside them)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import mia_amd  # noqa: E402

hip = mia_amd.MiaHip(0)
out = bench.section_myers(hip, no_cpu="--no-cpu" in sys.argv)
hip.close()
print(json.dumps({k: v for k, v in out.items() if k.startswith("pair_") or k in ("kernel_ms", "kernel_pairs_per_s", "cpu_baseline")}, indent=1))
