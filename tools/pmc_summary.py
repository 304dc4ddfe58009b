#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one file per counter pass; FETCH_SIZE and
WRITE_SIZE need separate passes on gfx950).
usage: pmc_summary.py out.json reads_per_gpu pass1_counter_collection.csv [...]
Values are per dispatch, averaged over the LAST 40 % of the kernel's dispatches in time order: `bench.py --pmc-run` does
2 warm-up iterations (the first against mt311 itself, whose ambiguity codes send every read through the full-window
kernels), 4 timed and 4 instrumented ones -- the last four are the steady state the stage table describes.  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports
them, uncorrected; bench.py applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x 2) and this file records
what k_peak_copy -- a streaming copy of a known size in the same passes -- showed, as the calibration of that factor.
_meta.source_hash names the build (sha256 over csrc/): bench.py does not replay counters of another build."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = name.split("(")[0].replace("void ", "").strip()
    name = re.sub(r"<.*>", "", name)
    return name.split("::")[-1]


def main():
    out = defaultdict(dict)
    for path in sys.argv[3:]:
        per = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for name, ctrs in per.items():
            for c, vals3 in ctrs.items():
                vals3.sort()
                k = max(1, int(round(len(vals3) * 0.4)))
                keep = [(v, t) for _, v, t in vals3[-k:]]
                out[name][c] = sum(v for v, _ in keep) / len(keep)
                out[name]["dispatches_" + c] = len(keep)
                out[name]["kernel_ms_under_" + c] = sum(t for _, t in keep) / len(keep) / 1e6
    import bench
    meta = {"source_hash": bench.source_hash(), "reads_per_gpu": int(sys.argv[2]), "steps_kept": 4,
            "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --pmc-run <tag> --config <k> (tools/pmc_collect.sh)"}
    pc = out.get("k_peak_copy")
    if pc and "FETCH_SIZE" in pc:
        copied = float(1 << 28)                  # bench.py --pmc-run: measure_peaks(1 << 28), bytes read = bytes written
        meta["fetch_calibration"] = {"k_peak_copy_bytes_read": copied, "FETCH_SIZE_bytes": pc["FETCH_SIZE"] * 1024,
                                     "ratio": pc["FETCH_SIZE"] * 1024 / copied,
                                     "WRITE_SIZE_bytes": pc.get("WRITE_SIZE", 0) * 1024, "write_ratio": pc.get("WRITE_SIZE", 0) * 1024 / copied}
    out["_meta"] = meta
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name in sorted((k for k in out if k != "_meta"), key=lambda k: -out[k].get("kernel_ms_under_FETCH_SIZE", 0))[:14]:
        print(name, {k: round(v, 2) for k, v in out[name].items() if not k.startswith("dispatches")})
    print(meta)


if __name__ == "__main__":
    main()
