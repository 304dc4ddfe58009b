#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one file per counter pass; FETCH_SIZE and
WRITE_SIZE need separate passes on gfx950).  usage: pmc_summary.py out.json pass1_counter_collection.csv [...]
Values are per dispatch, averaged over the dispatches of the kernel that processed at least half as much as the largest
one (the timed steps; warm-up and tiny launches are left out).  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them."""
import csv
import json
import sys
from collections import defaultdict


def main():
    out = defaultdict(dict)
    for path in sys.argv[2:]:
        per = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[name][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for name, ctrs in per.items():
            for c, vals in ctrs.items():
                big = max(v for v, _ in vals)
                keep = [(v, t) for v, t in vals if v >= 0.5 * big] or vals
                out[name][c] = sum(v for v, _ in keep) / len(keep)
                out[name]["dispatches_" + c] = len(keep)
                out[name]["kernel_ms_under_" + c] = sum(t for _, t in keep) / len(keep) / 1e6
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name in sorted(out, key=lambda k: -out[k].get("kernel_ms_under_FETCH_SIZE", 0))[:12]:
        print(name, {k: round(v, 2) for k, v in out[name].items() if not k.startswith("dispatches")})


if __name__ == "__main__":
    main()
