# usage (GPU box): bash tools/pmc_probe.sh <config> <reads> <tag> COUNTER...   -- any counters of one pass, averaged per dispatch and kernel of the timed steps
set -e
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; c=$1; reads=$2; tag=$3; shift 3
O=$R/gpurun_out/pmcp_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/run -o p -- python3 $R/bench.py --pmc-run cfg$c --config $c --reads $reads --steps 4 --warmup 2 > $O/run.log 2>&1
python3 - "$O" "$@" <<'PY'
import csv, glob, collections, sys
O, names = sys.argv[1], sys.argv[2:]
f = glob.glob(O + "/run/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mia::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add((k, r["Dispatch_Id"]))
nd = collections.Counter(k for k, _ in seen)
with open(O + ".txt", "w") as out:
    print("%-34s %4s " % ("kernel", "n") + " ".join("%16s" % n[:16] for n in names), file=out)
    for k, v in sorted(agg.items(), key=lambda x: -x[1].get(names[0], 0) / nd[x[0]]):
        print("%-34s %4d " % (k[:34], nd[k]) + " ".join("%16.0f" % (v.get(n, 0) / nd[k]) for n in names), file=out)
PY
rm -rf $O
cat $O.txt
