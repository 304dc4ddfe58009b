# usage (GPU box): bash tools/icache_probe.sh <config> <reads>  -- instruction-cache requests / misses per kernel of the timed steps
set -e
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/run -o p -- python3 $R/bench.py --pmc-run cfg$1 --config $1 --reads $2 --steps 4 --warmup 2 > $O/run.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/run/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("mia::","")
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    seen.add((k,r["Dispatch_Id"]))
for k,d in seen: nd[k]+=1
out=open("$O/icache.txt","w")
for k,v in sorted(agg.items(), key=lambda x:-x[1].get("SQ_WAVE_CYCLES",0)):
    n=nd[k]
    req,miss=v.get("SQC_ICACHE_REQ",0)/n,v.get("SQC_ICACHE_MISSES",0)/n
    print("%-34s x%-3d icache req %10.0f miss %9.0f (%.3f)  ifetch %10.0f  wave_cycles %11.0f wait_any %11.0f busy %9.0f" % (k[:34],n,req,miss,miss/max(req,1),v.get("SQ_IFETCH",0)/n,v.get("SQ_WAVE_CYCLES",0)/n,v.get("SQ_WAIT_INST_ANY",0)/n,v.get("SQ_BUSY_CYCLES",0)/n),file=out)
out.close()
PY
cat $O/icache.txt
