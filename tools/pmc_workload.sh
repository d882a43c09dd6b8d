# Run on the GPU box (gpurun): FETCH_SIZE / WRITE_SIZE passes of bench.py for one workload; per-kernel means.
# Usage: bash tools/pmc_workload.sh WORKLOAD TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-c2store}
TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcw_${TAG}_$c -- python3 $R/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcw_${TAG}_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmcw_${TAG}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "zn::" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1); w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    print("%-56s n=%-3d FETCH %11.0f KB  WRITE %11.0f KB  bytes %14.0f" % (k[:56], len(d["FETCH_SIZE"]), f, w, (2 * f + w) * 1024))
PY
