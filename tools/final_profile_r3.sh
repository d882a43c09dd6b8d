# Run on the GPU box (gpurun): round-3 evidence for profiles/.
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (per-kernel durations)
#   2. separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (HBM bytes per launch, every kernel)
#   3. the same two passes for --workload c4store (the run that used to die inside the profiler, profiles/README.md)
# Usage: bash tools/final_profile_r3.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
set -e
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 $R/bench.py --steps 20 --warmup 3 > $O/prof_${TAG}_bench.log 2>&1
grep "^{\"metric\"" $O/prof_${TAG}_bench.log | head -1 > $O/${TAG}_bench_c2.json
# the same command with the headline leg alone: the other read legs launch the same kernels, so only here does the
# stats file's per-kernel average describe the leg `roofline` is quoted on
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_headline -- python3 $R/bench.py --steps 20 --warmup 3 --headline-only --no-cpu-baseline > $O/prof_${TAG}_headline.log 2>&1
grep "^{\"metric\"" $O/prof_${TAG}_headline.log | head -1 > $O/${TAG}_bench_c2_headline.json
for wl in c2 c4store; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${TAG}_${wl}_$c -- python3 $R/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_${TAG}_${wl}_$c.log 2>&1
    echo "$wl $c exit $?"
  done
done
python3 - <<PY > $O/${TAG}_traffic_summary.txt
import csv, glob, collections
for wl in ("c2", "c4store"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$O/pmc_${TAG}_%s_%s/**/*counter_collection.csv" % (wl, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if k.startswith("void "): k = k[5:]
                if "zn::" in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", wl, "(KB per launch; HBM bytes = (2*FETCH + WRITE) * 1024, MI355X_MICROARCH.md gfx950 correction)")
    for k, d in sorted(acc.items()):
        f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        print("%-60s n=%-4d FETCH %12.0f  WRITE %12.0f  bytes %15.0f" % (k[:60], len(d["FETCH_SIZE"]), f, w, (2 * f + w) * 1024))
PY
cat $O/${TAG}_traffic_summary.txt
find $O/prof_${TAG} -name "*kernel_stats.csv" | head -3
cat $O/${TAG}_bench_c2.json
