# GPU box: kernel-trace timelines of one step of each leg of a workload.  Usage: bash tools/timeline.sh WORKLOAD TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-c2}; TAG=${2:-t}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_${TAG}_${WL} -- python3 $R/bench.py --workload $WL --steps 12 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tl_${TAG}_${WL}.log 2>&1
F=$(find $R/gpurun_out/tl_${TAG}_${WL} -name "*kernel_trace.csv" | head -1)
for anchor in "$3" "$4"; do
  [ -n "$anchor" ] && { echo "== $WL: steps starting at $anchor"; python3 $R/tools/timeline.py $F "$anchor" 20 2; }
done
