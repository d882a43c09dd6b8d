# Run on the GPU box: SQ counters of the C2 write step's encoder kernel — fused with the hash (default), and encoder only
# (ZNIPPY_NOHASH=1: the hash switched off).  Per-row figures = per-launch / 100,000.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mode in fused nohash; do
  if [ $mode = nohash ]; then export ZNIPPY_NOHASH=1; else unset ZNIPPY_NOHASH; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_encf_$mode -- python3 $R/tools/kt_write.py > $R/gpurun_out/pmc_encf_$mode.log 2>&1
  python3 - <<PY
import csv, glob, collections
print("== $mode")
for f in glob.glob("$R/gpurun_out/pmc_encf_$mode/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_zstd_encode" in r["Kernel_Name"] and "11" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print("%-20s n=%d mean=%.4g per row=%.1f" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / 1e5))
PY
  grep -h "wall" $R/gpurun_out/pmc_encf_$mode.log | head -2
done
