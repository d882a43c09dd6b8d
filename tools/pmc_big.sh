# Run on the GPU box: HBM bytes per launch of the fused block kernel (64 x 8 MiB text rows = 512 MiB out).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_big_$c -- python3 $R/tools/kt_big.py 64 ${1:-text} > $R/gpurun_out/pmc_big_$c.log 2>&1
tail -1 $R/gpurun_out/pmc_big_$c.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pmc_big_$c/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fused_blocks" in r["Kernel_Name"] or "k_hash_tiles" in r["Kernel_Name"] or "k_zstd_decode" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, "n=%d" % len(v), "mean KB=%.0f" % (sum(v) / len(v)))
PY
done
