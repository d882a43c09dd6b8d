# Run on the GPU box: SQ counters of the small-block encoder on the C2 write step (hash switched off).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export ZNIPPY_NOHASH=1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_enc -- python3 $R/tools/kt_write.py > $R/gpurun_out/pmc_enc.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pmc_enc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_zstd_encode" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)), "per row=%.1f" % (sum(v) / len(v) / 1e5))
PY
