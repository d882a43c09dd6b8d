# Run on the GPU box: L2 (TCC) request counters of the fused kernel on the C2 read step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_WRITE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_tcc -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_tcc.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pmc_tcc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fused_small" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(k, "%.4g" % (sum(v) / len(v)))
PY
tail -3 $R/gpurun_out/pmc_tcc.log
