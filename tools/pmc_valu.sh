# Run on the GPU box: VALU/SALU instruction counts of the fused kernel under ZNIPPY_DBG ablations.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for d in ${@:-0 1 16 17}; do
export ZNIPPY_DBG=$d
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_valu_$d -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_valu_$d.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/pmc_valu_$d/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fused_small" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("DBG $d", {k: round(sum(v) / len(v) / 16667) for k, v in acc.items()}, "per tile")
PY
done
