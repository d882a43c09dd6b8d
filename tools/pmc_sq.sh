# Run on the GPU box (gpurun): SQ counter passes of the C2 read step (tools/kt.py), fused kernel only.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq1_${TAG} -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_sq1_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_sq2_${TAG} -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_sq2_${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq1_${TAG}", "pmc_sq2_${TAG}"):
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_fused_small" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)))
PY
