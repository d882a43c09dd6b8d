# Run on the GPU box (gpurun): SQ counter passes of the C2 read step (tools/diag_roles.py, libzstd-19 frames),
# role-split kernel and (ZNIPPY_NO_ROLES=1) the one-kernel path.  Usage: bash tools/pmc_roles.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
for mode in roles plain; do
  if [ $mode = plain ]; then export ZNIPPY_NO_ROLES=1; else unset ZNIPPY_NO_ROLES; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_1 -- python3 $R/tools/diag_roles.py > $R/gpurun_out/pmc_${TAG}_${mode}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_2 -- python3 $R/tools/diag_roles.py > $R/gpurun_out/pmc_${TAG}_${mode}_2.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for mode in ("roles", "plain"):
    for p in (1, 2):
        for f in glob.glob("$R/gpurun_out/pmc_${TAG}_%s_%d/**/*counter_collection.csv" % (mode, p), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k_fused" in k:
                    acc[k.split("(")[0][-22:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, d in acc.items():
                print(mode, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
