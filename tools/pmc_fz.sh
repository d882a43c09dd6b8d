# GPU box: SQ counters of the foreign-frame kernels on one frame (tools/diag_fz.py).  Usage: bash tools/pmc_fz.sh TAG [kind] [MiB]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; KIND=${2:-text}; MIB=${3:-4}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_fz_${TAG}_1 -- python3 $R/tools/diag_fz.py $KIND $MIB 1 > $R/gpurun_out/pmc_fz_${TAG}_1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_fz_${TAG}_2 -- python3 $R/tools/diag_fz.py $KIND $MIB 1 > $R/gpurun_out/pmc_fz_${TAG}_2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in (1, 2):
    for f in glob.glob("$R/gpurun_out/pmc_fz_${TAG}_%d/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_fz_" in k:
                acc[k.split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in sorted(acc.items()):
            print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
