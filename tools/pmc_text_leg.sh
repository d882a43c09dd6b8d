# Run on the GPU box (gpurun): SQ instruction counters of the batch path's kernels on the default bench's text leg.
# Usage: bash tools/pmc_text_leg.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_txt1_${TAG} -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_txt1_${TAG}.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_txt2_${TAG} -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_txt2_${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_txt1_${TAG}", "pmc_txt2_${TAG}"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_bx_" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in sorted(acc.items()):
        print(d, k[:48], " ".join("%s=%.4g" % (c.replace("SQ_", ""), max(v)) for c, v in sorted(dd.items())))
PY
