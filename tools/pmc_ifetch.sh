# Run on the GPU box (gpurun): instruction-fetch side of the read kernels vs the BLAKE3-only microbenchmark.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
for mode in roles plain; do
  if [ $mode = plain ]; then export ZNIPPY_NO_ROLES=1; else unset ZNIPPY_NO_ROLES; fi
  UBENCH=1 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/if_${TAG}_${mode} -- python3 $R/tools/diag_roles.py > $R/gpurun_out/if_${TAG}_${mode}.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for mode in ("roles", "plain"):
    for f in glob.glob("$R/gpurun_out/if_${TAG}_%s/**/*counter_collection.csv" % mode, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_fused_roles" in k or "k_fused_small" in k or "ubench" in k:
                acc[k.split("(")[0][-20:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(mode, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
