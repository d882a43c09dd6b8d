# Run on the GPU box (gpurun): effective shader clock of the read kernels and of the BLAKE3-only microbenchmark
# (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; MI355X_MICROARCH.md "DVFS give-back").  Usage: bash tools/pmc_clock.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
for mode in roles plain; do
  if [ $mode = plain ]; then export ZNIPPY_NO_ROLES=1; else unset ZNIPPY_NO_ROLES; fi
  UBENCH=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_${TAG}_${mode} -- python3 $R/tools/diag_roles.py > $R/gpurun_out/clk_${TAG}_${mode}.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for mode in ("roles", "plain"):
    for f in glob.glob("$R/gpurun_out/clk_${TAG}_%s/**/*counter_collection.csv" % mode, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_fused" in k or "ubench" in k or "hash_tiles" in k or "zstd_encode" in k:
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                acc[k.split("(")[0][-26:]].append((float(r["Counter_Value"]), dur))
        for k, v in acc.items():
            g = sum(x for x, _ in v) / len(v); d = sum(y for _, y in v) / len(v)
            print(mode, k, "n=%d" % len(v), "GUI_ACTIVE=%.4g" % g, "dur=%.1f us" % (d / 1e3), "clock=%.3f GHz" % (g / 8 / d))
PY
