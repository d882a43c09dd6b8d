# A/B of prebuilt library variants (build_variants/lib_*.so, built here by hand from -D flags): each takes the library's
# place in turn and runs the named workloads; one line per variant and workload.  Usage: bash tools/variants_run.sh "c4store c5" v0 v1 ...
set -e
W="$1"; shift
cp znippy_amd/libznippy_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp build_variants/lib_$v.so znippy_amd/libznippy_hip.so
  for w in $W; do
    timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/var_${v}_$w.log 2>&1 || { echo "FAILED $v $w"; tail -5 gpurun_out/var_${v}_$w.log; cp /tmp/lib_keep.so znippy_amd/libznippy_hip.so; exit 1; }
    tail -1 gpurun_out/var_${v}_$w.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '$w', 'read ms', d['ms_per_step'], 'write ms', d['compress_ms_per_step'], d['roofline']['kernel_ms'], d['compress_kernel_ms'])
"
  done
done
cp /tmp/lib_keep.so znippy_amd/libznippy_hip.so
