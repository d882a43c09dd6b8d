# A/B of prebuilt library variants on the default bench's text leg (see tools/variants_run.sh).  Usage: bash tools/variants_text.sh v1 v2 ...
set -e
cp znippy_amd/libznippy_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp build_variants/lib_$v.so znippy_amd/libznippy_hip.so
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/vart_${v}.log 2>&1 || { echo "FAILED $v"; tail -5 gpurun_out/vart_${v}.log; cp /tmp/lib_keep.so znippy_amd/libznippy_hip.so; exit 1; }
  tail -1 gpurun_out/vart_${v}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
t=d['read_text_archive']
print('$v', 'c2 read', d['ms_per_step'], 'write', d['compress_ms_per_step'], 'text ms', t['ms_per_step'], {k:v for k,v in t['kernel_ms'].items() if v>0.05})
"
done
cp /tmp/lib_keep.so znippy_amd/libznippy_hip.so
