set -e
for w in ${WORKLOADS:-c2 c3 c3slot c4store c4codec c5 c5text}; do
  timeout -k 10 400 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1 || { echo "FAILED $w"; tail -5 gpurun_out/bench_$w.log; exit 1; }
  tail -1 gpurun_out/bench_$w.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', 'read MB/s', d['value'], 'ms', d['ms_per_step'], '| write MB/s', d['compress_MBps'], 'ms', d['compress_ms_per_step'], '| frac', d['roofline']['frac'], d['roofline']['kernel_ms'], d['compress_kernel_ms'], 'blob', d['config']['blob_bytes_per_gpu'])
"
done
