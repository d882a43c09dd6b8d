# GPU box: does the per-kernel event bracketing cost the step anything?  (same box, back to back)
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in 2 1 0 2 1 0; do
  ZNIPPY_KTIME=$k timeout -k 10 200 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ktime=$k ms_per_step', d['ms_per_step'], 'own', d.get('read_own_archive',{}).get('ms_per_step'), 'compress', d.get('compress_ms_per_step'), 'kernel_ms', (d.get('roofline') or {}).get('kernel_ms'))" || exit 1
done
