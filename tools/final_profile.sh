# Run on the GPU box (gpurun): kernel-trace summary of bench.py + PMC passes of the C2 read step.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1f}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG} -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_fetch_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG} -- python3 $R/tools/kt.py > $R/gpurun_out/pmc_write_${TAG}.log 2>&1
find $R/gpurun_out/prof_${TAG} $R/gpurun_out/pmc_fetch_${TAG} $R/gpurun_out/pmc_write_${TAG} -name "*.csv" | head -20
tail -1 $R/gpurun_out/prof_${TAG}_bench.log
