# Same-box A/B of the C2 read step: role-split kernel vs one-kernel path, each twice, plus the BLAKE3-only
# microbenchmark for normalisation (boxes differ by >10 %).  Usage (gpurun): bash tools/ab_read.sh [extra env]
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for m in 0 1; do
    echo "NO_ROLES=$m $(ZNIPPY_NO_ROLES=$m UBENCH=1 ZNIPPY_DBG=${DBG:-32768} N=100000 python3 $R/tools/diag_roles.py 2>&1 | grep "decode_verify\|GHz\|pass ns" | tr '\n' ' ' | sed 's/zstd_decode_general.*shader/shader/')"
  done
done
