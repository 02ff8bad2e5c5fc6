# usage: bash tools/diag/lds_ab.sh -- wave slots per SIMD the search may take beside the inside test (option v2v_lds: 0 = no
# cap, -7 / -6 / -5 by register count): quick bench lines, twice
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 -7 -6; do
  echo "== v2v_lds=$v"
  TUCH_V2V_LDS=$v bash $R/tools/quick_bench.sh lds_$v 2>&1 | head -2
done
done
