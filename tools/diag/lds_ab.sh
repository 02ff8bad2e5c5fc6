# usage: bash tools/diag/lds_ab.sh -- how many wave slots per SIMD the search takes beside the inside test (option v2v_lds < 0:
# by register count): quick bench line per setting, twice
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in -7 -6 -5 -4; do
  echo "== v2v_lds=$v"
  TUCH_V2V_LDS=$v bash $R/tools/quick_bench.sh lds_$v --no-extras 2>&1 | head -1
done
done
