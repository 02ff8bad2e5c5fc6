# usage: bash tools/diag/search_ab.sh  -- the default library against tuch_amd/libtuch_amd_prev.so (built by hand from an
# earlier v2v.hip) on the same box: the search alone (tools/diag/fresh_search.py) and the quick bench line of each, twice
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "" $R/tuch_amd/libtuch_amd_prev.so; do
    echo "== ${lib:-default}"
    TUCH_AMD_LIB=$lib python $R/tools/diag/fresh_search.py 64 2>&1 | grep -E "search (same|rotate)"
    TUCH_AMD_LIB=$lib bash $R/tools/quick_bench.sh ab_$rep --no-extras 2>&1 | head -1
  done
done
