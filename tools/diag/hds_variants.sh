# usage (on the GPU box): bash tools/diag/hds_variants.sh "<flags>" ...  -> HD search kernel time per build variant
R=$GRAFT_REPO_ROOT
for f in "$@"; do
  touch $R/tuch_amd/csrc/hd_search.hip
  TUCH_HDS_FLAGS="$f" python -m tuch_amd._build   # (needs EXTRA_FLAGS in _build.py to append os.environ TUCH_HDS_FLAGS) > /dev/null 2>&1 || echo build failed
  bash $R/tools/prof.sh v hd > /dev/null 2>&1
  echo "[$f] $(grep hd_search $R/gpurun_out/v_hd_kernels.txt | cut -c60-130)"
done
