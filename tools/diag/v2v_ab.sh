R=$GRAFT_REPO_ROOT
for cfg in "2 65536" "2 30000" "2 14000" "2 7000" "1 65536" "0 65536"; do
  set -- $cfg
  TUCH_V2V_FLAT=$1 TUCH_V2V_WAVES=$2 TUCH_OVERLAP=0 bash $R/tools/prof.sh r03_dbg step > /dev/null 2>&1
  echo "flat=$1 waves=$2: $(grep -E 'v2v_(scan|leaves|tree)_kernel' $R/gpurun_out/r03_dbg_step_kernels.txt | awk '{print $(NF-9), $(NF-8)}')"
done
