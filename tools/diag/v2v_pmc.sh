R=$GRAFT_REPO_ROOT
for cfg in "2 65536" "2 14000" "0 65536"; do
  set -- $cfg
  TUCH_V2V_FLAT=$1 TUCH_V2V_WAVES=$2 bash $R/tools/pmc.sh $R/tools/step_once.py $R/gpurun_out/pmc_dbg.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1
  echo "flat=$1 waves=$2: $(grep -E 'v2v_(scan|leaves|tree)_kernel' $R/gpurun_out/pmc_dbg.txt | cut -c60-400)"
done
