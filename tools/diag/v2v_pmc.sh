R=$GRAFT_REPO_ROOT
bash $R/tools/pmc.sh $R/tools/step_once.py $R/gpurun_out/pmc_dbg.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null 2>&1
echo "$(grep -E 'v2v_(scan|leaves|tree)_kernel' $R/gpurun_out/pmc_dbg.txt | cut -c60-400)"
bash $R/tools/pmc.sh $R/tools/step_once.py $R/gpurun_out/pmc_dbg2.txt SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC > /dev/null 2>&1
echo "$(grep -E 'v2v_(scan|leaves|tree)_kernel' $R/gpurun_out/pmc_dbg2.txt | cut -c60-400)"
