# usage: bash tools/diag/search_pmc.sh -- counter passes over the search ALONE (scalar cache, instruction cache, LDS, waits)
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc.sh $R/tools/diag/search_once.py $R/gpurun_out/search_pmc1.txt SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES > /dev/null 2>&1
bash $R/tools/pmc.sh $R/tools/diag/search_once.py $R/gpurun_out/search_pmc2.txt SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH > /dev/null 2>&1
bash $R/tools/pmc.sh $R/tools/diag/search_once.py $R/gpurun_out/search_pmc3.txt SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU > /dev/null 2>&1
bash $R/tools/pmc.sh $R/tools/diag/search_once.py $R/gpurun_out/search_pmc4.txt SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE > /dev/null 2>&1
for i in 1 2 3 4; do grep -E "v2v_scan" $R/gpurun_out/search_pmc$i.txt | cut -c60-600; done
