R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/prof.sh r02_m step; cp $O/r02_m_step_kernels.txt $O/r02_m_bench_step_kernel_trace.txt
TUCH_OVERLAP=0 bash $R/tools/prof.sh r02_m_no step; cp $O/r02_m_no_step_kernels.txt $O/r02_m_bench_step_kernel_trace_no_overlap.txt
bash $R/tools/prof.sh r02_m hd; cp $O/r02_m_hd_kernels.txt $O/r02_m_hd_train_step_kernel_trace.txt
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_m_pmc_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_m_pmc_sq2.txt SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_m_pmc_fetch.txt FETCH_SIZE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_m_pmc_write.txt WRITE_SIZE
bash $R/tools/pmc.sh $R/tools/hd_once.py $O/r02_m_pmc_hd_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_m_pmc_mfma.txt SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
ls -la $O/r02_m_*
