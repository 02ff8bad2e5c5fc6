R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/prof.sh r02_n step; cp $O/r02_n_step_kernels.txt $O/r02_n_bench_step_kernel_trace.txt
TUCH_OVERLAP=0 bash $R/tools/prof.sh r02_n_no step; cp $O/r02_n_no_step_kernels.txt $O/r02_n_bench_step_kernel_trace_no_overlap.txt
bash $R/tools/prof.sh r02_n hd; cp $O/r02_n_hd_kernels.txt $O/r02_n_hd_train_step_kernel_trace.txt
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_n_pmc_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_n_pmc_sq2.txt SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_n_pmc_fetch.txt FETCH_SIZE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_n_pmc_write.txt WRITE_SIZE
bash $R/tools/pmc.sh $R/tools/hd_once.py $O/r02_n_pmc_hd_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/r02_n_pmc_mfma.txt SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
ls -la $O/r02_n_*
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/gt; rocprofv3 --kernel-trace -d /tmp/gt -o kt -- python $R/tools/graph_timeline.py run > /tmp/gt.log 2>&1; python $R/tools/graph_timeline.py show $(find /tmp/gt -name "*results.db" | head -1) > $O/r02_n_graph_timeline.txt
