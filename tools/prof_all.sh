# usage: bash tools/prof_all.sh <tag>   (e.g. r03_a) -> gpurun_out/<tag>_* : kernel traces, PMC passes, graph timelines
T=${1:-r03_a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/prof.sh ${T} step; cp $O/${T}_step_kernels.txt $O/${T}_bench_step_kernel_trace.txt
TUCH_OVERLAP=0 bash $R/tools/prof.sh ${T}_no step; cp $O/${T}_no_step_kernels.txt $O/${T}_bench_step_kernel_trace_no_overlap.txt
bash $R/tools/prof.sh ${T} hd; cp $O/${T}_hd_kernels.txt $O/${T}_hd_train_step_kernel_trace.txt
bash $R/tools/pmc.sh $R/tools/step_once.py $O/${T}_pmc_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/${T}_pmc_sq2.txt SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS
bash $R/tools/pmc.sh $R/tools/step_once.py $O/${T}_pmc_fetch.txt FETCH_SIZE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/${T}_pmc_write.txt WRITE_SIZE
bash $R/tools/pmc.sh $R/tools/hd_once.py $O/${T}_pmc_hd_sq.txt SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
bash $R/tools/pmc.sh $R/tools/step_once.py $O/${T}_pmc_mfma.txt SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
ls -la $O/${T}_*
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/gt; rocprofv3 --kernel-trace -d /tmp/gt -o kt -- python $R/tools/graph_timeline.py run > /tmp/gt.log 2>&1; python $R/tools/graph_timeline.py show $(find /tmp/gt -name "*results.db" | head -1) > $O/${T}_graph_timeline.txt
rm -rf /tmp/gt8; rocprofv3 --kernel-trace -d /tmp/gt8 -o kt -- python $R/tools/graph_timeline.py run 8 > /tmp/gt8.log 2>&1; python $R/tools/graph_timeline.py show $(find /tmp/gt8 -name "*results.db" | head -1) > $O/${T}_graph_timeline_b8.txt
