R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests/test_gpu_contact.py tests/test_gpu_properties.py -q -x 2>&1 | tail -3
TUCH_OVERLAP=0 bash $R/tools/prof.sh r03_dbg step > /dev/null 2>&1
grep -E 'v2v_|ray_near_kernel|ray_tiles' $R/gpurun_out/r03_dbg_step_kernels.txt | awk '{print $1,$2,$3, $(NF-8), $(NF-7)}' | cut -c1-120
bash tools/quick_bench.sh qb 2>&1 | tail -4
