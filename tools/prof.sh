# usage: bash tools/prof.sh <tag> <what> [extra env]   -> gpurun_out/<tag>_<what>_kernels.txt
cd /tmp && export TMPDIR=/tmp
tag=$1; what=$2
rm -rf /tmp/kt_$what
rocprofv3 --kernel-trace --stats -d /tmp/kt_$what -o kt -- python $GRAFT_REPO_ROOT/tools/profile_step.py $what > /tmp/kt_$what.log 2>&1
tail -2 /tmp/kt_$what.log
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/kt_$what -name "*results.db" | head -1) 45 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_${what}_kernels.txt
