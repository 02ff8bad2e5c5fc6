# usage: bash tools/pmc.sh <script.py> <out.txt> COUNTER...
cd /tmp && export TMPDIR=/tmp
script=$1; out=$2; shift 2
rm -rf /tmp/pmc_run
rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_run -o p -- python $script > /tmp/pmc_run.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_pmc_summary.py $(find /tmp/pmc_run -name "*results.db" | head -1) > $out 2>&1 || tail -5 /tmp/pmc_run.log
