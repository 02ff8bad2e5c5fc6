# usage: bash tools/timelines.sh <tag>  -> gpurun_out/<tag>_graph_timeline.txt and _b8.txt (one replayed step each)
T=${1:-r03_x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt; rocprofv3 --kernel-trace -d /tmp/gt -o kt -- python $R/tools/graph_timeline.py run > /tmp/gt.log 2>&1; python $R/tools/graph_timeline.py show $(find /tmp/gt -name "*results.db" | head -1) > $O/${T}_graph_timeline.txt
rm -rf /tmp/gt8; rocprofv3 --kernel-trace -d /tmp/gt8 -o kt -- python $R/tools/graph_timeline.py run 8 > /tmp/gt8.log 2>&1; python $R/tools/graph_timeline.py show $(find /tmp/gt8 -name "*results.db" | head -1) > $O/${T}_graph_timeline_b8.txt
