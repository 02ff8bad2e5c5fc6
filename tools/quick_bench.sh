# usage: bash tools/quick_bench.sh <name> [extra bench args]  -> gpurun_out/<name>.json + a one-line summary
N=$1; shift
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rccl-smoke "$@" > $R/gpurun_out/$N.json 2> $R/gpurun_out/$N.err || tail -c 2000 $R/gpurun_out/$N.err
python - <<PY
import json
l=json.loads(open("$R/gpurun_out/$N.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "median", l["repeat_ms_per_step"]["median"], "kernels", l.get("kernels_per_step"), "selfcheck ok", l.get("selfcheck", {}).get("ok"))
if "shard_sweep" in l: print("shards", {k:v["graph_ms"] for k,v in l["shard_sweep"].items()})
if "contact_loss_eval" in l: print("cle", {k:v for k,v in l["contact_loss_eval"].items() if "step" in k})
PY
