# kernel trace of exterior_flags (ray path) at batch 64
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ef.py <<'PY'
import os, sys; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
for _ in range(10):
    model.exterior_flags(verts, apply_segments=False)
torch.cuda.synchronize()
print(model.ray_work(verts) if hasattr(model, 'ray_work') else '')
PY
rm -rf /tmp/kt_ef
rocprofv3 --kernel-trace --stats -d /tmp/kt_ef -o kt -- python /tmp/ef.py > /tmp/ef.log 2>&1
grep -E "elements|Error|error" /tmp/ef.log | tail -3
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/kt_ef -name "*results.db" | head -1) 12 | cut -c1-200
