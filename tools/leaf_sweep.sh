for lf in 32 48 64 96; do
  for ch in 8 16; do
    echo "leaf_faces=$lf chunks=$ch"
    TUCH_TREE_LEAF_FACES=$lf TUCH_RAY_CHUNKS=$ch python bench.py --no-cpu-baseline --no-extras --steps 50 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('  step', l['ms_per_step'], l['repeat_ms_per_step']['median'], 'exterior', l['roofline']['launch_ms'], 'v2v', l['roofline_v2v']['launch_ms'])"
  done
done
