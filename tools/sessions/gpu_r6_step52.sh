# A/B of the one-copy 3-tap weight-gradient kernel on the whole bf16 training step (same box, alternating)
for i in 1 2; do
for u in 0 1; do
  echo "== SET_AMD_WGRAD3_UNITS=$u"
  SET_AMD_WGRAD3_UNITS=$u python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','host_enqueue_ms_per_step','value','loss')})"
done; done
