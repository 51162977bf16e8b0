# same-box A/B: quad-interleaved bf16 intermediates against the [C][T] layout (library built from the commit before), bf16 training step
mkdir -p gpurun_out
for i in 1 2 3; do
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$PWD/speech-editing-toolkit_amd/libset_amd_before_q4.so; else unset SET_AMD_LIB; fi
  python bench.py --mode train --dtype bf16 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k:round(d.get(k),3) for k in ('ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['kernel'], round(d['roofline']['launch_ms']*1e3,1), 'us')"
done; done
unset SET_AMD_LIB
cd /tmp && export TMPDIR=/tmp
for v in before after; do
  if [ $v = before ]; then export SET_AMD_LIB=$GRAFT_REPO_ROOT/speech-editing-toolkit_amd/libset_amd_before_q4.so; else unset SET_AMD_LIB; fi
  rm -rf /tmp/prof_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --dtype bf16 --steps 10 --warmup 3 > /tmp/prof_$v.log 2>&1
  S=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -6 "$S" | cut -c1-150
done
