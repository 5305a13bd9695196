#!/bin/bash
# N rank processes on ONE GPU through the peer-copy transport at the headline size: the GPU does the same work as with
# one rank, so what ms/frame gains over the single-rank line is the per-frame cost of the N-rank pipeline itself
# (hand-shakes, copies, assembly, host work) -- not a scaling measurement.
for n in 1 2 4; do
  if [ $n = 1 ]; then extra="--force-distributed --transport ipc"; else extra="--gpus $n --transport ipc"; fi
  python bench.py $extra --no-cpu-baseline --steps 48 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ranks on one GPU: $n  ms/frame %.3f  Mrays/s %.0f  private strips %s  trials %s' % (d['ms_per_step'], d['value'], d['config']['strip_calibration']['private_strips_of_16'], d['config']['strip_calibration'].get('ms_per_frame_by_private_strips')))"
done
python bench.py --no-cpu-baseline --steps 48 --warmup 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('single rank, direct render: ms/frame %.3f  (two frames in flight %.3f)' % (d['ms_per_step'], d['two_frames_in_flight']['ms_per_step']))"
