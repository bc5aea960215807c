#!/bin/bash
# quick A/B runs of bench.py: tools/exp.sh "<env> :: <bench args>" ...
for spec in "$@"; do
  envs="${spec%%::*}"; args="${spec##*::}"
  echo "== $spec"
  env $envs python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 $args 2>&1 | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  Mrays/s %.0f  ms %.2f  kernel_ms %.2f  nodes/ray %.2f tris/ray %.2f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['nodes_per_ray'], r['tris_per_ray']))"
done
