#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for c in 0 1; do
RT_CUSHARE=$c python3 bench.py --workload C2 --no-pmc --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cus=$c', d['value'], {k: round(r[k],2) for k in ('lanes_per_node_step','leaf_phases_per_node_step','lanes_at_leaf_per_node_step','lanes_without_ray_per_node_step','nodes_per_ray')})"
done
