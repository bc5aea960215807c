#!/bin/bash
# quick A/B on the GPU box: tools/ab.sh "<env assignments>" <workloads...>  -> one line per run
envs="$1"; shift
for wl in "$@"; do
  r=$(env $envs python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload $wl 2>&1 | tail -1)
  echo "$wl [$envs] $(echo "$r" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.0f Mrays/s  %.2f ms  kernel %.2f ms"%(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]))' 2>/dev/null || echo "$r" | tail -c 300)"
done
