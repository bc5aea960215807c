#!/bin/bash
cd "$GRAFT_REPO_ROOT"
tools/run_guarded.sh gpurun_out/t_all.log 1100 python3 -m pytest tests -q -m gpu || exit 1
tools/run_guarded.sh gpurun_out/smoke.log 200 python3 -c "import __graft_entry__ as g; g.smoke()" || exit 1
