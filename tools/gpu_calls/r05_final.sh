#!/bin/bash
# Dev tool (GPU box, via gpurun): round 5's evidence run = smoke() + the r04 evidence script (suite, bench line at the driver's
# settings, rocprofv3 stats of the three numerics modes and of cond=relation, SQ counter passes) -> gpurun_out/$1/.
set -u
TAG=${1:-r05_final}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -2 gpurun_out/$TAG/smoke.log
bash tools/gpu_calls/r04_final.sh $TAG
