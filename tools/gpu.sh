#!/bin/bash
# Build the library for gfx950, refuse to spend GPU time on a tree that does not build, then run the given command on the MI355X box.
#   tools/gpu.sh [--timeout S] '<command>'
set -e
cd "$(dirname "$0")/.."
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
python -m layout_dm_amd.build > /tmp/ldm_build.log 2>&1 || { tail -20 /tmp/ldm_build.log; echo "BUILD FAILED"; exit 1; }
python -c "from layout_dm_amd import binding; binding.load_library()" || { echo "LOAD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$1"
