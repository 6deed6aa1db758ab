#!/bin/bash
# dev-container helper: rebuild everything in-tree (the .so files travel with the snapshot), then run a command on an MI355X box.
# Usage: tools/gpu.sh TIMEOUT_SECONDS 'command ...'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -c "import __graft_entry__ as g; g.build()" | tail -1
# the commit the tree was built from travels with the snapshot (.git does not): bench.py and the evidence summaries quote it
echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- flash-attention-turing_amd include bench.py || echo +dirty)" > flash-attention-turing_amd/BUILD_COMMIT
T=$1; shift
exec timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
