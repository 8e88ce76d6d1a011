#!/bin/bash
set -u
OUT=gpurun_out/r03final
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -14 "$OUT/pytest_gpu.log"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
bash tools/collect_profiles.sh "$OUT/profiles" > "$OUT/collect.log" 2>&1; tail -5 "$OUT/collect.log"
