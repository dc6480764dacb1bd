#!/bin/bash
# Kernel trace of a few frames under rocprofv3 and its per-kernel summary.
# Usage (on the GPU box): tools/kt.sh <out-dir under gpurun_out> [scene w h] ; RT_* environment passes through
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; shift
scene=${1:-glass_monkey}; w=${2:-1920}; h=${3:-1080}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/tools/prof_frame.py" $scene $w $h 3 1 nocount > "$OUT/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/kt.log"; exit 1; }
grep "^frame" "$OUT/kt.log" | tail -1
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %6s total %9.3f ms avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / 3, float(r["AverageNs"]) / 1e3))
PY
