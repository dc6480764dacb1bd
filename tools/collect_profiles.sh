#!/bin/bash
# Collects, on the GPU box, the rocprofv3 runs behind profiles/: for every benchmark configuration a kernel
# trace (3 frames), the SQ counters, and FETCH_SIZE / WRITE_SIZE in passes of their own (1 frame each).
# Usage: tools/collect_profiles.sh <out-dir under gpurun_out> [config ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; shift
CONFIGS=${*:-"glass_monkey glass_monkey_1chain light_room many_balls_forest subtraction_outside subtraction_outside_ss"}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in $CONFIGS; do
  unset RT_PROF_AA_THRESHOLD RT_WF_CHAINS
  case $c in
    glass_monkey) scene=glass_monkey; w=1920; h=1080;;
    glass_monkey_1chain) scene=glass_monkey; w=1920; h=1080; export RT_WF_CHAINS=1;;
    light_room) scene=light_room; w=1920; h=1080;;
    many_balls_forest) scene=many_balls_forest; w=3840; h=2160;;
    subtraction_outside) scene=subtraction_outside1; w=3840; h=2160;;
    subtraction_outside_ss) scene=subtraction_outside1; w=3840; h=2160; export RT_PROF_AA_THRESHOLD=0;;
    *) echo "unknown config $c"; exit 1;;
  esac
  echo "== $c"
  timeout -s KILL 180 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$c/kt" -o kt -- python3 "$ROOT/tools/prof_frame.py" $scene $w $h 3 1 nocount > "$OUT/$c.kt.log" 2>&1 || { echo "kernel trace failed for $c"; exit 1; }
  grep "^frame" "$OUT/$c.kt.log" | tail -1
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$c/fetch" -o pmc -- python3 "$ROOT/tools/prof_frame.py" $scene $w $h 1 1 nocount > "$OUT/$c.fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed for $c"; exit 1; }
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$c/write" -o pmc -- python3 "$ROOT/tools/prof_frame.py" $scene $w $h 1 1 nocount > "$OUT/$c.write.log" 2>&1 || { echo "WRITE_SIZE pass failed for $c"; exit 1; }
  timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/$c/sq" -o pmc -- python3 "$ROOT/tools/prof_frame.py" $scene $w $h 1 1 nocount > "$OUT/$c.sq.log" 2>&1 || { echo "SQ pass failed for $c"; exit 1; }
done
# calibration of the two traffic counters on known byte counts
if [ -x "$ROOT/exp/hbm_calib" ]; then
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib/fetch" -o pmc -- "$ROOT/exp/hbm_calib" > "$OUT/calib.fetch.log" 2>&1 || echo "calibration (fetch) failed"
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib/write" -o pmc -- "$ROOT/exp/hbm_calib" > "$OUT/calib.write.log" 2>&1 || echo "calibration (write) failed"
fi
echo done
