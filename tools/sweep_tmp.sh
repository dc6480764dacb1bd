timeout -k 10 600 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/r2_t9.log 2>&1; tail -4 gpurun_out/r2_t9.log
for c in subtraction_outside subtraction_outside_ss; do
  timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s', d['stage_ms_rank0'])"
done
