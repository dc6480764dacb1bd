one() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms_rank0'])"; }
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t6.log 2>&1; tail -3 gpurun_out/r2_t6.log
echo default; one || exit 1
for r in 8 12 16 20 32; do echo "refill_min $r"; RT_REFILL_MIN=$r one || exit 1; done
for f in 1 3 4; do echo "descend_frac $f"; RT_DESCEND_FRAC=$f one || exit 1; done
echo "chains 1"; RT_WF_CHAINS=1 one
