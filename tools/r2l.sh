cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['ms'],4) for k,v in d['stages'].items()})"
bash tools/pmc_traffic.sh; cat gpurun_out/traffic/pmc_fetch_write_8views.csv
bash tools/prof_stats.sh r2l | head -30
