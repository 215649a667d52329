#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_check; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_write_through_gpu.py tests/test_policy_step_gpu.py tests/test_fused_rollout_gpu.py tests/test_counter_rng_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python3 bench.py --no-configs --no-boundary --no-counter-rng --no-cpu-baseline --no-closed-loop 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default  per_step %.3f G fused %.3f G' % (d['value']/1e9, d['fused']['value']/1e9))"
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-configs --no-boundary --no-counter-rng --no-cpu-baseline --no-closed-loop 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver   per_step %.3f G fused %.3f G' % (d['value']/1e9, d['fused']['value']/1e9))"
done
