python -m pytest tests/test_gpu_walk_queue.py tests/test_gpu_nested_sampling.py tests/test_gpu_device_walk.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --steps 20 --warmup 5 --cpu-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step')}), 'walk', d['device_walk']['ms_per_mcmc_step'], 'queue', {k:d['device_walk_queue'][k] for k in ('map_ms','device_ms','evals_per_s','fraction_of_inner_loop_rate')})"
NMMA_WALK_NO_FUSE=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NO_FUSE queue', {k:d['device_walk_queue'][k] for k in ('map_ms','device_ms','evals_per_s')})"
