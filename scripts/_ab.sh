CFM_ATTN_Q128=1 timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -m gpu -x -q -k "attention or mhsa or config2_full or chained" 2>&1 | tail -1
for i in 1 2 3; do
CFM_ATTN_Q128=1 python bench.py --no-fp16 --train-steps 0 --no-cpu-baseline --no-live-traffic --no-parity --all-kernels 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q128', r['ms_per_step'])
"
python bench.py --no-fp16 --train-steps 0 --no-cpu-baseline --no-live-traffic --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q64 ', r['ms_per_step'])
"
done
