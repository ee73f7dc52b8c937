python -m pytest tests/test_sparse_gpu.py tests/test_config4_train_gpu.py -x -q -m gpu 2>&1 | tail -2
GLX_SCONV_WGRAD_F16=2 python -m pytest tests/test_sparse_gpu.py tests/test_config4_train_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 2 0 1; do GLX_SCONV_WGRAD_F16=$v python3 bench.py --config4-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); t=d['config4']['train']
print('wgrad_f16=$v fwd_bwd_ms', t['fwd_bwd_ms'], {k:v['us_per_call'] for k,v in t['wgrad_kernels']['per_kernel'].items()})"; done
