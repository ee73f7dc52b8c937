mkdir -p gpurun_out/r06j
for i in 1 2 3; do for v in True False; do
python -c "
import sys, runpy
import glenet_amd.dense_path as dp
dp.OWN_WIDE_LINEAR = $v
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-config1', '--no-stages', '--no-strict', '--no-cpu-baseline', '--no-extra']
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/r06j/wide_${v}_$i.json 2>/dev/null
done; done
