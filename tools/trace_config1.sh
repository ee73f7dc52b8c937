#!/bin/bash
# Launch sequence of configs[1]'s recorded frame under rocprofv3 --kernel-trace, next to its untraced time
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 2 --no-stages --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('untraced config1', d['config1'])"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_c1 -o c1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-stages --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 $R/tools/trace_window.py /tmp/p_c1/c1_kernel_trace.csv --steps 10 --marker "k_dense_from_index<" --top 5 --seq $R/gpurun_out/c1_seq.txt | head -3
python3 $R/tools/seq_gaps.py $R/gpurun_out/c1_seq.txt | head -8
