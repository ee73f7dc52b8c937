#!/bin/bash
# average duration of the kernels matching PATTERN in a short run of the training step under rocprofv3:
#   tools/kernel_avg.sh PATTERN [ENV=VALUE ...]
R=$GRAFT_REPO_ROOT; PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export $kv; done
rm -rf /tmp/p_ka
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ka -o ka -- python3 $R/bench.py --steps 20 --warmup 5 --no-config1 --no-stages --no-cpu-baseline --no-extra > /tmp/ka.log 2>&1
echo "== $@"
grep -E "$PAT" /tmp/p_ka/ka_kernel_stats.csv | awk -F'",' '{n=split($2,a,","); printf "%-70s calls %s avg_ns %s\n", substr($1,2,70), a[1], a[3]}'
