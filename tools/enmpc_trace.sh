#!/bin/bash
# Development probe (GPU box): per-kernel times of the economic workloads (kernel trace, one stream) + a check against the C restatement.
#   bash tools/enmpc_trace.sh [tag]
export TMPDIR=/tmp
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in enmpc mhe; do
  rm -rf "$R/gpurun_out/tr_${TAG}_$c"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_${TAG}_$c -- python3 $R/bench.py --config $c --steps 20 --warmup 2 --no-cpu-baseline --groups 1 --repeats 3 > /dev/null 2>&1
  f=$(find $R/gpurun_out/tr_${TAG}_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"; head -4 $f | cut -d, -f1-4
done
cd $R
python3 tools/enmpc_gpu_vs_c.py 8 600 40 10 2>&1 | tail -3
for c in enmpc mhe; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['metric'][:44], round(d['value']), round(d['ms_per_step'],3), d['solver'])"; done
