#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace of the same command, PMC passes (each on its own, bounded).
#   bash tools/profile_round.sh          (writes under gpurun_out/prof/)
set -u
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 600 "$OUT/bench.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --warmup 0 > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 180 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- python3 tools/run_loop.py --batch 4096 --steps 100 > "$OUT/pmc_$tag.log" 2>&1; echo "pmc $tag rc=$?"
done
find "$OUT" -name "*kernel_stats.csv" | head -2
python3 tools/pmc_summary.py loop_kernel "$OUT/pmc_summary.json" "$OUT"/pmc_* > /dev/null 2>&1; cat "$OUT/pmc_summary.json" | head -60
