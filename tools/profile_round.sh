#!/bin/bash
# Round profile on the GPU box: bench line (the driver's command), rocprofv3 kernel trace of the same workload, PMC passes (each on
# its own, bounded).   bash tools/profile_round.sh [round tag, default r02] [nmpc|enmpc|mhe]     (writes under gpurun_out/prof/; copy into profiles/)
# With "nmpc": the non-linear workload (bench.py --config nmpc), files <tag>_nmpc_*; its dominant kernel is the wave-style launch of the
# split pipeline, one launch = one step of every instance.
set -u
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
if [ "${2:-}" = "nmpc" ]; then
    OUT=$OUT/nmpc; rm -rf "$OUT"; mkdir -p "$OUT"
    # (--groups 1: the batch on one stream - launches that overlap on streams of their own have no duration of their own)
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config nmpc --steps 20 --warmup 2 --no-cpu-baseline --groups 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
    CMD="python3 bench.py --config nmpc --steps 20 --warmup 0 --repeats 2 --no-cpu-baseline --groups 1"
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
        tag=$(echo $pass | cut -d' ' -f1)
        timeout 180 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- $CMD > "$OUT/pmc_$tag.log" 2>&1; echo "pmc $tag rc=$?"
    done
    find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${TAG}_nmpc_kernel_stats.csv"
    python3 tools/pmc_summary.py nmpc_loop_kernel_wv "$OUT/${TAG}_nmpc_pmc_summary.json" 16384 "$CMD" "$OUT"/pmc_* > /dev/null 2>&1; head -c 1500 "$OUT/${TAG}_nmpc_pmc_summary.json"
    # the bench line last: it cites the PMC summary of the same round (bench.py: PROFILE_ROUND), which has to exist first
    cp "$OUT/${TAG}_nmpc_pmc_summary.json" profiles/
    timeout 600 python3 bench.py --config nmpc --steps 20 --warmup 2 > "$OUT/${TAG}_nmpc_bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 600 "$OUT/${TAG}_nmpc_bench.json"
    head -5 "$OUT/${TAG}_nmpc_kernel_stats.csv"
    exit 0
fi
if [ "${2:-}" = "enmpc" ] || [ "${2:-}" = "mhe" ]; then      # the economic workloads (bench.py --config enmpc | mhe), files <tag>_<config>_*
    CFG=$2; OUT=$OUT/$CFG; rm -rf "$OUT"; mkdir -p "$OUT"
    # (--groups 1: the batch on one stream - launches that overlap on streams of their own have no duration of their own)
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --config $CFG --steps 20 --warmup 2 --no-cpu-baseline --groups 1 > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
    CMD="python3 bench.py --config $CFG --steps 20 --warmup 0 --repeats 2 --no-cpu-baseline --groups 1"
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
        tag=$(echo $pass | cut -d' ' -f1)
        timeout 180 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- $CMD > "$OUT/pmc_$tag.log" 2>&1; echo "pmc $tag rc=$?"
    done
    find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${TAG}_${CFG}_kernel_stats.csv"
    B=16384; [ "$CFG" = "mhe" ] && B=4096
    KERN=enmpc_ocp_kernel; [ "$CFG" = "mhe" ] && KERN=enmpc_mhe_kernel      # the kernel bench.py prices: the largest share of the device time
    python3 tools/pmc_summary.py $KERN "$OUT/${TAG}_${CFG}_pmc_summary.json" $B "$CMD" "$OUT"/pmc_* > /dev/null 2>&1; head -c 1200 "$OUT/${TAG}_${CFG}_pmc_summary.json"
    cp "$OUT/${TAG}_${CFG}_pmc_summary.json" profiles/      # (the bench line last: it cites this summary)
    timeout 600 python3 bench.py --config $CFG --steps 20 --warmup 2 > "$OUT/${TAG}_${CFG}_bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 600 "$OUT/${TAG}_${CFG}_bench.json"
    head -6 "$OUT/${TAG}_${CFG}_kernel_stats.csv"
    exit 0
fi
rm -rf "$OUT"/pmc_* "$OUT"/trace; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
CMD="python3 bench.py --steps 20 --warmup 0 --repeats 4 --no-cpu-baseline --no-other-configs"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 180 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$tag" -- $CMD > "$OUT/pmc_$tag.log" 2>&1; echo "pmc $tag rc=$?"
done
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${TAG}_kernel_stats.csv"
python3 tools/pmc_summary.py loop_kernel_wv "$OUT/${TAG}_pmc_summary.json" 81920 "$CMD" "$OUT"/pmc_* > /dev/null 2>&1; head -c 1500 "$OUT/${TAG}_pmc_summary.json"
cp "$OUT/${TAG}_pmc_summary.json" profiles/      # (the bench line last: it cites this summary; its other_configs cite theirs - run the nmpc / enmpc / mhe rounds first)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 900 "$OUT/${TAG}_bench.json"
head -5 "$OUT/${TAG}_kernel_stats.csv"
