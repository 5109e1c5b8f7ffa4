#!/bin/bash
# Collects everything profiles/ holds for the current build, on the GPU box:
#   scripts/collect_profiles.sh gpurun_out/<name>
# kernel trace of bench.py, PMC passes (separate runs), the bench line with the CPU baseline, and the other configurations.
set -u
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > "$out/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$out/trace.log"; exit 1; }
cp "$out"/trace/*/*kernel_stats.csv "$out/bench_kernel_stats.csv" 2>/dev/null
bash scripts/pmc_passes.sh "$out/pmc" > "$out/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.log"; exit 1; }
cp "$out/pmc/summary.txt" "$out/bench_pmc_summary.txt"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/bench.log" 2>&1 || { echo "bench failed"; tail -5 "$out/bench.log"; exit 1; }
tail -n 1 "$out/bench.log" > "$out/bench_line.json"
: > "$out/configs.jsonl"; : > "$out/configs_with_phase_stats.jsonl"; : > "$out/configs_with_phase_timing.jsonl"
for c in c1 c2 c2d10 c3 c4 c5 c6; do
  timeout -k 10 200 python3 scripts/run_config.py $c --steps 5 2>/dev/null | tail -n 1 >> "$out/configs.jsonl" || { echo "config $c failed"; exit 1; }
  ms=$(tail -n 1 "$out/configs.jsonl" | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])")
  timeout -k 10 200 python3 scripts/run_config.py $c --steps 1 --stats --kernel-ms $ms 2>/dev/null | tail -n 1 >> "$out/configs_with_phase_stats.jsonl" || { echo "config $c stats failed"; exit 1; }
  timeout -k 10 200 python3 scripts/run_config.py $c --steps 2 --timing 2>/dev/null | tail -n 1 >> "$out/configs_with_phase_timing.jsonl" || { echo "config $c timing failed"; exit 1; }
done
# counters of the other two kernels that matter: nested shells (scene in LDS, deep paths) and the 1M-sphere tree in HBM
PMC_CMD="scripts/run_config.py c4 --steps 1" bash scripts/pmc_passes.sh "$out/pmc_c4" > "$out/pmc_c4.log" 2>&1 && cp "$out/pmc_c4/summary.txt" "$out/c4_pmc_summary_2launches.txt"
PMC_CMD="scripts/run_config.py c5 --steps 1" bash scripts/pmc_passes.sh "$out/pmc_c5" > "$out/pmc_c5.log" 2>&1 && cp "$out/pmc_c5/summary.txt" "$out/c5_pmc_summary_2launches.txt"
# the device-built tree on the two large configurations, and warm scene-creation times of both builders
: > "$out/configs_lbvh.jsonl"
for c in c5 c6; do
  timeout -k 10 300 python3 scripts/run_config.py $c --steps 5 --builder lbvh 2>/dev/null | tail -n 1 >> "$out/configs_lbvh.jsonl" || echo "config $c lbvh failed"
  timeout -k 10 300 python3 scripts/run_config.py $c --steps 1 --stats --builder lbvh 2>/dev/null | tail -n 1 >> "$out/configs_lbvh.jsonl" || echo "config $c lbvh stats failed"
done
timeout -k 10 300 python3 scripts/scene_creation_times.py > "$out/scene_creation.jsonl" 2> "$out/scene_creation_laps.txt" || echo "scene creation times failed"
# the deferred-leaf experiment of the statistics build, beside the same build without it
: > "$out/deferred_leaf_experiment.jsonl"
for f in 0 32; do timeout -k 10 200 python3 scripts/run_config.py c2 --steps 3 --stats --flags $f 2>/dev/null | tail -n 1 >> "$out/deferred_leaf_experiment.jsonl"; done
# the 4K strong-scaling configuration on the one GPU (all 1000 samples)
timeout -k 10 300 python3 bench.py --config c3 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -n 1 > "$out/bench_c3_line.json" || echo "bench c3 failed"
timeout -k 10 60 raytracer_amd/bin/instr_cost > "$out/instr_cost.txt" 2>&1 || echo "instr_cost failed"
timeout -k 10 120 raytracer_amd/bin/gather_rate > "$out/gather_rate.txt" 2>&1 || echo "gather_rate failed"
timeout -k 10 60 raytracer_amd/bin/inner_step > "$out/inner_step.txt" 2>&1 || echo "inner_step failed"
echo collected
