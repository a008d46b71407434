#!/bin/bash
# Round 4's evidence in one GPU call: every step writes under gpurun_out/ (what travels back); copy into profiles/ afterwards.
# usage: bash scripts/r04_profiles.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
HEAD="bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
echo "== kernel stats of the headline step (default: two registrations in flight -- durations include the co-running lane)"
bash scripts/prof_stats.sh r04_bench 23 -- $HEAD > gpurun_out/r04_bench_stats.txt 2>&1 || exit 1
echo "== kernel stats of the serial step (--in-flight 1: every kernel alone on the chip)"
bash scripts/prof_stats.sh r04_bench_serial 23 -- $HEAD --in-flight 1 > gpurun_out/r04_bench_serial_stats.txt 2>&1 || exit 1
echo "== kernel stats of the batched cfg 1 RANSAC (ransac_hyp32_kernel)"
bash scripts/prof_stats.sh r04_cfg1b 4 -m ransac_hyp32 -- scripts/ransac_extras.py ransac_cfg1_batched > gpurun_out/r04_cfg1b_stats.txt 2>&1 || exit 1
echo "== sweep kernel stats (Poisson rows)"
bash scripts/prof_stats.sh r04_sweep 2 -- scripts/sweep_prof.py 2 > gpurun_out/r04_sweep_stats.txt 2>&1 || exit 1
echo "== counters of the sweep's exact re-rank (segp_rerank_pairs_kernel)"
bash scripts/pmc_counters.sh r04_segpairs segp_rerank_pairs gpurun_out/r04_pmc_segp_rerank_pairs.json -- scripts/sweep_prof.py 2 > gpurun_out/r04_pmc_segpairs.txt 2>&1 || exit 1
echo "== traffic of the headline step"
bash scripts/pmc_traffic.sh r04_step gpurun_out/r04_pmc_step_traffic.json "knn_candidates_f16,prep_model_f16" -- bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --in-flight 1 > gpurun_out/r04_step_traffic.txt 2>&1 || exit 1
echo "== descriptors at cfg 4: kernel stats"
bash scripts/prof_stats.sh r04_desc_1m 1 -m desc_kernel -- scripts/desc_dev_bench.py 1000000 1000000 > gpurun_out/r04_desc_1m_stats.txt 2>&1 || exit 1
echo "== full bench line"
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err || exit 1
echo done
