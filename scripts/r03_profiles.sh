#!/bin/bash
# Round 3's evidence in one GPU call: every step writes under gpurun_out/ (what travels back); parse / copy into profiles/ afterwards.
# usage: bash scripts/r03_profiles.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
HEAD="bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
echo "== kernel stats of the headline step"
bash scripts/prof_stats.sh r03_bench 23 -- $HEAD > gpurun_out/r03_bench_stats.txt 2>&1 || exit 1
echo "== traffic of the headline step"
bash scripts/pmc_traffic.sh r03_step gpurun_out/r03_pmc_step_traffic.json "knn_candidates_f16,prep_model_f16" -- bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03_step_traffic.txt 2>&1 || exit 1
echo "== SQ counters of the search kernel"
bash scripts/pmc_counters.sh r03_knn knn_candidates_f16 gpurun_out/r03_pmc_knn_f16_pipe.json -- bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03_knn_counters.txt 2>&1 || exit 1
echo "== SQ counters of the refit kernel"
bash scripts/pmc_counters.sh r03_mom rs_moments_mfma gpurun_out/r03_pmc_rs_moments_mfma.json -- bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03_mom_counters.txt 2>&1 || exit 1
echo "== sweep kernel stats"
bash scripts/prof_stats.sh r03_sweep 2 -- scripts/sweep_prof.py 2 > gpurun_out/r03_sweep_stats.txt 2>&1 || exit 1
echo "== align counters"
bash scripts/pmc_counters.sh r03_align align_points_knn_reg gpurun_out/r03_pmc_align.json -- scripts/align_dev_bench.py > gpurun_out/r03_align_counters.txt 2>&1 || exit 1
bash scripts/pmc_traffic.sh r03_align gpurun_out/r03_pmc_align_traffic.json "" -- scripts/align_dev_bench.py > gpurun_out/r03_align_traffic.txt 2>&1 || exit 1
echo "== descriptors at cfg 4 (1 M keypoints on a 1 M-point cloud): kernel stats, counters, traffic"
bash scripts/prof_stats.sh r03_desc_1m 1 -m desc_kernel -- scripts/desc_dev_bench.py 1000000 1000000 > gpurun_out/r03_desc_1m_stats.txt 2>&1 || exit 1
bash scripts/pmc_counters.sh r03_desc desc_kernel gpurun_out/r03_pmc_desc_kernel.json -- scripts/desc_dev_bench.py 1000000 1000000 > gpurun_out/r03_desc_counters.txt 2>&1 || exit 1
bash scripts/pmc_traffic.sh r03_desc gpurun_out/r03_pmc_desc_traffic.json "" -- scripts/desc_dev_bench.py 1000000 1000000 > gpurun_out/r03_desc_traffic.txt 2>&1 || exit 1
echo "== sweep bench"
python scripts/sweep_bench.py > gpurun_out/r03_sweep_bench.json 2> gpurun_out/r03_sweep_bench.err || exit 1
echo "== full bench line"
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err || exit 1
echo done
