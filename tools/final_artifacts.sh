#!/usr/bin/env bash
# Runs on the GPU box (through gpurun): the round's GPU tests, smoke, every bench workload and the diagnostic probes,
# written under gpurun_out/$TAG/ (copied into profiles/ by hand afterwards, see profiles/README.md).
#   tools/final_artifacts.sh TAG
set -uo pipefail
TAG="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd "$R"
run() { local name="$1"; shift; echo "== $name"; timeout -k 10 600 "$@" > "$O/$name.json" 2> "$O/$name.err" || { echo "FAILED $name"; tail -5 "$O/$name.err"; return 1; }; }
timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > "$O/tests.log" 2>&1; tail -1 "$O/tests.log"
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
run bench_european252 python3 bench.py --gpus 1 --steps 20 --warmup 5 \
 && run bench_european252_f32 python3 bench.py --workload european252_f32 --steps 20 --warmup 5 --no-cpu-baseline \
 && run bench_vanilla1 python3 bench.py --workload vanilla1 --steps 20 --warmup 5 --no-cpu-baseline \
 && run bench_store python3 bench.py --workload store --steps 10 --warmup 3 --no-cpu-baseline \
 && run bench_nmc python3 bench.py --workload nmc --steps 3 --warmup 1 \
 && run bench_nmc_european_window python3 bench.py --workload nmc --nmc-window european --steps 1 --warmup 0 --no-cpu-baseline \
 && run bench_config5_1gpu python3 bench.py --global-paths 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-store-roofline --no-sweep --no-accuracy --no-nmc \
 && run bench_2rank_gloo_one_gpu python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline \
 && run bench_nmc_fused python3 bench.py --workload nmc --nmc-strategy fused --steps 3 --warmup 1 --no-cpu-baseline \
 && run nmc_strategies python3 tools/nmc_strategies.py \
 && run form_ab python3 tools/form_ab.py \
 && run fold_ab python3 tools/fold_ab.py \
 && run nmc_form_ab python3 tools/nmc_form_ab.py \
 && run fuzz_nmc python3 tools/fuzz_nmc.py --seconds 60 --seed 41 \
 && run fuzz_window_price python3 tools/fuzz_window_price.py --seconds 60 --seed 42
