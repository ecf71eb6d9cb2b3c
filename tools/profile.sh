#!/usr/bin/env bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats and separate PMC passes over a bench
# command, written under gpurun_out/prof_$TAG/ .  Counter passes are kept apart from --kernel-trace/--stats and
# from each other (TCC slot limits; see MI355X_MICROARCH.md).  The program after `--` is python3 itself.
#   tools/profile.sh TAG [default|nmc|nmc_eu]
set -uo pipefail
TAG="${1:-r02}"
MODE="${2:-default}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import importlib; print(importlib.import_module('monte-carlo-project-cuda_amd').capi.build_id())" > "$OUT/build_id.txt"
case "$MODE" in
  default) BENCH=(python3 "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --no-accuracy --no-sweep --no-nmc) ;;
  nmc)     BENCH=(python3 "$R/bench.py" --workload nmc --steps 1 --warmup 1 --no-cpu-baseline) ;;
  nmc_all) BENCH=(python3 "$R/tools/nmc_strategies.py") ;;
  nmc_eu)  BENCH=(python3 "$R/bench.py" --workload nmc --nmc-window european --steps 1 --warmup 0 --no-cpu-baseline) ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "${BENCH[@]}" > "$OUT/trace.log" 2>&1 \
 && timeout -k 10 400 rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/pmc_sq" -- "${BENCH[@]}" > "$OUT/pmc_sq.log" 2>&1 \
 && timeout -k 10 400 rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/pmc_sq2" -- "${BENCH[@]}" > "$OUT/pmc_sq2.log" 2>&1 \
 && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_wr" -- "${BENCH[@]}" > "$OUT/pmc_wr.log" 2>&1 \
 && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_rd" -- "${BENCH[@]}" > "$OUT/pmc_rd.log" 2>&1
rc=$?
# keep the merge small: drop everything but csv + logs
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' ! -name 'build_id.txt' -delete 2>/dev/null
echo "profile $TAG/$MODE rc=$rc"
exit $rc
