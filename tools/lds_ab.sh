#!/usr/bin/env bash
# Runs on the GPU box (through gpurun): same-box A/B of the shipped library against the LDS-planes variant
# (-DMCAMD_LDS_PLANES), timing (tools/ab_lib.py, alternating worker processes) and one PMC pass per library over
# the same worker.  Build the variant first, in the build container:
#   python3 -c "import importlib; importlib.import_module('monte-carlo-project-cuda_amd.build').build_variant('lds_planes', ['-DMCAMD_LDS_PLANES'])"
# Result: profiles/r03_lds_planes_ab.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/r03_lds"
mkdir -p "$O"
cd "$R"
python3 tools/ab_lib.py --rounds 3 monte-carlo-project-cuda_amd/libmcamd.so monte-carlo-project-cuda_amd/libmcamd_lds_planes.so > "$O/ab.jsonl" 2> "$O/ab.err"; echo "ab rc=$?"
cd /tmp && export TMPDIR=/tmp
for v in libmcamd libmcamd_lds_planes; do
  export MCAMD_LIB="$R/monte-carlo-project-cuda_amd/$v.so"
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_$v" -- python3 "$R/tools/ab_lib.py" --worker > "$O/pmc_$v.log" 2>&1; echo "pmc $v rc=$?"
done
find "$O" -type f ! -name '*.csv' ! -name '*.log' ! -name '*.jsonl' ! -name '*.err' -delete 2>/dev/null
cat "$O/ab.jsonl"
