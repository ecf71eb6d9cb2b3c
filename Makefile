# Convenience targets; the driver entry points are __graft_entry__.py (build / smoke) and bench.py.
PY ?= python3

build:            ## hipcc --offload-arch=gfx950 -> monte-carlo-project-cuda_amd/libmcamd.so, oracle, oracle/_ref
	$(PY) __graft_entry__.py build

examples: build   ## shim-based drivers (plain g++)
	$(MAKE) -C examples

test-cpu: build   ## everything that runs without a GPU
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build   ## parity tests on an MI355X
	$(PY) -m pytest tests -x -q -m gpu

smoke:
	$(PY) __graft_entry__.py smoke

bench:
	$(PY) bench.py

slots:            ## recount VALU issue slots of the shipped inner loops -> profiles/
	$(PY) tools/count_valu_slots.py

tables:           ## regenerate the fp64 math tables (needs mpmath)
	$(PY) tools/gen_tables64.py

HIPCC ?= hipcc
PROBE_FLAGS = --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imonte-carlo-project-cuda_amd/csrc
PROBES = clock_probe init_probe store_variants ubench_bank ubench_valu ubench_hbm_write naive_port_baseline launch_overhead

probes: $(addprefix tools/,$(PROBES))   ## the diagnostic binaries behind profiles/ (run them on an MI355X)

tools/%: tools/%.hip
	$(HIPCC) $(PROBE_FLAGS) $< -o $@

clean:
	rm -rf monte-carlo-project-cuda_amd/csrc/build monte-carlo-project-cuda_amd/libmcamd.so oracle/liboracle.so oracle/_ref
	$(MAKE) -C examples clean

.PHONY: build examples test-cpu test-gpu smoke bench slots tables probes clean
