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

clean:
	rm -rf monte-carlo-project-cuda_amd/csrc/build monte-carlo-project-cuda_amd/libmcamd.so oracle/liboracle.so oracle/_ref
	$(MAKE) -C examples clean

.PHONY: build examples test-cpu test-gpu smoke bench slots tables clean
