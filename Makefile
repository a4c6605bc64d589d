# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
PY ?= python
.PHONY: build test test-gpu bench clean
build:            ## libbourse_amd.so (hipcc, gfx950) + the CPU oracle (test infrastructure)
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build       ## CPU suite: oracle vs reference KATs / golden fixtures, C ABI symbols, gloo sharding, plain-C client
	$(PY) -m pytest tests -x -q -m "not gpu"
test-gpu: build   ## parity suite on an MI355X (through the C ABI, bit-exact vs the oracle)
	$(PY) -m pytest tests -x -q -m gpu
bench:            ## one JSON line: book-steps/s, roofline, issue-slot utilisation, CPU baseline
	$(PY) bench.py
clean:
	rm -f bourse_amd/csrc/libbourse_amd.so oracle/libbourse_oracle.so oracle/libbourse_oracle_asan.so
