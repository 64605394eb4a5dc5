# convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly)
build:
	python -c "import __graft_entry__ as g; g.build()"
test:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -m pytest tests -q -m gpu
bench:
	python bench.py
asan:                          # CPU sanitizer build of the C host, the oracle and the C consumer (tests/test_asan.py runs them)
	$(MAKE) -C aligntools/c_amd asan
goldens:                       # needs /root/reference (the real reference is compiled in place, never copied)
	$(MAKE) -C oracle && python oracle/make_golden.py && python oracle/make_cli_golden.py
.PHONY: build test test-gpu bench goldens asan
