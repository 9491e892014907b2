# Convenience targets; the contract entry points are __graft_entry__.py (build/smoke), tests/ and bench.py.
.PHONY: build test test-gpu bench smoke clean
build:
	$(MAKE) -C instant-distance_amd/csrc
	$(MAKE) -C oracle
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
smoke: build
	python -c "import __graft_entry__ as g; g.smoke()"
bench: build
	python bench.py
clean:
	$(MAKE) -C instant-distance_amd/csrc clean
	rm -rf tests/simt/_build oracle/*.so
