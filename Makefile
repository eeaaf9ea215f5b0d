# Convenience targets (the driver uses __graft_entry__.py, pytest and bench.py directly).
.PHONY: build test-cpu test-gpu bench smoke clean

build:
	python -c "import __graft_entry__ as g; g.build()"

test-cpu: build
	python -m pytest tests -q -m "not gpu"

test-gpu: build
	python -m pytest tests -q -m gpu

smoke: build
	python -c "import __graft_entry__ as g; g.smoke()"

bench: build
	python bench.py

clean:
	rm -f hedgehog.jl_amd/lib/*.so oracle/*.so tools/ubench/valu_rates
	rm -rf hedgehog.jl_amd/lib/variants
