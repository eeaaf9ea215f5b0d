# Convenience targets (the driver uses __graft_entry__.py, pytest and bench.py directly).
.PHONY: build test-cpu test-cpu-asan test-gpu bench smoke clean

build:
	python -c "import __graft_entry__ as g; g.build()"

test-cpu: build
	python -m pytest tests -q -m "not gpu"

# One sanitizer pass over everything of this repository that runs on the CPU (GPU AddressSanitizer is not available on
# the pool): the oracle (oracle/_asan/libhh_oracle.so, loaded into a python that has libasan preloaded) driven by its
# own pin and property tests, and the host builds of the device headers — hh_math.h, hh_bessel.h (g++), hh_rng.h
# (hipcc, host side only) — each under AddressSanitizer + UBSan.  detect_leaks=0: the interpreter's own allocations.
ASAN_RT := $(shell gcc -print-file-name=libasan.so)
test-cpu-asan:
	$(MAKE) -C oracle -s asan
	HH_SANITIZE=1 LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	  python -m pytest tests/test_oracle_pins.py tests/test_properties.py tests/test_math_host.py tests/test_bessel_host.py \
	  tests/test_rng_host.py -q -m "not gpu" -p no:cacheprovider

test-gpu: build
	python -m pytest tests -q -m gpu

smoke: build
	python -c "import __graft_entry__ as g; g.smoke()"

bench: build
	python bench.py

clean:
	rm -f hedgehog.jl_amd/lib/*.so oracle/*.so tools/ubench/valu_rates
	rm -rf hedgehog.jl_amd/lib/variants
