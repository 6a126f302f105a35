# Builds the HIP library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: the NTT's exactness argument needs every multiply and FMA exactly as written
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -std=c++17 -Wall
SRC := tfhe_amd/csrc/kernels.hip tfhe_amd/csrc/generic_kernels.hip tfhe_amd/csrc/wg_kernels.hip tfhe_amd/csrc/keygen_kernels.hip tfhe_amd/csrc/replica_kernels.hip tfhe_amd/csrc/capi.hip
HDR := tfhe_amd/csrc/kernels.hpp tfhe_amd/csrc/ntt512.hpp tfhe_amd/csrc/ntt512_half.hpp tfhe_amd/csrc/fft256_wave.hpp tfhe_amd/csrc/ntt_wg.hpp tfhe_amd/csrc/field.hpp include/tfhe_amd.h

all: tfhe_amd/libtfhe_amd.so oracle tests/cpp/dropin_test tools/gs_plan_dump tests/hip/convert_check

tfhe_amd/libtfhe_amd.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared $(SRC) -o $@

# behavioural tests of the drop-in headers (src/*.hpp) on top of the library
DROPIN_HDR := $(wildcard src/*.hpp)
tests/cpp/dropin_test: tests/cpp/test_dropin.cpp tests/cpp/test_dropin_tu2.cpp $(DROPIN_HDR) tfhe_amd/libtfhe_amd.so
	g++ -std=gnu++17 -O2 -Wall -pthread -I src -I include tests/cpp/test_dropin.cpp tests/cpp/test_dropin_tu2.cpp -o $@ \
	    -L tfhe_amd -ltfhe_amd -Wl,-rpath,'$$ORIGIN/../../tfhe_amd'

# the reduction plans of the inverse transforms, printed by host code (tests/test_reduction_plan.py)
tools/gs_plan_dump: tools/gs_plan_dump.hip tfhe_amd/csrc/ntt512.hpp tfhe_amd/csrc/ntt_wg.hpp
	$(HIPCC) -O1 --offload-arch=$(ARCH) -std=c++17 $< -o $@

# device check of the verification mode's double -> torus conversion (tests/test_gpu_convert.py)
tests/hip/convert_check: tests/hip/convert_check.hip tfhe_amd/csrc/fft256_wave.hpp tfhe_amd/csrc/ntt512_half.hpp tfhe_amd/csrc/ntt512.hpp
	$(HIPCC) -O2 --offload-arch=$(ARCH) -ffp-contract=off -std=c++17 $< -o $@

oracle:
	$(MAKE) -C oracle liboracle.so

ref: tfhe_amd/libtfhe_amd.so
	$(MAKE) -C oracle ref

clean:
	rm -f tfhe_amd/libtfhe_amd.so tests/cpp/dropin_test tools/gs_plan_dump tests/hip/convert_check
	$(MAKE) -C oracle clean

.PHONY: all oracle ref clean
