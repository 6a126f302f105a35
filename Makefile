# Builds the HIP library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: the NTT's exactness argument needs every multiply and FMA exactly as written
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -std=c++17 -Wall
SRC := tfhe_amd/csrc/kernels.hip tfhe_amd/csrc/generic_kernels.hip tfhe_amd/csrc/capi.hip
HDR := tfhe_amd/csrc/kernels.hpp tfhe_amd/csrc/ntt512.hpp tfhe_amd/csrc/field.hpp include/tfhe_amd.h

all: tfhe_amd/libtfhe_amd.so oracle

tfhe_amd/libtfhe_amd.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared $(SRC) -o $@

oracle:
	$(MAKE) -C oracle liboracle.so

ref:
	$(MAKE) -C oracle ref

clean:
	rm -f tfhe_amd/libtfhe_amd.so
	$(MAKE) -C oracle clean

.PHONY: all oracle ref clean
