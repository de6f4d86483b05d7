# Builds the HIP C-ABI library in-tree (travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard nodal_amd/csrc/*.hip)
HDR   := $(wildcard nodal_amd/csrc/*.h) include/nodal_hip.h
LIB   := nodal_amd/libnodal_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -ffp-contract=on -mllvm -pragma-unroll-threshold=1000000 -Wall -Wno-unused-function

all: $(LIB) oracle

$(LIB): $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
