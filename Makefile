# Builds the HIP C-ABI library in-tree (travels to the GPU box with the snapshot).
# One object per source so that `make -j` rebuilds only what changed.
# --wrap: every kernel launch of the library passes through the launch log of csrc/wait.hip (the diagnostics of a
# host wait that times out name the last kernel enqueued on the stream).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard nodal_amd/csrc/*.hip)
HDR   := $(wildcard nodal_amd/csrc/*.h) include/nodal_hip.h
OBJDIR := build/obj
OBJ   := $(patsubst nodal_amd/csrc/%.hip,$(OBJDIR)/%.o,$(SRC))
LIB   := nodal_amd/libnodal_hip.so
CSVLIB := nodal_amd/libnodal_csv.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -pragma-unroll-threshold=1000000 -Wall -Wno-unused-function -Wno-pass-failed

all: $(LIB) $(CSVLIB) oracle

$(OBJDIR)/%.o: nodal_amd/csrc/%.hip $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--wrap=hipLaunchKernel -Wl,--wrap=hipExtLaunchKernel -o $@ $(OBJ)

# host-side netlist tokenizer (front-end, optional: fastparse.py falls back to pandas without it)
$(CSVLIB): nodal_amd/csrc/fastcsv.cpp
	g++ -O2 -std=c++17 -fPIC -shared -pthread -Wall -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) $(CSVLIB) $(OBJDIR)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
