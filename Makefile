# Builds the gfx950 C-ABI library in-tree (the .so travels to the GPU box with the
# snapshot; it is git-ignored).  No cmake needed: a handful of hipcc invocations.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := embodied_clip_amd/csrc
OUT   := embodied_clip_amd/lib/libec_amd.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++20 -fPIC -Wall -Wno-unused-function -Wno-unused-but-set-variable -ffp-contract=fast $(EXTRA)

all: $(OUT)

build/%.o: $(CSRC)/%.hip $(CSRC)/common.h include/ec_amd.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(OUT)

.PHONY: all clean
