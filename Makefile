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

# Tools-only build: the same sources with -DEC_TOOLS -DEC_CONV8_PROFILE -- adds the profiling exports that are NOT part of
# the product library or of include/ec_amd.h (ec_bneck_set_debug, ec_debug_stamps, the EC_CONV_ABLATE instances of the
# 8-wave kernel).  Used by tools/bench_bneck.py --stamps and tools/stamps8.py through tools/_toolslib.py.
TOOLS_OUT  := embodied_clip_amd/lib/libec_amd_tools.so
TOOLS_OBJS := $(patsubst $(CSRC)/%.hip,build_tools/%.o,$(SRCS))
build_tools/%.o: $(CSRC)/%.hip $(CSRC)/common.h include/ec_amd.h
	@mkdir -p build_tools
	$(HIPCC) $(FLAGS) -DEC_TOOLS -DEC_CONV8_PROFILE -c $< -o $@
$(TOOLS_OUT): $(TOOLS_OBJS)
	@mkdir -p $(dir $(TOOLS_OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(TOOLS_OBJS)
tools: $(TOOLS_OUT)

clean:
	rm -rf build build_tools $(OUT) $(TOOLS_OUT)

.PHONY: all clean tools
