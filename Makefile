# Builds liboniris_hip.so (gfx950) in-tree.  `make -j8`
HIPCC ?= /opt/rocm/bin/hipcc
CSRC := autoregressive_diffusion_amd/csrc
OUT := autoregressive_diffusion_amd/liboniris_hip.so
HIPFLAGS := --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Wno-unused-result
SRCS := $(wildcard $(CSRC)/*.hip) $(CSRC)/misc.cpp
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) include/oniris.h

all: $(OUT)

build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/misc.cpp.o: $(CSRC)/misc.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf build $(OUT)
