# Builds liboniris_hip.so (gfx950) in-tree.  `make -j8`
HIPCC ?= /opt/rocm/bin/hipcc
CSRC := autoregressive_diffusion_amd/csrc
OUT := autoregressive_diffusion_amd/liboniris_hip.so
HIPFLAGS := --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Wno-unused-result
SRCS := $(wildcard $(CSRC)/*.hip) $(CSRC)/misc.cpp
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) include/oniris.h

all: $(OUT)

# attention: MFMA results in arch VGPRs (every S^T accumulator feeds v_exp_f32; the AGPR form hipcc otherwise picks for
# kernels it cannot fit into 256 registers costs one v_accvgpr_read per element)
build/attention.hip.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form=1
build/attention_stamp.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form=1

build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/misc.cpp.o: $(CSRC)/misc.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

# diagnostic build: the attention kernels with in-kernel cycle stamps (scratch/attn_stamp.py loads it through
# ONIRIS_LIB_NAME); never used by the product path
STAMP_OUT := autoregressive_diffusion_amd/liboniris_hip_stamp.so
build/attention_stamp.o: $(CSRC)/attention.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DATTN_STAMP -c $< -o $@
build/conv_fwd_s2ctx_stamp.o: $(CSRC)/conv_fwd_s2ctx.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DCONV_STAMP -c $< -o $@
stamp: build/attention_stamp.o build/conv_fwd_s2ctx_stamp.o $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $(STAMP_OUT) build/attention_stamp.o build/conv_fwd_s2ctx_stamp.o $(filter-out build/attention.hip.o build/conv_fwd_s2ctx.hip.o,$(OBJS))

clean:
	rm -rf build $(OUT)

# experiment builds of the gated-conv translation unit with other -D settings (VNAME / VDEF), e.g.
#   make variant VNAME=m816 VDEF="-DGLDS_MID_OWN=8 -DGLDS_MID_CTX=16"   -> autoregressive_diffusion_amd/liboniris_hip_m816.so
#   make variant VSRC=conv_wgrad VNAME=wro0 VDEF="-DWGRAD_ROWORDER=0"     (another translation unit: VSRC, default conv_fwd_s2ctx)
VSRC ?= conv_fwd_s2ctx
variant: $(OBJS)
	$(HIPCC) $(HIPFLAGS) $(VDEF) -c $(CSRC)/$(VSRC).hip -o build/$(VSRC)_$(VNAME).o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o autoregressive_diffusion_amd/liboniris_hip_$(VNAME).so build/$(VSRC)_$(VNAME).o $(filter-out build/$(VSRC).hip.o,$(OBJS))
