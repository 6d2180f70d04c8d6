# Builds libJoshUpscale.so (C ABI + C++ plugin surface + gfx950 kernels) in-tree.
# No cmake on purpose: the GPU box runs the prebuilt .so that travels with the
# snapshot; `python -c "import __graft_entry__ as g; g.build()"` drives this.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := joshupscale_amd/csrc
OUT        := joshupscale_amd/lib
OBJ        := build/obj
CXXFLAGS   := -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -I$(CSRC) -Wall -Wextra \
              -Wno-unused-parameter
HIPFLAGS   := --offload-arch=$(ARCH) $(CXXFLAGS)
SRCS_CPP   := model.cpp engine.cpp c_api.cpp core_api.cpp log.cpp comm.cpp graphics.cpp dev_switch.cpp
SRCS_HIP   := conv_kernels.hip tower_kernels.hip frame_kernels.hip fp8_kernels.hip flow_kernels.hip res_block_kernels.hip splitk_kernels.hip tower8_kernels.hip
OBJS       := $(addprefix $(OBJ)/,$(SRCS_CPP:.cpp=.o)) $(addprefix $(OBJ)/,$(SRCS_HIP:.hip=.o))

# Two libraries from the same objects: the product (exactly the C ABI of include/joshupscale_amd.h + the C++ plugin
# surface of include/JoshUpscale/core.h) and its test flavour, in which c_api.cpp, graphics.cpp and dev_switch.cpp are
# compiled with -DJU_TEST_HOOKS: it additionally exports include/joshupscale_amd_test.h (ju_debug_*, ju_read_tensor,
# ju_time_steps) and reads the developer switches of csrc/dev_switch.h from the environment -- the product reads none of them.
HOOK_CPP   := c_api.cpp graphics.cpp dev_switch.cpp
OBJS_TEST  := $(filter-out $(addprefix $(OBJ)/,$(HOOK_CPP:.cpp=.o)),$(OBJS)) $(addprefix $(OBJ)/,$(HOOK_CPP:.cpp=_hooks.o))

all: $(OUT)/libJoshUpscale.so $(OUT)/libJoshUpscale_test.so

# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs.  The epilogues
# consume every accumulator with VALU ops; in AGPR form each value first costs a
# v_accvgpr_read (128 per layer and lane in the tower kernel, ~2 % of its time).
KERNELFLAGS := -mllvm -amdgpu-mfma-vgpr-form
$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/kernels.h $(CSRC)/kernel_common.h $(CSRC)/flow_block_common.h Makefile
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KERNELFLAGS) -c $< -o $@

$(OBJ)/%.o: $(CSRC)/%.cpp Makefile $(wildcard $(CSRC)/*.h) include/joshupscale_amd.h include/JoshUpscale/core.h
	@mkdir -p $(OBJ)
	$(HIPCC) -x hip $(HIPFLAGS) -c $< -o $@

$(OBJ)/%_hooks.o: $(CSRC)/%.cpp Makefile $(wildcard $(CSRC)/*.h) include/joshupscale_amd.h include/joshupscale_amd_test.h include/JoshUpscale/core.h
	@mkdir -p $(OBJ)
	$(HIPCC) -x hip $(HIPFLAGS) -DJU_TEST_HOOKS -c $< -o $@

$(OUT)/libJoshUpscale.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -Wl,--exclude-libs,ALL -ldl

$(OUT)/libJoshUpscale_test.so: $(OBJS_TEST)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS_TEST) -Wl,--exclude-libs,ALL -ldl

harness: $(OUT)/libJoshUpscale.so tools/plugin_harness.cpp
	g++ -O2 -std=c++17 -Iinclude tools/plugin_harness.cpp -o build/plugin_harness \
	    -L$(OUT) -lJoshUpscale -Wl,-rpath,'$$ORIGIN/../$(OUT)'

# probe build with the JU_FB_SKIP timing-ablation branches compiled in (tools/*_ablate.sh);
# the product library above never carries them
ablate:
	$(MAKE) OBJ=build/obj_ablate OUT=build/ablate KERNELFLAGS="$(KERNELFLAGS) -DJU_ABLATE"

clean:
	rm -rf build $(OUT)

.PHONY: all clean harness ablate
