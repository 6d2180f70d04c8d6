// Developer switches (A/B paths, cross-check kernels, traces) -- environment variables that exist in the TEST flavour of
// the library only.  The reference's shared library reads no environment at all and exports nothing but its
// JOSHUPSCALE_EXPORT symbols (reference core/CMakeLists.txt:29-36); a plugin host's environment must not be able to route
// frames onto an ablation or cross-check kernel.  So every such switch goes through devSwitch(): dev_switch.cpp is compiled
// twice (Makefile) -- for libJoshUpscale.so the function returns nullptr and the object holds no variable NAME at all
// (tests/test_c_abi.py holds `strings libJoshUpscale.so | grep ^JU_` to the list in INTEGRATION.md), for
// libJoshUpscale_test.so (-DJU_TEST_HOOKS) it is std::getenv of the table in dev_switch.cpp.  The callers (engine.cpp, the
// launchers in *_kernels.hip) are the same objects in both libraries and name a switch by this enum, never by its string.
//
// What the PRODUCT reads from the environment is documented in INTEGRATION.md and nothing else: JU_VERBOSE (log.cpp),
// JU_NO_GRAPH, JU_RESIDENT_RETRY, JU_LOOKAHEAD (engine.cpp).
#pragma once

namespace ju {

enum class Dev : int {
	Tail,         // JU_TAIL=fused|split        generator tail as its own launch / as two kernels
	Pack,         // JU_PACK=split              pack_frames_kernel as its own launch
	Pool,         // JU_POOL=split              max-pool as its own launch
	Upsample,     // JU_UPSAMPLE=split          bilinear x2 as its own launch
	FlowConv,     // JU_FLOW_CONV=generic       one conv_mfma_kernel launch per flow layer
	Tower,        // JU_TOWER=layers|convs      per-block / per-convolution tower
	Calibrate,    // JU_CALIBRATE=1             per-layer maxima (tools/calibrate.py)
	Flow,         // JU_FLOW=layers|convs       flow-resnet outside the resident kernel
	Direct,       // JU_DIRECT=0                device frames through the staging buffers
	DirectGraph,  // JU_DIRECT_GRAPH=0          device frames always as eager launches
	SyncSpinUs,   // JU_SYNC_SPIN_US=<us>       how long process() polls before it blocks
	TraceSteps,   // JU_TRACE_STEPS=<file>      launch list of the constructor's eager pass
	TraceNoSync,  // JU_TRACE_NOSYNC=1
	ResBlock,     // JU_RES_BLOCK=plain|tile    the unpipelined residual-block kernels
	FlowTile,     // JU_FLOW_TILE=<rows>        forced flow_block_kernel tile height
	FlowWide,     // JU_FLOW_WIDE=0|1           128-filter flow blocks as separate launches
	WavePrio,     // JU_WAVE_PRIO=<mode>        static wave priority (A/B)
	ConvDbuf,     // JU_CONV_DBUF=0|1           conv_mfma_kernel staging depth
	TowerFast,    // JU_TOWER_FAST=0            the resident tower's general schedule
	Fp8Grid,      // JU_FP8_GRID=<n>            grid of the per-conv 8-bit kernel
	Fp8Block,     // JU_FP8_BLOCK=solo|duo      form of res_block_fp8_kernel
	Count
};

// The switch's value in the process environment -- in the test flavour; nullptr, always, in the product library.
const char *devSwitch(Dev which);

}  // namespace ju
