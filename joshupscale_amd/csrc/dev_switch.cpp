// dev_switch.h: compiled once per library flavour (Makefile).  Without JU_TEST_HOOKS -- libJoshUpscale.so -- no developer
// switch exists: no name, no getenv.
#include "dev_switch.h"

#ifdef JU_TEST_HOOKS
#include <cstdlib>
#endif

namespace ju {

#ifdef JU_TEST_HOOKS
const char *devSwitch(Dev which) {
	static const char *const kNames[static_cast<int>(Dev::Count)] = {
	    "JU_TAIL", "JU_PACK", "JU_POOL", "JU_UPSAMPLE", "JU_FLOW_CONV", "JU_TOWER", "JU_CALIBRATE", "JU_FLOW", "JU_DIRECT",
	    "JU_DIRECT_GRAPH", "JU_SYNC_SPIN_US", "JU_TRACE_STEPS", "JU_TRACE_NOSYNC", "JU_RES_BLOCK", "JU_FLOW_TILE",
	    "JU_FLOW_WIDE", "JU_WAVE_PRIO", "JU_CONV_DBUF", "JU_TOWER_FAST", "JU_FP8_GRID", "JU_FP8_BLOCK",
	};
	const int i = static_cast<int>(which);
	if (i < 0 || i >= static_cast<int>(Dev::Count)) return nullptr;
	return std::getenv(kNames[i]);
}
#else
const char *devSwitch(Dev) {
	return nullptr;
}
#endif

}  // namespace ju
