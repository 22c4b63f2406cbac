// Error reporting and device introspection for libanncur_hip.
#include <string.h>
#include "common.hpp"

namespace {
thread_local char g_err[512] = "";
}

void anncur_set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" int anncur_version(void) { return 1000 * 0 + 1; }

extern "C" const char *anncur_last_error(void) { return g_err; }

extern "C" int anncur_device_info(int *n_cu, int *wave_size, char *arch_name, int arch_name_len) {
	int dev = 0;
	ANNCUR_HIP_OK(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	ANNCUR_HIP_OK(hipGetDeviceProperties(&prop, dev));
	if (n_cu) *n_cu = prop.multiProcessorCount;
	if (wave_size) *wave_size = prop.warpSize;
	if (arch_name && arch_name_len > 0) {
		strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
		arch_name[arch_name_len - 1] = 0;
	}
	return ANNCUR_OK;
}
