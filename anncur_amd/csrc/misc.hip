// Error reporting and device introspection for libanncur_hip.
#include <string.h>
#include <mutex>
#include <vector>
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

// hipFuncSetAttribute and the CU count belong to a DEVICE: both are remembered per (function, device) / per device, so a process
// that drives several GPUs (CURApprox(device=...), --device cuda:N) sets the > 64 KiB dynamic-LDS attribute on each of them.
namespace {
struct AttrEntry { const void *fn; int dev; int bytes; };
std::mutex g_attr_mu;
std::vector<AttrEntry> g_attr;
int g_cu[64] = {0};
}

int anncur_ensure_dyn_lds(const void *fn, int bytes) {
	int dev = 0;
	ANNCUR_HIP_OK(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lock(g_attr_mu);
	for (auto &e : g_attr)
		if (e.fn == fn && e.dev == dev) {
			if (e.bytes >= bytes) return ANNCUR_OK;
			ANNCUR_HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
			e.bytes = bytes;
			return ANNCUR_OK;
		}
	ANNCUR_HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
	g_attr.push_back({fn, dev, bytes});
	return ANNCUR_OK;
}

int anncur_num_cu() {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
	std::lock_guard<std::mutex> lock(g_attr_mu);
	if (g_cu[dev] == 0) {
		hipDeviceProp_t prop;
		g_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;  // 256 = MI355X
	}
	return g_cu[dev];
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
