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

// A small pool of timing-less events per (host thread, device) for the fork / join edges of anncur_eval_topk: created once, reused
// by every call of the thread (a recorded event may be re-recorded as soon as the wait that reads it has been enqueued).
namespace {
constexpr int EVENT_POOL = 16;
thread_local hipEvent_t tl_events[64][EVENT_POOL];
thread_local bool tl_events_ready[64] = {false};
}
int anncur_event_pool(hipEvent_t **out, int n) {
	int dev = 0;
	ANNCUR_HIP_OK(hipGetDevice(&dev));
	ANNCUR_REQUIRE(dev >= 0 && dev < 64 && n <= EVENT_POOL, ANNCUR_E_INVALID, "event_pool: bad device / count");
	if (!tl_events_ready[dev]) {
		for (int i = 0; i < EVENT_POOL; ++i) ANNCUR_HIP_OK(hipEventCreateWithFlags(&tl_events[dev][i], hipEventDisableTiming));
		tl_events_ready[dev] = true;
	}
	*out = tl_events[dev];
	return ANNCUR_OK;
}

#ifdef ANNCUR_TIMING_EXPERIMENTS
// Where a workgroup runs: {XCC_ID, HW_ID} (diagnostic build: which CUs a stream's CU mask leaves -- scripts/cumask_map.py)
namespace {
__global__ void where_kernel(uint32_t *out) {
	if (threadIdx.x == 0) {
		uint32_t xcc = 0, hw = 0;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hw));
		out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
	}
	// a little work so that the workgroups of a launch spread over every CU the queue may use
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) { }   // 20 us
}
}
extern "C" int anncur_debug_where(uint32_t *out, int n_wg, void *stream) {
	ANNCUR_REQUIRE(out && n_wg > 0, ANNCUR_E_INVALID, "debug_where: bad arguments");
	hipLaunchKernelGGL(where_kernel, dim3(n_wg), dim3(64), 0, (hipStream_t)stream, out);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}
#endif

extern "C" int anncur_version(void) { return 1000 * 0 + 3; }

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
