// Error reporting and library info for the C ABI (include/locate_hip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_last_error[512] = "";

void locate_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

LOCATE_API const char* locate_last_error(void) { return g_last_error; }

LOCATE_API int locate_abi_version(void) { return 10; }   // 10: locate_pool2_upsample2x_fwd / _bwd; 9: fused RootTanh epilogues on every map (act_out, mul_pre, out_absmax; locate_conv_dgrad takes act_epilogue); 8: window panels / window kernels (panel format bit 2, precision bit 4, locate_conv_win_ok, pack passes); 7: direct panel re-packing (locate_conv_pack_job weight_absmax, Nadam record with absmax words), deferred / call-aligned split reductions; 6: locate_add3, locate_wgrad_batch*; 5: fp16-piece contractions (precision 2, panel format bit, absmax words); 4: batched finalisers

// Fills name (<= name_len bytes), compute-unit count and wavefront size of the current device.
// The library only ships gfx950 code objects; callers use this to fail loudly on anything else.
LOCATE_API int locate_device_info(char* arch_name, int name_len, int* cu_count, int* wave_size) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        locate_set_error("locate_device_info: no HIP device");
        return LOCATE_ERR_LAUNCH;
    }
    if (arch_name && name_len > 0) snprintf(arch_name, name_len, "%s", prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    return LOCATE_OK;
}
