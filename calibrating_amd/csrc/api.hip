// api.hip -- error reporting and device probing for libcalibrating_amd.so
#include "common.hpp"

#include <cstring>

namespace camd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace camd

extern "C" {

const char* camd_last_error(void) { return camd::g_err; }

int camd_version(void) { return 100; }  // 0.1.0

int camd_device_ok(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        camd::set_error("no HIP device available (%s): the MI355X kernels cannot run and there is no CPU "
                        "fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        (void)hipGetLastError();
        return CAMD_ERR_NO_DEVICE;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        camd::set_error("hipGetDeviceProperties failed");
        return CAMD_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        camd::set_error("device %d is %s; this library carries gfx950 (MI355X) code only", dev,
                        prop.gcnArchName);
        return CAMD_ERR_NO_DEVICE;
    }
    return CAMD_OK;
}

int camd_stream_create_cu_mask(const uint32_t* cu_mask, int nwords, void** stream)
{
    if (!cu_mask || nwords <= 0 || !stream) { camd::set_error("camd_stream_create_cu_mask: bad arguments"); return CAMD_ERR_BAD_ARG; }
    int rc = camd_device_ok();
    if (rc != CAMD_OK) return rc;
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, cu_mask);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        camd::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
        return CAMD_ERR_HIP;
    }
    *stream = (void*)st;
    return CAMD_OK;
}

int camd_stream_destroy(void* stream)
{
    if (!stream) return CAMD_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        camd::set_error("hipStreamDestroy: %s", hipGetErrorString(e));
        return CAMD_ERR_HIP;
    }
    return CAMD_OK;
}

}  // extern "C"
