// Version / error plumbing of libhalo_hip.so.
#include "halo_common.hpp"

namespace halo {
char *err_buf()
{
    static thread_local char buf[ERR_LEN] = {0};
    return buf;
}
}  // namespace halo

extern "C" int halo_version(void) { return HALO_ABI_VERSION; }
extern "C" const char *halo_last_error(void) { return halo::err_buf(); }

extern "C" void *halo_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { halo::fail(HALO_E_LAUNCH, "hipEventCreate failed"); return nullptr; }
    return (void *)e;
}
extern "C" int halo_event_record(void *event, void *stream)
{
    if (!event) return halo::fail(HALO_E_ARG, "halo_event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    return e == hipSuccess ? HALO_OK : halo::fail(HALO_E_LAUNCH, "hipEventRecord: %s", hipGetErrorString(e));
}
extern "C" int halo_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return halo::fail(HALO_E_ARG, "halo_event_elapsed_ms: null argument");
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return e == hipSuccess ? HALO_OK : halo::fail(HALO_E_LAUNCH, "hipEventElapsedTime: %s", hipGetErrorString(e));
}
extern "C" int halo_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy((hipEvent_t)event);
    return HALO_OK;
}
