// runtime.hip -- version / error plumbing of libcloudaae_hip.so.
#include "common.h"
#include "../../include/cloudaae_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace cloudaae {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace cloudaae

CLOUDAAE_API int cloudaae_version(void) { return 100; }
CLOUDAAE_API const char *cloudaae_last_error(void) { return cloudaae::g_err; }

// ---- a second stream for work off the critical path ------------------------------------------
// The backward pass has products nobody waits for until the optimiser runs (dW of the edge
// convolutions and of dgcnn_agg) and small kernels that depend on nothing recent (reverse neighbour
// lists).  They CAN go to a low-priority side stream to fill the CUs the critical path leaves idle;
// measured at B=32 it loses (2.35 vs 2.25 ms/step: the overlapped GEMM takes L2 and CUs from gather-bound
// kernels, each cross-stream dependency costs microseconds), so the Python host leaves it off by default.
// Ordering between the two streams is expressed with events from a ring (an event is re-recorded
// only long after its waiters were enqueued; a wait refers to the record that preceded it).
namespace cloudaae {
constexpr int EVENT_RING = 256;
static hipEvent_t g_events[EVENT_RING];
static bool g_events_ready = false;
static unsigned g_next_event = 0;
static hipStream_t g_side = nullptr;
} // namespace cloudaae

CLOUDAAE_API int cloudaae_stream_wait(cloudaae_stream_t waiter, cloudaae_stream_t signaller)
{
    using namespace cloudaae;
    const char *name = "cloudaae_stream_wait";
    if (waiter == signaller)
        return 0;
    if (!g_events_ready) {
        for (int i = 0; i < EVENT_RING; ++i)
            CLOUDAAE_CHECK_HIP(hipEventCreateWithFlags(&g_events[i], hipEventDisableTiming), name);
        g_events_ready = true;
    }
    hipEvent_t e = g_events[g_next_event++ % EVENT_RING];
    CLOUDAAE_CHECK_HIP(hipEventRecord(e, (hipStream_t)signaller), name);
    CLOUDAAE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)waiter, e, 0), name);
    return 0;
}

CLOUDAAE_API cloudaae_stream_t cloudaae_side_stream(void)
{
    using namespace cloudaae;
    if (g_side == nullptr) {
        int least = 0, greatest = 0;    // least = the LOWEST priority the device offers
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
            least = 0;
        if (hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, least) != hipSuccess) {
            set_error("cloudaae_side_stream: cannot create a stream");
            g_side = nullptr;
        }
    }
    return (cloudaae_stream_t)g_side;
}
