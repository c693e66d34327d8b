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

CLOUDAAE_API int cloudaae_version(void) { return CLOUDAAE_ABI_VERSION; }

// ---- development knobs (common.h) ---------------------------------------------------------------------
#include <stdlib.h>
#include <string.h>
#include <mutex>
namespace cloudaae {
constexpr int MAX_KNOBS = 128;
static Knob g_knobs[MAX_KNOBS];
static char g_knob_names[MAX_KNOBS][48];
static int g_nknobs = 0;
static std::mutex g_knob_mutex;
// A full table never aliases names: every further name shares ONE slot that is never set (a reader falls back to its
// default; cloudaae_set_knob / _unset_knob refuse it).  The fields are plain ints: knobs are set between launches by the
// thread that launches (tests, sweeps), not concurrently with it.
static Knob g_knob_overflow = {"(knob table full)", 0, false};
Knob *knob_slot(const char *name)
{
    std::lock_guard<std::mutex> lock(g_knob_mutex);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].name, name) == 0)
            return &g_knobs[i];
    if (g_nknobs == MAX_KNOBS)
        return &g_knob_overflow;
    Knob &k = g_knobs[g_nknobs];
    strncpy(g_knob_names[g_nknobs], name, sizeof(g_knob_names[0]) - 1);
    k.name = g_knob_names[g_nknobs];
    const char *e = getenv(name);       // the only place the library reads the environment
    k.set = e != nullptr && *e != 0;
    k.value = k.set ? atoi(e) : 0;
    ++g_nknobs;
    return &k;
}
} // namespace cloudaae

CLOUDAAE_API int cloudaae_set_knob(const char *name, int value)
{
    if (name == nullptr || strlen(name) >= 48) {
        cloudaae::set_error("cloudaae_set_knob: bad name");
        return (int)hipErrorInvalidValue;
    }
    cloudaae::Knob *k = cloudaae::knob_slot(name);
    if (k == &cloudaae::g_knob_overflow) {
        cloudaae::set_error("cloudaae_set_knob: knob table full (%d names)", cloudaae::MAX_KNOBS);
        return (int)hipErrorInvalidValue;
    }
    k->value = value;
    k->set = true;
    return 0;
}

CLOUDAAE_API int cloudaae_unset_knob(const char *name)
{
    if (name == nullptr || strlen(name) >= 48) {
        cloudaae::set_error("cloudaae_unset_knob: bad name");
        return (int)hipErrorInvalidValue;
    }
    cloudaae::Knob *k = cloudaae::knob_slot(name);
    if (k == &cloudaae::g_knob_overflow) {
        cloudaae::set_error("cloudaae_unset_knob: knob table full (%d names)", cloudaae::MAX_KNOBS);
        return (int)hipErrorInvalidValue;
    }
    k->set = false;
    return 0;
}

// ---- CRC-32C (Castagnoli) on the HOST: the checksum of the TFRecord framing (train_cloudAAE_ycbv.py:80-135
// reads such files) and of TensorFlow checkpoints (tf.train.Saver, :276 / :418-430), which TensorFlow computes
// with the SSE4.2 crc32 instruction; a byte-at-a-time Python loop needs minutes for the 262 MB of a checkpoint.
namespace cloudaae {
static unsigned crc32c_table(const unsigned char *p, unsigned long long n, unsigned c)
{
    static unsigned tab[8][256];
    static bool ready = false;
    if (!ready) {
        for (unsigned i = 0; i < 256; ++i) {
            unsigned v = i;
            for (int k = 0; k < 8; ++k)
                v = (v >> 1) ^ ((v & 1) ? 0x82F63B78u : 0u);
            tab[0][i] = v;
        }
        for (unsigned i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t)
                tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
        ready = true;
    }
    while (n >= 8) {            // slicing by eight
        const unsigned lo = ((unsigned)p[0] | (unsigned)p[1] << 8 | (unsigned)p[2] << 16 | (unsigned)p[3] << 24) ^ c;
        c = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
            tab[3][p[4]] ^ tab[2][p[5]] ^ tab[1][p[6]] ^ tab[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--)
        c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static unsigned crc32c_hw(const unsigned char *p, unsigned long long n, unsigned c)
{
    unsigned long long c64 = c;
    while (n >= 8) {
        unsigned long long v;
        __builtin_memcpy(&v, p, 8);
        c64 = __builtin_ia32_crc32di(c64, v);
        p += 8;
        n -= 8;
    }
    c = (unsigned)c64;
    while (n--)
        c = __builtin_ia32_crc32qi(c, *p++);
    return c;
}
#endif
} // namespace cloudaae

// crc = cloudaae_crc32c(data, n, 0) for a whole buffer; pass the previous result to continue over a next piece
CLOUDAAE_API unsigned cloudaae_crc32c(const void *data, unsigned long long n, unsigned crc)
{
    const unsigned char *p = (const unsigned char *)data;
    unsigned c = ~crc;
#if defined(__x86_64__)
    if (__builtin_cpu_supports("sse4.2"))
        return ~cloudaae::crc32c_hw(p, n, c);
#endif
    return ~cloudaae::crc32c_table(p, n, c);
}
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
