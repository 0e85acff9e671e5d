// Launch tape: replays a recorded list of libpsld_hip launches (and the cross-stream edges between them) from C, so a
// training step at a launch-bound batch size costs the host ~2700 plain function calls instead of ~2700 trips through
// the Python executor (the reference's per-GPU batch of 16: train_uncond_psld.sh:25-30).
//
// A tape entry is a function index (psld_tape_fn_index) plus that entry point's arguments as 64-bit words.  The host
// side (psld_amd/tape.py) records them while the step runs once under stream capture, which also pins every buffer
// the step allocates at a fixed address; replay issues the same launches, on the same two streams, with the same
// pointers.  Unlike the captured hipGraph of the same step, the replayed launches are ordinary stream launches: the
// side stream really overlaps the backward chain, and there is no per-node graph edge cost (3.5 us per cross-stream
// edge).
#include <hip/hip_runtime_api.h>
#include <cstring>

#include "common.h"
#include "psld_hip.h"

namespace {

struct TapeFn {
    const char* name;
    int (*call)(const unsigned long long*);
    int nargs;
};

inline float u2f(unsigned long long w) {
    const unsigned int lo = (unsigned int)w;
    float f;
    std::memcpy(&f, &lo, 4);
    return f;
}
inline double u2d(unsigned long long w) {
    double d;
    std::memcpy(&d, &w, 8);
    return d;
}

#include "tape_stubs.inc"

constexpr int N_FNS = (int)(sizeof(TAPE_FNS) / sizeof(TAPE_FNS[0]));

}  // namespace

extern "C" {

int psld_tape_fn_index(const char* name) {
    for (int i = 0; i < N_FNS; ++i)
        if (!std::strcmp(TAPE_FNS[i].name, name)) return i;
    return -1;
}

void* psld_tape_event_create(void) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    return ev;
}

int psld_tape_event_destroy(void* event) {
    return event && hipEventDestroy(static_cast<hipEvent_t>(event)) != hipSuccess ? PSLD_ERR_LAUNCH : 0;
}

int psld_tape_replay(const psld_tape_entry* entries, int n, int* failed_at) {
    for (int i = 0; i < n; ++i) {
        const psld_tape_entry& e = entries[i];
        int rc = 0;
        if (e.fn == PSLD_TAPE_EDGE) {           // everything queued on a[0] so far happens before what a[1] gets next
            hipEvent_t ev = reinterpret_cast<hipEvent_t>(e.a[2]);
            if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(e.a[0])) != hipSuccess ||
                hipStreamWaitEvent(reinterpret_cast<hipStream_t>(e.a[1]), ev, 0) != hipSuccess) {
                psld_set_error("psld_tape_replay: cross-stream edge %d failed: %s", i, hipGetErrorString(hipGetLastError()));
                rc = PSLD_ERR_LAUNCH;
            }
        } else if (e.fn < 0 || e.fn >= N_FNS || e.nargs != TAPE_FNS[e.fn].nargs) {
            psld_set_error("psld_tape_replay: entry %d names function %d with %d arguments", i, e.fn, e.nargs);
            rc = PSLD_ERR_ARG;
        } else {
            rc = TAPE_FNS[e.fn].call(e.a);
        }
        if (rc) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return 0;
}

}  // extern "C"
