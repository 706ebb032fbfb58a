// libvcx: error reporting, device query, HIP-event profiling of kernel families.
#include "vcx_common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void vcx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vcx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    // hipErrorNotReady is never the result of a launch: it is the sticky residue of a hipEventQuery / hipStreamQuery poll made by
    // the host framework on this thread (PyTorch's caching allocator polls events when tensors cross streams)
    if (e != hipSuccess && e != hipErrorNotReady) {
        vcx_set_error("%s: %s", what, hipGetErrorString(e));
        return VCX_ELAUNCH;
    }
    return VCX_OK;
}

extern "C" int vcx_abi_version(void) { return VCX_ABI_VERSION; }
extern "C" const char* vcx_last_error(void) { return g_err; }

extern "C" int vcx_device_arch(char* name_host, int len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        vcx_set_error("no HIP device");
        return VCX_ENODEV;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        vcx_set_error("hipGetDeviceProperties failed");
        return VCX_ENODEV;
    }
    if (name_host && len > 0) {
        strncpy(name_host, p.gcnArchName, len - 1);
        name_host[len - 1] = 0;
    }
    return VCX_OK;
}

// ---------------------------------------------------------------------------------------
// Experiment knobs: process-wide atomics, seeded once from VCX_TUNE_<NAME>; the dispatchers read them with a relaxed load
// (no getenv on the launch path).
// ---------------------------------------------------------------------------------------
static std::atomic<int> g_tune[VCX_TUNE_COUNT];
static std::once_flag g_tune_once;
static const struct { const char* name; int dflt; } g_tune_def[VCX_TUNE_COUNT] = {
    {"GEMM_CFG", -1}, {"GEMM_DMA", 1}, {"FLASH_QB", 0}, {"XATTN_RESIDENT", 1}, {"FLASH_IMPL", 0}, {"EXP0", 0}, {"EXP1", 0}, {"GEMM_WS", 1}};

static void tune_init() {
    for (int i = 0; i < VCX_TUNE_COUNT; ++i) {
        char key[64];
        snprintf(key, sizeof(key), "VCX_TUNE_%s", g_tune_def[i].name);
        const char* e = getenv(key);
        g_tune[i].store((e && e[0]) ? atoi(e) : g_tune_def[i].dflt, std::memory_order_relaxed);
    }
}
int vcx_tune(int knob) {
    std::call_once(g_tune_once, tune_init);
    return g_tune[knob].load(std::memory_order_relaxed);
}
extern "C" int vcx_tune_get(int knob) {
    if (knob < 0 || knob >= VCX_TUNE_COUNT) { vcx_set_error("vcx_tune_get: no knob %d", knob); return VCX_EINVAL; }
    return vcx_tune(knob);
}
extern "C" int vcx_tune_set(int knob, int value) {
    if (knob < 0 || knob >= VCX_TUNE_COUNT) { vcx_set_error("vcx_tune_set: no knob %d", knob); return VCX_EINVAL; }
    std::call_once(g_tune_once, tune_init);
    return g_tune[knob].exchange(value, std::memory_order_relaxed);
}

// ---------------------------------------------------------------------------------------
// Profiling: HIP-event brackets around RUNS of launches of one kernel family on one stream.  A run is opened by the
// first launch of a family and closed - one hipEventRecord - when a launch of another family (or vcx_profile_end)
// follows; the closing event of a run is the opening event of the next one.  Each hipEventRecord costs ~2-3 us of GPU
// time, so bracketing runs instead of launches roughly halves the overhead inside bench.py's timed region.
// ---------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    int start_rec;   // >= 0: the start timestamp is record start_rec's END event
    int family;
    double launches, flops, bytes;
    hipStream_t stream;
};
// The profiler is one process-wide recorder (bench.py brackets its timed region with it).  Entry points may be called from
// several host threads: every access to the recorder below happens under g_prof_mu; calls made while profiling is off only
// read the atomic flag, so the compute entry points stay lock-free and re-entrant in normal operation.
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;
static int g_nrec = 0;
static std::atomic<bool> g_prof_on{false};
static int g_open = -1;         // the run still open (its end event not yet recorded)

static void prof_close_open_run() {
    if (g_open >= 0) {
        (void)hipEventRecord(g_recs[g_open].b, g_recs[g_open].stream);
        g_open = -1;
    }
}

extern "C" int vcx_profile_begin(int max_records) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (max_records <= 0) max_records = 1;
    while ((int)g_recs.size() < max_records) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            vcx_set_error("hipEventCreate failed");
            return VCX_ELAUNCH;
        }
        r.family = 0;
        r.launches = r.flops = r.bytes = 0;
        r.start_rec = -1;
        r.stream = nullptr;
        g_recs.push_back(r);
    }
    g_nrec = 0;
    g_open = -1;
    g_prof_on.store(true, std::memory_order_release);
    return VCX_OK;
}

extern "C" int vcx_profile_end(double* out_host) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    prof_close_open_run();
    g_prof_on.store(false, std::memory_order_release);
    for (int i = 0; i < VCX_PROF_FAMILIES * 4; ++i) out_host[i] = 0.0;
    for (int i = 0; i < g_nrec; ++i) {
        ProfRec& r = g_recs[i];
        if (hipEventSynchronize(r.b) != hipSuccess) {
            vcx_set_error("hipEventSynchronize failed");
            return VCX_ELAUNCH;
        }
        float ms = 0.f;
        hipEvent_t start = r.start_rec >= 0 ? g_recs[r.start_rec].b : r.a;
        if (hipEventElapsedTime(&ms, start, r.b) != hipSuccess) {
            vcx_set_error("hipEventElapsedTime failed");
            return VCX_ELAUNCH;
        }
        double* o = out_host + 4 * r.family;
        o[0] += r.launches;
        o[1] += (double)ms;
        o[2] += r.flops;
        o[3] += r.bytes;
    }
    g_nrec = 0;
    return VCX_OK;
}

VcxProfScope::VcxProfScope(int family, hipStream_t stream, double flops, double bytes) : rec(-1), s(stream) {
    if (!g_prof_on.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    if (g_open >= 0 && g_recs[g_open].family == family && g_recs[g_open].stream == s) {   // same run: just account
        ProfRec& r = g_recs[g_open];
        r.launches += 1.0;
        r.flops += flops;
        r.bytes += bytes;
        return;
    }
    if (g_nrec >= (int)g_recs.size()) {       // pool exhausted: stop recording (the open run is closed at its true end)
        prof_close_open_run();
        return;
    }
    const int prev = g_open;
    const bool chain = prev >= 0 && g_recs[prev].stream == s;
    prof_close_open_run();
    rec = g_nrec++;
    ProfRec& r = g_recs[rec];
    r.family = family;
    r.launches = 1.0;
    r.flops = flops;
    r.bytes = bytes;
    r.stream = s;
    if (chain) {
        r.start_rec = prev;                  // the previous run's end event doubles as this run's start
    } else {
        r.start_rec = -1;
        (void)hipEventRecord(r.a, s);
    }
    g_open = rec;
}
VcxProfScope::~VcxProfScope() {}
