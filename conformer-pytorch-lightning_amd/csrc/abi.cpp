// abi.cpp -- version, error text, device probe and the per-kernel profiling table of libconformer_gfx950.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cfm_common.h"

static thread_local char g_err[512] = "";

int cfm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int cfm_version(void) { return CFM_VERSION; }
extern "C" const char* cfm_last_error(void) { return g_err; }

extern "C" int cfm_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        cfm_fail(CFM_ERR_UNSUPPORTED, "no HIP device visible");
        return 0;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        cfm_fail(CFM_ERR_UNSUPPORTED, "cannot query HIP device");
        return 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        cfm_fail(CFM_ERR_UNSUPPORTED, "device is %s, this library is built for gfx950 only", prop.gcnArchName);
        return 0;
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------
// profiling table
// ---------------------------------------------------------------------------------------------
namespace {
struct Rec {
    const char* name;
    hipEvent_t start, stop;
    double flops, bytes;
};
struct Agg {
    int64_t calls = 0;
    double ms = 0, flops = 0, bytes = 0;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec*> g_recs;
std::map<std::string, Agg> g_agg;
std::vector<std::pair<std::string, Agg>> g_sorted;
}  // namespace

CfmProfScope::CfmProfScope(const char* name, hipStream_t s, double flops, double bytes) : rec(nullptr), stream(s), ev_start(nullptr), ev_stop(nullptr) {
    if (!g_on) return;
    Rec* r = new Rec{name, nullptr, nullptr, flops, bytes};
    if (hipEventCreate(&r->start) != hipSuccess || hipEventCreate(&r->stop) != hipSuccess) {
        delete r;
        return;
    }
    ev_start = r->start;                                   // stamped by the dispatch (CFM_LAUNCH), not recorded here
    ev_stop = r->stop;
    rec = r;
}

CfmProfScope::~CfmProfScope() {
    if (!rec) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back((Rec*)rec);
}

extern "C" void cfm_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
}

extern "C" void cfm_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Rec* r : g_recs) {
        (void)hipEventDestroy(r->start);
        (void)hipEventDestroy(r->stop);
        delete r;
    }
    g_recs.clear();
    g_agg.clear();
    g_sorted.clear();
}

extern "C" int cfm_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Rec* r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r->stop) == hipSuccess && hipEventElapsedTime(&ms, r->start, r->stop) == hipSuccess) {
            Agg& a = g_agg[r->name];
            a.calls += 1;
            a.ms += ms;
            a.flops += r->flops;
            a.bytes += r->bytes;
        }
        (void)hipEventDestroy(r->start);
        (void)hipEventDestroy(r->stop);
        delete r;
    }
    g_recs.clear();
    g_sorted.assign(g_agg.begin(), g_agg.end());
    return (int)g_sorted.size();
}

extern "C" int cfm_prof_entry(int i, char* name, int name_cap, int64_t* calls, double* total_ms, double* flops,
                              double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || i >= (int)g_sorted.size()) return cfm_fail(CFM_ERR_ARG, "prof entry %d out of range", i);
    if (name && name_cap > 0) {
        strncpy(name, g_sorted[i].first.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (calls) *calls = g_sorted[i].second.calls;
    if (total_ms) *total_ms = g_sorted[i].second.ms;
    if (flops) *flops = g_sorted[i].second.flops;
    if (bytes) *bytes = g_sorted[i].second.bytes;
    return CFM_OK;
}
