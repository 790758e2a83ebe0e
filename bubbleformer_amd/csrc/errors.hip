// Error reporting for the C ABI: entry points return a negative code, the text is kept per thread.
#include "bf_common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int bf_fail(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s:%d: HIP error %d (%s)", file, line, (int)e, hipGetErrorString(e));
    return -(int)e - 1000;
}
int bf_fail_msg(const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, msg);
    return -1;
}
// a refusal (return code 1: shape / workspace not covered, nothing launched) also leaves its reason behind, so that a caller who treats
// it as an error has a text to show
int bf_decline(const char* msg) {
    snprintf(g_err, sizeof(g_err), "declined (nothing launched): %s", msg);
    return 1;
}
extern "C" const char* bf_last_error(void) { return g_err; }
extern "C" int bf_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------ launch profiler
#include <map>
#include <string>
#include <vector>
namespace {
struct Rec { hipEvent_t a, b; std::string name; double flops, bytes; };
bool g_prof_on = false;
std::vector<Rec> g_recs;
}
bool bf_prof_is_on() { return g_prof_on; }
int bf_prof_begin(hipStream_t st) {
    if (!g_prof_on) return -1;
    Rec r{};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void bf_prof_end(int idx, hipStream_t st, const char* name, double flops, double bytes) {
    Rec& r = g_recs[idx];
    (void)hipEventRecord(r.b, st);
    r.name = name; r.flops = flops; r.bytes = bytes;
}
extern "C" void bf_prof_enable(int on) {
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_prof_on = on != 0;
}
// JSON object {name: {"calls": n, "ms": total, "flops": total, "bytes": total}}; returns bytes written or -1
extern "C" int bf_prof_report(char* buf, int n) {
    struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        if (r.name.empty()) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -1;
        Agg& a = agg[r.name];
        a.calls++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"calls\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += tmp;
        first = false;
    }
    out += "}";
    if ((int)out.size() + 1 > n) return -1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}
