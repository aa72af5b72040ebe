// Version, thread-local error string and the GEMM profiling brackets of libvitk.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void vitk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vitk_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vitk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return VITK_ELAUNCH;
    }
    return VITK_OK;
}

// ---- profiling: one (start, stop) event pair per bracketed launch, kept until vitk_prof_read()
struct ProfRec { hipEvent_t a, b; double flop; int kind; };
static std::mutex g_prof_mu;
static int g_prof_on = 0;
static std::vector<ProfRec> g_prof;

bool vitk_prof_begin(void* stream, double flop, int kind) {
    if (!g_prof_on) return false;
    ProfRec r;
    r.flop = flop;
    r.kind = kind;
    if (hipEventCreate(&r.a) != hipSuccess) return false;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return false; }
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    return true;
}
void vitk_prof_end(void* stream) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, (hipStream_t)stream);
}

extern "C" {

int vitk_version(void) { return VITK_VERSION; }
const char* vitk_last_error(void) { return g_err; }

int vitk_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on ? 1 : 0;
    return VITK_OK;
}

int vitk_prof_read(int kind, double* total_ms, double* total_flop, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0;
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (kind != VITK_PROF_ALL && r.kind != kind) continue;
        float t = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms += t; fl += r.flop; ++n;
        }
    }
    if (total_ms) *total_ms = ms;
    if (total_flop) *total_flop = fl;
    if (launches) *launches = n;
    return VITK_OK;
}

int vitk_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return VITK_OK;
}

}  // extern "C"
