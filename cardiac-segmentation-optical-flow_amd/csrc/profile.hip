#include "profile.h"

#include <mutex>
#include <vector>

namespace cf {

struct Rec { int kid; double work; hipEvent_t e0, e1; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;
static size_t g_next = 0;

bool profile_on() { return g_on; }

bool profile_events(int kid, double work, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_next + 2 > g_pool.size()) return false;
    *start = g_pool[g_next++];
    *stop = g_pool[g_next++];
    g_recs.push_back({kid, work, *start, *stop});
    return true;
}

}  // namespace cf

using namespace cf;

// Enable with a pool of `max_launches` event pairs (created here, outside any timed region); 0 disables and frees.
extern "C" int cf_profile_enable(int max_launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (hipEvent_t e : g_pool) hipEventDestroy(e);
    g_pool.clear();
    g_recs.clear();
    g_next = 0;
    g_on = max_launches > 0;
    if (g_on) {
        g_pool.resize((size_t)max_launches * 2);
        for (auto& e : g_pool)
            if (hipEventCreate(&e) != hipSuccess) { set_error("cf_profile_enable: hipEventCreate failed"); g_on = false; return CF_ERR_LAUNCH; }
        g_recs.reserve(max_launches);
    }
    return CF_OK;
}

// Forget the launches recorded so far (events are reused).
extern "C" int cf_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.clear();
    g_next = 0;
    return CF_OK;
}

// Sum of kernel durations (ms), sum of algorithmic work and launch count for kernel id `kid` since the last reset.
// Synchronises on the recorded events (call it OUTSIDE the timed region).
extern "C" int cf_profile_read(int kid, double* total_ms, double* total_work, long* launches) {
    CF_REQUIRE(kid >= 0 && kid < PK_COUNT && total_ms && total_work && launches, "bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0, work = 0;
    long n = 0;
    for (const Rec& r : g_recs) {
        if (r.kid != kid) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) { set_error("cf_profile_read: event sync failed"); return CF_ERR_LAUNCH; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { set_error("cf_profile_read: elapsed failed"); return CF_ERR_LAUNCH; }
        ms += t;
        work += r.work;
        ++n;
    }
    *total_ms = ms;
    *total_work = work;
    *launches = n;
    return CF_OK;
}
