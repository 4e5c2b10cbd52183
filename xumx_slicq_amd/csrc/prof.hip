#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/xumx_slicq_hip.h"
#include "prof.h"

namespace xsq {

struct Pending {
    const char* name;
    hipEvent_t a, b;
};
static std::mutex g_mu;
static bool g_on = false;
static std::string g_only;          // when non-empty: only the kernel of this event name is timed (xsq_profile_filter)
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static std::map<std::string, std::pair<double, int64_t>> g_acc;   // name -> (ms, launches)

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_enabled() { return g_on; }
bool prof_wanted(const char* name) { return g_on && (g_only.empty() || g_only == name); }

void prof_begin(const char* name, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Pending p{name, get_event(), get_event()};
    (void)hipEventRecord(p.a, stream);
    g_pending.push_back(p);
}

void prof_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pending.back().b, stream);
}

static void collect_locked() {
    for (Pending& p : g_pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto& e = g_acc[p.name];
            e.first += ms;
            e.second += 1;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
}

}  // namespace xsq

using namespace xsq;

extern "C" {

int xsq_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return XSQ_OK;
}

int xsq_profile_filter(const char* name) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = name ? name : "";
    return XSQ_OK;
}

int xsq_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    collect_locked();
    g_acc.clear();
    return XSQ_OK;
}

int xsq_profile_read(char* names, size_t names_bytes, double* ms, int64_t* launches, int max_entries) {
    std::lock_guard<std::mutex> lk(g_mu);
    collect_locked();
    int n = 0;
    size_t off = 0;
    for (auto& kv : g_acc) {
        if (n >= max_entries || off + kv.first.size() + 2 > names_bytes) break;
        memcpy(names + off, kv.first.c_str(), kv.first.size());
        off += kv.first.size();
        names[off++] = '\n';
        ms[n] = kv.second.first;
        launches[n] = kv.second.second;
        ++n;
    }
    if (names_bytes) names[off < names_bytes ? off : names_bytes - 1] = 0;
    return n;
}

}  // extern "C"
