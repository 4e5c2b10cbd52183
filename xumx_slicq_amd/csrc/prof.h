// Optional per-kernel timing with HIP events recorded on the launch stream
// (bench.py reads it for the roofline object; off by default, zero cost when off).
#pragma once
#include "common.h"

namespace xsq {
bool prof_enabled();
bool prof_wanted(const char* name);
void prof_begin(const char* name, hipStream_t stream);
void prof_end(hipStream_t stream);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char* name, hipStream_t stream) : s(stream), on(prof_wanted(name)) {
        if (on) prof_begin(name, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};
}  // namespace xsq

#define XSQ_PROF(name, stream) ::xsq::ProfScope prof_scope_##__LINE__(name, stream)
