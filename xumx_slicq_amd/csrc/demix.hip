// The chunk loop of Separator.forward (/root/reference/xumx_slicq_v2/separator.py:133-232) issued from native code:
// one C call enqueues every launch of a track -- stacked full chunks on the caller's stream, the short tail chunk on a
// side stream beside them -- through row-offset tables cached per call shape.  No kernels of its own: it sequences
// the entry points of slicqt.hip / cdae.hip / wiener.hip (include/xumx_slicq_hip.h, "the whole call").
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cdae_api.h"
#include "common.h"
#include "plan.h"
#include "xumx_slicq_hip.h"

using namespace xsq;

namespace {

// what the 32-bit float offsets of the band kernels allow for the 8-channel estimate arena of one pass:
// 2 * (8 B) * S * sumFT < 2^31 (slicqt.hip: inverse_impl) -> B * S <= 7168 for the Bark-262 plan (sumFT = 18640)
static int default_max_item_slices(const xsq_plan* P) {
    const int64_t lim = ((1ll << 31) - 1) / (16 * std::max<int64_t>(P->sumFT, P->nbins));
    return (int)std::min<int64_t>(lim, 65535 / 8);          // and BC * S <= 65535 rows per launch
}

struct PassPlan {
    int64_t n = 0, n_pad = 0;        // samples per item that exist / that the slice count is taken from
    int B = 0, group = 1;            // stacked items (chunk, sample); samples per chunk in this pass
    int tail = 0;                    // 1: runs on the tail stream
    // a batch split over several passes under Wiener-EM: the passes of one chunk range (a "set") share a table of window
    // maxima taken over ALL their samples (norbert/__init__.py:257 spans the batch) -- float offset into the table buffer at
    // the head of the workspace, -1 = none; set_first / set_count: the passes of the set (consecutive)
    int64_t ext_off = -1;
    int set_first = 0, set_count = 1;
    int64_t* d_xrows = nullptr;      // [2B] input row offsets
    int64_t* d_orows = nullptr;      // [8B] output row offsets
};

struct ForwardPlan {
    std::vector<PassPlan> passes;
    int64_t* d_tables = nullptr;     // one allocation behind all d_xrows / d_orows
    int64_t* h_tables = nullptr;     // pinned host copy: uploaded when the plan is made, on the demixer's own copy stream (no device-wide stall)
    size_t table_words = 0;
    bool pinned = false;             // a HIP graph captured a call of this shape: its nodes hold d_xrows / d_orows, never evicted
    uint64_t last_use = 0;           // LRU stamp (xsq_demixer::clock)
    size_t main_bytes = 0, tail_bytes = 0;
    size_t ext_floats = 0;           // window-maximum tables of the split sets (zeroed at the start of every call)
};

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

struct xsq_demixer {
    xsq_plan* plan = nullptr;
    int max_item_slices = 0;
    std::vector<int32_t> F, T;
    std::mutex mu;
    // schedules + device row tables per call shape, least-recently-used bound: a test set of distinct track lengths would
    // otherwise leave one device allocation per length behind
    std::map<std::vector<int64_t>, ForwardPlan> plans;
    uint64_t clock = 0;
    static constexpr size_t kMaxPlans = 64;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t up_stream = nullptr;     // row-table uploads: a non-blocking stream of its own, waited for on the host per NEW shape
};

namespace {

struct PassLayout {
    size_t X, masks, Y, fwd, cdae, inv, wien, total;
    size_t fwd_bytes, cdae_bytes, inv_bytes, wien_bytes;
};

// workspace of one pass: X arena | masks arena | [Y arena] | forward ws | CDAE ws | inverse ws | [Wiener ws]
static int pass_layout(xsq_demixer* d, const xsq_model* Mo, int B, int64_t n_pad, int wiener, PassLayout* L) {
    xsq_plan* P = d->plan;
    const int S = xsq_plan_num_slices(P, n_pad);
    XSQ_REQUIRE(S >= 3, "xsq_demix_pass: %lld samples give %d slices; the conv stack needs 3 (pad to sllen/2 + 1)", (long long)n_pad, S);
    XSQ_REQUIRE(Mo->sumFT == P->sumFT && Mo->nblocks == P->nblocks, "xsq_demix_pass: the model's block table is not the plan's");
    const size_t coefs = (size_t)S * P->sumFT;
    size_t o = 0;
    L->X = o;     o += al256((size_t)2 * B * coefs * 8);
    L->masks = o; o += al256((size_t)8 * B * coefs * 4);
    L->Y = o;     if (wiener) o += al256((size_t)8 * B * coefs * 8);
    L->fwd_bytes = xsq_slicqt_forward_workspace(P, 2 * B, n_pad);
    L->cdae_bytes = xsq_cdae_workspace(Mo, B, S);
    L->inv_bytes = xsq_slicqt_inverse_workspace(P, 8 * B, S);
    L->wien_bytes = wiener ? xsq_wiener_workspace(P->nblocks, d->F.data(), d->T.data(), B, S, 5000) : 0;
    if (!L->fwd_bytes || !L->cdae_bytes || !L->inv_bytes || (wiener && !L->wien_bytes)) {
        const std::string why = xsq_last_error();
        set_error("xsq_demix_pass: B=%d, %lld samples: a stage's workspace query failed (%s)", B, (long long)n_pad, why.c_str());
        return XSQ_ERR_ARG;
    }
    L->fwd = o;  o += al256(L->fwd_bytes);
    L->cdae = o; o += al256(L->cdae_bytes);
    L->inv = o;  o += al256(L->inv_bytes);
    L->wien = o; o += al256(L->wien_bytes);
    L->total = o + 256;
    return XSQ_OK;
}

static int run_pass(xsq_demixer* d, xsq_model* Mo, const float* x, const float* const* x_slot, const int64_t* x_rows, int B, int64_t n, int64_t n_pad,
                    int group, int wiener, float* out, const int64_t* out_rows, void* ws, size_t ws_bytes, hipStream_t stream,
                    const float* ext_max = nullptr) {
    xsq_plan* P = d->plan;
    PassLayout L;
    int rc = pass_layout(d, Mo, B, n_pad, wiener, &L);
    if (rc) return rc;
    if (L.total > ws_bytes) {
        set_error("xsq_demix_pass: workspace too small (%zu needed, %zu given)", L.total, ws_bytes);
        return XSQ_ERR_WORKSPACE;
    }
    const int S = xsq_plan_num_slices(P, n_pad);
    char* w = (char*)ws;
    float* X = (float*)(w + L.X);
    float* masks = (float*)(w + L.masks);
    float* Y = (float*)(w + L.Y);
    const float *mean, *scale;
    int split;
    if ((rc = xsq_model_whitening(Mo, &mean, &scale, &split))) return rc;
    // the analysis kernels write the whitened magnitude into the head of the CDAE workspace (xsq_cdae_forward_xin, xin_ready)
    float* xin = (float*)(w + L.cdae);
    if ((rc = xsq_slicqt_forward_rows_indirect(P, x, x_slot, x_rows, 2 * B, n, n_pad, X, xin, mean, scale, split, w + L.fwd, L.fwd_bytes, stream)))
        return rc;
    if ((rc = xsq_cdae_forward_xin(Mo, X, B, S, nullptr, masks, w + L.cdae, L.cdae_bytes, stream, 1))) return rc;
    if (!wiener)
        return xsq_slicqt_inverse_masked(P, masks, X, 8 * B, 2 * B, S, n, out, out_rows, w + L.inv, L.inv_bytes, stream);
    if ((rc = xsq_wiener_em_masked_ext(P->nblocks, d->F.data(), d->T.data(), X, masks, Y, B, S, 5000, group, ext_max, w + L.wien,
                                       L.wien_bytes, stream)))
        return rc;
    return xsq_slicqt_inverse_rows(P, Y, 8 * B, S, n, out, out_rows, w + L.inv, L.inv_bytes, stream);
}

// One pass of a split batch, first half: its mix transform and the window maxima of its samples folded into the set's table.
static int run_prepass(xsq_demixer* d, xsq_model* Mo, const float* x, const float* const* x_slot, const PassPlan& p, void* ws, size_t ws_bytes, float* ext,
                       hipStream_t stream) {
    xsq_plan* P = d->plan;
    PassLayout L;
    int rc = pass_layout(d, Mo, p.B, p.n_pad, 1, &L);
    if (rc) return rc;
    XSQ_REQUIRE(L.total <= ws_bytes, "xsq_separator_forward: workspace too small");
    const int S = xsq_plan_num_slices(P, p.n_pad);
    char* w = (char*)ws;
    float* X = (float*)(w + L.X);
    if ((rc = xsq_slicqt_forward_rows_indirect(P, x, x_slot, p.d_xrows, 2 * p.B, p.n, p.n_pad, X, nullptr, nullptr, nullptr, 0, w + L.fwd, L.fwd_bytes,
                                               stream)))
        return rc;
    return xsq_wiener_window_max(P->nblocks, d->F.data(), d->T.data(), X, p.B, S, 5000, p.group, ext + p.ext_off, stream);
}

// ---- the schedule of one call shape, pure host arithmetic (no device call: tests/test_abi_cpu.py drives it through
// xsq_separator_schedule): stacked passes over the full chunks (at most max_stack (chunk, sample) pairs and `cap`
// item-slices per pass), then the remaining chunks one by one (the tail passes); a batch too large for one pass is split
// over sample ranges, and under Wiener-EM the passes of such a set share a table of window maxima.
struct SchedPass {
    int64_t start, n;        // first sample of the first chunk of the pass; samples per chunk that exist
    int k, b0, nbb;          // chunks stacked, first sample index, samples
    int tail;                // 1: may run on the tail stream
    int set_first, set_count;
    int needs_ext;           // 1: the set shares a window-maximum table (Wiener-EM, set_count > 1)
};

static inline int slices_of(int64_t n, int h) { return (int)(((n + h - 1) / h + 1) / 2 + 1); }      // nsgt/slicing.py:47-72

static void build_schedule(int L, int nb, int64_t N, int64_t cs, int max_stack, int wiener, int cap, std::vector<SchedPass>* out) {
    const int h = L / 4;
    const int64_t min_samples = L / 2 + 1;                         // separator.py:162
    auto close_set = [&](size_t first) {
        const int count = (int)(out->size() - first);
        for (size_t i = first; i < out->size(); ++i) {
            (*out)[i].set_first = (int)first; (*out)[i].set_count = count; (*out)[i].needs_ext = (wiener && count > 1) ? 1 : 0;
        }
    };
    const int64_t full = N / cs;
    const int S_full = slices_of(std::max(cs, min_samples), h);
    // samples per pass: the whole batch when one chunk of it fits a launch, else the largest share that does
    const int nbb_max = std::max(1, std::min(nb, cap / std::max(1, S_full)));
    const int per_pass = (int)std::max<int64_t>(1, std::min<int64_t>(max_stack / std::max(1, nbb_max), cap / ((int64_t)nbb_max * S_full)));
    int64_t start = 0;
    while (full - start / cs >= 2 && per_pass >= 2) {
        const int k = (int)std::min<int64_t>(per_pass, full - start / cs);
        const size_t first = out->size();
        for (int b0 = 0; b0 < nb; b0 += nbb_max) out->push_back(SchedPass{start, cs, k, b0, std::min(nbb_max, nb - b0), 0, 0, 1, 0});
        close_set(first);
        start += (int64_t)k * cs;
    }
    const bool stacked = !out->empty();
    for (; start < N; start += cs) {
        const int64_t n = std::min(cs, N - start);
        const int S_n = slices_of(std::max(n, min_samples), h);      // the sample split of a short chunk follows its own slice count
        const int nbb_n = std::max(1, std::min(nb, cap / std::max(1, S_n)));
        const size_t first = out->size();
        for (int b0 = 0; b0 < nb; b0 += nbb_n) out->push_back(SchedPass{start, n, 1, b0, std::min(nbb_n, nb - b0), stacked ? 1 : 0, 0, 1, 0});
        close_set(first);
    }
    bool any_ext = false;
    for (const SchedPass& p : *out) any_ext = any_ext || p.needs_ext;
    if (any_ext)                             // split sets under Wiener-EM run in order on the caller's stream
        for (SchedPass& p : *out) p.tail = 0;
}

// The schedule with its device tables and workspace sizes.
static int get_forward_plan(xsq_demixer* d, const xsq_model* Mo, int nb, int64_t N, int64_t cs, int max_stack, int wiener,
                            ForwardPlan** out) {
    xsq_plan* P = d->plan;
    const std::vector<int64_t> key{nb, N, cs, max_stack, wiener, d->max_item_slices, Mo->causal};
    auto it = d->plans.find(key);
    if (it != d->plans.end()) { it->second.last_use = ++d->clock; *out = &it->second; return XSQ_OK; }
    if (d->plans.size() >= xsq_demixer::kMaxPlans) {
        // evict the least recently used shape.  Its tables may still be read by kernels in flight (any stream): wait for the
        // device once -- this happens once per kMaxPlans NEW shapes, never in a steady loop over known shapes.  Shapes a HIP
        // graph has captured are never victims (a replay reads the tables without coming through here: ADVICE round 5);
        // when every cached shape is pinned the cache simply grows.
        auto victim = d->plans.end();
        for (auto p = d->plans.begin(); p != d->plans.end(); ++p)
            if (!p->second.pinned && (victim == d->plans.end() || p->second.last_use < victim->second.last_use)) victim = p;
        if (victim != d->plans.end()) {
            XSQ_HIP(hipDeviceSynchronize());
            (void)hipFree(victim->second.d_tables);
            (void)hipHostFree(victim->second.h_tables);
            d->plans.erase(victim);
        }
    }
    const int64_t min_samples = P->L / 2 + 1;                      // separator.py:162
    const int cap = d->max_item_slices > 0 ? d->max_item_slices : default_max_item_slices(P);
    ForwardPlan fp;
    std::vector<SchedPass> sched;
    build_schedule(P->L, nb, N, cs, max_stack, wiener, cap, &sched);
    std::vector<std::vector<int64_t>> xr, orw;
    for (const SchedPass& sp : sched) {
        PassPlan p;
        p.n = sp.n; p.n_pad = std::max(sp.n, min_samples); p.B = sp.k * sp.nbb; p.group = sp.nbb; p.tail = sp.tail;
        p.set_first = sp.set_first; p.set_count = sp.set_count;
        std::vector<int64_t> xrow((size_t)2 * p.B), orow((size_t)8 * p.B);
        for (int j2 = 0; j2 < sp.k; ++j2)
            for (int b = 0; b < sp.nbb; ++b)
                for (int c = 0; c < 2; ++c) {
                    const int item = j2 * sp.nbb + b;
                    xrow[(size_t)item * 2 + c] = ((int64_t)(sp.b0 + b) * 2 + c) * N + sp.start + j2 * cs;
                    for (int t = 0; t < 4; ++t)
                        orow[((size_t)t * p.B + item) * 2 + c] = (((int64_t)t * nb + sp.b0 + b) * 2 + c) * N + sp.start + j2 * cs;
                }
        xr.push_back(xrow); orw.push_back(orow);
        fp.passes.push_back(p);
    }
    // the window-maximum tables of the split sets (one per set, shared by its passes)
    for (size_t i = 0; i < sched.size(); ++i) {
        if (!sched[i].needs_ext) continue;
        if ((int)i == sched[i].set_first) {
            const int64_t off = (int64_t)fp.ext_floats;
            fp.ext_floats += (size_t)xsq_wiener_num_windows(P->nblocks, d->F.data(), d->T.data(), sched[i].k,
                                                            xsq_plan_num_slices(P, fp.passes[i].n_pad), 5000, 1);
            for (int q = 0; q < sched[i].set_count; ++q) fp.passes[i + q].ext_off = off;
        }
    }
    size_t total = 0;
    for (auto& p : fp.passes) total += (size_t)10 * p.B;
    XSQ_HIP(hipMalloc(&fp.d_tables, std::max<size_t>(total, 1) * sizeof(int64_t)));
    if (hipHostMalloc(&fp.h_tables, std::max<size_t>(total, 1) * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(fp.d_tables);
        set_error("xsq_separator_forward: hipHostMalloc of %zu table words failed", total);
        return XSQ_ERR_HIP;
    }
    fp.table_words = total;
    int64_t* host = fp.h_tables;
    size_t o = 0;
    for (size_t i = 0; i < fp.passes.size(); ++i) {
        PassPlan& p = fp.passes[i];
        p.d_xrows = fp.d_tables + o; std::copy(xr[i].begin(), xr[i].end(), host + o); o += xr[i].size();
        p.d_orows = fp.d_tables + o; std::copy(orw[i].begin(), orw[i].end(), host + o); o += orw[i].size();
        PassLayout L;
        int rc = pass_layout(d, Mo, p.B, p.n_pad, wiener, &L);
        if (rc) { (void)hipFree(fp.d_tables); (void)hipHostFree(fp.h_tables); return rc; }
        size_t& dst = p.tail ? fp.tail_bytes : fp.main_bytes;
        dst = std::max(dst, L.total);
    }
    fp.main_bytes += al256(fp.ext_floats * 4);
    fp.last_use = ++d->clock;
    // The tables are on the device before the plan is visible to any caller (any thread, any stream, eager or capturing):
    // copied on the demixer's own non-blocking stream and waited for on the host -- microseconds per NEW shape, no wait for
    // other streams' work.  A plan cannot be made while a stream of this thread is capturing in global mode (the wait is
    // refused): size the call first (xsq_separator_workspace), then capture.
    hipError_t ue = hipMemcpyAsync(fp.d_tables, fp.h_tables, fp.table_words * sizeof(int64_t), hipMemcpyHostToDevice, d->up_stream);
    if (ue == hipSuccess) ue = hipStreamSynchronize(d->up_stream);
    if (ue != hipSuccess) {
        (void)hipFree(fp.d_tables); (void)hipHostFree(fp.h_tables);
        set_error("xsq_separator_forward: upload of the row tables of a new call shape -> %s (a new shape cannot be planned during "
                  "stream capture: call xsq_separator_workspace for it first)", hipGetErrorString(ue));
        return XSQ_ERR_HIP;
    }
    auto ins = d->plans.emplace(key, fp);
    *out = &ins.first->second;
    return XSQ_OK;
}

}  // namespace

extern "C" {

int xsq_demixer_create(xsq_demixer** out, xsq_plan* plan) {
    XSQ_REQUIRE(out && plan, "xsq_demixer_create: null argument");
    xsq_demixer* d = new xsq_demixer();
    d->plan = plan;
    for (const BlockHost& b : plan->blocks) { d->F.push_back(b.F); d->T.push_back(b.T); }
    hipError_t e = hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->up_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("xsq_demixer_create: hipEventCreate -> %s", hipGetErrorString(e));
        delete d;
        return XSQ_ERR_HIP;
    }
    *out = d;
    return XSQ_OK;
}

int xsq_demixer_destroy(xsq_demixer* d) {
    if (!d) return XSQ_OK;
    for (auto& kv : d->plans) { (void)hipFree(kv.second.d_tables); (void)hipHostFree(kv.second.h_tables); }
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    if (d->ev_join) (void)hipEventDestroy(d->ev_join);
    if (d->up_stream) (void)hipStreamDestroy(d->up_stream);
    delete d;
    return XSQ_OK;
}

int xsq_separator_schedule(int L, int64_t coefs_per_slice, int nb, int64_t N, int64_t cs, int max_stack, int wiener,
                           int max_item_slices, int64_t* passes, int max_passes) {
    XSQ_REQUIRE(L > 0 && L % 4 == 0 && coefs_per_slice > 0 && nb > 0 && N > 0 && cs > 0 && max_stack > 0 && (passes || max_passes == 0),
                "xsq_separator_schedule: bad argument");
    const int64_t lim = ((1ll << 31) - 1) / (16 * std::max<int64_t>(coefs_per_slice, L / 2 + 1));
    const int dflt = (int)std::min<int64_t>(lim, 65535 / 8);
    const int cap = max_item_slices > 0 ? std::min(max_item_slices, dflt) : dflt;
    std::vector<SchedPass> sched;
    build_schedule(L, nb, N, cs, max_stack, wiener ? 1 : 0, cap, &sched);
    for (size_t i = 0; i < sched.size() && (int)i < max_passes; ++i) {
        const SchedPass& p = sched[i];
        int64_t* o = passes + 8 * i;
        o[0] = p.start; o[1] = p.n; o[2] = p.k; o[3] = p.b0; o[4] = p.nbb; o[5] = p.tail; o[6] = p.set_first; o[7] = p.needs_ext;
    }
    return (int)sched.size();
}

int xsq_demixer_set_max_rows(xsq_demixer* d, int max_item_slices) {
    XSQ_REQUIRE(d, "xsq_demixer_set_max_rows: null argument");
    std::lock_guard<std::mutex> lk(d->mu);
    d->max_item_slices = max_item_slices > 0 ? std::min(max_item_slices, default_max_item_slices(d->plan)) : 0;
    return XSQ_OK;
}

size_t xsq_demix_pass_workspace(xsq_demixer* d, const xsq_model* Mo, int B, int64_t n_pad, int wiener) {
    if (!d || !Mo || B <= 0 || n_pad <= 0) return 0;
    PassLayout L;
    return pass_layout(d, Mo, B, n_pad, wiener, &L) ? 0 : L.total;
}

int xsq_demix_pass(xsq_demixer* d, xsq_model* Mo, const float* x, const int64_t* x_rows, int B, int64_t n, int64_t n_pad,
                   int group, int wiener, float* out, const int64_t* out_rows, void* ws, size_t ws_bytes, void* stream) {
    XSQ_REQUIRE(d && Mo && x && out && out_rows && ws, "xsq_demix_pass: null argument");
    XSQ_REQUIRE(B > 0 && n > 0 && n_pad >= n, "xsq_demix_pass: B=%d n=%lld n_pad=%lld", B, (long long)n, (long long)n_pad);
    if (group <= 0) group = B;
    XSQ_REQUIRE(B % group == 0, "xsq_demix_pass: group=%d does not divide B=%d", group, B);
    return run_pass(d, Mo, x, nullptr, x_rows, B, n, n_pad, group, wiener, out, out_rows, ws, ws_bytes, (hipStream_t)stream);
}

int xsq_separator_workspace(xsq_demixer* d, const xsq_model* Mo, int nb, int64_t N, int64_t cs, int max_stack, int wiener,
                            size_t* main_bytes, size_t* tail_bytes) {
    XSQ_REQUIRE(d && Mo && main_bytes && tail_bytes, "xsq_separator_workspace: null argument");
    XSQ_REQUIRE(nb > 0 && N > 0 && cs > 0 && max_stack > 0, "xsq_separator_workspace: nb=%d N=%lld chunk_size=%lld max_stack=%d",
                nb, (long long)N, (long long)cs, max_stack);
    std::lock_guard<std::mutex> lk(d->mu);
    ForwardPlan* fp;
    int rc = get_forward_plan(d, Mo, nb, N, cs, max_stack, wiener ? 1 : 0, &fp);
    if (rc) return rc;
    *main_bytes = fp->main_bytes;
    *tail_bytes = fp->tail_bytes;
    return XSQ_OK;
}

static int separator_forward_impl(xsq_demixer* d, xsq_model* Mo, const float* audio, const float* const* x_slot, int nb, int64_t N, int64_t cs,
                                  int max_stack, int wiener, int overlap_tail, float* out, void* ws, size_t ws_bytes, void* tail_ws,
                                  size_t tail_ws_bytes, void* stream_, void* tail_stream_);

int xsq_separator_forward(xsq_demixer* d, xsq_model* Mo, const float* audio, int nb, int64_t N, int64_t cs, int max_stack,
                          int wiener, int overlap_tail, float* out, void* ws, size_t ws_bytes, void* tail_ws,
                          size_t tail_ws_bytes, void* stream_, void* tail_stream_) {
    XSQ_REQUIRE(audio, "xsq_separator_forward: null argument");
    return separator_forward_impl(d, Mo, audio, nullptr, nb, N, cs, max_stack, wiener, overlap_tail, out, ws, ws_bytes, tail_ws, tail_ws_bytes,
                                  stream_, tail_stream_);
}

int xsq_separator_forward_indirect(xsq_demixer* d, xsq_model* Mo, const float* const* audio_slot, int nb, int64_t N, int64_t cs,
                                   int max_stack, int wiener, int overlap_tail, float* out, void* ws, size_t ws_bytes, void* tail_ws,
                                   size_t tail_ws_bytes, void* stream_, void* tail_stream_) {
    XSQ_REQUIRE(audio_slot, "xsq_separator_forward_indirect: null argument");
    return separator_forward_impl(d, Mo, nullptr, audio_slot, nb, N, cs, max_stack, wiener, overlap_tail, out, ws, ws_bytes, tail_ws,
                                  tail_ws_bytes, stream_, tail_stream_);
}

static int separator_forward_impl(xsq_demixer* d, xsq_model* Mo, const float* audio, const float* const* x_slot, int nb, int64_t N, int64_t cs,
                                  int max_stack, int wiener, int overlap_tail, float* out, void* ws, size_t ws_bytes, void* tail_ws,
                                  size_t tail_ws_bytes, void* stream_, void* tail_stream_) {
    XSQ_REQUIRE(d && Mo && (audio || x_slot) && out && ws, "xsq_separator_forward: null argument");
    XSQ_REQUIRE(nb > 0 && N > 0 && cs > 0 && max_stack > 0, "xsq_separator_forward: nb=%d N=%lld chunk_size=%lld max_stack=%d",
                nb, (long long)N, (long long)cs, max_stack);
    hipStream_t main = (hipStream_t)stream_, side = (hipStream_t)tail_stream_;
    ForwardPlan* fp;
    {
        std::lock_guard<std::mutex> lk(d->mu);
        int rc = get_forward_plan(d, Mo, nb, N, cs, max_stack, wiener ? 1 : 0, &fp);
        if (rc) return rc;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) fp->pinned = true;
    }
    bool any_tail = false;
    for (const PassPlan& p : fp->passes) any_tail = any_tail || p.tail;
    const bool beside = any_tail && overlap_tail && side != main && tail_ws != nullptr;
    XSQ_REQUIRE(ws_bytes >= fp->main_bytes && (!beside || tail_ws_bytes >= fp->tail_bytes),
                "xsq_separator_forward: workspace too small (%zu / %zu needed, %zu / %zu given)", fp->main_bytes, fp->tail_bytes,
                ws_bytes, tail_ws_bytes);
    XSQ_REQUIRE(beside || ws_bytes >= std::max(fp->main_bytes, fp->tail_bytes), "xsq_separator_forward: workspace too small for the tail pass");
    int rc = XSQ_OK;
    bool forked = false;
    // The short tail chunks are launch-bound passes, independent of the stacked ones: they go out first, on the side
    // stream with their own workspace, and fill in beside the big launches; the caller's stream joins at the end -- on the
    // error paths too: whatever the side stream was given stays ordered in front of the caller's next use of `out`.
    auto join = [&]() -> int {
        if (!forked) return XSQ_OK;
        forked = false;
        XSQ_HIP(hipEventRecord(d->ev_join, side));
        XSQ_HIP(hipStreamWaitEvent(main, d->ev_join, 0));
        return XSQ_OK;
    };
    if (beside) {
        XSQ_HIP(hipEventRecord(d->ev_fork, main));
        XSQ_HIP(hipStreamWaitEvent(side, d->ev_fork, 0));
        forked = true;
        for (const PassPlan& p : fp->passes)
            if (p.tail && (rc = run_pass(d, Mo, audio, x_slot, p.d_xrows, p.B, p.n, p.n_pad, p.group, wiener, out, p.d_orows, tail_ws, tail_ws_bytes, side)))
                break;
        if (rc) { const std::string why = xsq_last_error(); (void)join(); set_error("%s", why.c_str()); return rc; }
    }
    float* ext = (float*)ws;                                  // window-maximum tables of split sets (head of the workspace)
    const size_t ext_bytes = al256(fp->ext_floats * 4);
    char* pws = (char*)ws + ext_bytes;
    const size_t pws_bytes = ws_bytes - ext_bytes;
    if (fp->ext_floats && hipMemsetAsync(ext, 0, fp->ext_floats * 4, main) != hipSuccess) {
        (void)join();
        set_error("xsq_separator_forward: hipMemsetAsync failed");
        return XSQ_ERR_HIP;
    }
    for (size_t i = 0; i < fp->passes.size() && rc == XSQ_OK; ++i) {
        const PassPlan& p = fp->passes[i];
        if (p.tail && beside) continue;
        if (p.ext_off >= 0 && (int)i == p.set_first)       // first pass of a split set: every pass's maxima first
            for (int j = 0; j < p.set_count && rc == XSQ_OK; ++j)
                rc = run_prepass(d, Mo, audio, x_slot, fp->passes[i + j], pws, pws_bytes, ext, main);
        if (rc == XSQ_OK)
            rc = run_pass(d, Mo, audio, x_slot, p.d_xrows, p.B, p.n, p.n_pad, p.group, wiener, out, p.d_orows, pws, pws_bytes, main,
                          p.ext_off >= 0 ? ext + p.ext_off : nullptr);
    }
    if (rc) { const std::string why = xsq_last_error(); (void)join(); set_error("%s", why.c_str()); return rc; }
    return join();
}

}  // extern "C"
