// The exchange step of the sharded demix (replaces the hard torch.cat of
// /root/reference/xumx_slicq_v2/separator.py:229-231 for stems computed on OTHER ranks): every rank keeps the SAME flat
// per-track layout, the kernels write the rows a rank owns in place, and one grouped ncclSend / ncclRecv per pass moves
// every row owner -> peers AT IDENTICAL OFFSETS -- RCCL over the point-to-point xGMI links, one link per peer, no packing
// buffer and no placement pass (the all-gather form needed both: 18 ms and 38 GB of HBM traffic per step and rank that did
// not shrink with the number of ranks).
//
// RCCL is resolved at run time from the library the process already uses (torch's librccl.so: xsq_comm_load(path)), the
// same way _lib.py shares torch's HIP runtime; nothing here links against it.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "common.h"
#include "xumx_slicq_hip.h"

using namespace xsq;

namespace {

typedef void* nccl_comm_t;
struct nccl_uid { char internal[128]; };
enum { NCCL_FLOAT32 = 7 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*CommAbort)(nccl_comm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return XSQ_OK;
    void* h = nullptr;
    std::string tried;
    const char* cands[] = {path, getenv("XSQ_RCCL_LIB"), "librccl.so", "librccl.so.1"};
    for (const char* c : cands) {
        if (!c || !*c) continue;
        // the copy the process already mapped wins (one RCCL per process); else load it -- RTLD_LOCAL either way: every
        // symbol is reached through dlsym on this handle, nothing of RCCL's is exported process-wide on our account
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (!h) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : ", ") + c;
    }
    if (!h) {
        set_error("xsq_comm_load: librccl not found (tried: %s): %s", tried.c_str(), dlerror());
        return XSQ_ERR_ARG;
    }
    Rccl r;
    r.handle = h;
#define XSQ_SYM(field, name)                                                         \
    *(void**)(&r.field) = dlsym(h, name);                                            \
    if (!r.field) { set_error("xsq_comm_load: symbol %s missing in librccl", name); return XSQ_ERR_ARG; }
    XSQ_SYM(GetUniqueId, "ncclGetUniqueId")
    XSQ_SYM(CommInitRank, "ncclCommInitRank")
    XSQ_SYM(CommDestroy, "ncclCommDestroy")
    XSQ_SYM(CommAbort, "ncclCommAbort")
    XSQ_SYM(Send, "ncclSend")
    XSQ_SYM(Recv, "ncclRecv")
    XSQ_SYM(GroupStart, "ncclGroupStart")
    XSQ_SYM(GroupEnd, "ncclGroupEnd")
    XSQ_SYM(GetErrorString, "ncclGetErrorString")
    XSQ_SYM(GetVersion, "ncclGetVersion")
#undef XSQ_SYM
    g_rccl = r;
    return XSQ_OK;
}

#define XSQ_NCCL(expr)                                                                      \
    do {                                                                                    \
        int e_ = (expr);                                                                    \
        if (e_ != 0) {                                                                      \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(e_)); \
            return XSQ_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

}  // namespace

struct xsq_comm {
    nccl_comm_t comm = nullptr;
    int world = 0, rank = 0;
};

extern "C" {

int xsq_comm_load(const char* librccl_path) { return load_rccl(librccl_path); }

int xsq_comm_version(void) {
    if (load_rccl(nullptr)) return -1;
    int v = 0;
    return g_rccl.GetVersion(&v) == 0 ? v : -1;
}

int xsq_comm_unique_id(void* id128) {
    XSQ_REQUIRE(id128, "xsq_comm_unique_id: null argument");
    if (int rc = load_rccl(nullptr)) return rc;
    nccl_uid id;
    XSQ_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return XSQ_OK;
}

int xsq_comm_create(xsq_comm** out, const void* id128, int world, int rank) {
    XSQ_REQUIRE(out && id128 && world > 0 && rank >= 0 && rank < world, "xsq_comm_create: bad argument (world=%d rank=%d)", world, rank);
    if (int rc = load_rccl(nullptr)) return rc;
    nccl_uid id;
    memcpy(id.internal, id128, sizeof(id.internal));
    xsq_comm* c = new xsq_comm();
    c->world = world; c->rank = rank;
    int e = g_rccl.CommInitRank(&c->comm, world, id, rank);         // on the CURRENT device, collective over the ranks
    if (e != 0) {
        set_error("xsq_comm_create: ncclCommInitRank(world=%d, rank=%d) -> %s", world, rank, g_rccl.GetErrorString(e));
        delete c;
        return XSQ_ERR_HIP;
    }
    *out = c;
    return XSQ_OK;
}

int xsq_comm_destroy(xsq_comm* c) {
    if (!c) return XSQ_OK;
    if (c->comm) g_rccl.CommDestroy(c->comm);
    delete c;
    return XSQ_OK;
}

// Tears the communicator down WITHOUT waiting for queued operations (ncclCommAbort): what a rank calls when the ranks have
// voted that an exchange did not get queued everywhere -- receives whose sender never arrives would otherwise hold the
// stream for ever.  The handle is freed.
int xsq_comm_abort(xsq_comm* c) {
    if (!c) return XSQ_OK;
    if (c->comm) g_rccl.CommAbort(c->comm);
    delete c;
    return XSQ_OK;
}

// Rows per ncclGroupStart / ncclGroupEnd.  The cut is by ROW INDEX, never by operation count: the table is identical on every
// rank, so every rank closes its groups behind the same rows and the sends of one rank's group k meet the receives of its
// peers' group k (an owner issues world - 1 sends per row, everybody else one receive: a cut by operations would fall at
// different rows on different ranks).  Default: ~1024 point-to-point operations per group on the busiest rank.
static int rows_per_group(int world) {
    const char* e = getenv("XSQ_EXCHANGE_GROUP_ROWS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : std::max(1, 1024 / std::max(1, world - 1));
}

int xsq_exchange_rows(xsq_comm* c, const float* src, int64_t src_len, float* dst, int64_t dst_len, const int64_t* rows, int nrows,
                      int self_loop, void* stream_) {
    XSQ_REQUIRE(c && c->comm && src && dst && (rows || nrows == 0) && nrows >= 0 && src_len >= 0 && dst_len >= 0,
                "xsq_exchange_rows: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    // the whole table is checked before the first operation is queued: a bad row must not leave half a group behind
    for (int i = 0; i < nrows; ++i) {
        const int64_t owner = rows[4 * i], so = rows[4 * i + 1], dof = rows[4 * i + 2], len = rows[4 * i + 3];
        XSQ_REQUIRE(owner >= 0 && owner < c->world && len >= 0, "xsq_exchange_rows: row %d: owner %lld of %d ranks, length %lld", i,
                    (long long)owner, c->world, (long long)len);
        XSQ_REQUIRE(so >= 0 && so <= src_len - len, "xsq_exchange_rows: row %d: source span [%lld, +%lld) leaves the buffer of %lld floats",
                    i, (long long)so, (long long)len, (long long)src_len);
        XSQ_REQUIRE(dof >= 0 && dof <= dst_len - len, "xsq_exchange_rows: row %d: destination span [%lld, +%lld) leaves the buffer of %lld floats",
                    i, (long long)dof, (long long)len, (long long)dst_len);
    }
    const int per_group = rows_per_group(c->world);
    for (int g0 = 0; g0 < nrows; g0 += per_group) {
        const int g1 = std::min(nrows, g0 + per_group);
        XSQ_NCCL(g_rccl.GroupStart());
        int rc = XSQ_OK;
        for (int i = g0; i < g1 && rc == XSQ_OK; ++i) {
            const int owner = (int)rows[4 * i];
            const int64_t so = rows[4 * i + 1], dof = rows[4 * i + 2], len = rows[4 * i + 3];
            if (len == 0) continue;
            if (owner == c->rank) {
                for (int peer = 0; peer < c->world && rc == XSQ_OK; ++peer) {
                    if (peer == c->rank && !self_loop) continue;
                    if (g_rccl.Send(src + so, (size_t)len, NCCL_FLOAT32, peer, c->comm, stream) != 0) rc = XSQ_ERR_HIP;
                }
                if (self_loop && rc == XSQ_OK && g_rccl.Recv(dst + dof, (size_t)len, NCCL_FLOAT32, c->rank, c->comm, stream) != 0) rc = XSQ_ERR_HIP;
            } else if (g_rccl.Recv(dst + dof, (size_t)len, NCCL_FLOAT32, owner, c->comm, stream) != 0) {
                rc = XSQ_ERR_HIP;
            }
        }
        const int e = g_rccl.GroupEnd();
        if (rc != XSQ_OK || e != 0) {
            set_error("xsq_exchange_rows: ncclSend / ncclRecv / ncclGroupEnd failed in the group of rows [%d, %d): %s", g0, g1,
                      e ? g_rccl.GetErrorString(e) : "enqueue error");
            return XSQ_ERR_HIP;
        }
    }
    return XSQ_OK;
}

}  // extern "C"
