// Placement of packed stems into their tracks: the "final waveform concat" of the sharded path, gfx950.
//
// Reference: xumx_slicq_v2/separator.py:229-231 -- every chunk's (4, nb_samples, 2, n) estimate is appended to a
// list and the list is joined by a hard torch.cat along the sample axis.  On one GPU the inverse transform writes
// its rows straight into the result (xsq_slicqt_inverse_rows).  With the chunk items sharded over the ranks the
// stems of the OTHER ranks arrive packed in an all-gather buffer; one launch of k_place_rows moves every row
// (item, target, sample, channel) of a round to its span of the per-track tensors: an HBM copy driven by a row
// table, 16-byte accesses wherever source and destination share their alignment.
#include "../../include/xumx_slicq_hip.h"
#include "common.h"
#include "prof.h"

namespace xsq {

constexpr int PLACE_THREADS = 256;
constexpr int PLACE_VEC_PER_THREAD = 8;                                   // float4 per thread and grid-x step
constexpr int64_t PLACE_SPAN = (int64_t)PLACE_THREADS * PLACE_VEC_PER_THREAD * 4;   // floats per workgroup

// table: nrows x (src offset, dst offset, length), all in floats.
__global__ __launch_bounds__(PLACE_THREADS) void k_place_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                               const int64_t* __restrict__ table) {
    const int64_t so = table[3 * blockIdx.y], dof = table[3 * blockIdx.y + 1], len = table[3 * blockIdx.y + 2];
    const int64_t first = (int64_t)blockIdx.x * PLACE_SPAN;
    if (first >= len) return;
    const float* s = src + so;
    float* d = dst + dof;
    // floats in front of the first 16-byte boundary of the DESTINATION row (same for the source when the two
    // offsets agree modulo 4, the common case: chunk and track lengths are multiples of 4 except odd-length tracks)
    const int head = (int)((4 - (dof & 3)) & 3);
    const bool same = ((so ^ dof) & 3) == 0;
    if (same) {
        const int64_t nvec = len > head ? (len - head) >> 2 : 0;
        if (blockIdx.x == 0) {                                // the row's ragged ends: < 4 floats on either side
            const int64_t t = threadIdx.x, tail = head + 4 * nvec + t;
            if (t < head && t < len) d[t] = s[t];
            if (t < 3 && tail >= head && tail < len) d[tail] = s[tail];
        }
        const float4* s4 = reinterpret_cast<const float4*>(s + head);
        float4* d4 = reinterpret_cast<float4*>(d + head);
        const int64_t v0 = (int64_t)blockIdx.x * (PLACE_THREADS * PLACE_VEC_PER_THREAD) + threadIdx.x;
        float4 r[PLACE_VEC_PER_THREAD];
#pragma unroll
        for (int i = 0; i < PLACE_VEC_PER_THREAD; ++i) {
            const int64_t v = v0 + (int64_t)i * PLACE_THREADS;
            if (v < nvec) r[i] = s4[v];
        }
#pragma unroll
        for (int i = 0; i < PLACE_VEC_PER_THREAD; ++i) {
            const int64_t v = v0 + (int64_t)i * PLACE_THREADS;
            if (v < nvec) d4[v] = r[i];
        }
    } else {                                                  // rows of different alignment: 4-byte path
        const int64_t end = first + PLACE_SPAN < len ? first + PLACE_SPAN : len;
        for (int64_t i = first + threadIdx.x; i < end; i += PLACE_THREADS) d[i] = s[i];
    }
}

}  // namespace xsq

using namespace xsq;

extern "C" int xsq_place_rows(const float* src, float* dst, const int64_t* table, int nrows, int64_t max_len,
                              void* stream) {
    XSQ_REQUIRE(src && dst && table, "xsq_place_rows: null pointer");
    XSQ_REQUIRE(nrows >= 0 && nrows <= 65535 && max_len >= 0, "xsq_place_rows: nrows %d (<= 65535), max_len %lld", nrows,
                (long long)max_len);
    if (nrows == 0 || max_len == 0) return XSQ_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t gx = (max_len + PLACE_SPAN - 1) / PLACE_SPAN;
    XSQ_REQUIRE(gx < (1ll << 31), "xsq_place_rows: max_len too large");
    XSQ_PROF("place_rows", st);
    hipLaunchKernelGGL(k_place_rows, dim3((unsigned)gx, (unsigned)nrows), dim3(PLACE_THREADS), 0, st, src, dst, table);
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}
