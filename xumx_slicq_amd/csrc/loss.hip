// Loss forward of the reference's training/validation loop on the coefficient arenas, gfx950.
//
// Reference: xumx_slicq_v2/loss.py:37-76 (ComplexMSELossCriterion: per block the mean squared error of
// the 4 + 6 + 4 one-, two- and three-target sums of estimates vs targets, /14, averaged over blocks) and
// loss.py:79-96 (MaskSumLossCriterion: per block mean((sum_j mask_j - 1)^2), averaged over blocks);
// training.py:86-103 adds the two.  With e_j = pred_j - target_j per real element, s1 = sum_j e_j and
// s2 = sum_j e_j^2, the 14 squared sums add up to 4*s2 + 3*s1^2 (pairs: 2*s2 + s1^2, triples: 2*s1^2 + s2).
// One streaming pass, fp64 partial sums, fixed-order reductions (bitwise reproducible, no atomics).
#include <map>
#include <mutex>
#include <vector>

#include "../../include/xumx_slicq_hip.h"
#include "plan.h"
#include "prof.h"

namespace xsq {

struct LossWork {     // one workgroup: `count` float4 quads of one block's per-target sub-arena
    int block;
    int pad;
    int64_t first;    // first quad (in floats / 4) inside the per-target sub-arena of the block
    int64_t count;
    int64_t base_c;   // float offset of the block in a complex arena holding 8B channels
    int64_t base_r;   // float offset of the block in the real (mask) arena
    int64_t tstride_c, tstride_r;   // floats between consecutive targets (complex / real arena)
    int64_t nreal;    // floats per target in the real arena (complex has 2x)
};

// BWD: the same pass also writes the loss gradients (training step, training.py:107): d/dpred_j = (8 e_j + 6 sum_k e_k)
// / (14 n_b nblocks) into gY and the mask-sum gradient 2 (sum_j m_j - 1) / (n'_b nblocks) into gM (the same value for
// the four targets) -- the operands are in registers already, a second read of pred / target / masks is saved.
template <bool BWD>
__global__ __launch_bounds__(256) void k_loss_partial(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       const float* __restrict__ masks,
                                                       const LossWork* __restrict__ work, double* __restrict__ partial,
                                                       float* __restrict__ gY, float* __restrict__ gM, int nblocks) {
    const LossWork w = work[blockIdx.x];
    double mse = 0.0, msk = 0.0;
    const float cb = BWD ? 1.f / (14.f * (float)(2 * w.nreal) * (float)nblocks) : 0.f;
    const float cm = BWD ? 1.f / ((float)w.nreal * (float)nblocks) : 0.f;
    for (int64_t q = threadIdx.x; q < w.count; q += 256) {
        const int64_t i = 4 * (w.first + q);                 // float index inside the per-target complex sub-arena
        float4 e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 p = *reinterpret_cast<const float4*>(pred + w.base_c + j * w.tstride_c + i);
            const float4 t = *reinterpret_cast<const float4*>(tgt + w.base_c + j * w.tstride_c + i);
            e[j] = make_float4(p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w);
        }
        const float s1x = e[0].x + e[1].x + e[2].x + e[3].x, s1y = e[0].y + e[1].y + e[2].y + e[3].y;
        const float s1z = e[0].z + e[1].z + e[2].z + e[3].z, s1w = e[0].w + e[1].w + e[2].w + e[3].w;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s2 += e[j].x * e[j].x + e[j].y * e[j].y + e[j].z * e[j].z + e[j].w * e[j].w;
        mse += (double)(4.f * s2 + 3.f * (s1x * s1x + s1y * s1y + s1z * s1z + s1w * s1w));
        if (BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(gY + w.base_c + j * w.tstride_c + i) =
                    make_float4((8.f * e[j].x + 6.f * s1x) * cb, (8.f * e[j].y + 6.f * s1y) * cb,
                                (8.f * e[j].z + 6.f * s1z) * cb, (8.f * e[j].w + 6.f * s1w) * cb);
        }
        if (masks && i < w.nreal) {       // the real arena is half as long: quads [0, nreal/4) of this block
            float4 m = *reinterpret_cast<const float4*>(masks + w.base_r + i);
            float4 sm = make_float4(-1.f + m.x, -1.f + m.y, -1.f + m.z, -1.f + m.w);      // the gradient's own summation order
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const float4 mj = *reinterpret_cast<const float4*>(masks + w.base_r + j * w.tstride_r + i);
                m.x += mj.x; m.y += mj.y; m.z += mj.z; m.w += mj.w;
                sm.x += mj.x; sm.y += mj.y; sm.z += mj.z; sm.w += mj.w;
            }
            m.x -= 1.f; m.y -= 1.f; m.z -= 1.f; m.w -= 1.f;
            msk += (double)(m.x * m.x + m.y * m.y + m.z * m.z + m.w * m.w);
            if (BWD) {
                const float4 gm = make_float4(2.f * sm.x * cm, 2.f * sm.y * cm, 2.f * sm.z * cm, 2.f * sm.w * cm);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(gM + w.base_r + j * w.tstride_r + i) = gm;
            }
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = mse;
    red[1][threadIdx.x] = msk;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}

// one thread per block: fixed-order sum of its workgroups' partials, scaled to the block's mean
__global__ void k_loss_combine(const double* __restrict__ partial, const int* __restrict__ first_wg,
                               const double* __restrict__ inv_n, double* __restrict__ out, int nblocks) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    double a = 0.0, m = 0.0;
    for (int g = first_wg[b]; g < first_wg[b + 1]; ++g) { a += partial[2 * g]; m += partial[2 * g + 1]; }
    out[2 * b] = a * inv_n[2 * b];          // / (14 * elements)
    out[2 * b + 1] = m * inv_n[2 * b + 1];  // / mask elements
}

}  // namespace xsq

using namespace xsq;

// work tables of one (device, block table, B, S): built once on first use -- a synchronous upload, so a shape has to be warmed up
// before it is captured in a graph -- and resident for the life of the process (a few KB per shape; the step must not wait
// for the host in mid-flight)
struct LossTables { LossWork* d_work = nullptr; double* d_inv = nullptr; int* d_first = nullptr; int nwork = 0; };
static std::mutex g_loss_mu;
static std::map<std::vector<int>, LossTables> g_loss_tables;

static int loss_tables(int nblocks, const int32_t* F, const int32_t* T, int Bn, int S, LossTables* out) {
    int dev = 0;
    XSQ_HIP(hipGetDevice(&dev));                 // the tables live in one device's memory: keyed by it
    std::vector<int> key{dev, nblocks, Bn, S};
    for (int b = 0; b < nblocks; ++b) { key.push_back(F[b]); key.push_back(T[b]); }
    std::lock_guard<std::mutex> lk(g_loss_mu);
    auto it = g_loss_tables.find(key);
    if (it != g_loss_tables.end()) { *out = it->second; return XSQ_OK; }
    std::vector<LossWork> work;
    std::vector<int> first(nblocks + 1, 0);
    std::vector<double> inv(2 * nblocks);
    int64_t cum = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int64_t per_t_real = (int64_t)Bn * 2 * F[b] * S * T[b];      // floats per target, real arena
        const int64_t quads = per_t_real * 2 / 4;                         // complex sub-arena in float4
        first[b] = (int)work.size();
        for (int64_t q0 = 0; q0 < quads; q0 += 4096) {
            LossWork w;
            w.block = b; w.pad = 0; w.first = q0; w.count = quads - q0 < 4096 ? quads - q0 : 4096;
            w.base_c = 2 * (int64_t)Bn * 8 * S * cum; w.base_r = (int64_t)Bn * 8 * S * cum;
            w.tstride_c = 2 * per_t_real; w.tstride_r = per_t_real; w.nreal = per_t_real;
            work.push_back(w);
        }
        inv[2 * b] = 1.0 / (14.0 * (double)(2 * per_t_real));
        inv[2 * b + 1] = 1.0 / (double)per_t_real;
        cum += (int64_t)F[b] * T[b];
    }
    first[nblocks] = (int)work.size();
    LossTables t;
    t.nwork = (int)work.size();
    XSQ_HIP(hipMalloc(&t.d_work, work.size() * sizeof(LossWork)));
    XSQ_HIP(hipMemcpy(t.d_work, work.data(), work.size() * sizeof(LossWork), hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&t.d_inv, inv.size() * sizeof(double)));
    XSQ_HIP(hipMemcpy(t.d_inv, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&t.d_first, first.size() * sizeof(int)));
    XSQ_HIP(hipMemcpy(t.d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
    g_loss_tables[key] = t;
    *out = t;
    return XSQ_OK;
}

namespace xsq {
// loss forward, optionally with the gradients of both terms in the same pass (gY / gM non-null: training step)
int loss_forward_backward(int nblocks, const int32_t* F, const int32_t* T, const float* pred, const float* target,
                          const float* masks, int Bn, int S, double* out, float* gY, float* gM, void* ws, hipStream_t stream) {
    LossTables t;
    if (int rc = loss_tables(nblocks, F, T, Bn, S, &t)) return rc;
    double* d_partial = (double*)ws;
    { XSQ_PROF("loss_partial", stream);
      if (gY && gM && masks)
          hipLaunchKernelGGL(k_loss_partial<true>, dim3((unsigned)t.nwork), dim3(256), 0, stream, pred, target, masks, t.d_work, d_partial, gY, gM, nblocks);
      else
          hipLaunchKernelGGL(k_loss_partial<false>, dim3((unsigned)t.nwork), dim3(256), 0, stream, pred, target, masks, t.d_work, d_partial, nullptr, nullptr, nblocks); }
    hipLaunchKernelGGL(k_loss_combine, dim3((nblocks + 63) / 64), dim3(64), 0, stream, d_partial, t.d_first, t.d_inv, out, nblocks);
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}
}  // namespace xsq

extern "C" {

size_t xsq_loss_workspace(int nblocks, const int32_t* F, const int32_t* T, int Bn, int S) {
    if (nblocks <= 0 || !F || !T || Bn <= 0 || S <= 0) return 0;
    size_t nwg = 0;
    for (int b = 0; b < nblocks; ++b) nwg += ((size_t)Bn * 2 * F[b] * S * T[b] * 2 / 4 + 4095) / 4096;
    return nwg * (sizeof(LossWork) + 16) + (size_t)nblocks * 64 + 1024;
}

int xsq_loss_forward(int nblocks, const int32_t* F, const int32_t* T, const float* pred, const float* target,
                     const float* masks, int Bn, int S, double* out, void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(nblocks > 0 && F && T && pred && target && out && ws, "xsq_loss_forward: null argument");
    XSQ_REQUIRE(Bn > 0 && S > 0, "xsq_loss_forward: B=%d S=%d", Bn, S);
    XSQ_REQUIRE(ws_bytes >= xsq_loss_workspace(nblocks, F, T, Bn, S), "xsq_loss_forward: workspace too small");
    return xsq::loss_forward_backward(nblocks, F, T, pred, target, masks, Bn, S, out, nullptr, nullptr, ws, (hipStream_t)stream_);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Dataset statistics (training.get_statistics, training.py:115-154): per block and frequency bin the
// count, sum and sum of squares over frames of the channel-mean magnitude  m = mean_c |X[c, f, frame]|.
// One workgroup per (block, bin) row, fp64 accumulation, fixed-order reduction.
// ------------------------------------------------------------------------------------------------
namespace xsq {
struct StatRow {
    int F, T, f, pad;
    int64_t cum;       // sum over earlier blocks of F*T
    int64_t out;       // index of the row in the output (sum over earlier blocks of F, plus f)
};

__global__ __launch_bounds__(256) void k_magnitude_stats(const float2* __restrict__ X, const StatRow* __restrict__ rows,
                                                          double* __restrict__ out, int C, int S) {
    const StatRow r = rows[blockIdx.x];
    const int64_t N = (int64_t)S * r.T;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += 256) {
        float m = 0.f;
        for (int c = 0; c < C; ++c) {
            const float2 z = X[(int64_t)C * S * r.cum + ((int64_t)c * r.F + r.f) * N + n];
            m += sqrtf(z.x * z.x + z.y * z.y);
        }
        m /= (float)C;
        s1 += (double)m;
        s2 += (double)m * (double)m;
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * r.out] = red[0][0]; out[2 * r.out + 1] = red[1][0]; }
}
}  // namespace xsq

extern "C" int xsq_magnitude_stats(int nblocks, const int32_t* F, const int32_t* T, const float* X, int C, int S,
                                   double* out, void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(nblocks > 0 && F && T && X && out && ws, "xsq_magnitude_stats: null argument");
    XSQ_REQUIRE(C > 0 && S > 0, "xsq_magnitude_stats: C=%d S=%d", C, S);
    std::vector<xsq::StatRow> rows;
    int64_t cum = 0, o = 0;
    for (int b = 0; b < nblocks; ++b) {
        for (int f = 0; f < F[b]; ++f) rows.push_back(xsq::StatRow{F[b], T[b], f, 0, cum, o++});
        cum += (int64_t)F[b] * T[b];
    }
    XSQ_REQUIRE(ws_bytes >= rows.size() * sizeof(xsq::StatRow), "xsq_magnitude_stats: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    XSQ_HIP(hipMemcpyAsync(ws, rows.data(), rows.size() * sizeof(xsq::StatRow), hipMemcpyHostToDevice, stream));
    XSQ_HIP(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(xsq::k_magnitude_stats, dim3((unsigned)rows.size()), dim3(256), 0, stream, (const float2*)X,
                       (const xsq::StatRow*)ws, out, C, S);
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}
