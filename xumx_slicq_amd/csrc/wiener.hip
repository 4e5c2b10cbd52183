// Mix-phase estimate and norbert Wiener-EM (one iteration) on the coefficient arena, gfx950.
//
// Reference: xumx_slicq_v2/phase.py:18-69 (blockwise_wiener: windows of <= 5000 frames over the
// flattened (slice, time) axis), :96-113 (blockwise_phasemix_sep); norbert/__init__.py:153-260
// (wiener, iterations=1, use_softmask=False) -> expectation_maximization :10-150 ->
// get_local_gaussian_model :458-494, get_mix_model :416-437, _invert :312-350,
// wiener_gain :353-388, apply_filter :391-413.
//
// Per (block, window w, batch item b, bin f) with frames n in the window, channels c,d in {0,1},
// sources j in {0..3}, y0 = initial estimate (mask * X), ma = max(1, 0.1 * max |x|) over the whole
// window INCLUDING the batch dimension (norbert :257), y' = y0/ma, x' = x/ma:
//   v'[n,j]  = mean_c |y'[n,c,j]|^2
//   R[j]     = sum_n y'[n,:,j] y'[n,:,j]^H / (sum_n v'[n,j] + eps)          eps = FLT_EPSILON
//   Cxx[n]   = sum_j v'[n,j] R[j] + sqrt(eps) I ;  y[n,:,j] = ma * v'[n,j] R[j] Cxx[n]^-1 x'[n]
// Three launches: k_wiener_stats (raw sums + max per row/window, fixed-order LDS tree, no atomics
// -> bitwise reproducible), k_wiener_finalize (window max, R), k_wiener_apply (elementwise 2x2
// solve, in place on Y).  All reads/writes are contiguous along the frame axis.
#include <cfloat>
#include <cmath>
#include <vector>

#include "../../include/xumx_slicq_hip.h"
#include "plan.h"
#include "prof.h"

namespace xsq {

static const int STAT = 24;   // floats per (row, window): 4 sources x (C00, C11, Re C01, Im C01), max|x|^2, pad[3],
                              // 4 x 1/(sum_n v + eps) (kept for the backward pass)

struct WRow {          // one (block, batch item, bin) row of the arena
    int F, T;          // block geometry
    int b, f;          // batch item, bin
    int nwin;          // windows of this block
    int first_row;     // first row index of this row's block (rows of a block are consecutive)
    int nrows;         // rows in the block (B*F)
    int pad;
    int64_t cum;       // sum over earlier blocks of F*T
    int64_t stat;      // float offset of this row's window 0 in the stats buffer
};

struct WTable {
    WRow* d_rows = nullptr;
    int* d_work = nullptr;     // (row, window) pairs for the stats / finalize launches
    int nrows = 0, nwork = 0, nblockwin = 0;
    int* d_blockwin = nullptr; // (first_row, window) per (block, window)
    int* d_bw_of_work = nullptr;   // per (row, window) work item: index of its (block, group, window) in d_blockwin order
    int64_t stat_floats = 0;
    int64_t max_frames = 0;
};

static std::mutex g_wmu;
static std::map<std::vector<int>, WTable> g_wtables;

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)

// complex arena index of (chan, f, frame n) for a block; nchan = packed channels of the arena
__device__ inline int64_t cidx(const WRow& r, int nchan, int S, int chan, int64_t n) {
    return (int64_t)nchan * S * r.cum + ((int64_t)chan * r.F + r.f) * ((int64_t)S * r.T) + n;
}

// ---- Y = mag * x/|x|  (phase.py:96-113; angle(0) = 0) --------------------------------------------
__global__ __launch_bounds__(256) void k_phasemix(const float2* __restrict__ X, const float* __restrict__ mag,
                                                   float2* __restrict__ Y, const WRow* __restrict__ rows, int Bn,
                                                   int S) {
    const WRow r = rows[blockIdx.y];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float2 x = X[cidx(r, 2 * Bn, S, r.b * 2 + c, n)];
        const float ax = sqrtf(x.x * x.x + x.y * x.y);
        const float2 u = ax > 0.f ? make_float2(x.x / ax, x.y / ax) : make_float2(1.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t yi = cidx(r, 8 * Bn, S, (j * Bn + r.b) * 2 + c, n);
            const float m = mag[yi];
            Y[yi] = make_float2(m * u.x, m * u.y);
        }
    }
}

// ---- pass 1: raw sums and max per (row, window).  One workgroup per (row, window). -----------------
__global__ __launch_bounds__(256) void k_wiener_stats(const float2* __restrict__ X, const float2* __restrict__ Y,
                                                       const WRow* __restrict__ rows, const int* __restrict__ work,
                                                       float* __restrict__ stats, int Bn, int S, int win_len) {
    const int row = work[2 * blockIdx.x], w = work[2 * blockIdx.x + 1];
    const WRow r = rows[row];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n0 = (int64_t)w * win_len;
    const int64_t n1 = n0 + win_len < N ? n0 + win_len : N;
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    const float2* x0 = X + cidx(r, 2 * Bn, S, r.b * 2, 0);
    const float2* x1 = X + cidx(r, 2 * Bn, S, r.b * 2 + 1, 0);
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
        const float2 a = x0[n], b = x1[n];
        acc[16] = fmaxf(acc[16], fmaxf(a.x * a.x + a.y * a.y, b.x * b.x + b.y * b.y));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 y0 = Y[cidx(r, 8 * Bn, S, (j * Bn + r.b) * 2, n)];
            const float2 y1 = Y[cidx(r, 8 * Bn, S, (j * Bn + r.b) * 2 + 1, n)];
            const float2 c01 = cmulc(y0, y1);
            acc[4 * j + 0] += y0.x * y0.x + y0.y * y0.y;
            acc[4 * j + 1] += y1.x * y1.x + y1.y * y1.y;
            acc[4 * j + 2] += c01.x;
            acc[4 * j + 3] += c01.y;
        }
    }
    // wavefront butterfly, then a fixed-order sum of the 4 wave partials
    __shared__ float red[4][17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v, off, 64);
            v = (i == 16) ? fmaxf(v, o) : v + o;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        const int i = threadIdx.x;
        float v;
        if (i == 16) v = fmaxf(fmaxf(red[0][i], red[1][i]), fmaxf(red[2][i], red[3][i]));
        else v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        stats[r.stat + (int64_t)w * STAT + i] = v;
    }
}

// ---- pass 2: window max over all rows of the block, then R per row.  One workgroup per (block, window).
// ext_max (optional): max |x|^2 per (block, group, window) in blockwin order, taken over MORE batch items than this call holds
// (a batch split into several passes, demix.hip): the window maximum of norbert :257 spans the whole batch.
__global__ __launch_bounds__(256) void k_wiener_finalize(const WRow* __restrict__ rows,
                                                          const int* __restrict__ blockwin,
                                                          float* __restrict__ stats, const float* __restrict__ ext_max = nullptr) {
    const int first = blockwin[2 * blockIdx.x], w = blockwin[2 * blockIdx.x + 1];
    const int nrows = rows[first].nrows;
    __shared__ float smax[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < nrows; i += 256) m = fmaxf(m, stats[rows[first + i].stat + (int64_t)w * STAT + 16]);
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
        __syncthreads();
    }
    const float mx2 = ext_max ? fmaxf(smax[0], ext_max[blockIdx.x]) : smax[0];
    const float ma = fmaxf(1.f, 0.1f * sqrtf(mx2));         // norbert :257
    const float inv_ma2 = 1.f / (ma * ma);
    const float eps = FLT_EPSILON;
    for (int i = threadIdx.x; i < nrows; i += 256) {
        float* st = stats + rows[first + i].stat + (int64_t)w * STAT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float c00 = st[4 * j] * inv_ma2, c11 = st[4 * j + 1] * inv_ma2;
            const float den = 1.f / (0.5f * (c00 + c11) + eps);   // sum_n mean_c |y'|^2 + eps   (:491-493)
            st[4 * j] = c00 * den;
            st[4 * j + 1] = c11 * den;
            st[4 * j + 2] = st[4 * j + 2] * inv_ma2 * den;
            st[4 * j + 3] = st[4 * j + 3] * inv_ma2 * den;
            st[20 + j] = den;
        }
        st[16] = inv_ma2;
    }
}

// ---- pass 3: per time-frequency point 2x2 solve and filter, in place on Y ---------------------------
__global__ __launch_bounds__(256) void k_wiener_apply(const float2* __restrict__ X, float2* __restrict__ Y,
                                                       const WRow* __restrict__ rows, const float* __restrict__ stats,
                                                       int Bn, int S, int win_len) {
    const WRow r = rows[blockIdx.y];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* st = stats + r.stat + (n / win_len) * STAT;
    const float inv_ma2 = st[16];
    const float2 x0 = X[cidx(r, 2 * Bn, S, r.b * 2, n)];
    const float2 x1 = X[cidx(r, 2 * Bn, S, r.b * 2 + 1, n)];
    float v[4];
    float2 R01[4];
    float R00[4], R11[4];
    int64_t yi[4];
    const float reg = sqrtf(FLT_EPSILON);
    float c00 = reg, c11 = reg;
    float2 c01 = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        yi[j] = cidx(r, 8 * Bn, S, (j * Bn + r.b) * 2, n);
        const float2 y0 = Y[yi[j]];
        const float2 y1 = Y[yi[j] + (int64_t)r.F * N];
        v[j] = 0.5f * ((y0.x * y0.x + y0.y * y0.y) * inv_ma2 + (y1.x * y1.x + y1.y * y1.y) * inv_ma2);
        R00[j] = st[4 * j]; R11[j] = st[4 * j + 1]; R01[j] = make_float2(st[4 * j + 2], st[4 * j + 3]);
        c00 += v[j] * R00[j];
        c11 += v[j] * R11[j];
        c01.x += v[j] * R01[j].x;
        c01.y += v[j] * R01[j].y;
    }
    // Cxx = [[c00, c01], [conj(c01), c11]];  analytic inverse (norbert _invert :337-346)
    const float det = c00 * c11 - (c01.x * c01.x + c01.y * c01.y);
    const float idet = 1.f / det;
    const float i00 = c11 * idet, i11 = c00 * idet;
    const float2 i01 = make_float2(-c01.x * idet, -c01.y * idet);    // -c01/det
    const float2 i10 = make_float2(-c01.x * idet, c01.y * idet);     // -conj(c01)/det
    // z = Cxx^-1 x
    const float2 z0 = make_float2(i00 * x0.x + (i01.x * x1.x - i01.y * x1.y), i00 * x0.y + (i01.x * x1.y + i01.y * x1.x));
    const float2 z1 = make_float2((i10.x * x0.x - i10.y * x0.y) + i11 * x1.x, (i10.x * x0.y + i10.y * x0.x) + i11 * x1.y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // y_j = v_j R_j z ;  R_j = [[R00, R01], [conj(R01), R11]]
        const float2 a = cmul(R01[j], z1);
        const float2 b = cmulc(z0, R01[j]);      // conj(R01) * z0
        Y[yi[j]] = make_float2(v[j] * (R00[j] * z0.x + a.x), v[j] * (R00[j] * z0.y + a.y));
        Y[yi[j] + (int64_t)r.F * N] = make_float2(v[j] * (b.x + R11[j] * z1.x), v[j] * (b.y + R11[j] * z1.y));
    }
}

// ---- the same two passes fed by MASKS: y0 = mask * x is formed on the way in ---------------------------------
// The CDAE's last layer then writes 4 bytes per coefficient instead of 8 (and does not read the mix), pass 1 reads
// 48 instead of 80 bytes per time-frequency point, pass 3 reads 48 and writes 64 instead of 80 + 64.  The products
// mask * re, mask * im are the ones the layer-4 epilogue would have stored (one fp32 rounding each), and everything
// downstream is the same expression tree: bitwise the two-step result.  Two frames per thread (N = S*T and the
// window length are even on this path): 16-byte loads of the mix, 8-byte loads of the masks, 16-byte stores.
__device__ inline int64_t ridx(const WRow& r, int nchan, int S, int chan, int64_t n) {      // real arena (masks)
    return (int64_t)nchan * S * r.cum + ((int64_t)chan * r.F + r.f) * ((int64_t)S * r.T) + n;
}

__global__ __launch_bounds__(256) void k_wiener_stats_masked(const float2* __restrict__ X, const float* __restrict__ Mk,
                                                              const WRow* __restrict__ rows, const int* __restrict__ work,
                                                              float* __restrict__ stats, int Bn, int S, int win_len) {
    const int row = work[2 * blockIdx.x], w = work[2 * blockIdx.x + 1];
    const WRow r = rows[row];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n0 = (int64_t)w * win_len;
    const int64_t n1 = n0 + win_len < N ? n0 + win_len : N;
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
    const float2* x0 = X + cidx(r, 2 * Bn, S, r.b * 2, 0);
    const float2* x1 = X + cidx(r, 2 * Bn, S, r.b * 2 + 1, 0);
    const float* m0 = Mk + ridx(r, 8 * Bn, S, r.b * 2, 0);            // target 0, channel 0 of this row
    const int64_t cstride = (int64_t)r.F * N, jstride = (int64_t)Bn * 2 * cstride;
    // The reduction keeps the two-step kernel's shape -- lane t adds frames n0 + t, n0 + t + 256, ... in order -- so the
    // sums round identically; the pair (t, t + 256) is what one thread of THIS kernel owns per 512-frame step.
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
        const float2 a = x0[n], b = x1[n];
        acc[16] = fmaxf(acc[16], fmaxf(a.x * a.x + a.y * a.y, b.x * b.x + b.y * b.y));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ma = m0[j * jstride + n], mb = m0[j * jstride + cstride + n];
            const float2 y0 = make_float2(ma * a.x, ma * a.y);
            const float2 y1 = make_float2(mb * b.x, mb * b.y);
            const float2 c01 = cmulc(y0, y1);
            acc[4 * j + 0] += y0.x * y0.x + y0.y * y0.y;
            acc[4 * j + 1] += y1.x * y1.x + y1.y * y1.y;
            acc[4 * j + 2] += c01.x;
            acc[4 * j + 3] += c01.y;
        }
    }
    __shared__ float red[4][17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v, off, 64);
            v = (i == 16) ? fmaxf(v, o) : v + o;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        const int i = threadIdx.x;
        float v;
        if (i == 16) v = fmaxf(fmaxf(red[0][i], red[1][i]), fmaxf(red[2][i], red[3][i]));
        else v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        stats[r.stat + (int64_t)w * STAT + i] = v;
    }
}

// one time-frequency point of pass 3 (shared by both apply kernels: same expression tree, same bits)
__device__ inline void wiener_point(const float* __restrict__ st, float2 x0, float2 x1, const float2 (&y)[4][2],
                                    float2 (&o)[4][2]) {
    const float inv_ma2 = st[16];
    float v[4];
    float2 R01[4];
    float R00[4], R11[4];
    const float reg = sqrtf(FLT_EPSILON);
    float c00 = reg, c11 = reg;
    float2 c01 = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 y0 = y[j][0], y1 = y[j][1];
        v[j] = 0.5f * ((y0.x * y0.x + y0.y * y0.y) * inv_ma2 + (y1.x * y1.x + y1.y * y1.y) * inv_ma2);
        R00[j] = st[4 * j]; R11[j] = st[4 * j + 1]; R01[j] = make_float2(st[4 * j + 2], st[4 * j + 3]);
        c00 += v[j] * R00[j];
        c11 += v[j] * R11[j];
        c01.x += v[j] * R01[j].x;
        c01.y += v[j] * R01[j].y;
    }
    const float det = c00 * c11 - (c01.x * c01.x + c01.y * c01.y);
    const float idet = 1.f / det;
    const float i00 = c11 * idet, i11 = c00 * idet;
    const float2 i01 = make_float2(-c01.x * idet, -c01.y * idet);
    const float2 i10 = make_float2(-c01.x * idet, c01.y * idet);
    const float2 z0 = make_float2(i00 * x0.x + (i01.x * x1.x - i01.y * x1.y), i00 * x0.y + (i01.x * x1.y + i01.y * x1.x));
    const float2 z1 = make_float2((i10.x * x0.x - i10.y * x0.y) + i11 * x1.x, (i10.x * x0.y + i10.y * x0.x) + i11 * x1.y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 a = cmul(R01[j], z1);
        const float2 b = cmulc(z0, R01[j]);
        o[j][0] = make_float2(v[j] * (R00[j] * z0.x + a.x), v[j] * (R00[j] * z0.y + a.y));
        o[j][1] = make_float2(v[j] * (b.x + R11[j] * z1.x), v[j] * (b.y + R11[j] * z1.y));
    }
}

__global__ __launch_bounds__(256) void k_wiener_apply_masked(const float2* __restrict__ X, const float* __restrict__ Mk,
                                                              float2* __restrict__ Y, const WRow* __restrict__ rows,
                                                              const float* __restrict__ stats, int Bn, int S, int win_len) {
    const WRow r = rows[blockIdx.y];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x);       // frames n, n + 1 (same window: both even)
    if (n >= N) return;
    const float* st = stats + r.stat + (n / win_len) * STAT;
    const float4 xa = *reinterpret_cast<const float4*>(X + cidx(r, 2 * Bn, S, r.b * 2, n));
    const float4 xb = *reinterpret_cast<const float4*>(X + cidx(r, 2 * Bn, S, r.b * 2 + 1, n));
    const int64_t cstride = (int64_t)r.F * N, jstride = (int64_t)Bn * 2 * cstride;
    const float* m0 = Mk + ridx(r, 8 * Bn, S, r.b * 2, n);
    float2 ma[4], mb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ma[j] = *reinterpret_cast<const float2*>(m0 + j * jstride);
        mb[j] = *reinterpret_cast<const float2*>(m0 + j * jstride + cstride);
    }
    float2 y[4][2], o0[4][2], o1[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j][0] = make_float2(ma[j].x * xa.x, ma[j].x * xa.y);
        y[j][1] = make_float2(mb[j].x * xb.x, mb[j].x * xb.y);
    }
    wiener_point(st, make_float2(xa.x, xa.y), make_float2(xb.x, xb.y), y, o0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j][0] = make_float2(ma[j].y * xa.z, ma[j].y * xa.w);
        y[j][1] = make_float2(mb[j].y * xb.z, mb[j].y * xb.w);
    }
    wiener_point(st, make_float2(xa.z, xa.w), make_float2(xb.z, xb.w), y, o1);
    float2* y0p = Y + cidx(r, 8 * Bn, S, r.b * 2, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<float4*>(y0p + j * jstride) = make_float4(o0[j][0].x, o0[j][0].y, o1[j][0].x, o1[j][0].y);
        *reinterpret_cast<float4*>(y0p + j * jstride + cstride) = make_float4(o0[j][1].x, o0[j][1].y, o1[j][1].x, o1[j][1].y);
    }
}

// ---- backward of the EM iteration (training: loss.backward() through norbert, training.py:107) --------
// Notation of the header; per point n: w = Cxx^-1 x, out_j = v_j R_j w.  With gO_j the incoming gradient,
//   q = sum_j v_j R_j gO_j,  p = Cxx^-1 q,  u_j = gO_j - p,
//   d/dv_j = Re(u_j^H R_j w),   d/dR_j = sum_n v_j u_j w^H  (only its Hermitian part K_j matters),
//   R_j = A_j / D_j:  H_j = K_j / D_j - Re tr(K_j/2 R_j) / D_j * I,
//   d/dy0[n,:,j] = (d/dv_j * y0 + H_j y0) / ma^2.
// Pass 1 accumulates K_j per (row, window) (same fixed-order reduction as the forward statistics), pass 2
// forms H_j, pass 3 rewrites the gradient arena in place.
struct WPoint {
    float v[4], gv[4];
    float2 u[4][2];
    float2 w0, w1;
};

__device__ inline void wiener_bwd_point(const float* __restrict__ st, float2 x0, float2 x1, const float2 (&y)[4][2],
                                        const float2 (&g)[4][2], WPoint& P) {
    const float inv_ma2 = st[16];
    const float reg = sqrtf(FLT_EPSILON);
    float c00 = reg, c11 = reg;
    float2 c01 = make_float2(0.f, 0.f);
    float R00[4], R11[4];
    float2 R01[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        P.v[j] = 0.5f * ((y[j][0].x * y[j][0].x + y[j][0].y * y[j][0].y) * inv_ma2 +
                         (y[j][1].x * y[j][1].x + y[j][1].y * y[j][1].y) * inv_ma2);
        R00[j] = st[4 * j]; R11[j] = st[4 * j + 1]; R01[j] = make_float2(st[4 * j + 2], st[4 * j + 3]);
        c00 += P.v[j] * R00[j];
        c11 += P.v[j] * R11[j];
        c01.x += P.v[j] * R01[j].x;
        c01.y += P.v[j] * R01[j].y;
    }
    const float idet = 1.f / (c00 * c11 - (c01.x * c01.x + c01.y * c01.y));
    const float i00 = c11 * idet, i11 = c00 * idet;
    const float2 i01 = make_float2(-c01.x * idet, -c01.y * idet);
    auto solve = [&](float2 a0, float2 a1, float2& o0, float2& o1) {      // Cxx^-1 a
        const float2 t = cmul(i01, a1), s = cmulc(a0, i01);               // i10 = conj(i01)
        o0 = make_float2(i00 * a0.x + t.x, i00 * a0.y + t.y);
        o1 = make_float2(s.x + i11 * a1.x, s.y + i11 * a1.y);
    };
    solve(x0, x1, P.w0, P.w1);
    float2 q0 = make_float2(0.f, 0.f), q1 = q0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // R_j gO_j
        const float2 a = cmul(R01[j], g[j][1]), b = cmulc(g[j][0], R01[j]);
        q0.x += P.v[j] * (R00[j] * g[j][0].x + a.x); q0.y += P.v[j] * (R00[j] * g[j][0].y + a.y);
        q1.x += P.v[j] * (b.x + R11[j] * g[j][1].x); q1.y += P.v[j] * (b.y + R11[j] * g[j][1].y);
    }
    float2 p0, p1;
    solve(q0, q1, p0, p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        P.u[j][0] = make_float2(g[j][0].x - p0.x, g[j][0].y - p0.y);
        P.u[j][1] = make_float2(g[j][1].x - p1.x, g[j][1].y - p1.y);
        const float2 a = cmul(R01[j], P.w1), b = cmulc(P.w0, R01[j]);
        const float2 rw0 = make_float2(R00[j] * P.w0.x + a.x, R00[j] * P.w0.y + a.y);
        const float2 rw1 = make_float2(b.x + R11[j] * P.w1.x, b.y + R11[j] * P.w1.y);
        P.gv[j] = (P.u[j][0].x * rw0.x + P.u[j][0].y * rw0.y) + (P.u[j][1].x * rw1.x + P.u[j][1].y * rw1.y);
    }
}

// Mk != nullptr: the pre-filter estimate is mask * x (the products the layer-4 epilogue would have stored)
__device__ inline void wiener_load_point(const float2* __restrict__ X, const float2* __restrict__ Y0, const float* __restrict__ Mk,
                                         const float2* __restrict__ G, const WRow& r, int Bn, int S, int64_t n,
                                         float2& x0, float2& x1, float2 (&y)[4][2], float2 (&g)[4][2], int64_t (&yi)[4]) {
    const int64_t N = (int64_t)S * r.T;
    x0 = X[cidx(r, 2 * Bn, S, r.b * 2, n)];
    x1 = X[cidx(r, 2 * Bn, S, r.b * 2 + 1, n)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        yi[j] = cidx(r, 8 * Bn, S, (j * Bn + r.b) * 2, n);
        if (Mk) {
            const float ma = Mk[yi[j]], mb = Mk[yi[j] + (int64_t)r.F * N];      // real arena: same index, one float each
            y[j][0] = make_float2(ma * x0.x, ma * x0.y); y[j][1] = make_float2(mb * x1.x, mb * x1.y);
        } else {
            y[j][0] = Y0[yi[j]]; y[j][1] = Y0[yi[j] + (int64_t)r.F * N];
        }
        g[j][0] = G[yi[j]]; g[j][1] = G[yi[j] + (int64_t)r.F * N];
    }
}

__global__ __launch_bounds__(256) void k_wiener_bwd_stats(const float2* __restrict__ X, const float2* __restrict__ Y0,
                                                           const float* __restrict__ Mk,
                                                           const float2* __restrict__ G, const WRow* __restrict__ rows,
                                                           const int* __restrict__ work, const float* __restrict__ stats,
                                                           float* __restrict__ bstats, int Bn, int S, int win_len) {
    const int row = work[2 * blockIdx.x], w = work[2 * blockIdx.x + 1];
    const WRow r = rows[row];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n0 = (int64_t)w * win_len;
    const int64_t n1 = n0 + win_len < N ? n0 + win_len : N;
    const float* st = stats + r.stat + (int64_t)w * STAT;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
        float2 x0, x1, y[4][2], g[4][2];
        int64_t yi[4];
        wiener_load_point(X, Y0, Mk, G, r, Bn, S, n, x0, x1, y, g, yi);
        WPoint P;
        wiener_bwd_point(st, x0, x1, y, g, P);
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // K = v (u w^H + w u^H)
            const float2 a = cmulc(P.u[j][0], P.w1), b = cmulc(P.w0, P.u[j][1]);
            acc[4 * j + 0] += 2.f * P.v[j] * (P.u[j][0].x * P.w0.x + P.u[j][0].y * P.w0.y);
            acc[4 * j + 1] += 2.f * P.v[j] * (P.u[j][1].x * P.w1.x + P.u[j][1].y * P.w1.y);
            acc[4 * j + 2] += P.v[j] * (a.x + b.x);
            acc[4 * j + 3] += P.v[j] * (a.y + b.y);
        }
    }
    __shared__ float red[4][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int i = threadIdx.x;
        bstats[r.stat + (int64_t)w * STAT + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// K_j -> H_j, one thread per (row, window)
__global__ void k_wiener_bwd_finalize(const WRow* __restrict__ rows, const int* __restrict__ work, int nwork,
                                      const float* __restrict__ stats, float* __restrict__ bstats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwork) return;
    const WRow r = rows[work[2 * i]];
    const int64_t o = r.stat + (int64_t)work[2 * i + 1] * STAT;
    const float* st = stats + o;
    float* bs = bstats + o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float den = st[20 + j];
        const float k00 = bs[4 * j], k11 = bs[4 * j + 1], k01x = bs[4 * j + 2], k01y = bs[4 * j + 3];
        const float s = (0.5f * (k00 * st[4 * j] + k11 * st[4 * j + 1]) + (k01x * st[4 * j + 2] + k01y * st[4 * j + 3])) * den;
        bs[4 * j] = k00 * den - s;
        bs[4 * j + 1] = k11 * den - s;
        bs[4 * j + 2] = k01x * den;
        bs[4 * j + 3] = k01y * den;
    }
}

__global__ __launch_bounds__(256) void k_wiener_bwd_apply(const float2* __restrict__ X, const float2* __restrict__ Y0,
                                                           const float* __restrict__ Mk,
                                                           float2* __restrict__ G, const WRow* __restrict__ rows,
                                                           const float* __restrict__ stats, const float* __restrict__ bstats,
                                                           int Bn, int S, int win_len, float* __restrict__ gM = nullptr) {
    const WRow r = rows[blockIdx.y];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t o = r.stat + (n / win_len) * STAT;
    const float* st = stats + o;
    const float* bs = bstats + o;
    float2 x0, x1, y[4][2], g[4][2];
    int64_t yi[4];
    wiener_load_point(X, Y0, Mk, G, r, Bn, S, n, x0, x1, y, g, yi);
    WPoint P;
    wiener_bwd_point(st, x0, x1, y, g, P);
    const float inv_ma2 = st[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float h00 = bs[4 * j], h11 = bs[4 * j + 1];
        const float2 h01 = make_float2(bs[4 * j + 2], bs[4 * j + 3]);
        const float2 a = cmul(h01, y[j][1]), b = cmulc(y[j][0], h01);
        const float2 g0 = make_float2((P.gv[j] * y[j][0].x + h00 * y[j][0].x + a.x) * inv_ma2,
                                      (P.gv[j] * y[j][0].y + h00 * y[j][0].y + a.y) * inv_ma2);
        const float2 g1 = make_float2((P.gv[j] * y[j][1].x + b.x + h11 * y[j][1].x) * inv_ma2,
                                      (P.gv[j] * y[j][1].y + b.y + h11 * y[j][1].y) * inv_ma2);
        const int64_t i0 = yi[j], i1 = yi[j] + (int64_t)r.F * N;
        if (gM) {
            // training step: the gradient of the pre-filter estimate y0 = m x goes straight on through the product and the
            // sigmoid (train.hip: k_mask_bwd) -- d/dm = Re(conj(x) g) on top of the mask-sum gradient already in gM, times
            // m (1 - m) -- instead of being stored (64 B per point) and read back by a second pass
            const float m0 = Mk[i0], m1 = Mk[i1];
            gM[i0] = (x0.x * g0.x + x0.y * g0.y + gM[i0]) * m0 * (1.f - m0);
            gM[i1] = (x1.x * g1.x + x1.y * g1.y + gM[i1]) * m1 * (1.f - m1);
        } else {
            G[i0] = g0;
            G[i1] = g1;
        }
    }
}

// ---- max |x|^2 per (block, group, window) of ONE pass of a split batch, folded into ext_max (non-negative floats order
// like their bit patterns: atomicMax on the words).  One workgroup per (row, window), as the statistics pass.
__global__ __launch_bounds__(256) void k_wiener_window_max(const float2* __restrict__ X, const WRow* __restrict__ rows,
                                                            const int* __restrict__ work, const int* __restrict__ bw_of_work,
                                                            float* __restrict__ ext_max, int Bn, int S, int win_len) {
    const int row = work[2 * blockIdx.x], w = work[2 * blockIdx.x + 1];
    const WRow r = rows[row];
    const int64_t N = (int64_t)S * r.T;
    const int64_t n0 = (int64_t)w * win_len;
    const int64_t n1 = n0 + win_len < N ? n0 + win_len : N;
    const float2* x0 = X + cidx(r, 2 * Bn, S, r.b * 2, 0);
    const float2* x1 = X + cidx(r, 2 * Bn, S, r.b * 2 + 1, 0);
    float m = 0.f;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += 256) {
        const float2 a = x0[n], b = x1[n];
        m = fmaxf(m, fmaxf(a.x * a.x + a.y * a.y, b.x * b.x + b.y * b.y));      // the expression of the statistics pass: same bits
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(ext_max) + bw_of_work[blockIdx.x], __builtin_bit_cast(unsigned, m));
}

// ------------------------------------------------------------------------------------------------
static int get_wtable(int nblocks, const int32_t* F, const int32_t* T, int Bn, int S, int win_len, int group, WTable* out) {
    std::vector<int> key;
    int dev = 0;
    XSQ_HIP(hipGetDevice(&dev));                 // the tables live in one device's memory: keyed by it
    key.push_back(dev);
    key.push_back(nblocks); key.push_back(Bn); key.push_back(S); key.push_back(win_len); key.push_back(group);
    for (int b = 0; b < nblocks; ++b) { key.push_back(F[b]); key.push_back(T[b]); }
    std::lock_guard<std::mutex> lk(g_wmu);
    auto it = g_wtables.find(key);
    if (it != g_wtables.end()) { *out = it->second; return XSQ_OK; }
    std::vector<WRow> rows;
    std::vector<int> work, blockwin, bw_of_work;
    int64_t cum = 0, stat = 0, maxN = 0;
    for (int k = 0; k < nblocks; ++k) {
        const int64_t N = (int64_t)S * T[k];
        const int nwin = (int)((N + win_len - 1) / win_len);
        const int first = (int)rows.size();
        const int bw0 = (int)blockwin.size() / 2;          // this block's (group, window) entries start here
        maxN = N > maxN ? N : maxN;
        for (int b = 0; b < Bn; ++b)
            for (int f = 0; f < F[k]; ++f) {
                for (int w = 0; w < nwin; ++w) bw_of_work.push_back(bw0 + (b / group) * nwin + w);
                WRow r;
                r.F = F[k]; r.T = T[k]; r.b = b; r.f = f; r.nwin = nwin;
                // rows sharing one window maximum: the `group` consecutive batch items of this row's group
                r.first_row = first + (b / group) * group * F[k];
                r.nrows = group * F[k]; r.pad = 0; r.cum = cum; r.stat = stat;
                for (int w = 0; w < nwin; ++w) { work.push_back((int)rows.size()); work.push_back(w); }
                rows.push_back(r);
                stat += (int64_t)nwin * STAT;
            }
        for (int gI = 0; gI < Bn / group; ++gI)
            for (int w = 0; w < nwin; ++w) { blockwin.push_back(first + gI * group * F[k]); blockwin.push_back(w); }
        cum += (int64_t)F[k] * T[k];
    }
    WTable t;
    t.nrows = (int)rows.size(); t.nwork = (int)work.size() / 2; t.nblockwin = (int)blockwin.size() / 2;
    t.stat_floats = stat; t.max_frames = maxN;
    XSQ_HIP(hipMalloc(&t.d_rows, rows.size() * sizeof(WRow)));
    XSQ_HIP(hipMemcpy(t.d_rows, rows.data(), rows.size() * sizeof(WRow), hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&t.d_work, work.size() * sizeof(int)));
    XSQ_HIP(hipMemcpy(t.d_work, work.data(), work.size() * sizeof(int), hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&t.d_blockwin, blockwin.size() * sizeof(int)));
    XSQ_HIP(hipMemcpy(t.d_blockwin, blockwin.data(), blockwin.size() * sizeof(int), hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&t.d_bw_of_work, bw_of_work.size() * sizeof(int)));
    XSQ_HIP(hipMemcpy(t.d_bw_of_work, bw_of_work.data(), bw_of_work.size() * sizeof(int), hipMemcpyHostToDevice));
    g_wtables[key] = t;
    *out = t;
    return XSQ_OK;
}

static int check_table(const char* who, int nblocks, const int32_t* F, const int32_t* T, int Bn, int S) {
    XSQ_REQUIRE(nblocks > 0 && F && T, "%s: null block table", who);
    XSQ_REQUIRE(Bn > 0 && S > 0, "%s: B=%d S=%d", who, Bn, S);
    int64_t rows = 0;
    for (int b = 0; b < nblocks; ++b) {
        XSQ_REQUIRE(F[b] > 0 && T[b] > 0, "%s: block %d has F=%d T=%d", who, b, F[b], T[b]);
        rows += (int64_t)Bn * F[b];
    }
    XSQ_REQUIRE(rows <= 65535, "%s: %lld rows exceed one launch", who, (long long)rows);
    return XSQ_OK;
}

// backward of xsq_wiener_em: `stats` is the workspace the forward call left behind, G holds dL/d(out) on entry
// and dL/d(y0) on return, Y0 is the pre-filter estimate.  bstats: another workspace of the same size.
int wiener_em_backward(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* Y0, const float* masks,
                       float* G, int Bn, int S, int win_len, int batch_group, const void* stats, void* bstats, hipStream_t stream,
                       float* gM) {
    if (batch_group <= 0) batch_group = Bn;
    WTable t;
    int rc;
    if ((rc = get_wtable(nblocks, F, T, Bn, S, win_len, batch_group, &t))) return rc;
    { XSQ_PROF("wiener_bwd_stats", stream);
    hipLaunchKernelGGL(k_wiener_bwd_stats, dim3(t.nwork), dim3(256), 0, stream, (const float2*)X, (const float2*)Y0,
                       Y0 ? nullptr : masks, (const float2*)G, t.d_rows, t.d_work, (const float*)stats, (float*)bstats, Bn, S, win_len); }
    { XSQ_PROF("wiener_bwd_finalize", stream);
    hipLaunchKernelGGL(k_wiener_bwd_finalize, dim3((t.nwork + 255) / 256), dim3(256), 0, stream, t.d_rows, t.d_work, t.nwork,
                       (const float*)stats, (float*)bstats); }
    { XSQ_PROF("wiener_bwd_apply", stream);
    hipLaunchKernelGGL(k_wiener_bwd_apply, dim3((unsigned)((t.max_frames + 255) / 256), t.nrows), dim3(256), 0, stream,
                       (const float2*)X, (const float2*)Y0, Y0 ? nullptr : masks, (float2*)G, t.d_rows, (const float*)stats,
                       (const float*)bstats, Bn, S, win_len, (Y0 == nullptr && masks) ? gM : nullptr); }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

}  // namespace xsq

using namespace xsq;

extern "C" {

int xsq_phasemix(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* mag, float* Y,
                 int Bn, int S, void* stream_) {
    int rc = check_table("xsq_phasemix", nblocks, F, T, Bn, S);
    if (rc) return rc;
    XSQ_REQUIRE(X && mag && Y, "xsq_phasemix: null argument");
    WTable t;
    if ((rc = get_wtable(nblocks, F, T, Bn, S, 5000, Bn, &t))) return rc;
    hipLaunchKernelGGL(k_phasemix, dim3((unsigned)((t.max_frames + 255) / 256), t.nrows), dim3(256), 0,
                       (hipStream_t)stream_, (const float2*)X, mag, (float2*)Y, t.d_rows, Bn, S);
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

size_t xsq_wiener_workspace(int nblocks, const int32_t* F, const int32_t* T, int Bn, int S, int win_len) {
    if (nblocks <= 0 || !F || !T || Bn <= 0 || S <= 0 || win_len <= 0) return 0;
    int64_t stat = 0;
    for (int k = 0; k < nblocks; ++k)
        stat += (int64_t)Bn * F[k] * (((int64_t)S * T[k] + win_len - 1) / win_len) * STAT;
    return (size_t)stat * 4 + 256;
}

int xsq_wiener_em(int nblocks, const int32_t* F, const int32_t* T, const float* X, float* Y, int Bn, int S,
                  int win_len, int batch_group, void* ws, size_t ws_bytes, void* stream_) {
    int rc = check_table("xsq_wiener_em", nblocks, F, T, Bn, S);
    if (rc) return rc;
    XSQ_REQUIRE(X && Y && ws, "xsq_wiener_em: null argument");
    XSQ_REQUIRE(win_len > 0, "xsq_wiener_em: win_len=%d", win_len);
    if (batch_group <= 0) batch_group = Bn;
    XSQ_REQUIRE(Bn % batch_group == 0, "xsq_wiener_em: batch_group=%d does not divide B=%d", batch_group, Bn);
    XSQ_REQUIRE(ws_bytes >= xsq_wiener_workspace(nblocks, F, T, Bn, S, win_len), "xsq_wiener_em: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    WTable t;
    if ((rc = get_wtable(nblocks, F, T, Bn, S, win_len, batch_group, &t))) return rc;
    float* stats = (float*)ws;
    { XSQ_PROF("wiener_stats", stream);
    hipLaunchKernelGGL(k_wiener_stats, dim3(t.nwork), dim3(256), 0, stream, (const float2*)X, (const float2*)Y,
                       t.d_rows, t.d_work, stats, Bn, S, win_len); }
    { XSQ_PROF("wiener_finalize", stream);
    hipLaunchKernelGGL(k_wiener_finalize, dim3(t.nblockwin), dim3(256), 0, stream, t.d_rows, t.d_blockwin, stats); }
    { XSQ_PROF("wiener_apply", stream);
    hipLaunchKernelGGL(k_wiener_apply, dim3((unsigned)((t.max_frames + 255) / 256), t.nrows), dim3(256), 0, stream,
                       (const float2*)X, (float2*)Y, t.d_rows, stats, Bn, S, win_len); }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

int xsq_wiener_em_masked(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* masks, float* Y,
                         int Bn, int S, int win_len, int batch_group, void* ws, size_t ws_bytes, void* stream_) {
    return xsq_wiener_em_masked_ext(nblocks, F, T, X, masks, Y, Bn, S, win_len, batch_group, nullptr, ws, ws_bytes, stream_);
}

int64_t xsq_wiener_num_windows(int nblocks, const int32_t* F, const int32_t* T, int Bn, int S, int win_len, int batch_group) {
    if (nblocks <= 0 || !F || !T || Bn <= 0 || S <= 0 || win_len <= 0) return 0;
    if (batch_group <= 0) batch_group = Bn;
    if (Bn % batch_group) return 0;
    int64_t n = 0;
    for (int k = 0; k < nblocks; ++k) n += (int64_t)(Bn / batch_group) * (((int64_t)S * T[k] + win_len - 1) / win_len);
    return n;
}

int xsq_wiener_window_max(int nblocks, const int32_t* F, const int32_t* T, const float* X, int Bn, int S, int win_len,
                          int batch_group, float* ext_max, void* stream_) {
    int rc = check_table("xsq_wiener_window_max", nblocks, F, T, Bn, S);
    if (rc) return rc;
    XSQ_REQUIRE(X && ext_max && win_len > 0, "xsq_wiener_window_max: bad argument");
    if (batch_group <= 0) batch_group = Bn;
    XSQ_REQUIRE(Bn % batch_group == 0, "xsq_wiener_window_max: batch_group=%d does not divide B=%d", batch_group, Bn);
    WTable t;
    if ((rc = get_wtable(nblocks, F, T, Bn, S, win_len, batch_group, &t))) return rc;
    XSQ_PROF("wiener_window_max", (hipStream_t)stream_);
    hipLaunchKernelGGL(k_wiener_window_max, dim3(t.nwork), dim3(256), 0, (hipStream_t)stream_, (const float2*)X, t.d_rows, t.d_work,
                       t.d_bw_of_work, ext_max, Bn, S, win_len);
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

int xsq_wiener_em_masked_ext(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* masks, float* Y,
                             int Bn, int S, int win_len, int batch_group, const float* ext_max, void* ws, size_t ws_bytes,
                             void* stream_) {
    int rc = check_table("xsq_wiener_em_masked", nblocks, F, T, Bn, S);
    if (rc) return rc;
    XSQ_REQUIRE(X && masks && Y && ws, "xsq_wiener_em_masked: null argument");
    XSQ_REQUIRE(win_len > 0 && win_len % 2 == 0, "xsq_wiener_em_masked: win_len=%d must be even (two frames per thread)", win_len);
    for (int b = 0; b < nblocks; ++b)
        XSQ_REQUIRE(((int64_t)S * T[b]) % 2 == 0, "xsq_wiener_em_masked: block %d has an odd frame count S*T=%lld", b, (long long)S * T[b]);
    if (batch_group <= 0) batch_group = Bn;
    XSQ_REQUIRE(Bn % batch_group == 0, "xsq_wiener_em_masked: batch_group=%d does not divide B=%d", batch_group, Bn);
    XSQ_REQUIRE(ws_bytes >= xsq_wiener_workspace(nblocks, F, T, Bn, S, win_len), "xsq_wiener_em_masked: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    WTable t;
    if ((rc = get_wtable(nblocks, F, T, Bn, S, win_len, batch_group, &t))) return rc;
    float* stats = (float*)ws;
    { XSQ_PROF("wiener_stats", stream);
    hipLaunchKernelGGL(k_wiener_stats_masked, dim3(t.nwork), dim3(256), 0, stream, (const float2*)X, masks,
                       t.d_rows, t.d_work, stats, Bn, S, win_len); }
    { XSQ_PROF("wiener_finalize", stream);
    hipLaunchKernelGGL(k_wiener_finalize, dim3(t.nblockwin), dim3(256), 0, stream, t.d_rows, t.d_blockwin, stats, ext_max); }
    { XSQ_PROF("wiener_apply", stream);
    hipLaunchKernelGGL(k_wiener_apply_masked, dim3((unsigned)((t.max_frames / 2 + 255) / 256), t.nrows), dim3(256), 0, stream,
                       (const float2*)X, masks, (float2*)Y, t.d_rows, stats, Bn, S, win_len); }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

}  // extern "C"
