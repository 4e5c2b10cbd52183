// Training step of the CDAE (BASELINE config 5; reference training.py:66-108) for gfx950.
//
//   forward   |X| whitening -> L1..L3 (MFMA engine of cdae.hip, raw outputs) each followed by BatchNorm on
//             BATCH statistics + ReLU -> L4 + sigmoid -> masks, Y = mask * X [-> Wiener-EM, wiener.hip]
//   loss      ComplexMSE over the 14 target combinations + MaskSum (loss.hip), and their gradients
//   backward  direct-form kernels for every layer (weight, bias, BatchNorm affine, input whitening and
//             data gradients) -- correctness first: one thread per output element, no tiling; the MFMA
//             forms (the data gradients are the forward operators of the mirrored layers) are the next step
//   update    AdamW (training.py:391-393: lr 1e-3, weight decay 1e-5) on the canonical parameter pool
//
// Parameters, gradients and optimizer moments live in ONE flat fp32 pool each, in the reference's
// state_dict order (the order xsq_model_create consumes), so gradients compare key by key with the
// reference's autograd and a checkpoint is a single copy.  The GEMM-layout weights the MFMA forward
// needs are re-gathered from the pool on the device every step through an index map.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/xumx_slicq_hip.h"
#include "cdae_api.h"
#include "prof.h"
#include "wgrad.h"

namespace xsq {

static const float BN_EPS_F = 1e-5f;

struct TrainGroup {     // one (block, target)
    int block, tgt, F, T, hop, kf, F1, F2;
    int cumF1, cumF2;
    int C1, C2;         // real channel counts (50, 51)
    int64_t cum, cumF;
    int64_t p_w1, p_bn1, p_w2, p_bn2, p_w3, p_bn3, p_w4, p_b4;   // canonical pool offsets; bn: weight, bias, rm, rv
    int64_t p_mean, p_scale;
};

struct TrainDims {
    int Bn, S, T1, T2, causal;
};

__device__ __forceinline__ int64_t act1_off(const TrainGroup& g, const TrainDims& d) {
    return (int64_t)CS * d.Bn * d.T1 * (4 * (int64_t)g.cumF1 + (int64_t)g.tgt * g.F1);
}
__device__ __forceinline__ int64_t act2_off(const TrainGroup& g, const TrainDims& d) {
    return (int64_t)CS * d.Bn * d.T2 * (4 * (int64_t)g.cumF2 + (int64_t)g.tgt * g.F2);
}
// real arena (8B channels): element (tgt, b, c, f, tau) of the group's block
__device__ __forceinline__ int64_t r8_idx(const TrainGroup& g, const TrainDims& d, int b, int c, int f, int64_t tau) {
    const int64_t ST = (int64_t)d.S * g.T;
    return (int64_t)d.Bn * 8 * d.S * g.cum + ((int64_t)((g.tgt * d.Bn + b) * 2 + c) * g.F + f) * ST + tau;
}
// real arena (2B channels): element (b, c, f, tau)
__device__ __forceinline__ int64_t r2_idx(const TrainGroup& g, const TrainDims& d, int b, int c, int f, int64_t tau) {
    const int64_t ST = (int64_t)d.S * g.T;
    return (int64_t)d.Bn * 2 * d.S * g.cum + ((int64_t)(b * 2 + c) * g.F + f) * ST + tau;
}

// ---- gather: dst[i] = map[i] >= 0 ? src[map[i]] : 0 -----------------------------------------------
__global__ void k_gather(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = map[i] >= 0 ? src[map[i]] : 0.f;
}

// ---- BatchNorm (train mode) -----------------------------------------------------------------------
// stats layout per (group, layer): mean[64] | invstd[64] | sum_g[64] | sum_gz[64]
__device__ __forceinline__ float* bn_slot(float* stats, int group, int layer) { return stats + ((int64_t)group * 3 + layer) * 256; }

// Reductions over the rows of a group run in two deterministic stages: row tiles of BN_ROWS rows write double partial
// sums (fixed-order tree inside the workgroup), one small workgroup per group adds its tiles in order.
static const int BN_ROWS = 256;
struct BnTile { int group, r0, r1; };
struct BnInfo { int base, n; };

// The BatchNorm output in front of the ReLU, spelled ONCE: the forward pass applies the ReLU to it and the two backward
// passes re-derive the ReLU's mask from it (a > 0 <=> pre > 0) instead of reading the activation array back -- a third
// less traffic in both -- which only holds if all three round identically: no contraction left to the compiler here.
__device__ __forceinline__ float bn_pre(float z, float mean, float inv, float gamma, float beta) {
#pragma clang fp contract(off)
    const float zh = (z - mean) * inv;
    const float t = zh * gamma;
    return t + beta;
}

// stage 1, one workgroup per row tile: per channel sum and sum of squares.  A row is 13 float4 (52 channels, 208 B,
// 16-byte aligned): thread = (row lane tid / 13 of 16, channel quad tid % 13), 16-byte loads, 16 rows of the tile per
// step (the first form read 4 bytes per lane and walked 64 rows per thread: 0.05 ms per launch of pure latency).
static const int BN_RL = 16, BN_Q = CS / 4;             // row lanes, float4 per row
__device__ __forceinline__ void bn_tile_reduce(double (&s1)[4], double (&s2)[4], double* __restrict__ o) {
    __shared__ double q1[BN_RL][CS], q2[BN_RL][CS];
    const int tid = threadIdx.x, rl = tid / BN_Q, q = tid - rl * BN_Q;
    if (tid < BN_RL * BN_Q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { q1[rl][4 * q + k] = s1[k]; q2[rl][4 * q + k] = s2[k]; }
    }
    __syncthreads();
    if (tid < CS) {                  // fixed order: bitwise reproducible
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int r = 0; r < BN_RL; ++r) { a += q1[r][tid]; b += q2[r][tid]; }
        o[tid] = a;
        o[64 + tid] = b;
    }
}

__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ z, const TrainGroup* __restrict__ groups,
                                                           const BnTile* __restrict__ tiles, TrainDims d, int layer,
                                                           double* __restrict__ part) {
    const BnTile t = tiles[blockIdx.x];
    const TrainGroup g = groups[t.group];
    const float4* zz = reinterpret_cast<const float4*>(z + (layer == 1 ? act2_off(g, d) : act1_off(g, d)));
    const int tid = threadIdx.x, rl = tid / BN_Q, q = tid - rl * BN_Q;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (tid < BN_RL * BN_Q)
        for (int m = t.r0 + rl; m < t.r1; m += BN_RL) {
            const float4 v = zz[(int64_t)m * BN_Q + q];
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            s1[0] += x0; s1[1] += x1; s1[2] += x2; s1[3] += x3;
            s2[0] += x0 * x0; s2[1] += x1 * x1; s2[2] += x2 * x2; s2[3] += x3 * x3;
        }
    bn_tile_reduce(s1, s2, part + (int64_t)blockIdx.x * 128);
}

// stage 2, one workgroup per group: the group's tile partials added in a FIXED order -- tile lane ty (of 16) adds tiles ty,
// ty + 16, ... in sequence, the 16 lane sums are added pairwise in a fixed tree (bitwise reproducible).  (Round 3: one wave
// per group walked all tiles, one dependent double load after the other -- 35-70 us per launch for the 108 tiles of the
// 86-bin block, six launches per step.)
static const int BN_FL = 16;
__device__ __forceinline__ void bn_final_sum(const double* __restrict__ part, const BnInfo bi, int c, int ty, double& a, double& b) {
    __shared__ double sa[BN_FL][64], sb[BN_FL][64];
    double x = 0.0, y = 0.0;
    for (int ch = ty; ch < bi.n; ch += BN_FL) { const double* o = part + (int64_t)(bi.base + ch) * 128; x += o[c]; y += o[64 + c]; }
    sa[ty][c] = x; sb[ty][c] = y;
    __syncthreads();
#pragma unroll
    for (int s = BN_FL / 2; s > 0; s >>= 1) {
        if (ty < s) { sa[ty][c] += sa[ty + s][c]; sb[ty][c] += sb[ty + s][c]; }
        __syncthreads();
    }
    a = sa[0][c]; b = sb[0][c];
}

// batch mean / biased variance, running-stat update (momentum 0.1)
__global__ __launch_bounds__(64 * BN_FL) void k_bn_stats_final(const double* __restrict__ part, const TrainGroup* __restrict__ groups,
                                                        const BnInfo* __restrict__ info, TrainDims d, int layer,
                                                        float* __restrict__ stats, float* __restrict__ pool, int update_running) {
    const TrainGroup g = groups[blockIdx.x];
    const int64_t M = (int64_t)d.Bn * (layer == 1 ? g.F2 : g.F1) * (layer == 1 ? d.T2 : d.T1);
    const int C = layer == 1 ? g.C2 : g.C1;
    const int c = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double a, b;
    bn_final_sum(part, info[blockIdx.x], c, ty, a, b);
    if (c >= C || ty != 0) return;
    const double mean = a / (double)M;
    double var = b / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    float* st = bn_slot(stats, blockIdx.x, layer);
    st[c] = (float)mean;
    st[64 + c] = (float)(1.0 / sqrt(var + (double)BN_EPS_F));
    if (update_running) {
        const int64_t pb = layer == 0 ? g.p_bn1 : (layer == 1 ? g.p_bn2 : g.p_bn3);
        float* rm = pool + pb + 2 * C;
        float* rv = pool + pb + 3 * C;
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rm[c] = (float)(0.9 * (double)rm[c] + 0.1 * mean);
        rv[c] = (float)(0.9 * (double)rv[c] + 0.1 * unbiased);
    }
}

// The channels-last arrays are contiguous over all groups and every frequency row of a group is rows_per_f = Bn * T consecutive
// rows of 13 float4: one workgroup per frequency row, thread = (row r0 = tid / 13 of 19, float4 q = tid % 13).  A thread keeps
// the constants of its four channels in registers and walks rows r0, r0 + 19, ... two at a time.  (As one thread per float4 of the
// flat array every thread fetched its 16-24 per-channel constants -- a dependent frow -> group -> table chain of 4-byte gathers
// per 32 streamed bytes: 0.31 + 0.18 ms per step at 2.2 TB/s.)  frow maps the (group-major) frequency-row index to its group.
//   a = relu((z - mean) * invstd * gamma + beta); pad channels -> 0
constexpr int BN_AROWS = 19;         // rows per trip: 13 * 19 = 247 of 256 threads
__global__ __launch_bounds__(256) void k_bn_relu_apply(const float4* __restrict__ z, float4* __restrict__ a,
                                                        const TrainGroup* __restrict__ groups, const int* __restrict__ frow,
                                                        int rows_per_f, int64_t nquads, int layer,
                                                        const float* __restrict__ stats, const float* __restrict__ pool) {
    const int r0 = threadIdx.x / 13, q = threadIdx.x - 13 * r0;
    if (r0 >= BN_AROWS) return;
    const int gid = frow[blockIdx.x];
    const TrainGroup& g = groups[gid];
    const int C = layer == 1 ? g.C2 : g.C1;
    const int64_t pb = layer == 0 ? g.p_bn1 : (layer == 1 ? g.p_bn2 : g.p_bn3);
    const float* st = stats + ((int64_t)gid * 3 + layer) * 256;
    float mean[4], inv[4], gam[4], bet[4];
    bool on[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k, cc = c < C ? c : 0;
        on[k] = c < C;
        mean[k] = st[cc]; inv[k] = st[64 + cc]; gam[k] = pool[pb + cc]; bet[k] = pool[pb + C + cc];
    }
    const float4* zr = z + (int64_t)blockIdx.x * rows_per_f * 13 + q;
    float4* ar = a + (int64_t)blockIdx.x * rows_per_f * 13 + q;
    auto one = [&](float4 v) {
        const float in[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = on[k] ? fmaxf(bn_pre(in[k], mean[k], inv[k], gam[k], bet[k]), 0.f) : 0.f;
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    int r = r0;
    for (; r + BN_AROWS < rows_per_f; r += 2 * BN_AROWS) {
        const float4 v0 = zr[r * 13], v1 = zr[(r + BN_AROWS) * 13];
        ar[r * 13] = one(v0);
        ar[(r + BN_AROWS) * 13] = one(v1);
    }
    if (r < rows_per_f) ar[r * 13] = one(zr[r * 13]);
}

// backward, step 1 (two stages like the statistics): per channel sum(g_bn) and sum(g_bn * zhat), g_bn = g_a * [a > 0]
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ z, const float* __restrict__ ga,
                                                         const TrainGroup* __restrict__ groups,
                                                         const BnTile* __restrict__ tiles, TrainDims d, int layer,
                                                         const float* __restrict__ stats, const float* __restrict__ pool,
                                                         double* __restrict__ part) {
    const BnTile t = tiles[blockIdx.x];
    const TrainGroup g = groups[t.group];
    const int64_t off = layer == 1 ? act2_off(g, d) : act1_off(g, d);
    const float4* z4 = reinterpret_cast<const float4*>(z + off);
    const float4* g4 = reinterpret_cast<const float4*>(ga + off);
    const int tid = threadIdx.x, rl = tid / BN_Q, q = tid - rl * BN_Q;
    const float* st = stats + ((int64_t)t.group * 3 + layer) * 256;
    const int C = layer == 1 ? g.C2 : g.C1;
    const int64_t pb = layer == 0 ? g.p_bn1 : (layer == 1 ? g.p_bn2 : g.p_bn3);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (tid < BN_RL * BN_Q) {
        const float4 mean = *reinterpret_cast<const float4*>(st + 4 * q), inv = *reinterpret_cast<const float4*>(st + 64 + 4 * q);
        float gam[4], bet[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {           // pad channels: gamma = beta = 0 -> pre = 0 -> masked, as the stored zero was
            const int c = 4 * q + k;
            gam[k] = c < C ? pool[pb + c] : 0.f;
            bet[k] = c < C ? pool[pb + C + c] : 0.f;
        }
        for (int m = t.r0 + rl; m < t.r1; m += BN_RL) {
            const int64_t i = (int64_t)m * BN_Q + q;
            const float4 zv = z4[i], gv = g4[i];
            const float g0 = bn_pre(zv.x, mean.x, inv.x, gam[0], bet[0]) > 0.f ? gv.x : 0.f;
            const float g1 = bn_pre(zv.y, mean.y, inv.y, gam[1], bet[1]) > 0.f ? gv.y : 0.f;
            const float g2 = bn_pre(zv.z, mean.z, inv.z, gam[2], bet[2]) > 0.f ? gv.z : 0.f;
            const float g3 = bn_pre(zv.w, mean.w, inv.w, gam[3], bet[3]) > 0.f ? gv.w : 0.f;
            s1[0] += g0; s1[1] += g1; s1[2] += g2; s1[3] += g3;
            s2[0] += (double)g0 * (double)((zv.x - mean.x) * inv.x);
            s2[1] += (double)g1 * (double)((zv.y - mean.y) * inv.y);
            s2[2] += (double)g2 * (double)((zv.z - mean.z) * inv.z);
            s2[3] += (double)g3 * (double)((zv.w - mean.w) * inv.w);
        }
    }
    bn_tile_reduce(s1, s2, part + (int64_t)blockIdx.x * 128);
}

__global__ __launch_bounds__(64 * BN_FL) void k_bn_bwd_final(const double* __restrict__ part, const TrainGroup* __restrict__ groups,
                                                      const BnInfo* __restrict__ info, TrainDims d, int layer,
                                                      float* __restrict__ stats, float* __restrict__ gpool) {
    const TrainGroup g = groups[blockIdx.x];
    const int64_t M = (int64_t)d.Bn * (layer == 1 ? g.F2 : g.F1) * (layer == 1 ? d.T2 : d.T1);
    const int C = layer == 1 ? g.C2 : g.C1;
    const int c = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double sg, sgz;
    bn_final_sum(part, info[blockIdx.x], c, ty, sg, sgz);
    if (c >= C || ty != 0) return;
    float* st = bn_slot(stats, blockIdx.x, layer);
    st[128 + c] = (float)(sg / (double)M);
    st[192 + c] = (float)(sgz / (double)M);
    const int64_t pb = layer == 0 ? g.p_bn1 : (layer == 1 ? g.p_bn2 : g.p_bn3);
    gpool[pb + c] = (float)sgz;        // d gamma
    gpool[pb + C + c] = (float)sg;     // d beta
}

// backward, step 2 (in place on ga; same workgroup shape as k_bn_relu_apply): g_z = gamma * invstd * (g_bn - mean(g_bn) - zhat * mean(g_bn * zhat))
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float4* __restrict__ z,
                                                       float4* __restrict__ ga, const TrainGroup* __restrict__ groups,
                                                       const int* __restrict__ frow, int rows_per_f, int64_t nquads, int layer,
                                                       const float* __restrict__ stats, const float* __restrict__ pool) {
    const int r0 = threadIdx.x / 13, q = threadIdx.x - 13 * r0;
    if (r0 >= BN_AROWS) return;
    const int gid = frow[blockIdx.x];
    const TrainGroup& g = groups[gid];
    const int C = layer == 1 ? g.C2 : g.C1;
    const int64_t pb = layer == 0 ? g.p_bn1 : (layer == 1 ? g.p_bn2 : g.p_bn3);
    const float* st = stats + ((int64_t)gid * 3 + layer) * 256;
    float mean[4], inv[4], gam[4], bet[4], mg[4], mgz[4];
    bool on[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k, cc = c < C ? c : 0;
        on[k] = c < C;
        mean[k] = st[cc]; inv[k] = st[64 + cc]; mg[k] = st[128 + cc]; mgz[k] = st[192 + cc];
        gam[k] = pool[pb + cc]; bet[k] = pool[pb + C + cc];
    }
    const float4* zr = z + (int64_t)blockIdx.x * rows_per_f * 13 + q;
    float4* gr = ga + (int64_t)blockIdx.x * rows_per_f * 13 + q;
    auto one = [&](float4 zv, float4 gv) {
        const float zi[4] = {zv.x, zv.y, zv.z, zv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float zh = (zi[k] - mean[k]) * inv[k];
            const float gb = bn_pre(zi[k], mean[k], inv[k], gam[k], bet[k]) > 0.f ? gi[k] : 0.f;     // the ReLU's mask, re-derived
            o[k] = on[k] ? gam[k] * inv[k] * (gb - mg[k] - zh * mgz[k]) : 0.f;
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    int r = r0;
    for (; r + BN_AROWS < rows_per_f; r += 2 * BN_AROWS) {
        const float4 z0 = zr[r * 13], g0 = gr[r * 13], z1 = zr[(r + BN_AROWS) * 13], g1 = gr[(r + BN_AROWS) * 13];
        gr[r * 13] = one(z0, g0);
        gr[(r + BN_AROWS) * 13] = one(z1, g1);
    }
    if (r < rows_per_f) gr[r * 13] = one(zr[r * 13], gr[r * 13]);
}

// ---- loss gradients ---------------------------------------------------------------------------------
// gY_j = (8 e_j + 6 s1) / (14 * n_b * nblocks) and the mask-sum gradient 2 (sum_j m_j - 1) / (n'_b * nblocks) (the same
// for the 4 targets) come out of the loss pass itself (loss.hip: k_loss_partial<true>).
struct BlockGeo { int F, T; int64_t cum; };

// g_p4 = (Re(conj(X) * gY0) + gM) * m * (1 - m), in place on gM.  grid (ceil(per-target reals/256), groups)
__global__ __launch_bounds__(256) void k_mask_bwd(const float2* __restrict__ X, const float2* __restrict__ gY0,
                                                   const float* __restrict__ masks, float* __restrict__ gM,
                                                   const TrainGroup* __restrict__ groups, TrainDims d) {
    const TrainGroup g = groups[blockIdx.y];
    const int64_t ST = (int64_t)d.S * g.T;
    const int64_t per = (int64_t)d.Bn * 2 * g.F * ST;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const int64_t yi = (int64_t)d.Bn * 8 * d.S * g.cum + (int64_t)g.tgt * per + i;
    const int64_t xi = (int64_t)d.Bn * 2 * d.S * g.cum + i;
    const float2 x = X[xi], gy = gY0[yi];
    const float m = masks[yi];
    gM[yi] = (x.x * gy.x + x.y * gy.y + gM[yi]) * m * (1.f - m);
}

// ---- layer 4 backward ---------------------------------------------------------------------------------
// bias: stage 1, grid (2B, groups): one (b, c) plane per workgroup -> double partial; stage 2 adds the B planes of a channel
__global__ __launch_bounds__(256) void k_l4_bias_partial(const float* __restrict__ gp4, const TrainGroup* __restrict__ groups,
                                                          TrainDims d, double* __restrict__ part) {
    const TrainGroup g = groups[blockIdx.y];
    const int64_t n = (int64_t)g.F * d.S * g.T;
    const float* p = gp4 + r8_idx(g, d, blockIdx.x >> 1, blockIdx.x & 1, 0, 0);
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += p[i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * 2 * d.Bn + blockIdx.x] = red[0];
}
__global__ void k_l4_bias_final(const double* __restrict__ part, const TrainGroup* __restrict__ groups, int ngroups, TrainDims d,
                                float* __restrict__ gpool) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * ngroups) return;
    const int gid = i >> 1, c = i & 1;
    double s = 0.0;
    for (int b = 0; b < d.Bn; ++b) s += part[(int64_t)gid * 2 * d.Bn + 2 * b + c];
    gpool[groups[gid].p_b4 + c] = (float)s;
}

// ---- weight gradients (wgrad.h) -----------------------------------------------------------------------------
// layers 2 / 3:  gw2[c2][c1,df,dt] = sum_k gz2[k][c2] * a1[(b, f2+df, t2+dt)][c1]
//                gw3[c2][c3,df,dt] = sum_k a2[k][c2]  * gz3[(b, f2+df, t2+dt)][c3]          k = (b, f2, t2)
// A = an act2-like array (rows k), B = 4 x 52 contiguous floats of an act1-like array per frequency tap df;
// one column tile per df, column = dt*52 + c.
struct WgL23Op {
    static constexpr int NTL = 224;
    struct Group { const float* A; const float* B; int F1, F2; };
    struct Row { const float* p; };
    struct Cols { int n; };
    const float* Aarr; const float* Barr; const TrainGroup* groups; TrainDims d;
    __device__ Group group(int gid) const {
        const TrainGroup g = groups[gid];
        return Group{Aarr + act2_off(g, d), Barr + act1_off(g, d), g.F1, g.F2};
    }
    __device__ const float* a_row(const Group& g, int k) const { return g.A + (int64_t)k * CS; }
    __device__ Row row(const Group& g, int k, int df) const {
        const int per = g.F2 * d.T2;
        const int b = k / per, r = k - b * per;
        const int f2 = r / d.T2, t2 = r - f2 * d.T2;
        return Row{g.B + ((int64_t)(b * g.F1 + f2 + df) * d.T1 + t2) * CS};
    }
    __device__ Cols cols(const Group&, int, int n) const { return Cols{n}; }
    __device__ float4 load_b4(const Group&, const Row& r, const Cols& c) const {
        return c.n < 4 * CS ? *reinterpret_cast<const float4*>(r.p + c.n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // consecutive rows k, k + 1, ... without a (b, f2, t2) split per row (wgrad_bf16p_kernel)
    struct RowIt { const float* p; int f2, t2; };
    __device__ RowIt row_it(const Group& g, int k, int df) const {
        const int per = g.F2 * d.T2;
        const int b = k / per, r = k - b * per;
        const int f2 = r / d.T2, t2 = r - f2 * d.T2;
        return RowIt{g.B + ((int64_t)(b * g.F1 + f2 + df) * d.T1 + t2) * CS, f2, t2};
    }
    __device__ Row row_of(const RowIt& it) const { return Row{it.p}; }
    __device__ void advance(const Group& g, RowIt& it) const {
        it.p += CS;
        if (++it.t2 == d.T2) {
            it.t2 = 0; it.p += (int64_t)(d.T1 - d.T2) * CS;
            if (++it.f2 == g.F2) { it.f2 = 0; it.p += (int64_t)(g.F1 - g.F2) * d.T1 * CS; }
        }
    }
};

// layers 1 / 4:  gw1[co][ci,df,dt] = sum_k gz1[k][co] * xin[b, ci, f1+df, t1*hop + dt - pad]
//                gw4[c3][c,df,dt]  = sum_k a3[k][c3]  * gp4[tgt, b, c, f1+df, t1*hop + dt]    k = (b, f1, t1)
// A = an act1-like array, B = (ci, df) spans of W samples of a real arena (2B channels, or the target's
// 2B channels of an 8B-channel arena); samples outside [0, S*T) are zero (causal padding / crop).
struct WgL14Op {
    static constexpr int NTL = 256;
    struct Group { const float* A; const float* R; int F, F1, T, hop, kf, K1, pad; int64_t ST; };
    struct Row { const float* p; int64_t tau0; };
    struct Cols { int64_t off; int dt; int valid; };
    const float* Aarr; const float* Rarr; const TrainGroup* groups; TrainDims d; int chan8, padleft;
    __device__ Group group(int gid) const {
        const TrainGroup g = groups[gid];
        Group o;
        o.ST = (int64_t)d.S * g.T;
        o.A = Aarr + act1_off(g, d);
        o.R = chan8 ? Rarr + (int64_t)d.Bn * 8 * d.S * g.cum + (int64_t)g.tgt * d.Bn * 2 * g.F * o.ST
                    : Rarr + (int64_t)d.Bn * 2 * d.S * g.cum;
        o.F = g.F; o.F1 = g.F1; o.T = g.T; o.hop = g.hop; o.kf = g.kf; o.K1 = 2 * g.kf * g.T;
        o.pad = (padleft && d.causal) ? g.T - 1 : 0;
        return o;
    }
    __device__ const float* a_row(const Group& g, int k) const { return g.A + (int64_t)k * CS; }
    __device__ Row row(const Group& g, int k, int) const {
        const int per = g.F1 * d.T1;
        const int b = k / per, r = k - b * per;
        const int f1 = r / d.T1, t1 = r - f1 * d.T1;
        const int64_t tau0 = (int64_t)t1 * g.hop - g.pad;
        return Row{g.R + ((int64_t)b * 2 * g.F + f1) * g.ST + tau0, tau0};
    }
    __device__ Cols cols(const Group& g, int ntile, int n) const {
        const int ng = ntile * NTL + n;
        Cols c; c.valid = ng < g.K1 && n < NTL; c.off = 0; c.dt = 0;
        if (c.valid) {
            const int seg = ng / g.T; c.dt = ng - seg * g.T;
            const int ci = seg / g.kf, df = seg - ci * g.kf;
            c.off = ((int64_t)ci * g.F + df) * g.ST + c.dt;
        }
        return c;
    }
    __device__ float4 load_b4(const Group& g, const Row& r, const Cols& c) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!c.valid) return v;
        const float* p = r.p + c.off;
        const int64_t t = r.tau0 + c.dt;
        if (t >= 0 && t + 3 < g.ST) { v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; return v; }
        if (t >= 0 && t < g.ST) v.x = p[0];
        if (t + 1 >= 0 && t + 1 < g.ST) v.y = p[1];
        if (t + 2 >= 0 && t + 2 < g.ST) v.z = p[2];
        if (t + 3 >= 0 && t + 3 < g.ST) v.w = p[3];
        return v;
    }
    // consecutive rows k, k + 1, ... without a (b, f1, t1) split per row (wgrad_bf16p_kernel)
    struct RowIt { const float* p; int64_t tau0; int f1, t1; };
    __device__ RowIt row_it(const Group& g, int k, int) const {
        const int per = g.F1 * d.T1;
        const int b = k / per, r = k - b * per;
        const int f1 = r / d.T1, t1 = r - f1 * d.T1;
        const int64_t tau0 = (int64_t)t1 * g.hop - g.pad;
        return RowIt{g.R + ((int64_t)b * 2 * g.F + f1) * g.ST + tau0, tau0, f1, t1};
    }
    __device__ Row row_of(const RowIt& it) const { return Row{it.p, it.tau0}; }
    __device__ void advance(const Group& g, RowIt& it) const {
        it.p += g.hop; it.tau0 += g.hop;
        if (++it.t1 == d.T1) {
            it.t1 = 0; it.p += g.ST - (int64_t)d.T1 * g.hop; it.tau0 = -g.pad;
            if (++it.f1 == g.F1) { it.f1 = 0; it.p += (int64_t)(2 * g.F - g.F1) * g.ST; }
        }
    }
};

// chunk sums -> canonical gradient layout.  grid (ceil(64 * kf_max * 208 / 256), groups)
__global__ __launch_bounds__(256) void k_wgrad_reduce23(const float* __restrict__ partial, const WgGroupInfo* __restrict__ info,
                                                         const TrainGroup* __restrict__ groups, int layer, float* __restrict__ gpool) {
    const TrainGroup g = groups[blockIdx.y];
    const int per = g.kf * 4 * CS;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t / per;
    if (m >= g.C2) return;
    const int r = t - m * per, df = r / (4 * CS), col = r - df * 4 * CS;
    const int dt = col / CS, cx = col - dt * CS;
    if (cx >= g.C1) return;
    const WgGroupInfo gi = info[blockIdx.y];
    const float* p = partial + ((int64_t)(gi.tile_base + df * gi.nch) * 64 + m) * WgL23Op::NTL + col;
    float acc = 0.f;
    for (int ch = 0; ch < gi.nch; ++ch) acc += p[(int64_t)ch * 64 * WgL23Op::NTL];
    gpool[(layer == 2 ? g.p_w2 : g.p_w3) + (((int64_t)m * g.C1 + cx) * g.kf + df) * 4 + dt] = acc;
}

// grid (ceil(50 * K1_max / 256), groups)
__global__ __launch_bounds__(256) void k_wgrad_reduce14(const float* __restrict__ partial, const WgGroupInfo* __restrict__ info,
                                                         const TrainGroup* __restrict__ groups, int layer, float* __restrict__ gpool) {
    const TrainGroup g = groups[blockIdx.y];
    const int K1 = 2 * g.kf * g.T;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.C1 * K1) return;
    const int m = t / K1, n = t - m * K1;
    const int ntile = n / WgL14Op::NTL, col = n - ntile * WgL14Op::NTL;
    const WgGroupInfo gi = info[blockIdx.y];
    const float* p = partial + ((int64_t)(gi.tile_base + ntile * gi.nch) * 64 + m) * WgL14Op::NTL + col;
    float acc = 0.f;
    for (int ch = 0; ch < gi.nch; ++ch) acc += p[(int64_t)ch * 64 * WgL14Op::NTL];
    gpool[(layer == 1 ? g.p_w1 : g.p_w4) + t] = acc;
}

// input whitening: xin = (|X| + mean_f) * scale_f.  gx8 holds the layer-1 data gradient per target in padded
// coordinates s = tau + pad (layer-4 operator, see cdae_api.h).
//   d mean_f = scale_f * sum g_xin,   d scale_f = sum g_xin * xin / scale_f      (sums over target, b, channel, tau)
// stage 1, grid (sum F, 4B): one (target, b) pair of planes per workgroup; stage 2 adds the 4B partials in order.
__global__ __launch_bounds__(256) void k_input_grad_partial(const float* __restrict__ xin, const float* __restrict__ gx8,
                                                             const TrainGroup* __restrict__ groups, const int2* __restrict__ rows,
                                                             TrainDims d, double* __restrict__ part) {
    const int2 r = rows[blockIdx.x];           // (first group of the block, f)
    const TrainGroup g = groups[r.x];
    const int f = r.y;
    const int To = d.T1 + 1;
    const int64_t ST = (int64_t)d.S * g.T, STp = (int64_t)To * g.hop;
    const int pad = d.causal ? g.T - 1 : 0;
    const int64_t s_end = pad + ST < STp ? pad + ST : STp;
    const float* gb = gx8 + 4 * (int64_t)d.Bn * To * g.cum;
    const int b = blockIdx.y % d.Bn;
    double sm = 0.0, ss = 0.0;
    for (int c = 0; c < 2; ++c) {
        const float* gr = gb + ((int64_t)(blockIdx.y * 2 + c) * g.F + f) * STp;       // blockIdx.y = tgt * Bn + b
        const float* xr = xin + r2_idx(g, d, b, c, f, 0) - pad;
        for (int64_t s = pad + threadIdx.x; s < s_end; s += 256) {
            const float gx = gr[s];
            sm += gx;
            ss += (double)gx * (double)xr[s];
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = sm; red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 2;
        o[0] = red[0][0]; o[1] = red[1][0];
    }
}
__global__ void k_input_grad_final(const double* __restrict__ part, const float* __restrict__ pool,
                                   const TrainGroup* __restrict__ groups, const int2* __restrict__ rows, int nrows, int nplanes,
                                   float* __restrict__ gpool) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int2 r = rows[i];
    const TrainGroup& g = groups[r.x];
    double sm = 0.0, ss = 0.0;
    for (int k = 0; k < nplanes; ++k) { sm += part[((int64_t)i * nplanes + k) * 2]; ss += part[((int64_t)i * nplanes + k) * 2 + 1]; }
    const float sc = pool[g.p_scale + r.y];
    gpool[g.p_mean + r.y] = (float)(sm * sc);
    gpool[g.p_scale + r.y] = (float)(ss / sc);
}

// ---- AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments) ----------------------
__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        const unsigned char* __restrict__ trainable, int64_t n, float lr, float wd, float b1, float b2,
                        float eps, float bc1, float bc2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !trainable[i]) return;
    float w = p[i] * (1.f - lr * wd);
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
}

}  // namespace xsq

// =====================================================================================================
struct xsq_train {
    xsq_model* model = nullptr;             // GEMM-layout forward (tables, tile caches); its pool is re-gathered every step
    int nblocks = 0, causal = 0, ngroups = 0;
    int64_t nparams = 0, pool_floats = 0, sumF = 0;
    std::vector<xsq::TrainGroup> groups;
    xsq::TrainGroup* d_groups = nullptr;
    xsq::BlockGeo* d_geo = nullptr;
    int2* d_rows = nullptr;                 // (first group, f) per whitening row
    float *d_params = nullptr, *d_grads = nullptr, *d_m = nullptr, *d_v = nullptr;
    unsigned char* d_trainable = nullptr;
    int *d_map_pool = nullptr, *d_map_mean = nullptr, *d_map_scale = nullptr;
    int* d_map_bwd = nullptr;               // canonical pool -> weights of the data-gradient operators (same slots as the forward pool)
    bool pools_valid = false;               // the two GEMM pools hold the current parameters (re-gathered behind every update)
    mutable std::mutex size_mu;
    mutable std::map<std::pair<int, int>, std::pair<size_t, size_t>> size_cache;   // (B, S) -> (reduction scratch doubles, weight-gradient partial floats)
    float* d_pool_bwd = nullptr;
    int64_t step = 0;
    std::vector<int32_t> Fv, Tv;
    struct WgTables {
        xsq::WgTile *d_t23 = nullptr, *d_t14 = nullptr; xsq::WgGroupInfo *d_i23 = nullptr, *d_i14 = nullptr; int n23 = 0, n14 = 0;
        xsq::BnTile *d_bt1 = nullptr, *d_bt2 = nullptr; xsq::BnInfo *d_bi1 = nullptr, *d_bi2 = nullptr; int nbt1 = 0, nbt2 = 0;   // act1-like / act2-like row tiles
    };
    int *d_frow1 = nullptr, *d_frow2 = nullptr;    // frequency-row index of the act1-like / act2-like arrays -> group
    std::mutex mu;
    std::map<std::pair<int, int>, WgTables> wg;     // (B, S) -> weight-gradient tile tables
    // The weight gradient of a layer and the data gradient that continues the chain both start from the same g and
    // are independent of each other; at B = 16 x S = 11 each is a 0.2 ms launch with a long tail, so the weight
    // gradients go to a side stream (forked / joined with events: capturable into a HIP graph).
    hipStream_t side = nullptr;
    hipEvent_t ev_fork[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join = nullptr;
    // loss read-back without stalling the stream (xsq_train_step with loss_out == NULL): the per-block terms of the
    // last LOSS_RING steps land in pinned host memory behind an event each; xsq_train_loss waits for one of them
    static const int LOSS_RING = 4;
    double* h_loss = nullptr;
    hipEvent_t ev_loss[LOSS_RING] = {nullptr, nullptr, nullptr, nullptr};
    int64_t seq = 0;                        // steps issued so far (the ticket of the next one)
};

using namespace xsq;

static const int WG_KC = 512;      // rows of one weight-gradient chunk

// tiles of the two weight-gradient launches: (group, column tile, row chunk); chunks of one (group, column tile) consecutive
static void wg_build(const xsq_train* Tr, int Bn, int S, std::vector<WgTile>* t23, std::vector<WgGroupInfo>* i23,
                     std::vector<WgTile>* t14, std::vector<WgGroupInfo>* i14) {
    const int T1 = Tr->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    for (int gi = 0; gi < Tr->ngroups; ++gi) {
        const TrainGroup& g = Tr->groups[gi];
        const int M2 = Bn * g.F2 * T2, M1 = Bn * g.F1 * T1;
        const int nch2 = (M2 + WG_KC - 1) / WG_KC, nch1 = (M1 + WG_KC - 1) / WG_KC;
        i23->push_back(WgGroupInfo{(int)t23->size(), nch2});
        for (int nt = 0; nt < g.kf; ++nt)
            for (int ch = 0; ch < nch2; ++ch) t23->push_back(WgTile{gi, nt, ch * WG_KC, std::min(M2, (ch + 1) * WG_KC)});
        i14->push_back(WgGroupInfo{(int)t14->size(), nch1});
        const int ntiles = (2 * g.kf * g.T + WgL14Op::NTL - 1) / WgL14Op::NTL;
        for (int nt = 0; nt < ntiles; ++nt)
            for (int ch = 0; ch < nch1; ++ch) t14->push_back(WgTile{gi, nt, ch * WG_KC, std::min(M1, (ch + 1) * WG_KC)});
    }
}

static void bn_build(const xsq_train* Tr, int Bn, int S, int kind, std::vector<BnTile>* t, std::vector<BnInfo>* info) {
    const int T1 = Tr->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    for (int gi = 0; gi < Tr->ngroups; ++gi) {
        const TrainGroup& g = Tr->groups[gi];
        const int M = kind == 0 ? Bn * g.F1 * T1 : Bn * g.F2 * T2;
        const int n = (M + BN_ROWS - 1) / BN_ROWS;
        info->push_back(BnInfo{(int)t->size(), n});
        for (int k = 0; k < n; ++k) t->push_back(BnTile{gi, k * BN_ROWS, std::min(M, (k + 1) * BN_ROWS)});
    }
}

// doubles of the reduction scratch: BatchNorm row tiles, bias planes, whitening planes
static size_t part_doubles(const xsq_train* Tr, int Bn, int S) {
    std::vector<BnTile> a, b; std::vector<BnInfo> ia, ib;
    bn_build(Tr, Bn, S, 0, &a, &ia); bn_build(Tr, Bn, S, 1, &b, &ib);
    return std::max({a.size() * 128, b.size() * 128, (size_t)Tr->ngroups * 2 * Bn, (size_t)Tr->sumF * 4 * Bn * 2});
}

static size_t wg_partial_floats(const xsq_train* Tr, int Bn, int S) {
    std::vector<WgTile> a, b; std::vector<WgGroupInfo> ia, ib;
    wg_build(Tr, Bn, S, &a, &ia, &b, &ib);
    return std::max(a.size() * 64 * WgL23Op::NTL, b.size() * 64 * WgL14Op::NTL);
}

// both sizes walk every group's tile list on the host (~0.1 ms): once per shape, not three times per step
static std::pair<size_t, size_t> train_sizes(const xsq_train* Tr, int Bn, int S) {
    std::lock_guard<std::mutex> lk(Tr->size_mu);
    auto it = Tr->size_cache.find({Bn, S});
    if (it != Tr->size_cache.end()) return it->second;
    const std::pair<size_t, size_t> v{part_doubles(Tr, Bn, S), wg_partial_floats(Tr, Bn, S)};
    Tr->size_cache[{Bn, S}] = v;
    return v;
}

static int wg_tables(xsq_train* Tr, int Bn, int S, xsq_train::WgTables* out) {
    std::lock_guard<std::mutex> lk(Tr->mu);
    auto it = Tr->wg.find({Bn, S});
    if (it != Tr->wg.end()) { *out = it->second; return XSQ_OK; }
    std::vector<WgTile> a, b; std::vector<WgGroupInfo> ia, ib;
    wg_build(Tr, Bn, S, &a, &ia, &b, &ib);
    xsq_train::WgTables w;
    w.n23 = (int)a.size(); w.n14 = (int)b.size();
#define UPW(dst, vec, TY)                                                                         \
    do {                                                                                          \
        XSQ_HIP(hipMalloc(&(dst), (vec).size() * sizeof(TY)));                                    \
        XSQ_HIP(hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(TY), hipMemcpyHostToDevice)); \
    } while (0)
    UPW(w.d_t23, a, WgTile); UPW(w.d_i23, ia, WgGroupInfo); UPW(w.d_t14, b, WgTile); UPW(w.d_i14, ib, WgGroupInfo);
    std::vector<BnTile> b1, b2; std::vector<BnInfo> j1, j2;
    bn_build(Tr, Bn, S, 0, &b1, &j1); bn_build(Tr, Bn, S, 1, &b2, &j2);
    w.nbt1 = (int)b1.size(); w.nbt2 = (int)b2.size();
    UPW(w.d_bt1, b1, BnTile); UPW(w.d_bi1, j1, BnInfo); UPW(w.d_bt2, b2, BnTile); UPW(w.d_bi2, j2, BnInfo);
#undef UPW
    Tr->wg[{Bn, S}] = w;
    *out = w;
    return XSQ_OK;
}

static inline size_t alt(size_t x) { return (x + 255) / 256 * 256; }
static int kf_of_t(int F) { return F < 10 ? 1 : (F < 20 ? 3 : 5); }

extern "C" {

int xsq_train_set_precision(xsq_train* T, int mode) {
    XSQ_REQUIRE(T && T->model, "xsq_train_set_precision: null handle");
    XSQ_REQUIRE(mode == 0 || mode == 1 || mode == 2, "xsq_train_set_precision: mode %d (0 = fp32, 1 = bf16, 2 = bf16x6)", mode);
    T->model->precision = mode == 1 ? 3 : mode;      // (3: plain bf16 operands, training only -- mode 1 of the MODEL is the split-bf16 inference format)
    return XSQ_OK;
}

int xsq_train_destroy(xsq_train* T) {
    if (!T) return XSQ_OK;
    if (T->model) xsq_model_destroy(T->model);
    (void)hipFree(T->d_groups); (void)hipFree(T->d_geo); (void)hipFree(T->d_rows); (void)hipFree(T->d_params);
    (void)hipFree(T->d_grads); (void)hipFree(T->d_m); (void)hipFree(T->d_v); (void)hipFree(T->d_trainable);
    (void)hipFree(T->d_map_pool); (void)hipFree(T->d_map_mean); (void)hipFree(T->d_map_scale);
    (void)hipFree(T->d_map_bwd); (void)hipFree(T->d_pool_bwd);
    for (auto& kv : T->wg) { (void)hipFree(kv.second.d_t23); (void)hipFree(kv.second.d_t14); (void)hipFree(kv.second.d_i23); (void)hipFree(kv.second.d_i14);
        (void)hipFree(kv.second.d_bt1); (void)hipFree(kv.second.d_bt2); (void)hipFree(kv.second.d_bi1); (void)hipFree(kv.second.d_bi2); }
    (void)hipFree(T->d_frow1); (void)hipFree(T->d_frow2);
    if (T->side) (void)hipStreamDestroy(T->side);
    for (hipEvent_t e : T->ev_fork) if (e) (void)hipEventDestroy(e);
    if (T->ev_join) (void)hipEventDestroy(T->ev_join);
    for (hipEvent_t e : T->ev_loss) if (e) (void)hipEventDestroy(e);
    if (T->h_loss) (void)hipHostFree(T->h_loss);
    delete T;
    return XSQ_OK;
}

static int train_build(xsq_train* Tr, int nblocks, const int32_t* F, const int32_t* T, int causal, const float* params,
                       int64_t nparams) {
    int rc = xsq_model_create(&Tr->model, nblocks, F, T, causal, params, nparams);
    if (rc) return rc;
    xsq_model* Mo = Tr->model;
    Tr->nblocks = nblocks; Tr->causal = causal ? 1 : 0; Tr->nparams = nparams;
    Tr->Fv.assign(F, F + nblocks); Tr->Tv.assign(T, T + nblocks);
    // walk the canonical order once more: offsets of every tensor, trainable mask, gather maps
    std::vector<unsigned char> trainable((size_t)nparams, 1);
    std::vector<BlockGeo> geo;
    std::vector<int2> rows;
    std::vector<int> map_mean, map_scale;
    int64_t p = 0;
    for (int b = 0; b < nblocks; ++b) {
        const CdaeBlockDev& cb = Mo->blocks[b];
        const int kf = cb.kf, W = cb.T;
        geo.push_back(BlockGeo{cb.F, cb.T, cb.cum});
        const int64_t p_mean = p, p_scale = p + cb.F;
        for (int f = 0; f < cb.F; ++f) { map_mean.push_back((int)(p_mean + f)); map_scale.push_back((int)(p_scale + f)); }
        for (int f = 0; f < cb.F; ++f) rows.push_back(make_int2((int)Tr->groups.size(), f));
        p += 2 * cb.F;
        for (int t = 0; t < NT; ++t) {
            TrainGroup g;
            memset(&g, 0, sizeof(g));
            g.block = b; g.tgt = t; g.F = cb.F; g.T = cb.T; g.hop = cb.hop; g.kf = kf; g.F1 = cb.F1; g.F2 = cb.F2;
            g.cumF1 = cb.cumF1; g.cumF2 = cb.cumF2; g.C1 = H1; g.C2 = H2; g.cum = cb.cum; g.cumF = cb.cumF;
            g.p_mean = p_mean; g.p_scale = p_scale;
            g.p_w1 = p; p += (int64_t)H1 * 2 * kf * W;
            g.p_bn1 = p; for (int64_t i = p + 2 * H1; i < p + 4 * H1; ++i) trainable[i] = 0; p += 4 * H1;
            g.p_w2 = p; p += (int64_t)H2 * H1 * kf * 4;
            g.p_bn2 = p; for (int64_t i = p + 2 * H2; i < p + 4 * H2; ++i) trainable[i] = 0; p += 4 * H2;
            g.p_w3 = p; p += (int64_t)H2 * H1 * kf * 4;
            g.p_bn3 = p; for (int64_t i = p + 2 * H1; i < p + 4 * H1; ++i) trainable[i] = 0; p += 4 * H1;
            g.p_w4 = p; p += (int64_t)H1 * 2 * kf * W;
            g.p_b4 = p; p += 2;
            Tr->groups.push_back(g);
        }
    }
    XSQ_REQUIRE(p == nparams, "xsq_train_create: parameter walk mismatch");
    Tr->ngroups = (int)Tr->groups.size();
    Tr->sumF = Mo->sumF;
    // GEMM-pool gather map: the same placement as xsq_model_create, without the BatchNorm fold
    const int64_t extent = Mo->pool_floats;
    Tr->pool_floats = extent;
    std::vector<int> map((size_t)extent, -1);
    for (int b = 0; b < nblocks; ++b) {
        const CdaeBlockDev& cb = Mo->blocks[b];
        const int kf = cb.kf, W = cb.T, hop = cb.hop;
        const int K1 = 2 * kf * W, K2 = kf * 4 * CS;
        for (int t = 0; t < NT; ++t) {
            const TrainGroup& g = Tr->groups[b * 4 + t];
            for (int co = 0; co < H1; ++co)
                for (int k = 0; k < K1; ++k) map[cb.w1[t] + (int64_t)co * cb.ld1 + k] = (int)(g.p_w1 + (int64_t)co * K1 + k);
            for (int co = 0; co < H2; ++co)
                for (int ci = 0; ci < H1; ++ci)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            map[cb.w2[t] + (int64_t)co * K2 + (df * 4 + dt) * CS + ci] =
                                (int)(g.p_w2 + (((int64_t)co * H1 + ci) * kf + df) * 4 + dt);
            for (int co = 0; co < H1; ++co)
                for (int ci = 0; ci < H2; ++ci)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            map[cb.w3[t] + (int64_t)co * K2 + (df * 4 + (3 - dt)) * CS + ci] =
                                (int)(g.p_w3 + (((int64_t)ci * H1 + co) * kf + df) * 4 + dt);
            for (int ci = 0; ci < H1; ++ci)
                for (int c = 0; c < 2; ++c)
                    for (int df = 0; df < kf; ++df)
                        for (int tap = 0; tap < 2; ++tap)
                            for (int dt = 0; dt < hop; ++dt)
                                map[cb.w4[t] + (int64_t)(c * hop + dt) * cb.ld4 + (df * 2 + (1 - tap)) * CS + ci] =
                                    (int)(g.p_w4 + (((int64_t)ci * 2 + c) * kf + df) * W + dt + tap * hop);
            map[cb.b4[t]] = (int)g.p_b4;
            map[cb.b4[t] + 1] = (int)(g.p_b4 + 1);
        }
    }
    // data-gradient operators reuse the forward GEMM operators on the gradient arenas (cdae_api.h):
    //   slot w1 <- w4 (layer-1 operator = d/d a3 of layer 4)      slot w2 <- w3 (layer-2 operator = d/d a2 of layer 3)
    //   slot w3 <- w2 (layer-3 operator = d/d a1 of layer 2)      slot w4 <- w1 (layer-4 operator = d/d xin of layer 1)
    std::vector<int> mapb((size_t)extent, -1);
    for (int b = 0; b < nblocks; ++b) {
        const CdaeBlockDev& cb = Mo->blocks[b];
        const int kf = cb.kf, W = cb.T, hop = cb.hop;
        const int K1 = 2 * kf * W, K2 = kf * 4 * CS;
        for (int t = 0; t < NT; ++t) {
            const TrainGroup& g = Tr->groups[b * 4 + t];
            for (int co = 0; co < H1; ++co)
                for (int k = 0; k < K1; ++k) mapb[cb.w1[t] + (int64_t)co * cb.ld1 + k] = (int)(g.p_w4 + (int64_t)co * K1 + k);
            for (int c2 = 0; c2 < H2; ++c2)
                for (int c3 = 0; c3 < H1; ++c3)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            mapb[cb.w2[t] + (int64_t)c2 * K2 + (df * 4 + dt) * CS + c3] =
                                (int)(g.p_w3 + (((int64_t)c2 * H1 + c3) * kf + df) * 4 + dt);
            for (int c1 = 0; c1 < H1; ++c1)
                for (int c2 = 0; c2 < H2; ++c2)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            mapb[cb.w3[t] + (int64_t)c1 * K2 + (df * 4 + (3 - dt)) * CS + c2] =
                                (int)(g.p_w2 + (((int64_t)c2 * H1 + c1) * kf + df) * 4 + dt);
            for (int co = 0; co < H1; ++co)
                for (int c = 0; c < 2; ++c)
                    for (int df = 0; df < kf; ++df)
                        for (int tap = 0; tap < 2; ++tap)
                            for (int dt = 0; dt < hop; ++dt)
                                mapb[cb.w4[t] + (int64_t)(c * hop + dt) * cb.ld4 + (df * 2 + (1 - tap)) * CS + co] =
                                    (int)(g.p_w1 + (((int64_t)co * 2 + c) * kf + df) * W + dt + tap * hop);
        }
    }
#define UPV(dst, vec, TY)                                                                         \
    do {                                                                                          \
        XSQ_HIP(hipMalloc(&(dst), (vec).size() * sizeof(TY)));                                    \
        XSQ_HIP(hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(TY), hipMemcpyHostToDevice)); \
    } while (0)
    UPV(Tr->d_groups, Tr->groups, TrainGroup);
    UPV(Tr->d_geo, geo, BlockGeo);
    UPV(Tr->d_rows, rows, int2);
    UPV(Tr->d_trainable, trainable, unsigned char);
    UPV(Tr->d_map_pool, map, int);
    UPV(Tr->d_map_mean, map_mean, int);
    UPV(Tr->d_map_scale, map_scale, int);
    UPV(Tr->d_map_bwd, mapb, int);
    std::vector<int> frow1((size_t)4 * Mo->sumF1), frow2((size_t)4 * Mo->sumF2);
    for (int gi = 0; gi < Tr->ngroups; ++gi) {
        const TrainGroup& g = Tr->groups[gi];
        for (int f = 0; f < g.F1; ++f) frow1[4 * g.cumF1 + g.tgt * g.F1 + f] = gi;
        for (int f = 0; f < g.F2; ++f) frow2[4 * g.cumF2 + g.tgt * g.F2 + f] = gi;
    }
    UPV(Tr->d_frow1, frow1, int);
    UPV(Tr->d_frow2, frow2, int);
    XSQ_HIP(hipMalloc(&Tr->d_pool_bwd, (size_t)extent * 4));
#undef UPV
    XSQ_HIP(hipMalloc(&Tr->d_params, (size_t)nparams * 4));
    XSQ_HIP(hipMemcpy(Tr->d_params, params, (size_t)nparams * 4, hipMemcpyHostToDevice));
    XSQ_HIP(hipMalloc(&Tr->d_grads, (size_t)nparams * 4));
    XSQ_HIP(hipMalloc(&Tr->d_m, (size_t)nparams * 4));
    XSQ_HIP(hipMalloc(&Tr->d_v, (size_t)nparams * 4));
    XSQ_HIP(hipMemset(Tr->d_grads, 0, (size_t)nparams * 4));
    XSQ_HIP(hipMemset(Tr->d_m, 0, (size_t)nparams * 4));
    XSQ_HIP(hipMemset(Tr->d_v, 0, (size_t)nparams * 4));
    return XSQ_OK;
}

int xsq_train_create(xsq_train** out, int nblocks, const int32_t* F, const int32_t* T, int causal, const float* params,
                     int64_t nparams) {
    XSQ_REQUIRE(out && F && T && params && nblocks > 0, "xsq_train_create: null argument");
    // the step runs the Wiener-EM from the masks (xsq_wiener_em_masked: two frames per thread), which needs an even frame
    // count S * T_b in every block for any S: refuse other plans here, with the reason, instead of failing a step later
    // (every plan nsgfwin builds has T_b = 4 k, nsgt/nsgfwin_sl.py:70-72)
    for (int b = 0; b < nblocks; ++b)
        XSQ_REQUIRE(T[b] % 2 == 0, "xsq_train_create: block %d has an odd T = %d; the training step needs even band lengths "
                    "(Wiener-EM from the masks processes two frames per thread)", b, T[b]);
    xsq_train* Tr = new xsq_train();
    int rc = train_build(Tr, nblocks, F, T, causal, params, nparams);
    if (!rc && hipStreamCreateWithFlags(&Tr->side, hipStreamNonBlocking) != hipSuccess) rc = XSQ_ERR_HIP;
    for (hipEvent_t& e : Tr->ev_fork) if (!rc && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = XSQ_ERR_HIP;
    if (!rc && hipEventCreateWithFlags(&Tr->ev_join, hipEventDisableTiming) != hipSuccess) rc = XSQ_ERR_HIP;
    for (hipEvent_t& e : Tr->ev_loss) if (!rc && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = XSQ_ERR_HIP;
    if (!rc && hipHostMalloc((void**)&Tr->h_loss, (size_t)xsq_train::LOSS_RING * nblocks * 16, hipHostMallocDefault) != hipSuccess) rc = XSQ_ERR_HIP;
    if (rc == XSQ_ERR_HIP) set_error("xsq_train_create: side stream / events / pinned loss buffer");
    if (rc) { xsq_train_destroy(Tr); return rc; }
    *out = Tr;
    return XSQ_OK;
}

static float* train_pool(xsq_train* Tr, int what) {
    return what == 0 ? Tr->d_params : what == 1 ? Tr->d_grads : what == 2 ? Tr->d_m : Tr->d_v;
}

int xsq_train_read(xsq_train* Tr, int what, float* host_out) {
    XSQ_REQUIRE(Tr && host_out && what >= 0 && what <= 3, "xsq_train_read: bad argument (what = 0..3)");
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpy(host_out, train_pool(Tr, what), (size_t)Tr->nparams * 4, hipMemcpyDeviceToHost));
    return XSQ_OK;
}

// Restores a pool from the host (resuming from a checkpoint: parameters + both AdamW moments).  Gradients are
// an output of the step and cannot be written.
int xsq_train_write(xsq_train* Tr, int what, const float* host_in) {
    XSQ_REQUIRE(Tr && host_in && (what == 0 || what == 2 || what == 3), "xsq_train_write: bad argument (what = 0, 2 or 3)");
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpy(train_pool(Tr, what), host_in, (size_t)Tr->nparams * 4, hipMemcpyHostToDevice));
    if (what == 0) Tr->pools_valid = false;        // the GEMM pools are re-gathered by the next step
    return XSQ_OK;
}

// AdamW step counter (bias correction 1 - beta^step); set >= 0 to restore it, -1 to only read.  Returns the counter.
int64_t xsq_train_step_count(xsq_train* Tr, int64_t set) {
    if (!Tr) return XSQ_ERR_ARG;
    if (set >= 0) Tr->step = set;
    return Tr->step;
}

// workspace (floats): xin | z1 a1 g1 | z2 a2 g2 | z3 a3 g3 | masks gM | Y gY [Y0] | bn stats | mean scale | loss
size_t xsq_train_workspace(const xsq_train* Tr, int Bn, int S, int wiener) {
    if (!Tr || Bn <= 0 || S < 3) return 0;
    const xsq_model* Mo = Tr->model;
    const int64_t T1 = Tr->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    const size_t n2 = (size_t)Bn * 2 * S * Mo->sumFT, n8 = 4 * n2;
    const size_t a1 = (size_t)CS * Bn * T1 * 4 * Mo->sumF1, a2 = (size_t)CS * Bn * T2 * 4 * Mo->sumF2;
    size_t b = alt(n2 * 4) + 6 * alt(a1 * 4) + 3 * alt(a2 * 4) + 2 * alt(n8 * 4) + 2 * alt(n8 * 8);
    const auto sz = train_sizes(Tr, Bn, S);
    b += alt((size_t)Tr->ngroups * 3 * 256 * 4) + 2 * alt((size_t)Tr->sumF * 4) + alt(sz.first * 8) + alt(sz.second * 4);
    b += xsq_loss_workspace(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), Bn, S) + alt((size_t)Tr->nblocks * 16) + 4096;
    if (wiener) b += 2 * alt(xsq_wiener_workspace(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), Bn, S, 5000)) + 4096;
    return b;
}

// One step.  X: mix arena (2B ch), Yt: target arena (8B ch).  apply_update = 0 leaves the parameters untouched
// (gradients stay readable through xsq_train_read).  loss_out: HOST double[2] (complex MSE, mask sum).
int xsq_train_step(xsq_train* Tr, const float* X, const float* Yt, int Bn, int S, int wiener, float lr, float wd,
                   int apply_update, double* loss_out, void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(Tr && X && Yt && ws, "xsq_train_step: null argument");
    XSQ_REQUIRE(Bn > 0 && S >= 3, "xsq_train_step: B=%d S=%d", Bn, S);
    XSQ_REQUIRE(ws_bytes >= xsq_train_workspace(Tr, Bn, S, wiener), "xsq_train_step: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    xsq_model* Mo = Tr->model;
    const int T1 = Tr->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    const TrainDims d{Bn, S, T1, T2, Tr->causal};
    const size_t n2 = (size_t)Bn * 2 * S * Mo->sumFT, n8 = 4 * n2;
    const size_t na1 = (size_t)CS * Bn * T1 * 4 * Mo->sumF1, na2 = (size_t)CS * Bn * T2 * 4 * Mo->sumF2;
    char* w = (char*)ws;
    auto take = [&](size_t bytes) { void* p = w; w += alt(bytes); return p; };
    float* xin = (float*)take(n2 * 4);
    float *z1 = (float*)take(na1 * 4), *a1 = (float*)take(na1 * 4), *g1 = (float*)take(na1 * 4);
    float *z2 = (float*)take(na2 * 4), *a2 = (float*)take(na2 * 4), *g2 = (float*)take(na2 * 4);
    float *z3 = (float*)take(na1 * 4), *a3 = (float*)take(na1 * 4), *g3 = (float*)take(na1 * 4);
    float *masks = (float*)take(n8 * 4), *gM = (float*)take(n8 * 4);
    float *Y = (float*)take(n8 * 8), *gY = (float*)take(n8 * 8);
    const size_t wst_bytes = wiener ? xsq_wiener_workspace(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), Bn, S, 5000) : 0;
    void* wst = wiener ? take(wst_bytes) : nullptr;
    void* wbst = wiener ? take(wst_bytes) : nullptr;
    float* stats = (float*)take((size_t)Tr->ngroups * 3 * 256 * 4);
    double* part = (double*)take(train_sizes(Tr, Bn, S).first * 8);
    xsq_train::WgTables wt;
    if (int rcw = wg_tables(Tr, Bn, S, &wt)) return rcw;
    float* wpart = (float*)take(std::max((size_t)wt.n23 * 64 * WgL23Op::NTL, (size_t)wt.n14 * 64 * WgL14Op::NTL) * 4);
    float *mean = (float*)take((size_t)Tr->sumF * 4), *scale = (float*)take((size_t)Tr->sumF * 4);
    double* d_loss = (double*)take((size_t)Tr->nblocks * 16);
    void* loss_ws = w;

    auto grid1 = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // ---- parameters -> GEMM layouts -------------------------------------------------------------
    auto gather_pools = [&]() {
        { XSQ_PROF("train_gather", stream); hipLaunchKernelGGL(k_gather, grid1(Tr->pool_floats), dim3(256), 0, stream, Tr->d_params, Tr->d_map_pool, Mo->d_pool, Tr->pool_floats); }
        { XSQ_PROF("train_gather", stream); hipLaunchKernelGGL(k_gather, grid1(Tr->pool_floats), dim3(256), 0, stream, Tr->d_params, Tr->d_map_bwd, Tr->d_pool_bwd, Tr->pool_floats); }
        Tr->pools_valid = true;
    };
    // the weights in GEMM layout: gathered here only on the first step or after the parameters were written from outside --
    // otherwise the gather of step k + 1 was issued BEHIND the update of step k (below), where it runs while the host looks at
    // the loss and prepares the next batch (the device idles there: profiles/r07h_timeline_train.txt) instead of in front
    // of the first GEMM
    if (!Tr->pools_valid) gather_pools();
    { XSQ_PROF("train_gather", stream); hipLaunchKernelGGL(k_gather, grid1(Tr->sumF), dim3(256), 0, stream, Tr->d_params, Tr->d_map_mean, mean, Tr->sumF); }
    { XSQ_PROF("train_gather", stream); hipLaunchKernelGGL(k_gather, grid1(Tr->sumF), dim3(256), 0, stream, Tr->d_params, Tr->d_map_scale, scale, Tr->sumF); }
    // ---- forward ----------------------------------------------------------------------------------
    int rc;
    cdae_launch_magnitude(Mo, X, xin, mean, scale, Bn, S, stream);
    int64_t maxM1 = 0, maxM2 = 0, maxP = 0, maxW14 = 0, maxR23 = 0;
    for (const TrainGroup& g : Tr->groups) {
        maxM1 = std::max<int64_t>(maxM1, (int64_t)Bn * g.F1 * T1 * CS);
        maxM2 = std::max<int64_t>(maxM2, (int64_t)Bn * g.F2 * T2 * CS);
        maxP = std::max<int64_t>(maxP, (int64_t)Bn * 2 * g.F * S * g.T);
        maxW14 = std::max<int64_t>(maxW14, (int64_t)H1 * 2 * g.kf * g.T);
        maxR23 = std::max<int64_t>(maxR23, (int64_t)H2 * g.kf * 4 * CS);
    }
    const unsigned G = (unsigned)Tr->ngroups;
    const int64_t nq1 = (int64_t)na1 / 4, nq2 = (int64_t)na2 / 4;      // float4s of the act1-like / act2-like arrays
    CdaeArgs a{Mo->d_blocks, Mo->d_pool, xin, z1, z2, z3, X, Y, masks, Bn, S, T1, T2, Tr->causal, 1, nullptr, nullptr};
    if ((rc = cdae_launch_layer(Mo, 1, a, stream))) return rc;                       // z1
    { XSQ_PROF("train_bn_stats", stream); hipLaunchKernelGGL(k_bn_stats_partial, dim3(wt.nbt1), dim3(256), 0, stream, z1, Tr->d_groups, wt.d_bt1, d, 0, part);
      hipLaunchKernelGGL(k_bn_stats_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi1, d, 0, stats, Tr->d_params, apply_update); }
    { XSQ_PROF("train_bn_relu_apply", stream); hipLaunchKernelGGL(k_bn_relu_apply, dim3((unsigned)(nq1 / ((int64_t)(Bn * T1) * 13))), dim3(256), 0, stream, (const float4*)z1, (float4*)a1, Tr->d_groups, Tr->d_frow1, Bn * T1, nq1, 0, stats, Tr->d_params); }
    a.act1 = a1; a.act2 = z2;
    if ((rc = cdae_launch_layer(Mo, 2, a, stream))) return rc;                       // z2 from a1
    { XSQ_PROF("train_bn_stats", stream); hipLaunchKernelGGL(k_bn_stats_partial, dim3(wt.nbt2), dim3(256), 0, stream, z2, Tr->d_groups, wt.d_bt2, d, 1, part);
      hipLaunchKernelGGL(k_bn_stats_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi2, d, 1, stats, Tr->d_params, apply_update); }
    { XSQ_PROF("train_bn_relu_apply", stream); hipLaunchKernelGGL(k_bn_relu_apply, dim3((unsigned)(nq2 / ((int64_t)(Bn * T2) * 13))), dim3(256), 0, stream, (const float4*)z2, (float4*)a2, Tr->d_groups, Tr->d_frow2, Bn * T2, nq2, 1, stats, Tr->d_params); }
    a.act2 = a2; a.act3 = z3;
    if ((rc = cdae_launch_layer(Mo, 3, a, stream))) return rc;                       // z3 from a2
    { XSQ_PROF("train_bn_stats", stream); hipLaunchKernelGGL(k_bn_stats_partial, dim3(wt.nbt1), dim3(256), 0, stream, z3, Tr->d_groups, wt.d_bt1, d, 2, part);
      hipLaunchKernelGGL(k_bn_stats_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi1, d, 2, stats, Tr->d_params, apply_update); }
    { XSQ_PROF("train_bn_relu_apply", stream); hipLaunchKernelGGL(k_bn_relu_apply, dim3((unsigned)(nq1 / ((int64_t)(Bn * T1) * 13))), dim3(256), 0, stream, (const float4*)z3, (float4*)a3, Tr->d_groups, Tr->d_frow1, Bn * T1, nq1, 2, stats, Tr->d_params); }
    a.act3 = a3;
    // model.py:264-268: the offline model filters the mix-phase estimate (phase.py:18-69).  Layer 4 then stores the
    // masks only; the EM passes -- forward and backward -- form mask * X while they load (no estimate arena, no copy of it).
    if (wiener) a.Y = nullptr;
    if ((rc = cdae_launch_layer(Mo, 4, a, stream))) return rc;                       // masks [, Y = mask * X]
    if (wiener && (rc = xsq_wiener_em_masked(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), X, masks, Y, Bn, S, 5000, Bn, wst, wst_bytes, stream)))
        return rc;
    // ---- loss + its gradients (one pass) ------------------------------------------------------------
    if ((rc = loss_forward_backward(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), Y, Yt, masks, Bn, S, d_loss, gY, gM, loss_ws, stream))) return rc;
    // (Wiener-EM: the last pass of its backward forms the mask gradient itself -- no gradient arena written and re-read)
    if (wiener && (rc = wiener_em_backward(Tr->nblocks, Tr->Fv.data(), Tr->Tv.data(), X, nullptr, masks, gY, Bn, S, 5000, Bn, wst, wbst, stream, gM)))
        return rc;
    if (!wiener) { XSQ_PROF("train_mask_bwd", stream); hipLaunchKernelGGL(k_mask_bwd, dim3(grid1(maxP).x, G), dim3(256), 0, stream, (const float2*)X, (const float2*)gY, masks, gM, Tr->d_groups, d); }
    // ---- backward -----------------------------------------------------------------------------------
    float* gp = Tr->d_grads;
    { XSQ_PROF("train_l4_bias_grad", stream); hipLaunchKernelGGL(k_l4_bias_partial, dim3(2 * Bn, G), dim3(256), 0, stream, gM, Tr->d_groups, d, part);
      hipLaunchKernelGGL(k_l4_bias_final, grid1(2 * Tr->ngroups), dim3(256), 0, stream, part, Tr->d_groups, Tr->ngroups, d, gp); }
    // weight gradients: on the side stream, each behind the kernel that finishes its g (XSQ_TRAIN_SIDE=0: in line)
    static const bool use_side = !(getenv("XSQ_TRAIN_SIDE") && atoi(getenv("XSQ_TRAIN_SIDE")) == 0);
    hipStream_t ws_ = use_side ? Tr->side : stream;
    auto fork = [&](int i) {
        if (!use_side) return hipSuccess;
        hipError_t e = hipEventRecord(Tr->ev_fork[i], stream);
        return e != hipSuccess ? e : hipStreamWaitEvent(Tr->side, Tr->ev_fork[i], 0);
    };
    // bf16 arm: the weight gradients on bf16 operands as well (XSQ_TRAIN_WGRAD_FP32=1: keep them on the fp32 pipe, an A/B arm)
    static const bool wgrad_fp32 = getenv("XSQ_TRAIN_WGRAD_FP32") && atoi(getenv("XSQ_TRAIN_WGRAD_FP32")) != 0;
    const bool bf16w = Mo->precision == 3 && !wgrad_fp32;
    // XSQ_TRAIN_WGRAD_PACKED=1: the bf16 weight gradients on wgrad_bf16p_kernel (operands rounded once at staging time, k-pair-packed
    // in LDS, one 16-byte fragment read per MFMA operand) -- an A/B arm, bitwise the default's gradients and measured SLOWER
    // (0.30-0.34 ms per launch against 0.24, step 3.19 against 3.02 on one box: its 8-byte staging writes land 16 lanes on four
    // banks and it holds four rows of loads per thread; profiles/r11_ab_runs.txt r11wg).  Default: round 5's kernel.
    const char* wp_ = getenv("XSQ_TRAIN_WGRAD_PACKED");          // (read per step: tests/test_training.py flips it inside one process)
    const bool wgrad_packed = wp_ && atoi(wp_) != 0;
    XSQ_HIP(fork(0));
    { XSQ_PROF("train_l4_wgrad", ws_);
      if (bf16w && wgrad_packed) hipLaunchKernelGGL((wgrad_bf16p_kernel<WgL14Op>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{a3, gM, Tr->d_groups, d, 1, 0}, wt.d_t14, wpart);
      else if (bf16w) hipLaunchKernelGGL((wgrad_kernel<WgL14Op, true>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{a3, gM, Tr->d_groups, d, 1, 0}, wt.d_t14, wpart);
      else hipLaunchKernelGGL((wgrad_kernel<WgL14Op>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{a3, gM, Tr->d_groups, d, 1, 0}, wt.d_t14, wpart);
      hipLaunchKernelGGL(k_wgrad_reduce14, dim3(grid1(maxW14).x, G), dim3(256), 0, ws_, wpart, wt.d_i14, Tr->d_groups, 4, gp); }
    CdaeArgs bw{Mo->d_blocks, Tr->d_pool_bwd, xin, g1, g2, g3, X, Y, nullptr, Bn, S, T1, T2, Tr->causal, 1, nullptr, nullptr};
    bw.xin8 = gM; bw.act1 = g3;                                                    // g_a3 <- g_p4   (layer-1 operator)
    if ((rc = cdae_launch_layer(Mo, 1, bw, stream, "train_l4_dgrad_gemm"))) return rc;
    bw.xin8 = nullptr;
    { XSQ_PROF("train_bn_bwd_reduce", stream); hipLaunchKernelGGL(k_bn_bwd_partial, dim3(wt.nbt1), dim3(256), 0, stream, z3, g3, Tr->d_groups, wt.d_bt1, d, 2, stats, Tr->d_params, part);
      hipLaunchKernelGGL(k_bn_bwd_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi1, d, 2, stats, gp); }
    { XSQ_PROF("train_bn_bwd_apply", stream); hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)(nq1 / ((int64_t)(Bn * T1) * 13))), dim3(256), 0, stream, (const float4*)z3, (float4*)g3, Tr->d_groups, Tr->d_frow1, Bn * T1, nq1, 2, stats, Tr->d_params); }
    XSQ_HIP(fork(1));
    { XSQ_PROF("train_l3_wgrad", ws_);
      if (bf16w && wgrad_packed) hipLaunchKernelGGL((wgrad_bf16p_kernel<WgL23Op>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{a2, g3, Tr->d_groups, d}, wt.d_t23, wpart);
      else if (bf16w) hipLaunchKernelGGL((wgrad_kernel<WgL23Op, true>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{a2, g3, Tr->d_groups, d}, wt.d_t23, wpart);
      else hipLaunchKernelGGL((wgrad_kernel<WgL23Op>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{a2, g3, Tr->d_groups, d}, wt.d_t23, wpart);
      hipLaunchKernelGGL(k_wgrad_reduce23, dim3(grid1(maxR23).x, G), dim3(256), 0, ws_, wpart, wt.d_i23, Tr->d_groups, 3, gp); }
    bw.act1 = g3; bw.act2 = g2;                                                    // g_a2 <- g_z3   (layer-2 operator)
    if ((rc = cdae_launch_layer(Mo, 2, bw, stream, "train_l3_dgrad_gemm"))) return rc;
    { XSQ_PROF("train_bn_bwd_reduce", stream); hipLaunchKernelGGL(k_bn_bwd_partial, dim3(wt.nbt2), dim3(256), 0, stream, z2, g2, Tr->d_groups, wt.d_bt2, d, 1, stats, Tr->d_params, part);
      hipLaunchKernelGGL(k_bn_bwd_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi2, d, 1, stats, gp); }
    { XSQ_PROF("train_bn_bwd_apply", stream); hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)(nq2 / ((int64_t)(Bn * T2) * 13))), dim3(256), 0, stream, (const float4*)z2, (float4*)g2, Tr->d_groups, Tr->d_frow2, Bn * T2, nq2, 1, stats, Tr->d_params); }
    XSQ_HIP(fork(2));
    { XSQ_PROF("train_l2_wgrad", ws_);
      if (bf16w && wgrad_packed) hipLaunchKernelGGL((wgrad_bf16p_kernel<WgL23Op>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{g2, a1, Tr->d_groups, d}, wt.d_t23, wpart);
      else if (bf16w) hipLaunchKernelGGL((wgrad_kernel<WgL23Op, true>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{g2, a1, Tr->d_groups, d}, wt.d_t23, wpart);
      else hipLaunchKernelGGL((wgrad_kernel<WgL23Op>), dim3(wt.n23), dim3(256), 0, ws_, WgL23Op{g2, a1, Tr->d_groups, d}, wt.d_t23, wpart);
      hipLaunchKernelGGL(k_wgrad_reduce23, dim3(grid1(maxR23).x, G), dim3(256), 0, ws_, wpart, wt.d_i23, Tr->d_groups, 2, gp); }
    bw.act2 = g2; bw.act3 = g1;                                                    // g_a1 <- g_z2   (layer-3 operator)
    if ((rc = cdae_launch_layer(Mo, 3, bw, stream, "train_l2_dgrad_gemm"))) return rc;
    { XSQ_PROF("train_bn_bwd_reduce", stream); hipLaunchKernelGGL(k_bn_bwd_partial, dim3(wt.nbt1), dim3(256), 0, stream, z1, g1, Tr->d_groups, wt.d_bt1, d, 0, stats, Tr->d_params, part);
      hipLaunchKernelGGL(k_bn_bwd_final, dim3(G), dim3(64 * BN_FL), 0, stream, part, Tr->d_groups, wt.d_bi1, d, 0, stats, gp); }
    { XSQ_PROF("train_bn_bwd_apply", stream); hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)(nq1 / ((int64_t)(Bn * T1) * 13))), dim3(256), 0, stream, (const float4*)z1, (float4*)g1, Tr->d_groups, Tr->d_frow1, Bn * T1, nq1, 0, stats, Tr->d_params); }
    XSQ_HIP(fork(3));
    { XSQ_PROF("train_l1_wgrad", ws_);
      if (bf16w && wgrad_packed) hipLaunchKernelGGL((wgrad_bf16p_kernel<WgL14Op>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{g1, xin, Tr->d_groups, d, 0, 1}, wt.d_t14, wpart);
      else if (bf16w) hipLaunchKernelGGL((wgrad_kernel<WgL14Op, true>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{g1, xin, Tr->d_groups, d, 0, 1}, wt.d_t14, wpart);
      else hipLaunchKernelGGL((wgrad_kernel<WgL14Op>), dim3(wt.n14), dim3(256), 0, ws_, WgL14Op{g1, xin, Tr->d_groups, d, 0, 1}, wt.d_t14, wpart);
      hipLaunchKernelGGL(k_wgrad_reduce14, dim3(grid1(maxW14).x, G), dim3(256), 0, ws_, wpart, wt.d_i14, Tr->d_groups, 1, gp); }
    bw.act3 = g1; bw.gx8 = gY;                                                     // g_xin (per target) <- g_z1   (layer-4 operator);
    if ((rc = cdae_launch_layer(Mo, 4, bw, stream, "train_l1_dgrad_gemm"))) return rc;   // gY is free by now
    { XSQ_PROF("train_input_grad_reduce", stream); hipLaunchKernelGGL(k_input_grad_partial, dim3((unsigned)Tr->sumF, 4 * Bn), dim3(256), 0, stream, xin, gY, Tr->d_groups, Tr->d_rows, d, part);
      hipLaunchKernelGGL(k_input_grad_final, grid1(Tr->sumF), dim3(256), 0, stream, part, Tr->d_params, Tr->d_groups, Tr->d_rows, (int)Tr->sumF, 4 * Bn, gp); }
    if (use_side) {      // the caller's stream joins: every gradient is complete before the update / before returning
        XSQ_HIP(hipEventRecord(Tr->ev_join, Tr->side));
        XSQ_HIP(hipStreamWaitEvent(stream, Tr->ev_join, 0));
    }
    // ---- update ---------------------------------------------------------------------------------------
    if (apply_update) {
        Tr->step += 1;
        const float b1 = 0.9f, b2 = 0.999f;
        { XSQ_PROF("train_adamw", stream); hipLaunchKernelGGL(k_adamw, grid1(Tr->nparams), dim3(256), 0, stream, Tr->d_params, gp, Tr->d_m, Tr->d_v, Tr->d_trainable,
                           Tr->nparams, lr, wd, b1, b2, 1e-8f, 1.f - powf(b1, (float)Tr->step), 1.f - powf(b2, (float)Tr->step)); }
    }
    XSQ_HIP(hipGetLastError());
    // loss scalars: per-block terms to pinned host memory behind an event (ring slot = ticket % LOSS_RING)
    const int64_t ticket = Tr->seq++;
    const int slot = (int)(ticket % xsq_train::LOSS_RING);
    XSQ_HIP(hipMemcpyAsync(Tr->h_loss + (size_t)slot * Tr->nblocks * 2, d_loss, (size_t)Tr->nblocks * 16, hipMemcpyDeviceToHost, stream));
    XSQ_HIP(hipEventRecord(Tr->ev_loss[slot], stream));
    // the next step's weights in GEMM layout (see above): BEHIND the loss event, so that a caller who waits for the loss --
    // the step and its update are complete by then -- does not wait for them
    if (apply_update) gather_pools();
    // loss_out given: wait here, as loss.item() does every step in the reference (training.py:110)
    return loss_out ? xsq_train_loss(Tr, ticket, loss_out) : XSQ_OK;
}

int64_t xsq_train_ticket(xsq_train* Tr) { return Tr ? Tr->seq - 1 : XSQ_ERR_ARG; }

int xsq_train_loss(xsq_train* Tr, int64_t ticket, double* loss_out) {
    XSQ_REQUIRE(Tr && loss_out, "xsq_train_loss: null argument");
    XSQ_REQUIRE(ticket >= 0 && ticket < Tr->seq && ticket >= Tr->seq - xsq_train::LOSS_RING,
                "xsq_train_loss: ticket %lld is not among the last %d steps (next ticket %lld)", (long long)ticket,
                xsq_train::LOSS_RING, (long long)Tr->seq);
    const int slot = (int)(ticket % xsq_train::LOSS_RING);
    XSQ_HIP(hipEventSynchronize(Tr->ev_loss[slot]));
    const double* per = Tr->h_loss + (size_t)slot * Tr->nblocks * 2;
    loss_out[0] = loss_out[1] = 0.0;
    for (int b = 0; b < Tr->nblocks; ++b) { loss_out[0] += per[2 * b]; loss_out[1] += per[2 * b + 1]; }
    loss_out[0] /= Tr->nblocks; loss_out[1] /= Tr->nblocks;
    return XSQ_OK;
}

}  // extern "C"
