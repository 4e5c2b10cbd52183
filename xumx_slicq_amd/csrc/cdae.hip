// Per-block convolutional denoising auto-encoders (CDAE) of xumx-sliCQ-V2 for gfx950.
//
// Reference: xumx_slicq_v2/model.py:213-271 (_SlicedUnmixCDAE.forward), layer stack :130-181,
// causal first layer :274-290, phasemix phase.py:96-113 (== mask * X).
// Per block b (F bins, W = T_b coefficients per slice, hop = W/2, kf = freq kernel) and target:
//   xin  = (|X| + input_mean[f]) * input_scale[f]                       k_magnitude_whiten
//   L1   Conv2d(2->50,(kf,W),stride(1,hop)) + BN + ReLU      GEMM  M=(b,f1,t1)  K=(ci,df,dt)    N=50
//   L2   Conv2d(50->51,(kf,4)) + BN + ReLU                   GEMM  M=(b,f2,t2)  K=(df,dt,c1)    N=51
//   L3   ConvTranspose2d(51->50,(kf,4)) + BN + ReLU          GEMM  M=(b,f3,t3)  K=(df,dt',c2)   N=50
//   L4   ConvTranspose2d(50->2,(kf,W),stride(1,hop)) + bias  GEMM  M=(b,f4,u)   K=(df,1-tap,c3) N=(c,dt<hop)
//        + sigmoid -> mask;  Y[target] = mask * X  (complex)   fused epilogue
// BatchNorm (eval, eps 1e-5) is folded into the weights (scale) and a per-channel shift.
// Activations are channels-last with a channel stride of 52 floats (16-byte rows), so every
// K-run of an implicit-GEMM row is one contiguous, aligned span; all 70 blocks x 4 targets
// of a layer run in ONE grouped launch driven by a tile table.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/xumx_slicq_hip.h"
#include "gemm_tile.h"
#include "gemm_tile_bf3.h"
#include "gemm_tile_bf6.h"
#include "plan.h"
#include "prof.h"

#include "cdae_api.h"

namespace xsq {

// ------------------------------------------------------------------------------------------
// |X| + whitening.  One thread per FOUR consecutive complex coefficients of the 2B-channel arena
// (two 16-byte loads, one 16-byte store); block sizes are multiples of 4, so a quad never
// straddles two blocks or two frequency rows.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_magnitude_whiten(const float4* __restrict__ X, float4* __restrict__ xin,
                                                           const int64_t* __restrict__ cum,
                                                           const CdaeBlockDev* __restrict__ blocks,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ scale, int nblocks,
                                                           int BC, int S, int64_t nquads, int split) {
    // block boundaries in LDS: the binary search below was seven DEPENDENT global loads per thread
    __shared__ int64_t cum_s[128];
    for (int i = threadIdx.x; i < nblocks && i < 128; i += 256) cum_s[i] = cum[i];
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    const int64_t i = 4 * q;
    const int64_t per = (int64_t)BC * S;            // arena offset of block b is per*cum[b]
    int lo = 0, hi = nblocks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (per * (nblocks <= 128 ? cum_s[mid] : cum[mid]) <= i) lo = mid; else hi = mid - 1;
    }
    const CdaeBlockDev& b = blocks[lo];
    const int64_t r = i - per * (nblocks <= 128 ? cum_s[lo] : cum[lo]);
    const int f = (int)((r / ((int64_t)S * b.T)) % b.F);
    const float mu = mean[b.cumF + f], sc = scale[b.cumF + f];
    const float4 z0 = X[2 * q], z1 = X[2 * q + 1];
    float4 o;
    o.x = whiten_mag(z0.x, z0.y, mu, sc);
    o.y = whiten_mag(z0.z, z0.w, mu, sc);
    o.z = whiten_mag(z1.x, z1.y, mu, sc);
    o.w = whiten_mag(z1.z, z1.w, mu, sc);
    if (split) { bf3_words2(o.x, o.y, o.x, o.y); bf3_words2(o.z, o.w, o.z, o.w); }
    xin[q] = o;
}

__global__ __launch_bounds__(256) void k_split_pool(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = bf3_word(src[i]);
}

// ------------------------------------------------------------------------------------------
// implicit-GEMM operators.  group id = block*4 + target.
// ------------------------------------------------------------------------------------------
struct CdaeGroup {
    int M, N, K, ldb;
    const float* B;
    const float* shift;
    const float* in;     // input activation base for this (block[,target])
    float* out;          // output activation base
    int Fo, To;          // output rows / time positions per batch item
    int Fi, Ti;          // input extents (bounds of the transposed convolutions)
    int F, T, hop, kf, tgt;
    int64_t cum;
};

struct RowFT {
    const float* p;   // row base pointer (may be out of range for transposed convs; checked per load)
    int f, t;         // row's frequency / time coordinate
    unsigned vo;      // layers 1 and 4: byte offset of the row base inside the group's input (buffer loads), BUF_OOB past M
};

__device__ inline void split_row(int m, int Fo, int To, int& b, int& f, int& t) {
    const int per = Fo * To;
    b = m / per;
    const int r = m - b * per;
    f = r / To;
    t = r - f * To;
}

// BN shift + ReLU, channels-last store (layers 1-3): column n and n+32 of 52 padded channels
__device__ __forceinline__ void relu_shift_epilogue(const CdaeGroup& g, int row0, int n, const f32x16& a0,
                                                    const f32x16& a1, bool raw, bool split = false) {
    if (split) {       // operand format of the split-bf16 engine: (bf16 hi << 16) | bf16 lo per value
        const float s0 = g.shift[n];
        const bool c1 = n + 32 < CS;
        const float s1 = c1 ? g.shift[n + 32] : 0.f;
        float* d0 = g.out + (int64_t)row0 * CS + n;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float w0, w1, u0 = 0.f, u1 = 0.f;
            bf3_words2(fmaxf(a0[r] + s0, 0.f), fmaxf(a0[r + 1] + s0, 0.f), w0, w1);
            if (c1) bf3_words2(fmaxf(a1[r] + s1, 0.f), fmaxf(a1[r + 1] + s1, 0.f), u0, u1);
            if (row0 + acc_row(r) < g.M) { d0[acc_row(r) * CS] = w0; if (c1) d0[acc_row(r) * CS + 32] = u0; }
            if (row0 + acc_row(r + 1) < g.M) { d0[acc_row(r + 1) * CS] = w1; if (c1) d0[acc_row(r + 1) * CS + 32] = u1; }
        }
        return;
    }
    const float s0 = g.shift[n];
    const bool c1 = n + 32 < CS;
    const float s1 = c1 ? g.shift[n + 32] : 0.f;
    float* d0 = g.out + (int64_t)row0 * CS + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (row0 + acc_row(r) >= g.M) break;
        float* d = d0 + acc_row(r) * CS;
        d[0] = raw ? a0[r] : fmaxf(a0[r] + s0, 0.f);
        if (c1) d[32] = raw ? a1[r] : fmaxf(a1[r] + s1, 0.f);
    }
}

// the same for the exact-width column layout of the fp32 engines (gemm_tile.h XW = 1, cdae_slab.h MODE 3):
// columns 0..31 in the 32x32 accumulator, 32..47 in two 16x16 blocks, 48..47+nv as half-wave partial sums
// Wide stores: the wave's 32 x 52 output block is one contiguous 6,656-byte span of the channels-last activation, but
// the accumulator layout gives a lane one COLUMN of it (25 four-byte stores per lane, each behind its own 64-bit
// address and row predicate: 0.2 ms of layer 1's 0.68, measured with the epilogue ablated).  The block goes through a
// per-wave LDS image (the staging buffers are dead once the K loop has passed its last barrier) and leaves as seven
// 16-byte stores per lane, lanes consecutive: whole 1 KB wave-instructions.
constexpr int XW_TILE = 32 * CS;          // floats of one wave's image
__device__ __forceinline__ void relu_shift_epilogue_xw(const CdaeGroup& g, int rowb, int lane, const f32x16& a0,
                                                       const f32x4 (&a16)[2], float (&av)[4], int nv, float* img) {
    const int lrow = lane & 31, lk = lane >> 5, q16 = lane >> 4;
    {
        const float sh = g.shift[lrow];
#pragma unroll
        for (int r = 0; r < 16; ++r) img[(acc_row(r) + 4 * lk) * CS + lrow] = fmaxf(a0[r] + sh, 0.f);
    }
    {
        const int col = 32 + (lane & 15);
        const float sh = g.shift[col];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * rb + 4 * q16 + r) * CS + col] = fmaxf(a16[rb][r] + sh, 0.f);
    }
    {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) av[cc] = cc < nv ? av[cc] + __shfl_xor(av[cc], 32) : 0.f;
        if (lk == 0) {
            const float4 sh = *reinterpret_cast<const float4*>(g.shift + 48);
            *reinterpret_cast<float4*>(img + lrow * CS + 48) =
                make_float4(fmaxf(av[0] + sh.x, 0.f), fmaxf(av[1] + sh.y, 0.f), fmaxf(av[2] + sh.z, 0.f), fmaxf(av[3] + sh.w, 0.f));
        }
    }
    __builtin_amdgcn_wave_barrier();          // same wave, LDS operations complete in order: no workgroup barrier needed
    const int nvalid = max(0, min(g.M - rowb, 32)) * (CS / 4);               // float4 slots of the rows that exist (none for a wave past M)
    // buffer stores: the descriptor ends behind the last row that exists (and behind the image's 416 slots) -- no
    // predicate per store
    const __amdgpu_buffer_rsrc_t ro = buf_rsrc(g.out + (int64_t)__builtin_amdgcn_readfirstlane(rowb) * CS, 16u * (unsigned)__builtin_amdgcn_readfirstlane(nvalid));
#pragma unroll
    for (int i = 0; i < (XW_TILE / 4 + 63) / 64; ++i)
        buf_st4(*reinterpret_cast<const float4*>(img + 4 * (lane + 64 * i)), ro, 16u * (unsigned)lane + 1024u * i, 0);      // (displacement in the lane offset: common.h, buf_st4)
}

// ---- layer 1 -----------------------------------------------------------------------------
// CAUSAL (the realtime model's first layer, model.py:274-290) is a compile-time fact of the operator: as a run-time flag the
// K loop carried the predicated 4-byte path of the causal layer beside the buffer path (a uniform branch per load and
// twice the code; tools/isa_budget.py counted both).
template <bool CAUSAL>
struct CdaeL1OpT {
    typedef CdaeGroup Group;
    typedef RowFT RowA;
    static constexpr bool STAMPED = XSQ_GEMM_STAMP == 2;      // (diagnostic builds with XSQ_GEMM_STAMP=2: tools/gemm_phases.py)
    CdaeArgs a;
    __device__ Group group(int gid) const {
        const CdaeBlockDev& b = a.blocks[gid >> 2];
        const int tgt = gid & 3;
        Group g;
        g.F = b.F; g.T = b.T; g.hop = b.hop; g.kf = b.kf; g.tgt = tgt; g.cum = b.cum;
        g.Fo = b.F1; g.To = a.T1; g.Fi = b.F; g.Ti = a.S * b.T;
        g.M = a.Bn * g.Fo * g.To; g.N = CS; g.K = 2 * b.kf * b.T; g.ldb = b.ld1;
        g.B = (a.poolB ? a.poolB : a.pool) + b.w1[tgt]; g.shift = a.pool + b.s1[tgt];
        g.in = a.xin8 ? a.xin8 + (int64_t)a.Bn * 8 * a.S * b.cum + (int64_t)tgt * a.Bn * 2 * b.F * g.Ti
                      : a.xin + (int64_t)a.Bn * 2 * a.S * b.cum;
        g.out = a.act1 + (int64_t)CS * a.Bn * a.T1 * (4 * (int64_t)b.cumF1 + (int64_t)tgt * b.F1);
        return g;
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r; r.p = nullptr; r.f = 0; r.t = 0; r.vo = BUF_OOB;
        if (m >= g.M) return r;
        int b, f, t;
        split_row(m, g.Fo, g.To, b, f, t);
        r.f = f;
        r.t = t * g.hop - ((CAUSAL && !a.xin8) ? g.T - 1 : 0);   // first input sample of the window
        if constexpr (CAUSAL) r.p = g.in + ((int64_t)b * 2 * g.F + f) * g.Ti + r.t;
        r.vo = 4u * (unsigned)((b * 2 * g.F + f) * g.Ti + r.t);    // (non-causal: r.t >= 0; the group's input is < 2^30 bytes, launch check)
        return r;
    }
    // k = (ci*kf + df)*T + dt.  The cursor keeps dt, df and the element offset (ci*F + df)*Ti of the segment; a step of
    // 16 crosses one segment boundary at most when T >= 16 (every Bark-262 block) and several for T = 4, 8, 12, which
    // xsq_model_create accepts: hence a loop.
    struct Cursor { int k, dt, df, off; };
    __device__ Cursor cursor(const Group& g, int k) const {
        Cursor c;
        const int seg = k / g.T, ci = seg / g.kf;
        c.k = k; c.dt = k - seg * g.T; c.df = seg - ci * g.kf;
        c.off = (ci * g.F + c.df) * g.Ti;                      // < 2^31: the input arena holds fewer floats (launch check)
        return c;
    }
    __device__ void advance(const Group& g, Cursor& c) const {
        c.k += 16; c.dt += 16;
        while (c.dt >= g.T) {
            c.dt -= g.T; c.off += g.Ti;
            if (++c.df == g.kf) { c.df = 0; c.off += (g.F - g.kf) * g.Ti; }
        }
    }
    __device__ float4 load_a4(const Group& g, const RowA& r, const Cursor& c) const { return load_at(g, r, c.k, c.off, c.dt); }
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        const int seg = k / g.T, dt = k - seg * g.T;          // seg = ci*kf + df
        const int ci = seg / g.kf, df = seg - ci * g.kf;
        return load_at(g, r, k, (ci * g.F + df) * g.Ti, dt);
    }
    __device__ float4 load_at(const Group& g, const RowA& r, int k, int off, int dt) const {
        if constexpr (!CAUSAL) {
            // one 16-byte buffer load (dword aligned is enough): row offset + cursor, rows past M and k past K switched
            // out of range -- no pointer sums, zero fills or exec-masked branches in the K loop
            const unsigned vo = r.vo + 4u * (unsigned)(off + dt);
            return buf_ld4(buf_rsrc(g.in, 0x40000000u), k < g.K ? vo : BUF_OOB, 0);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r.p == nullptr || k >= g.K) return v;
        const float* p = r.p + (off + dt);
        const int t0 = r.t + dt;     // causal: zero left padding of W-1 samples (forward) / cropped right edge (xin8)
        if (t0 >= 0 && t0 < g.Ti) v.x = p[0];
        if (t0 + 1 >= 0 && t0 + 1 < g.Ti) v.y = p[1];
        if (t0 + 2 >= 0 && t0 + 2 < g.Ti) v.z = p[2];
        if (t0 + 3 >= 0 && t0 + 3 < g.Ti) v.w = p[3];
        return v;
    }
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool) const {
        relu_shift_epilogue(g, row0, n, a0, a1, a.raw != 0, a.split != 0);
    }
    static constexpr int NV = H1 - 48;        // real output channels past 47
    __device__ void epilogue_xw(const Group& g, int rowb, int lane, const f32x16& a0, const f32x4 (&a16)[2], float (&av)[4], float* img) const {
        relu_shift_epilogue_xw(g, rowb, lane, a0, a16, av, NV, img);
    }
};
struct CdaeL1Op : CdaeL1OpT<false> { __host__ __device__ CdaeL1Op(const CdaeArgs& a_) { a = a_; } };
struct CdaeL1CausalOp : CdaeL1OpT<true> { __host__ __device__ CdaeL1CausalOp(const CdaeArgs& a_) { a = a_; } };

// ---- layer 2 -----------------------------------------------------------------------------
struct CdaeL2Op {
    typedef CdaeGroup Group;
    typedef RowFT RowA;
    CdaeArgs a;
    __device__ Group group(int gid) const {
        const CdaeBlockDev& b = a.blocks[gid >> 2];
        const int tgt = gid & 3;
        Group g;
        g.F = b.F; g.T = b.T; g.hop = b.hop; g.kf = b.kf; g.tgt = tgt; g.cum = b.cum;
        g.Fo = b.F2; g.To = a.T2; g.Fi = b.F1; g.Ti = a.T1;
        g.M = a.Bn * g.Fo * g.To; g.N = CS; g.K = b.kf * 4 * CS; g.ldb = b.kf * 4 * CS;
        g.B = (a.poolB ? a.poolB : a.pool) + b.w2[tgt]; g.shift = a.pool + b.s2[tgt];
        g.in = a.act1 + (int64_t)CS * a.Bn * a.T1 * (4 * (int64_t)b.cumF1 + (int64_t)tgt * b.F1);
        g.out = a.act2 + (int64_t)CS * a.Bn * a.T2 * (4 * (int64_t)b.cumF2 + (int64_t)tgt * b.F2);
        return g;
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r; r.p = nullptr; r.f = 0; r.t = 0;
        if (m >= g.M) return r;
        int b, f, t;
        split_row(m, g.Fo, g.To, b, f, t);
        r.f = f; r.t = t;
        r.p = g.in + (((int64_t)b * g.Fi + f) * g.Ti + t) * CS;
        return r;
    }
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        if (r.p == nullptr || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int df = k / (4 * CS), rem = k - df * 4 * CS;   // rem = dt*52 + c1, contiguous in memory
        return *reinterpret_cast<const float4*>(r.p + (int64_t)df * g.Ti * CS + rem);
    }
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool) const {
        relu_shift_epilogue(g, row0, n, a0, a1, a.raw != 0, a.split != 0);
    }
    static constexpr int NV = H2 - 48;        // real output channels past 47
    __device__ void epilogue_xw(const Group& g, int rowb, int lane, const f32x16& a0, const f32x4 (&a16)[2], float (&av)[4], float* img) const {
        relu_shift_epilogue_xw(g, rowb, lane, a0, a16, av, NV, img);
    }
};

// ---- layer 3 (transposed conv as a gather over a zero-extended input) ----------------------
struct CdaeL3Op {
    typedef CdaeGroup Group;
    typedef RowFT RowA;
    CdaeArgs a;
    __device__ Group group(int gid) const {
        const CdaeBlockDev& b = a.blocks[gid >> 2];
        const int tgt = gid & 3;
        Group g;
        g.F = b.F; g.T = b.T; g.hop = b.hop; g.kf = b.kf; g.tgt = tgt; g.cum = b.cum;
        g.Fo = b.F1; g.To = a.T1; g.Fi = b.F2; g.Ti = a.T2;
        g.M = a.Bn * g.Fo * g.To; g.N = CS; g.K = b.kf * 4 * CS; g.ldb = b.kf * 4 * CS;
        g.B = (a.poolB ? a.poolB : a.pool) + b.w3[tgt]; g.shift = a.pool + b.s3[tgt];
        g.in = a.act2 + (int64_t)CS * a.Bn * a.T2 * (4 * (int64_t)b.cumF2 + (int64_t)tgt * b.F2);
        g.out = a.act3 + (int64_t)CS * a.Bn * a.T1 * (4 * (int64_t)b.cumF1 + (int64_t)tgt * b.F1);
        return g;
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r; r.p = nullptr; r.f = 0; r.t = 0;
        if (m >= g.M) return r;
        int b, f, t;
        split_row(m, g.Fo, g.To, b, f, t);
        r.f = f; r.t = t;
        r.p = g.in + (((int64_t)b * g.Fi + f) * g.Ti + (t - 3)) * CS;   // (f3, t3-3); checked per load
        return r;
    }
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        if (r.p == nullptr || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
        const int df = k / (4 * CS), rem = k - df * 4 * CS;
        const int dtp = rem / CS;                              // dt' = 3 - dt
        const int fi = r.f - df, ti = r.t - 3 + dtp;
        if (fi < 0 || fi >= g.Fi || ti < 0 || ti >= g.Ti) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(r.p - (int64_t)df * g.Ti * CS + rem);
    }
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool) const {
        relu_shift_epilogue(g, row0, n, a0, a1, a.raw != 0, a.split != 0);
    }
    static constexpr int NV = H1 - 48;        // real output channels past 47
    __device__ void epilogue_xw(const Group& g, int rowb, int lane, const f32x16& a0, const f32x4 (&a16)[2], float (&av)[4], float* img) const {
        relu_shift_epilogue_xw(g, rowb, lane, a0, a16, av, NV, img);
    }
};

// ---- layer 4 + sigmoid + mask * X ------------------------------------------------------------
// Output-stationary form of the strided transposed conv: output sample tau = u*hop + dt (dt < hop)
// receives taps t3 = u (kernel column dt) and t3 = u-1 (kernel column dt + hop).
struct CdaeL4Op {
    typedef CdaeGroup Group;
    typedef RowFT RowA;
    static constexpr bool STAMPED = XSQ_GEMM_STAMP == 1;      // (diagnostic builds with XSQ_GEMM_STAMP=1: tools/gemm_phases.py)
    CdaeArgs a;
    __device__ Group group(int gid) const {
        const CdaeBlockDev& b = a.blocks[gid >> 2];
        const int tgt = gid & 3;
        Group g;
        g.F = b.F; g.T = b.T; g.hop = b.hop; g.kf = b.kf; g.tgt = tgt; g.cum = b.cum;
        g.Fo = b.F; g.To = a.gx8 ? a.T1 + 1 : 2 * a.S; g.Fi = b.F1; g.Ti = a.T1;
        g.M = a.Bn * g.Fo * g.To; g.N = b.T; g.K = b.kf * 2 * CS; g.ldb = b.ld4;
        g.B = (a.poolB ? a.poolB : a.pool) + b.w4[tgt]; g.shift = a.pool + b.b4[tgt];
        g.in = a.act3 + (int64_t)CS * a.Bn * a.T1 * (4 * (int64_t)b.cumF1 + (int64_t)tgt * b.F1);
        g.out = nullptr;
        return g;
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r; r.p = nullptr; r.f = 0; r.t = 0; r.vo = BUF_OOB;
        if (m >= g.M) return r;
        int b, f, t;
        split_row(m, g.Fo, g.To, b, f, t);
        r.f = f; r.t = t;
        r.p = g.in + (((int64_t)b * g.Fi + f) * g.Ti + t) * CS;
        r.vo = 4u * (unsigned)(((b * g.Fi + f) * g.Ti + t) * CS);  // (f, t may lie past the input: such loads are switched off below)
        return r;
    }
    // k = (df*2 + (1 - tap))*52 + c3: the TWO taps of a row, stored tap 1 first, are ONE contiguous run of 104 floats --
    // positions u - 1 and u of input row f - df -- so the operand address is "row base - df rows + kk" with no division
    // and one range test per half (kk < 52: position u - 1 exists, else: position u exists).  The cursor carries kk and
    // the row offset; the engine asks for k = s_kq, s_kq + 16, ... in order.
    struct Cursor { int k, kk, df; };
    __device__ Cursor cursor(const Group& g, int k) const {
        Cursor c;
        c.k = k; c.df = k / (2 * CS); c.kk = k - c.df * 2 * CS;
        return c;
    }
    __device__ void advance(const Group& g, Cursor& c) const {
        c.k += 16; c.kk += 16;
        if (c.kk >= 2 * CS) { c.kk -= 2 * CS; ++c.df; }
    }
    __device__ float4 load_a4(const Group& g, const RowA& r, const Cursor& c) const { return load_at(g, r, c.k, c.df, c.kk); }
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        const int df = k / (2 * CS);
        return load_at(g, r, k, df, k - df * 2 * CS);
    }
    __device__ float4 load_at(const Group& g, const RowA& r, int k, int df, int kk) const {
        const int fi = r.f - df;
        const bool pos_ok = kk < CS ? r.t >= 1 : r.t < g.Ti;           // tap 1 reads position u - 1, tap 0 position u
        const bool ok = k < g.K && (unsigned)fi < (unsigned)g.Fi && pos_ok;
        // buffer load: a row past M keeps its switch through the backward displacement (common.h); everything else that
        // must read as zero is switched off here -- no zero fill, no exec-masked branch, no 64-bit address
        const unsigned vo = r.vo + 4u * (unsigned)(kk - (df * g.Ti + 1) * CS);
        return buf_ld4(buf_rsrc(g.in, 0x40000000u), ok ? vo : BUF_OOB, 0);
    }
    // mask = sigmoid(acc + bias[c]);  Y[target] = mask * X.  Column n = c*hop + dt with n < 2*hop, so
    // c is a compare.
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool wide) const {
        const bool v0 = n < g.N, v1 = wide && n + 32 < g.N;
        const int n1 = n + 32;
        const int ca = n >= g.hop, cb = n1 >= g.hop;
        const int dta = n - ca * g.hop, dtb = n1 - cb * g.hop;
        if (a.gx8) {        // data gradient of layer 1: raw store, (target, b, c, f, u*hop + dt), row length To*hop
            const int64_t STp = (int64_t)g.To * g.hop, FSTp = (int64_t)g.F * STp;
            float* o = a.gx8 + 4 * (int64_t)a.Bn * g.To * g.cum + (int64_t)g.tgt * a.Bn * 2 * FSTp;
            int b, f, u;
            split_row(row0, g.Fo, g.To, b, f, u);
            int prev = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u += acc_row(r) - prev; prev = acc_row(r);
                while (u >= g.To) { u -= g.To; if (++f == g.Fo) { f = 0; ++b; } }
                if (row0 + acc_row(r) >= g.M) break;
                float* d = o + (int64_t)(b * 2) * FSTp + (int64_t)f * STp + (int64_t)u * g.hop;
                if (v0) d[ca * FSTp + dta] = a0[r];
                if (v1) d[cb * FSTp + dtb] = a1[r];
            }
            return;
        }
        const float ba = g.shift[ca], bb = g.shift[cb];
        // Rows m = (b, f, u) of one batch item are CONTIGUOUS in time: tau = f*S*T + u*hop = (m - b*F*To)*hop
        // (To*hop = S*T), so element (b, c, f, tau) of the mix arena and of the target's sub-arena sits at
        // m*hop + (b + c)*F*S*T: one multiply per row instead of a (b, f, u) split and 64-bit products.
        const int FST = g.F * a.S * g.T;                      // 2*Bn*FST < 2^31 is checked at launch
        const int perb = g.Fo * g.To;
        const float2* X2 = reinterpret_cast<const float2*>(a.X) + (int64_t)a.Bn * 2 * a.S * g.cum;
        float2* Y2 = reinterpret_cast<float2*>(a.Y) + (int64_t)a.Bn * 8 * a.S * g.cum + (int64_t)g.tgt * a.Bn * 2 * FST;
        float* Mk = a.masks ? a.masks + (int64_t)a.Bn * 8 * a.S * g.cum + (int64_t)g.tgt * a.Bn * 2 * FST : nullptr;
        const int offa = ca * FST + dta, offb = cb * FST + dtb;
        // Straight-line code matters here: with per-row / per-column branches the compiler has to place
        // conservative s_waitcnt vmcnt(0) at every join, and each one also waits for the STORES issued just
        // before it -- the epilogue then runs at one store round trip per row (measured: 0.67 ms of the
        // layer's 1.31 ms).  Fast path (every row of the slab valid, at most one batch boundary inside it):
        // per column block one divergent region holding 16 loads, then 16 sigmoids and 16 stores, no joins.
        const int slab0 = row0 & ~31;                          // tiles start at multiples of 128, slabs at multiples of 32
        const int b0 = slab0 / perb;
        const int next_b = (b0 + 1) * perb;
        if (slab0 + 32 <= g.M && perb >= 32) {
            const int obase = row0 * g.hop + b0 * FST, ocut = next_b - row0;     // row r: obase + acc_row(r)*hop (+ FST past the batch boundary)
            auto off = [&](int r) { return obase + acc_row(r) * g.hop + (acc_row(r) >= ocut ? FST : 0); };
            if (!a.Y && next_b >= slab0 + 32) {
                // masks only, no batch boundary inside the slab (all but one slab in Fo * To / 32): buffer stores with the
                // row's displacement as the SCALAR offset -- per element the sigmoid's five instructions and nothing else
                // (the pointer form below spent six more on the element's address; beside fp32 MFMAs they are not hidden).
                // Same arithmetic as below: bitwise the same masks.
                const int s0 = __builtin_amdgcn_readfirstlane(slab0);       // (wave-uniform by construction; the descriptor wants scalars)
                const int b0u = __builtin_amdgcn_readfirstlane(b0);
                const __amdgpu_buffer_rsrc_t rm = buf_rsrc(Mk + (int64_t)s0 * g.hop + (int64_t)b0u * FST, 0x40000000u);   // (launch check: < 2^30 bytes)
                const unsigned lane_off = 4u * (unsigned)((row0 - slab0) * g.hop);
                const unsigned va = v0 ? lane_off + 4u * (unsigned)offa : BUF_OOB, vb = v1 ? lane_off + 4u * (unsigned)offb : BUF_OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_amdgcn_rcpf(1.f + __expf(-(a0[r] + ba)))), rm, (int)va, 4 * acc_row(r) * g.hop, 0);
                if (wide) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_amdgcn_rcpf(1.f + __expf(-(a1[r] + bb)))), rm, (int)vb, 4 * acc_row(r) * g.hop, 0);
                }
                return;
            }
            if (!a.Y) {      // masks only (the inverse transform multiplies by the mix on its way in)
                if (v0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Mk[off(r) + offa] = __builtin_amdgcn_rcpf(1.f + __expf(-(a0[r] + ba)));
                }
                if (v1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Mk[off(r) + offb] = __builtin_amdgcn_rcpf(1.f + __expf(-(a1[r] + bb)));
                }
                return;
            }
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 4) {
                if (v0) {
                    float2 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = X2[off(r0 + q) + offa];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float mk = __builtin_amdgcn_rcpf(1.f + __expf(-(a0[r0 + q] + ba)));
                        Y2[off(r0 + q) + offa] = make_float2(mk * x[q].x, mk * x[q].y);
                    }
                }
                if (v1) {
                    float2 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = X2[off(r0 + q) + offb];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float mk = __builtin_amdgcn_rcpf(1.f + __expf(-(a1[r0 + q] + bb)));
                        Y2[off(r0 + q) + offb] = make_float2(mk * x[q].x, mk * x[q].y);
                    }
                }
            }
            if (Mk) {        // training: the masks as well (second pass, sigmoid recomputed)
                if (v0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Mk[off(r) + offa] = __builtin_amdgcn_rcpf(1.f + __expf(-(a0[r] + ba)));
                }
                if (v1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Mk[off(r) + offb] = __builtin_amdgcn_rcpf(1.f + __expf(-(a1[r] + bb)));
                }
            }
            return;
        }
        // generic path: ragged last slab, or batch items shorter than a slab
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + acc_row(r);
            if (m >= g.M) break;
            const int o = m * g.hop + (m / perb) * FST;
            if (v0) {
                const float mk = __builtin_amdgcn_rcpf(1.f + __expf(-(a0[r] + ba)));
                if (a.Y) { const float2 x = X2[o + offa]; Y2[o + offa] = make_float2(mk * x.x, mk * x.y); }
                if (Mk) Mk[o + offa] = mk;
            }
            if (v1) {
                const float mk = __builtin_amdgcn_rcpf(1.f + __expf(-(a1[r] + bb)));
                if (a.Y) { const float2 x = X2[o + offb]; Y2[o + offb] = make_float2(mk * x.x, mk * x.y); }
                if (Mk) Mk[o + offb] = mk;
            }
        }
    }
    // 16-column blocks of the exact-width tiles (gemm_tile.h, XW = 2): lane = column n0 + (l & 15), its four
    // registers are rows 4 (l >> 4) .. + 3 of each of the two 16-row blocks
    __device__ void epilogue16(const Group& g, int rowb, int lane, int n0, const f32x4 (&acc)[2]) const {
        const int n = n0 + (lane & 15), q16 = lane >> 4;
        if (n >= g.N) return;
        const int c = n >= g.hop, dt = n - c * g.hop;
        const float bias = g.shift[c];
        const int FST = g.F * a.S * g.T, perb = g.Fo * g.To;
        const float2* X2 = reinterpret_cast<const float2*>(a.X) + (int64_t)a.Bn * 2 * a.S * g.cum;
        float2* Y2 = a.Y ? reinterpret_cast<float2*>(a.Y) + (int64_t)a.Bn * 8 * a.S * g.cum + (int64_t)g.tgt * a.Bn * 2 * FST : nullptr;
        float* Mk = a.masks ? a.masks + (int64_t)a.Bn * 8 * a.S * g.cum + (int64_t)g.tgt * a.Bn * 2 * FST : nullptr;
        const int off = c * FST + dt;
        if (!Y2 && rowb + 32 <= g.M && perb >= 32) {
            // masks only (inference), every row of the block valid (uniform): eight stores back to back.  In the general
            // loop below the load of X sits between the stores and each row is a divergent region; the s_waitcnt vmcnt(0)
            // the compiler places at the top of every region (for the bias, for X) is executed whether or not anything
            // was loaded, and waits for the previous row's STORE: one round trip per row.  The T = 16 .. 60 bands -- a
            // third of all tiles -- go through this function only.
            const int m0 = rowb + 4 * q16;
            const int b0 = rowb / perb;                       // uniform; at most one batch boundary inside 32 rows
            const int obase = m0 * g.hop + b0 * FST + off, ocut = (b0 + 1) * perb - m0;
            float mk[2][4];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[rb][r] = __builtin_amdgcn_rcpf(1.f + __expf(-(acc[rb][r] + bias)));
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dm = 16 * rb + r;
                    Mk[obase + dm * g.hop + (dm >= ocut ? FST : 0)] = mk[rb][r];
                }
            return;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = rowb + 16 * rb + 4 * q16 + r;
                if (m >= g.M) continue;
                const int o = m * g.hop + (m / perb) * FST + off;
                const float mk = __builtin_amdgcn_rcpf(1.f + __expf(-(acc[rb][r] + bias)));
                if (Y2) { const float2 x = X2[o]; Y2[o] = make_float2(mk * x.x, mk * x.y); }
                if (Mk) Mk[o] = mk;
            }
    }
};

}  // namespace xsq
#include "cdae_slab.h"
#include "cdae_wino.h"
#include "cdae_wino4.h"
#include "cdae_l1f.h"
#include "cdae_l4f.h"
namespace xsq {

// ------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------
}  // namespace xsq
namespace xsq {

// F(4, 4) weights of layers 2 / 3 (cdae_wino4.h: an A/B arm, off by default) are built with a model only when XSQ_WINO4=1 is set at
// xsq_model_create -- 7 / 5 of the F(2, 4) pool on top, and their share of the fp64 transform at load time
static bool wino4_wanted() { return getenv("XSQ_WINO4") && atoi(getenv("XSQ_WINO4")) != 0; }

static const int L23_MT = 1;     // 256-row tiles (MT = 2) measured slower: 192 VGPR -> 2 waves per SIMD (L3 1.51 -> 1.74 ms)

static int kf_of(int F) { return F < 10 ? 1 : (F < 20 ? 3 : 5); }   // model.py:112-117

static int get_cdae_tiles(xsq_model* Mo, int layer, int Bn, int S, TileTable* out, int mt23 = L23_MT, bool n16 = false, bool tgt_outer = false) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(layer + 16 * mt23 + (n16 ? 128 : 0) + (tgt_outer ? 256 : 0), Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    std::vector<TileDev> t;
    // longest tiles first: order the blocks by the K of this layer, descending
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    auto kof = [&](int b) {
        const CdaeBlockDev& d = Mo->blocks[b];
        return layer == 1 ? 2 * d.kf * d.T : (layer >= 4 ? d.kf * 2 * CS : d.kf * 4 * CS);
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return kof(x) > kof(y); });
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        int64_t M; int N;
        switch (layer) {
            case 1: M = (int64_t)Bn * d.F1 * T1; N = CS; break;
            case 2: M = (int64_t)Bn * d.F2 * T2; N = CS; break;
            case 3: M = (int64_t)Bn * d.F1 * T1; N = CS; break;
            case 4: M = (int64_t)Bn * d.F * 2 * S; N = d.T; break;
            default: M = (int64_t)Bn * d.F * (T1 + 1); N = d.T; break;     // 6: layer-4 operator as layer-1 data gradient
        }
        if (layer == 1 || layer == 4 || layer == 6) {
            // the four targets read the same input (L1: whitened magnitude) / the same mix X (L4 epilogue):
            // keep their tiles of one (row, column) patch adjacent so the re-reads hit the XCD's L2
            auto push = [&](int64_t m0, int n0, int tgt) {
                const int rem = N - n0;      // n16 (layer 4, fp32 inference): widths 16 / 32 / 48 / 64, see gemm_tile.h XW = 2
                const int kind = n16 ? (rem <= 16 ? 2 : rem <= 32 ? 1 : rem <= 48 ? 3 : 0) : (rem <= 32 ? 1 : 0);
                t.push_back(TileDev{b * 4 + tgt, (int)m0, n0, kind});
            };
            if (tgt_outer) {
                // layer 4 storing masks only: the targets share nothing (each reads its own layer-3 activations), while
                // the column tiles of a row block share its operand rows and neighbouring row blocks share rows through the
                // frequency taps: one (block, target) after the other (r4t: 0.689 -> 0.675 ms; target innermost and
                // column tiles adjacent per target measured 0.694)
                for (int tgt = 0; tgt < NT; ++tgt)
                    for (int64_t m0 = 0; m0 < M; m0 += 128)
                        for (int n0 = 0; n0 < N; n0 += 64) push(m0, n0, tgt);
            } else {
                for (int64_t m0 = 0; m0 < M; m0 += 128)
                    for (int n0 = 0; n0 < N; n0 += 64)
                        for (int tgt = 0; tgt < NT; ++tgt) push(m0, n0, tgt);
            }
        } else {
            for (int tgt = 0; tgt < NT; ++tgt) push_group_tiles(t, b * 4 + tgt, M, N, 128 * mt23);
        }
    }
    TileTable tt;
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc(&tt.d_tiles, t.size() * sizeof(TileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(TileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the slab kernels (cdae_slab.h): 256 consecutive rows inside one batch item, all 64 columns
static int get_slab_tiles(xsq_model* Mo, int layer, int Bn, int S, TileTable* out, int max_kf = 1 << 30) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(layer + 64 + (max_kf < (1 << 30) ? 1000 * max_kf : 0), Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Mo->blocks[x].kf > Mo->blocks[y].kf; });
    std::vector<SlabTileDev> t;
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        if (d.kf > max_kf) continue;
        const int64_t perb = layer == 2 ? (int64_t)d.F2 * T2 : (int64_t)d.F1 * T1;
        for (int tgt = 0; tgt < NT; ++tgt) {
            const int64_t off1 = (int64_t)CS * Bn * T1 * (4 * (int64_t)d.cumF1 + (int64_t)tgt * d.F1);   // act1 / act3 of the (block, target)
            const int64_t off2 = (int64_t)CS * Bn * T2 * (4 * (int64_t)d.cumF2 + (int64_t)tgt * d.F2);   // act2
            SlabTileDev e;
            e.kf = d.kf;
            if (layer == 2) { e.Fo = d.F2; e.Fi = d.F1; e.in_off = off1; e.out_off = off2; e.shift_off = d.s2[tgt]; e.w_off = d.w2[tgt]; }
            else { e.Fo = d.F1; e.Fi = d.F2; e.in_off = off2; e.out_off = off1; e.shift_off = d.s3[tgt]; e.w_off = d.w3[tgt]; }
            const int To = layer == 2 ? T2 : T1;
            e.pad = 0;
            for (int bi = 0; bi < Bn; ++bi)
                for (int64_t r = 0; r < perb; r += SLAB_ROWS) {
                    e.m0 = (int)(bi * perb + r); e.b = bi; e.f0 = (int)(r / To); e.t0 = (int)(r % To);
                    t.push_back(e);
                }
        }
    }
    TileTable tt;                    // (d_tiles holds SlabTileDev entries for this key: cast at the launch sites)
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc((void**)&tt.d_tiles, t.size() * sizeof(SlabTileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(SlabTileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the Winograd kernels (cdae_wino.h): 64 consecutive output PAIRS of one batch item in the flattened (f, pair) space
static int get_wino_tiles(xsq_model* Mo, int layer, int Bn, int S, TileTable* out, int min_kf = 0) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(layer + 96 + 1000 * min_kf, Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Mo->blocks[x].kf > Mo->blocks[y].kf; });
    std::vector<WinoTileDev> t;
    const int To = layer == 2 ? T2 : T1, P = (To + 1) / 2;
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        if (d.kf < min_kf) continue;
        for (int tgt = 0; tgt < NT; ++tgt) {
            const int64_t off1 = (int64_t)CS * Bn * T1 * (4 * (int64_t)d.cumF1 + (int64_t)tgt * d.F1);   // act1 / act3 of the (block, target)
            const int64_t off2 = (int64_t)CS * Bn * T2 * (4 * (int64_t)d.cumF2 + (int64_t)tgt * d.F2);   // act2
            WinoTileDev e;
            e.kf = d.kf; e.P = P;
            if (layer == 2) { e.Fo = d.F2; e.Fi = d.F1; e.in_off = off1; e.out_off = off2; e.shift_off = d.s2[tgt]; e.u_off = d.u2[tgt]; }
            else { e.Fo = d.F1; e.Fi = d.F2; e.in_off = off2; e.out_off = off1; e.shift_off = d.s3[tgt]; e.u_off = d.u3[tgt]; }
            const int perb = e.Fo * P;
            e.pad0 = e.pad1 = 0;
            for (int bi = 0; bi < Bn; ++bi)
                for (int Q = 0; Q < perb; Q += WN_PAIRS) {
                    e.Q0 = Q; e.b = bi;
                    t.push_back(e);
                }
        }
    }
    TileTable tt;                    // (d_tiles holds WinoTileDev entries for this key: cast at the launch site)
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc((void**)&tt.d_tiles, t.size() * sizeof(WinoTileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(WinoTileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the F(4, 4) kernels (cdae_wino4.h): 64 consecutive output QUADS of one batch item in the flattened (f, quad) space
static int get_wino4_tiles(xsq_model* Mo, int layer, int Bn, int S, TileTable* out) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(layer + 192, Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Mo->blocks[x].kf > Mo->blocks[y].kf; });
    std::vector<WinoTileDev> t;
    const int To = layer == 2 ? T2 : T1, P = (To + 3) / 4;
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        for (int tgt = 0; tgt < NT; ++tgt) {
            const int64_t off1 = (int64_t)CS * Bn * T1 * (4 * (int64_t)d.cumF1 + (int64_t)tgt * d.F1);   // act1 / act3 of the (block, target)
            const int64_t off2 = (int64_t)CS * Bn * T2 * (4 * (int64_t)d.cumF2 + (int64_t)tgt * d.F2);   // act2
            WinoTileDev e;
            e.kf = d.kf; e.P = P;
            if (layer == 2) { e.Fo = d.F2; e.Fi = d.F1; e.in_off = off1; e.out_off = off2; e.shift_off = d.s2[tgt]; e.u_off = d.uq2[tgt]; }
            else { e.Fo = d.F1; e.Fi = d.F2; e.in_off = off2; e.out_off = off1; e.shift_off = d.s3[tgt]; e.u_off = d.uq3[tgt]; }
            const int perb = e.Fo * P;
            e.pad0 = e.pad1 = 0;
            for (int bi = 0; bi < Bn; ++bi)
                for (int Q = 0; Q < perb; Q += W4_QUADS) {
                    e.Q0 = Q; e.b = bi;
                    t.push_back(e);
                }
        }
    }
    TileTable tt;                    // (d_tiles holds WinoTileDev entries for this key: cast at the launch site)
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc((void**)&tt.d_tiles, t.size() * sizeof(WinoTileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(WinoTileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the layer-1 F(2, 2) kernel (cdae_l1f.h): 64 consecutive output PAIRS of one batch item in the flattened (f1, pair)
// space; the four targets of a patch adjacent (they read the same whitened magnitudes: the re-reads hit the XCD's L2)
static int get_l1f_tiles(xsq_model* Mo, int Bn, int S, TileTable* out) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(1 + 160, Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = 2 * S - 1, P = (T1 + 1) / 2;
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Mo->blocks[x].nch1 > Mo->blocks[y].nch1; });
    std::vector<L1fTileDev> t;
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        L1fTileDev e;
        e.kf = d.kf; e.F = d.F; e.F1 = d.F1; e.hop = d.hop; e.nchunks = d.nch1; e.P = P;
        e.in_off = (int64_t)Bn * 2 * S * d.cum;
        const int total = Bn * d.F1 * P;                   // tiles run across batch items: (b, f1, pair) flattened
        e.pad0 = 0;
        for (int Q = 0; Q < total; Q += LF_PAIRS)
            for (int tgt = 0; tgt < NT; ++tgt) {
                e.Q0 = Q;
                e.out_off = (int64_t)CS * Bn * T1 * (4 * (int64_t)d.cumF1 + (int64_t)tgt * d.F1);
                e.shift_off = d.s1[tgt]; e.u_off = d.u1[tgt];
                t.push_back(e);
            }
    }
    TileTable tt;                    // (d_tiles holds L1fTileDev entries for this key: cast at the launch site)
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc((void**)&tt.d_tiles, t.size() * sizeof(L1fTileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(L1fTileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the layer-4 F(2, 2) kernel (cdae_l4f.h): 64 consecutive output PAIRS of one batch item in the flattened (f, pair)
// space x one column tile of <= 64 columns; one (block, target) after the other (the targets share nothing; the column tiles
// of a row block share its operand rows, neighbouring row blocks share rows through the frequency taps)
// taps = 0: the one-tap blocks (cdae_l4f_kernel); taps = 1: the multi-tap blocks, whose weight tiles of all taps fit the LDS of
// cdae_l4f_taps_kernel (kf * 3 * 16 NCB rows <= L4_RES_ROWS) -- a launch of their own; multi-tap blocks that do not fit stay with
// taps = 0 (one row tile per workgroup, weights re-staged per tap)
static bool l4f_resident(const CdaeBlockDev& d) {
    const int cols = l4f_cols(d.T);
    return d.kf > 1 && cols <= 32 && d.kf * 3 * cols <= L4_RES_ROWS;
}
static int get_l4f_tiles(xsq_model* Mo, int Bn, int S, TileTable* out, int taps = 0) {
    std::lock_guard<std::mutex> lk(Mo->mu);
    auto key = std::make_tuple(4 + 160 + 32 * taps, Bn, S);
    auto it = Mo->tiles.find(key);
    if (it != Mo->tiles.end()) { *out = it->second; return XSQ_OK; }
    const int T1 = 2 * S - 1, P = S;
    std::vector<int> order(Mo->nblocks);
    for (int b = 0; b < Mo->nblocks; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return Mo->blocks[x].kf > Mo->blocks[y].kf; });
    std::vector<L4fTileDev> t;
    for (int b : order) {
        const CdaeBlockDev& d = Mo->blocks[b];
        // XSQ_L4F_RESIDENT=1 (A/B arm, off): the multi-tap blocks on cdae_l4f_taps_kernel, a launch of their own with all taps' weight
        // tiles resident -- measured SLOWER (0.570-0.572 against 0.541-0.542 ms for the layer, profiles/r11_ab_runs.txt r11res)
        static const bool res_on = getenv("XSQ_L4F_RESIDENT") && atoi(getenv("XSQ_L4F_RESIDENT")) != 0;
        if ((l4f_resident(d) && res_on) != (taps != 0)) continue;
        const int W = d.T, cols = l4f_cols(W);
        const int64_t FST = (int64_t)d.F * S * d.T;
        L4fTileDev e;
        e.kf = d.kf; e.F = d.F; e.hop = d.hop; e.P = P; e.x_off = (int)((int64_t)Bn * 2 * S * d.cum);
        const int total = Bn * d.F * P;                    // tiles run across batch items: (b, f, pair) flattened
        const int rtiles = (total + L4_PAIRS - 1) / L4_PAIRS;
        // one tap: a workgroup keeps its column tile's weights in LDS for a run of consecutive row tiles (cdae_l4f.h); runs of
        // equal length, at most l4f_run row tiles
        static const int l4f_run = getenv("XSQ_L4F_RUN") ? std::max(1, atoi(getenv("XSQ_L4F_RUN"))) : 6;
        const int nruns = (d.kf == 1 || taps) ? (rtiles + l4f_run - 1) / l4f_run : rtiles;
        for (int tgt = 0; tgt < NT; ++tgt) {
            e.in_off = (int64_t)CS * Bn * T1 * (4 * (int64_t)d.cumF1 + (int64_t)tgt * d.F1);
            e.out_off = (int64_t)Bn * 8 * S * d.cum + (int64_t)tgt * Bn * 2 * FST;
            e.bias_off = d.b4[tgt];
            for (int r = 0; r < nruns; ++r) {
                const int t0 = (int)((int64_t)rtiles * r / nruns), t1 = (int)((int64_t)rtiles * (r + 1) / nruns);
                for (int n0 = 0; n0 < cols; n0 += 64) {
                    e.Q0 = t0 * L4_PAIRS; e.run = t1 - t0; e.n0 = n0;
                    e.u_off = d.u4[tgt] + (int64_t)3 * CS * n0;
                    t.push_back(e);
                }
            }
        }
    }
    TileTable tt;                    // (d_tiles holds L4fTileDev entries for this key: cast at the launch site)
    tt.ntiles = (int)t.size();
    XSQ_HIP(hipMalloc((void**)&tt.d_tiles, t.size() * sizeof(L4fTileDev)));
    XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(L4fTileDev), hipMemcpyHostToDevice));
    Mo->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace xsq

using namespace xsq;

extern "C" {

int64_t xsq_model_num_params(int nblocks, const int32_t* F, const int32_t* T) {
    if (!F || !T || nblocks <= 0) return XSQ_ERR_ARG;
    int64_t n = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int kf = kf_of(F[b]);
        n += 2 * F[b];
        n += NT * ((int64_t)H1 * 2 * kf * T[b] + 4 * H1 + (int64_t)H2 * H1 * kf * 4 + 4 * H2 +
                   (int64_t)H2 * H1 * kf * 4 + 4 * H1 + (int64_t)H1 * 2 * kf * T[b] + 2);
    }
    return n;
}

static int model_build(xsq_model** out, int nblocks, const int32_t* F, const int32_t* T, int causal,
                       const float* params, int64_t nparams, xsq_model** partial);

int xsq_model_create(xsq_model** out, int nblocks, const int32_t* F, const int32_t* T, int causal,
                     const float* params, int64_t nparams) {
    xsq_model* partial = nullptr;
    const int rc = model_build(out, nblocks, F, T, causal, params, nparams, &partial);
    if (rc != XSQ_OK && partial) xsq_model_destroy(partial);   // frees whatever was allocated before the failure
    return rc;
}

static int model_build(xsq_model** out, int nblocks, const int32_t* F, const int32_t* T, int causal,
                       const float* params, int64_t nparams, xsq_model** partial) {
    XSQ_REQUIRE(out && F && T && params && nblocks > 0, "xsq_model_create: null argument");
    XSQ_REQUIRE(nparams == xsq_model_num_params(nblocks, F, T),
                "xsq_model_create: got %lld parameters, the block table needs %lld", (long long)nparams,
                (long long)xsq_model_num_params(nblocks, F, T));
    struct PlanLike { std::vector<BlockHost> blocks; int nblocks; int64_t sumFT; } PL;
    PL.nblocks = nblocks; PL.sumFT = 0;
    for (int b = 0; b < nblocks; ++b) {
        XSQ_REQUIRE(F[b] >= 1 && T[b] >= 4 && T[b] % 4 == 0, "xsq_model_create: block %d has F=%d T=%d", b, F[b], T[b]);
        XSQ_REQUIRE(F[b] - 2 * (kf_of(F[b]) - 1) >= 1, "xsq_model_create: block with F=%d too small", F[b]);
        PL.blocks.push_back(BlockHost{0, F[b], T[b], PL.sumFT});
        PL.sumFT += (int64_t)F[b] * T[b];
    }
    PlanLike* P = &PL;
    xsq_model* Mo = new xsq_model();
    *partial = Mo;
    Mo->causal = causal ? 1 : 0; Mo->nblocks = nblocks; Mo->sumFT = PL.sumFT; Mo->table = PL.blocks;
    Mo->wino4 = wino4_wanted();
    const double eps = 1e-5;
    std::vector<float> pool, upool, mean, scale;
    std::vector<int64_t> cum(P->nblocks + 1, 0);
    std::vector<int> blockF(P->nblocks);
    const float* p = params;
    int cumF1 = 0, cumF2 = 0;
    int64_t cumF = 0;
    auto alloc = [&](size_t n) { size_t o = pool.size(); pool.resize(o + n, 0.f); return (int64_t)o; };
    for (int bi = 0; bi < P->nblocks; ++bi) {
        const BlockHost& hb = P->blocks[bi];
        CdaeBlockDev d;
        memset(&d, 0, sizeof(d));
        d.F = hb.F; d.T = hb.T; d.hop = hb.T / 2; d.kf = kf_of(hb.F);
        d.F1 = d.F - d.kf + 1; d.F2 = d.F1 - d.kf + 1;
        d.cumF1 = cumF1; d.cumF2 = cumF2; d.cum = hb.cum; d.cumF = cumF;
        d.ld1 = (int)round_up(2 * d.kf * d.T, 16); d.ld4 = (int)round_up(d.kf * 2 * CS, 16);
        cumF1 += d.F1; cumF2 += d.F2; cumF += d.F;
        cum[bi] = hb.cum; blockF[bi] = d.F;
        const int kf = d.kf, W = d.T, hop = d.hop;
        mean.insert(mean.end(), p, p + d.F); p += d.F;
        scale.insert(scale.end(), p, p + d.F); p += d.F;
        for (int t = 0; t < NT; ++t) {
            // ---- L1: Conv2d weight (50,2,kf,W); BN(50)
            const float* w = p; p += (size_t)H1 * 2 * kf * W;
            const float *bw = p, *bb = p + H1, *rm = p + 2 * H1, *rv = p + 3 * H1; p += 4 * H1;
            const int K1 = 2 * kf * W;
            d.w1[t] = alloc((size_t)64 * d.ld1);                 // Wt[n = co][k]
            d.s1[t] = alloc(64);
            for (int co = 0; co < H1; ++co) {
                const double s = (double)bw[co] / std::sqrt((double)rv[co] + eps);
                pool[d.s1[t] + co] = (float)((double)bb[co] - (double)rm[co] * s);
                for (int k = 0; k < K1; ++k)       // k = (ci*kf + df)*W + dt, the weight's own order
                    pool[d.w1[t] + (size_t)co * d.ld1 + k] = (float)((double)w[(size_t)co * K1 + k] * s);
            }
            // ---- F(2, 2) along the hop (cdae_l1f.h): per chunk of 16 k the three tiles W0 | W0 + W1 | W1 of the FOLDED fp32
            //      weights above; k order = (segment (ci, df), d < hop) with every segment padded to whole quads (zero weights)
            {
                d.nch1 = l1f_chunks(kf, hop);
                const int64_t u = (int64_t)upool.size();
                upool.resize(upool.size() + (size_t)d.nch1 * LF_U16, 0.f);
                d.u1[t] = u;
                const int hq = (hop + 3) / 4;
                for (int s = 0; s < d.nch1; ++s)
                    for (int kk = 0; kk < 16; ++kk) {
                        const int e4 = 4 * s + kk / 4, seg = e4 / hq, dd = 4 * (e4 % hq) + kk % 4;
                        if (seg >= 2 * kf || dd >= hop) continue;
                        for (int co = 0; co < H1; ++co) {
                            const double w0 = pool[d.w1[t] + (size_t)co * d.ld1 + (size_t)seg * W + dd];
                            const double w1v = pool[d.w1[t] + (size_t)co * d.ld1 + (size_t)seg * W + hop + dd];
                            upool[u + l1f_u_off(s, 0, co, kk)] = (float)w0;
                            upool[u + l1f_u_off(s, 1, co, kk)] = (float)(w0 + w1v);
                            upool[u + l1f_u_off(s, 2, co, kk)] = (float)w1v;
                        }
                    }
            }
            // ---- L2: Conv2d weight (51,50,kf,4); BN(51);  k = (df*4 + dt)*52 + c1
            w = p; p += (size_t)H2 * H1 * kf * 4;
            bw = p; bb = p + H2; rm = p + 2 * H2; rv = p + 3 * H2; p += 4 * H2;
            const int K2 = kf * 4 * CS;
            d.w2[t] = alloc((size_t)64 * K2);                    // K2 % 16 == 0
            d.s2[t] = alloc(64);
            for (int co = 0; co < H2; ++co) {
                const double s = (double)bw[co] / std::sqrt((double)rv[co] + eps);
                pool[d.s2[t] + co] = (float)((double)bb[co] - (double)rm[co] * s);
                for (int ci = 0; ci < H1; ++ci)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            pool[d.w2[t] + (size_t)co * K2 + (df * 4 + dt) * CS + ci] =
                                (float)((double)w[(((size_t)co * H1 + ci) * kf + df) * 4 + dt] * s);
            }
            // ---- L3: ConvTranspose2d weight (51,50,kf,4) = (in,out,kH,kW); BN(50)
            //      out3[c3,f3,t3] = sum w[c2,c3,df,dt] out2[c2,f3-df,t3-dt];  k = (df*4 + dt')*52 + c2, dt' = 3-dt
            w = p; p += (size_t)H2 * H1 * kf * 4;
            bw = p; bb = p + H1; rm = p + 2 * H1; rv = p + 3 * H1; p += 4 * H1;
            d.w3[t] = alloc((size_t)64 * K2);
            d.s3[t] = alloc(64);
            for (int co = 0; co < H1; ++co) {
                const double s = (double)bw[co] / std::sqrt((double)rv[co] + eps);
                pool[d.s3[t] + co] = (float)((double)bb[co] - (double)rm[co] * s);
                for (int ci = 0; ci < H2; ++ci)
                    for (int df = 0; df < kf; ++df)
                        for (int dt = 0; dt < 4; ++dt)
                            pool[d.w3[t] + (size_t)co * K2 + (df * 4 + (3 - dt)) * CS + ci] =
                                (float)((double)w[(((size_t)ci * H1 + co) * kf + df) * 4 + dt] * s);
            }
            // ---- Winograd F(2, 4) along the time taps (cdae_wino.h): U_j = sum_dt G[j][dt] w[dt] of the FOLDED fp32 weights above
            //      (what the direct kernels contract with), summed in fp64, as [df][chunk][component][col < 51][k] tiles
            for (int layer = 2; layer <= 3; ++layer) {
                const int64_t wsrc = layer == 2 ? d.w2[t] : d.w3[t];
                const int64_t u = (int64_t)upool.size();
                upool.resize(upool.size() + (size_t)kf * WN_UDF, 0.f);
                (layer == 2 ? d.u2[t] : d.u3[t]) = u;
                for (int df = 0; df < kf; ++df)
                    for (int j = 0; j < 5; ++j)
                        for (int col = 0; col < WN_COLS; ++col)
                            for (int ci = 0; ci < CS; ++ci) {
                                double acc = 0.0;
                                for (int dt = 0; dt < 4; ++dt)
                                    acc += WN_G[j][dt] * (double)pool[wsrc + (size_t)col * K2 + (df * 4 + dt) * CS + ci];
                                upool[u + (size_t)df * WN_UDF + wino_u_off(j, col, ci)] = (float)acc;
                            }
            }
            // ---- Winograd F(4, 4) along the time taps (cdae_wino4.h), the same way: seven components
            for (int layer = 2; layer <= 3; ++layer) {
                (layer == 2 ? d.uq2[t] : d.uq3[t]) = -1;
                if (!Mo->wino4) continue;
                const int64_t wsrc = layer == 2 ? d.w2[t] : d.w3[t];
                const int64_t u = (int64_t)upool.size();
                upool.resize(upool.size() + (size_t)kf * W4_UDF, 0.f);
                (layer == 2 ? d.uq2[t] : d.uq3[t]) = u;
                for (int df = 0; df < kf; ++df)
                    for (int j = 0; j < W4_NC; ++j)
                        for (int col = 0; col < WN_COLS; ++col)
                            for (int ci = 0; ci < CS; ++ci) {
                                double acc = 0.0;
                                for (int dt = 0; dt < 4; ++dt)
                                    acc += W4_G[j][dt] * (double)pool[wsrc + (size_t)col * K2 + (df * 4 + dt) * CS + ci];
                                upool[u + (size_t)df * W4_UDF + wino4_u_off(j, col, ci)] = (float)acc;
                            }
            }
            // ---- L4: ConvTranspose2d weight (50,2,kf,W) = (in,out,kH,kW); bias(2)
            //      k = (df*2 + (1 - tap))*52 + c3 (tap 1 first: CdaeL4Op) ;  n = c*hop + dtlo ;  kernel column = dtlo + tap*hop
            w = p; p += (size_t)H1 * 2 * kf * W;
            const float* bias = p; p += 2;
            d.w4[t] = alloc((size_t)round_up(W, 64) * d.ld4);    // Wt[n = c*hop + dtlo][k]
            d.b4[t] = alloc(64);
            pool[d.b4[t]] = bias[0]; pool[d.b4[t] + 1] = bias[1];
            for (int ci = 0; ci < H1; ++ci)
                for (int c = 0; c < 2; ++c)
                    for (int df = 0; df < kf; ++df)
                        for (int tap = 0; tap < 2; ++tap)
                            for (int dt = 0; dt < hop; ++dt)
                                pool[d.w4[t] + (size_t)(c * hop + dt) * d.ld4 + (df * 2 + (1 - tap)) * CS + ci] =
                                    w[(((size_t)ci * 2 + c) * kf + df) * W + dt + tap * hop];
            // ---- F(2, 2) along the hop (cdae_l4f.h): per frequency tap and column tile the three tiles Wb | Wa + Wb | Wa
            //      ([component][column][52 k]; Wa = kernel columns dt < hop, Wb = columns dt + hop), summed in fp64
            {
                const int64_t u = (int64_t)upool.size();
                const int tapf = l4f_tap_floats(W), cols = l4f_cols(W);
                upool.resize(upool.size() + (size_t)kf * tapf, 0.f);
                d.u4[t] = u;
                for (int df = 0; df < kf; ++df)
                    for (int n0 = 0; n0 < cols; n0 += 64)
                        for (int n = n0; n < std::min(cols, n0 + 64) && n < W; ++n)
                            for (int ci = 0; ci < H1; ++ci) {
                                const double wa = pool[d.w4[t] + (size_t)n * d.ld4 + (df * 2 + 1) * CS + ci];
                                const double wb = pool[d.w4[t] + (size_t)n * d.ld4 + (df * 2 + 0) * CS + ci];
                                const size_t o = (size_t)u + (size_t)df * tapf;
                                upool[o + l4f_u_off(W, n0, 0, n, ci)] = (float)wb;
                                upool[o + l4f_u_off(W, n0, 1, n, ci)] = (float)(wa + wb);
                                upool[o + l4f_u_off(W, n0, 2, n, ci)] = (float)wa;
                            }
            }
        }
        Mo->blocks.push_back(d);
    }
    cum[P->nblocks] = P->sumFT;
    Mo->sumF = cumF; Mo->sumF1 = cumF1; Mo->sumF2 = cumF2;
    if (p - params != nparams) {
        set_error("xsq_model_create: internal parameter walk mismatch");
        return XSQ_ERR_ARG;
    }
#define UP(dst, vec, T)                                                                           \
    do {                                                                                          \
        XSQ_HIP(hipMalloc(&(dst), (vec).size() * sizeof(T)));                                     \
        XSQ_HIP(hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice)); \
    } while (0)
    UP(Mo->d_pool, pool, float);
    UP(Mo->d_upool, upool, float);
    Mo->pool_floats = (int64_t)pool.size();
    UP(Mo->d_mean, mean, float);
    UP(Mo->d_scale, scale, float);
    UP(Mo->d_blocks, Mo->blocks, CdaeBlockDev);
    UP(Mo->d_cum, cum, int64_t);
    UP(Mo->d_blockF, blockF, int);
#undef UP
    *partial = nullptr;
    *out = Mo;
    return XSQ_OK;
}

int xsq_model_set_precision(xsq_model* Mo, int mode) {
    XSQ_REQUIRE(Mo, "xsq_model_set_precision: null model");
    XSQ_REQUIRE(mode >= 0 && mode <= 2, "xsq_model_set_precision: mode %d (0 = fp32, 1 = bf16x3, 2 = bf16x6)", mode);
    if (mode == 1 && !Mo->d_pool_split) {      // one-off: the weight pool in the split operand format
        XSQ_HIP(hipMalloc(&Mo->d_pool_split, (size_t)Mo->pool_floats * 4));
        hipLaunchKernelGGL(k_split_pool, dim3((unsigned)((Mo->pool_floats + 255) / 256)), dim3(256), 0, 0, Mo->d_pool,
                           Mo->d_pool_split, Mo->pool_floats);
        XSQ_HIP(hipDeviceSynchronize());
    }
    Mo->precision = mode;
    return XSQ_OK;
}

#if XSQ_WINO_STAMPS
extern "C" int xsq_debug_wn_stamps(unsigned long long* out, int reset) {     // diagnostic builds only (cdae_wino.h)
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(xsq::g_wn_stamps), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; XSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(xsq::g_wn_stamps), z, sizeof(z))); }
    return XSQ_OK;
}
#endif
#if XSQ_WINO4_STAMPS
extern "C" int xsq_debug_w4_stamps(unsigned long long* out, int reset) {     // diagnostic builds only (cdae_wino4.h)
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(xsq::g_w4_stamps), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; XSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(xsq::g_w4_stamps), z, sizeof(z))); }
    return XSQ_OK;
}
#endif

int xsq_model_set_winograd(xsq_model* Mo, int on) {
    XSQ_REQUIRE(Mo && on >= 0 && on <= 15, "xsq_model_set_winograd: null model or mask %d (1 = layers 2 / 3 Winograd F(2, 4), 2 / 4 = layer 1 / 4 F(2, 2), 8 = F(4, 4) for long rows)", on);
    XSQ_REQUIRE(!(on & 8) || Mo->wino4, "xsq_model_set_winograd: bit 8 (F(4, 4)) needs a model created with XSQ_WINO4=1 in the environment");
    Mo->winograd = on;
    return XSQ_OK;
}

int xsq_model_destroy(xsq_model* Mo) {
    if (!Mo) return XSQ_OK;
    for (auto& kv : Mo->tiles) (void)hipFree(kv.second.d_tiles);
    (void)hipFree(Mo->d_pool); (void)hipFree(Mo->d_upool); (void)hipFree(Mo->d_pool_split); (void)hipFree(Mo->d_mean); (void)hipFree(Mo->d_scale);
    (void)hipFree(Mo->d_blocks); (void)hipFree(Mo->d_cum); (void)hipFree(Mo->d_blockF);
    delete Mo;
    return XSQ_OK;
}

// workspace: xin | act1 | act2 | act3
size_t xsq_cdae_workspace(const xsq_model* Mo, int Bn, int S) {
    if (!Mo || Bn <= 0 || S < 3) return 0;
    const int64_t T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    return al((size_t)Bn * 2 * S * Mo->sumFT * 4) + 2 * al((size_t)CS * Bn * T1 * 4 * Mo->sumF1 * 4) +
           al((size_t)CS * Bn * T2 * 4 * Mo->sumF2 * 4) + 256;
}

}  // extern "C"

namespace xsq {

int cdae_launch_magnitude(const xsq_model* Mo, const float* X, float* xin, const float* mean, const float* scale,
                          int Bn, int S, hipStream_t stream, int split) {
    const int64_t total = (int64_t)Bn * 2 * S * Mo->sumFT;
    XSQ_PROF("magnitude_whiten", stream);
    hipLaunchKernelGGL(k_magnitude_whiten, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream,
                       (const float4*)X, (float4*)xin, Mo->d_cum, Mo->d_blocks, mean, scale, Mo->nblocks, Bn * 2, S,
                       total / 4, split);
    return XSQ_OK;
}

int cdae_launch_layer(xsq_model* Mo, int layer, const CdaeArgs& a, hipStream_t stream, const char* prof_name) {
    TileTable tt;
    // layers 1 and 4 read their operand rows through buffer descriptors with 32-bit byte offsets and out-of-range switches
    // (common.h), layer 4 stores its masks that way: a block's whitened magnitudes / masks of one target and its layer-3
    // activations stay below 2^30 bytes
    if (layer == 1 || layer == 4)
        for (const CdaeBlockDev& d : Mo->blocks)
            XSQ_REQUIRE(std::max((int64_t)4 * 2 * a.Bn * d.F * a.S * d.T, layer == 4 ? (int64_t)4 * CS * a.Bn * a.T1 * d.F1 : 0) < ((int64_t)1 << 30),
                        "xsq_cdae_forward: B=%d S=%d: a block's layer-%d input exceeds 2^30 bytes; split the batch", a.Bn, a.S, layer);
    if (layer == 4 && !a.gx8)
        for (const CdaeBlockDev& d : Mo->blocks)
            XSQ_REQUIRE((int64_t)2 * a.Bn * d.F * a.S * d.T < ((int64_t)1 << 31),
                        "xsq_cdae_forward: B=%d S=%d overflows the 32-bit offsets of a block; split the batch", a.Bn, a.S);
    if (layer == 4 && a.gx8) layer = 6;
    const bool bf3 = a.split != 0;      // set by xsq_cdae_forward (inference only); operands are in the split format
    const bool bf1 = Mo->precision == 3;       // training only (xsq_train_set_precision mode 1): operands rounded to bf16, one MFMA per product
    const bool bf6 = Mo->precision == 2 || bf1; // fp32 operands, cut (or rounded) in the kernel (any operator / epilogue: also the training step)
    // Diagnostic A/B switches (default 0 = the product configuration; results stay correct in every setting):
    //   1 bf16x3 generic engine with 256-row tiles      2 ... with 32-value K-steps      4 no slab kernels at all
    //   8 no slab kernels on the fp32 path              64 fp32 slab kernels padded to 64 columns (MODE 0)
    //   128 generic fp32 engine padded to 64 columns    256 layer 4 with 32/64-column tiles only
    //   512 fp32 slab kernels request the next slab a whole df ahead (28 VGPRs held across the MFMA slots, spilled: r01)
    static const int variant = getenv("XSQ_CDAE_VARIANT") ? atoi(getenv("XSQ_CDAE_VARIANT")) : 0;
    const int mt23 = bf3 && (variant & 1) ? 2 : L23_MT;
    const bool xw = !bf3 && !bf6 && !a.raw && !a.xin8 && !a.gx8 && layer <= 3 && !(variant & 128);   // fp32 inference: no column padding
    if (!(variant & 4) && (layer == 2 || layer == 3) && (layer == 2 ? a.T2 : a.T1) >= 86 && !a.raw && !a.xin8 && !a.gx8 &&
        (bf3 || bf6 || !(variant & 8))) {
        // slab kernels: the tile's distinct input positions held once in LDS (cdae_slab.h); they address a (block, target)'s
        // input through 32-bit float offsets
        for (const CdaeBlockDev& d : Mo->blocks)
            XSQ_REQUIRE((int64_t)4 * CS * a.Bn * a.T1 * d.F1 < ((int64_t)1 << 30), "xsq_cdae_forward: B=%d S=%d overflows the 32-bit "
                        "offsets of a block's activations; split the batch", a.Bn, a.S);
        // fp32: Winograd F(2, 4) along the four time taps (cdae_wino.h) -- 5 instead of 8 MFMA products per output pair;
        // rows of >= 64 pairs, i.e. To >= 127.  xsq_model_set_winograd(0) / XSQ_CDAE_VARIANT=2048: the direct slab kernels.
        if (!bf3 && !bf6 && a.upool && (Mo->winograd & 1) && !(variant & 2048) && ((layer == 2 ? a.T2 : a.T1) + 1) / 2 >= WN_PAIRS) {
            // bit 8: F(4, 4) (cdae_wino4.h) -- 7 products per output quad instead of 10; rows of >= 64 quads, i.e. To >= 253
            if ((Mo->winograd & 8) && Mo->wino4 && ((layer == 2 ? a.T2 : a.T1) + 3) / 4 >= W4_QUADS) {
                int rcq = get_wino4_tiles(Mo, layer, a.Bn, a.S, &tt);
                if (rcq) return rcq;
                XSQ_PROF(prof_name ? prof_name : (layer == 2 ? "cdae_l2_slab" : "cdae_l3_slab"), stream);
                if (layer == 2) hipLaunchKernelGGL((cdae_wino4_kernel<false>), dim3(tt.ntiles), dim3(512), 0, stream, a, (const WinoTileDev*)tt.d_tiles, tt.ntiles);
                else hipLaunchKernelGGL((cdae_wino4_kernel<true>), dim3(tt.ntiles), dim3(512), 0, stream, a, (const WinoTileDev*)tt.d_tiles, tt.ntiles);
                return XSQ_OK;
            }
            // XSQ_WINO_MIN_KF (A/B): blocks with fewer frequency taps stay on the direct kernel (their tiles are short: a
            // prologue per 64 pairs and tap), the Winograd kernel takes the rest
            static const int min_kf = getenv("XSQ_WINO_MIN_KF") ? atoi(getenv("XSQ_WINO_MIN_KF")) : 0;
            int rcw = get_wino_tiles(Mo, layer, a.Bn, a.S, &tt, min_kf);
            if (rcw) return rcw;
            XSQ_PROF(prof_name ? prof_name : (layer == 2 ? "cdae_l2_slab" : "cdae_l3_slab"), stream);
            if (layer == 2) hipLaunchKernelGGL((cdae_wino_kernel<false>), dim3(tt.ntiles), dim3(256), 0, stream, a, (const WinoTileDev*)tt.d_tiles, tt.ntiles);
            else hipLaunchKernelGGL((cdae_wino_kernel<true>), dim3(tt.ntiles), dim3(256), 0, stream, a, (const WinoTileDev*)tt.d_tiles, tt.ntiles);
            if (min_kf > 1) {
                TileTable ts;
                if ((rcw = get_slab_tiles(Mo, layer, a.Bn, a.S, &ts, min_kf - 1))) return rcw;
                if (layer == 2) hipLaunchKernelGGL((cdae_slab_kernel<false, 3, true>), dim3(ts.ntiles), dim3(512), 0, stream, a, (const SlabTileDev*)ts.d_tiles, ts.ntiles);
                else hipLaunchKernelGGL((cdae_slab_kernel<true, 3, true>), dim3(ts.ntiles), dim3(512), 0, stream, a, (const SlabTileDev*)ts.d_tiles, ts.ntiles);
            }
            return XSQ_OK;
        }
        int rc = get_slab_tiles(Mo, layer, a.Bn, a.S, &tt);
        if (rc) return rc;
        XSQ_PROF(prof_name ? prof_name : (layer == 2 ? "cdae_l2_slab" : "cdae_l3_slab"), stream);      // its own event name: one kernel, one name
#define XSQ_SLAB(TR_, MODE_) hipLaunchKernelGGL((cdae_slab_kernel<TR_, MODE_>), dim3(tt.ntiles), dim3(512), 0, stream, a, (const SlabTileDev*)tt.d_tiles, tt.ntiles)
        const bool exw = !(variant & 64);      // fp32: exact-width columns (MODE 3) unless switched back to MODE 0
#define XSQ_SLAB_LATE(TR_) hipLaunchKernelGGL((cdae_slab_kernel<TR_, 3, true>), dim3(tt.ntiles), dim3(512), 0, stream, a, (const SlabTileDev*)tt.d_tiles, tt.ntiles)
        const bool late = !(variant & 512);      // fp32 exact-width kernel: next slab fetched in its own slot (no registers held across the MFMA slots: no scratch)
        if (layer == 2) { if (bf3) XSQ_SLAB(false, 1); else if (bf6) XSQ_SLAB(false, 2); else if (exw && late) XSQ_SLAB_LATE(false); else if (exw) XSQ_SLAB(false, 3); else XSQ_SLAB(false, 0); }
        else { if (bf3) XSQ_SLAB(true, 1); else if (bf6) XSQ_SLAB(true, 2); else if (exw && late) XSQ_SLAB_LATE(true); else if (exw) XSQ_SLAB(true, 3); else XSQ_SLAB(true, 0); }
#undef XSQ_SLAB_LATE
#undef XSQ_SLAB
        return XSQ_OK;
    }
    if (layer == 1 && !bf3 && !bf6 && !a.raw && !a.xin8 && !a.gx8 && !a.causal && a.upool && (Mo->winograd & 2)) {
        // fp32 inference, non-causal: F(2, 2) along the hop (cdae_l1f.h) -- three half-window products per output pair
        // instead of four.  xsq_model_set_winograd without bit 2: the implicit GEMM below.
        int rcf = get_l1f_tiles(Mo, a.Bn, a.S, &tt);
        if (rcf) return rcf;
        XSQ_PROF(prof_name ? prof_name : "cdae_l1_gemm", stream);
        hipLaunchKernelGGL(cdae_l1f_kernel, dim3(tt.ntiles), dim3(256), 0, stream, a, (const L1fTileDev*)tt.d_tiles, tt.ntiles);
        return XSQ_OK;
    }
    bool l4f_fits = true;            // cdae_l4f.h runs several frequency taps only on column tiles of <= 32 columns
    for (const CdaeBlockDev& d : Mo->blocks) l4f_fits = l4f_fits && (d.kf == 1 || d.T <= 32);
    if (layer == 4 && l4f_fits && !bf3 && !bf6 && !a.raw && !a.xin8 && !a.gx8 && !a.causal && (a.Y || a.masks) && a.upool && (Mo->winograd & 4)) {
        // fp32 inference, non-causal: F(2, 2) along the hop (cdae_l4f.h) -- masks only (the separator's path) or with the
        // estimates materialised (the module API): one kernel, the same masks bit for bit
        for (const CdaeBlockDev& d : Mo->blocks)
            XSQ_REQUIRE((int64_t)8 * 2 * a.Bn * d.F * a.S * d.T < ((int64_t)1 << 31) && (int64_t)a.Bn * 2 * a.S * d.cum < ((int64_t)1 << 31),
                        "xsq_cdae_forward: B=%d S=%d overflows the 32-bit offsets of a block's coefficients; split the batch", a.Bn, a.S);
        int rcf = get_l4f_tiles(Mo, a.Bn, a.S, &tt);
        if (rcf) return rcf;
        TileTable t2;
        if ((rcf = get_l4f_tiles(Mo, a.Bn, a.S, &t2, 1))) return rcf;
        XSQ_PROF(prof_name ? prof_name : "cdae_l4_gemm", stream);
        if (t2.ntiles) {           // the multi-tap blocks first (the longest tiles of the layer)
            if (a.Y) hipLaunchKernelGGL(cdae_l4f_taps_kernel<true>, dim3(t2.ntiles), dim3(256), 0, stream, a, (const L4fTileDev*)t2.d_tiles, t2.ntiles);
            else hipLaunchKernelGGL(cdae_l4f_taps_kernel<false>, dim3(t2.ntiles), dim3(256), 0, stream, a, (const L4fTileDev*)t2.d_tiles, t2.ntiles);
        }
        if (tt.ntiles) {
            if (a.Y) hipLaunchKernelGGL(cdae_l4f_kernel<true>, dim3(tt.ntiles), dim3(256), 0, stream, a, (const L4fTileDev*)tt.d_tiles, tt.ntiles);
            else hipLaunchKernelGGL(cdae_l4f_kernel<false>, dim3(tt.ntiles), dim3(256), 0, stream, a, (const L4fTileDev*)tt.d_tiles, tt.ntiles);
        }
        return XSQ_OK;
    }
    const bool n16 = !bf3 && !bf6 && layer == 4 && !a.raw && !a.xin8 && !a.gx8 && !(variant & 256);      // fp32 inference: 16-column granularity
    const bool tgt_outer = layer == 4 && !a.Y && !a.gx8;        // masks only (the separator's path, also with Wiener-EM from the masks)
    int rc = get_cdae_tiles(Mo, layer, a.Bn, a.S, &tt, mt23, n16, tgt_outer);
    if (rc) return rc;
#define XSQ_LAUNCH(OP, MT_, XW_)                                                                                    \
    do {                                                                                                            \
        if (bf1) hipLaunchKernelGGL((grouped_gemm_bf6_kernel<OP, true>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles);      \
        else if (bf6) hipLaunchKernelGGL((grouped_gemm_bf6_kernel<OP>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles);      \
        else if (!bf3 && xw) hipLaunchKernelGGL((grouped_gemm_kernel<OP, 1, XW_>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles); \
        else if (!bf3) hipLaunchKernelGGL((grouped_gemm_kernel<OP, MT_>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles);         \
        else if (variant & 2) hipLaunchKernelGGL((grouped_gemm_bf3_kernel<OP, MT_, 2>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles); \
        else hipLaunchKernelGGL((grouped_gemm_bf3_kernel<OP, MT_, 1>), dim3(tt.ntiles), dim3(256), 0, stream, OP{a}, tt.d_tiles, tt.ntiles); \
    } while (0)
    switch (layer) {
        case 1: { XSQ_PROF(prof_name ? prof_name : "cdae_l1_gemm", stream); if (a.causal) XSQ_LAUNCH(CdaeL1CausalOp, 1, 1); else XSQ_LAUNCH(CdaeL1Op, 1, 1); } break;
        case 2: { XSQ_PROF(prof_name ? prof_name : "cdae_l2_gemm", stream); if (mt23 == 2) XSQ_LAUNCH(CdaeL2Op, 2, 0); else XSQ_LAUNCH(CdaeL2Op, 1, 1); } break;
        case 3: { XSQ_PROF(prof_name ? prof_name : "cdae_l3_gemm", stream); if (mt23 == 2) XSQ_LAUNCH(CdaeL3Op, 2, 0); else XSQ_LAUNCH(CdaeL3Op, 1, 1); } break;
        default: { XSQ_PROF(prof_name ? prof_name : "cdae_l4_gemm", stream);
            if (n16) hipLaunchKernelGGL((grouped_gemm_kernel<CdaeL4Op, 1, 2>), dim3(tt.ntiles), dim3(256), 0, stream, CdaeL4Op{a}, tt.d_tiles, tt.ntiles);
            else XSQ_LAUNCH(CdaeL4Op, 1, 0); } break;
    }
#undef XSQ_LAUNCH
    return XSQ_OK;
}

}  // namespace xsq

#if XSQ_SLAB_STAMP
extern "C" int xsq_debug_slab_stamps(unsigned long long* host, int tiles) {
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_slab_stamps), (size_t)(tiles < SLAB_STAMP_TILES ? tiles : SLAB_STAMP_TILES) * 64));
    XSQ_HIP(hipMemcpyFromSymbol(host + (size_t)SLAB_STAMP_TILES * 8, HIP_SYMBOL(g_slab_stamps2), (size_t)SLAB_STAMP_TILES * 32));
    int occ = 0;
    XSQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (cdae_slab_kernel<true, 3, true>), 512, 0));
    return occ;
}
#endif
#if XSQ_GEMM_STAMP
extern "C" int xsq_debug_occupancy(int* out) {       // the runtime's own count of resident workgroups per CU
    XSQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], (grouped_gemm_kernel<CdaeL4Op, 1, 2>), 256, 0));
    XSQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], (grouped_gemm_kernel<CdaeL1Op, 1, 1>), 256, 0));
    XSQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[2], (cdae_slab_kernel<true, 3, true>), 512, 0));
    hipFuncAttributes fa;
    XSQ_HIP(hipFuncGetAttributes(&fa, (const void*)(grouped_gemm_kernel<CdaeL4Op, 1, 2>)));
    out[3] = fa.numRegs; out[4] = (int)fa.sharedSizeBytes; out[5] = (int)fa.localSizeBytes; out[6] = fa.maxThreadsPerBlock;
    return XSQ_OK;
}
extern "C" int xsq_debug_gemm_stamps(unsigned long long* host, int tiles) {
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), (size_t)(tiles < GEMM_STAMP_TILES ? tiles : GEMM_STAMP_TILES) * 64));
    return XSQ_OK;
}
#endif

extern "C" {

int xsq_model_whitening(xsq_model* Mo, const float** mean, const float** scale, int* split) {
    XSQ_REQUIRE(Mo && mean && scale && split, "xsq_model_whitening: null argument");
    *mean = Mo->d_mean; *scale = Mo->d_scale; *split = Mo->precision == 1 ? 1 : 0;
    return XSQ_OK;
}

int xsq_cdae_forward(xsq_model* Mo, const float* X, int Bn, int S, float* Y, float* masks, void* ws,
                     size_t ws_bytes, void* stream_) {
    return xsq_cdae_forward_xin(Mo, X, Bn, S, Y, masks, ws, ws_bytes, stream_, 0);
}

int xsq_cdae_forward_xin(xsq_model* Mo, const float* X, int Bn, int S, float* Y, float* masks, void* ws,
                         size_t ws_bytes, void* stream_, int xin_ready) {
    XSQ_REQUIRE(Mo && X && (Y || masks) && ws, "xsq_cdae_forward: null argument");
    XSQ_REQUIRE(Bn > 0 && S >= 3, "xsq_cdae_forward: Bn=%d S=%d (the conv stack needs >= 3 slices)", Bn, S);
    XSQ_REQUIRE(ws_bytes >= xsq_cdae_workspace(Mo, Bn, S), "xsq_cdae_forward: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int T1 = Mo->causal ? 2 * S : 2 * S - 1, T2 = T1 - 3;
    char* w = (char*)ws;
    float* xin = (float*)w;  w += al((size_t)Bn * 2 * S * Mo->sumFT * 4);
    float* act1 = (float*)w; w += al((size_t)CS * Bn * T1 * 4 * Mo->sumF1 * 4);
    float* act3 = (float*)w; w += al((size_t)CS * Bn * T1 * 4 * Mo->sumF1 * 4);
    float* act2 = (float*)w;
    const int split = Mo->precision == 1 ? 1 : 0;
    if (!xin_ready) cdae_launch_magnitude(Mo, X, xin, Mo->d_mean, Mo->d_scale, Bn, S, stream, split);
    CdaeArgs a{Mo->d_blocks, Mo->d_pool, xin, act1, act2, act3, X, Y, masks, Bn, S, T1, T2, Mo->causal, 0, nullptr, nullptr};
    a.split = split;
    a.poolB = split ? Mo->d_pool_split : nullptr;
    a.upool = Mo->d_upool;
    for (int layer = 1; layer <= 4; ++layer) {
        const int rc = cdae_launch_layer(Mo, layer, a, stream);
        if (rc) return rc;
    }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

}  // extern "C"
