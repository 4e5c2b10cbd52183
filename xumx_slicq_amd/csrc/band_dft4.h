// Per-band DFT with one radix-4 decimation-in-frequency stage fused into the operand staging.
//
// Every band length is a multiple of 4 (Lg = 4m).  For X[q] = sum_t x[t] w^(q t), w = exp(-+2 pi i / Lg):
//   X[4k + r] = sum_{t1 < m}  y_r[t1] * exp(-+2 pi i k t1 / m),
//   y_r[t1]   = w^(r t1) * sum_{a < 4} x[t1 + a m] * (-+i)^(r a)          (r = 0..3)
// i.e. four length-m DFTs that share ONE m x m matrix: 4x fewer MFMA flops than the dense Lg x Lg
// product of the generic engine (gemm_tile.h), which keeps the bands with Lg < 64 (5 % of the flops).
// The four quarters x[t1 + a m] are contiguous in memory, so the butterflies cost 8 independent 8-byte
// loads per lane and K-step and happen in registers on the way into LDS.
//
// One workgroup = 4 wavefronts; wave r owns residue r for a tile of 64 rows x 64 real columns (32 k's):
// all four waves read the same B fragments (DFT_m) and their own A fragments (y_r), 32 MFMAs per wave and
// K-step between barriers.  LDS: 2 x (4 x 64 + 64) rows x 20 floats = 51,200 B -> 3 workgroups per CU.
//
// FWD (analysis, nsgt/nsgtf.py:50-81 closed form F*):  x[q] = g'[q] * U~[bin0 + (q + Lg/2) mod Lg]
//      (window, sign and 1/Lg folded into g'; Hermitian reflection outside [0, L/2]), inverse-DFT sign,
//      output = coefficient t = 4k + r of the arena row.
// INV (synthesis, nsgt/nsigtf.py:85-95 closed form I2): x = coefficient row, forward-DFT sign, output
//      X[q] * wd[q] (dual window, Lg, sign, 1/L) at spectrum position p = (q + Lg/2) mod Lg of the
//      phase-ordered row (slice_fft.h) or of the arena row (rocFFT fallback).
#pragma once
#include "common.h"
#include "gemm_tile.h"

namespace xsq {

struct Band4Dev {
    int Lg, m, bin0, f, F, ent;
    int ldd;            // row length of the transposed DFT_m matrix: round_up(2m, 16)
    int pad;
    int64_t cum;        // arena offset of the band's block (complex per channel-slice)
    int64_t d_off;      // float offset of Dt[n = (k, re/im)][kk = (t1, re/im)] inside the direction's pool
    int64_t tw_off;     // float offset of the twiddles w^(r t1), r = 1..3: [3][round_up(m, 8)] complex
    int64_t win_off;    // float offset of g' (FWD) / wd (INV), Lg floats in window order q
};

struct Band4Args {
    const Band4Dev* bands;
    const float* pool;      // matrices, twiddles and windows of this direction
    const float* src;       // FWD: U (rows x nbins complex)      INV: coefficient arena
    float* dst;             // FWD: coefficient arena              INV: Z (row-major phase-ordered, or arena)
    int BC, S, nbins, L;
    int row_len;            // INV: > 0 -> row-major phase-ordered output with rows of row_len complex
    // INV, optional: the coefficients are mask * mix, formed on the way in (the separator's mix-phase path:
    // the CDAE then writes only the real masks).  mask = real arena with BC channels; src = the mix arena
    // with BCx channels, coefficient channel bc reads mix channel bc % BCx.
    const float* mask;
    int BCx;
};

constexpr int D4_BM = 64, D4_LD = 20, D4_MPAD = 80;     // bands up to Lg = 320 (the plan builder routes longer ones to the dense engine)

#ifndef XSQ_D4_WPE
#define XSQ_D4_WPE __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
template <bool FWD>
__global__ __launch_bounds__(256) XSQ_D4_WPE void band_dft4_kernel(Band4Args a, const TileDev* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (4 * D4_BM + 64) * D4_LD];
    __shared__ __attribute__((aligned(16))) float2 twl[3 * D4_MPAD];      // w^(r t1), r = 1..3, of this tile's band
    float* const As0 = lds;                               // [buf][r][row][20]
    float* const Bs0 = lds + 2 * 4 * D4_BM * D4_LD;       // [buf][col][20]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const bool wide = t.narrow == 0;
    const Band4Dev& bd = a.bands[t.group];
    const int m_ = bd.m, Lg = bd.Lg, K = 2 * m_, M = a.BC * a.S;
    const int64_t BCS = (int64_t)a.BC * a.S;

    // ---- staging assignment: row s_row, complex pair 2*s_kq, 2*s_kq+1 of the K-step -------------
    const int s_row = tid >> 2, s_kq = tid & 3;
    const int row = t.m0 + s_row;
    const bool row_ok = row < M;
    const int bc = row / a.S, s = row - bc * a.S;
    const float* xrow = nullptr;
    const float* mrow = nullptr;
    if (row_ok) {
        if (FWD) xrow = a.src + (int64_t)row * 2 * a.nbins;
        else if (!a.mask) xrow = a.src + 2 * (BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + s) * Lg);
        else {
            mrow = a.mask + (BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + s) * Lg);
            xrow = a.src + 2 * ((int64_t)a.BCx * a.S * bd.cum + (((int64_t)(bc % a.BCx) * bd.F + bd.f) * a.S + s) * Lg);
        }
    }
    const float* win = a.pool + bd.win_off;
    const int mpad = (m_ + 7) & ~7;
    const float2* tw = reinterpret_cast<const float2*>(a.pool + bd.tw_off);
    for (int i = tid; i < 3 * mpad; i += 256) twl[i] = tw[i];      // read back in store_set, after the prologue barrier below
    const float* bp = a.pool + bd.d_off + (int64_t)(t.n0 + s_row) * bd.ldd + 4 * s_kq;
    const bool b_on = wide || s_row < 32;

    float2 gx[4][2];       // quarters a = 0..3, two consecutive t1
    float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
    int g_t1 = 0;          // t1 of the staged pair (twiddles are read from LDS in store_set)
    struct __attribute__((aligned(8))) F4 { float x, y, z, w; };     // two consecutive complex values, 8-byte aligned
    struct __attribute__((aligned(4))) F2 { float x, y; };           // two consecutive masks

    auto load_set = [&](int k0) {      // k0 = first real k of the K-step (16 per step = 8 complex t1)
        const int t1 = (k0 >> 1) + 2 * s_kq;
        g_t1 = t1;
        if (!FWD) {       // synthesis: the pair (t1, t1 + 1) of every quarter is one 16-byte load (+ one 8-byte mask load)
            const bool ok0 = row_ok && t1 < m_, ok1 = row_ok && t1 + 1 < m_;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                gx[q4][0] = gx[q4][1] = make_float2(0.f, 0.f);
                const float* p = xrow + 2 * (t1 + q4 * m_);
                if (ok1) { const F4 v = *reinterpret_cast<const F4*>(p); gx[q4][0] = make_float2(v.x, v.y); gx[q4][1] = make_float2(v.z, v.w); }
                else if (ok0) gx[q4][0] = *reinterpret_cast<const float2*>(p);
                if (mrow) {
                    const float* pm = mrow + t1 + q4 * m_;
                    float mk0 = 0.f, mk1 = 0.f;
                    if (ok1) { const F2 v = *reinterpret_cast<const F2*>(pm); mk0 = v.x; mk1 = v.y; }
                    else if (ok0) mk0 = pm[0];
                    gx[q4][0].x *= mk0; gx[q4][0].y *= mk0; gx[q4][1].x *= mk1; gx[q4][1].y *= mk1;
                }
            }
            if (b_on) gb = *reinterpret_cast<const float4*>(bp + k0);
            return;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tt = t1 + e;
            const bool ok = row_ok && tt < m_;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                float2 v = make_float2(0.f, 0.f);
                if (ok) {
                    if (FWD) {
                        const int q = tt + q4 * m_;                        // window index
                        int idx = bd.bin0 + tt + ((q4 + 2) & 3) * m_;      // spectrum bin of that index
                        float cj = 1.f;
                        if (idx < 0) { idx = -idx; cj = -1.f; }
                        else if (idx > a.L / 2) { idx = a.L - idx; cj = -1.f; }
                        const float2 u = *reinterpret_cast<const float2*>(xrow + 2 * idx);
                        const float gq = win[q];
                        v = make_float2(u.x * gq, cj * u.y * gq);
                    } else {
                        v = *reinterpret_cast<const float2*>(xrow + 2 * (tt + q4 * m_));      // (not reached: synthesis returns above)
                    }
                }
                gx[q4][e] = v;
            }
        }
        if (b_on) gb = *reinterpret_cast<const float4*>(bp + k0);
    };
    auto store_set = [&](int buf) {
        float* Aw = As0 + buf * 4 * D4_BM * D4_LD + s_row * D4_LD + 4 * s_kq;
        float4 y[4];
        float2 gt[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {      // t1 even, mpad % 8 == 0: one 16-byte LDS read per residue (zeros of the padded table past m)
            const float4 w = g_t1 + 1 < mpad ? *reinterpret_cast<const float4*>(&twl[r * mpad + g_t1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            gt[r][0] = make_float2(w.x, w.y); gt[r][1] = make_float2(w.z, w.w);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float2 x0 = gx[0][e], x1 = gx[1][e], x2 = gx[2][e], x3 = gx[3][e];
            const float2 s0 = make_float2(x0.x + x2.x, x0.y + x2.y), s1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            const float2 d0 = make_float2(x0.x - x2.x, x0.y - x2.y), d1 = make_float2(x1.x - x3.x, x1.y - x3.y);
            const float2 y0 = make_float2(s0.x + s1.x, s0.y + s1.y);
            const float2 y2 = make_float2(s0.x - s1.x, s0.y - s1.y);
            // INV (forward DFT sign): y1 = d0 - i d1, y3 = d0 + i d1;  FWD (inverse sign): swapped
            const float2 ym = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - i d1
            const float2 yp = make_float2(d0.x - d1.y, d0.y + d1.x);      // d0 + i d1
            const float2 y1 = FWD ? yp : ym, y3 = FWD ? ym : yp;
            const float2 z1 = make_float2(y1.x * gt[0][e].x - y1.y * gt[0][e].y, y1.x * gt[0][e].y + y1.y * gt[0][e].x);
            const float2 z2 = make_float2(y2.x * gt[1][e].x - y2.y * gt[1][e].y, y2.x * gt[1][e].y + y2.y * gt[1][e].x);
            const float2 z3 = make_float2(y3.x * gt[2][e].x - y3.y * gt[2][e].y, y3.x * gt[2][e].y + y3.y * gt[2][e].x);
            if (e == 0) { y[0].x = y0.x; y[0].y = y0.y; y[1].x = z1.x; y[1].y = z1.y; y[2].x = z2.x; y[2].y = z2.y; y[3].x = z3.x; y[3].y = z3.y; }
            else        { y[0].z = y0.x; y[0].w = y0.y; y[1].z = z1.x; y[1].w = z1.y; y[2].z = z2.x; y[2].w = z2.y; y[3].z = z3.x; y[3].w = z3.y; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(Aw + r * D4_BM * D4_LD) = y[r];
        *reinterpret_cast<float4*>(Bs0 + buf * 64 * D4_LD + s_row * D4_LD + 4 * s_kq) = gb;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31, lk = lane >> 5;
    load_set(0);
    __syncthreads();             // twiddle table complete
    store_set(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const bool more = k0 + 16 < K;
        if (more) load_set(k0 + 16);
        const float* As = As0 + (cur * 4 + wave) * D4_BM * D4_LD;     // this wave's residue
        const float* Bs = Bs0 + cur * 64 * D4_LD;
        float av[2][8], bv[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 lo = *reinterpret_cast<const float4*>(&As[(i * 32 + lrow) * D4_LD + 8 * lk]);
            const float4 hi = *reinterpret_cast<const float4*>(&As[(i * 32 + lrow) * D4_LD + 8 * lk + 4]);
            av[i][0] = lo.x; av[i][1] = lo.y; av[i][2] = lo.z; av[i][3] = lo.w;
            av[i][4] = hi.x; av[i][5] = hi.y; av[i][6] = hi.z; av[i][7] = hi.w;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !wide) break;
            const float4 lo = *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * D4_LD + 8 * lk]);
            const float4 hi = *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * D4_LD + 8 * lk + 4]);
            bv[j][0] = lo.x; bv[j][1] = lo.y; bv[j][2] = lo.z; bv[j][3] = lo.w;
            bv[j][4] = hi.x; bv[j][5] = hi.y; bv[j][6] = hi.z; bv[j][7] = hi.w;
        }
        if (wide) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][kk], bv[j][kk], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][kk], bv[0][kk], acc[i][0], 0, 0, 0);
        }
        if (more) store_set(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: wave = residue r, lane column n -> (k = n/2, re/im), output index q = 4k + r.
    // The four waves hold interleaved parts of every output row, so the tile is transposed through
    // LDS (32 rows x 256 floats at a time, row stride 260) and written as 16-byte stores of two
    // consecutive complex outputs per lane -- full 128-byte lines instead of 4-byte scatters.
    constexpr int TLD = 260;
    float* const Tt = lds;                       // reuses the staging buffers (33,280 B needed)
    const int kq0 = t.n0 >> 1;                   // first k of the tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !wide) break;
            const int col = 8 * ((j * 32 + lrow) >> 1) + 2 * wave + (lrow & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) Tt[(acc_row(r) + 4 * lk) * TLD + col] = acc[i][j][r];
        }
        __syncthreads();
        const int c4 = tid & 63;                 // float4 slot of the row: complex outputs 2*c4, 2*c4+1
        const int q = 4 * kq0 + 2 * c4;
        if (c4 < (wide ? 64 : 32) && q < Lg) {
            float w0 = 1.f, w1 = 1.f;
            int pos = q;
            if (!FWD) {
                w0 = win[q]; w1 = win[q + 1];
                pos = q + 2 * m_;                // spectrum position p = (q + Lg/2) mod Lg
                if (pos >= Lg) pos -= Lg;
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rl = (tid >> 6) + 4 * it;
                const int mrow = t.m0 + i * 32 + rl;
                if (mrow >= M) break;
                float4 v = *reinterpret_cast<const float4*>(&Tt[rl * TLD + 4 * c4]);
                v.x *= w0; v.y *= w0; v.z *= w1; v.w *= w1;
                float* d;
                if (!FWD && a.row_len) {
                    d = a.dst + 2 * ((int64_t)mrow * a.row_len + bd.ent + pos);
                } else {
                    const int rb = mrow / a.S, rs = mrow - rb * a.S;
                    d = a.dst + 2 * (BCS * bd.cum + (((int64_t)rb * bd.F + bd.f) * a.S + rs) * Lg + pos);
                }
                *reinterpret_cast<float4*>(d) = v;
            }
        }
    }
}

}  // namespace xsq
