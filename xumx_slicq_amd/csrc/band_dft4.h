// Per-band DFT with one radix-4 decimation-in-frequency stage fused into the operand staging.
//
// Every band length is a multiple of 4 (Lg = 4m).  For X[q] = sum_t x[t] w^(q t), w = exp(-+2 pi i / Lg):
//   X[4k + r] = sum_{t1 < m}  y_r[t1] * exp(-+2 pi i k t1 / m),
//   y_r[t1]   = w^(r t1) * sum_{a < 4} x[t1 + a m] * (-+i)^(r a)          (r = 0..3)
// i.e. four length-m DFTs that share ONE m x m matrix: 4x fewer MFMA flops than the dense Lg x Lg
// product of the generic engine (gemm_tile.h), which keeps the bands with Lg < 48 (3 % of the flops).
// The four quarters x[t1 + a m] are contiguous in memory, so the butterflies cost 8 independent 8-byte
// loads per lane and K-step and happen in registers on the way into LDS.
//
// One workgroup = 4 wavefronts on 32 rows x ALL 2m real columns of the band (up to 10 blocks of 16): every coefficient
// is staged exactly once -- loads, butterflies, twiddles, mask products.  (Round 2 started with 64 x 64 tiles: each
// 64-column tile of a band staged the same operand rows again, 1.85 times per coefficient over the plan, and the narrow
// tiles of the short bands carried the vector work of a wide one for a quarter of the MFMAs: 5.3 vector instructions
// per MFMA over the launch, waves waiting 60 % of their cycles.)
//   wave w:  rows 16 (w & 1) .. + 15,  residues 2 (w >> 1) and 2 (w >> 1) + 1,  every column block:  acc[2][10] f32x4
//   K-step:  thread (row = tid >> 3, t1 = tid & 7) stages ONE complex t1 of one row (four quarters -> four residues),
//            all threads together stage the 16 x 2m slab of the DFT matrix; 2 + ncb 16-byte LDS reads and
//            8 ncb MFMAs per wave (v_mfma_f32_16x16x4_f32: lane = row l & 15 / column l & 15, k quad l >> 4; MFMA j of
//            a 16-k chunk takes k = 4 (l >> 4) + j, so a fragment is one 16-byte LDS read).
//   LDS:     2 x (4 x 32 + 160) rows x 20 floats + twiddles + window = 49,280 B -> 3 workgroups per CU.
// Epilogue: a lane pair (Re, Im column of one k) holds residues 2 rp, 2 rp + 1 of four rows; the even lane takes
// rows 0, 1 and the odd lane rows 2, 3 of the quad, two DPP exchanges each, and stores (q, q + 1) = 16 bytes; the
// other half of each 32-byte group comes from the wave with the other residue pair.  No LDS transpose, no barrier.
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
#include "gemm_tile_bf3.h"

namespace xsq {

struct Band4Dev {
    int Lg, m, bin0, f, F, ent;
    int ldd;            // row length of the transposed DFT_m matrix: round_up(2m, 16)
    int jband;          // band index inside the plan (= row of the model's input_mean / input_scale tables)
    int64_t cum;        // arena offset of the band's block (complex per channel-slice)
    int64_t d_off;      // float offset of Dt[n = (k, re/im)][kk = (t1, re/im)] inside the direction's pool
    int64_t tw_off;     // float offset of the twiddles w^(r t1), r = 1..3: [3][round_up(m, 8)] complex
    int64_t win_off;    // float offset of g' (FWD) / wd (INV), Lg floats in window order q
};

// One tile of the radix-4 kernel WITH its band's descriptor (72 bytes, read with scalar loads in one go).  Through a
// tile -> band index -> band table chain the prologue made two dependent round trips before the first operand address
// was known.
struct Tile4Dev {
    int m0, ncb;        // first row; 16-column blocks of the band (1..10)
    Band4Dev bd;
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
    // FWD, optional: the CDAE's whitened magnitude (|coef| + mean[band]) * scale[band] (model.py:238-242) written beside
    // the coefficients, same arena layout, real -- the separate magnitude pass then never runs.  split = 1: stored in
    // the split-bf16 operand format (gemm_tile_bf3.h).
    float* xin;
    const float* mean;
    const float* scale;
    int split;
};

#ifndef XSQ_D4_STAMP
#define XSQ_D4_STAMP 0      // diagnostic build: phase stamps of the synthesis launch (tools/band_phases.py)
#endif
#if XSQ_D4_STAMP
// per tile: s_memrealtime (100 MHz) at 0 start, 1 first operands arrived, 2 K loop done, 3 epilogue stores issued; [4] = ncb, [5] = K-steps
constexpr int D4_STAMP_TILES = 1 << 17;
__device__ unsigned long long g_d4_stamps[D4_STAMP_TILES * 8];
#define XSQ_D4S(i) do { if (!FWD && tid == 0 && blockIdx.x < D4_STAMP_TILES) g_d4_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSQ_D4S(i) do { } while (0)
#endif

constexpr int D4_LD = 20, D4_MPAD = 80;     // bands up to Lg = 320 (the plan builder routes longer ones to the dense engine)
constexpr int D4H_ROWS = 32;
// NCBMAX = most 16-column blocks a band of this instantiation has.  10 (Lg <= 320): 49 KB of LDS and 80 accumulator
// registers, three workgroups per CU -- the product configuration.  5 (Lg <= 160): 35 KB and 40 accumulator registers
// -> FOUR workgroups per CU, an A/B arm (XSQ_D4_SPLIT=1) that measured no faster.  The kernel waits: in-kernel stamps
// (tools/band_phases.py) put a K-step at 1.8-4.2 us of wall time for 0.24-1.2 us of MFMAs per wave (the next operands
// arrive after 2-5 us under load) and the prologue at 3.4-8.4 us; a fourth resident workgroup on the narrow bands -- 41 %
// of the summed tile time -- did not shorten them: the round trips lengthen with the requests in flight.
template <int NCBMAX> struct D4Cfg { static constexpr int waves = NCBMAX > 5 ? 3 : 4, mpad = NCBMAX > 5 ? D4_MPAD : 40; };
template <bool FWD, int NCBMAX = 10>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(D4Cfg<NCBMAX>::waves, D4Cfg<NCBMAX>::waves)))
void band_dft4_full_kernel(Band4Args a, const Tile4Dev* __restrict__ tiles, int ntiles) {
    constexpr int D4H_NCB = NCBMAX, MPADL = D4Cfg<NCBMAX>::mpad;
    __shared__ __attribute__((aligned(16))) float lds[2 * (4 * D4H_ROWS + 16 * D4H_NCB) * D4_LD];
    __shared__ __attribute__((aligned(16))) float2 twl[3 * MPADL];        // w^(r t1), r = 1..3, of this tile's band
    __shared__ __attribute__((aligned(16))) float winl[4 * MPADL];        // INV: dual window wd[q] of this tile's band
    constexpr int ABUF = 4 * D4H_ROWS * D4_LD, BBUF = 16 * D4H_NCB * D4_LD;
    float* const As0 = lds;                     // [buf][r][row][20]
    float* const Bs0 = lds + 2 * ABUF;          // [buf][col][20]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    XSQ_D4S(0);
    // By value: through a reference into global memory the compiler had to RE-LOAD the fields after every store of the
    // epilogue (they might alias), and each reload's s_waitcnt vmcnt(0) also waited for the store before it -- one HBM
    // round trip per 16-byte store (tools/scan_isa.py counts such loads).
    const Tile4Dev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const int ncb = t.ncb;                      // 16-column blocks of the band, 1..10 (uniform)
    const Band4Dev bd = t.bd;
    const int m_ = bd.m, Lg = bd.Lg, K = 2 * m_, M = a.BC * a.S;
    const int64_t BCS = (int64_t)a.BC * a.S;

    // ---- staging assignment: row s_row, complex t1 = K-step base + s_t ---------------------------
    const int s_row = tid >> 3, s_t = tid & 7;
    const int row = t.m0 + s_row;
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : M - 1;
    const int bc = rowc / a.S, sl = rowc - bc * a.S;
    const float* const xbase = a.src;
    const float* const mbase = a.mask;
    const bool masked = !FWD && a.mask != nullptr;
    int xoff, moff = 0;
    if (FWD) xoff = rowc * 2 * a.nbins;
    else if (!masked) xoff = (int)(2 * (BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + sl) * Lg));
    else {
        moff = (int)(BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + sl) * Lg);
        xoff = (int)(2 * ((int64_t)a.BCx * a.S * bd.cum + (((int64_t)(bc % a.BCx) * bd.F + bd.f) * a.S + sl) * Lg));
    }
    const float* win = a.pool + bd.win_off;
    const int mpad = (m_ + 7) & ~7;
    const float2* tw = reinterpret_cast<const float2*>(a.pool + bd.tw_off);
    // DFT matrix slab of a K-step: 16 ncb rows (n) x 16 floats = 64 ncb float4, item i = tid + 256 u: n = i >> 2, k quad i & 3
    const float* bp = a.pool + bd.d_off + (int64_t)(tid >> 2) * bd.ldd + 4 * (tid & 3);
    const int nb4 = 64 * ncb;
    constexpr int NBU = (64 * D4H_NCB + 255) / 256;     // float4 items of the matrix slab per thread (3 / 2)
    int bover[NBU];                              // rows by which item u of this thread lies past the slab (0 inside)
#pragma unroll
    for (int u = 0; u < NBU; ++u) { const int n = (tid >> 2) + 64 * u; bover[u] = n < 16 * ncb ? 0 : n - (16 * ncb - 1); }

    float2 raw[4];         // INV: quarter a, complex tc;  FWD: spectrum value of quarter a
    float aux[4];          // INV: mask of quarter a;  FWD: window value
    float4 gb[NBU];
#pragma unroll
    for (int u = 0; u < NBU; ++u) gb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int g_t1 = 0;

    auto fwd_idx = [&](int tt, int q4, float& cj) {        // spectrum bin of window index tt + q4*m, Hermitian reflection
        int idx = bd.bin0 + tt + ((q4 + 2) & 3) * m_;
        cj = 1.f;
        if (idx < 0) { idx = -idx; cj = -1.f; }
        else if (idx > a.L / 2) { idx = a.L - idx; cj = -1.f; }
        return idx;
    };
    // load_set only issues loads (clamped addresses, nothing consumed); products, selects and butterflies in store_set
    auto load_set = [&](int k0) {
        const int t1 = (k0 >> 1) + s_t;
        g_t1 = t1;
        const int tc = t1 < m_ ? t1 : m_ - 1;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (!FWD) {
                raw[q4] = *reinterpret_cast<const float2*>(xbase + (xoff + 2 * (tc + q4 * m_)));
                if (masked) aux[q4] = mbase[moff + tc + q4 * m_];
            } else {
                float cj;
                aux[q4] = win[tc + q4 * m_];
                raw[q4] = *reinterpret_cast<const float2*>(xbase + (xoff + 2 * fwd_idx(tc, q4, cj)));
            }
        }
#pragma unroll
        for (int u = 0; u < NBU; ++u)       // uniform test; rows past the band's blocks re-read the last row (not stored)
            if (256 * u < nb4) gb[u] = *reinterpret_cast<const float4*>(bp + (int64_t)(64 * u - bover[u]) * bd.ldd + k0);
    };
    auto store_set = [&](int buf) {
        const int t1 = g_t1;
        const bool ok = row_ok && t1 < m_;
        float2 x[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float2 v = raw[q4];
            if (!FWD) {
                if (masked) { v.x *= aux[q4]; v.y *= aux[q4]; }
            } else {
                float cj;
                (void)fwd_idx(t1 < m_ ? t1 : m_ - 1, q4, cj);
                v = make_float2(v.x * aux[q4], cj * v.y * aux[q4]);
            }
            x[q4] = ok ? v : make_float2(0.f, 0.f);
        }
        const int tl = t1 < mpad ? t1 : 0;       // table is zero past m; t1 >= mpad only on rows that are zero anyway
        const float2 w1 = twl[tl], w2 = twl[mpad + tl], w3 = twl[2 * mpad + tl];
        const float2 s0 = make_float2(x[0].x + x[2].x, x[0].y + x[2].y), s1 = make_float2(x[1].x + x[3].x, x[1].y + x[3].y);
        const float2 d0 = make_float2(x[0].x - x[2].x, x[0].y - x[2].y), d1 = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
        const float2 y0 = make_float2(s0.x + s1.x, s0.y + s1.y);
        const float2 y2 = make_float2(s0.x - s1.x, s0.y - s1.y);
        const float2 ym = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - i d1
        const float2 yp = make_float2(d0.x - d1.y, d0.y + d1.x);      // d0 + i d1
        const float2 y1 = FWD ? yp : ym, y3 = FWD ? ym : yp;          // INV (forward DFT sign): y1 = d0 - i d1
        const float2 z1 = make_float2(y1.x * w1.x - y1.y * w1.y, y1.x * w1.y + y1.y * w1.x);
        const float2 z2 = make_float2(y2.x * w2.x - y2.y * w2.y, y2.x * w2.y + y2.y * w2.x);
        const float2 z3 = make_float2(y3.x * w3.x - y3.y * w3.y, y3.x * w3.y + y3.y * w3.x);
        float* Aw = As0 + buf * ABUF + s_row * D4_LD + 2 * s_t;
        *reinterpret_cast<float2*>(Aw) = y0;
        *reinterpret_cast<float2*>(Aw + D4H_ROWS * D4_LD) = z1;
        *reinterpret_cast<float2*>(Aw + 2 * D4H_ROWS * D4_LD) = z2;
        *reinterpret_cast<float2*>(Aw + 3 * D4H_ROWS * D4_LD) = z3;
        float* Bw = Bs0 + buf * BBUF + (tid >> 2) * D4_LD + 4 * (tid & 3);
#pragma unroll
        for (int u = 0; u < NBU; ++u)
            if (tid + 256 * u < nb4) *reinterpret_cast<float4*>(Bw + 64 * u * D4_LD) = gb[u];
    };

    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc[2][D4H_NCB];                     // [residue of the pair][16-column block]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int cb = 0; cb < D4H_NCB; ++cb) acc[e][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, kq = lane >> 4;
    const int rh = wave & 1, rp = wave >> 1;
    load_set(0);
    float w_mu = 0.f, w_sc = 1.f;                // whitening constants of this tile's band (uniform)
    if (FWD && a.xin) { w_mu = a.mean[bd.jband]; w_sc = a.scale[bd.jband]; }
    {   // tables of the band into LDS: every value of this thread requested before the first is stored (a load -> store
        // loop made Lg > 256 two dependent round trips)
        const float2 tv = tid < 3 * mpad ? tw[tid] : make_float2(0.f, 0.f);
        float wv0 = 0.f, wv1 = 0.f;
        if (!FWD) { wv0 = tid < Lg ? win[tid] : 0.f; wv1 = tid + 256 < Lg ? win[tid + 256] : 0.f; }
        if (tid < 3 * mpad) twl[tid] = tv;
        if (!FWD) { if (tid < Lg) winl[tid] = wv0; if (tid + 256 < Lg) winl[tid + 256] = wv1; }
    }
    __syncthreads();             // tables complete (store_set reads the twiddles)
    store_set(0);
    XSQ_D4S(1);
    __syncthreads();
    int cur = 0;
    auto k_step = [&]() {
        const float* As = As0 + cur * ABUF + (2 * rp * D4H_ROWS + 16 * rh + l16) * D4_LD + 4 * kq;
        const float* Bs = Bs0 + cur * BBUF + l16 * D4_LD + 4 * kq;
        const float4 a0 = *reinterpret_cast<const float4*>(As);
        const float4 a1 = *reinterpret_cast<const float4*>(As + D4H_ROWS * D4_LD);
#pragma unroll
        for (int cb = 0; cb < D4H_NCB; ++cb) {
            if (cb >= ncb) continue;              // (not break: the compiler refuses to unroll the multi-exit loop and moves acc to scratch)
            const float4 b = *reinterpret_cast<const float4*>(Bs + 16 * cb * D4_LD);
            acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[0][cb], 0, 0, 0);
            acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc[1][cb], 0, 0, 0);
            acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[0][cb], 0, 0, 0);
            acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc[1][cb], 0, 0, 0);
            acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc[0][cb], 0, 0, 0);
            acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc[1][cb], 0, 0, 0);
            acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc[0][cb], 0, 0, 0);
            acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc[1][cb], 0, 0, 0);
        }
    };
    int k0 = 0;
    for (; k0 + 16 < K; k0 += 16) {
        load_set(k0 + 16);
        k_step();
        store_set(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    k_step();
    XSQ_D4S(2);
#if XSQ_D4_STAMP
    if (!FWD && tid == 0 && blockIdx.x < D4_STAMP_TILES) { g_d4_stamps[blockIdx.x * 8 + 4] = ncb; g_d4_stamps[blockIdx.x * 8 + 5] = (K + 15) / 16; }
#endif

    // ---- epilogue.  Register rr of acc[e][cb]: row 16 rh + 4 kq + rr, column l16 = (k' = l16 >> 1, Re / Im) of residue
    // 2 rp + e, i.e. output q = 4 (8 cb + k') + 2 rp + e.  The even lane (Re) finishes rows rr = 0, 1, the odd lane (Im)
    // rows rr = 2, 3: each sends the partner what it holds of the partner's rows and receives the missing half.
    auto swap1 = [](float v) {                   // value of lane l ^ 1 (DPP quad_perm [1, 0, 3, 2])
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    };
    const int part = lane & 1;
    float* rowp[2];
    bool rok[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
        const int mrow = t.m0 + 16 * rh + 4 * kq + 2 * part + sx;
        rok[sx] = mrow < M;
        const int mr = rok[sx] ? mrow : 0;
        if (!FWD && a.row_len) {
            rowp[sx] = a.dst + 2 * ((int64_t)mr * a.row_len + bd.ent);
        } else {
            const int rb = mr / a.S, rs = mr - rb * a.S;
            rowp[sx] = a.dst + 2 * (BCS * bd.cum + (((int64_t)rb * bd.F + bd.f) * a.S + rs) * Lg);
        }
    }
    float* const xin = FWD ? a.xin : nullptr;
    const bool split = a.split;
#pragma unroll
    for (int cb = 0; cb < D4H_NCB; ++cb) {
        if (cb >= ncb) continue;
        const int k = 8 * cb + (l16 >> 1);
        const int q = 4 * k + 2 * rp;
        const bool on = k < m_;
        int pos = q;
        float w0 = 1.f, w1 = 1.f;
        if (!FWD) {
            const float2 w = *reinterpret_cast<const float2*>(&winl[on ? q : 0]);
            w0 = w.x; w1 = w.y;
            pos = q + 2 * m_;                    // spectrum position p = (q + Lg/2) mod Lg
            if (pos >= Lg) pos -= Lg;
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            // even lane: keeps Re of row sx, sends Re of row 2 + sx;  odd lane: keeps Im of row 2 + sx, sends Im of row sx
            const float g0 = swap1(part ? acc[0][cb][sx] : acc[0][cb][2 + sx]);
            const float g1 = swap1(part ? acc[1][cb][sx] : acc[1][cb][2 + sx]);
            float4 v = part ? make_float4(g0, acc[0][cb][2 + sx], g1, acc[1][cb][2 + sx])
                            : make_float4(acc[0][cb][sx], g0, acc[1][cb][sx], g1);
            v.x *= w0; v.y *= w0; v.z *= w1; v.w *= w1;
            if (!on || !rok[sx]) continue;
            float* const d = rowp[sx] + 2 * pos;
            *reinterpret_cast<float4*>(d) = v;
            if (FWD && xin) {
                float2 o = make_float2(whiten_mag(v.x, v.y, w_mu, w_sc), whiten_mag(v.z, v.w, w_mu, w_sc));
                if (split) bf3_words2(o.x, o.y, o.x, o.y);
                *reinterpret_cast<float2*>(xin + ((d - a.dst) >> 1)) = o;
            }
        }
    }
    XSQ_D4S(3);
}

}  // namespace xsq
