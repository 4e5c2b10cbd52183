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
// One workgroup = 4 wavefronts on a tile of 64 rows x 64 real columns (32 k's; N tails of 16, 32 or 48 columns): wave w
// owns rows 16 w .. 16 w + 15 for all four residues (see the MFMA section), all waves read the same B fragments
// (DFT_m), up to 64 MFMAs per wave and K-step between barriers.
// LDS: 2 x (4 x 64 + 64) rows x 20 floats = 51,200 B -> 3 workgroups per CU.
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

constexpr int D4_BM = 64, D4_LD = 20, D4_MPAD = 80;     // bands up to Lg = 320 (the plan builder routes longer ones to the dense engine)

#ifndef XSQ_D4_ABL
#define XSQ_D4_ABL 0      // diagnostic builds: 1 no MFMAs, 2 no operand loads, 4 no matrix loads, 8 no epilogue, 16 no epilogue stores
#endif
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
    const int ncb = t.narrow;                  // 16-column blocks of this tile, 1..4 (uniform)
    const Band4Dev bd = a.bands[t.group];      // by value: see the epilogue
    const int m_ = bd.m, Lg = bd.Lg, K = 2 * m_, M = a.BC * a.S;
    const int64_t BCS = (int64_t)a.BC * a.S;

    // ---- staging assignment: row s_row, complex pair 2*s_kq, 2*s_kq+1 of the K-step -------------
    const int s_row = tid >> 2, s_kq = tid & 3;
    const int row = t.m0 + s_row;
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : M - 1;             // rows past M load row M-1 (any valid memory) and are zeroed below
    const int bc = rowc / a.S, s = rowc - bc * a.S;
    // 32-bit element offsets from the (uniform) arena pointers: the arenas hold < 2^31 floats (checked by the
    // host) and two lane-varying 64-bit pointers would not fit under the 3-workgroups-per-CU register cap
    const float* const xbase = a.src;
    const float* const mbase = a.mask;
    const bool masked = !FWD && a.mask != nullptr;
    int xoff, moff = 0;
    if (FWD) xoff = rowc * 2 * a.nbins;
    else if (!masked) xoff = (int)(2 * (BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + s) * Lg));
    else {
        moff = (int)(BCS * bd.cum + (((int64_t)bc * bd.F + bd.f) * a.S + s) * Lg);
        xoff = (int)(2 * ((int64_t)a.BCx * a.S * bd.cum + (((int64_t)(bc % a.BCx) * bd.F + bd.f) * a.S + s) * Lg));
    }
    const float* win = a.pool + bd.win_off;
    const int mpad = (m_ + 7) & ~7;
    const float2* tw = reinterpret_cast<const float2*>(a.pool + bd.tw_off);
    const float* bp = a.pool + bd.d_off + (int64_t)(t.n0 + s_row) * bd.ldd + 4 * s_kq;
    const bool b_on = s_row < 16 * ncb;

    // Operand staging.  load_set only ISSUES loads -- unconditionally, from clamped addresses, nothing consumed --
    // so that the K-step's ~9 loads per lane are in flight together while the previous step's MFMAs run; every
    // product, select and butterfly happens in store_set.  (With the mask multiply / window multiply / bounds
    // predicates inside load_set the compiler had to wait for each quarter's loads in turn: four global round
    // trips per K-step, none of them overlapped with the MFMAs -- 45 us per tile, measured.)
    struct __attribute__((aligned(8))) F4 { float x, y, z, w; };     // two consecutive complex values, 8-byte aligned
    struct __attribute__((aligned(4))) F2 { float x, y; };           // two consecutive masks / window values
    F4 raw[4];             // INV: quarters a = 0..3, complex (tc, tc + 1);  FWD: element e = 0: quarters 0,1 / ...
    float2 rawf[4][2];     // FWD: spectrum values of quarter a, element e
    F2 aux[4];             // INV: masks (tc, tc + 1) of quarter a;  FWD: window values
    float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
    int g_t1 = 0;          // t1 of the staged pair (twiddles are read from LDS in store_set)

    auto fwd_idx = [&](int tt, int q4, float& cj) {        // spectrum bin of window index tt + q4*m, Hermitian reflection
        int idx = bd.bin0 + tt + ((q4 + 2) & 3) * m_;
        cj = 1.f;
        if (idx < 0) { idx = -idx; cj = -1.f; }
        else if (idx > a.L / 2) { idx = a.L - idx; cj = -1.f; }
        return idx;
    };
    auto load_set = [&](int k0) {      // k0 = first real k of the K-step (16 per step = 8 complex t1)
        const int t1 = (k0 >> 1) + 2 * s_kq;
        g_t1 = t1;
        const int tc = t1 < m_ - 1 ? t1 : m_ - 2;          // the pair (tc, tc + 1) always lies inside the quarter
        if ((XSQ_D4_ABL & 2) && k0 > 0) { if (b_on && !(XSQ_D4_ABL & 4)) gb = *reinterpret_cast<const float4*>(bp + k0); return; }
        if (!FWD) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                raw[q4] = *reinterpret_cast<const F4*>(xbase + (xoff + 2 * (tc + q4 * m_)));
                if (masked) aux[q4] = *reinterpret_cast<const F2*>(mbase + (moff + tc + q4 * m_));
            }
        } else {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                aux[q4] = *reinterpret_cast<const F2*>(win + tc + q4 * m_);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float cj;
                    rawf[q4][e] = *reinterpret_cast<const float2*>(xbase + (xoff + 2 * fwd_idx(tc + e, q4, cj)));
                }
            }
        }
        if (b_on && (!(XSQ_D4_ABL & 4) || k0 == 0)) gb = *reinterpret_cast<const float4*>(bp + k0);
    };
    auto store_set = [&](int buf) {
        float* Aw = As0 + buf * 4 * D4_BM * D4_LD + s_row * D4_LD + 4 * s_kq;
        // unpack the staged pair: element e is t1 + e; the loads were taken at (tc, tc + 1)
        const int t1 = g_t1;
        const int tc = t1 < m_ - 1 ? t1 : m_ - 2;
        const bool ok0 = row_ok && t1 < m_, ok1 = row_ok && t1 + 1 < m_;
        const bool shifted = t1 != tc;                     // t1 = m - 1 (odd m): element 0 is the SECOND value of the loaded pair
        float2 gx[4][2];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float2 v0, v1;
            if (!FWD) {
                v0 = shifted ? make_float2(raw[q4].z, raw[q4].w) : make_float2(raw[q4].x, raw[q4].y);
                v1 = make_float2(raw[q4].z, raw[q4].w);
                if (masked) {
                    const float mk0 = shifted ? aux[q4].y : aux[q4].x, mk1 = aux[q4].y;
                    v0.x *= mk0; v0.y *= mk0; v1.x *= mk1; v1.y *= mk1;
                }
            } else {
                float cj0, cj1;
                (void)fwd_idx(tc, q4, cj0);
                (void)fwd_idx(tc + 1, q4, cj1);
                const float g0 = shifted ? aux[q4].y : aux[q4].x, g1 = aux[q4].y;
                const float2 u0 = shifted ? rawf[q4][1] : rawf[q4][0], u1 = rawf[q4][1];
                const float c0 = shifted ? cj1 : cj0;
                v0 = make_float2(u0.x * g0, c0 * u0.y * g0);
                v1 = make_float2(u1.x * g1, cj1 * u1.y * g1);
            }
            gx[q4][0] = ok0 ? v0 : make_float2(0.f, 0.f);
            gx[q4][1] = ok1 ? v1 : make_float2(0.f, 0.f);
        }
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

    // ---- MFMA section: wave w owns rows 16 w .. 16 w + 15 of the tile for ALL FOUR residues (v_mfma_f32_16x16x4_f32,
    // lane = row l & 15 / column l & 15, k quad l >> 4; MFMA j of a 16-k chunk takes k = 4 (l >> 4) + j, so a fragment is
    // one 16-byte LDS read).  A lane pair (columns 2k', 2k'+1 = Re, Im) then holds outputs q = 4k .. 4k + 3 of its rows
    // across the four residue accumulators: one DPP exchange per pair gives each lane two consecutive complex outputs
    // -- a 16-byte store -- with no LDS transpose and no barrier in the epilogue.  (The first version gave every wave
    // one residue of 64 rows, v_mfma_f32_32x32x2_f32: the waves then held interleaved parts of every output row and
    // the tile went through LDS twice behind four barriers.)
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc[4][4];                           // [residue][16-column block]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[r][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, kq = lane >> 4;
    // epilogue constants: this lane's outputs are q = 4 (k0 + 8 cb + k') + 2 (l & 1) + {0, 1}, k' = (l & 15) >> 1
    const int e_k = (t.n0 >> 1) + (l16 >> 1);    // + 8 cb
    const int e_odd = lane & 1;
    float e_w[4][2];
    load_set(0);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int q = 4 * (e_k + 8 * cb) + 2 * e_odd;
        const int qc = q + 1 < Lg ? q : 0;        // clamped: unconditional loads
        e_w[cb][0] = FWD ? 1.f : win[qc];
        e_w[cb][1] = FWD ? 1.f : win[qc + 1];
    }
    float w_mu = 0.f, w_sc = 1.f;                // whitening constants of this tile's band (uniform)
    if (FWD && a.xin) { w_mu = a.mean[bd.jband]; w_sc = a.scale[bd.jband]; }
    for (int i = tid; i < 3 * mpad; i += 256) twl[i] = tw[i];      // read back in store_set, after the barrier below
    __syncthreads();             // twiddle table complete
    store_set(0);
    __syncthreads();
    int cur = 0;
    // One K-step of one wave: 4 A fragments (its 16 rows, four residues) and 4 B fragments from LDS, then 16 MFMAs per
    // 16-column block of the tile, the block loop OUTERMOST: one uniform branch per block.  (With the width test in
    // front of every MFMA pair -- the innermost position -- the K-step carried 40 scalar branches between its 64 MFMAs;
    // a switch over four fully specialised K-steps spilled 60 registers.)  MFMA j takes k = 4 (l >> 4) + j of the
    // chunk, i.e. every MFMA spans the whole K-step: a ragged last K-step (K = 2m not a multiple of 16) cannot skip any.
    auto k_step = [&]() {
        const float* As = As0 + cur * 4 * D4_BM * D4_LD + (wave * 16 + l16) * D4_LD + 4 * kq;     // + residue * D4_BM * D4_LD
        const float* Bs = Bs0 + cur * 64 * D4_LD + l16 * D4_LD + 4 * kq;                          // + 16 cb * D4_LD
        float4 av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = *reinterpret_cast<const float4*>(As + r * D4_BM * D4_LD);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bv[cb] = *reinterpret_cast<const float4*>(Bs + 16 * cb * D4_LD);   // zero rows past the tile's width
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (cb >= ncb) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bx = j == 0 ? bv[cb].x : j == 1 ? bv[cb].y : j == 2 ? bv[cb].z : bv[cb].w;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ax = j == 0 ? av[r].x : j == 1 ? av[r].y : j == 2 ? av[r].z : av[r].w;
                    if (XSQ_D4_ABL & 1) { acc[r][cb][j] += ax * bx; continue; }
                    acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, acc[r][cb], 0, 0, 0);
                }
            }
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
    k_step();      // the band's last K-step: nothing to stage, no barrier behind it

    // ---- epilogue: register rr of acc[r][cb] is row 16 w + 4 (l >> 4) + rr, column l & 15 = (k', Re / Im) of residue r,
    // i.e. output q = 4 (k0 + 8 cb + k') + r.  Even lanes end up with (q, q + 1) = residues 0, 1, odd lanes with
    // residues 2, 3: the even lane takes Im of residues 0 / 1 from its neighbour, the odd lane Re of residues 2 / 3.
    if (XSQ_D4_ABL & 8) { if (acc[0][0][0] + acc[1][1][3] + acc[2][1][1] + acc[3][0][2] == 1.2345e-30f) __builtin_trap(); return; }
    auto swap1 = [](float v) {                   // value of lane l ^ 1 (DPP quad_perm [1, 0, 3, 2])
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    };
    // Everything the stores need is in registers before the first one is issued: the band descriptor is a by-value
    // copy (through the reference into a.bands the compiler had to RE-LOAD its fields after every store -- they might
    // alias -- and each reload's s_waitcnt vmcnt(0) also waited for the store before it: one HBM round trip per
    // 16-byte store, half of the kernel's time, measured with the XSQ_D4_ABL builds), and the four row bases are
    // computed once.
    float* rowp[4];
    bool rok[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int mrow = t.m0 + wave * 16 + 4 * kq + rr;
        rok[rr] = mrow < M;
        const int mr = rok[rr] ? mrow : 0;
        if (!FWD && a.row_len) {
            rowp[rr] = a.dst + 2 * ((int64_t)mr * a.row_len + bd.ent);
        } else {
            const int rb = mr / a.S, rs = mr - rb * a.S;
            rowp[rr] = a.dst + 2 * (BCS * bd.cum + (((int64_t)rb * bd.F + bd.f) * a.S + rs) * Lg);
        }
    }
    float* const xin = FWD ? a.xin : nullptr;
    const bool split = a.split;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (cb >= ncb) break;
        const int k = e_k + 8 * cb;
        const int q = 4 * k + 2 * e_odd;
        const bool on = k < m_;
        int pos = q;
        if (!FWD) {
            pos = q + 2 * m_;                    // spectrum position p = (q + Lg/2) mod Lg
            if (pos >= Lg) pos -= Lg;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float x = swap1(e_odd ? acc[0][cb][rr] : acc[2][cb][rr]);      // even lane receives Im of residue 0, odd lane Re of residue 2
            const float y = swap1(e_odd ? acc[1][cb][rr] : acc[3][cb][rr]);      //                    Im of residue 1,          Re of residue 3
            float4 v = e_odd ? make_float4(x, acc[2][cb][rr], y, acc[3][cb][rr]) : make_float4(acc[0][cb][rr], x, acc[1][cb][rr], y);
            v.x *= e_w[cb][0]; v.y *= e_w[cb][0]; v.z *= e_w[cb][1]; v.w *= e_w[cb][1];
            if (!on || !rok[rr]) continue;
            float* const d = rowp[rr] + 2 * pos;
            if (!(XSQ_D4_ABL & 16) || v.x == 1.2345e-30f) *reinterpret_cast<float4*>(d) = v;
            if (FWD && xin) {
                float2 o = make_float2(whiten_mag(v.x, v.y, w_mu, w_sc), whiten_mag(v.z, v.w, w_mu, w_sc));
                if (split) bf3_words2(o.x, o.y, o.x, o.y);
                *reinterpret_cast<float2*>(xin + ((d - a.dst) >> 1)) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Full-width variant: one workgroup = 32 rows x ALL 2m real columns of the band (up to 10 blocks of 16).
//
// In band_dft4_kernel every 64-column tile of a band stages the same operand rows again -- loads, radix-4
// butterflies, twiddles, mask products -- 1.85 times per coefficient on average over the plan, and the narrow tiles
// of the short bands carry as many vector-ALU instructions as the wide ones for a quarter of the MFMAs (measured:
// 5.3 vector instructions per MFMA over the launch, waves waiting 60 % of their cycles).  Here each coefficient is
// staged exactly once:
//   wave w:  rows 16 (w & 1) .. + 15,  residues 2 (w >> 1) and 2 (w >> 1) + 1,  every column block:  acc[2][10] f32x4
//   K-step:  thread (row = tid >> 3, t1 = tid & 7) stages ONE complex t1 of one row (four quarters -> four residues),
//            all threads together stage the 16 x 2m slab of the DFT matrix; 2 + ncb 16-byte LDS reads and
//            8 ncb MFMAs per wave.
//   LDS:     2 x (4 x 32 + 160) rows x 20 floats + twiddles + window = 49,280 B -> 3 workgroups per CU.
// Epilogue: a lane pair (Re, Im column of one k) holds residues 2 rp, 2 rp + 1 of four rows; the even lane takes
// rows 0, 1 and the odd lane rows 2, 3 of the quad, two DPP exchanges each, and stores (q, q + 1) = 16 bytes; the
// other half of each 32-byte group comes from the wave with the other residue pair.
constexpr int D4H_ROWS = 32, D4H_NCB = 10;
template <bool FWD>
__global__ __launch_bounds__(256) XSQ_D4_WPE void band_dft4_full_kernel(Band4Args a, const TileDev* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (4 * D4H_ROWS + 16 * D4H_NCB) * D4_LD];
    __shared__ __attribute__((aligned(16))) float2 twl[3 * D4_MPAD];      // w^(r t1), r = 1..3, of this tile's band
    __shared__ __attribute__((aligned(16))) float winl[4 * D4_MPAD];      // INV: dual window wd[q] of this tile's band
    constexpr int ABUF = 4 * D4H_ROWS * D4_LD, BBUF = 16 * D4H_NCB * D4_LD;
    float* const As0 = lds;                     // [buf][r][row][20]
    float* const Bs0 = lds + 2 * ABUF;          // [buf][col][20]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const int ncb = t.narrow;                   // 16-column blocks of the band, 1..10 (uniform)
    const Band4Dev bd = a.bands[t.group];       // by value (see band_dft4_kernel's epilogue)
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
    int bover[3];                                // rows by which item u of this thread lies past the slab (0 inside)
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int n = (tid >> 2) + 64 * u; bover[u] = n < 16 * ncb ? 0 : n - (16 * ncb - 1); }

    float2 raw[4];         // INV: quarter a, complex tc;  FWD: spectrum value of quarter a
    float aux[4];          // INV: mask of quarter a;  FWD: window value
    float4 gb[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
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
        for (int u = 0; u < 3; ++u)         // uniform test; rows past the band's blocks re-read the last row (not stored)
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
        for (int u = 0; u < 3; ++u)
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
    for (int i = tid; i < 3 * mpad; i += 256) twl[i] = tw[i];
    if (!FWD) for (int i = tid; i < Lg; i += 256) winl[i] = win[i];
    __syncthreads();             // tables complete (store_set reads the twiddles)
    store_set(0);
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
}

}  // namespace xsq
