// Per-band DFT, radix-4 stage fused into the staging (band_dft4.h), with the m-point DFTs contracted over PAIRS of inputs.
//
// band_dft4.h runs the four length-m DFTs of a band as complex products on the matrix cores: 4 m^2 real multiply-adds per
// residue and row.  The DFT matrix is Hermitian-symmetric in its input index -- cos(2 pi k (m - n) / m) = cos(2 pi k n / m),
// sin(..) = -sin(..) -- so with
//     a[n] = y[n] + y[m - n],   b[n] = y[n] - y[m - n]        (n = 1 .. (m - 1) / 2;  a[0] = y[0], a[m/2] = y[m/2], b = anything there)
//     P[k] = sum_n cos(2 pi n k / m) a[n],      Q[k] = sum_n sin(2 pi n k / m) b[n]          (complex a, b; REAL matrices)
// the outputs come in mirrored pairs  X[k] = P[k] -+ i Q[k],  X[m - k] = P[k] +- i Q[k]  for k = 0 .. m / 2:
// two real (m/2 + 1) x (m/2 + 1) matrices applied to (Re, Im) of a and b as separate ROWS -- m^2 multiply-adds per residue and
// row, a quarter of the complex product's (the same pairing the slice FFT's small codelets use: slice_fft.h, dft_small).
// Over the Bark-262 bands (tile padding included) the kernel issues 0.38 of the MFMA cycles of band_dft4.h at 1.03 x its other
// vector instructions (tools/isa_budget.py, bench.py issue_bound).
//
// One workgroup = 4 waves on 32 rows x the whole band, as in band_dft4.h:
//   K-step:   8 pairs n.  Thread (row = tid >> 3, p = tid & 7) stages pair n = k0 + p of its row: the four quarters of t1 = n and
//             of t1 = m - n (mask products, radix-4 butterflies, twiddles), then a = y(n) + y(m - n), b = y(n) - y(m - n) per
//             residue.  A pair that is its own mirror (n = 0, n = m / 2) has its second load set switched off (zeros): a = y,
//             and its b meets a zero row of the sine matrix.  LDS rows = (quantity Re a | Im a | Re b | Im b, residue, row),
//             8 floats each (one K-step), so a staging store is one dword and a fragment one ds_read_b64.
//   wave w:   rows 16 (w & 1) .. + 15, residues 2 (w >> 1), + 1, all four quantities, every 16-column block of outputs
//             k = 0 .. m / 2 (at most 3):  acc[4][2][3] f32x4.  v_mfma_f32_16x16x4_f32 j of a K-step takes n = 2 (l >> 4) + j.
//   epilogue: a lane holds P and Q of (row, residue pair, k) itself: X[k] and X[m - k] of both residues are two 16-byte stores
//             each, no cross-lane exchange.
//   LDS:      2 x (512 x 8 + 2 x 48 x 8) floats + twiddles + window = 42.1 KB -> three workgroups per CU.
#pragma once
#include "band_dft4.h"

#ifndef XSQ_S4_EARLY
#define XSQ_S4_EARLY 0      // 1: the operands of K-step i + 2 are requested right behind the staging of K-step i + 1 (in front of the barrier:
                            // they travel during the barrier wait AND the next K-step's MFMAs; same single register set) -- A/B r11y
#endif
#ifndef XSQ_S4_SCHED
#define XSQ_S4_SCHED -1     // >= 0: a scheduling barrier with this mask behind every column block's MFMAs of a K-step (A/B: r11w)
#endif

namespace xsq {

constexpr int S4_KP = 8, S4_NCB = 3, S4_AROWS = 16 * D4H_ROWS, S4_BROWS = 16 * S4_NCB;

template <bool FWD, bool MASKED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void band_dft4s_kernel(Band4Args a, const Tile4Dev* __restrict__ tiles, int ntiles) {
#pragma clang fp contract(off)          // fused multiply-adds only where written (fmaf): same bits from every instantiation
    static_assert(!(FWD && MASKED), "the mask product belongs to the synthesis");
    static_assert(D4H_ROWS == 32, "32-row tiles");
    constexpr int ABUF = S4_AROWS * S4_KP, BBUF = 2 * S4_BROWS * S4_KP;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ABUF + BBUF)];
    __shared__ __attribute__((aligned(16))) float2 twl[3 * D4_MPAD];        // w^(r t1), r = 1..3, of this tile's band
    __shared__ __attribute__((aligned(16))) float winl[4 * D4_MPAD];        // window of this tile's band: g'[q] (FWD) / wd[q] (INV)
    float* const As0 = lds;                      // [buf][quantity][residue][row][8]
    float* const Bs0 = lds + 2 * ABUF;           // [buf][cos | sin][column][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Tile4Dev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const int ncb = t.ncb;                       // 16-column blocks of the outputs k = 0 .. m / 2: 1..3 (uniform)
    const Band4Dev bd = t.bd;
    const int m_ = bd.m, Lg = bd.Lg, K2 = bd.K2, M = a.BC * a.S;
    const int64_t BCS = (int64_t)a.BC * a.S;
    const int mpad = (m_ + 7) & ~7;

    // ---- staging assignment: row s_row, pair n = K-step base + s_p ---------------------------------------------
    const int s_row = tid >> 3, s_p = tid & 7;
    const int row = t.m0 + s_row;
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : M - 1;
    const int bc = rowc / a.S, sl = rowc - bc * a.S;
    const bool reflect = FWD && (bd.bin0 < 0 || bd.bin0 + Lg - 1 > a.L / 2);
    __amdgpu_buffer_rsrc_t rx, rm = buf_rsrc(a.src, 0);
    unsigned vx, vm = BUF_OOB;                   // byte offsets of (row, t1 = 0)
    const unsigned blk = (unsigned)(bd.F * Lg) * (unsigned)a.S;
    if (FWD) {
        rx = buf_rsrc(a.src + 2 * ((int64_t)t.m0 * a.nbins), 8u * D4H_ROWS * a.nbins);
        vx = row_ok ? 8u * (unsigned)(s_row * a.nbins + bd.bin0) : BUF_OOB;
    } else if (!MASKED) {
        rx = buf_rsrc(a.src + 2 * (BCS * bd.cum), 8u * a.BC * blk);
        vx = row_ok ? 8u * (unsigned)(((bc * bd.F + bd.f) * a.S + sl) * Lg) : BUF_OOB;
    } else {
        rx = buf_rsrc(a.src + 2 * ((int64_t)a.BCx * a.S * bd.cum), 8u * a.BCx * blk);
        rm = buf_rsrc(a.mask + BCS * bd.cum, 4u * a.BC * blk);
        vx = row_ok ? 8u * (unsigned)((((bc % a.BCx) * bd.F + bd.f) * a.S + sl) * Lg) : BUF_OOB;
        vm = row_ok ? 4u * (unsigned)(((bc * bd.F + bd.f) * a.S + sl) * Lg) : BUF_OOB;
    }
    const float* const xrow = a.src + (int64_t)rowc * 2 * a.nbins;       // FWD, reflecting bands only
    // cosine / sine slabs of a K-step: thread -> (matrix tid >> 7, column (tid & 127) >> 1, half tid & 1), one float4
    const int csz = ((K2 + 15) & ~15) * bd.ldc;
    const __amdgpu_buffer_rsrc_t rb = buf_rsrc(a.pool + bd.c_off, 4u * (unsigned)(2 * csz));
    const int b_col = (tid & 127) >> 1;
    const unsigned vb = b_col < 16 * ncb ? 4u * (unsigned)((tid >> 7) * csz + b_col * bd.ldc + 4 * (tid & 1)) : BUF_OOB;

    float2 ra[4], rbq[4];      // quarters of t1 = n and of t1 = m - n
    float ma[4], mb[4];        // INV: their masks
    float4 gb;
    int n_cur = 0;             // pair of the set in flight
    auto load_set = [&](int k0) {
        gb = buf_ld4(rb, vb, 4 * k0);
        const int n = k0 + s_p;
        n_cur = n;
        const bool ok = n < K2;
        const bool own = n == 0 || 2 * n == m_;                // its own mirror
        const int t1b = ok && !own ? m_ - n : 0;
        if (!reflect) {
            const unsigned oa = ok ? vx + 8u * (unsigned)n : BUF_OOB, ob = ok && !own ? vx + 8u * (unsigned)t1b : BUF_OOB;
            const unsigned qa = ok ? vm + 4u * (unsigned)n : BUF_OOB, qb = ok && !own ? vm + 4u * (unsigned)t1b : BUF_OOB;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int qq = FWD ? ((q4 + 2) & 3) : q4;      // FWD: x[q] sits at bin bin0 + (q + Lg/2) mod Lg
                ra[q4] = buf_ld2(rx, oa, 8 * qq * m_);
                rbq[q4] = buf_ld2(rx, ob, 8 * qq * m_);
                if (MASKED) { ma[q4] = buf_ld1(rm, qa, 4 * q4 * m_); mb[q4] = buf_ld1(rm, qb, 4 * q4 * m_); }
            }
        } else {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int ia = bd.bin0 + (ok ? n : 0) + ((q4 + 2) & 3) * m_, ib = bd.bin0 + t1b + ((q4 + 2) & 3) * m_;
                if (ia < 0) ia = -ia; else if (ia > a.L / 2) ia = a.L - ia;
                if (ib < 0) ib = -ib; else if (ib > a.L / 2) ib = a.L - ib;
                const float2 va = *reinterpret_cast<const float2*>(xrow + 2 * ia), vb2 = *reinterpret_cast<const float2*>(xrow + 2 * ib);
                ra[q4] = ok && row_ok ? va : make_float2(0.f, 0.f);
                rbq[q4] = ok && !own && row_ok ? vb2 : make_float2(0.f, 0.f);
            }
        }
    };
    // one t1: mask product / window, radix-4 butterfly, twiddles -> the four residues' inputs y_r[t1]
    auto butterfly = [&](const float2 (&raw)[4], const float (&mk)[4], int t1, float2 (&y)[4]) {
        float2 x[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float2 v = raw[q4];
            if (!FWD) {
                // (a rounded product -- contraction is off -- as the layer-4 epilogue stores it when it materialises mask * X)
                if (MASKED) { v.x *= mk[q4]; v.y *= mk[q4]; }
            } else {
                const float g = winl[t1 + q4 * m_];
                float cj = 1.f;
                if (reflect) {
                    const int idx = bd.bin0 + t1 + ((q4 + 2) & 3) * m_;
                    cj = (idx < 0 || idx > a.L / 2) ? -1.f : 1.f;
                }
                v = make_float2(v.x * g, cj * v.y * g);
            }
            x[q4] = v;
        }
        const float2 w1 = twl[t1], w2 = twl[mpad + t1], w3 = twl[2 * mpad + t1];
        const float2 s0 = make_float2(x[0].x + x[2].x, x[0].y + x[2].y), s1 = make_float2(x[1].x + x[3].x, x[1].y + x[3].y);
        const float2 d0 = make_float2(x[0].x - x[2].x, x[0].y - x[2].y), d1 = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
        const float2 y2 = make_float2(s0.x - s1.x, s0.y - s1.y);
        const float2 ym = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - i d1
        const float2 yp = make_float2(d0.x - d1.y, d0.y + d1.x);      // d0 + i d1
        const float2 y1 = FWD ? yp : ym, y3 = FWD ? ym : yp;
        y[0] = make_float2(s0.x + s1.x, s0.y + s1.y);
        // (fused multiply-adds spelled out, contraction off in this kernel: the masked and the plain instantiation must round alike --
        //  the separator's masks-only path and the decode of materialised estimates are held bitwise equal)
        y[1] = make_float2(fmaf(y1.x, w1.x, -(y1.y * w1.y)), fmaf(y1.x, w1.y, y1.y * w1.x));
        y[2] = make_float2(fmaf(y2.x, w2.x, -(y2.y * w2.y)), fmaf(y2.x, w2.y, y2.y * w2.x));
        y[3] = make_float2(fmaf(y3.x, w3.x, -(y3.y * w3.y)), fmaf(y3.x, w3.y, y3.y * w3.x));
    };
    const int a_st = s_row * S4_KP + s_p;
    const int b_st = (tid >> 7) * S4_BROWS * S4_KP + b_col * S4_KP + 4 * (tid & 1);
    auto store_set = [&](int buf) {
        const int n = n_cur;
        const bool ok = n < K2;
        const int t1a = ok ? n : 0, t1b = ok && !(n == 0 || 2 * n == m_) ? m_ - n : 0;
        float2 ya[4], yb[4];
        butterfly(ra, ma, t1a, ya);
        butterfly(rbq, mb, t1b, yb);
        float* Aw = As0 + buf * ABUF + a_st;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Aw[(0 * 4 + r) * D4H_ROWS * S4_KP] = ya[r].x + yb[r].x;
            Aw[(1 * 4 + r) * D4H_ROWS * S4_KP] = ya[r].y + yb[r].y;
            Aw[(2 * 4 + r) * D4H_ROWS * S4_KP] = ya[r].x - yb[r].x;
            Aw[(3 * 4 + r) * D4H_ROWS * S4_KP] = ya[r].y - yb[r].y;
        }
        if (b_col < 16 * ncb) *reinterpret_cast<float4*>(Bs0 + buf * BBUF + b_st) = gb;
    };

    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc[4][2][S4_NCB];                   // [Re P | Im P | Re Q | Im Q][residue of the pair][16-column block]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int cb = 0; cb < S4_NCB; ++cb) acc[q][e][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, kq = lane >> 4;
    const int rh = wave & 1, rp = wave >> 1;
    for (int q4 = 0; q4 < 4; ++q4) { ma[q4] = 1.f; mb[q4] = 1.f; }
    load_set(0);
    float w_mu = 0.f, w_sc = 1.f;
    if (FWD && a.xin) { w_mu = a.mean[bd.jband]; w_sc = a.scale[bd.jband]; }
    {   // tables of the band into LDS
        const float2* tw = reinterpret_cast<const float2*>(a.pool + bd.tw_off);
        const float* win = a.pool + bd.win_off;
        const float2 tv = tid < 3 * mpad ? tw[tid] : make_float2(0.f, 0.f);
        const float wv0 = tid < Lg ? win[tid] : 0.f, wv1 = tid + 256 < Lg ? win[tid + 256] : 0.f;
        if (tid < 3 * mpad) twl[tid] = tv;
        winl[tid] = wv0;
        if (tid + 256 < 4 * D4_MPAD) winl[tid + 256] = wv1;
    }
    __syncthreads();
    store_set(0);
    __syncthreads();
    int cur = 0;
    const int a_rd = ((2 * rp) * D4H_ROWS + 16 * rh + l16) * S4_KP + 2 * kq, b_rd = l16 * S4_KP + 2 * kq;
    auto k_step = [&]() {
        const float* As = As0 + cur * ABUF + a_rd;
        const float* Bs = Bs0 + cur * BBUF + b_rd;
        float2 af[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) af[q][e] = *reinterpret_cast<const float2*>(As + ((q * 4 + e) * D4H_ROWS) * S4_KP);
#pragma unroll
        for (int cb = 0; cb < S4_NCB; ++cb) {
            if (cb >= ncb) continue;
            const float2 c = *reinterpret_cast<const float2*>(Bs + 16 * cb * S4_KP);
            const float2 s = *reinterpret_cast<const float2*>(Bs + (S4_BROWS + 16 * cb) * S4_KP);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc[0][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][e].x, c.x, acc[0][e][cb], 0, 0, 0);
                acc[1][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][e].x, c.x, acc[1][e][cb], 0, 0, 0);
                acc[2][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2][e].x, s.x, acc[2][e][cb], 0, 0, 0);
                acc[3][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3][e].x, s.x, acc[3][e][cb], 0, 0, 0);
                acc[0][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][e].y, c.y, acc[0][e][cb], 0, 0, 0);
                acc[1][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][e].y, c.y, acc[1][e][cb], 0, 0, 0);
                acc[2][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2][e].y, s.y, acc[2][e][cb], 0, 0, 0);
                acc[3][e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3][e].y, s.y, acc[3][e][cb], 0, 0, 0);
            }
            if (XSQ_S4_SCHED >= 0) __builtin_amdgcn_sched_barrier(XSQ_S4_SCHED);
        }
    };
    int k0 = 0;
    if (XSQ_S4_EARLY) {
        if (S4_KP < K2) load_set(S4_KP);
        for (; k0 + S4_KP < K2; k0 += S4_KP) {
            k_step();
            store_set(cur ^ 1);
            if (k0 + 2 * S4_KP < K2) load_set(k0 + 2 * S4_KP);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (; k0 + S4_KP < K2; k0 += S4_KP) {
            load_set(k0 + S4_KP);
            k_step();
            store_set(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    k_step();

    // ---- epilogue.  Register rr of acc[.][e][cb]: row 16 rh + 4 kq + rr, output k = 16 cb + l16 of residue 2 rp + e:
    //   X[k] = P - i Q = (Re P + Im Q, Im P - Re Q)  ->  coefficient q = 4 k + 2 rp + e
    //   X[m - k] = P + i Q = (Re P - Im Q, Im P + Re Q)  ->  q' = 4 (m - k) + 2 rp + e      (not for k = 0, k = m / 2: their own mirrors)
    // (the direction's sign sits in the sine table).  Both residues of a k are adjacent in memory: one 16-byte store per
    // (row, k) and one per (row, m - k).
    const bool rowmajor = !FWD && a.row_len;
    const float* const dbase = rowmajor ? a.dst + 2 * ((int64_t)t.m0 * a.row_len + bd.ent) : a.dst + 2 * (BCS * bd.cum);
    const unsigned dbytes = rowmajor ? 8u * (unsigned)(D4H_ROWS * a.row_len) : 8u * a.BC * blk;
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(dbase, dbytes);
    const __amdgpu_buffer_rsrc_t rxin = buf_rsrc(FWD && a.xin ? a.xin + BCS * bd.cum : a.dst, FWD && a.xin ? 4u * a.BC * blk : 0u);
    unsigned vrow[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r0 = 16 * rh + 4 * kq + rr;
        const int mr = t.m0 + r0, mc = mr < M ? mr : 0;
        unsigned o = 8u * (unsigned)(r0 * a.row_len);
        if (!rowmajor) {
            const int rb_ = mc / a.S, rs = mc - rb_ * a.S;
            o = 8u * (unsigned)(((rb_ * bd.F + bd.f) * a.S + rs) * Lg);
        }
        vrow[rr] = mr < M ? o : BUF_OOB;
    }
    const bool split = a.split;
#pragma unroll
    for (int cb = 0; cb < S4_NCB; ++cb) {
        if (cb >= ncb) continue;
        const int k = 16 * cb + l16;
        const bool k_ok = k < K2, mir_ok = k_ok && k != 0 && 2 * k != m_;
        const int q1 = 4 * k + 2 * rp, q2 = 4 * (m_ - k) + 2 * rp;       // first residue of the pair; q2 < Lg only when mir_ok
        float2 w1 = make_float2(1.f, 1.f), w2 = make_float2(1.f, 1.f);
        unsigned p1 = 8u * (unsigned)q1, p2 = 8u * (unsigned)(mir_ok ? q2 : 0);
        if (!FWD) {
            w1 = *reinterpret_cast<const float2*>(&winl[k_ok ? q1 : 0]);
            w2 = *reinterpret_cast<const float2*>(&winl[mir_ok ? q2 : 0]);
            p1 += 16u * (unsigned)m_; if (p1 >= 8u * (unsigned)Lg) p1 -= 8u * (unsigned)Lg;      // spectrum position p = (q + Lg/2) mod Lg
            p2 += 16u * (unsigned)m_; if (p2 >= 8u * (unsigned)Lg) p2 -= 8u * (unsigned)Lg;
        }
        const unsigned o1 = k_ok ? p1 : BUF_OOB_COL, o2 = mir_ok ? p2 : BUF_OOB_COL;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            float4 v1, v2;
            v1.x = (acc[0][0][cb][rr] + acc[3][0][cb][rr]) * w1.x; v1.y = (acc[1][0][cb][rr] - acc[2][0][cb][rr]) * w1.x;
            v1.z = (acc[0][1][cb][rr] + acc[3][1][cb][rr]) * w1.y; v1.w = (acc[1][1][cb][rr] - acc[2][1][cb][rr]) * w1.y;
            v2.x = (acc[0][0][cb][rr] - acc[3][0][cb][rr]) * w2.x; v2.y = (acc[1][0][cb][rr] + acc[2][0][cb][rr]) * w2.x;
            v2.z = (acc[0][1][cb][rr] - acc[3][1][cb][rr]) * w2.y; v2.w = (acc[1][1][cb][rr] + acc[2][1][cb][rr]) * w2.y;
            const unsigned vo1 = vrow[rr] + o1, vo2 = vrow[rr] + o2;
            buf_st4(v1, rd, vo1, 0);
            buf_st4(v2, rd, vo2, 0);
            if (FWD && a.xin) {
                float2 u1 = make_float2(whiten_mag(v1.x, v1.y, w_mu, w_sc), whiten_mag(v1.z, v1.w, w_mu, w_sc));
                float2 u2 = make_float2(whiten_mag(v2.x, v2.y, w_mu, w_sc), whiten_mag(v2.z, v2.w, w_mu, w_sc));
                if (split) { bf3_words2(u1.x, u1.y, u1.x, u1.y); bf3_words2(u2.x, u2.y, u2.x, u2.y); }
                buf_st2(u1, rxin, (vo1 >> 1) | (vo1 & (BUF_OOB | BUF_OOB_COL)), 0);
                buf_st2(u2, rxin, (vo2 >> 1) | (vo2 & (BUF_OOB | BUF_OOB_COL)), 0);
            }
        }
    }
}

}  // namespace xsq
