// Per-band DFT with one radix-4 decimation-in-frequency stage fused into the operand staging.
//
// Every band length is a multiple of 4 (Lg = 4m).  For X[q] = sum_t x[t] w^(q t), w = exp(-+2 pi i / Lg):
//   X[4k + r] = sum_{t1 < m}  y_r[t1] * exp(-+2 pi i k t1 / m),
//   y_r[t1]   = w^(r t1) * sum_{a < 4} x[t1 + a m] * (-+i)^(r a)          (r = 0..3)
// i.e. four length-m DFTs that share ONE m x m matrix: 4x fewer MFMA flops than the dense Lg x Lg
// product of the generic engine (gemm_tile.h), which keeps the bands with Lg < 24.
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
    // band_dft4s.h (pair-contracted form): K2 = m / 2 + 1 pairs / outputs; Ct[k][n] = cos(2 pi n k / m) at c_off, rows of ldc =
    // round_up(K2, 8) floats, round_up(K2, 16) rows; St[k][n] = -+sin(2 pi n k / m) (the direction's sign) right behind it
    int K2, ldc;
    int64_t c_off;
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
#ifndef XSQ_D4_ABL
#define XSQ_D4_ABL 0        // diagnostic build (timings only, results wrong): 1 no MFMAs, 2 no operand loads inside the K loop, 4 no epilogue stores
#endif

constexpr int D4_LD = 20, D4_MPAD = 80;     // bands up to Lg = 320 (the plan builder routes longer ones to the dense engine)
#ifndef XSQ_D4_ROWS
#define XSQ_D4_ROWS 32          // rows of a tile: 32 (4 waves, three workgroups per CU) or 64 (8 waves, two workgroups per CU: an A/B arm)
#endif
constexpr int D4H_ROWS = XSQ_D4_ROWS, D4H_NT = 8 * D4H_ROWS;
static_assert(D4H_ROWS == 32 || D4H_ROWS == 64, "tile height");

// Vector issue.  On gfx950 an fp32 MFMA runs at exactly the rate of the SIMD's packed-fp32 vector ALU, and measured it does
// not overlap with the vector instructions of the SIMD's other waves: a K-step of this kernel takes the SUM of its MFMA
// cycles (256 per 16-column block and wave) and 4 cycles per other vector instruction, times the three resident waves
// (ablations r4a, profiles/r05_ab_runs.txt: no MFMAs -10 %, no K-loop loads -5 %, both -46 %; K-step 0.56 + 0.41 ncb us; tools/valu_mfma.py
// counts the two from the assembly).  The round-3 form spent 108 vector instructions per K-step and 63 per 16-column
// block of the epilogue -- as many issue cycles as the MFMAs of a 4-block band -- two thirds of them address arithmetic:
// 64-bit pointer sums per load, clamps, selects around loads and stores.  Here every operand and result goes through a
// BUFFER descriptor: the per-thread offset is formed once per tile, the K-step / quarter / column-block displacement is a
// scalar offset, rows past the end and columns past the band are switched off by an out-of-range offset (loads return 0,
// stores are dropped) instead of clamps and predicates.
// NCBMAX = most 16-column blocks a band of this instantiation has.  10 (Lg <= 320): 49 KB of LDS and 80 accumulator
// registers, three workgroups per CU -- the product configuration.  5 (Lg <= 160): 35 KB and 40 accumulator registers
// -> FOUR workgroups per CU, an A/B arm (XSQ_D4_SPLIT=1) that measured no faster.
template <int NCBMAX> struct D4Cfg { static constexpr int waves = D4H_ROWS == 64 ? 4 : (NCBMAX > 5 ? 3 : 4), mpad = NCBMAX > 5 ? D4_MPAD : 40; };
// MASKED (synthesis only): the coefficients are mask * mix, formed on the way in (Band4Args.mask).
template <bool FWD, int NCBMAX = 10, bool MASKED = false>
__global__ __launch_bounds__(D4H_NT) __attribute__((amdgpu_waves_per_eu(D4Cfg<NCBMAX>::waves, D4Cfg<NCBMAX>::waves)))
void band_dft4_full_kernel(Band4Args a, const Tile4Dev* __restrict__ tiles, int ntiles) {
    static_assert(!(FWD && MASKED), "the mask product belongs to the synthesis");
    constexpr int D4H_NCB = NCBMAX, MPADL = D4Cfg<NCBMAX>::mpad;
    __shared__ __attribute__((aligned(16))) float lds[2 * (4 * D4H_ROWS + 16 * D4H_NCB) * D4_LD];
    __shared__ __attribute__((aligned(16))) float2 twl[3 * MPADL];        // w^(r t1), r = 1..3, of this tile's band
    __shared__ __attribute__((aligned(16))) float winl[4 * MPADL];        // window of this tile's band: g'[q] (FWD) / wd[q] (INV)
    constexpr int ABUF = 4 * D4H_ROWS * D4_LD, BBUF = 16 * D4H_NCB * D4_LD;
    float* const As0 = lds;                     // [buf][r][row][20]
    float* const Bs0 = lds + 2 * ABUF;          // [buf][col][20]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    XSQ_D4S(0);
    const Tile4Dev t = tiles[xcd_remap(blockIdx.x, ntiles)];     // by value: one batched scalar load
    const int ncb = t.ncb;                      // 16-column blocks of the band, 1..10 (uniform)
    const Band4Dev bd = t.bd;
    const int m_ = bd.m, Lg = bd.Lg, K = 2 * m_, M = a.BC * a.S;
    const int64_t BCS = (int64_t)a.BC * a.S;
    const int mpad = (m_ + 7) & ~7;

    // ---- staging assignment: row s_row, complex t1 = K-step base + s_t ---------------------------
    const int s_row = tid >> 3, s_t = tid & 7;
    const int row = t.m0 + s_row;
    const bool row_ok = row < M;
    const int rowc = row_ok ? row : M - 1;
    const int bc = rowc / a.S, sl = rowc - bc * a.S;
    constexpr bool masked = MASKED;
    // FWD: a band whose window leaves [0, L/2] reads Hermitian reflections (the lowest and the highest bands): pointer path
    const bool reflect = FWD && (bd.bin0 < 0 || bd.bin0 + Lg - 1 > a.L / 2);
    // operand descriptors and this thread's offsets inside them (bytes)
    __amdgpu_buffer_rsrc_t rx, rm = buf_rsrc(a.src, 0);
    unsigned vx, vm = BUF_OOB;
    const unsigned blk = (unsigned)(bd.F * Lg) * (unsigned)a.S;          // elements of one channel in the band's block
    if (FWD) {                                   // spectrum rows of this tile: x[q] sits at bin bin0 + (q + Lg/2) mod Lg
        rx = buf_rsrc(a.src + 2 * ((int64_t)t.m0 * a.nbins), 8u * D4H_ROWS * a.nbins);
        vx = row_ok ? 8u * (unsigned)(s_row * a.nbins + bd.bin0 + s_t) : BUF_OOB;
    } else if (!masked) {
        rx = buf_rsrc(a.src + 2 * (BCS * bd.cum), 8u * a.BC * blk);
        vx = row_ok ? 8u * (unsigned)(((bc * bd.F + bd.f) * a.S + sl) * Lg + s_t) : BUF_OOB;
    } else {
        rx = buf_rsrc(a.src + 2 * ((int64_t)a.BCx * a.S * bd.cum), 8u * a.BCx * blk);
        rm = buf_rsrc(a.mask + BCS * bd.cum, 4u * a.BC * blk);
        vx = row_ok ? 8u * (unsigned)((((bc % a.BCx) * bd.F + bd.f) * a.S + sl) * Lg + s_t) : BUF_OOB;
        vm = row_ok ? 4u * (unsigned)(((bc * bd.F + bd.f) * a.S + sl) * Lg + s_t) : BUF_OOB;
    }
    const float* const xrow = a.src + (int64_t)rowc * 2 * a.nbins;       // FWD, reflecting bands only
    // DFT matrix slab of a K-step: 16 ncb rows (n) x 16 floats = 64 ncb float4, item i = tid + 256 u: n = i >> 2, k quad i & 3.
    // The pool holds round_up(2m, 64) rows (zero past 2m), so every item of a requested group lies inside it.
    const __amdgpu_buffer_rsrc_t rb = buf_rsrc(a.pool + bd.d_off, 4u * (unsigned)(((K + 63) & ~63) * bd.ldd));
    const unsigned vb = 4u * (unsigned)((tid >> 2) * bd.ldd + 4 * (tid & 3));
    const int sb64 = 4 * (D4H_NT / 4) * bd.ldd;  // D4H_NT / 4 rows further down
    const int nb4 = 64 * ncb;
    constexpr int NBU = (64 * D4H_NCB + D4H_NT - 1) / D4H_NT;     // float4 items of the matrix slab per thread (3 / 2)

    float2 raw[4];         // quarter a, complex t1 (FWD: spectrum value of the quarter)
    float aux[4];          // INV: mask of quarter a
    float4 gb[NBU];
#pragma unroll
    for (int u = 0; u < NBU; ++u) gb[u] = make_float4(0.f, 0.f, 0.f, 0.f);

    // load_set only issues loads; products, selects and butterflies in store_set.  The matrix slab (L2) is requested first.
    auto load_set = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NBU; ++u)       // uniform test
            if (D4H_NT * u < nb4) gb[u] = buf_ld4(rb, vb, u * sb64 + 4 * k0);
        if (!FWD) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                raw[q4] = buf_ld2(rx, vx, 4 * k0 + 8 * q4 * m_);
                if (masked) aux[q4] = buf_ld1(rm, vm, 2 * k0 + 4 * q4 * m_);
            }
        } else if (!reflect) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) raw[q4] = buf_ld2(rx, vx, 4 * k0 + 8 * ((q4 + 2) & 3) * m_);
        } else {
            const int t1 = (k0 >> 1) + s_t, tc = t1 < m_ ? t1 : m_ - 1;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                int idx = bd.bin0 + tc + ((q4 + 2) & 3) * m_;
                if (idx < 0) idx = -idx;
                else if (idx > a.L / 2) idx = a.L - idx;
                raw[q4] = *reinterpret_cast<const float2*>(xrow + 2 * idx);
            }
        }
    };
    // LDS addresses of this thread's staging stores and of the twiddle / window reads (floats)
    const int a_st = s_row * D4_LD + 2 * s_t, b_st = (tid >> 2) * D4_LD + 4 * (tid & 3);
    auto store_set = [&](int buf, int k0) {
        const int t1 = (k0 >> 1) + s_t;          // < mpad always (mpad = 8 * K-steps)
        float2 x[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float2 v = raw[q4];
            if (!FWD) {
                if (masked) { v.x *= aux[q4]; v.y *= aux[q4]; }
            } else {
                const float g = winl[t1 + q4 * m_];          // window value (0 past the band: table zero-filled)
                float cj = 1.f;
                if (reflect) {
                    const int idx = bd.bin0 + (t1 < m_ ? t1 : m_ - 1) + ((q4 + 2) & 3) * m_;
                    cj = (idx < 0 || idx > a.L / 2) ? -1.f : 1.f;
                }
                v = make_float2(v.x * g, cj * v.y * g);
            }
            x[q4] = v;
        }
        if (k0 + 16 > K) {                       // the band's last, partial K-step: complex t1 >= m belong to the next quarter
            const bool ok = t1 < m_;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) x[q4] = ok ? x[q4] : make_float2(0.f, 0.f);
        }
        const float2 w1 = twl[t1], w2 = twl[mpad + t1], w3 = twl[2 * mpad + t1];
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
        float* Aw = As0 + buf * ABUF + a_st;
        *reinterpret_cast<float2*>(Aw) = y0;
        *reinterpret_cast<float2*>(Aw + D4H_ROWS * D4_LD) = z1;
        *reinterpret_cast<float2*>(Aw + 2 * D4H_ROWS * D4_LD) = z2;
        *reinterpret_cast<float2*>(Aw + 3 * D4H_ROWS * D4_LD) = z3;
        float* Bw = Bs0 + buf * BBUF + b_st;
#pragma unroll
        for (int u = 0; u < NBU; ++u)
            if (tid + D4H_NT * u < nb4) *reinterpret_cast<float4*>(Bw + (D4H_NT / 4) * u * D4_LD) = gb[u];
    };

    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc[2][D4H_NCB];                     // [residue of the pair][16-column block]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int cb = 0; cb < D4H_NCB; ++cb) acc[e][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int l16 = lane & 15, kq = lane >> 4;
    const int rh = wave & (D4H_ROWS / 16 - 1), rp = wave / (D4H_ROWS / 16);      // row block of 16, residue pair
    load_set(0);
    float w_mu = 0.f, w_sc = 1.f;                // whitening constants of this tile's band (uniform)
    if (FWD && a.xin) { w_mu = a.mean[bd.jband]; w_sc = a.scale[bd.jband]; }
    {   // tables of the band into LDS: every value of this thread requested before the first is stored
        const float2* tw = reinterpret_cast<const float2*>(a.pool + bd.tw_off);
        const float* win = a.pool + bd.win_off;
        const float2 tv = tid < 3 * mpad ? tw[tid] : make_float2(0.f, 0.f);
        const float wv0 = tid < Lg ? win[tid] : 0.f, wv1 = (D4H_NT == 256 && tid + 256 < Lg) ? win[tid + 256] : 0.f;
        if (tid < 3 * mpad) twl[tid] = tv;
        if (tid < 4 * MPADL) winl[tid] = wv0;     // zero past Lg: the FWD staging reads up to 4 mpad - 1
        if (D4H_NT == 256 && tid + 256 < 4 * MPADL) winl[tid + 256] = wv1;
    }
    __syncthreads();             // tables complete (store_set reads the twiddles)
    store_set(0, 0);
    XSQ_D4S(1);
    __syncthreads();
    int cur = 0;
    const int a_rd = (2 * rp * D4H_ROWS + 16 * rh + l16) * D4_LD + 4 * kq, b_rd = l16 * D4_LD + 4 * kq;
    auto k_step = [&]() {
        const float* As = As0 + cur * ABUF + a_rd;
        const float* Bs = Bs0 + cur * BBUF + b_rd;
        const float4 a0 = *reinterpret_cast<const float4*>(As);
        const float4 a1 = *reinterpret_cast<const float4*>(As + D4H_ROWS * D4_LD);
        float4 bf[2];                             // fragment of block cb in bf[cb & 1]; the next block's is read under this block's MFMAs
        bf[0] = *reinterpret_cast<const float4*>(Bs);
#pragma unroll
        for (int cb = 0; cb < D4H_NCB; ++cb) {
            if (cb >= ncb) continue;              // (not break: the compiler refuses to unroll the multi-exit loop and moves acc to scratch)
            if (cb + 1 < D4H_NCB && cb + 1 < ncb) bf[(cb + 1) & 1] = *reinterpret_cast<const float4*>(Bs + 16 * (cb + 1) * D4_LD);
            const float4 b = bf[cb & 1];
#if XSQ_D4_ABL & 1
            asm volatile("" :: "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(a0.x), "v"(a0.w), "v"(a1.x), "v"(a1.w));
            continue;
#endif
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
        if (!(XSQ_D4_ABL & 2)) load_set(k0 + 16);
        k_step();
        store_set(cur ^ 1, k0 + 16);
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
    // rows rr = 2, 3: each takes from its partner (DPP quad_perm [1, 0, 3, 2], folded into the select) the other half of
    // its rows.  One 16-byte buffer store per row and block; the block's displacement (32 q = 256 bytes) is the scalar
    // offset, so a lane's offsets are formed once: row offset + position, switched out of range for rows past M and for
    // columns past the band.
    auto swap1 = [](float v) {                   // value of lane l ^ 1
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    };
    const bool part = lane & 1;
    const bool rowmajor = !FWD && a.row_len;     // phase-ordered rows of row_len complex (slice_fft.h) instead of the arena
    // (one descriptor from scalar selects: two descriptors joined by a branch became a per-lane value and every store a
    // readfirstlane loop)
    const float* const dbase = rowmajor ? a.dst + 2 * ((int64_t)t.m0 * a.row_len + bd.ent) : a.dst + 2 * (BCS * bd.cum);
    const unsigned dbytes = rowmajor ? 8u * (unsigned)(D4H_ROWS * a.row_len) : 8u * a.BC * blk;
    const __amdgpu_buffer_rsrc_t rd = buf_rsrc(dbase, dbytes);
    const __amdgpu_buffer_rsrc_t rxin = buf_rsrc(FWD && a.xin ? a.xin + BCS * bd.cum : a.dst, FWD && a.xin ? 4u * a.BC * blk : 0u);
    unsigned vrow[2];                            // byte offset of (row, q = 0) inside rd
    {
        const int r0 = 16 * rh + 4 * kq + 2 * (int)part;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const int mr = t.m0 + r0 + sx, mc = mr < M ? mr : 0;
            unsigned o = 8u * (unsigned)((r0 + sx) * a.row_len);
            if (!rowmajor) {
                const int rb_ = mc / a.S, rs = mc - rb_ * a.S;
                o = 8u * (unsigned)(((rb_ * bd.F + bd.f) * a.S + rs) * Lg);
            }
            vrow[sx] = mr < M ? o : BUF_OOB;
        }
    }
    const bool split = a.split;
    // q of this lane in block 0 and the byte offset of its position; block cb adds 32 q (the position wraps at Lg once)
    const int q0 = 4 * (l16 >> 1) + 2 * rp;
    unsigned p8 = 8u * (unsigned)q0;
    if (!FWD) { p8 += 16u * (unsigned)m_; if (p8 >= 8u * (unsigned)Lg) p8 -= 8u * (unsigned)Lg; }    // spectrum position p = (q + Lg/2) mod Lg
#pragma unroll
    for (int cb = 0; cb < D4H_NCB; ++cb) {
        if (cb >= ncb) continue;
        float w0 = 1.f, w1 = 1.f;
        if (!FWD) {
            const float2 w = *reinterpret_cast<const float2*>(&winl[q0 + 32 * cb]);      // (table is zero past Lg)
            w0 = w.x; w1 = w.y;
        }
        unsigned qoff = p8;
        if (cb + 1 == ncb) qoff = q0 + 32 * cb < Lg ? p8 : BUF_OOB_COL;       // columns past the band: only in its last block
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            // even lane: keeps Re of row sx, takes Im of row sx;  odd lane: keeps Im of row 2 + sx, takes Re of row 2 + sx
            // (both exchanges run on every lane -- a DPP move under a lane condition becomes a branch -- and are selected)
            const float x0 = swap1(acc[0][cb][2 + sx]), y0 = swap1(acc[0][cb][sx]);
            const float x1 = swap1(acc[1][cb][2 + sx]), y1 = swap1(acc[1][cb][sx]);
            float4 v;
            v.x = part ? x0 : acc[0][cb][sx];
            v.y = part ? acc[0][cb][2 + sx] : y0;
            v.z = part ? x1 : acc[1][cb][sx];
            v.w = part ? acc[1][cb][2 + sx] : y1;
            v.x *= w0; v.y *= w0; v.z *= w1; v.w *= w1;
            const unsigned vo = vrow[sx] + qoff;         // either switch alone or both together stay past every range
            if (!(XSQ_D4_ABL & 4)) buf_st4(v, rd, vo, 0);
            if (FWD && a.xin) {
                float2 o = make_float2(whiten_mag(v.x, v.y, w_mu, w_sc), whiten_mag(v.z, v.w, w_mu, w_sc));
                if (split) bf3_words2(o.x, o.y, o.x, o.y);
                buf_st2(o, rxin, (vo >> 1) | (vo & (BUF_OOB | BUF_OOB_COL)), 0);
            }
        }
        const unsigned pn = p8 + 256u, pw = pn - 8u * (unsigned)Lg;         // (wraps to a huge value while pn < 8 Lg)
        p8 = FWD ? pn : (pn < pw ? pn : pw);
    }
    XSQ_D4S(3);
}

}  // namespace xsq
