// LDS-resident real FFT of one slice (length L = 18060 = 2 * 43 * 14 * 15) for gfx950.
//
// rocFFT needs Bluestein for the prime factor 43 (7 kernels and ~6 passes over HBM per
// transform; 4.5 % of the HBM roofline measured in profiles/r01).  Here one workgroup owns one
// (channel, slice) row: the half-length complex sequence z[n] = v[2n] + i v[2n+1] (9030 points,
// 72,240 B) lives in LDS for the whole transform, so HBM sees each row exactly once in and once
// out, and the neighbouring streaming kernels are fused in:
//   forward  k_slice_rfft : slice + Tukey window (nsgt/slicing.py:21-72) on the load side,
//                           radix 43 x 14 x 15 Cooley-Tukey in LDS, real post-processing on the store side
//   inverse  k_slice_irfft: gather-sum of the per-band synthesis spectra (nsgt/nsigtf.py:85-95) and
//                           real pre-processing on the load side, same FFT with conjugate twiddles
// Index maps (N = 9030, M1 = 210):
//   step 1: n = n1*210 + m,   DFT_43 over n1,  x W_N^(m*k1)        -> Z[k1*210 + m]
//   step 2: m = n2*15 + n3,   DFT_14 over n2,  x W_N^(43*n3*k2)    -> Z[k1*210 + k2*15 + n3]
//   step 3:                   DFT_15 over n3                        -> Z[k1*210 + k2*15 + k3] = X[k1 + 43*k2 + 602*k3]
// Small DFTs pair x[n] +- x[R-n] (R^2/2 real FMAs instead of 2R^2) with compile-time twiddles.
#pragma once
#include <type_traits>

#include "common.h"
#include "dft_tables.h"

#ifndef XSQ_FFT_STAMP
#define XSQ_FFT_STAMP 0             // diagnostic build: phase time stamps of k_slice_irfft (tools/fft_phases.py)
#endif
#ifndef XSQ_FFT_DEPHASE
#define XSQ_FFT_DEPHASE 0           // s_sleep(127) units (~3.9 us each) by which odd workgroups of the first wave start late
#endif
#ifndef XSQ_FFT_DEPHASE_SLOTS
#define XSQ_FFT_DEPHASE_SLOTS 512   // workgroup slots of the chip for the 512-thread transforms (256 CUs x 2)
#endif
#ifndef XSQ_ABLATE
#define XSQ_ABLATE 0      // diagnostic builds (tools/ablate.sh): 16 no gather, 32 no radix-43, 64 no steps 2/3, 128 no output
#endif

namespace xsq {

constexpr int FFT_L = 18060, FFT_N = 9030, FFT_R1 = 43, FFT_R2 = 14, FFT_R3 = 15, FFT_M1 = 210;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// Complex products with the fused multiply-adds spelled out (and contraction off around them): left to the compiler,
// `a.x b.x - a.y b.y` may become fma(a.x, b.x, -(a.y b.y)) in one instantiation of a kernel and fma(-a.y, b.y, a.x b.x)
// in another -- the scalar and the packed-butterfly builds of the forward transform differed in the last bit of 90 % of
// their outputs that way (round 3), and "the packed kernels return the bits of the scalar ones" is the property the
// packed-fp32 hazard tests stand on.
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
#pragma clang fp contract(off)
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {   // a * conj(b)
#pragma clang fp contract(off)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}

// ---- packed-fp32 forms (PK = true) ---------------------------------------------------------------------------------
// The butterflies below do the SAME thing to the real and the imaginary part of a value with one constant:
// P.x = fma(a.x, c, P.x); P.y = fma(a.y, c, P.y) -- one v_pk_fma_f32 on the (x, y) register pair with the constant
// broadcast from a uniform register.  Each lane of a packed instruction is the IEEE operation of its scalar twin, in
// the same order on the same operands: the PK codelets are BITWISE the scalar ones at half the vector instructions
// (the transforms are bound by their instruction count: 64 % vector-ALU issue utilisation, profiles/r03z_pmc_sq.csv).
// Hand-placed because the compiler's own packing needs more registers than the 128 of a 512-thread workgroup pair
// (122 spilled registers, 0.66 -> 1.09 ms, round 3) and because the library is BUILT with packed ops off (Makefile:
// next to split-bf16 MFMAs of another stream a packed-ops transform returned wrong values; the packed kernels are only
// launched while the model runs its contractions in fp32 -- xsq_plan_set_packed_fft).  Operand selects / negates:
// tools/probe/pk_ops.hip pins their meaning on the hardware.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v2(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// a + i b = (a.x - b.y, a.y + b.x)   /   a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f pk_add_i(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub_i(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// acc + a * u, u = the LOW (HI = false) or HIGH half of the uniform pair cs, negated when NEG
template <bool HI, bool NEG>
__device__ __forceinline__ v2f pk_fma_u(v2f a, v2f cs, v2f acc) {
    v2f d;
    if constexpr (!HI && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(cs), "v"(acc));
    else if constexpr (!HI && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(cs), "v"(acc));
    else if constexpr (HI && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(cs), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(cs), "v"(acc));
    return d;
}

// X[k] = sum_n x[n] exp(SIGN * 2 pi i n k / R), results handed to emit(k, X[k]) as they are produced.
// PARTS = 2 splits the OUTPUTS between two callers that hold the same inputs (two wave groups of a 512-thread
// workgroup): PART 0 emits k = 0 and the pairs k = 1 .. KSPLIT, PART 1 the pairs KSPLIT + 1 .. H (odd R only).
template <int R, int SIGN, int PART = 0, int PARTS = 1, bool PK = false, class Emit>
__device__ __forceinline__ void dft_small(const float2 (&x)[R], Emit&& emit) {
    constexpr int H = (R - 1) / 2;
    constexpr bool EVEN = (R % 2) == 0;
    static_assert(PARTS == 1 || !EVEN, "split outputs: odd radix only");
    constexpr int KSPLIT = (H + 1) / 2;
    constexpr int K_LO = PARTS == 1 ? 1 : (PART == 0 ? 1 : KSPLIT + 1);
    constexpr int K_HI = PARTS == 1 ? H : (PART == 0 ? KSPLIT : H);
    if constexpr (PK) {
        v2f a[H], b[H];
        const v2f x0 = to_v2(x[0]);
        v2f s0 = x0;
        static_for<1, H + 1>([&](auto nc) {
            constexpr int n = nc;
            a[n - 1] = pk_add(to_v2(x[n]), to_v2(x[R - n]));
            b[n - 1] = pk_sub(to_v2(x[n]), to_v2(x[R - n]));
            s0 = pk_add(s0, a[n - 1]);
        });
        if constexpr (EVEN) s0 = pk_add(s0, to_v2(x[R / 2]));
        if constexpr (PART == 0) emit(0, to_f2(s0));
        static_for<K_LO, K_HI + 1>([&](auto kc) {
            constexpr int k = kc;
            v2f P = x0;
            if constexpr (EVEN) P = (k & 1) ? pk_sub(P, to_v2(x[R / 2])) : pk_add(P, to_v2(x[R / 2]));
            v2f Q = v2f{0.f, 0.f};
            static_for<1, H + 1>([&](auto nc) {
                constexpr int n = nc;
                constexpr int i = (n * k) % R;
                constexpr float c = DftTw<R>::c[i];
                constexpr float sn = DftTw<R>::s[i];
                constexpr float sa = sn < 0.f ? -sn : sn;
                const v2f cs = v2f{c, sa};                       // uniform pair: cos in the low half, |sin| in the high half
                P = pk_fma_u<false, false>(a[n - 1], cs, P);     // P += a * c
                Q = pk_fma_u<true, (sn < 0.f)>(b[n - 1], cs, Q); // Q += b * s
            });
            // X[k] = P + SIGN*i*Q,  X[R-k] = P - SIGN*i*Q
            if constexpr (SIGN > 0) { emit(k, to_f2(pk_add_i(P, Q))); emit(R - k, to_f2(pk_sub_i(P, Q))); }
            else { emit(k, to_f2(pk_sub_i(P, Q))); emit(R - k, to_f2(pk_add_i(P, Q))); }
        });
        if constexpr (EVEN) {
            v2f P = ((R / 2) & 1) ? pk_sub(x0, to_v2(x[R / 2])) : pk_add(x0, to_v2(x[R / 2]));
            static_for<1, H + 1>([&](auto nc) {
                constexpr int n = nc;
                P = (n & 1) ? pk_sub(P, a[n - 1]) : pk_add(P, a[n - 1]);
            });
            emit(R / 2, to_f2(P));
        }
        return;
    }
    float2 a[H], b[H];
    float2 s0 = x[0];
    static_for<1, H + 1>([&](auto nc) {
        constexpr int n = nc;
        a[n - 1] = c_add(x[n], x[R - n]);
        b[n - 1] = c_sub(x[n], x[R - n]);
        s0 = c_add(s0, a[n - 1]);
    });
    if constexpr (EVEN) s0 = c_add(s0, x[R / 2]);
    if constexpr (PART == 0) emit(0, s0);
    static_for<K_LO, K_HI + 1>([&](auto kc) {
        constexpr int k = kc;
        float2 P = x[0];
        if constexpr (EVEN) P = (k & 1) ? c_sub(P, x[R / 2]) : c_add(P, x[R / 2]);
        float2 Q = make_float2(0.f, 0.f);
        static_for<1, H + 1>([&](auto nc) {
            constexpr int n = nc;
            constexpr float c = DftTw<R>::c[(n * k) % R];
            constexpr float s = DftTw<R>::s[(n * k) % R];
            P.x = fmaf(a[n - 1].x, c, P.x);
            P.y = fmaf(a[n - 1].y, c, P.y);
            Q.x = fmaf(b[n - 1].x, s, Q.x);
            Q.y = fmaf(b[n - 1].y, s, Q.y);
        });
        // X[k] = P + SIGN*i*Q,  X[R-k] = P - SIGN*i*Q,  i*Q = (-Q.y, Q.x)
        const float2 iq = make_float2(-SIGN * Q.y, SIGN * Q.x);
        emit(k, c_add(P, iq));
        emit(R - k, c_sub(P, iq));
    });
    if constexpr (EVEN) {
        float2 P = ((R / 2) & 1) ? c_sub(x[0], x[R / 2]) : c_add(x[0], x[R / 2]);
        static_for<1, H + 1>([&](auto nc) {
            constexpr int n = nc;
            P = (n & 1) ? c_sub(P, a[n - 1]) : c_add(P, a[n - 1]);
        });
        emit(R / 2, P);
    }
}

// LDS position of output bin k (k = k1 + 43*k2 + 602*k3)
__device__ __forceinline__ int fft_pos(int k) {
    const int k3 = k / (FFT_R1 * FFT_R2);
    const int r = k - k3 * (FFT_R1 * FFT_R2);
    const int k2 = r / FFT_R1;
    const int k1 = r - k2 * FFT_R1;
    return k1 * FFT_M1 + k2 * FFT_R3 + k3;
}

// Twiddle tables of one plan, laid out so that consecutive lanes read consecutive entries:
//   w1[k1*210 + m] = W_N^(m*k1)   w2[k2*15 + n3] = W_N^(43*n3*k2)   wl[k] = W_L^k (real pre/post-processing)
struct FftTables {
    const float2* w1;
    const float2* w2;
    const float2* wl;
};

// Inverse-side gather schedule.  The synthesis GEMM writes each row's 18640 band values
// row-major, bands ordered by phase (band index mod 4) -- bands of one phase never overlap in
// the spectrum, so a phase is accumulated into LDS with plain read-modify-writes (no atomics,
// fixed order => bitwise reproducible) from contiguous, independent, coalesced global loads.
struct GatherSched {
    const int* tgt;        // (sumLg) target bin of each entry, -1 when outside [0, N]
    const unsigned short* tgt16;   // the same as 16-bit values (0xFFFF = none), padded: entries (2j, 2j + 1) load as one dword
    int begin[5];          // entry range of phase p is [begin[p], begin[p+1])
    int lo[4];             // first entry of phase p that is gathered from Z (= begin[p] unless the short bands run in-kernel)
    int row_len;           // entries per row (sum of band lengths)
};

// Short bands (Lg = 4m < 64) synthesised INSIDE k_slice_irfft, straight from the coefficient arena (or from
// mask * mix): for them the dense DFT-matrix GEMM (band_synthesis_gemm) moved 8 Lg^2 flops and a Z round trip per
// band and row for what is a 16..60-point FFT.  Here: one radix-4 decimation-in-frequency stage while the
// coefficients are loaded (one lane per (band, t1): the four quarters x[t1 + a m] are contiguous runs), the four
// m-point DFTs per band from compile-time-twiddle codelets (dft_small<4..15>, one lane per (band, residue), in
// place in an LDS scratch area), and a 4-phase accumulation into the spectrum bins (bands of one phase are
// disjoint: plain read-modify-write, fixed order, bitwise reproducible) with the dual window applied on the way.
struct ShortItem1 {        // one (band, t1) butterfly
    int cum, F, f, Lg, t1, sc;    // block offset (complex per channel-slice), block rows, row, band length, t1, scratch offset of the band
};
struct ShortSched {
    const ShortItem1* item1;   // n1 butterflies, band-major
    const float2* tw1;         // 3 per butterfly: w^(r t1), r = 1..3, w = exp(-2 pi i / Lg)
    const int* item2;          // n2 m-point DFTs sorted by m: (scratch offset of (band, r)) << 4 | m
    const int* stgt;           // nent: spectrum bin of scratch entry e (phase-major), -1 outside [0, N]
    const float* swd;          // nent: dual window * Lg * sign / L of that entry
    int n1, n2, nent, sc0;     // sc0: first spectrum slot used as scratch (above every short band's bins)
    int begin[5];              // scratch entry range of phase p
};
struct ShortIn {           // per call
    const float* coef;     // coefficient arena (BC channels), or the mix arena (BCx channels) when mask != nullptr
    const float* mask;     // optional real mask arena (BC channels)
    int BC, BCx;
};

template <int R>
__device__ __forceinline__ void short_dft(float2* base) {      // in place: X[k] = sum_n x[n] exp(-2 pi i n k / R)
    float2 v[R];
#pragma unroll
    for (int n = 0; n < R; ++n) v[n] = base[n];
    dft_small<R, -1>(v, [&](int k, float2 X) { base[k] = X; });
}

// Steps 2 and 3 of the complex FFT, in place on Z.  The step-1 twiddles W_N^(m*k1) are applied
// here on the load side, where the 14 table reads of a butterfly are independent loads issued
// together (inside step 1 each one sat behind a 43-point butterfly).  w2s = step-2 twiddles in LDS.
// SIGN = +1 uses the conjugate twiddles.
template <int SIGN, int NT, bool PK = false>
__device__ __forceinline__ void fft_steps_2_3(float2* Z, const float2* __restrict__ w1, const float2* w2s, int tid) {
#pragma clang fp contract(off)
    // 645 butterflies over NT threads = 3 (2) rounds; the 14 step-1 twiddles of round i + 1 are requested before
    // round i computes (they come from L2: one exposed round trip per round otherwise)
    constexpr int NB = FFT_R1 * FFT_R3, ROUNDS = (NB + NT - 1) / NT;
    float2 w[2][FFT_R2];
    auto load_w = [&](int set, int bf) {
        const int bfc = bf < NB ? bf : NB - 1;
        const int k1 = bfc / FFT_R3, n3 = bfc - k1 * FFT_R3;
        const float2* wb = w1 + k1 * FFT_M1 + n3;
#pragma unroll
        for (int n2 = 0; n2 < FFT_R2; ++n2) w[set][n2] = wb[n2 * FFT_R3];
    };
    load_w(0, tid);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int bf = tid + NT * r;
        if (r + 1 < ROUNDS) load_w((r + 1) & 1, bf + NT);
        if (bf < NB) {
            const int k1 = bf / FFT_R3, n3 = bf - k1 * FFT_R3;
            float2* base = Z + k1 * FFT_M1 + n3;
            float2 v[FFT_R2];
#pragma unroll
            for (int n2 = 0; n2 < FFT_R2; ++n2) v[n2] = base[n2 * FFT_R3];
#pragma unroll
            for (int n2 = 0; n2 < FFT_R2; ++n2) v[n2] = SIGN < 0 ? c_mul(v[n2], w[r & 1][n2]) : c_mulc(v[n2], w[r & 1][n2]);
            dft_small<FFT_R2, SIGN, 0, 1, PK>(v, [&](int k2, float2 X) {
                const float2 t = w2s[k2 * FFT_R3 + n3];
                base[k2 * FFT_R3] = SIGN < 0 ? c_mul(X, t) : c_mulc(X, t);
            });
        }
    }
    __syncthreads();
    for (int bf = tid; bf < FFT_R1 * FFT_R2; bf += NT) {
        float2* base = Z + bf * FFT_R3;
        float2 v[FFT_R3];
#pragma unroll
        for (int n3 = 0; n3 < FFT_R3; ++n3) v[n3] = base[n3];
        dft_small<FFT_R3, SIGN, 0, 1, PK>(v, [&](int k3, float2 X) { base[k3] = X; });
    }
    __syncthreads();
}

// ---- forward: U[row, 0..N] = rfft_L( tw * xpad[(2s-2)h : (2s+2)h] ) --------------------------------
// NT = 256: one lane per 43-point butterfly (210 of 256 lanes), 2 workgroups of 4 waves per CU.
// NT = 512: the OUTPUTS of every 43-point butterfly are split between wave group 0 (lanes 0..255) and wave group 1
//           (lanes 256..511), both holding the column's 43 inputs: the longest serial stage of the transform is
//           halved, steps 2 / 3 take 2 rounds instead of 3, and a CU holds 16 waves instead of 8 -- the kernel is
//           bound by dependent LDS / memory round trips, not by issue slots (27 % VALU utilisation measured).
template <int NT, bool PK = false>
__global__ __launch_bounds__(NT, NT / 128) void k_slice_rfft(const float* __restrict__ x, const float* __restrict__ tw,
                                                              const FftTables T, float2* __restrict__ U,
                                                              int S, int64_t n, int h,
                                                              const int64_t* __restrict__ xrows = nullptr,
                                                              const float* const* __restrict__ xslot = nullptr) {
#pragma clang fp contract(off)          // fused multiply-adds only where written (fmaf): same bits from every instantiation
    __shared__ float2 Z[FFT_N];
    __shared__ float2 w2s[FFT_R2 * FFT_R3];
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const int bc = row / S, s = row - bc * S;
    // packed channel bc starts at x + xrows[bc] (a row of the caller's (nb, 2, N) track: the stacked chunks are read in
    // place, no packing copy) or, without a table, at x + bc * n.  xslot: the base pointer is read from DEVICE memory
    // instead (one uniform 8-byte load) -- a captured HIP graph then follows the caller's tensor from replay to replay
    // without a copy into a static input buffer (Separator.forward_graphed).
    const float* xb = xslot ? *xslot : x;
    const float* xr = xb + (xrows ? xrows[bc] : (int64_t)bc * n);
    const int64_t i0 = (int64_t)(2 * s - 2) * h;
    const int part = tid >> 8, m = tid & 255;            // wave-uniform part
    const float2* tw2 = reinterpret_cast<const float2*>(tw);
    if constexpr (NT == 256) {
        if (m < FFT_M1) {
            float2 v[FFT_R1];
            // Straight-line loads: with a predicate around every sample the compiler emitted 86 load -> wait
            // round trips in sequence (57 us per row, measured); here every load is unconditional and the 129 of a
            // thread are in flight together.  Slices inside the signal (all but the first two and last three of a
            // channel) take 8-byte loads; edge slices clamp the index and select afterwards.
            if (i0 >= 0 && i0 + FFT_L <= n) {            // workgroup-uniform
#pragma unroll
                for (int n1 = 0; n1 < FFT_R1; ++n1) {
                    const int p2 = n1 * FFT_M1 + m;
                    const float2 w = tw2[p2];
                    const float xa = xr[i0 + 2 * p2], xb = xr[i0 + 2 * p2 + 1];
                    v[n1] = make_float2(w.x * xa, w.y * xb);
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < FFT_R1; ++n1) {
                    const int p2 = n1 * FFT_M1 + m;
                    const int64_t i = i0 + 2 * p2;
                    const float2 w = tw2[p2];
                    const int64_t ia = i < 0 ? 0 : (i >= n ? n - 1 : i), ib = i + 1 < 0 ? 0 : (i + 1 >= n ? n - 1 : i + 1);
                    const float xa = xr[ia], xb = xr[ib];
                    v[n1].x = (i >= 0 && i < n) ? w.x * xa : 0.f;
                    v[n1].y = (i + 1 >= 0 && i + 1 < n) ? w.y * xb : 0.f;
                }
            }
            dft_small<FFT_R1, -1, 0, 1, PK>(v, [&](int k1, float2 X) { Z[k1 * FFT_M1 + m] = X; });
        }
    } else {
        // 512 threads: the windowed samples go through LDS once (every lane 18 consecutive-lane pairs, all loads in
        // flight together), then both wave groups read the columns they share from there
        constexpr int NLD = (FFT_N + NT - 1) / NT;
        float2 xv[NLD], wv[NLD];
        const bool inner = i0 >= 0 && i0 + FFT_L <= n;      // workgroup-uniform
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int p2 = tid + NT * it;
            const int pc = p2 < FFT_N ? p2 : FFT_N - 1;
            wv[it] = tw2[pc];
            const int64_t i = i0 + 2 * pc;
            if (inner) { xv[it].x = xr[i]; xv[it].y = xr[i + 1]; }
            else {
                const int64_t ia = i < 0 ? 0 : (i >= n ? n - 1 : i), ib = i + 1 < 0 ? 0 : (i + 1 >= n ? n - 1 : i + 1);
                const float xa = xr[ia], xb = xr[ib];
                xv[it].x = (i >= 0 && i < n) ? xa : 0.f;
                xv[it].y = (i + 1 >= 0 && i + 1 < n) ? xb : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int p2 = tid + NT * it;
            if (p2 < FFT_N) Z[p2] = make_float2(wv[it].x * xv[it].x, wv[it].y * xv[it].y);
        }
        __syncthreads();
        const bool on = m < FFT_M1;
        float2 v[FFT_R1];
        if (on) {
#pragma unroll
            for (int n1 = 0; n1 < FFT_R1; ++n1) v[n1] = Z[n1 * FFT_M1 + m];
        }
        __syncthreads();
        if (on) {
            auto put = [&](int k1, float2 X) { Z[k1 * FFT_M1 + m] = X; };
            if (part == 0) dft_small<FFT_R1, -1, 0, 2, PK>(v, put);
            else dft_small<FFT_R1, -1, 1, 2, PK>(v, put);
        }
    }
    if (tid < FFT_R2 * FFT_R3) w2s[tid] = T.w2[tid];     // (behind the sample loads, not in front of them: see k_slice_irfft)
    __syncthreads();
    fft_steps_2_3<-1, NT, PK>(Z, T.w1, w2s, tid);
    // real post-processing: U[k] = E + G, U[N-k] = conj(E - G), E = (Z[k] + conj Z[N-k])/2,
    // G = -i/2 * W_L^k * (Z[k] - conj Z[N-k])
    float2* Ur = U + (int64_t)row * (FFT_N + 1);
    // the W_L^k of this lane are requested together, ahead of the loop: a table load inside the loop body
    // (waited for in every iteration, and on this target a wait for a load also waits for the stores issued
    // before it) made the loop one memory round trip per iteration
    constexpr int NPP = (FFT_N / 2 + NT) / NT;             // iterations that cover k = 0 .. N/2
    float2 wlr[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) { const int k = tid + NT * i; wlr[i] = T.wl[k <= FFT_N / 2 ? k : FFT_N / 2]; }
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int k = tid + NT * i;
        if (k > FFT_N / 2) break;
        const float2 zk = Z[fft_pos(k)];
        const float2 zn = Z[fft_pos(k == 0 ? 0 : FFT_N - k)];
        const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 D = make_float2(0.5f * (zk.x - zn.x), 0.5f * (zk.y + zn.y));
        const float2 td = c_mul(wlr[i], D);
        const float2 G = make_float2(td.y, -td.x);   // -i * td
        Ur[k] = c_add(E, G);
        Ur[FFT_N - k] = make_float2(E.x - G.x, -(E.y - G.y));
    }
}

// ---- inverse: seg = L * irfft_L( sum over covering bands of the synthesis spectra ), overlap-added -----
// Overlap-add fused in (nsgt/unslicing.py:6-69, no synthesis window): slice s holds samples (2s-2)h .. (2s+2)h,
// every output sample is the sum of exactly two neighbouring slices -- one even, one odd.  The launch with
// parity 0 transforms the even slices and STORES them (together they cover every sample once); the launch
// with parity 1 transforms the odd slices and ADDS them (load, add, store: each sample belongs to one odd
// slice, so no atomics and a fixed order: even + odd, bitwise the same sum as any other order of two terms).
// The L-sample segments never go to HBM.  Samples past the last even slice (S even, tail of the last odd
// slice) have no partner and are stored by the odd launch.  row_off: element offset of packed channel bc in y
// (the caller's final tensor: the hard concat of separator.py:231 by placement), nullptr = bc * length.
#if XSQ_FFT_STAMP
// phase stamps of the inverse transform, diagnostic builds only: s_memrealtime (100 MHz, the same clock on every CU) of
// thread 0 at: 0 start, 1 band spectra gathered, 2 pre-processed, 3 43-point stage done, 4 steps 2 / 3 done, 5 stores issued
constexpr int FFT_STAMP_ROWS = 16384;
__device__ unsigned long long g_fft_stamps[FFT_STAMP_ROWS * 8];
#define XSQ_STAMP(i) do { if (tid == 0 && blockIdx.x < FFT_STAMP_ROWS) g_fft_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSQ_STAMP(i) do { } while (0)
#endif

struct OlaArgs {
    float* y;
    const int64_t* row_off;
    int S, h, parity;
    int64_t length;
};


// SHORT: the short bands are synthesised inside this kernel (ShortSched; an A/B switch that measured slower) -- a template
// parameter, so that the default instantiation carries neither that code nor its registers (as a run-time branch it
// pushed the gather prologue of the default path over the 128-register cap: spilled loads, each waited for in turn).
// FAST4: the four-phases-in-flight gather (see below); chosen by the host when the plan's phases fit.
template <int NT, bool PK = false, bool SHORT = false, bool FAST4 = false>
__global__ __launch_bounds__(NT, NT / 128) void k_slice_irfft(const float2* __restrict__ Zrow, const GatherSched G,
                                                      const FftTables T, const OlaArgs O, const ShortSched SS,
                                                      const ShortIn SI) {
#pragma clang fp contract(off)          // fused multiply-adds only where written (fmaf): same bits from every instantiation
    __shared__ float2 Z[FFT_N + 1];        // bins 0..N while gathering, then the complex sequence
    __shared__ float2 w2s[FFT_R2 * FFT_R3];
    const int tid = threadIdx.x;
#if XSQ_FFT_DEPHASE > 0
    // Every row alternates HBM phases (band-spectrum gather, output) with compute phases of fixed lengths, and a launch
    // is only ~9 rows deep per workgroup slot: all slots start together and STAY together -- the whole chip gathers,
    // then the whole chip computes with HBM idle.  Half of the first wave of workgroups starts late by about half a
    // row, once; the slots then run in two groups whose HBM phases face the other group's compute phases.
    if (blockIdx.x < XSQ_FFT_DEPHASE_SLOTS && (blockIdx.x & 1))
        for (int i = 0; i < XSQ_FFT_DEPHASE; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    XSQ_STAMP(0);
    const int nsl = (O.S + 1 - O.parity) >> 1;              // slices of this parity per channel
    const int bc = blockIdx.x / nsl, s = 2 * (blockIdx.x - bc * nsl) + O.parity;
    const int row = bc * O.S + s;
    // (the step-2 twiddles go into LDS AFTER the first gather loads have been issued: in front of them, the table load
    //  and the wait for it -- the LDS store needs the value -- cost every row one exposed L2 round trip before its first
    //  request left the CU; in-kernel stamps, tools/fft_phases.py)
    // Gather registers of the long bands: two phases per memory round trip, all loads of a pair of phases issued
    // (unconditionally: clamped entry index, validity kept in k) before any is consumed.
    constexpr int UN = (4864 + NT - 1) / NT;     // covers a whole phase of the Bark-262 plan in one chunk
    const float2* zr = Zrow + (int64_t)row * G.row_len;
    float2 z[2][UN];
    int k[2][UN];
    auto gather_load1 = [&](int h, int ph, int off) {       // one phase into register set h (compile-time h)
        const int e0 = G.lo[ph], e1 = G.begin[ph + 1];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int ee = e0 + off + NT * u;
            const int ec = ee < e1 ? ee : (e1 > e0 ? e1 - 1 : 0);
            const int kk = G.tgt[ec];
            k[h][u] = ee < e1 ? kk : -1;
            z[h][u] = zr[ec];
        }
    };
    auto gather_accum1 = [&](int h) {
        // bins of one phase are distinct: read all, add, write all (no read-after-write chain)
        float2 acc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc[u] = Z[k[h][u] >= 0 ? k[h][u] : 0];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (k[h][u] >= 0) Z[k[h][u]] = make_float2(acc[u].x + z[h][u].x, acc[u].y + z[h][u].y);
        __syncthreads();
    };
    // Fast path (every phase of the plan fits NT * 2 * UP entries -- the Bark-262 plan -- and SHORT is off): the row's
    // first THREE phases are requested at once and the fourth as soon as the first has been added, two entries per lane
    // and load (16-byte loads of the spectra, one dword of two 16-bit target bins): 112 KB in flight per row instead of
    // 75, 40 loads per lane instead of 80 in two dependent batches.  In-kernel stamps (tools/fft_phases.py) had the gather at half of the row's 31 us, waiting: two exposed
    // round trips at ~10 GB/s per workgroup, the second one behind the first pair's accumulation.  The accumulation
    // order of a bin is unchanged (phase 0, 1, 2, 3): same bits.
    constexpr int UP = (4864 / 2 + 1 + NT - 1) / NT;        // entry pairs per lane and phase (5 at 512 threads)
    float4 zz[3][UP];       // three register sets: phases 0, 1, 2 are requested at the start, phase 3 into set 0 once
    unsigned kk[3][UP];     // phase 0 has been added (all four at once spill under the 128-register cap)
    static_assert(!FAST4 || !SHORT, "the four-phase gather is the default path's");
    constexpr bool fast4 = FAST4;
    const bool do_gather = !(XSQ_ABLATE & 16);
    auto load_phase4 = [&](int slot, int ph) {       // slot, ph: compile-time at every call site
        const int lo = G.lo[ph], e0 = lo & ~1, e1 = G.begin[ph + 1];       // pairs are aligned to even entries of the row
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int ee = e0 + 2 * (tid + NT * u);
            const int ec = ee < e1 ? ee : e0;
            unsigned t2 = *reinterpret_cast<const unsigned*>(G.tgt16 + ec);
            if (!(ee >= lo && ee < e1)) t2 |= 0xFFFFu;                      // first entry of the pair outside the phase
            if (!(ee + 1 >= lo && ee + 1 < e1)) t2 |= 0xFFFF0000u;           // second one
            kk[slot][u] = t2;
            zz[slot][u] = *reinterpret_cast<const float4*>(zr + ec);
        }
    };
    if constexpr (FAST4) if (do_gather) { load_phase4(0, 0); load_phase4(1, 1); load_phase4(2, 2); }
    if constexpr (SHORT) {
        // ---- short bands in-kernel (see ShortSched) ------------------------------------------------------
        constexpr int U1 = (1280 + NT - 1) / NT, U2 = (768 + NT - 1) / NT, UG = (1280 + NT - 1) / NT;   // 1280 butterflies, 768 small DFTs, 1280 entries per phase
        float2* const scr = Z + SS.sc0;
        const bool masked = SI.mask != nullptr;
        const int64_t BCS = (int64_t)SI.BC * O.S, BCSx = (int64_t)SI.BCx * O.S;
        const int bcx = masked ? bc % SI.BCx : bc;
        const float2* const c2 = reinterpret_cast<const float2*>(SI.coef);
        float2 xv[U1][4], tw[U1][3];
        float mk[U1][4];
        int sc1[U1], m1[U1];
#pragma unroll
        for (int u = 0; u < U1; ++u) {               // all loads first (unconditional, clamped item index)
            const int i = tid + NT * u;
            const int ii = i < SS.n1 ? i : SS.n1 - 1;
            const ShortItem1 it = SS.item1[ii];
            const int m = it.Lg >> 2;
            sc1[u] = i < SS.n1 ? it.sc + it.t1 : -1;
            m1[u] = m;
            const int64_t xi = (masked ? BCSx : BCS) * it.cum + (((int64_t)bcx * it.F + it.f) * O.S + s) * it.Lg + it.t1;
            const int64_t mi = BCS * it.cum + (((int64_t)bc * it.F + it.f) * O.S + s) * it.Lg + it.t1;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xv[u][a] = c2[xi + a * m];
                mk[u][a] = masked ? SI.mask[mi + a * m] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) tw[u][r] = SS.tw1[3 * ii + r];
        }
        // the long bands' first pair of phases travels in the same memory round trip as the short bands' inputs
        if (do_gather) { gather_load1(0, 0, tid); gather_load1(1, 1, tid); }
        // spectrum slots outside the scratch area start at zero (the scratch area is written in full below)
        for (int k = tid; k <= FFT_N; k += NT)
            if (k < SS.sc0 || k >= SS.sc0 + SS.nent) Z[k] = make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U1; ++u) {
            if (sc1[u] < 0) continue;
            float2 x[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) x[a] = masked ? make_float2(xv[u][a].x * mk[u][a], xv[u][a].y * mk[u][a]) : xv[u][a];
            const float2 s0 = c_add(x[0], x[2]), s1 = c_add(x[1], x[3]);
            const float2 d0 = c_sub(x[0], x[2]), d1 = c_sub(x[1], x[3]);
            const float2 y1 = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - i d1
            const float2 y3 = make_float2(d0.x - d1.y, d0.y + d1.x);      // d0 + i d1
            float2* o = scr + sc1[u];
            o[0] = c_add(s0, s1);
            o[m1[u]] = c_mul(y1, tw[u][0]);
            o[2 * m1[u]] = c_mul(c_sub(s0, s1), tw[u][1]);
            o[3 * m1[u]] = c_mul(y3, tw[u][2]);
        }
        // gather tables of the four phases, requested before the small DFTs run
        int gk[4][UG];
        float gw[4][UG];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int e = SS.begin[ph] + tid + NT * u;
                const bool ok = e < SS.begin[ph + 1];
                gk[ph][u] = ok ? SS.stgt[e] : -1;
                gw[ph][u] = ok ? SS.swd[e] : 0.f;
            }
        int code2[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = tid + NT * u; code2[u] = i < SS.n2 ? SS.item2[i] : 0; }
        __syncthreads();
#pragma unroll 1                                   // one copy of the twelve codelets (instruction cache)
        for (int u = 0; u < U2; ++u) {
            float2* base = scr + (code2[u] >> 4);
            switch (code2[u] & 15) {               // items are sorted by m: a wave sees one or two cases
                case 4: short_dft<4>(base); break;
                case 5: short_dft<5>(base); break;
                case 6: short_dft<6>(base); break;
                case 7: short_dft<7>(base); break;
                case 8: short_dft<8>(base); break;
                case 9: short_dft<9>(base); break;
                case 10: short_dft<10>(base); break;
                case 11: short_dft<11>(base); break;
                case 12: short_dft<12>(base); break;
                case 13: short_dft<13>(base); break;
                case 14: short_dft<14>(base); break;
                case 15: short_dft<15>(base); break;
                default: break;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int e = SS.begin[ph] + tid + NT * u;
                if (gk[ph][u] >= 0) {
                    const float2 v = scr[e];
                    float2 acc = Z[gk[ph][u]];
                    acc.x += v.x * gw[ph][u];
                    acc.y += v.y * gw[ph][u];
                    Z[gk[ph][u]] = acc;
                }
            }
            __syncthreads();
        }
        for (int e = tid; e < SS.nent; e += NT) scr[e] = make_float2(0.f, 0.f);
    } else {
        if (do_gather && !fast4) { gather_load1(0, 0, tid); gather_load1(1, 1, tid); }
        for (int k = tid; k <= FFT_N; k += NT) Z[k] = make_float2(0.f, 0.f);
    }
    if (tid < FFT_R2 * FFT_R3) w2s[tid] = T.w2[tid];
    __syncthreads();
    // gather-sum of the band spectra, one phase of mutually disjoint bands at a time; the accumulation stays
    // phase by phase with a barrier in between, because neighbouring phases overlap in bins.  Every phase fits one
    // chunk in the Bark-262 plan: phase p + 2 is requested as soon as phase p has been added and its registers are
    // free, so its round trip runs beside the accumulation of phase p + 1 instead of behind it.
    if constexpr (FAST4) { if (do_gather) {
#if XSQ_FFT_STAMP
        if (zz[0][0].x == 1.2345e-30f) __builtin_trap();
        XSQ_STAMP(6);
#endif
        auto accum_phase4 = [&](int slot) {
            // bins of one phase are distinct: read all, add, write all (no read-after-write chain)
            float2 a0[UP], a1[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const unsigned k0 = kk[slot][u] & 0xFFFFu, k1 = kk[slot][u] >> 16;
                a0[u] = Z[k0 != 0xFFFFu ? k0 : 0];
                a1[u] = Z[k1 != 0xFFFFu ? k1 : 0];
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const unsigned k0 = kk[slot][u] & 0xFFFFu, k1 = kk[slot][u] >> 16;
                if (k0 != 0xFFFFu) Z[k0] = make_float2(a0[u].x + zz[slot][u].x, a0[u].y + zz[slot][u].y);
                if (k1 != 0xFFFFu) Z[k1] = make_float2(a1[u].x + zz[slot][u].z, a1[u].y + zz[slot][u].w);
            }
            __syncthreads();
        };
        accum_phase4(0);
        load_phase4(0, 3);
        accum_phase4(1);
#if XSQ_FFT_STAMP
        XSQ_STAMP(7);
#endif
        accum_phase4(2);
        accum_phase4(0);
    } } else if (do_gather) {      // other band layouts: chunk by chunk (the first chunks of phases 0 / 1 are in flight)
        for (int ph = 0; ph < 4; ++ph) {
            const int span = G.begin[ph + 1] - G.lo[ph];
            for (int off = 0; off < span; off += NT * UN) {
                if (ph == 1 && off == 0) { gather_accum1(1); continue; }
                if (ph != 0 || off > 0) gather_load1(0, ph, off + tid);
                gather_accum1(0);
            }
        }
    }
    XSQ_STAMP(1);
    // real pre-processing in place: Zin[k] = E + i O, Zin[N-k] = conj(E) + i conj(O),
    // E = U[k] + conj U[N-k], O = (U[k] - conj U[N-k]) * conj(W_L^k)   (no 1/2: output = L * irfft)
    constexpr int NPP = (FFT_N / 2 + NT) / NT;             // table values requested together (see k_slice_rfft)
    float2 wlr[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) { const int k = tid + NT * i; wlr[i] = T.wl[k <= FFT_N / 2 ? k : FFT_N / 2]; }
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int k = tid + NT * i;
        if (k > FFT_N / 2) break;
        float2 uk = Z[k], un = Z[FFT_N - k];
        if (k == 0) { uk.y = 0.f; un.y = 0.f; }   // irfft ignores Im of DC / Nyquist (nsigtf.py:103)
        const float2 E = make_float2(uk.x + un.x, uk.y - un.y);
        const float2 D = make_float2(uk.x - un.x, uk.y + un.y);
        const float2 O = c_mulc(D, wlr[i]);
        Z[k] = make_float2(E.x - O.y, E.y + O.x);
        if (k != 0) Z[FFT_N - k] = make_float2(E.x + O.y, O.x - E.y);
    }
    __syncthreads();
    XSQ_STAMP(2);
    {   // 43-point butterflies, in place: every lane reads its column first; with NT = 512 the outputs are split between
        // the two wave groups, which both hold the column (barrier between the reads and the writes of the pair)
        const int part = tid >> 8, m = tid & 255;
        const bool on = m < FFT_M1 && !(XSQ_ABLATE & 32);
        float2 v[FFT_R1];
        if (on) {
#pragma unroll
            for (int n1 = 0; n1 < FFT_R1; ++n1) v[n1] = Z[n1 * FFT_M1 + m];
        }
        if constexpr (NT != 256) __syncthreads();
        if (on) {
            auto put = [&](int k1, float2 X) { Z[k1 * FFT_M1 + m] = X; };
            if constexpr (NT == 256) dft_small<FFT_R1, +1, 0, 1, PK>(v, put);
            else if (part == 0) dft_small<FFT_R1, +1, 0, 2, PK>(v, put);
            else dft_small<FFT_R1, +1, 1, 2, PK>(v, put);
        }
    }
    __syncthreads();
    XSQ_STAMP(3);
    // samples 2nn, 2nn+1 of the segment = Re / Im of sequence element nn; output index i = (2s-2)h + 2nn
    float* const yr = O.y + (O.row_off ? O.row_off[bc] : (int64_t)bc * O.length);
    const int64_t i0 = (int64_t)(2 * s - 2) * O.h;
    // the second half of the last slice has no partner: plain store even in the adding launch
    const bool add_lo = O.parity != 0, add_hi = O.parity != 0 && s + 1 < O.S;
    const bool al8 = ((reinterpret_cast<uintptr_t>(yr + i0)) & 7) == 0;       // workgroup-uniform (h may be odd: i0 too)
    constexpr int NIT = (FFT_N + NT - 1) / NT;
    const bool interior = al8 && i0 >= 0 && i0 + FFT_L <= O.length;
    float2* const y2 = reinterpret_cast<float2*>(yr + i0);
    if (!(XSQ_ABLATE & 64)) fft_steps_2_3<+1, NT, PK>(Z, T.w1, w2s, tid);
    XSQ_STAMP(4);
    if (XSQ_ABLATE & 128) return;
    if (interior) {
        // interior slice, 8-byte aligned: all loads of the partner sums first, then all stores
        float2 prev[NIT];
        if (O.parity) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int nn = tid + NT * it;
                prev[it] = (nn < FFT_N && (2 * nn < 2 * O.h ? add_lo : add_hi)) ? y2[nn] : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int nn = tid + NT * it;
            if (nn >= FFT_N) break;
            float2 v = Z[fft_pos(nn)];
            if (O.parity) { v.x += prev[it].x; v.y += prev[it].y; }
            y2[nn] = v;
        }
        XSQ_STAMP(5);
        return;
    }
    // edge slices (first / last of a channel) and odd row offsets: per-sample bounds, 4-byte accesses
    for (int nn = tid; nn < FFT_N; nn += NT) {
        const float2 v = Z[fft_pos(nn)];
        const bool add = 2 * nn < 2 * O.h ? add_lo : add_hi;     // 2h is even: both samples of nn lie in the same half
        const int64_t ia = i0 + 2 * nn, ib = ia + 1;
        if (ia >= 0 && ia < O.length) yr[ia] = add ? yr[ia] + v.x : v.x;
        if (ib >= 0 && ib < O.length) yr[ib] = add ? yr[ib] + v.y : v.y;
    }
}

}  // namespace xsq
