// Forward / inverse sliced Constant-Q transform for gfx950.
//
// Forward  (closed form, SURVEY.md 8(a) F*; reference nsgt/slicing.py, nsgt/nsgtf.py, nsgt/slicq.py:13-33)
//   seg[bc,s,p]   = tw[p] * xpad[bc, (2s-2)h + p]                         k_slice_window   (HBM stream)
//   U[bc,s,:]     = rfft_L(seg[bc,s,:])                                   rocFFT
//   coef[bc,j,s,:] = U[bc,s, bin0_j : bin0_j+Lg_j] (Hermitian-reflected at DC/Nyquist) x Wf_j
//                                                                         grouped fp32-MFMA GEMM
//   Wf_j = diag(g_j * (-1)^(c_j/2) / Lg_j) * IDFT_Lg, real-ified on interleaved (re,im).
// Inverse  (closed form I2; reference nsgt/nsigtf.py, nsgt/unslicing.py)
//   Z[bc,j,s,:]   = coef[bc,j,s,:] x Wi_j,  Wi_j = DFT_Lg * diag(gd_j * Lg_j * (-1)^(c_j/2) / L)
//   fr[bc,s,k]    = sum over bands covering bin k of Z[bc,j,s,k-bin0_j]   k_spectrum_gather (HBM stream)
//   seg[bc,s,:]   = L * irfft_L(fr[bc,s,:])  (unnormalised c2r; the 1/L sits in Wi)   rocFFT
//   y[bc,i]       = seg[bc,s0,i-(2s0-2)h] + seg[bc,s0+1,i-2h*s0],  s0 = i/(2h)        k_overlap_add
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/xumx_slicq_hip.h"
#include "gemm_tile.h"
#include "plan.h"
#include "prof.h"
#include "slice_fft.h"
#include "band_dft4.h"
#include "band_dft4s.h"

namespace xsq {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
// streaming kernels
// ------------------------------------------------------------------------------------------
// grid (ceil(L/256), BC*S).  Reads x once (each sample lands in two slices), writes seg once.
__global__ __launch_bounds__(256) void k_slice_window(const float* __restrict__ x,
                                                       const float* __restrict__ tw,
                                                       float* __restrict__ seg, int S, int64_t n, int L,
                                                       int h, const int64_t* __restrict__ xrows = nullptr,
                                                       const float* const* __restrict__ xslot = nullptr) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= L) return;
    const int row = blockIdx.y;  // bc*S + s
    const int bc = row / S, s = row - bc * S;
    const int64_t i = (int64_t)(2 * s - 2) * h + p;
    float v = 0.f;
    if (i >= 0 && i < n) v = tw[p] * (xslot ? *xslot : x)[(xrows ? xrows[bc] : (int64_t)bc * n) + i];
    seg[(int64_t)row * L + p] = v;
}

// grid (ceil(nbins/256), BC*S).  fr[row][k] = sum of the covering bands' synthesis outputs.
__global__ __launch_bounds__(256) void k_spectrum_gather(const float* __restrict__ Z,
                                                          const BandDev* __restrict__ bands,
                                                          const int* __restrict__ cov_ptr,
                                                          const int* __restrict__ cov_band,
                                                          float2* __restrict__ fr, int BC, int S, int nbins) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nbins) return;
    const int row = blockIdx.y;
    const int bc = row / S, s = row - bc * S;
    const int64_t BCS = (int64_t)BC * S;
    float2 acc = make_float2(0.f, 0.f);
    for (int e = cov_ptr[k]; e < cov_ptr[k + 1]; ++e) {
        const BandDev b = bands[cov_band[e]];
        const int64_t off = 2 * (BCS * b.cum + (((int64_t)bc * b.F + b.f) * S + s) * b.Lg + (k - b.bin0));
        const float2 z = *reinterpret_cast<const float2*>(Z + off);
        acc.x += z.x;
        acc.y += z.y;
    }
    // irfft ignores the imaginary part of the DC and Nyquist bins (torch.fft.irfft, nsigtf.py:103)
    if (k == 0 || k == nbins - 1) acc.y = 0.f;
    fr[(int64_t)row * nbins + k] = acc;
}

// grid (ceil(length/256), BC).  Each output sample is the sum of exactly two slices.
__global__ __launch_bounds__(256) void k_overlap_add(const float* __restrict__ seg, float* __restrict__ y,
                                                      const int64_t* __restrict__ row_off,
                                                      int S, int64_t length, int L, int h) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= length) return;
    const int bc = blockIdx.y;
    const int s0 = (int)(i / (2 * h));
    const int p1 = (int)(i - (int64_t)2 * h * s0);  // position inside slice s0+1
    const float* base = seg + (int64_t)bc * S * L;
    float v = base[(int64_t)s0 * L + p1 + 2 * h];
    if (s0 + 1 < S) v += base[(int64_t)(s0 + 1) * L + p1];
    y[(row_off ? row_off[bc] : (int64_t)bc * length) + i] = v;
}

// ------------------------------------------------------------------------------------------
// grouped GEMM operators
// ------------------------------------------------------------------------------------------
struct BandGroup {
    int M, N, K, ldb;
    const float* B;
    int Lg, bin0, f, F;
    int64_t base;  // float offset of (bc=0, f, s=0) of this band in the arena
    int64_t cum;   // complex coefficients per channel-slice before this band's block
    int j;         // band index inside the plan
};

// coef = U_window x Wf
struct BandFwdOp {
    typedef BandGroup Group;
    struct RowA {
        const float* p;  // U row (bc,s), nullptr when past M
    };
    const float* U;
    float* coef;
    const BandDev* bands;
    const float* W;
    int BC, S, nbins, L;
    float* xin;            // optional whitened magnitude beside the coefficients (see Band4Args)
    const float* mean;
    const float* scale;
    int split;

    __device__ Group group(int j) const {
        const BandDev b = bands[j];
        Group g;
        g.M = BC * S; g.N = 2 * b.Lg; g.K = 2 * b.Lg; g.ldb = b.ldw; g.B = W + b.w_off;
        g.Lg = b.Lg; g.bin0 = b.bin0; g.f = b.f; g.F = b.F;
        g.base = 2 * ((int64_t)BC * S * b.cum + (int64_t)b.f * S * b.Lg);
        g.cum = b.cum;
        g.j = j;
        return g;
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r;
        r.p = m < g.M ? U + (int64_t)m * 2 * nbins : nullptr;
        return r;
    }
    // two complex bins per call; bins outside [0, L/2] come from the Hermitian mirror (their
    // conjugation is folded into the signs of Wf's imaginary rows)
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        const bool ok = r.p != nullptr && k < g.K;                       // out of range: both loads read zero16()
        int i0 = g.bin0 + (k >> 1), i1 = i0 + 1;
        i0 = i0 < 0 ? -i0 : (i0 > L / 2 ? L - i0 : i0);
        i1 = i1 < 0 ? -i1 : (i1 > L / 2 ? L - i1 : i1);
        const float2 a = *reinterpret_cast<const float2*>(ok ? r.p + 2 * i0 : zero16());
        const float2 b = *reinterpret_cast<const float2*>(ok ? r.p + 2 * i1 : zero16());
        return make_float4(a.x, a.y, b.x, b.y);
    }
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool wide) const {
        const bool c0 = n < g.N, c1 = wide && n + 32 < g.N;
        int bc = row0 / S, s = row0 - bc * S;          // one division per lane, then carry
        int prev = 0;
        // loaded BEFORE the first store: a load between two stores makes the compiler wait for the store as well
        const float mu = xin ? mean[g.j] : 0.f, sc = xin ? scale[g.j] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s += acc_row(r) - prev; prev = acc_row(r);
            while (s >= S) { s -= S; ++bc; }
            if (row0 + acc_row(r) >= g.M) break;
            float* d = coef + g.base + ((int64_t)bc * g.F * S + s) * (2 * g.Lg) + n;
            if (c0) d[0] = a0[r];
            if (c1) d[32] = a1[r];
            if (xin) {       // uniform.  Column n = 2t + (re/im): the even lane of a pair forms |c| and stores it
                const float o0 = __shfl_xor(a0[r], 1), o1 = __shfl_xor(a1[r], 1);
                if (!(n & 1)) {
                    float* x = xin + ((d - coef) >> 1);
                    if (c0) { float v = whiten_mag(a0[r], o0, mu, sc); if (split) v = bf3_word(v); x[0] = v; }
                    if (c1) { float v = whiten_mag(a1[r], o1, mu, sc); if (split) v = bf3_word(v); x[16] = v; }
                }
            }
        }
    }
};

// Z = coef x Wi   (dense rows in, dense rows out, same arena layout)
struct BandInvOp {
    typedef BandGroup Group;
    struct RowA {
        const float* p;
        const float* mk;
    };
    const float* coef;
    float* Z;
    const BandDev* bands;
    const float* W;
    int BC, S;
    int row_len;   // > 0: write row-major, phase-ordered (rows of row_len complex entries) for k_slice_irfft
    const float* mask;   // optional: coefficients = mask * mix (see Band4Args); coef is then the mix arena, BCx channels
    int BCx;

    __device__ Group group(int j) const {
        const BandDev b = bands[j];
        Group g;
        g.M = BC * S; g.N = 2 * b.Lg; g.K = 2 * b.Lg; g.ldb = b.ldw; g.B = W + b.w_off;
        g.Lg = b.Lg; g.bin0 = b.ent; g.f = b.f; g.F = b.F;   // bin0 slot carries the entry offset here
        g.base = 2 * ((int64_t)BC * S * b.cum + (int64_t)b.f * S * b.Lg);
        g.cum = b.cum;
        return g;
    }
    __device__ int64_t row_off(const Group& g, int m) const {
        const int bc = m / S, s = m - bc * S;
        return g.base + ((int64_t)bc * g.F * S + s) * (2 * g.Lg);
    }
    __device__ RowA row_a(const Group& g, int m) const {
        RowA r;
        r.p = nullptr; r.mk = nullptr;
        if (m >= g.M) return r;
        if (!mask) { r.p = coef + row_off(g, m); return r; }
        const int bc = m / S, s = m - bc * S;
        r.mk = mask + row_off(g, m) / 2;
        r.p = coef + 2 * ((int64_t)BCx * S * g.cum + (int64_t)g.f * S * g.Lg) + ((int64_t)(bc % BCx) * g.F * S + s) * (2 * g.Lg);
        return r;
    }
    __device__ float4 load_a4(const Group& g, const RowA& r, int k) const {
        const bool ok = r.p != nullptr && k < g.K;
        return *reinterpret_cast<const float4*>(ok ? r.p + k : zero16());  // rows are 32-byte aligned (Lg % 4 == 0)
    }
    // masked synthesis: the two masks of the K-chunk travel beside the coefficients and are applied when the
    // K-step goes into LDS (gemm_tile.h, aux hook)
    typedef float2 Aux;
    __device__ Aux load_aux(const Group& g, const RowA& r, int k) const {
        const bool ok = r.mk != nullptr && k < g.K;
        return *reinterpret_cast<const float2*>(ok ? r.mk + (k >> 1) : zero16());
    }
    __device__ float4 finish(float4 v, const Aux& mk) const {
        if (mask) { v.x *= mk.x; v.y *= mk.x; v.z *= mk.y; v.w *= mk.y; }      // uniform
        return v;
    }
    __device__ void epilogue(const Group& g, int row0, int n, const f32x16& a0, const f32x16& a1, bool wide) const {
        const bool c0 = n < g.N, c1 = wide && n + 32 < g.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + acc_row(r);
            if (m >= g.M) break;
            float* d = (row_len ? Z + 2 * ((int64_t)m * row_len + g.bin0) : Z + row_off(g, m)) + n;
            if (c0) d[0] = a0[r];
            if (c1) d[32] = a1[r];
        }
    }
};

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int upload_tiles(const std::vector<TileDev>& t, TileTable* tt) {
    tt->ntiles = (int)t.size();
    tt->d_tiles = nullptr;
    if (t.empty()) return XSQ_OK;
    XSQ_HIP(hipMalloc(&tt->d_tiles, t.size() * sizeof(TileDev)));
    XSQ_HIP(hipMemcpy(tt->d_tiles, t.data(), t.size() * sizeof(TileDev), hipMemcpyHostToDevice));
    return XSQ_OK;
}

// dense-GEMM tiles: all bands, or only the short ones when the radix-4 kernel takes the rest
static int get_band_tiles(xsq_plan* P, int rows, TileTable* out) {
    std::lock_guard<std::mutex> lk(P->mu);
    const bool r4 = P->band_radix4 && P->nbands4 > 0;
    auto key = std::make_tuple(r4 ? 2 : 0, rows, 0);
    auto it = P->tiles.find(key);
    if (it != P->tiles.end()) { *out = it->second; return XSQ_OK; }
    std::vector<TileDev> t;
    // longest tiles first (K = 2*Lg grows along the band table): the launch ends on short tiles
    if (r4) {
        for (int i = (int)P->bands4_small.size() - 1; i >= 0; --i)
            push_group_tiles(t, P->bands4_small[i], rows, 2 * P->bands[P->bands4_small[i]].Lg);
    } else {
        for (int j = P->nbands - 1; j >= 0; --j) push_group_tiles(t, j, rows, 2 * P->bands[j].Lg);
    }
    TileTable tt;
    int rc = upload_tiles(t, &tt);
    if (rc) return rc;
    P->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// tiles of the radix-4 kernel (band_dft4_full_kernel): 32 rows x every column of the band; TileDev.narrow = number of
// 16-column blocks.
// `share` > 0 (masked synthesis): rows r and r + share, r + 2*share, ... read the same mix rows (the targets of one
// (sample, channel, slice)); their tiles are made neighbours so the mix is fetched once per XCD.  When share is not a
// multiple of the tile height the last tile of a copy runs into the next copy's first rows and recomputes them
// (same values, written twice).
// cls: 0 = every band (one launch of the 10-block kernel), 1 = bands of more than 5 blocks, 2 = bands of at most 5 blocks
// (their own instantiation: four workgroups per CU, band_dft4.h)
static int get_dft4_full_tiles(xsq_plan* P, int rows, TileTable* out, int share = 0, int cls = 0, bool sym = false) {
    std::lock_guard<std::mutex> lk(P->mu);
    auto key = std::make_tuple(3 + 16 * cls + (sym ? 256 : 0), rows, share);
    auto it = P->tiles.find(key);
    if (it != P->tiles.end()) { *out = it->second; return XSQ_OK; }
    std::vector<Tile4Dev> t;
    const Band4Dev* b4 = reinterpret_cast<const Band4Dev*>(P->bands4_host.data());
    const int span = share > 0 ? share : rows, copies = share > 0 ? rows / share : 1;
    for (int i = P->nbands4 - 1; i >= 0; --i) {
        int ncb = (2 * P->bands4_m[i] + 15) / 16;
        if ((cls == 1 && ncb <= 5) || (cls == 2 && ncb > 5)) continue;
        if (sym) ncb = (P->bands4_m[i] / 2 + 1 + 15) / 16;        // band_dft4s.h: blocks of the outputs k = 0 .. m / 2
        for (int m0 = 0; m0 < span; m0 += D4H_ROWS)
            for (int k = 0; k < copies; ++k) t.push_back(Tile4Dev{m0 + k * span, ncb, b4[i]});
    }
    TileTable tt;                     // (d_tiles holds Tile4Dev entries for this key: cast at the launch sites)
    tt.ntiles = (int)t.size();
    if (!t.empty()) {
        XSQ_HIP(hipMalloc(&tt.d_tiles, t.size() * sizeof(Tile4Dev)));
        XSQ_HIP(hipMemcpy(tt.d_tiles, t.data(), t.size() * sizeof(Tile4Dev), hipMemcpyHostToDevice));
    }
    P->tiles[key] = tt;
    *out = tt;
    return XSQ_OK;
}

// The radix-4 band kernel reaches operands and results through buffer descriptors per band block / tile with 32-bit byte
// offsets and switches lanes off with offsets of 2^30 and 2^31 (band_dft4.h, common.h): every such range stays below 2^30
// bytes.  Checked where a transform call enters, before anything is launched.
static int d4_ranges_ok(const xsq_plan* P, int64_t rows, const char* who) {
    if (!(P->band_radix4 && P->nbands4)) return XSQ_OK;
    XSQ_REQUIRE(8 * rows * P->d4_max_block < (1ll << 30) && (int64_t)8 * D4H_ROWS * std::max<int64_t>(P->nbins, P->sumFT) < (1ll << 30),
                "%s: %lld rows x %lld coefficients of a band block exceed 2^30 bytes; split the call (fewer stacked chunks)",
                who, (long long)rows, (long long)P->d4_max_block);
    return XSQ_OK;
}

// the radix-4 band kernel over every eligible band, one launch of the 10-block instantiation (three workgroups per CU).
// XSQ_D4_SPLIT=1 (A/B switch, measured no faster: synthesis 0.940-0.948 vs 0.916-0.945 ms, analysis 0.309-0.316 vs
// 0.291-0.301): the bands of at most five 16-column blocks in their own instantiation -- 35 KB of LDS, 96 registers,
// four workgroups per CU -- launched behind the wide ones.
template <bool FWD>
static int launch_dft4(xsq_plan* P, const Band4Args& a4, int rows, int share, hipStream_t stream) {
    static const bool split = getenv("XSQ_D4_SPLIT") && atoi(getenv("XSQ_D4_SPLIT")) == 1;
    // the pair-contracted form (band_dft4s.h) is the default; XSQ_D4_SYM=0 runs the complex-product form (band_dft4.h)
    static const bool sym = !(getenv("XSQ_D4_SYM") && atoi(getenv("XSQ_D4_SYM")) == 0);
    const bool masked = !FWD && a4.mask != nullptr;
    TileTable t;
    int rc;
    if (sym) {
        if ((rc = get_dft4_full_tiles(P, rows, &t, share, 0, true))) return rc;
        if (!t.ntiles) return XSQ_OK;
        if constexpr (!FWD) {
            if (masked) { hipLaunchKernelGGL((band_dft4s_kernel<false, true>), dim3(t.ntiles), dim3(256), 0, stream, a4, (const Tile4Dev*)t.d_tiles, t.ntiles); return XSQ_OK; }
        }
        hipLaunchKernelGGL((band_dft4s_kernel<FWD, false>), dim3(t.ntiles), dim3(256), 0, stream, a4, (const Tile4Dev*)t.d_tiles, t.ntiles);
        return XSQ_OK;
    }
    auto launch = [&](auto ncbmax) {
        constexpr int N = decltype(ncbmax)::value;
        if (!t.ntiles) return;
        if constexpr (!FWD) {
            if (masked) { hipLaunchKernelGGL((band_dft4_full_kernel<false, N, true>), dim3(t.ntiles), dim3(D4H_NT), 0, stream, a4, (const Tile4Dev*)t.d_tiles, t.ntiles); return; }
        }
        hipLaunchKernelGGL((band_dft4_full_kernel<FWD, N, false>), dim3(t.ntiles), dim3(D4H_NT), 0, stream, a4, (const Tile4Dev*)t.d_tiles, t.ntiles);
    };
    if (!split) {
        if ((rc = get_dft4_full_tiles(P, rows, &t, share, 0))) return rc;
        launch(std::integral_constant<int, 10>{});
        return XSQ_OK;
    }
    if ((rc = get_dft4_full_tiles(P, rows, &t, share, 1))) return rc;
    launch(std::integral_constant<int, 10>{});
    if ((rc = get_dft4_full_tiles(P, rows, &t, share, 2))) return rc;
    launch(std::integral_constant<int, 5>{});
    return XSQ_OK;
}

static int get_fft(xsq_plan* P, int inverse, int batch, FftPlan* out) {
    std::lock_guard<std::mutex> lk(P->mu);
    auto key = std::make_pair(inverse, batch);
    auto it = P->fft.find(key);
    if (it != P->fft.end()) {
        *out = it->second;
        return XSQ_OK;
    }
    static std::once_flag once;
    std::call_once(once, [] { rocfft_setup(); });
    FftPlan f;
    size_t len = (size_t)P->L;
    rocfft_status st = rocfft_plan_create(&f.plan, rocfft_placement_notinplace,
                                          inverse ? rocfft_transform_type_real_inverse
                                                  : rocfft_transform_type_real_forward,
                                          rocfft_precision_single, 1, &len, (size_t)batch, nullptr);
    if (st != rocfft_status_success) {
        set_error("rocfft_plan_create(L=%d, batch=%d, inverse=%d) failed: %d", P->L, batch, inverse, (int)st);
        return XSQ_ERR_FFT;
    }
    rocfft_plan_get_work_buffer_size(f.plan, &f.work_bytes);
    if (rocfft_execution_info_create(&f.info) != rocfft_status_success) {
        set_error("rocfft_execution_info_create failed");
        return XSQ_ERR_FFT;
    }
    P->fft[key] = f;
    *out = f;
    return XSQ_OK;
}

static int run_fft(const FftPlan& f, void* in, void* out, void* work, hipStream_t stream) {
    if (rocfft_execution_info_set_stream(f.info, stream) != rocfft_status_success) {
        set_error("rocfft_execution_info_set_stream failed");
        return XSQ_ERR_FFT;
    }
    if (f.work_bytes) {
        if (rocfft_execution_info_set_work_buffer(f.info, work, f.work_bytes) != rocfft_status_success) {
            set_error("rocfft_execution_info_set_work_buffer failed");
            return XSQ_ERR_FFT;
        }
    }
    void* ib[1] = {in};
    void* ob[1] = {out};
    rocfft_status st = rocfft_execute(f.plan, ib, ob, f.info);
    if (st != rocfft_status_success) {
        set_error("rocfft_execute failed: %d", (int)st);
        return XSQ_ERR_FFT;
    }
    return XSQ_OK;
}

static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

// Bands at least this long take the radix-4 kernel, shorter ones the dense engine.  Measured with the full-width kernel
// (r03o, sum of the four band kernels): 64 -> 1.53 ms, 48 -> 1.485, 32 -> 1.49, 16 -> 1.53; again with the buffer-addressed
// kernel (r4f, four full chunks): 48 -> 1.285 ms, 40 -> 1.272, 32 -> 1.280, 24 -> 1.252, 16 -> 1.304.
#ifndef XSQ_D4_MIN_LG_DEFAULT
#define XSQ_D4_MIN_LG_DEFAULT 24
#endif
#ifndef XSQ_FFT_NT_FWD
#define XSQ_FFT_NT_FWD 512
#endif
#ifndef XSQ_FFT_NT_INV
#define XSQ_FFT_NT_INV 512
#endif

// threads per row of the slice FFT kernels (slice_fft.h): XSQ_FFT_THREADS = "fwd,inv" of 256 / 512 overrides the
// defaults (diagnostic A/B switch; same results bit for bit)
static int fft_threads(int inverse) {
    static int u[2] = {-1, -1};
    if (u[0] < 0) {
        int f = XSQ_FFT_NT_FWD, i = XSQ_FFT_NT_INV;
        if (const char* e = getenv("XSQ_FFT_THREADS")) { if (sscanf(e, "%d,%d", &f, &i) < 2) i = f; }
        u[0] = f == 512 ? 512 : 256; u[1] = i == 512 ? 512 : 256;
    }
    return u[inverse ? 1 : 0];
}

static inline bool lds_fft(const xsq_plan* P) { return P->fft_backend == 0 && P->L == FFT_L && P->d_tgt != nullptr; }
static inline FftTables fft_tables(const xsq_plan* P) {
    return FftTables{P->d_T, P->d_T + FFT_R1 * FFT_M1, P->d_T + FFT_R1 * FFT_M1 + FFT_R2 * FFT_R3};
}

}  // namespace xsq

using namespace xsq;

extern "C" {

int xsq_abi_version(void) { return XSQ_ABI_VERSION; }

#ifndef XSQ_BUILD_ARCH
#define XSQ_BUILD_ARCH "unknown"
#endif
#ifndef XSQ_BUILD_FLAGS
#define XSQ_BUILD_FLAGS ""
#endif
const char* xsq_build_info(void) { return "arch=" XSQ_BUILD_ARCH "; flags=" XSQ_BUILD_FLAGS "; date=" __DATE__ " " __TIME__; }
const char* xsq_last_error(void) { return g_err; }

static int plan_build(xsq_plan* P, int L, int tr, int nbands, const int32_t* Lg, const int32_t* c,
                      const float* g, const double* gd, const float* tw);

int xsq_plan_create(xsq_plan** out, int L, int tr, int nbands, const int32_t* Lg, const int32_t* c,
                    const float* g, const double* gd, const float* tw) {
    XSQ_REQUIRE(out && Lg && c && g && gd && tw, "xsq_plan_create: null argument");
    XSQ_REQUIRE(L > 0 && L % 4 == 0 && tr % 2 == 0 && nbands >= 2, "xsq_plan_create: bad L/tr/nbands");
    xsq_plan* P = new xsq_plan();
    const int rc = plan_build(P, L, tr, nbands, Lg, c, g, gd, tw);
    if (rc != XSQ_OK) {          // frees whatever was allocated before the failure
        xsq_plan_destroy(P);
        return rc;
    }
    *out = P;
    return XSQ_OK;
}

static int plan_build(xsq_plan* P, int L, int tr, int nbands, const int32_t* Lg, const int32_t* c,
                      const float* g, const double* gd, const float* tw) {
    P->L = L; P->tr = tr; P->h = L / 4; P->nbins = L / 2 + 1; P->nbands = nbands;
    // blocks = runs of equal band length (nsgt/nsgtf.py:66-78)
    int64_t cum = 0;
    for (int j = 0; j < nbands;) {
        int k = j;
        while (k + 1 < nbands && Lg[k + 1] == Lg[j]) ++k;
        P->blocks.push_back(BlockHost{j, k - j + 1, Lg[j], cum});
        cum += (int64_t)(k - j + 1) * Lg[j];
        j = k + 1;
    }
    P->nblocks = (int)P->blocks.size();
    P->sumFT = cum;
    int64_t woff = 0, goff = 0;
    std::vector<int64_t> g_off(nbands);
    for (const BlockHost& b : P->blocks) {
        for (int f = 0; f < b.F; ++f) {
            const int j = b.first_band + f;
            if (Lg[j] % 4 != 0 || c[j] % 2 != 0 || Lg[j] > L / 2) {
                set_error("xsq_plan_create: band %d has Lg=%d c=%d (need Lg%%4==0, c even, Lg<=L/2)", j, Lg[j], c[j]);
                return XSQ_ERR_ARG;
            }
            BandDev d;
            d.Lg = Lg[j]; d.bin0 = c[j] - Lg[j] / 2; d.f = f; d.F = b.F; d.cum = b.cum;
            d.ldw = (int)round_up(2 * Lg[j], 16); d.w_off = woff; d.ent = 0;
            woff += round_up(2 * Lg[j], 64) * d.ldw;
            P->bands.push_back(d);
            g_off[j] = goff;
            goff += Lg[j];
        }
    }
    // ---- per-band real-ified DFT matrices --------------------------------------------
    // stored transposed, Wt[n][k] (K contiguous, as the tile engine wants its B operand).
    // analysis: k = 2p+ri over the band's window in spectrum order (bin = bin0 + p, window
    // index q = (p + Lg/2) mod Lg since windows are stored peak-at-0), n = 2t+ro over coefficients;
    // synthesis: k = 2t+ri over coefficients, n = 2p+ro over spectrum positions.
    std::vector<float> Wf((size_t)woff, 0.f), Wi((size_t)woff, 0.f);
    const double PI2 = 6.283185307179586476925286766559;
    for (int j = 0; j < nbands; ++j) {
        const BandDev& d = P->bands[j];
        const int n = d.Lg, ld = d.ldw;
        const double sign = ((c[j] / 2) % 2 == 0) ? 1.0 : -1.0;
        std::vector<double> cs(n), sn(n);
        for (int r = 0; r < n; ++r) { cs[r] = std::cos(PI2 * r / n); sn[r] = std::sin(PI2 * r / n); }
        float* wf = Wf.data() + d.w_off;
        float* wi = Wi.data() + d.w_off;
        for (int p = 0; p < n; ++p) {
            const int q = (p + n / 2) % n;
            const int bin = d.bin0 + p;
            const double conj = (bin < 0 || bin > L / 2) ? -1.0 : 1.0;  // mirrored bin: input is conj(U)
            const double ga = (double)g[g_off[j] + q] * sign / n;
            const double gs = gd[g_off[j] + q] * n * sign / L;
            for (int t = 0; t < n; ++t) {
                const int r = (int)(((int64_t)q * t) % n);
                // analysis: w = ga * e^{+i 2 pi q t / n};  (a_re + i conj a_im) * w
                const double wr = ga * cs[r], wim = ga * sn[r];
                wf[(size_t)(2 * t) * ld + 2 * p] = (float)wr;                    // re <- re
                wf[(size_t)(2 * t + 1) * ld + 2 * p] = (float)wim;               // im <- re
                wf[(size_t)(2 * t) * ld + 2 * p + 1] = (float)(-conj * wim);     // re <- im
                wf[(size_t)(2 * t + 1) * ld + 2 * p + 1] = (float)(conj * wr);   // im <- im
                // synthesis: rows are coefficients t, columns spectrum positions p:
                // w = gs * e^{-i 2 pi q t / n}
                const double vr = gs * cs[r], vi = -gs * sn[r];
                wi[(size_t)(2 * p) * ld + 2 * t] = (float)vr;                    // re <- re
                wi[(size_t)(2 * p + 1) * ld + 2 * t] = (float)vi;                // im <- re
                wi[(size_t)(2 * p) * ld + 2 * t + 1] = (float)(-vi);             // re <- im
                wi[(size_t)(2 * p + 1) * ld + 2 * t + 1] = (float)vr;            // im <- im
            }
        }
    }
    // ---- spectrum coverage (which bands add into bin k) ----------------------------------
    std::vector<int> cov_ptr(P->nbins + 1, 0), cov_band;
    {
        std::vector<std::vector<int>> cov(P->nbins);
        for (int j = 0; j < nbands; ++j)
            for (int p = 0; p < P->bands[j].Lg; ++p) {
                const int k = P->bands[j].bin0 + p;
                if (k >= 0 && k <= L / 2) cov[k].push_back(j);
            }
        for (int k = 0; k < P->nbins; ++k) {
            cov_ptr[k + 1] = cov_ptr[k] + (int)cov[k].size();
            cov_band.insert(cov_band.end(), cov[k].begin(), cov[k].end());
        }
    }
#define UP(dst, vec, T)                                                                           \
    do {                                                                                          \
        XSQ_HIP(hipMalloc(&(dst), (vec).size() * sizeof(T)));                                     \
        XSQ_HIP(hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice)); \
    } while (0)
    UP(P->d_Wf, Wf, float);
    UP(P->d_Wi, Wi, float);
    UP(P->d_bands, P->bands, BandDev);
    UP(P->d_cov_ptr, cov_ptr, int);
    UP(P->d_cov_band, cov_band, int);
#undef UP
    if (L == FFT_L) {   // tables of the hand-written slice FFT
        auto W = [&](int64_t j) {   // exp(-2 pi i j / L), argument reduced exactly
            j %= L;
            return make_float2((float)std::cos(PI2 * j / L), (float)(-std::sin(PI2 * j / L)));
        };
        std::vector<float2> T;
        for (int k1 = 0; k1 < FFT_R1; ++k1)
            for (int m = 0; m < FFT_M1; ++m) T.push_back(W(2 * (int64_t)k1 * m));
        for (int k2 = 0; k2 < FFT_R2; ++k2)
            for (int n3 = 0; n3 < FFT_R3; ++n3) T.push_back(W(2 * (int64_t)FFT_R1 * n3 * k2));
        for (int k = 0; k <= FFT_N; ++k) T.push_back(W(k));
        XSQ_HIP(hipMalloc(&P->d_T, T.size() * sizeof(float2)));
        XSQ_HIP(hipMemcpy(P->d_T, T.data(), T.size() * sizeof(float2), hipMemcpyHostToDevice));
        // phase-ordered entry table of the inverse gather: bands j with j % 4 == ph are laid out back
        // to back; valid only if the bands of one phase are pairwise disjoint inside [0, L/2]
        std::vector<int> tgt;
        bool ok = true;
        for (int ph = 0; ph < 4; ++ph) {
            P->phase_begin[ph] = (int)tgt.size();
            int last_end = -(1 << 30);
            for (int j = ph; j < nbands; j += 4) {
                BandDev& b = P->bands[j];
                b.ent = (int)tgt.size();
                const int lo = b.bin0 < 0 ? 0 : b.bin0;
                if (lo < last_end) ok = false;
                last_end = b.bin0 + b.Lg > L / 2 + 1 ? L / 2 + 1 : b.bin0 + b.Lg;
                for (int q = 0; q < b.Lg; ++q) {
                    const int k = b.bin0 + q;
                    tgt.push_back(k >= 0 && k <= L / 2 ? k : -1);
                }
            }
        }
        P->phase_begin[4] = (int)tgt.size();
        if (ok) {
            XSQ_HIP(hipMalloc(&P->d_tgt, tgt.size() * sizeof(int)));
            XSQ_HIP(hipMemcpy(P->d_tgt, tgt.data(), tgt.size() * sizeof(int), hipMemcpyHostToDevice));
            std::vector<unsigned short> t16(tgt.size() + 2, 0xFFFFu);
            for (size_t i = 0; i < tgt.size(); ++i) t16[i] = tgt[i] >= 0 ? (unsigned short)tgt[i] : 0xFFFFu;
            XSQ_HIP(hipMalloc(&P->d_tgt16, t16.size() * sizeof(unsigned short)));
            XSQ_HIP(hipMemcpy(P->d_tgt16, t16.data(), t16.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
            // the device copy of the band table was uploaded before .ent was known
            XSQ_HIP(hipMemcpy(P->d_bands, P->bands.data(), P->bands.size() * sizeof(BandDev), hipMemcpyHostToDevice));
        }
    }
    {   // ---- radix-4 band kernel tables (band_dft4.h): bands with Lg >= d4_min_lg -------------------------
        std::vector<Band4Dev> b4;
        std::vector<float> pf, pi;      // analysis / synthesis pools
        // bands at least this long take the radix-4 kernel (XSQ_D4_MIN_LG: diagnostic A/B of the split point)
        int d4_min_lg = XSQ_D4_MIN_LG_DEFAULT;
        if (const char* e = getenv("XSQ_D4_MIN_LG")) d4_min_lg = atoi(e) >= 16 ? atoi(e) : d4_min_lg;
        std::map<int, int64_t> doff, twoff, coff;
        auto alloc2 = [&](size_t n) { size_t o = pf.size(); pf.resize(o + n, 0.f); pi.resize(o + n, 0.f); return (int64_t)o; };
        for (int j = 0; j < nbands; ++j) {
            const BandDev& b = P->bands[j];
            if (b.Lg < d4_min_lg || b.Lg > 4 * D4_MPAD) { P->bands4_small.push_back(j); continue; }   // dense engine (gemm_tile.h)
            const int n = b.Lg, m = n / 4;
            Band4Dev d;
            memset(&d, 0, sizeof(d));
            d.Lg = n; d.m = m; d.bin0 = b.bin0; d.f = b.f; d.F = b.F; d.ent = b.ent; d.cum = b.cum; d.jband = j;
            d.ldd = (int)round_up(2 * m, 16);
            if (!doff.count(m)) {       // Dt[n' = (k, ro)][kk = (t1, ri)] of exp(-+2 pi i k t1 / m)
                const int64_t o = alloc2((size_t)round_up(2 * m, 64) * d.ldd);
                doff[m] = o;
                for (int k = 0; k < m; ++k)
                    for (int t1 = 0; t1 < m; ++t1) {
                        const int r = (int)(((int64_t)k * t1) % m);
                        const double cr = std::cos(PI2 * r / m), si = std::sin(PI2 * r / m);
                        for (int dir = 0; dir < 2; ++dir) {       // 0: analysis e^{+}, 1: synthesis e^{-}
                            std::vector<float>& pool = dir ? pi : pf;
                            const double dre = cr, dim = dir ? -si : si;
                            pool[o + (size_t)(2 * k) * d.ldd + 2 * t1] = (float)dre;
                            pool[o + (size_t)(2 * k) * d.ldd + 2 * t1 + 1] = (float)(-dim);
                            pool[o + (size_t)(2 * k + 1) * d.ldd + 2 * t1] = (float)dim;
                            pool[o + (size_t)(2 * k + 1) * d.ldd + 2 * t1 + 1] = (float)dre;
                        }
                    }
            }
            d.d_off = doff[m];
            d.K2 = m / 2 + 1; d.ldc = (int)round_up(d.K2, 8);
            if (!coff.count(m)) {       // band_dft4s.h: Ct[k][n] = cos(2 pi n k / m), St[k][n] = sin(..) with the direction's sign
                const size_t csz = (size_t)round_up(d.K2, 16) * d.ldc;
                const int64_t o = alloc2(2 * csz);
                coff[m] = o;
                for (int k = 0; k < d.K2; ++k)
                    for (int nn = 0; nn < d.K2; ++nn) {
                        const int r = (int)(((int64_t)k * nn) % m);
                        const double cr = std::cos(PI2 * r / m), si = (nn == 0 || 2 * nn == m) ? 0.0 : std::sin(PI2 * r / m);
                        pf[o + (size_t)k * d.ldc + nn] = (float)cr;
                        pi[o + (size_t)k * d.ldc + nn] = (float)cr;
                        pf[o + csz + (size_t)k * d.ldc + nn] = (float)(-si);     // analysis e^{+}: X[k] = P + i Q = P - i Q'
                        pi[o + csz + (size_t)k * d.ldc + nn] = (float)si;        // synthesis e^{-}: X[k] = P - i Q
                    }
            }
            d.c_off = coff[m];
            if (!twoff.count(n)) {      // twiddles w^(r t1), r = 1..3: [3][round_up(m, 8)] complex
                const int mpad = (m + 7) & ~7;
                const int64_t o = alloc2((size_t)3 * mpad * 2);
                twoff[n] = o;
                for (int r = 1; r <= 3; ++r)
                    for (int t1 = 0; t1 < m; ++t1) {
                        const int e = (r * t1) % n;
                        const double cr = std::cos(PI2 * e / n), si = std::sin(PI2 * e / n);
                        pf[o + 2 * ((size_t)(r - 1) * mpad + t1)] = (float)cr;
                        pf[o + 2 * ((size_t)(r - 1) * mpad + t1) + 1] = (float)si;
                        pi[o + 2 * ((size_t)(r - 1) * mpad + t1)] = (float)cr;
                        pi[o + 2 * ((size_t)(r - 1) * mpad + t1) + 1] = (float)(-si);
                    }
            }
            d.tw_off = twoff[n];
            d.win_off = alloc2((size_t)round_up(n, 4));
            const double sign = ((c[j] / 2) % 2 == 0) ? 1.0 : -1.0;
            for (int q = 0; q < n; ++q) {
                pf[d.win_off + q] = (float)((double)g[g_off[j] + q] * sign / n);
                pi[d.win_off + q] = (float)(gd[g_off[j] + q] * n * sign / L);
            }
            b4.push_back(d);
            P->bands4_m.push_back(m);
            P->d4_max_block = std::max<int64_t>(P->d4_max_block, (int64_t)b.F * b.Lg);
        }
        P->nbands4 = (int)b4.size();
        P->bands4_host.assign(reinterpret_cast<const unsigned char*>(b4.data()),
                              reinterpret_cast<const unsigned char*>(b4.data()) + b4.size() * sizeof(Band4Dev));
        if (P->nbands4) {
            XSQ_HIP(hipMalloc(&P->d_bands4, b4.size() * sizeof(Band4Dev)));
            XSQ_HIP(hipMemcpy(P->d_bands4, b4.data(), b4.size() * sizeof(Band4Dev), hipMemcpyHostToDevice));
            XSQ_HIP(hipMalloc(&P->d_pool4f, pf.size() * sizeof(float)));
            XSQ_HIP(hipMemcpy(P->d_pool4f, pf.data(), pf.size() * sizeof(float), hipMemcpyHostToDevice));
            XSQ_HIP(hipMalloc(&P->d_pool4i, pi.size() * sizeof(float)));
            XSQ_HIP(hipMemcpy(P->d_pool4i, pi.data(), pi.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    XSQ_HIP(hipMalloc(&P->d_tw, (size_t)L * sizeof(float)));
    XSQ_HIP(hipMemcpy(P->d_tw, tw, (size_t)L * sizeof(float), hipMemcpyHostToDevice));
    // ---- short bands inside k_slice_irfft (slice_fft.h: ShortSched) ----------------------------------------
    // eligible: the LDS slice FFT is in use, every band the radix-4 kernel does not take has Lg = 4m with
    // 4 <= m <= 15 and lies below the scratch area, and the scratch area fits above it
    if (P->d_tgt != nullptr && P->nbands4 > 0 && !P->bands4_small.empty()) {
        bool ok = true;
        int nent = 0, maxbin = 0;
        for (int j : P->bands4_small) {
            const BandDev& b = P->bands[j];
            if (b.Lg % 4 != 0 || b.Lg < 16 || b.Lg > 60) ok = false;
            nent += b.Lg;
            maxbin = std::max(maxbin, b.bin0 + b.Lg);
        }
        const int sc0 = 4096;
        if (maxbin >= sc0 || sc0 + nent > FFT_N + 1) ok = false;
        if (ok) {
            std::vector<ShortItem1> item1;
            std::vector<float2> tw1;
            std::vector<std::pair<int, int>> item2;     // (m, code)
            std::vector<int> stgt((size_t)nent, -1);
            std::vector<float> swd((size_t)nent, 0.f);
            int sc = 0;
            for (int ph = 0; ph < 4; ++ph) {
                P->short_begin[ph] = sc;
                for (int j : P->bands4_small) {
                    if (j % 4 != ph) continue;
                    const BandDev& b = P->bands[j];
                    const int n = b.Lg, m = n / 4;
                    const double sign = ((c[j] / 2) % 2 == 0) ? 1.0 : -1.0;
                    for (int t1 = 0; t1 < m; ++t1) {
                        item1.push_back(ShortItem1{(int)b.cum, b.F, b.f, n, t1, sc});
                        for (int r = 1; r <= 3; ++r) {
                            const int e = (r * t1) % n;
                            tw1.push_back(make_float2((float)std::cos(PI2 * e / n), (float)(-std::sin(PI2 * e / n))));
                        }
                    }
                    for (int r = 0; r < 4; ++r) {
                        item2.push_back({m, ((sc + r * m) << 4) | m});
                        for (int k = 0; k < m; ++k) {
                            const int q = 4 * k + r;
                            const int bin = b.bin0 + (q + n / 2) % n;
                            stgt[sc + r * m + k] = (bin >= 0 && bin <= L / 2) ? bin : -1;
                            swd[sc + r * m + k] = (float)(gd[g_off[j] + q] * n * sign / L);
                        }
                    }
                    sc += n;
                }
            }
            P->short_begin[4] = sc;
            std::stable_sort(item2.begin(), item2.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
            std::vector<int> codes;
            for (auto& it : item2) codes.push_back(it.second);
            // the gather of the long bands starts behind the short ones of each phase (bands of a phase are in band order)
            for (int ph = 0; ph < 4; ++ph) {
                int lo = P->phase_begin[ph + 1];
                for (int j = ph; j < nbands; j += 4) {
                    bool is_short = false;
                    for (int js : P->bands4_small) is_short = is_short || js == j;
                    if (!is_short) { lo = P->bands[j].ent; break; }
                }
                P->phase_long[ph] = lo;
                // every short band of the phase must precede every long one, or the split gather would skip entries
                for (int j = ph; j < nbands; j += 4) {
                    bool is_short = false;
                    for (int js : P->bands4_small) is_short = is_short || js == j;
                    if (is_short && P->bands[j].ent >= lo) ok = false;
                }
            }
            if (ok && item1.size() <= 1280 && codes.size() <= 768 && sc <= nent) {
                bool fits = true;
                for (int ph = 0; ph < 4; ++ph) fits = fits && (P->short_begin[ph + 1] - P->short_begin[ph] <= 1280);
                if (fits) {
#define UPV(dst, vec, T)                                                                          \
    do {                                                                                          \
        XSQ_HIP(hipMalloc((void**)&(dst), (vec).size() * sizeof(T)));                             \
        XSQ_HIP(hipMemcpy((dst), (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice)); \
    } while (0)
                    UPV(P->d_s_item1, item1, ShortItem1);
                    UPV(P->d_s_tw1, tw1, float2);
                    UPV(P->d_s_item2, codes, int);
                    UPV(P->d_s_tgt, stgt, int);
                    UPV(P->d_s_wd, swd, float);
#undef UPV
                    P->short_n1 = (int)item1.size(); P->short_n2 = (int)codes.size();
                    P->short_nent = nent; P->short_sc0 = sc0;
                }
            }
        }
    }
    return XSQ_OK;
}

int xsq_plan_destroy(xsq_plan* P) {
    if (!P) return XSQ_OK;
    for (auto& kv : P->fft) {
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
    }
    for (auto& kv : P->tiles) (void)hipFree(kv.second.d_tiles);
    (void)hipFree(P->d_bands4); (void)hipFree(P->d_pool4f); (void)hipFree(P->d_pool4i);
    (void)hipFree(P->d_T); (void)hipFree(P->d_tgt); (void)hipFree(P->d_tgt16); (void)hipFree(P->d_tw); (void)hipFree(P->d_Wf); (void)hipFree(P->d_Wi); (void)hipFree(P->d_bands);
    (void)hipFree(P->d_cov_ptr); (void)hipFree(P->d_cov_band);
    (void)hipFree(P->d_s_item1); (void)hipFree(P->d_s_tw1); (void)hipFree(P->d_s_item2); (void)hipFree(P->d_s_tgt); (void)hipFree(P->d_s_wd);
    delete P;
    return XSQ_OK;
}

int xsq_plan_num_blocks(const xsq_plan* P) { return P ? P->nblocks : XSQ_ERR_ARG; }

int xsq_plan_set_band_radix4(xsq_plan* P, int on) {
    XSQ_REQUIRE(P, "xsq_plan_set_band_radix4: null plan");
    P->band_radix4 = on ? 1 : 0;
    return XSQ_OK;
}

#if XSQ_D4_STAMP
extern "C" int xsq_debug_d4_stamps(unsigned long long* host, int tiles) {
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_d4_stamps), (size_t)(tiles < D4_STAMP_TILES ? tiles : D4_STAMP_TILES) * 64));
    return XSQ_OK;
}
#endif
#if XSQ_FFT_STAMP
// diagnostic builds only (not declared in the public header): copies the phase stamps of the last inverse launches
extern "C" int xsq_debug_fft_stamps(unsigned long long* host, int rows) {
    XSQ_HIP(hipDeviceSynchronize());
    XSQ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fft_stamps), (size_t)(rows < FFT_STAMP_ROWS ? rows : FFT_STAMP_ROWS) * 64));
    return XSQ_OK;
}
#endif

int xsq_plan_set_packed_fft(xsq_plan* P, int on) {
    XSQ_REQUIRE(P, "xsq_plan_set_packed_fft: null plan");
#if !XSQ_PACKED_FFT_KERNELS
    // The product library does not contain the packed-fp32 transform: next to v_mfma_f32_16x16x32_bf16 waves of another
    // stream it returned wrong values (tools/probe/pk_mfma_hazard.hip, cause below the ISA), no guard at this level can
    // see what other streams or a later precision switch put beside it, and it measured no faster.
    XSQ_REQUIRE(!on, "xsq_plan_set_packed_fft: this library is built without the packed-fp32 slice FFT kernels "
                "(diagnostic build: make PACKED_FFT=1)");
#endif
    P->packed_fft = on ? 1 : 0;
    return XSQ_OK;
}

int xsq_plan_set_short_inline(xsq_plan* P, int on) {
    XSQ_REQUIRE(P, "xsq_plan_set_short_inline: null plan");
    P->short_inline = on ? 1 : 0;
    return XSQ_OK;
}

int xsq_plan_set_fft_backend(xsq_plan* P, int backend) {
    XSQ_REQUIRE(P && (backend == 0 || backend == 1), "xsq_plan_set_fft_backend: backend must be 0 (auto) or 1 (rocFFT)");
    P->fft_backend = backend;
    return XSQ_OK;
}

int xsq_plan_block_table(const xsq_plan* P, int64_t* table) {
    XSQ_REQUIRE(P && table, "xsq_plan_block_table: null argument");
    for (int b = 0; b < P->nblocks; ++b) {
        table[4 * b + 0] = P->blocks[b].first_band;
        table[4 * b + 1] = P->blocks[b].F;
        table[4 * b + 2] = P->blocks[b].T;
        table[4 * b + 3] = P->blocks[b].cum;
    }
    return XSQ_OK;
}

int64_t xsq_plan_coefs_per_slice(const xsq_plan* P) { return P ? P->sumFT : XSQ_ERR_ARG; }

int xsq_plan_num_slices(const xsq_plan* P, int64_t n) {
    if (!P || n <= 0) return XSQ_ERR_ARG;
    const int64_t nb = (n + P->h - 1) / P->h;
    return (int)((nb + 1) / 2 + 1);
}

// workspace: seg | U | fft work
size_t xsq_slicqt_forward_workspace(xsq_plan* P, int BC, int64_t n) {
    if (!P || BC <= 0 || n <= 0) return 0;
    const size_t rows = (size_t)BC * xsq_plan_num_slices(P, n);
    FftPlan f;
    if (!lds_fft(P) && get_fft(P, 0, (int)rows, &f)) return 0;
    return al(rows * P->L * 4) + al(rows * P->nbins * 8) + al(f.work_bytes) + 256;
}

int xsq_slicqt_forward(xsq_plan* P, const float* x, int BC, int64_t n, float* coef, void* ws,
                       size_t ws_bytes, void* stream_) {
    return xsq_slicqt_forward_xin(P, x, BC, n, coef, nullptr, nullptr, nullptr, 0, ws, ws_bytes, stream_);
}

int xsq_slicqt_forward_xin(xsq_plan* P, const float* x, int BC, int64_t n, float* coef, float* xin, const float* mean,
                           const float* scale, int split, void* ws, size_t ws_bytes, void* stream_) {
    return xsq_slicqt_forward_rows(P, x, nullptr, BC, n, n, coef, xin, mean, scale, split, ws, ws_bytes, stream_);
}

int xsq_slicqt_forward_rows(xsq_plan* P, const float* x, const int64_t* x_rows, int BC, int64_t n, int64_t n_pad, float* coef,
                            float* xin, const float* mean, const float* scale, int split, void* ws, size_t ws_bytes,
                            void* stream_) {
    XSQ_REQUIRE(x, "xsq_slicqt_forward: null argument");
    return xsq_slicqt_forward_rows_indirect(P, x, nullptr, x_rows, BC, n, n_pad, coef, xin, mean, scale, split, ws, ws_bytes, stream_);
}

int xsq_slicqt_forward_rows_indirect(xsq_plan* P, const float* x, const float* const* x_slot, const int64_t* x_rows, int BC, int64_t n,
                                     int64_t n_pad, float* coef, float* xin, const float* mean, const float* scale, int split,
                                     void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(P && (x || x_slot) && coef && ws, "xsq_slicqt_forward: null argument");
    XSQ_REQUIRE(!xin || (mean && scale), "xsq_slicqt_forward_xin: xin needs the mean / scale tables");
    XSQ_REQUIRE(BC > 0 && n > 0 && n_pad >= n, "xsq_slicqt_forward: BC=%d n=%lld n_pad=%lld", BC, (long long)n, (long long)n_pad);
    hipStream_t stream = (hipStream_t)stream_;
    const int S = xsq_plan_num_slices(P, n_pad);        // the kernels read zeros outside [0, n): padding is a slice count
    const int rows = BC * S;
    XSQ_REQUIRE(S >= 2, "xsq_slicqt_forward: signal too short");
    XSQ_REQUIRE((int64_t)BC * S <= 65535, "xsq_slicqt_forward: BC*S=%lld rows exceed one launch", (long long)BC * S);
    // band_dft4.h addresses the slice spectra and the arena through 32-bit float offsets
    XSQ_REQUIRE((int64_t)2 * BC * S * (P->sumFT > P->nbins ? P->sumFT : P->nbins) < (1ll << 31),
                "xsq_slicqt_forward: BC*S=%lld rows exceed 2^31 floats of coefficients; split the call", (long long)BC * S);
    if (int rcg = d4_ranges_ok(P, (int64_t)BC * S, "xsq_slicqt_forward")) return rcg;
    FftPlan f;
    int rc = lds_fft(P) ? XSQ_OK : get_fft(P, 0, rows, &f);
    if (rc) return rc;
    char* w = (char*)ws;
    float* seg = (float*)w; w += al((size_t)rows * P->L * 4);
    float* U = (float*)w;   w += al((size_t)rows * P->nbins * 8);
    void* fwork = w;
    if ((size_t)(w - (char*)ws) + f.work_bytes > ws_bytes) {
        set_error("xsq_slicqt_forward: workspace too small (%zu needed, %zu given)",
                  (size_t)(w - (char*)ws) + f.work_bytes, ws_bytes);
        return XSQ_ERR_WORKSPACE;
    }
    if (lds_fft(P)) {
        XSQ_PROF("slice_rfft", stream);
#if XSQ_PACKED_FFT_KERNELS
        if (fft_threads(0) == 512 && P->packed_fft) hipLaunchKernelGGL((k_slice_rfft<512, true>), dim3(rows), dim3(512), 0, stream, x, P->d_tw, fft_tables(P), (float2*)U, S, n, P->h, x_rows, x_slot);
        else
#endif
        if (fft_threads(0) == 512) hipLaunchKernelGGL(k_slice_rfft<512>, dim3(rows), dim3(512), 0, stream, x, P->d_tw, fft_tables(P), (float2*)U, S, n, P->h, x_rows, x_slot);
        else hipLaunchKernelGGL(k_slice_rfft<256>, dim3(rows), dim3(256), 0, stream, x, P->d_tw, fft_tables(P), (float2*)U, S, n, P->h, x_rows, x_slot);
    } else {
        { XSQ_PROF("slice_window", stream);
        hipLaunchKernelGGL(k_slice_window, dim3((P->L + 255) / 256, rows), dim3(256), 0, stream, x, P->d_tw, seg,
                           S, n, P->L, P->h, x_rows, x_slot); }
        { XSQ_PROF("rfft_L", stream); rc = run_fft(f, seg, U, fwork, stream); }
        if (rc) return rc;
    }
    TileTable tt;
    rc = get_band_tiles(P, rows, &tt);
    if (rc) return rc;
    BandFwdOp op{U, coef, P->d_bands, P->d_Wf, BC, S, P->nbins, P->L, xin, mean, scale, split};
    if (P->band_radix4 && P->nbands4) {
        Band4Args a4{(const Band4Dev*)P->d_bands4, P->d_pool4f, U, coef, BC, S, P->nbins, P->L, 0, nullptr, 0, xin, mean, scale, split};
        XSQ_PROF("band_analysis_dft4", stream);
        if ((rc = launch_dft4<true>(P, a4, rows, 0, stream))) return rc;
    }
    if (tt.ntiles) { XSQ_PROF("band_analysis_gemm", stream);
    hipLaunchKernelGGL((grouped_gemm_kernel<BandFwdOp>), dim3(tt.ntiles), dim3(256), 0, stream, op,
                       tt.d_tiles, tt.ntiles); }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

// workspace: Z | fr | seg | fft work
size_t xsq_slicqt_inverse_workspace(xsq_plan* P, int BC, int S) {
    if (!P || BC <= 0 || S <= 0) return 0;
    const size_t rows = (size_t)BC * S;
    FftPlan f;
    if (!lds_fft(P) && get_fft(P, 1, (int)rows, &f)) return 0;
    if (lds_fft(P)) return al(rows * P->sumFT * 8) + 256;        // Z only: spectra and segments stay in LDS
    return al(rows * P->sumFT * 8) + al(rows * P->nbins * 8) + al(rows * P->L * 4) + al(f.work_bytes) + 256;
}

int xsq_slicqt_inverse(xsq_plan* P, const float* coef, int BC, int S, int64_t length, float* y, void* ws,
                       size_t ws_bytes, void* stream_) {
    return xsq_slicqt_inverse_rows(P, coef, BC, S, length, y, nullptr, ws, ws_bytes, stream_);
}

static int inverse_impl(xsq_plan* P, const float* coef, const float* mask, int BCx, int BC, int S, int64_t length, float* y,
                        const int64_t* row_offsets, void* ws, size_t ws_bytes, void* stream_);

int xsq_slicqt_inverse_rows(xsq_plan* P, const float* coef, int BC, int S, int64_t length, float* y,
                            const int64_t* row_offsets, void* ws, size_t ws_bytes, void* stream_) {
    return inverse_impl(P, coef, nullptr, 0, BC, S, length, y, row_offsets, ws, ws_bytes, stream_);
}

int xsq_slicqt_inverse_masked(xsq_plan* P, const float* masks, const float* mix, int BC, int BCx, int S, int64_t length,
                              float* y, const int64_t* row_offsets, void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(masks && mix, "xsq_slicqt_inverse_masked: null argument");
    XSQ_REQUIRE(BCx > 0 && BC % BCx == 0, "xsq_slicqt_inverse_masked: %d mix channels do not divide %d mask channels", BCx, BC);
    return inverse_impl(P, mix, masks, BCx, BC, S, length, y, row_offsets, ws, ws_bytes, stream_);
}

static int inverse_impl(xsq_plan* P, const float* coef, const float* mask, int BCx, int BC, int S, int64_t length, float* y,
                        const int64_t* row_offsets, void* ws, size_t ws_bytes, void* stream_) {
    XSQ_REQUIRE(P && coef && y && ws, "xsq_slicqt_inverse: null argument");
    XSQ_REQUIRE(BC > 0 && S >= 2 && length > 0, "xsq_slicqt_inverse: BC=%d S=%d length=%lld", BC, S,
                (long long)length);
    XSQ_REQUIRE(length <= (int64_t)2 * S * P->h, "xsq_slicqt_inverse: length %lld exceeds the %d slices",
                (long long)length, S);
    XSQ_REQUIRE((int64_t)BC * S <= 65535, "xsq_slicqt_inverse: BC*S=%lld rows exceed one launch", (long long)BC * S);
    // the band kernels carry arena offsets in 32 bits (band_dft4.h: xoff / moff; the mix arena is the smaller one)
    XSQ_REQUIRE((int64_t)2 * BC * S * P->sumFT < (1ll << 31), "xsq_slicqt_inverse: the coefficient arena of BC*S=%lld rows "
                "exceeds 2^31 floats; split the call (fewer stacked chunks)", (long long)BC * S);
    if (int rcg = d4_ranges_ok(P, (int64_t)BC * S, "xsq_slicqt_inverse")) return rcg;
    hipStream_t stream = (hipStream_t)stream_;
    const int rows = BC * S;
    FftPlan f;
    int rc = lds_fft(P) ? XSQ_OK : get_fft(P, 1, rows, &f);
    if (rc) return rc;
    char* w = (char*)ws;
    float* Z = (float*)w;    w += al((size_t)rows * P->sumFT * 8);
    float2* fr = (float2*)w; if (!lds_fft(P)) w += al((size_t)rows * P->nbins * 8);
    float* seg = (float*)w;  if (!lds_fft(P)) w += al((size_t)rows * P->L * 4);
    void* fwork = w;
    if ((size_t)(w - (char*)ws) + f.work_bytes > ws_bytes) {
        set_error("xsq_slicqt_inverse: workspace too small (%zu needed, %zu given)",
                  (size_t)(w - (char*)ws) + f.work_bytes, ws_bytes);
        return XSQ_ERR_WORKSPACE;
    }
    TileTable tt;
    rc = get_band_tiles(P, rows, &tt);
    if (rc) return rc;
    BandInvOp op{coef, Z, P->d_bands, P->d_Wi, BC, S, lds_fft(P) ? (int)P->sumFT : 0, mask, BCx};
    if (P->band_radix4 && P->nbands4) {
        Band4Args a4{(const Band4Dev*)P->d_bands4, P->d_pool4i, coef, Z, BC, S, P->nbins, P->L,
                     lds_fft(P) ? (int)P->sumFT : 0, mask, BCx, nullptr, nullptr, nullptr, 0};
        XSQ_PROF("band_synthesis_dft4", stream);
        if ((rc = launch_dft4<false>(P, a4, rows, mask ? BCx * S : 0, stream))) return rc;
    }
    // short bands: inside k_slice_irfft when the plan allows it, else dense GEMM + Z round trip
    const bool inl = lds_fft(P) && P->band_radix4 && P->short_inline && P->short_n1 > 0;
    if (tt.ntiles && !inl) { XSQ_PROF("band_synthesis_gemm", stream);
    hipLaunchKernelGGL((grouped_gemm_kernel<BandInvOp>), dim3(tt.ntiles), dim3(256), 0, stream, op,
                       tt.d_tiles, tt.ntiles); }
    if (lds_fft(P)) {
        // inverse slice FFT with the overlap-add fused in: even slices store, odd slices add (slice_fft.h)
        XSQ_PROF("slice_irfft_ola", stream);
        GatherSched G;
        G.tgt = P->d_tgt; G.tgt16 = P->d_tgt16; G.row_len = (int)P->sumFT;
        for (int i = 0; i < 5; ++i) G.begin[i] = P->phase_begin[i];
        for (int i = 0; i < 4; ++i) G.lo[i] = inl ? P->phase_long[i] : P->phase_begin[i];
        ShortSched SS;
        memset(&SS, 0, sizeof(SS));
        if (inl) {
            SS.item1 = (const ShortItem1*)P->d_s_item1; SS.tw1 = (const float2*)P->d_s_tw1; SS.item2 = P->d_s_item2;
            SS.stgt = P->d_s_tgt; SS.swd = P->d_s_wd;
            SS.n1 = P->short_n1; SS.n2 = P->short_n2; SS.nent = P->short_nent; SS.sc0 = P->short_sc0;
            for (int i = 0; i < 5; ++i) SS.begin[i] = P->short_begin[i];
        }
        const ShortIn SI{coef, mask, BC, mask ? BCx : BC};
        for (int parity = 0; parity < 2; ++parity) {
            const int nsl = (S + 1 - parity) / 2;
            OlaArgs O{y, row_offsets, S, P->h, parity, length};
            const bool sh = SS.n1 > 0;
            bool fast4 = !sh && fft_threads(1) == 512 && G.tgt16 != nullptr && !getenv("XSQ_FFT_GATHER2");
            for (int ph = 0; ph < 4; ++ph) fast4 = fast4 && (G.begin[ph + 1] - (G.lo[ph] & ~1) <= 2 * 512 * ((4864 / 2 + 1 + 511) / 512));
#if XSQ_PACKED_FFT_KERNELS
            if (fast4 && P->packed_fft) hipLaunchKernelGGL((k_slice_irfft<512, true, false, true>), dim3(BC * nsl), dim3(512), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else if (!fast4 && !sh && fft_threads(1) == 512 && P->packed_fft) hipLaunchKernelGGL((k_slice_irfft<512, true>), dim3(BC * nsl), dim3(512), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else
#endif
            if (fast4) hipLaunchKernelGGL((k_slice_irfft<512, false, false, true>), dim3(BC * nsl), dim3(512), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else if (fft_threads(1) == 512 && sh) hipLaunchKernelGGL((k_slice_irfft<512, false, true>), dim3(BC * nsl), dim3(512), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else if (fft_threads(1) == 512) hipLaunchKernelGGL(k_slice_irfft<512>, dim3(BC * nsl), dim3(512), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else if (sh) hipLaunchKernelGGL((k_slice_irfft<256, false, true>), dim3(BC * nsl), dim3(256), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
            else hipLaunchKernelGGL(k_slice_irfft<256>, dim3(BC * nsl), dim3(256), 0, stream, (const float2*)Z, G, fft_tables(P), O, SS, SI);
        }
        XSQ_HIP(hipGetLastError());
        return XSQ_OK;
    } else {
        { XSQ_PROF("spectrum_gather", stream);
        hipLaunchKernelGGL(k_spectrum_gather, dim3((P->nbins + 255) / 256, rows), dim3(256), 0, stream, Z,
                           P->d_bands, P->d_cov_ptr, P->d_cov_band, fr, BC, S, P->nbins); }
        { XSQ_PROF("irfft_L", stream); rc = run_fft(f, fr, seg, fwork, stream); }
        if (rc) return rc;
    }
    { XSQ_PROF("overlap_add", stream);
    hipLaunchKernelGGL(k_overlap_add, dim3((unsigned)((length + 255) / 256), BC), dim3(256), 0, stream, seg, y,
                       row_offsets, S, length, P->L, P->h); }
    XSQ_HIP(hipGetLastError());
    return XSQ_OK;
}

}  // extern "C"
