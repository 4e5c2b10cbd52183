// Shared host/device definitions for the xumx-sliCQ MI355X hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define XSQ_OK 0
#define XSQ_ERR_ARG (-1)
#define XSQ_ERR_HIP (-2)
#define XSQ_ERR_FFT (-3)
#define XSQ_ERR_WORKSPACE (-4)

namespace xsq {

void set_error(const char* fmt, ...);

#define XSQ_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::xsq::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,               \
                             hipGetErrorString(e_));                                    \
            return XSQ_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define XSQ_REQUIRE(cond, ...)                                                          \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::xsq::set_error(__VA_ARGS__);                                              \
            return XSQ_ERR_ARG;                                                         \
        }                                                                               \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Sixteen zero bytes in device memory: an out-of-range operand load can be pointed here instead of being
// branched around (used by the band operators of slicqt.hip, whose loads must stay unpredicated next to the
// mask stream).
// (not `const`: a constant-address-space object next to a global pointer in one select turns the load into a
// flat_load, which counts against both memory counters)
static __device__ float4 g_zero16;       // zero-initialised, never written
__device__ __forceinline__ const float* zero16() { return reinterpret_cast<const float*>(&g_zero16); }

// Whitened magnitude of one coefficient, (|c| + mean) * scale (model.py:238-242), spelled once so that the separate
// magnitude pass (cdae.hip) and the fused analysis epilogues (band_dft4.h, slicqt.hip) round identically.
__device__ __forceinline__ float whiten_mag(float re, float im, float mu, float sc) {
    return (sqrtf(fmaf(re, re, im * im)) + mu) * sc;
}

// ---- buffer addressing (gfx950).  A load or store through a buffer descriptor takes a 32-bit per-lane byte offset and a
// scalar offset; past the descriptor's range -- lane offset + scalar offset + immediate against num_records, probed with
// tools/probe/buf_range.hip -- a load returns 0 and a store is dropped.  The fp32 MFMA kernels use it to
// keep address arithmetic, clamps, zero fills and predicates out of their K loops: on this part those vector instructions
// are not hidden behind the MFMAs of the SIMD's other waves (band_dft4.h, "Vector issue").  Every range is kept below 2^30
// bytes where the launch is built: BUF_OOB (a row switched off) still lies past it after a backward displacement of up to
// 2^30, BUF_OOB_COL (a column switched off) after being added to an in-range offset, and so does their sum.
typedef unsigned buf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned buf_u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0x80000000u, BUF_OOB_COL = 0x40000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)vo, so, 0));
}
__device__ __forceinline__ float2 buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)vo, so, 0));
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, so, 0));
}
__device__ __forceinline__ void buf_st2(float2 v, __amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u32x2, v), r, (int)vo, so, 0);
}
// NOTE (round 6, cdae_l4f.h / cdae_wino.h): pass `so` = 0 to buf_st4.  With the scalar offset in an SGPR -- a run-time value, or
// a compile-time constant that is not an inline constant (128, 192, 1024 ...: the compiler moves it into an SGPR) -- a 16-byte
// store lost its first data dword to the next vector instruction that wrote that register: the store-data hazard the compiler
// guards only for stores WITHOUT an SGPR offset.  Put every displacement into the lane offset `vo`.
__device__ __forceinline__ void buf_st4(float4 v, __amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(buf_u32x4, v), r, (int)vo, so, 0);
}

struct TileDev;
// tiles of one group: 128-row x 64-column tiles, plus a 32-column tile when the N tail is <= 32
template <class Vec>
static inline void push_group_tiles(Vec& out, int group, int64_t M, int N, int BM = 128) {
    for (int64_t m0 = 0; m0 < M; m0 += BM)           // n fastest: the N-tiles of one M-tile run back
        for (int n0 = 0; n0 < N; n0 += 64)           // to back and re-read the same A rows from L2
            out.push_back({group, (int)m0, n0, (N - n0 <= 32) ? 1 : 0});
}

// One band of the transform as the kernels see it.
struct BandDev {
    int Lg;        // band length (coefficients per slice), multiple of 4
    int bin0;      // first spectrum bin of the band's window: c - Lg/2 (may be < 0 for DC)
    int f;         // row of the band inside its block
    int F;         // rows of the block
    int64_t cum;   // sum over earlier blocks of F_b*T_b (complex coefficients per channel-slice)
    int64_t w_off; // float offset of the band's DFT matrix inside Wf / Wi
    int ldw;       // row length (floats) of that matrix, stored transposed Wt[n][k]: round_up(2*Lg, 16)
    int ent;       // entry offset of the band inside a row of the phase-ordered synthesis output (slice_fft.h)
};

// One tile of work of a grouped GEMM launch.
struct TileDev {
    int group;  // band (sliCQT) or layer-group (CDAE)
    int m0;     // first row of the tile
    int n0;     // first column of the tile
    int narrow; // 1: the tile is 32 columns wide (N tail), 0: 64
};

}  // namespace xsq
