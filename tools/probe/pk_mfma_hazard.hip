// Standalone probe for the "packed-fp32 VALU next to bf16 MFMA" finding (csrc/Makefile NOPK):
// does a wave of k_slice_rfft (the LDS slice FFT, ~1,300 v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 per butterfly pass
// when packed-fp32 ops are enabled) return different bits when waves of ANOTHER kernel that only issues MFMAs are
// resident on the same CU?  No pipeline, no torch, no library: the victim kernel (included from csrc/slice_fft.h,
// unchanged) on one stream, a bare MFMA loop on a second stream, outputs compared bitwise with the victim run alone.
//
//   hipcc -O3 --offload-arch=gfx950 -I xumx_slicq_amd/csrc tools/probe/pk_mfma_hazard.hip -o /tmp/pk_on            (packed ops on)
//   hipcc -O3 --offload-arch=gfx950 -I xumx_slicq_amd/csrc -DPROBE_NO_PK -Xclang -target-feature -Xclang -packed-fp32-ops ... -o /tmp/pk_off
//   /tmp/pk_on; /tmp/pk_off          prints, per aggressor, the number of trials (of 20) with a corrupted transform
// Round 3: the library's slicqt.hip is built with packed ops ON and the SLP vectoriser OFF (csrc/Makefile), so that only
// the hand-placed v_pk_* codelets of k_slice_rfft<512, true> / k_slice_irfft<512, true> are packed:
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -DPROBE_HAND_PK -I xumx_slicq_amd/csrc ... -o /tmp/pk_hand
// runs the scalar k_slice_rfft<512> (control) and the hand-packed k_slice_rfft<512, true> as victims, with
// v_mfma_f32_16x16x4_f32 among the aggressors (the fp32 path runs 32x32x2 and 16x16x4 MFMAs beside the transforms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace xsq { void set_error(const char*, ...) {} }
#include "slice_fft.h"

using namespace xsq;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// aggressors: one wave per SIMD on every CU, MFMAs back to back for `iters` iterations, no memory traffic
template <int KIND>
__global__ __launch_bounds__(256) void k_aggressor(float* sink, int iters) {
    f32x16 acc = {0};
    f32x4 acc4 = {0};
    const float s = 1.0f + 1e-3f * (threadIdx.x & 7);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)s; b[i] = (__bf16)(0.5f * s); }
    float v = s;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 0.5f * s, acc, 0, 0, 0);
        } else if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
        } else if (KIND == 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, 0.5f * s, acc4, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) v = fmaf(v, 0.999f, 1e-3f);      // plain VALU, no MFMA
        }
    }
    if (acc[0] + acc4[0] + v == 1.2345e-30f) sink[0] = 1.f;
}

// second victim: registers only -- a chain of packed (or scalar) fp32 FMAs on exactly representable values, no LDS,
// no memory traffic inside the loop; out[t] must equal the same chain evaluated alone
template <bool PACKED>
__global__ __launch_bounds__(256) void k_victim_regs(float2* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float2 a = make_float2((float)(t & 1023), (float)((t >> 3) & 1023));
    const float2 m = make_float2(0.5f, 0.25f), c = make_float2(3.f, 5.f);
    for (int i = 0; i < iters; ++i) {
#ifndef PROBE_NO_PK          // (the assembler rejects the packed opcode when the target feature is switched off)
        if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
        else
#endif
            asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5" : "+v"(a.x), "+v"(a.y) : "v"(m.x), "v"(m.y), "v"(c.x), "v"(c.y));
    }
    out[t] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const int BC = 8, h = 4515, S = 292;
    const int64_t n = 2621440;
    const int rows = BC * S;
    std::vector<float> hx((size_t)BC * n), htw(FFT_L);
    srand(1);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < FFT_L; ++i) htw[i] = 0.5f + 0.5f * (float)i / FFT_L;
    const size_t nT = (size_t)FFT_R1 * FFT_M1 + FFT_R2 * FFT_R3 + FFT_N + 1;
    std::vector<float2> hT(nT);
    for (auto& t : hT) { const float ph = 6.2831853f * rand() / RAND_MAX; t = make_float2(cosf(ph), sinf(ph)); }   // any unit twiddles will do
    float *x, *tw, *sink;
    float2 *T, *U, *Uref;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&tw, FFT_L * 4)); CK(hipMalloc(&T, nT * 8)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&U, (size_t)rows * (FFT_N + 1) * 8)); CK(hipMalloc(&Uref, (size_t)rows * (FFT_N + 1) * 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw, htw.data(), FFT_L * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(T, hT.data(), nT * 8, hipMemcpyHostToDevice));
    const FftTables tabs{T, T + FFT_R1 * FFT_M1, T + FFT_R1 * FFT_M1 + FFT_R2 * FFT_R3};
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
#ifdef PROBE_HAND_PK
    const int nvictims = 2;
    const char* vnames[] = {"k_slice_rfft<512> (scalar)", "k_slice_rfft<512, true> (hand-placed v_pk_*)"};
#else
    const int nvictims = 1;
    const char* vnames[] = {"k_slice_rfft<256>"};
#endif
    std::vector<float2> a((size_t)rows * (FFT_N + 1)), b(a.size()), a0;
    const char* names[] = {"v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_bf16", "plain VALU (no MFMA)", "none",
                           "v_mfma_f32_16x16x4_f32"};
    for (int vic = 0; vic < nvictims; ++vic) {
    auto victim = [&](float2* out) {
#ifdef PROBE_HAND_PK
        if (vic == 1) hipLaunchKernelGGL((k_slice_rfft<512, true>), dim3(rows), dim3(512), 0, s1, x, tw, tabs, out, S, n, h);
        else hipLaunchKernelGGL((k_slice_rfft<512, false>), dim3(rows), dim3(512), 0, s1, x, tw, tabs, out, S, n, h);
#else
        hipLaunchKernelGGL(k_slice_rfft<256>, dim3(rows), dim3(256), 0, s1, x, tw, tabs, out, S, n, h);
#endif
    };
    victim(Uref);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(a.data(), Uref, a.size() * 8, hipMemcpyDeviceToHost));
    if (vic == 0) a0 = a;
    else printf("victim %s alone vs victim %s alone: %s\n", vnames[vic], vnames[0], memcmp(a.data(), a0.data(), a.size() * 8) ? "DIFFERENT BITS" : "bitwise equal");
    printf("victim %s\n", vnames[vic]);
    for (int kind = 0; kind < 6; ++kind) {
        int bad = 0;
        long badvals = 0;
        for (int trial = 0; trial < 20; ++trial) {
            CK(hipMemsetAsync(U, 0, a.size() * 8, s1));
            CK(hipDeviceSynchronize());
            const int iters = 60000;       // ~ a few ms: covers the victim
            if (kind == 0) hipLaunchKernelGGL(k_aggressor<0>, dim3(512), dim3(256), 0, s2, sink, iters);
            if (kind == 1) hipLaunchKernelGGL(k_aggressor<1>, dim3(512), dim3(256), 0, s2, sink, iters / 2);
            if (kind == 2) hipLaunchKernelGGL(k_aggressor<2>, dim3(512), dim3(256), 0, s2, sink, iters);
            if (kind == 3) hipLaunchKernelGGL(k_aggressor<3>, dim3(512), dim3(256), 0, s2, sink, iters / 4);
            if (kind == 5) hipLaunchKernelGGL(k_aggressor<4>, dim3(512), dim3(256), 0, s2, sink, iters);
            victim(U);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(b.data(), U, b.size() * 8, hipMemcpyDeviceToHost));
            if (memcmp(a.data(), b.data(), a.size() * 8) != 0) {
                ++bad;
                for (size_t i = 0; i < a.size(); ++i) badvals += (a[i].x != b[i].x) || (a[i].y != b[i].y);
            }
        }
        printf("aggressor %-28s: %2d / 20 transforms differ from the run alone (%ld values)\n", names[kind], bad, badvals);
        fflush(stdout);
    }
    }
    // ---- registers-only victim (explicit v_pk_fma_f32 vs v_fma_f32 chains; same binary in both builds) ----
    const int nb = 2048, nt = nb * 256, vit = 200000;
    float2 *o1, *o2;
    CK(hipMalloc(&o1, (size_t)nt * 8)); CK(hipMalloc(&o2, (size_t)nt * 8));
    std::vector<float2> r1(nt), r2(nt);
#ifdef PROBE_NO_PK
    const int first = 0;
#else
    const int first = 1;
#endif
    for (int packed = first; packed >= 0; --packed) {
        if (packed) hipLaunchKernelGGL(k_victim_regs<true>, dim3(nb), dim3(256), 0, s1, o1, vit);
        else hipLaunchKernelGGL(k_victim_regs<false>, dim3(nb), dim3(256), 0, s1, o1, vit);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r1.data(), o1, (size_t)nt * 8, hipMemcpyDeviceToHost));
        for (int kind = 0; kind < 3; ++kind) {
            int bad = 0;
            for (int trial = 0; trial < 10; ++trial) {
                if (kind == 0) hipLaunchKernelGGL(k_aggressor<0>, dim3(512), dim3(256), 0, s2, sink, 60000);
                if (kind == 1) hipLaunchKernelGGL(k_aggressor<1>, dim3(512), dim3(256), 0, s2, sink, 30000);
                if (kind == 2) hipLaunchKernelGGL(k_aggressor<2>, dim3(512), dim3(256), 0, s2, sink, 60000);
                if (packed) hipLaunchKernelGGL(k_victim_regs<true>, dim3(nb), dim3(256), 0, s1, o2, vit);
                else hipLaunchKernelGGL(k_victim_regs<false>, dim3(nb), dim3(256), 0, s1, o2, vit);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(r2.data(), o2, (size_t)nt * 8, hipMemcpyDeviceToHost));
                bad += memcmp(r1.data(), r2.data(), (size_t)nt * 8) != 0;
            }
            printf("registers-only %s chain beside %-26s: %2d / 10 runs differ\n", packed ? "v_pk_fma_f32" : "v_fma_f32   ", names[kind], bad);
            fflush(stdout);
        }
    }
    return 0;
}
