// Standalone probe of the SIMD issue model behind `roofline_issue`: how long does a K-step-shaped loop
// body take on gfx950 as a function of what it holds -- fp32 MFMAs, other vector instructions, LDS reads, LDS writes, global
// (buffer) loads, a workgroup barrier -- and of the number of waves per SIMD?  No library, no torch.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
//
// Every workgroup is 256 threads = one wave per SIMD; `wgs` workgroups per CU (dynamic LDS sized so that exactly that many
// fit) give `wgs` waves per SIMD.  A body = NM x v_mfma_f32_16x16x4_f32 (over four independent accumulators), NV x v_fmac_f32,
// NL x ds_read_b128, NW x ds_write_b128, NG x buffer_load_dwordx4 (L2-resident 64 KB window), BAR x s_barrier; waits for the
// memory instructions sit at the END of the body (the loads of body i are consumed by nobody: only their issue and their
// completion before the next body count).  Printed: cycles per body per SIMD (2.4 GHz wall clock) next to the model
//   32 NM + 4 NV   (the sum rule of section 4.1),
// so that the cost of each other ingredient reads off as the difference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NM, int NV, int NL, int NW, int NG, int BAR, int BUNCH = 0, int ND = 0, int NW2 = 0, int NG2 = 0, int NG1 = 0>
__global__ __launch_bounds__(256) void k_body(float* sink, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float va[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = 1.f + tid * 1e-6f * i;
    const float a = 1.0f + 1e-3f * (tid & 7), b = 0.5f;
    // conflict-free 16-byte slots: lane l -> slot l (+ 64 per instruction)
    const unsigned laddr = 16u * (unsigned)(tid & 63) + 4096u * (unsigned)(tid >> 6);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long pa = (unsigned long long)src;
    const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)pa), __builtin_amdgcn_readfirstlane((int)(unsigned)(pa >> 32) & 0xFFFF),
                      __builtin_amdgcn_readfirstlane(65536), __builtin_amdgcn_readfirstlane(0x00020000)};
    const unsigned goff = 16u * (unsigned)tid;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 65536, 0x00020000);
    f32x4 lv[NL > 0 ? NL : 1], gv[NG > 0 ? NG : 1];
    typedef float f32x2g __attribute__((ext_vector_type(2)));
    f32x2g gv2[NG2 > 0 ? NG2 : 1];
    float gv1[NG1 > 0 ? NG1 : 1];
    const f32x4 wv = {a, b, a, b};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 wv2 = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NG; ++i)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(gv[i]) : "v"(goff), "s"(rs), "n"(0) : "memory");
        // NG2 x buffer_load_dwordx2, NG1 x buffer_load_dword (8 / 4 bytes per lane)
#pragma unroll
        for (int i = 0; i < NG2; ++i)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(gv2[i]) : "v"(goff), "s"(rs), "n"(0) : "memory");
#pragma unroll
        for (int i = 0; i < NG1; ++i)
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(gv1[i]) : "v"(goff), "s"(rs), "n"(0) : "memory");
        // ND x buffer_load_dwordx4 ... lds (LDS-DMA: lane l's 16 bytes land at base + 16 l; no register, no ds_write)
#pragma unroll
        for (int i = 0; i < ND; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, lds + 8192 + 1024 * (threadIdx.x >> 6) + 256 * (i & 3) * 4, 16, (int)goff, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lv[i]) : "v"(laddr), "n"((i & 3) * 1024) : "memory");
        // MFMAs and vector instructions interleaved evenly
        constexpr int VPM = (NM > 0 && !BUNCH) ? (NV + NM - 1) / NM : 0;      // BUNCH: every vector instruction behind the MFMAs (a staging phase)
        int vdone = 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < VPM; ++j)
                if (vdone < NV) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(va[vdone & 7]) : "v"(a), "v"(b)); ++vdone; }
        }
        if (NM == 0 || BUNCH) {
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(va[j & 7]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < NW; ++i)
            asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(laddr), "v"(wv), "n"(16384 + (i & 3) * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < NW2; ++i)         // NW2 x ds_write_b64 (8 bytes per lane)
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(laddr), "v"(wv2), "n"(24576 + (i & 3) * 1024) : "memory");
        if (NG > 0 || ND > 0 || NG2 > 0 || NG1 > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NL > 0 || NW > 0 || NW2 > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) asm volatile("s_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i];
#pragma unroll
    for (int i = 0; i < NL; ++i) s += lv[i][0];
#pragma unroll
    for (int i = 0; i < NG; ++i) s += gv[i][0];
#pragma unroll
    for (int i = 0; i < NG2; ++i) s += gv2[i][0];
#pragma unroll
    for (int i = 0; i < NG1; ++i) s += gv1[i];
    if (s == 1.2345e-30f) sink[tid] = s;
}

static float* d_sink; static float* d_src; static int g_cus = 256;

template <int NM, int NV, int NL, int NW, int NG, int BAR, int BUNCH = 0, int ND = 0, int NW2 = 0, int NG2 = 0, int NG1 = 0>
static void run(const char* what) {
    printf("%-44s NM %3d NV %3d NL %2d NW %2d NG %2d BAR %d | 32 NM + 4 NV %5d |", what, NM, NV, NL, NW, NG, BAR, 32 * NM + 4 * NV);
    for (int wgs = 1; wgs <= 4; ++wgs) {
        // dynamic LDS: floor(160 KB / wgs) minus a margin -> exactly `wgs` workgroups per CU
        const size_t lds = (size_t)(160 * 1024 / wgs) - (wgs == 1 ? 0 : 1024);
        auto kern = k_body<NM, NV, NL, NW, NG, BAR, BUNCH, ND, NW2, NG2, NG1>;
        HIPCHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int iters = 8000;
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {          // (the first repetition also warms the clocks)
            HIPCHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(g_cus * wgs), dim3(256), lds, 0, d_sink, d_src, iters);
            HIPCHECK(hipEventRecord(e1));
            HIPCHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * wgs);      // 2.4 GHz cycles per body and wave = SIMD time per body
            if (rep > 0 && cyc < best) best = cyc;
        }
        printf(" %dw %7.1f", wgs, best);
        HIPCHECK(hipEventDestroy(e0)); HIPCHECK(hipEventDestroy(e1));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    HIPCHECK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, g_cus, p.clockRate);
    HIPCHECK(hipMalloc(&d_sink, 4096));
    HIPCHECK(hipMalloc(&d_src, 1 << 20));
    HIPCHECK(hipMemset(d_src, 0, 1 << 20));
    printf("columns: SIMD time per body in 2.4 GHz cycles with 1..4 waves per SIMD (best of 3)\n");
    run<40, 0, 0, 0, 0, 0>("MFMA only (calibration: 1280 at 2.4 GHz)");
    run<0, 64, 0, 0, 0, 0>("vector only");
    run<40, 20, 0, 0, 0, 0>("MFMA + 20 vector");
    run<40, 40, 0, 0, 0, 0>("MFMA + 40 vector");
    run<40, 80, 0, 0, 0, 0>("MFMA + 80 vector");
    run<40, 40, 8, 0, 0, 0>("MFMA + 40 vector + 8 ds_read_b128");
    run<40, 40, 16, 0, 0, 0>("MFMA + 40 vector + 16 ds_read_b128");
    run<40, 40, 0, 8, 0, 0>("MFMA + 40 vector + 8 ds_write_b128");
    run<40, 40, 0, 16, 0, 0>("MFMA + 40 vector + 16 ds_write_b128");
    run<40, 40, 0, 0, 8, 0>("MFMA + 40 vector + 8 buffer_load_dwordx4");
    run<40, 40, 0, 0, 16, 0>("MFMA + 40 vector + 16 buffer_load_dwordx4");
    run<40, 40, 0, 0, 0, 1>("MFMA + 40 vector + barrier");
    run<16, 55, 4, 7, 11, 1>("band K-step, ncb = 2");
    run<40, 55, 7, 7, 11, 1>("band K-step, ncb = 5");
    run<80, 55, 12, 7, 11, 1>("band K-step, ncb = 10");
    run<40, 0, 7, 7, 11, 1>("  ncb = 5 without its vector instructions");
    run<0, 55, 7, 7, 11, 1>("  ncb = 5 without its MFMAs");
    run<40, 55, 0, 0, 0, 0, 1>("MFMA + 55 vector, vector behind the MFMAs");
    run<40, 55, 7, 7, 11, 1, 1>("band K-step, ncb = 5, vector behind the MFMAs");
    run<40, 55, 7, 7, 7, 1, 1>("  with 7 instead of 11 global loads");
    run<40, 55, 7, 4, 11, 1, 1>("  with 4 instead of 7 LDS writes");
    run<40, 55, 7, 4, 7, 1, 1>("  with both");
    run<40, 55, 7, 3, 11, 1, 1, 0, 4>("  as built: 3 x ds_write_b128 + 4 x ds_write_b64");
    run<40, 55, 7, 5, 11, 1, 1, 0, 0>("  the four residues as two ds_write_b128 (5 x b128)");
    run<40, 55, 7, 4, 8, 1, 1, 3>("  matrix slab by LDS-DMA: 8 loads + 3 DMA, 4 LDS writes");
    run<40, 55, 7, 2, 8, 1, 1, 3>("  ... and the four residues in two 16-byte writes");
    run<16, 55, 4, 7, 11, 1, 1>("band K-step, ncb = 2, vector behind the MFMAs");
    run<80, 55, 12, 7, 11, 1, 1>("band K-step, ncb = 10, vector behind the MFMAs");
    // Winograd chunk (16 channels, one wave): 60 MFMAs, ~110 vector, 15 B + 5 A + 5 tail reads, 4 + 3 staging writes, 7 loads, barrier
    run<60, 110, 25, 7, 7, 1, 0>("Winograd 16-channel chunk (interleaved)");
    run<60, 110, 25, 7, 7, 1, 1>("Winograd 16-channel chunk (vector behind)");
    run<60, 110, 0, 0, 0, 0, 0>("  its MFMAs + vector only");
    // the pair-contracted band kernel (band_dft4s.h): K-step of 8 pairs = 16 ncb MFMAs, 107 vector, 13 LDS reads, 9 LDS writes, 17 loads
    run<16, 107, 13, 9, 1, 1, 1, 0, 0, 8, 8>("pair-contracted band K-step, 1 block (8 x 8-byte + 8 x 4-byte + 1 x 16-byte loads)");
    run<32, 107, 13, 9, 1, 1, 1, 0, 0, 8, 8>("pair-contracted band K-step, 2 blocks");
    run<48, 107, 13, 9, 1, 1, 1, 0, 0, 8, 8>("pair-contracted band K-step, 3 blocks");
    run<16, 107, 13, 9, 5, 1, 1, 0, 0, 4, 0>("  1 block, two pairs per thread: 4 x 16-byte + 4 x 8-byte + 1 loads");
    run<16, 107, 13, 9, 1, 1, 1, 0, 0, 8, 0>("  1 block, without the mask loads");
    run<16, 107, 0, 0, 0, 0, 1>("  1 block, MFMAs + vector only");
    return 0;
}
