// Microbenchmark: what fp32 MFMA rate and clock does this MI355X sustain, and what do a
// per-K-step barrier and LDS staging writes cost?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: bare, 1: barrier per 16 MFMA, 2: barrier + 3 LDS writes, 3: barrier + 6 ds_read_b128 + writes
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[7680];
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a[8], b0[8], b1[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b0[i] = blockIdx.x * 1e-3f + i; b1[i] = b0[i] + 1.f; }
    float4 w = make_float4(a[0], a[1], b0[0], b1[1]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            const float* p = lds + cur * 3840 + (threadIdx.x & 63) * 20 + (threadIdx.x >> 6) * 640;
            float4 x0 = *(const float4*)(p), x1 = *(const float4*)(p + 4), x2 = *(const float4*)(p + 1280), x3 = *(const float4*)(p + 1284);
            float4 x4 = *(const float4*)(p + 1920), x5 = *(const float4*)(p + 1924);
            a[0] = x0.x; a[1] = x0.y; a[2] = x0.z; a[3] = x0.w; a[4] = x1.x; a[5] = x1.y; a[6] = x1.z; a[7] = x1.w;
            b0[0] = x2.x; b0[1] = x2.y; b0[2] = x2.z; b0[3] = x2.w; b0[4] = x3.x; b0[5] = x3.y; b0[6] = x3.z; b0[7] = x3.w;
            b1[0] = x4.x; b1[1] = x4.y; b1[2] = x4.z; b1[3] = x4.w; b1[4] = x5.x; b1[5] = x5.y; b1[6] = x5.z; b1[7] = x5.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1[i], acc1, 0, 0, 0);
        }
        if (MODE >= 2) {
            float* q = lds + (cur ^ 1) * 3840 + threadIdx.x * 4;
            *(float4*)(q) = w; *(float4*)(q + 1024) = w; *(float4*)(q + 2048) = w;
        }
        if (MODE >= 1) __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE>
void run(int blocks, int iters) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("mode=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s  clock=%.0f MHz\n", MODE, blocks, iters, ms, flops / ms * 1e-9,
           (double)h[0] / (double)h[1] * 100.0);
}
int main() {
    for (int blocks : {256, 1024, 4096, 16384}) {
        int iters = 2048 * 1024 / blocks;    // same total work
        if (iters > 4096) iters = 4096;
        run<0>(blocks, iters); run<1>(blocks, iters); run<2>(blocks, iters); run<3>(blocks, iters);
    }
    // short tiles: many blocks with few K-steps each (the small bands)
    run<3>(65536, 4); run<3>(65536, 8); run<3>(32768, 37);
    return 0;
}
