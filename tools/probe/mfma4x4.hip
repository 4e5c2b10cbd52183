// Operand / result lane maps of v_mfma_f32_4x4x1_16B_f32 on gfx950 (not in the programming guides): A carries a
// tag of (lane), B a tag of (lane); D tells which (A lane, B lane) pair each (lane, register) multiplied.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // a = 1 + lane, b = 1000^? : use primes so that product identifies the pair: a = lane + 1, b = 100 * (lane + 1)
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 100.f * (float)(l + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            // product = (la+1) * 100 * (lb+1): find la, lb in the same block guess by brute force
            int fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb) if ((float)(la + 1) * 100.f * (float)(lb + 1) == h[l * 4 + r] && la / 4 == lb / 4) { fa = la; fb = lb; break; }
            printf("  r%d = A[lane %2d] x B[lane %2d]", r, fa, fb);
        }
        printf("\n");
        if (l == 7) l = 31 - 1;
        if (l == 35) break;
    }
    return 0;
}
