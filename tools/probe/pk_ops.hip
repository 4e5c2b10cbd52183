// Probe: operand-select / negate semantics of the packed-fp32 VOP3P instructions on gfx950, as used by the
// hand-placed v_pk_* codelets of csrc/slice_fft.h.  Build + run:  hipcc --offload-arch=gfx950 pk_ops.hip -o pk_ops && ./pk_ops
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out) {
    v2f a = {in[0], in[1]}, b = {in[2], in[3]}, c = {in[4], in[5]}, d;
    v2f cs = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in[6]))),
              __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in[7])))};
    int o = 0;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "s"(cs), "v"(c)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(cs), "v"(c)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(cs), "v"(c)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[o++] = d.x; out[o++] = d.y;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[o++] = d.x; out[o++] = d.y;                       // (a.x b.x, a.x b.y)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); out[o++] = d.x; out[o++] = d.y;   // (-a.y b.y + c.x, a.y b.x + c.y)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); out[o++] = d.x; out[o++] = d.y;          // (a.x b.x, -a.x b.y)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); out[o++] = d.x; out[o++] = d.y;                  // (a.y b.y + c.x, a.y b.x + c.y)
}
int main() {
    float h[8] = {2.f, 3.f, 5.f, 7.f, 11.f, 13.f, 17.f, 19.f}, *din, *dout, r[20];
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(r));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    const float a0 = 2, a1 = 3, b0 = 5, b1 = 7, c0 = 11, c1 = 13, s0 = 17, s1 = 19;
    const float want[20] = {a0 * s0 + c0, a1 * s0 + c1, a0 * s1 + c0, a1 * s1 + c1, -a0 * s1 + c0, -a1 * s1 + c1,
                            a0 - b1, a1 + b0, a0 + b1, a1 - b0, a0 - b0, a1 - b1, a0 * b0, a0 * b1,
                            -a1 * b1 + c0, a1 * b0 + c1, a0 * b0, -a0 * b1, a1 * b1 + c0, a1 * b0 + c1};
    int bad = 0;
    for (int i = 0; i < 20; ++i) { if (r[i] != want[i]) { ++bad; printf("MISMATCH %d: got %g want %g\n", i, r[i], want[i]); } }
    printf(bad ? "pk_ops: %d mismatches\n" : "pk_ops: all 20 values as expected\n", bad);
    return bad != 0;
}
