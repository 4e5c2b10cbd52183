// Probe: what a gfx950 raw buffer access checks against the descriptor's num_records.
// Finding (MI355X, ROCm 7.2): see the output -- the scalar offset IS / IS NOT part of the range check.
//   hipcc -O2 --offload-arch=gfx950 tools/probe/buf_range.hip -o /tmp/buf_range && /tmp/buf_range
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, float* out, int records, int voff, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, records, 0x00020000);
    if (threadIdx.x == 0) {
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 777.f), r, voff, soff, 0);
    }
}
int main() {
    float *p, *o; hipMalloc(&p, 4096); hipMalloc(&o, 16);
    float h[1024];
    int bad = 0;
    struct { int rec, vo, so; const char* what; } cases[] = {
        {64, 0, 0, "in range"}, {64, 128, 0, "voffset past the range"}, {64, 0, 128, "scalar offset past the range"},
        {64, 32, 48, "voffset + scalar offset past the range, each inside"}, {64, 60, 0, "last dword"}, {64, 64, 0, "first dword past"}};
    for (auto& c : cases) {
        for (int i = 0; i < 1024; ++i) h[i] = (float)i;
        hipMemcpy(p, h, 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, c.rec, c.vo, c.so);
        float got; hipMemcpy(&got, o, 4, hipMemcpyDeviceToHost); hipMemcpy(h, p, 4096, hipMemcpyDeviceToHost);
        const int idx = (c.vo + c.so) / 4;
        printf("num_records %d, voffset %d, soffset %d (%s): load returned %g (memory holds %d), store %s\n", c.rec, c.vo, c.so, c.what,
               got, idx, h[idx] == 777.f ? "WRITTEN" : "dropped");
        // what common.h's buf_ld* / buf_st* rely on (cdae.hip epilogues, band_dft4.h): voffset + soffset is checked
        // against num_records as ONE sum; inside -> the access happens, past it -> the load returns 0 and the store is dropped
        const bool inside = c.vo + c.so + 4 <= c.rec;
        const bool ok = inside ? (got == (float)idx && h[idx] == 777.f) : (got == 0.f && h[idx] == (float)idx);
        if (!ok) { printf("UNEXPECTED\n"); ++bad; }
    }
    printf("%s\n", bad ? "FAILED" : "ALL AS RELIED ON");
    return bad ? 1 : 0;
}
