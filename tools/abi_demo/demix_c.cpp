// The drop-in boundary without Python or PyTorch: a plain C++ host program that links libxumx_slicq_hip.so, uploads a
// plan and a model through the C ABI (include/xumx_slicq_hip.h), demixes a track with ONE call (xsq_separator_forward:
// Separator.forward of /root/reference/xumx_slicq_v2/separator.py:133-232) and writes the four stems.  Device memory comes
// from hipMalloc -- the library only ever sees raw pointers, sizes and a stream.
//   build: hipcc -O2 -I include tools/abi_demo/demix_c.cpp -L xumx_slicq_amd -lxumx_slicq_hip -Wl,-rpath,$PWD/xumx_slicq_amd -o demix_c
//   run:   demix_c plan.bin model.bin audio.bin stems.bin [chunk_size] [wiener 0|1]
// File formats (little endian, written by tests/test_abi_c_gpu.py):
//   plan.bin   int32 L, tr, nbands | int32 Lg[nbands] | int32 c[nbands] | float g[sum Lg] | double gd[sum Lg] | float tw[L]
//   model.bin  int32 nblocks, causal | int32 F[nblocks] | int32 T[nblocks] | int64 nparams | float params[nparams]
//   audio.bin  int32 nb | int64 N | float audio[nb][2][N]
//   stems.bin  float stems[4][nb][2][N]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "xumx_slicq_hip.h"

#define CHECK(expr)                                                                             \
    do {                                                                                        \
        int rc_ = (expr);                                                                       \
        if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, xsq_last_error()); return 2; } \
    } while (0)
#define HIP(expr)                                                                               \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #expr, hipGetErrorString(e_)); return 3; } \
    } while (0)

template <class T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s plan.bin model.bin audio.bin stems.bin [chunk_size] [wiener]\n", argv[0]); return 1; }
    const int64_t chunk = argc > 5 ? atoll(argv[5]) : 2621440;
    const int wiener = argc > 6 ? atoi(argv[6]) : 0;

    // ---- plan ----------------------------------------------------------------------------------------------
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[3];
    if (!rd(f, hdr, 3)) return 1;
    const int L = hdr[0], tr = hdr[1], nbands = hdr[2];
    std::vector<int32_t> Lg(nbands), c(nbands);
    if (!rd(f, Lg.data(), nbands) || !rd(f, c.data(), nbands)) return 1;
    size_t sumLg = 0;
    for (int v : Lg) sumLg += v;
    std::vector<float> g(sumLg), tw(L);
    std::vector<double> gd(sumLg);
    if (!rd(f, g.data(), sumLg) || !rd(f, gd.data(), sumLg) || !rd(f, tw.data(), L)) return 1;
    fclose(f);
    xsq_plan* plan = nullptr;
    CHECK(xsq_plan_create(&plan, L, tr, nbands, Lg.data(), c.data(), g.data(), gd.data(), tw.data()));

    // ---- model ---------------------------------------------------------------------------------------------
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    int32_t mh[2];
    if (!rd(f, mh, 2)) return 1;
    const int nblocks = mh[0], causal = mh[1];
    std::vector<int32_t> F(nblocks), T(nblocks);
    int64_t nparams = 0;
    if (!rd(f, F.data(), nblocks) || !rd(f, T.data(), nblocks) || !rd(f, &nparams, 1)) return 1;
    std::vector<float> params((size_t)nparams);
    if (!rd(f, params.data(), (size_t)nparams)) return 1;
    fclose(f);
    if (xsq_model_num_params(nblocks, F.data(), T.data()) != nparams) { fprintf(stderr, "model.bin: parameter count mismatch\n"); return 1; }
    xsq_model* model = nullptr;
    CHECK(xsq_model_create(&model, nblocks, F.data(), T.data(), causal, params.data(), nparams));

    // ---- audio ---------------------------------------------------------------------------------------------
    f = fopen(argv[3], "rb");
    if (!f) { perror(argv[3]); return 1; }
    int32_t nb = 0;
    int64_t N = 0;
    if (!rd(f, &nb, 1) || !rd(f, &N, 1)) return 1;
    std::vector<float> audio((size_t)nb * 2 * N);
    if (!rd(f, audio.data(), audio.size())) return 1;
    fclose(f);

    // ---- one call -------------------------------------------------------------------------------------------
    xsq_demixer* dmx = nullptr;
    CHECK(xsq_demixer_create(&dmx, plan));
    size_t main_bytes = 0, tail_bytes = 0;
    CHECK(xsq_separator_workspace(dmx, model, nb, N, chunk, 8, wiener, &main_bytes, &tail_bytes));
    float *d_audio = nullptr, *d_out = nullptr;
    void *ws = nullptr, *wt = nullptr;
    HIP(hipMalloc(&d_audio, audio.size() * 4));
    HIP(hipMalloc(&d_out, audio.size() * 4 * 4));
    HIP(hipMalloc(&ws, main_bytes));
    if (tail_bytes) HIP(hipMalloc(&wt, tail_bytes));
    HIP(hipMemcpy(d_audio, audio.data(), audio.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s_main, s_tail;
    HIP(hipStreamCreate(&s_main));
    HIP(hipStreamCreate(&s_tail));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {                 // the third call is the timed one (tables are built by the first)
        HIP(hipEventRecord(e0, s_main));
        CHECK(xsq_separator_forward(dmx, model, d_audio, nb, N, chunk, 8, wiener, 1, d_out, ws, main_bytes, wt, tail_bytes, s_main, s_tail));
        HIP(hipEventRecord(e1, s_main));
        HIP(hipStreamSynchronize(s_main));
        HIP(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<float> stems(audio.size() * 4);
    HIP(hipMemcpy(stems.data(), d_out, stems.size() * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 1; }
    fwrite(stems.data(), 4, stems.size(), f);
    fclose(f);
    printf("demix_c: nb=%d N=%lld chunk=%lld wiener=%d: %.3f ms on the device = %.0f x real time (workspace %.2f + %.2f GB)\n", nb,
           (long long)N, (long long)chunk, wiener, ms, (double)N / 44100.0 / (ms * 1e-3), main_bytes / 1e9, tail_bytes / 1e9);
    xsq_demixer_destroy(dmx);
    xsq_model_destroy(model);
    xsq_plan_destroy(plan);
    return 0;
}
