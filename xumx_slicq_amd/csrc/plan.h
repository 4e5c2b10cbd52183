// Host-side plan object behind the C ABI (include/xumx_slicq_hip.h).
#pragma once
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <rocfft/rocfft.h>

#include "common.h"

namespace xsq {

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    size_t work_bytes = 0;
};

struct TileTable {
    TileDev* d_tiles = nullptr;
    int ntiles = 0;
};

struct BlockHost {
    int first_band, F, T;
    int64_t cum;  // sum of F*T over earlier blocks
};

}  // namespace xsq

struct xsq_plan {
    int L = 0, tr = 0, h = 0, nbins = 0, nbands = 0, nblocks = 0;
    int64_t sumFT = 0;  // complex coefficients per channel-slice (18640 for Bark-262)
    std::vector<xsq::BandDev> bands;
    std::vector<xsq::BlockHost> blocks;
    // device tables
    int band_radix4 = 1;            // 1: bands with Lg >= 24 (XSQ_D4_MIN_LG_DEFAULT) run on the radix-4 kernel (band_dft4.h), 0: all on the dense GEMM
    void* d_bands4 = nullptr;       // Band4Dev table of the eligible bands
    float* d_pool4f = nullptr;      // DFT_m matrices, twiddles, windows: analysis direction
    float* d_pool4i = nullptr;      //                                     synthesis direction
    int nbands4 = 0;
    std::vector<unsigned char> bands4_host;    // host copy of the Band4Dev table (the tile entries carry their band's descriptor)
    int64_t d4_max_block = 0;                  // largest F * Lg of a radix-4 band's arena block (buffer ranges of band_dft4.h)
    std::vector<int> bands4_m, bands4_small;   // host copies: m of each eligible band; indices of the other bands
    int fft_backend = 0;            // 0: hand-written LDS FFT when L == 18060, else rocFFT; 1: always rocFFT
    float2* d_T = nullptr;          // twiddles of the hand-written slice FFT: w1 (43*210) | w2 (14*15) | wl (L/2+1)
    unsigned short* d_tgt16 = nullptr;   // the same table as 16-bit bins (0xFFFF = none) + one pad entry: pairs of entries load as one dword
    int* d_tgt = nullptr;           // (sumFT) target bin per phase-ordered entry; null if bands of one phase overlap
    int phase_begin[5] = {0, 0, 0, 0, 0};
    // short bands (below the radix-4 split) synthesised inside k_slice_irfft (slice_fft.h: ShortSched); valid when short_n1 > 0
    int packed_fft = 0;             // 1: slice FFT codelets on v_pk_* (xsq_plan_set_packed_fft; only beside fp32 contractions)
    int short_inline = 0;           // 0 (default, faster as measured): short bands on the dense GEMM + Z round trip (band_synthesis_gemm)
    void *d_s_item1 = nullptr, *d_s_tw1 = nullptr;
    int *d_s_item2 = nullptr, *d_s_tgt = nullptr;
    float* d_s_wd = nullptr;
    int short_n1 = 0, short_n2 = 0, short_nent = 0, short_sc0 = 0;
    int short_begin[5] = {0, 0, 0, 0, 0};
    int phase_long[4] = {0, 0, 0, 0};   // first entry of each gather phase that belongs to a band NOT handled in-kernel
    float* d_tw = nullptr;          // (L) slice window
    float* d_Wf = nullptr;          // per-band analysis matrices  (window, sign, 1/Lg folded in)
    float* d_Wi = nullptr;          // per-band synthesis matrices (dual window, Lg, sign, 1/L folded in)
    xsq::BandDev* d_bands = nullptr;
    int* d_cov_ptr = nullptr;       // (nbins+1) CSR over spectrum bins -> covering bands
    int* d_cov_band = nullptr;
    // caches keyed by the call shape
    std::mutex mu;
    std::map<std::pair<int, int>, xsq::FftPlan> fft;              // (direction, batch)
    std::map<std::tuple<int, int, int>, xsq::TileTable> tiles;   // (kind, rows, 0)
};
