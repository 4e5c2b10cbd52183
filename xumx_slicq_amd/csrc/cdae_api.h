// Shared declarations of the CDAE kernels (csrc/cdae.hip) for the training step (csrc/train.hip).
#pragma once
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"
#include "plan.h"

namespace xsq {

static const int H1 = 50, H2 = 51, CS = 52;  // hidden sizes (model.py:92-93), padded channel stride
static const int NT = 4;                      // targets

struct CdaeBlockDev {
    int F, T, hop, kf, F1, F2;
    int cumF1, cumF2;     // sums over earlier blocks of F1, F2 (activation arena offsets)
    int ld1, ld4;         // row lengths of the transposed layer-1 / layer-4 matrices (K padded to 16)
    int64_t cum;          // sum over earlier blocks of F*T
    int64_t cumF;         // sum over earlier blocks of F (input_mean / input_scale offset)
    int64_t w1[NT], w2[NT], w3[NT], w4[NT];   // float offsets of the folded weight matrices
    int64_t s1[NT], s2[NT], s3[NT], b4[NT];   // float offsets of shift vectors (64) / output bias (2)
    int64_t u2[NT], u3[NT];                   // Winograd F(2, 4) transformed weights of layers 2 / 3 (offsets into xsq_model::d_upool; cdae_wino.h: kf x WN_UDF floats)
    int64_t u1[NT];                           // F(2, 2)-along-the-hop weights of layer 1 (W0 | W0 + W1 | W1 per chunk of 16 k; cdae_l1f.h: nch1 x LF_U16 floats)
    int nch1, pad_;                           // chunks of 16 k of that form: l1f_chunks(kf, hop)
    int64_t u4[NT];                           // F(2, 2)-along-the-hop weights of layer 4 (Wb | Wa + Wb | Wa per frequency tap and column tile; cdae_l4f.h)
    int64_t uq2[NT], uq3[NT];                 // Winograd F(4, 4) transformed weights of layers 2 / 3 (cdae_wino4.h: kf x W4_UDF floats), -1: not built
};

}  // namespace xsq

struct xsq_model {
    int causal = 0;
    int precision = 0;             // 0 fp32 MFMA, 1 split-bf16 MFMA (xsq_model_set_precision)
    int winograd = 7;              // fast-convolution forms of the fp32 inference layers (xsq_model_set_winograd), a bit mask: 1 = layers 2 / 3 as Winograd
                                   // F(2, 4) along the time taps (cdae_wino.h), 2 / 4 = layer 1 / layer 4 as F(2, 2) along the hop (cdae_l1f.h, cdae_l4f.h),
                                   // 8 = layers 2 / 3 as F(4, 4) where the rows are long enough (cdae_wino4.h; needs bit 1); 0 = the direct kernels
    bool wino4 = false;            // the F(4, 4) weights of layers 2 / 3 exist (XSQ_WINO4=1 at xsq_model_create; cdae_wino4.h)
    int nblocks = 0;
    int64_t sumFT = 0;             // complex coefficients per channel-slice
    std::vector<xsq::BlockHost> table;
    std::vector<xsq::CdaeBlockDev> blocks;
    int64_t sumF = 0, sumF1 = 0, sumF2 = 0;
    xsq::CdaeBlockDev* d_blocks = nullptr;
    float* d_pool = nullptr;       // all folded weights / shifts
    float* d_pool_split = nullptr; // the same pool as (bf16 hi << 16) | bf16 lo words (precision 1)
    float* d_upool = nullptr;      // Winograd-transformed weights of layers 2 / 3 (cdae_wino.h): their own pool -- the training step
                                   // regathers d_pool from the canonical parameters every step and never contracts with these
    int64_t pool_floats = 0;
    float* d_mean = nullptr;       // (sumF) input_mean  (stored as -mean by the reference)
    float* d_scale = nullptr;      // (sumF) input_scale (stored as 1/std)
    int64_t* d_cum = nullptr;      // (nblocks+1) cumulative F*T, for the elementwise kernels
    int* d_blockF = nullptr;       // (nblocks)
    std::mutex mu;
    std::map<std::tuple<int, int, int>, xsq::TileTable> tiles;   // (layer, B, S)
};

namespace xsq {

struct CdaeArgs {
    const CdaeBlockDev* blocks;
    const float* pool;
    const float* xin;     // whitened magnitude, arena layout (2B channels, real)
    float* act1;          // (block, target, B, F1, T1, 52)
    float* act2;          // (block, target, B, F2, T2, 52)
    float* act3;          // (block, target, B, F1, T1, 52)
    const float* X;       // mix coefficients (complex arena, 2B channels)
    float* Y;             // estimates (complex arena, 8B channels, targets first)
    float* masks;         // optional real arena (8B channels), nullptr to skip
    int Bn, S, T1, T2, causal;
    int raw;              // 1: layers 1-3 store the raw convolution (training: BatchNorm on batch statistics follows)
    // training only (csrc/train.hip); nullptr on the inference path
    const float* xin8;    // layer-1 operator as the DATA GRADIENT of layer 4: input = real arena with 8B channels
                          // (target-specific), no left padding, right edge checked; weights = w4 in w1's layout
    float* gx8;           // layer-4 operator as the DATA GRADIENT of layer 1: raw store of (target, b, c, f, u*hop+dt)
                          // with u < T1+1 (padded coordinates s = tau + left pad); weights = w1 in w4's layout
    // split-bf16 inference (xsq_model_set_precision 1, gemm_tile_bf3.h): xin / act1..3 hold one 32-bit word
    // per value, (bf16 hi << 16) | bf16 lo, written by the producing kernel; poolB = the weight pool in the
    // same format (shifts / biases are still read from `pool`)
    int split = 0;
    const float* poolB = nullptr;
    const float* upool = nullptr;  // Winograd-transformed weights of layers 2 / 3 (xsq_model::d_upool; inference only)
};


// one grouped launch of CDAE layer 1..4 over all blocks x targets (tile table cached per (layer, B, S))
int cdae_launch_layer(xsq_model* Mo, int layer, const CdaeArgs& a, hipStream_t stream, const char* prof_name = nullptr);
// |X| -> whitened magnitude with explicit mean / scale tables (sum_b F_b floats each)
int cdae_launch_magnitude(const xsq_model* Mo, const float* X, float* xin, const float* mean, const float* scale,
                          int Bn, int S, hipStream_t stream, int split = 0);

// backward of the Wiener-EM iteration (wiener.hip), in place on the gradient arena G.  The pre-filter estimate is
// either given (Y0, complex arena) or formed from the masks while loading (Y0 == nullptr, masks = real arena).
// gM (optional, masks form only): the mask gradient is formed in the last pass -- gM = (Re(conj(x) dL/dy0) + gM) m (1 - m),
// the training step's k_mask_bwd -- and G is then NOT written.
int wiener_em_backward(int nblocks, const int32_t* F, const int32_t* T, const float* X, const float* Y0, const float* masks,
                       float* G, int Bn, int S, int win_len, int batch_group, const void* stats, void* bstats, hipStream_t stream,
                       float* gM = nullptr);
// loss forward (loss.hip), optionally with the gradients of both terms written in the same pass (gY / gM non-null)
int loss_forward_backward(int nblocks, const int32_t* F, const int32_t* T, const float* pred, const float* target,
                          const float* masks, int Bn, int S, double* out, float* gY, float* gM, void* ws, hipStream_t stream);

}  // namespace xsq
