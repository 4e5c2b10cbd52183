// Weight gradients of the CDAE layers as grouped "TN" GEMMs on the matrix cores (gfx950).
//
//   C[m][n] = sum_k A[k][m] * B[k][n]        k = one output position (b, f, t) of the layer
//   A[k][.] = the 52 channels-last values of that position (activation or its gradient)
//   B[k][.] = the receptive-field patch of that position, a few contiguous spans in memory
//
// The contraction index k is the SLOW axis of both operands (rows of the channels-last arrays), so the
// forward engine (gemm_tile.h, K-contiguous operands) does not fit.  Here a K-step of 16 rows is
// staged k-major in LDS -- As[16][64], Bs[16][NTL] -- and every v_mfma_f32_32x32x2_f32 operand is one
// ds_read_b32 per lane (lane l takes column l%32 of row 2*kp + l/32: 32 consecutive floats per half
// wave, conflict-free).  M is the channel count (52 of 64 used), so each B element feeds 64 MACs and
// LDS traffic is no concern; the kernel lives on L2-resident re-reads of the activation arrays.
//
// One workgroup = (group, column tile, chunk of KC rows); it writes its 64 x NTL partial tile, and
// k_wgrad_reduce adds the chunks of a group in order (deterministic) while scattering into the
// canonical (state_dict) gradient layout.  4 waves: wave w owns column blocks w*NBW .. w*NBW+NBW-1
// (32 columns each) x both 32-row blocks.
//
// BF16 = true (the "bf16" training arm, xsq_train_set_precision mode 1: torch.autocast runs the convolutions' weight
// gradients on bf16 operands as well, training.py:473-476): the same staging; a lane reads its eight k of a K-step
// (k = 8 (l / 32) + j), rounds pairs to bf16 (v_cvt_pk_bf16_f32, nearest even) and issues ONE v_mfma_f32_32x32x16_bf16
// per (row block, column block) and K-step where the fp32 form issues eight v_mfma_f32_32x32x2_f32 -- 1/16 of the
// matrix-pipe cycles; the loop is then bound by its operand staging.
#pragma once
#include "gemm_tile.h"
#include "gemm_tile_bf6.h"

namespace xsq {

struct WgTile { int group, ntile, k0, k1; };
struct WgGroupInfo { int tile_base, nch; };

template <class Op, bool BF16 = false>
__global__ __launch_bounds__(256) void wgrad_kernel(Op op, const WgTile* __restrict__ tiles, float* __restrict__ partial) {
    constexpr int NTL = Op::NTL, NB = NTL / 32, NBW = (NB + 3) / 4, NQ = (NTL / 4 + 15) / 16;
    const WgTile t = tiles[blockIdx.x];
    const typename Op::Group g = op.group(t.group);
    __shared__ float As[2][16][64];
    __shared__ float Bs[2][16][NTL];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = tid >> 4, q = tid & 15;
    typename Op::Cols cols[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) cols[i] = op.cols(g, t.ntile, 4 * (q + 16 * i));
    f32x16 acc[2][NBW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra, rb[NQ];
    auto gload = [&](int kbase) {
        const int k = kbase + kk;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        ra = z;
#pragma unroll
        for (int i = 0; i < NQ; ++i) rb[i] = z;
        if (k < t.k1) {
            if (q < 13) ra = *reinterpret_cast<const float4*>(op.a_row(g, k) + 4 * q);
            const typename Op::Row row = op.row(g, k, t.ntile);
#pragma unroll
            for (int i = 0; i < NQ; ++i) rb[i] = op.load_b4(g, row, cols[i]);
        }
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][kk][4 * q]) = ra;
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (4 * (q + 16 * i) < NTL) *reinterpret_cast<float4*>(&Bs[buf][kk][4 * (q + 16 * i)]) = rb[i];
    };
    gload(t.k0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    const int l32 = lane & 31, lk = lane >> 5;
    for (int kb = t.k0; kb < t.k1; kb += 16) {
        const bool more = kb + 16 < t.k1;
        if (more) gload(kb + 16);
        if constexpr (BF16) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto frag_a = [&](int m) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = As[buf][8 * lk + j][m];
                const u32x4 w = {bf16_rne2(v[0], v[1]), bf16_rne2(v[2], v[3]), bf16_rne2(v[4], v[5]), bf16_rne2(v[6], v[7])};
                return __builtin_bit_cast(bf16x8_t, w);
            };
            auto frag_b = [&](int n) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = Bs[buf][8 * lk + j][n];
                const u32x4 w = {bf16_rne2(v[0], v[1]), bf16_rne2(v[2], v[3]), bf16_rne2(v[4], v[5]), bf16_rne2(v[6], v[7])};
                return __builtin_bit_cast(bf16x8_t, w);
            };
            const bf16x8_t a0 = frag_a(l32), a1 = frag_a(32 + l32);
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int nb = wave * NBW + j;
                if (nb < NB) {          // wave-uniform
                    const bf16x8_t b = frag_b(nb * 32 + l32);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                const int k = 2 * kp + lk;
                const float a0 = As[buf][k][l32], a1 = As[buf][k][32 + l32];
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int nb = wave * NBW + j;
                    if (nb < NB) {          // wave-uniform
                        const float b = Bs[buf][k][nb * 32 + l32];
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][j], 0, 0, 0);
                    }
                }
            }
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = partial + (int64_t)blockIdx.x * 64 * NTL;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int nb = wave * NBW + j;
            if (nb < NB) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(int64_t)(i * 32 + acc_row(r) + 4 * lk) * NTL + nb * 32 + l32] = acc[i][j][r];
            }
        }
}

}  // namespace xsq
