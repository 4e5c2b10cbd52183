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

// A/B ARM (XSQ_TRAIN_WGRAD_PACKED=1; bitwise the default's results, measured 25-35 % SLOWER per launch: profiles/r11_ab_runs.txt r11wg).
// The bf16 arm with its K-step staged TRANSPOSED AND PACKED (round 6; DESIGN.md section 8 lead of round 5).  wgrad_kernel<Op, true>
// reads every MFMA operand as eight 4-byte LDS words and converts them lane by lane: per K-step and wave 16 + 8 NBW ds_read_b32
// and 4 (2 + NBW) conversions beside 2 NBW MFMAs of 32 cycles -- an LDS-instruction loop with some matrix work attached.  Here the
// rounding to bf16 (the same v_cvt_pk_bf16_f32 of the same values: bitwise the same operands) happens ONCE, at staging time, and
// LDS holds, per column and half K-step, the eight bf16 of rows 8 h .. 8 h + 7 as ONE 16-byte word: a fragment is one
// ds_read_b128, 2 + NBW of them per K-step.  Staging: thread (quarter h2 = tid >> 6: rows 4 h2 .. 4 h2 + 3 of the K-step, column
// quad cq = tid & 63) loads four consecutive rows' float4, packs the two row pairs of each of its four columns and writes four
// 8-byte words; the operator walks its rows with an iterator (one (b, f, t) split per K-step, three increments) instead of a split
// per row.
template <class Op>
__global__ __launch_bounds__(256) void wgrad_bf16p_kernel(Op op, const WgTile* __restrict__ tiles, float* __restrict__ partial) {
    constexpr int NTL = Op::NTL, NB = NTL / 32, NBW = (NB + 3) / 4, NCQ = NTL / 4;
    static_assert(NCQ <= 64, "one column quad per thread of a 64-thread staging quarter");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const WgTile t = tiles[blockIdx.x];
    const typename Op::Group g = op.group(t.group);
    // [buffer][half K-step][slot of the column][row pair]; column n sits in slot (n & 3) * QP + (n >> 2): the four columns of a
    // staging thread's quad land QP slots apart, consecutive quads in consecutive slots -- the 8-byte staging writes of a wave
    // walk the banks instead of putting 16 lanes on four of them, and a fragment read's four column phases start 128 bytes apart
    constexpr int QPA = 16 + 8, QPB = NCQ + 8;
    __shared__ __attribute__((aligned(16))) unsigned Ap[2][2][4 * QPA][4];
    __shared__ __attribute__((aligned(16))) unsigned Bp[2][2][4 * QPB][4];
    auto slot_a = [&](int n) { return (n & 3) * QPA + (n >> 2); };
    auto slot_b = [&](int n) { return (n & 3) * QPB + (n >> 2); };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int h2 = tid >> 6, cq = tid & 63;
    const bool b_on = cq < NCQ, a_on = cq < 13, a_pad = cq >= 13 && cq < 16;       // A: 52 channels = 13 quads, 3 quads of zeros
    const typename Op::Cols cols = op.cols(g, t.ntile, 4 * (b_on ? cq : 0));
    f32x16 acc[2][NBW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    auto gload = [&](int kbase) {
        const int k = kbase + 4 * h2;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        typename Op::RowIt it = op.row_it(g, k < t.k1 ? k : t.k0, t.ntile);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = k + j < t.k1;
            ra[j] = (live && a_on) ? *reinterpret_cast<const float4*>(op.a_row(g, k + j) + 4 * cq) : z;
            rb[j] = (live && b_on) ? op.load_b4(g, op.row_of(it), cols) : z;
            if (j < 3) op.advance(g, it);
        }
    };
    auto lstore = [&](int buf) {
        const int half = h2 >> 1, w0 = 2 * (h2 & 1);
        const float va[4][4] = {{ra[0].x, ra[0].y, ra[0].z, ra[0].w}, {ra[1].x, ra[1].y, ra[1].z, ra[1].w},
                                {ra[2].x, ra[2].y, ra[2].z, ra[2].w}, {ra[3].x, ra[3].y, ra[3].z, ra[3].w}};
        const float vb[4][4] = {{rb[0].x, rb[0].y, rb[0].z, rb[0].w}, {rb[1].x, rb[1].y, rb[1].z, rb[1].w},
                                {rb[2].x, rb[2].y, rb[2].z, rb[2].w}, {rb[3].x, rb[3].y, rb[3].z, rb[3].w}};
        if (a_on || a_pad) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<uint2*>(&Ap[buf][half][c * QPA + cq][w0]) = make_uint2(bf16_rne2(va[0][c], va[1][c]), bf16_rne2(va[2][c], va[3][c]));
        }
        if (b_on) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<uint2*>(&Bp[buf][half][c * QPB + cq][w0]) = make_uint2(bf16_rne2(vb[0][c], vb[1][c]), bf16_rne2(vb[2][c], vb[3][c]));
        }
    };
    gload(t.k0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    const int l32 = lane & 31, lk = lane >> 5;
    for (int kb = t.k0; kb < t.k1; kb += 16) {
        const bool more = kb + 16 < t.k1;
        if (more) gload(kb + 16);
        const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(&Ap[buf][lk][slot_a(l32)][0]));
        const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(&Ap[buf][lk][slot_a(32 + l32)][0]));
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int nb = wave * NBW + j;
            if (nb < NB) {          // wave-uniform
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(&Bp[buf][lk][slot_b(nb * 32 + l32)][0]));
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][j], 0, 0, 0);
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
