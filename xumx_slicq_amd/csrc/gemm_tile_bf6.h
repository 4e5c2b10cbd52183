// "bf16x6": fp32-grade contraction on the bf16 matrix pipe.
//
// Every fp32 operand is cut EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (8 + 8 + 8 significand bits, by
// truncation: x1 = top 16 bits of x, x2 = top 16 bits of x - x1, x3 = x - x1 - x2, all three subtractions exact),
// and a product is accumulated as the six partial products of weight >= 2^-16:
//        a*b  ~=  a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1          (each exact in the MFMA's fp32 accumulator)
// The dropped terms a2 b3 + a3 b2 + a3 b3 are <= 2^-23 |ab| -- the size of ONE fp32 rounding -- so the result
// differs from an fp32 FMA chain by what a different summation order would change.  Six v_mfma_f32_32x32x16_bf16
// replace sixteen v_mfma_f32_32x32x2_f32: 3/8 of the matrix-pipe time (xsq_model_set_precision mode 2).
//
// Same pipeline and tile tables as gemm_tile.h; operands are plain fp32 in memory and are cut between the
// global load and the LDS write (11 VALU ops per two values).  LDS image of a K-step: per tile row
// [16 x x1 | 16 x x2 | 16 x x3] = 96 B at a stride of 28 words (conflict-free ds_read_b128).
#pragma once
#include "gemm_tile_bf3.h"

namespace xsq {

// two values -> three words, each (piece of x in the low half, piece of y in the high half)
__device__ __forceinline__ void bf6_cut2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __builtin_bit_cast(unsigned, x), uy = __builtin_bit_cast(unsigned, y);
    const float rx = x - __builtin_bit_cast(float, ux & 0xffff0000u), ry = y - __builtin_bit_cast(float, uy & 0xffff0000u);
    const unsigned vx = __builtin_bit_cast(unsigned, rx), vy = __builtin_bit_cast(unsigned, ry);
    const float sx = rx - __builtin_bit_cast(float, vx & 0xffff0000u), sy = ry - __builtin_bit_cast(float, vy & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sy), __builtin_bit_cast(unsigned, sx), 0x07060302u);
}

// two values -> one word of two bf16, rounded to nearest even: what torch.autocast(dtype=bfloat16) makes of a conv operand
// (/root/reference/xumx_slicq_v2/training.py:473-476).
__device__ __forceinline__ unsigned bf16_rne2(float x, float y) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));      // v_cvt_pk_bf16_f32 (gfx950): one instruction per pair
}

// PLAIN = true: the "bf16" training arm (xsq_train_set_precision mode 1, BASELINE configs[4] as written): both operands
// rounded ONCE to bf16 (round to nearest even) between the global load and the LDS write, ONE v_mfma_f32_32x32x16_bf16 per
// 16-value K-step and 32 x 32 block, fp32 accumulation -- the arithmetic of the reference's bf16 autocast convolutions.
template <class Op, bool PLAIN = false>
__global__ __launch_bounds__(256) void grouped_gemm_bf6_kernel(Op op, const TileDev* __restrict__ tiles, int ntiles) {
    static_assert(!aux_of<Op>::on, "operators with an aux stream run on the fp32 engine");
#ifndef XSQ_BF6_LD
#define XSQ_BF6_LD 28
#endif
    constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = 16, LD = XSQ_BF6_LD;
    constexpr int RA = BM / 64;

    __shared__ __attribute__((aligned(16))) unsigned lds[2 * (BM + BN) * LD];
    unsigned* const As0 = lds;
    unsigned* const Bs0 = lds + 2 * BM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const bool wide = t.narrow == 0;          // wave-uniform
    const typename Op::Group g = op.group(t.group);
    const int K = g.K;

    const int s_row = tid >> 2;          // 0..63
    const int s_kq = (tid & 3) * 4;      // 0,4,8,12
    typename Op::RowA ra[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = op.row_a(g, t.m0 + s_row + 64 * i);
    // (weights pre-cut on the device -- 24 B per four values, three 8-byte loads -- measured no faster than cutting
    // them here from one 16-byte load)
    const float* bp = g.B + (int64_t)(t.n0 + s_row) * g.ldb + s_kq;   // Bt[n][k]
    const bool b_on = wide || s_row < 32;

    float4 ga[2][RA];
    float4 gb[2];
    auto load_set = [&](int set, int k) {
        if (k < K) {
#pragma unroll
            for (int i = 0; i < RA; ++i) ga[set][i] = op.load_a4(g, ra[i], k + s_kq);
            if (b_on) gb[set] = *reinterpret_cast<const float4*>(bp + k);
        }
    };
    auto put = [&](unsigned* row, const float4& v) {       // plane p: words [8p + 2q, 8p + 2q + 1], q = s_kq/4
        if constexpr (PLAIN) {
            *reinterpret_cast<uint2*>(row + (s_kq >> 1)) = make_uint2(bf16_rne2(v.x, v.y), bf16_rne2(v.z, v.w));
            return;
        }
        unsigned a1, a2, a3, b1, b2, b3;
        bf6_cut2(v.x, v.y, a1, a2, a3);
        bf6_cut2(v.z, v.w, b1, b2, b3);
        *reinterpret_cast<uint2*>(row + (s_kq >> 1)) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(row + 8 + (s_kq >> 1)) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(row + 16 + (s_kq >> 1)) = make_uint2(a3, b3);
    };
    auto store_set = [&](int set, int buf) {
        unsigned* Aw = As0 + buf * BM * LD;
        unsigned* Bw = Bs0 + buf * BN * LD;
#pragma unroll
        for (int i = 0; i < RA; ++i) put(&Aw[(s_row + 64 * i) * LD], ga[set][i]);
        put(&Bw[s_row * LD], gb[set]);
    };
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[0][i] = ga[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[0] = gb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_set(0, 0);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int lrow = lane & 31, lk = lane >> 5;
    const int a_frag = (wave * 32 + lrow) * LD + 4 * lk;
    const int b_frag = lrow * LD + 4 * lk;

    auto frag = [&](const unsigned* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p)); };
#define XSQ_MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
    auto mfma_step = [&](int buf) {
        const unsigned* As = As0 + buf * BM * LD;
        const unsigned* Bs = Bs0 + buf * BN * LD;
        if constexpr (PLAIN) {
            const bf16x8_t a1 = frag(&As[a_frag]), p1 = frag(&Bs[b_frag]);
            XSQ_MF(a1, p1, acc0);
            if (wide) { const bf16x8_t q1 = frag(&Bs[b_frag + 32 * LD]); XSQ_MF(a1, q1, acc1); }
            return;
        }
        const bf16x8_t a1 = frag(&As[a_frag]), a2 = frag(&As[a_frag + 8]), a3 = frag(&As[a_frag + 16]);
        const bf16x8_t p1 = frag(&Bs[b_frag]), p2 = frag(&Bs[b_frag + 8]), p3 = frag(&Bs[b_frag + 16]);
        if (wide) {
            const bf16x8_t q1 = frag(&Bs[b_frag + 32 * LD]), q2 = frag(&Bs[b_frag + 32 * LD + 8]), q3 = frag(&Bs[b_frag + 32 * LD + 16]);
            XSQ_MF(a1, p3, acc0); XSQ_MF(a1, q3, acc1);      // smallest terms first
            XSQ_MF(a3, p1, acc0); XSQ_MF(a3, q1, acc1);
            XSQ_MF(a2, p2, acc0); XSQ_MF(a2, q2, acc1);
            XSQ_MF(a1, p2, acc0); XSQ_MF(a1, q2, acc1);
            XSQ_MF(a2, p1, acc0); XSQ_MF(a2, q1, acc1);
            XSQ_MF(a1, p1, acc0); XSQ_MF(a1, q1, acc1);
        } else {
            XSQ_MF(a1, p3, acc0); XSQ_MF(a3, p1, acc0); XSQ_MF(a2, p2, acc0);
            XSQ_MF(a1, p2, acc0); XSQ_MF(a2, p1, acc0); XSQ_MF(a1, p1, acc0);
        }
    };
#undef XSQ_MF

    store_set(0, 0);
    load_set(1, BK);
    load_set(0, 2 * BK);
    __syncthreads();

    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
        mfma_step(0);
        if (k0 + BK < K) store_set(1, 1);
        load_set(1, k0 + 3 * BK);
        __syncthreads();
        if (k0 + BK >= K) break;
        mfma_step(1);
        if (k0 + 2 * BK < K) store_set(0, 0);
        load_set(0, k0 + 4 * BK);
        __syncthreads();
    }

    op.epilogue(g, t.m0 + wave * 32 + 4 * lk, t.n0 + lrow, acc0, acc1, wide);
}

}  // namespace xsq
