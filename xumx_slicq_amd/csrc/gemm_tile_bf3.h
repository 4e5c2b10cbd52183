// Split-bf16 ("bf16x3") variant of the grouped tile engine of gemm_tile.h: same operators, same tile
// tables, same epilogues; the contraction runs on v_mfma_f32_32x32x16_bf16 with every fp32 operand
// value x carried as the pair  hi = bf16(x), lo = bf16(x - hi)  (round to nearest even both times, so
// |x - hi - lo| <= 2^-18 |x|) and every product as
//        a*b  ~=  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (fp32 accumulation in the MFMA),
// dropping only a_lo*b_lo (<= 2^-18 |ab|).  Per product that is ~16-17 significant bits -- between
// fp32 and the TF32 convolutions the reference's own torch-cuda backend runs by default -- at 3/16 of
// the matrix-pipe time of v_mfma_f32_32x32x2_f32.  Selected per model (xsq_model_set_precision);
// the exact-fp32 engine stays the default of the C ABI.
//
// LDS image of a K-step (16 k-values) of one tile row: [16 x hi bf16 | 16 x lo bf16] = 64 bytes at the
// fp32 engine's row stride of 20 words (conflict-free ds_read_b128); lane l = (row l&31, half h = l>>5)
// reads hi[8h .. 8h+7] and lo[8h .. 8h+7] as two 16-byte fragments, which is exactly the operand map of
// the instruction (element j of the fragment is k = 8h + j, cdna guide section 3).  The split itself is
// done ONCE per value by whoever produces it (the previous layer's epilogue, the magnitude kernel, the
// weight pool conversion) and stored as one word (hi << 16) | lo -- same bytes as fp32, so the
// operators' loaders are unchanged -- because every activation is re-read kf x 4 times by the next layer.
#pragma once
#include "gemm_tile.h"

namespace xsq {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void bf3_split2(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2_t v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));       // v_cvt_pk_bf16_f32
    const float fx = __builtin_bit_cast(float, hi << 16), fy = __builtin_bit_cast(float, hi & 0xffff0000u);
    const f32x2_t r = {x - fx, y - fy};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// the storage format of split operands: one word per value, (bf16 hi << 16) | bf16 lo
__device__ __forceinline__ void bf3_words2(float x, float y, float& wx, float& wy) {
    unsigned hi, lo;
    bf3_split2(x, y, hi, lo);
    wx = __builtin_bit_cast(float, (hi << 16) | (lo & 0xffffu));
    wy = __builtin_bit_cast(float, (hi & 0xffff0000u) | (lo >> 16));
}
__device__ __forceinline__ float bf3_word(float x) {
    float wx, wy;
    bf3_words2(x, 0.f, wx, wy);
    return wx;
}

// KS = 16-value k-chunks per K-step (one barrier per K-step; LDS row = KS x (16 hi | 16 lo) + 4 pad words:
// 20 or 36 words, both conflict-free for ds_read_b128 over 16 consecutive rows)
template <class Op, int MT = 1, int KS = 1>
__global__ __launch_bounds__(256) void grouped_gemm_bf3_kernel(Op op, const TileDev* __restrict__ tiles, int ntiles) {
    static_assert(!aux_of<Op>::on, "operators with an aux stream run on the fp32 engine");
    constexpr int BM = GEMM_BM * MT, BN = GEMM_BN, BK = 16 * KS, LD = 16 * KS + 4;
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
    const float* bp = g.B + (int64_t)(t.n0 + s_row) * g.ldb + s_kq;   // Bt[n][k]
    const bool b_on = wide || s_row < 32;

    float4 ga[2][RA][KS];
    float4 gb[2][KS];
    auto load_set = [&](int set, int k) {
        if (k < K) {
#pragma unroll
            for (int c = 0; c < KS; ++c) {
#pragma unroll
                for (int i = 0; i < RA; ++i) ga[set][i][c] = op.load_a4(g, ra[i], k + 16 * c + s_kq);     // zeros past K
                if (b_on) gb[set][c] = (k + 16 * c < K) ? *reinterpret_cast<const float4*>(bp + k + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // Operands arrive PRE-SPLIT (one word (hi << 16) | lo per value, written by the producing kernel / the
    // weight pool conversion): staging is four byte permutes per 16-byte load, no arithmetic.
    auto put = [&](unsigned* row, const float4& v) {       // words [2q, 2q+1] = hi, [8+2q, 8+2q+1] = lo, q = s_kq/4
        const unsigned e0 = __builtin_bit_cast(unsigned, v.x), e1 = __builtin_bit_cast(unsigned, v.y);
        const unsigned e2 = __builtin_bit_cast(unsigned, v.z), e3 = __builtin_bit_cast(unsigned, v.w);
        const unsigned h0 = __builtin_amdgcn_perm(e1, e0, 0x07060302u), h1 = __builtin_amdgcn_perm(e3, e2, 0x07060302u);
        const unsigned l0 = __builtin_amdgcn_perm(e1, e0, 0x05040100u), l1 = __builtin_amdgcn_perm(e3, e2, 0x05040100u);
        *reinterpret_cast<uint2*>(row + (s_kq >> 1)) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(row + 8 + (s_kq >> 1)) = make_uint2(l0, l1);
    };
    auto store_set = [&](int set, int buf) {
        unsigned* Aw = As0 + buf * BM * LD;
        unsigned* Bw = Bs0 + buf * BN * LD;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
#pragma unroll
            for (int i = 0; i < RA; ++i) put(&Aw[(s_row + 64 * i) * LD + 16 * c], ga[set][i][c]);
            put(&Bw[s_row * LD + 16 * c], gb[set][c]);
        }
    };
#pragma unroll
    for (int c = 0; c < KS; ++c) {
#pragma unroll
        for (int i = 0; i < RA; ++i) ga[0][i][c] = ga[1][i][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        gb[0][c] = gb[1][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_set(0, 0);

    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; acc1[i][r] = 0.f; }

    const int lrow = lane & 31, lk = lane >> 5;
    const int a_frag = (wave * 32 * MT + lrow) * LD + 4 * lk;
    const int b_frag = lrow * LD + 4 * lk;

    auto frag = [&](const unsigned* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p)); };
    auto mfma_step = [&](int buf) {
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const unsigned* As = As0 + buf * BM * LD + 16 * c;
        const unsigned* Bs = Bs0 + buf * BN * LD + 16 * c;
        bf16x8_t ah[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) { ah[i] = frag(&As[a_frag + i * 32 * LD]); al[i] = frag(&As[a_frag + i * 32 * LD + 8]); }
        const bf16x8_t b0h = frag(&Bs[b_frag]), b0l = frag(&Bs[b_frag + 8]);
        if (wide) {
            const bf16x8_t b1h = frag(&Bs[b_frag + 32 * LD]), b1l = frag(&Bs[b_frag + 32 * LD + 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {      // small terms first
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b0h, acc0[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b1h, acc1[i], 0, 0, 0);
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b0l, acc0[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b1l, acc1[i], 0, 0, 0);
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b0h, acc0[i], 0, 0, 0);
                acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b1h, acc1[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b0h, acc0[i], 0, 0, 0);
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b0l, acc0[i], 0, 0, 0);
                acc0[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], b0h, acc0[i], 0, 0, 0);
            }
        }
      }
    };

    // same pipeline as the fp32 engine: LDS double-buffered (one barrier per K-step), global loads two
    // K-steps ahead in two register sets
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

#pragma unroll
    for (int i = 0; i < MT; ++i)
        op.epilogue(g, t.m0 + (wave * MT + i) * 32 + 4 * lk, t.n0 + lrow, acc0[i], acc1[i], wide);
}

}  // namespace xsq
