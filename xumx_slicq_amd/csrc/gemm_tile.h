// fp32 MFMA tile engine for the grouped (ragged) GEMM-shaped kernels of the hot path:
// per-band DFTs of the sliCQT / isliCQT and the four CDAE layers as implicit GEMMs.
//
// One workgroup = 256 threads = 4 wavefronts (64 lanes each) stacked along M.
// Tile BM x 64, K-step 16.  BM = 128 -> 32x64 per wave, BM = 256 -> 64x64 per wave,
// built from v_mfma_f32_32x32x2_f32 (exact fp32: the parity bar is 1e-4 RMS against a
// torch-CPU fp32 reference, so operands stay fp32).  Operand maps (cdna guide section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C: reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
//
// A is produced by an operator-specific loader (gathers, reflections, zero padding);
// B is always a dense row-major matrix padded to (K%16==0, N%64==0) with zeros, so its
// loads are unconditional 16-byte loads.  Global->LDS staging goes through registers
// with the next K-step's loads issued before the current step's MFMAs.
#pragma once
#include "common.h"

namespace xsq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Blocks b and b+8 share an XCD (observed round-robin placement; speed only).  Give each
// XCD a contiguous run of tiles so that neighbouring tiles, which share the B matrix of
// their group and often A rows, hit the same 4 MiB L2.  Bijective for any grid size.
__device__ inline int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7;
    return x * q + (x < r ? x : r) + (bid >> 3);
}

template <int BM, class Op>
__global__ __launch_bounds__(256) void grouped_gemm_kernel(Op op, const TileDev* __restrict__ tiles,
                                                            int ntiles) {
    constexpr int BN = 64, BK = 16, LDA = BK + 1;
    constexpr int WM = BM / 4;      // rows per wave
    constexpr int MT = WM / 32;     // 32-row MFMA tiles per wave along M
    constexpr int RA = BM / 64;     // A rows staged per thread

    __shared__ float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const typename Op::Group g = op.group(t.group);
    const int K = g.K;

    // ---- staging assignment -------------------------------------------------
    const int a_row = tid >> 2;          // 0..63 (+64*i)
    const int a_kq = (tid & 3) * 4;      // 0,4,8,12
    typename Op::RowA ra[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = op.row_a(g, t.m0 + a_row + 64 * i);
    const int b_k = tid >> 4;            // 0..15
    const int b_n = (tid & 15) * 4;      // 0..60
    const float* bp = g.B + (int64_t)b_k * g.ldb + t.n0 + b_n;

    float4 ga[RA];
    float4 gb;
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[i] = op.load_a4(g, ra[i], a_kq);
    gb = *reinterpret_cast<const float4*>(bp);

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31, lk = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += BK) {
        // registers -> LDS
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float* d = &As[(a_row + 64 * i) * LDA + a_kq];
            d[0] = ga[i].x; d[1] = ga[i].y; d[2] = ga[i].z; d[3] = ga[i].w;
        }
        *reinterpret_cast<float4*>(&Bs[b_k * BN + b_n]) = gb;
        __syncthreads();
        // issue next step's global loads; they land while the MFMAs run
        if (k0 + BK < K) {
#pragma unroll
            for (int i = 0; i < RA; ++i) ga[i] = op.load_a4(g, ra[i], k0 + BK + a_kq);
            gb = *reinterpret_cast<const float4*>(bp + (int64_t)(k0 + BK) * g.ldb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = As[(wave * WM + i * 32 + lrow) * LDA + kk + lk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + lk) * BN + j * 32 + lrow];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = t.m0 + wave * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            op.store_row(g, m, t.n0 + lrow, acc[i][0][r], acc[i][1][r]);
        }
    }
}

}  // namespace xsq
