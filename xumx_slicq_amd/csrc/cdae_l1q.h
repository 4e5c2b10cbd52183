// Layer 1 of the CDAE for the FOUR (or two) targets of a block in one tile (fp32 inference, non-causal first layer).
// AN A/B ARM, OFF BY DEFAULT: built in round 4, bitwise the per-target tiles, and measured SLOWER -- layer 1 0.545 ms per 240 s
// track on the per-target tiles, 0.58 with two targets per tile (three workgroups per CU), 0.63 (four, 32 B of scratch), 0.66
// with four targets per tile (two workgroups per CU); the step within noise (profiles/r07_ab_runs.txt, r07g).  Sharing the
// operand buys less than the resident waves it costs -- the pattern of rounds 2 and 3.
//
// Conv2d(2 -> 50, (kf, W), stride (1, W/2)) of model.py:130-139 reads the SAME whitened magnitude for every target
// (model.py:244-247: the four CDAEs of a block are applied to one input); only the weights differ.  The tile engine
// (gemm_tile.h, XW = 1) ran it as 4 x 70 independent groups: every 128 x 16 operand step was loaded, staged and read
// out of LDS four times, and a K = 160 tile spent 676 vector instructions outside its K loop against 930 inside -- on
// this part beside fp32 MFMAs those are not hidden (band_dft4.h, "Vector issue"): layer 1 sat at 0.55 of the matrix peak.
// Here a workgroup owns 128 rows x 4 targets x 52 channels: the A operand (row offsets, cursor, loads, LDS staging,
// fragment reads) is paid once per four targets' MFMAs, the prologue once per tile, and two workgroups of 4 waves per CU
// leave every wave 256 registers for its 4 x (32x32 + 2 16x16 + vector-column) accumulator sets.
//   K-step: A 128 x 16 (2 float4 per thread) + B 4 x 52 x 16 (4 float4 per thread of rows 0..51) -> LDS (53.8 KB,
//   double-buffered, one barrier per K-step); per wave 4 x (8 v_mfma_f32_32x32x2_f32 + 8 v_mfma_f32_16x16x4_f32 +
//   16 v_fmac for channels 48, 49) on ONE set of A fragments.
// Per accumulator the MFMA sequence and the k order are the tile engine's: the result is bitwise the XW = 1 path
// (tests/test_model_gpu.py::test_layer1_quad_tiles_are_bitwise_the_per_target_tiles).
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

namespace xsq {

#ifndef XSQ_L1Q_WAVES2
#define XSQ_L1Q_WAVES2 3      // waves per SIMD of the two-target instantiation (4: 128 registers, 32 B of scratch)
#endif
constexpr int L1Q_BM = 128, L1Q_LD = GEMM_LD;

// NTG = targets per tile (TileDev.n0 = the first one): 4 -> 212 registers, two workgroups per CU; 2 -> four workgroups per CU
template <class L1Op, class GroupT, int NTG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTG == 4 ? 2 : XSQ_L1Q_WAVES2, NTG == 4 ? 2 : XSQ_L1Q_WAVES2)))
void cdae_l1_quad_kernel(CdaeArgs a, const TileDev* __restrict__ tiles, int ntiles) {
    constexpr int BM = L1Q_BM, LD = L1Q_LD, BK = GEMM_BK, RA = BM / 64;
    constexpr int ABUF = BM * LD, BBUF = NTG * CS * LD;
    constexpr int NV = L1Op::NV;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ABUF + BBUF)];
    float* const As0 = lds;
    float* const Bs0 = lds + 2 * ABUF;
    static_assert(4 * 32 * CS <= 2 * (ABUF + BBUF), "epilogue images do not fit the staging buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    const L1Op op{a};
    const int blk = t.group, tg0 = t.n0;                       // block index; the tile covers targets tg0 .. tg0 + NTG - 1
    const GroupT g = op.group(4 * blk);                        // geometry, input and K are the same for the four
    const CdaeBlockDev& bd = a.blocks[blk];
    const int K = g.K;

    const int s_row = tid >> 2, s_kq = (tid & 3) * 4;
    typename L1Op::RowA ra[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = op.row_a(g, t.m0 + s_row + 64 * i);
    // the four weight matrices Bt[n][k] (transposed, K padded to 16): rows 0..51 of each through a buffer descriptor,
    // the K-step as the scalar offset; lanes of rows 52..63 switched off
    __amdgpu_buffer_rsrc_t rB[NTG];
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) rB[tg] = buf_rsrc(a.pool + bd.w1[tg0 + tg], 0x7FFFFFFFu);
    const bool b_on = s_row < CS;
    const unsigned bvo = b_on ? 4u * (unsigned)(s_row * g.ldb + s_kq) : BUF_OOB;

    float4 ga[2][RA], gb[2][NTG];
    typename L1Op::Cursor kc = op.cursor(g, s_kq);
    auto load_set = [&](int set, int k) {
        if (k < K) {
#pragma unroll
            for (int i = 0; i < RA; ++i) ga[set][i] = op.load_a4(g, ra[i], kc);
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) gb[set][tg] = buf_ld4(rB[tg], bvo, 4 * k);
            op.advance(g, kc);
        }
    };
    auto store_set = [&](int set, int buf) {
        float* Aw = As0 + buf * ABUF;
        float* Bw = Bs0 + buf * BBUF;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<float4*>(&Aw[(s_row + 64 * i) * LD + s_kq]) = ga[set][i];
        if (b_on) {
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) *reinterpret_cast<float4*>(&Bw[(tg * CS + s_row) * LD + s_kq]) = gb[set][tg];
        }
    };
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[0][i] = ga[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) gb[0][tg] = gb[1][tg] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_set(0, 0);

    f32x16 acc0[NTG];
    f32x4 acc16[NTG][2];
    float accv[NTG][4];
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[tg][r] = 0.f;
        acc16[tg][0] = acc16[tg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) accv[tg][cc] = 0.f;
    }

    const int lrow = lane & 31, lk = lane >> 5, q16 = lane >> 4;
    const int a_frag = (wave * 32 + lrow) * LD + 8 * lk;
    const int a16_frag = (wave * 32 + (lane & 15)) * LD + 4 * q16;
    const int b_frag = lrow * LD + 8 * lk;
    const int b16_frag = (32 + (lane & 15)) * LD + 4 * q16;
    const int bv_frag = 48 * LD + 8 * lk;

    auto mfma_step = [&](int buf) {
        const float* As = As0 + buf * ABUF;
        const float4 lo = *reinterpret_cast<const float4*>(&As[a_frag]);
        const float4 hi = *reinterpret_cast<const float4*>(&As[a_frag + 4]);
        const float4 x0 = *reinterpret_cast<const float4*>(&As[a16_frag]);
        const float4 x1 = *reinterpret_cast<const float4*>(&As[a16_frag + 16 * LD]);
        const float av[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) {
            const float* Bs = Bs0 + buf * BBUF + tg * CS * LD;
            const float4 b_lo = *reinterpret_cast<const float4*>(&Bs[b_frag]);
            const float4 b_hi = *reinterpret_cast<const float4*>(&Bs[b_frag + 4]);
            const float4 y = *reinterpret_cast<const float4*>(&Bs[b16_frag]);
            const float b0[8] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
            const float yb[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc0[tg] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0[tg], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc16[tg][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], yb[j], acc16[tg][0], 0, 0, 0);
                acc16[tg][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], yb[j], acc16[tg][1], 0, 0, 0);
            }
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) {
                const float4 v0 = *reinterpret_cast<const float4*>(&Bs[bv_frag + cc * LD]);
                const float4 v1 = *reinterpret_cast<const float4*>(&Bs[bv_frag + cc * LD + 4]);
                // (volatile asm, as in gemm_tile.h: plain fmaf chains were sunk past the K-step barrier with their operands spilled)
                asm volatile("v_fmac_f32 %0, %1, %9\n\tv_fmac_f32 %0, %2, %10\n\tv_fmac_f32 %0, %3, %11\n\tv_fmac_f32 %0, %4, %12\n\t"
                             "v_fmac_f32 %0, %5, %13\n\tv_fmac_f32 %0, %6, %14\n\tv_fmac_f32 %0, %7, %15\n\tv_fmac_f32 %0, %8, %16"
                             : "+v"(accv[tg][cc])
                             : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]),
                               "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
            }
        }
    };

    // prologue: K-step 0 -> LDS buffer 0; K-steps 1 and 2 in flight in register sets 1 and 0
    store_set(0, 0);
    load_set(1, BK);
    load_set(0, 2 * BK);
    __syncthreads();
    const int nk = (K + BK - 1) / BK;
    int k0 = 0;
    for (int pr = 0; pr < nk / 2; ++pr, k0 += 2 * BK) {
        mfma_step(0);
        store_set(1, 1);
        load_set(1, k0 + 3 * BK);
        __syncthreads();
        mfma_step(1);
        if (k0 + 2 * BK < K) store_set(0, 0);
        load_set(0, k0 + 4 * BK);
        __syncthreads();
    }
    if (nk & 1) {
        mfma_step(0);
        __syncthreads();        // (the epilogue images reuse the staging buffers)
    }

    // ---- epilogue: one target after the other through this wave's 32 x 52 LDS image (same wave, LDS in order) ----
#pragma unroll
    for (int tg = 0; tg < NTG; ++tg) {
        GroupT gt = g;
        gt.shift = a.pool + bd.s1[tg0 + tg];
        gt.out = a.act1 + (int64_t)CS * a.Bn * a.T1 * (4 * (int64_t)bd.cumF1 + (int64_t)(tg0 + tg) * bd.F1);
        relu_shift_epilogue_xw(gt, t.m0 + wave * 32, lane, acc0[tg], acc16[tg], accv[tg], NV, lds + wave * 32 * CS);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace xsq
