// Layer 1 of the CDAE (fp32, non-causal) as F(2, 2) along the HOP.
//
// Layer 1 is Conv2d(2 -> 50, (kf, W), stride (1, hop = W / 2)) over the whitened magnitudes (/root/reference/xumx_slicq_v2/
// model.py:130-139): output u of a (b, f) row contracts the window x[u hop .. u hop + W) of every (channel, frequency tap).  In
// units of the hop that is a TWO-tap convolution: with X[u] = the u-th half window (2 kf hop values) and the weights cut
// the same way (W0 = columns dt < hop, W1 = columns dt >= hop)
//     y[u] = W0 X[u] + W1 X[u + 1].
// The implicit GEMM (CdaeL1Op, gemm_tile.h) runs four half-length products per output pair.  F(2, 2) -- the same saving as
// cdae_wino.h takes from layers 2 / 3 -- needs three:
//     m1 = W0 (X[u] - X[u + 1]),   m2 = (W0 + W1) X[u + 1],   m3 = W1 (X[u + 2] - X[u + 1])
//     y[u] = m1 + m2,              y[u + 1] = m2 + m3
// with W0 + W1 summed on the host in fp64 from the BN-folded fp32 weights (xsq_model::d_upool).  Three quarters of the MFMA
// cycles of a kernel whose issue port they fill to three quarters.
//
// As a GEMM: rows = output PAIRS, three accumulator sets of (pairs x 50 columns), K = the 2 kf hop values of a half window.
// Tile = 64 consecutive pairs of the flattened (b, f1, pair) space, 256 threads = 4 waves x 16 pairs.
//  * A operand: NOT staged.  In the MFMA's own layout a lane (pair q, k-quad kq) needs, per chunk of 16 k, the four
//    consecutive values 4 kq .. 4 kq + 3 of its pair's three half windows -- three 16-byte buffer loads at byte offsets
//    0, 4 hop, 8 hop from ONE per-lane address (X[u + 1] IS X[u] displaced by a hop: the input row is contiguous in time).
//    Two subtractions per value make the three operands in registers; each transformed value feeds exactly one lane, so an
//    LDS round trip would buy nothing (cdae_wino.h transforms in registers for the same reason).  Requested one chunk ahead.
//  * K order: (segment = (channel, frequency tap), d < hop) with every segment padded to a multiple of 4 values, so a
//    lane's four values never straddle two segments; the pad values are whatever follows in the row (finite whitened
//    magnitudes of the next half window) against ZERO weights.  Chunks of 16: 2 kf ceil(hop / 4) quads, padded to 4 quads.
//  * B operand: the three weight tiles of a chunk ([component][column < 50][16 k], 9.6 KB) stream through two LDS buffers
//    exactly as the Winograd kernel's do (requested at the start of a chunk, written at its end, one barrier per chunk), row
//    stride 24 words (cdae_wino.h explains the bank pattern); columns 48 / 49 on the vector ALU through one LDS word and
//    v_fmac_f32_dpp row_newbcast.
//  * per chunk and wave: 36 v_mfma_f32_16x16x4_f32 (1,152 cycles) where the implicit GEMM issues 1,536 for the same outputs.
//  * the weights are the MFMA's ROW operand: a lane's accumulator registers are four consecutive output channels of its own pair --
//    epilogue: output sums, shift + ReLU, 16-byte stores straight from registers.
//   LDS 28.8 KB.
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

#ifndef XSQ_L1F_WAVES_PER_EU
#define XSQ_L1F_WAVES_PER_EU 3
#endif
#ifndef XSQ_L1F_SCHED
#define XSQ_L1F_SCHED 0     // >= 0: a scheduling barrier with this mask behind every component's MFMAs (0.407-0.409 against 0.414-0.423
                            // without: profiles/r11_ab_runs.txt r11r); -1: none
#endif
#ifndef XSQ_L1F_ABL
#define XSQ_L1F_ABL 0       // diagnostic builds (wrong results, timings only): 2 no vector columns, 4 no weight stream, 8 no operand loads, 16 no epilogue stores, 32 no MFMAs
#endif

namespace xsq {

constexpr int LF_PAIRS = 64;                              // output pairs per tile (4 waves x 16)
constexpr int LF_COLS = 50;                               // H1 output channels: 48 on the matrix pipe + 2 on the vector ALU
constexpr int LF_BLD = 24;                                // weight tile row: 16 k | 8 pad words (6 slots, cdae_wino.h)
constexpr int LF_BTILE = LF_COLS * LF_BLD;                // words per component tile in LDS (row = component * 50 + column)
constexpr int LF_U16 = 3 * LF_COLS * 16;                  // words of a chunk in global memory: [component][col][16 k] = 600 float4

// chunks of 16 k of a block: 2 kf segments of ceil(hop / 4) quads
__host__ __device__ constexpr int l1f_chunks(int kf, int hop) { return (2 * kf * ((hop + 3) / 4) + 3) / 4; }
// word offset of (chunk s, component j, column col, k kk) inside a (block, target)'s transformed layer-1 weights
__host__ __device__ constexpr int l1f_u_off(int s, int j, int col, int kk) { return s * LF_U16 + (j * LF_COLS + col) * 16 + kk; }

struct L1fTileDev {                // 64 bytes: one scalar load
    int Q0, kf, F, F1;             // first pair of the tile in the flattened (b, f1, pair) space; taps; input / output rows
    int64_t in_off, out_off;       // the block's whitened magnitudes inside xin / the (block, target)'s act1, in floats
    int64_t shift_off, u_off;      // shift vector inside the pool / transformed weights inside the Winograd pool
    int pad0, hop, nchunks, P;     // -, hop, chunks of 16 k, pairs per (b, f1) row = (T1 + 1) / 2
};
static_assert(sizeof(L1fTileDev) == 64, "L1fTileDev is meant to be one 64-byte scalar load");

__global__ __launch_bounds__(256, XSQ_L1F_WAVES_PER_EU) void cdae_l1f_kernel(CdaeArgs a, const L1fTileDev* __restrict__ tiles, int ntiles) {
#pragma clang fp contract(off)
    constexpr int NV = H1 - 48;                                  // real channels past 47
    __shared__ __attribute__((aligned(16))) float Bs[2 * 3 * LF_BTILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const L1fTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.F), "s"(t.F1), "s"(t.in_off), "s"(t.out_off), "s"(t.shift_off), "s"(t.u_off),
                 "s"(t.hop), "s"(t.nchunks), "s"(t.P));
    const int kf = t.kf, F = t.F, F1 = t.F1, P = t.P, hop = t.hop, nchunks = t.nchunks;
    const int Ti = a.S * 2 * hop, T1 = a.T1;
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.xin + t.in_off, 4u * (unsigned)(a.Bn * 2 * F * Ti));     // (< 2^30 bytes: cdae_launch_layer)

    // ---- this lane's pair: row base of its first half window; output offsets for the epilogue
    const int pl = wave * 16 + q;
    const int Qg = t.Q0 + pl;                                    // pair of the flattened (b, f1, pair) space: tiles run across batch items
    const bool pair_ok = Qg < a.Bn * F1 * P;
    const int b = Qg / (F1 * P), Q = Qg - b * (F1 * P);
    const int f1 = Q / P, p = Q - f1 * P;
    const unsigned vo_row = pair_ok ? 4u * (unsigned)((b * 2 * F + f1) * Ti + 2 * p * hop) : BUF_OOB;

    // ---- K cursor of this lane: quad e4 = kq + 4 s of the padded K order -> (segment, quad inside the segment).  seg_off =
    // floats from the row base to the segment's row: segment (c, df) is input row (c F + f1 + df)
    const int hq = (hop + 3) >> 2;
    int dq = kq, c_df = 0, c_c = 0, seg_off = 0;
    auto norm = [&]() {
        while (dq >= hq) {
            dq -= hq; seg_off += Ti;
            if (++c_df == kf) { c_df = 0; ++c_c; seg_off += (F - kf) * Ti; }
        }
    };
    norm();
    float4 xa[2][3];
    auto load_a = [&](int set) {
        const unsigned vo = c_c < 2 ? vo_row + 4u * (unsigned)(seg_off + 4 * dq) : BUF_OOB;
        if (XSQ_L1F_ABL & 8) { xa[set][0] = make_float4(1.f, 2.f, 3.f, (float)vo); xa[set][1] = make_float4(2.f, 3.f, 1.f, 4.f); xa[set][2] = make_float4(3.f, 1.f, 2.f, 4.f); }
        else {
        xa[set][0] = buf_ld4(rin, vo, 0);
        xa[set][1] = buf_ld4(rin, vo, 4 * hop);
        xa[set][2] = buf_ld4(rin, vo, 8 * hop);
        }
        dq += 4;
        norm();
    };

    // ---- weight stream: chunk s = three component tiles of 50 columns -> the LDS buffer of its parity.  600 float4: thread
    // tid takes float4 tid + 256 r; the spare threads of r = 2 load past the descriptor (zeros) and write into pad words
    const __amdgpu_buffer_rsrc_t ru = buf_rsrc(a.upool + t.u_off, 4u * (unsigned)(nchunks * LF_U16));
    const int b_lds0 = (tid >> 2) * LF_BLD + 4 * (tid & 3);      // float4 x -> LDS row x >> 2, k quad x & 3
    constexpr int LAST = LF_U16 / 4 - 512;                       // float4s of the third round (88)
    float4 gb[3];
    auto load_chunk = [&](int s) {
        if (XSQ_L1F_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < 3; ++r) gb[r] = buf_ld4(ru, (r < 2 || tid < LAST) ? 16u * (unsigned)(tid + 256 * r) : BUF_OOB, 4 * s * LF_U16);
    };
    auto store_chunk = [&](int buf) {
        if (XSQ_L1F_ABL & 4) return;
        float* Bw = Bs + buf * 3 * LF_BTILE;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            *reinterpret_cast<float4*>(&Bw[(r < 2 || tid < LAST) ? b_lds0 + 64 * r * LF_BLD : (tid & 127) * LF_BLD + 20]) = gb[r];
    };

    const int bf = q * LF_BLD + 4 * kq;                          // weight tile: column q of a 16-column block, k-quad kq
    const int bv = 48 * LF_BLD + 4 * kq + (q & 3);               // vector columns: one word per lane (cdae_wino.h)

    f32x4 acc[3][3];
    float accv[3][NV];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) accv[j][cc] = 0.f;
    }

    struct Frag { float4 w[3]; float u[NV]; };
    auto read_frag = [&](Frag& f, const float* Bt) {
        f.w[0] = *reinterpret_cast<const float4*>(&Bt[bf]);
        f.w[1] = *reinterpret_cast<const float4*>(&Bt[bf + 16 * LF_BLD]);
        f.w[2] = *reinterpret_cast<const float4*>(&Bt[bf + 32 * LF_BLD]);
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) f.u[cc] = Bt[bv + cc * LF_BLD];
    };

    // one chunk: operands of register set SET against the LDS buffer of the chunk's parity; the operands and the weights of chunk
    // s + 1 are requested first (uniform condition).  (Measured, profiles/r11_ab_runs.txt: operands TWO chunks ahead -- a third
    // register set, 135 registers, three waves per SIMD instead of four -- 5 % slower; s_setprio around the MFMAs: nothing.)
    int cnt = 0;
    auto chunk = [&](auto set_c, int s) {
        constexpr int SET = decltype(set_c)::value;
        const bool next = s + 1 < nchunks;
        if (next) { load_chunk(s + 1); load_a(SET ^ 1); }
        const int cur = cnt & 1;
        const float* Bc = Bs + cur * 3 * LF_BTILE;
        Frag fr[2];
        read_frag(fr[0], Bc);
        const float4 x0 = xa[SET][0], x1 = xa[SET][1], x2 = xa[SET][2];
        const float d[3][4] = {{x0.x - x1.x, x0.y - x1.y, x0.z - x1.z, x0.w - x1.w}, {x1.x, x1.y, x1.z, x1.w},
                               {x2.x - x1.x, x2.y - x1.y, x2.z - x1.z, x2.w - x1.w}};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const Frag& f = fr[j & 1];
            if (j < 2) read_frag(fr[(j + 1) & 1], Bc + (j + 1) * LF_BTILE);
            const float wa[4] = {f.w[0].x, f.w[0].y, f.w[0].z, f.w[0].w}, wb[4] = {f.w[1].x, f.w[1].y, f.w[1].z, f.w[1].w};
            const float wc[4] = {f.w[2].x, f.w[2].y, f.w[2].z, f.w[2].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (XSQ_L1F_ABL & 32) { acc[j][0][i] += d[j][i] * wa[i]; acc[j][1][i] += d[j][i] * wb[i]; acc[j][2][i] += d[j][i] * wc[i]; continue; }
                // (the WEIGHTS are the MFMA's row operand: accumulator register r of a lane is output channel 16 cb + 4 kq + r of the
                //  lane's OWN pair -- 16-byte stores straight from registers, cdae_wino.h / cdae_l4f.h)
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], d[j][i], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i], d[j][i], acc[j][1], 0, 0, 0);
                acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[i], d[j][i], acc[j][2], 0, 0, 0);
            }
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) {
                if (XSQ_L1F_ABL & 2) { accv[j][cc] += d[j][0] + f.u[cc]; continue; }
                asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                    : "+v"(accv[j][cc])
                    : "v"(f.u[cc]), "v"(d[j][0]), "v"(d[j][1]), "v"(d[j][2]), "v"(d[j][3]));
            }
            if (XSQ_L1F_SCHED >= 0) __builtin_amdgcn_sched_barrier(XSQ_L1F_SCHED);
        }
        if (next) store_chunk(cur ^ 1);          // the other buffer: last read in the chunk before, every wave is past that chunk's barrier
        cnt += 1;
        __syncthreads();
    };

    // ---- prologue: chunk 0's weights in LDS buffer 0, its operands in register set 0
    load_chunk(0);
    load_a(0);
    store_chunk(0);
    __syncthreads();
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int s = 0;
        for (int pr = 0; pr < nchunks / 2; ++pr, s += 2) { chunk(I0{}, s); chunk(I1{}, s + 1); }
        if (nchunks & 1) chunk(I0{}, s);
    }

    // ---- epilogue: y[u] = m1 + m2, y[u + 1] = m2 + m3, shift + ReLU, one 16-byte store per column block and output from the
    // lane's own registers (displacements in the lane offset: common.h, buf_st4).  A pair that does not exist and the phantom
    // second output of a row's last pair (T1 is odd: it would land on the NEXT row's first output) are switched out of range.
    const float* shift = a.pool + t.shift_off;
    const __amdgpu_buffer_rsrc_t ro = buf_rsrc(a.act1 + t.out_off, 0x40000000u);
    const bool ok0 = pair_ok, ok1 = pair_ok && 2 * p + 1 < T1;
    const unsigned vo = 4u * (unsigned)(((b * F1 + f1) * T1 + 2 * p) * CS + 4 * kq);
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const float4 sh = *reinterpret_cast<const float4*>(shift + 16 * cb + 4 * kq);
        const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float y0[4], y1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m1 = acc[0][cb][r], m2 = acc[1][cb][r], m3 = acc[2][cb][r];
            y0[r] = fmaxf((m1 + m2) + shv[r], 0.f);
            y1[r] = fmaxf((m2 + m3) + shv[r], 0.f);
        }
        const bool live = !((XSQ_L1F_ABL & 16) && y0[0] != 1.2345e-30f);
        buf_st4(make_float4(y0[0], y0[1], y0[2], y0[3]), ro, (ok0 && live) ? vo + 64u * cb : BUF_OOB, 0);
        buf_st4(make_float4(y1[0], y1[1], y1[2], y1[3]), ro, (ok1 && live) ? vo + 64u * cb + 4u * CS : BUF_OOB, 0);
    }
    float y0v[4] = {0.f, 0.f, 0.f, 0.f}, y1v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NV; ++cc) {
        float m[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                // the four k-quads' partial sums meet here (fixed order)
            float x = accv[j][cc];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            m[j] = x;
        }
        y0v[cc] = m[0] + m[1];
        y1v[cc] = m[1] + m[2];
    }
    const float4 sh = *reinterpret_cast<const float4*>(shift + 48);
    const unsigned v48 = vo - 16u * (unsigned)kq + 192u;             // channel 48 of the row
    buf_st4(make_float4(fmaxf(y0v[0] + sh.x, 0.f), fmaxf(y0v[1] + sh.y, 0.f), fmaxf(y0v[2] + sh.z, 0.f), fmaxf(y0v[3] + sh.w, 0.f)), ro,
            (ok0 && kq == 0) ? v48 : BUF_OOB, 0);
    buf_st4(make_float4(fmaxf(y1v[0] + sh.x, 0.f), fmaxf(y1v[1] + sh.y, 0.f), fmaxf(y1v[2] + sh.z, 0.f), fmaxf(y1v[3] + sh.w, 0.f)), ro,
            (ok1 && kq == 0) ? v48 + 4u * CS : BUF_OOB, 0);
}

}  // namespace xsq
