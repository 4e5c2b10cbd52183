// Layer 4 of the CDAE (fp32 inference, masks only) as F(2, 2) along the HOP.
//
// Layer 4 is ConvTranspose2d(50 -> 2, (kf, W), stride (1, hop = W / 2)) + bias + sigmoid (/root/reference/xumx_slicq_v2/
// model.py:171-181).  Output half window u of a (b, f) row -- the hop values tau = u hop + dt of both channels, N = W columns
// n = c hop + dt -- receives position u of the layer-3 activations through the kernel columns dt < hop (Wa) and position
// u - 1 through the columns dt + hop (Wb):
//     out[u] = A[u] Wa + A[u - 1] Wb                (A[u] = the 52 kf channels of input rows f - df at position u)
// -- a two-tap convolution in units of the hop, as layer 1 is (cdae_l1f.h).  The implicit GEMM (CdaeL4Op) runs four products
// of K = 52 kf per output pair; F(2, 2) needs three:
//     m1 = (A[u - 1] - A[u]) Wb,   m2 = A[u] (Wa + Wb),   m3 = (A[u + 1] - A[u]) Wa
//     out[u] = m1 + m2,            out[u + 1] = m2 + m3
// with Wa + Wb summed on the host in fp64 (xsq_model::d_upool), and its K is exact: 52 channels as three chunks of 16 and
// one MFMA of the four tail channels per product and column block, where the GEMM pads 104 to 112.
//
// Tile = 64 consecutive output pairs of the flattened (b, f, pair) space x one column tile of 16 NCB <= 64
// columns; 256 threads = 4 waves x 16 pairs.  To = 2 S is even: every pair has both outputs.
//  * B operand: the column tile's three weight tiles of ONE frequency tap ([component][column][52 k], 13 KB per 16 columns)
//    go into LDS whole, once per tap -- for the 67 of 70 blocks with one tap the K loop has NO barrier and no stream.
//    Row stride 56 words (cdae_wino.h explains the bank pattern of the 16-byte reads).
//  * A operand: not staged (cdae_l1f.h): a lane (pair q, k-quad kq) loads channels 4 kq .. 4 kq + 3 of each chunk and channel
//    48 + kq of its pair's three positions 2 p - 1, 2 p, 2 p + 1 straight into the MFMA layout -- nine 16-byte and three
//    4-byte buffer loads per tap, all in flight together --, two subtractions per value make the three operands.
//  * per tap and wave: 39 NCB v_mfma_f32_16x16x4_f32 (4,992 cycles at NCB = 4) where the GEMM issues 7,168 for the same outputs.
//  * epilogue: output sums, bias, sigmoid, 4-byte buffer stores (16 consecutive lanes = 64 bytes of one output row).
//   LDS 43.0 KB + 256 B: three workgroups per CU.
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

#ifndef XSQ_L4F_WAVES_PER_EU
#define XSQ_L4F_WAVES_PER_EU 3
#endif
#ifndef XSQ_L4F_ABL
#define XSQ_L4F_ABL 0       // diagnostic builds (wrong results, timings only): 4 no weight tiles, 8 no operand loads, 16 no epilogue stores
#endif

namespace xsq {

constexpr int L4_PAIRS = 64;                              // output pairs per tile (4 waves x 16)
constexpr int L4_BLD = 56;                                // weight tile row: 52 k + 4 pad words (14 slots)
constexpr int L4_BROWS = 3 * 64;                          // rows of a staged tap: component * 64 + column

// floats of one frequency tap of a (block, target)'s transformed layer-4 weights: column tiles of 64 (the last one 16 NCB wide),
// each [component][column][52 k]
__host__ __device__ constexpr int l4f_cols(int W) { return (W + 15) / 16 * 16; }
__host__ __device__ constexpr int l4f_tap_floats(int W) { return 3 * CS * l4f_cols(W); }
// float offset of (column tile starting at n0, component j, column n, channel ci) inside a tap
__host__ __device__ constexpr int l4f_u_off(int W, int n0, int j, int n, int ci) {
    return 3 * CS * n0 + (j * (l4f_cols(W) - n0 < 64 ? l4f_cols(W) - n0 : 64) + (n - n0)) * CS + ci;
}

struct L4fTileDev {                // 64 bytes: one scalar load
    int Q0, kf, F, F1;             // first pair of the tile in the flattened (b, f, pair) space; taps; output / input rows
    int64_t in_off, out_off;       // the (block, target)'s act3 / masks, in floats
    int64_t bias_off, u_off;       // output bias (2) inside the pool / the column tile's weights of tap 0 inside the Winograd pool
    int pad0, hop, n0, P;          // -, hop, first column of the tile, pairs per (b, f) row = S
};
static_assert(sizeof(L4fTileDev) == 64, "L4fTileDev is meant to be one 64-byte scalar load");

template <int NCB>
__device__ __forceinline__ void cdae_l4f_body(const CdaeArgs& a, const L4fTileDev& t, float* const Bs, unsigned* const obase) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int kf = t.kf, F = t.F, F1 = t.F1, P = t.P, hop = t.hop, n0 = t.n0;
    const int T1 = a.T1, W = 2 * hop;
    const int ST = a.S * W;                                      // floats of one (b, c, f) output row
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.act3 + t.in_off, 4u * (unsigned)(a.Bn * F1 * T1 * CS));   // (< 2^30 bytes: cdae_launch_layer)

    // ---- this lane's pair
    const int pl = wave * 16 + q;
    const int Qg = t.Q0 + pl;                                    // pair of the flattened (b, f, pair) space: tiles run across batch items
    const bool pair_ok = Qg < a.Bn * F * P;
    const int b = Qg / (F * P), Q = Qg - b * (F * P);
    const int f = Q / P, p = Q - f * P;
    if (kq == 0) obase[pl] = pair_ok ? 4u * (unsigned)((b * 2 * F + f) * ST + 2 * p * hop) : 0xffffffffu;
    // positions 2 p - 1 (exists for p > 0), 2 p, 2 p + 1 (exists below T1 = 2 S - 1: not for the row's last pair)
    const bool ok0 = pair_ok && p > 0, ok2 = pair_ok && 2 * p + 1 < T1;

    // ---- weight tiles of a tap -> LDS.  Global: [component][column < 16 NCB][52 k] = 39 x 16 NCB float4; float4 x -> LDS row
    // component * 64 + column, words 4 (x % 13) ..
    constexpr int NB4 = 3 * 16 * NCB * (CS / 4);
    constexpr int NLD = (NB4 + 255) / 256;
    const int tapf = l4f_tap_floats(W);
    const __amdgpu_buffer_rsrc_t ru = buf_rsrc(a.upool + t.u_off, 4u * (unsigned)((kf - 1) * tapf + 3 * CS * 16 * NCB));
    int b_lds[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int x = tid + 256 * r, row = x / (CS / 4), k4 = x - row * (CS / 4);
        const int j = row / (16 * NCB), col = row - j * (16 * NCB);
        b_lds[r] = x < NB4 ? (j * 64 + col) * L4_BLD + 4 * k4 : -1;
    }
    float4 gb[NLD];
    auto load_b = [&](int df) {
        if (XSQ_L4F_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < NLD; ++r) gb[r] = buf_ld4(ru, (r < NLD - 1 || tid + 256 * r < NB4) ? 16u * (unsigned)(tid + 256 * r) : BUF_OOB, 4 * df * tapf);
    };
    auto store_b = [&]() {
        if (XSQ_L4F_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < NLD; ++r)
            if (r < NLD - 1 || b_lds[r] >= 0) *reinterpret_cast<float4*>(&Bs[b_lds[r]]) = gb[r];
    };

    // ---- operands of a tap: input row f - df, three positions, channels 16 s + 4 kq .. (s < 3) and 48 + kq
    float4 xa[3][3];
    float xt[3];
    auto load_a = [&](int df) {
        const int fi = f - df;
        const bool row_ok = (unsigned)fi < (unsigned)F1;
        const unsigned base = 4u * (unsigned)(((b * F1 + fi) * T1 + 2 * p - 1) * CS + 4 * kq);
        const unsigned v0 = (row_ok && ok0) ? base : BUF_OOB, v1 = (row_ok && pair_ok) ? base + 4u * CS : BUF_OOB, v2 = (row_ok && ok2) ? base + 8u * CS : BUF_OOB;
        if (XSQ_L4F_ABL & 8) {
#pragma unroll
            for (int s = 0; s < 3; ++s) { xa[0][s] = make_float4(1.f, 2.f, 3.f, (float)v0); xa[1][s] = make_float4(2.f, 3.f, 1.f, (float)v1); xa[2][s] = make_float4(3.f, 1.f, 2.f, (float)v2); }
            xt[0] = 1.f; xt[1] = 2.f; xt[2] = 3.f;
            return;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            xa[0][s] = buf_ld4(rin, v0, 64 * s);
            xa[1][s] = buf_ld4(rin, v1, 64 * s);
            xa[2][s] = buf_ld4(rin, v2, 64 * s);
        }
        // channel 48 + kq: the lane's own word of the tail quad (base carries 4 kq channels = 16 kq bytes: 192 + 4 kq from the row = base + 192 - 12 kq)
        xt[0] = buf_ld1(rin, v0 - 12u * (unsigned)kq, 192);
        xt[1] = buf_ld1(rin, v1 - 12u * (unsigned)kq, 192);
        xt[2] = buf_ld1(rin, v2 - 12u * (unsigned)kq, 192);
    };

    f32x4 acc[3][NCB];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int bf = q * L4_BLD + 4 * kq;                          // weight tile: column q of a 16-column block, k-quad kq
    // TAPS (column tiles of <= 32 columns): any number of frequency taps, the next tap's weights and operands requested while
    // the tap computes.  Wider tiles hold 10 staging + 39 operand registers per tap in flight: they run ONE tap (every block of
    // the Bark-262 plan with more than one tap has W <= 24; cdae_launch_layer sends other plans to the implicit GEMM).
    constexpr bool TAPS = NCB <= 2;
    load_b(0);
    load_a(0);
    for (int df = 0; df < (TAPS ? kf : 1); ++df) {
        if (TAPS && df > 0) __syncthreads();                     // every wave is past the tap before: its weight tiles may go
        store_b();
        __syncthreads();
        if (TAPS && df + 1 < kf) load_b(df + 1);
        float4 xc[3][3];
        float xtc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xtc[i] = xt[i];
#pragma unroll
            for (int s = 0; s < 3; ++s) xc[i][s] = xa[i][s];
        }
        if (TAPS && df + 1 < kf) load_a(df + 1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float4 x0 = xc[0][s], x1 = xc[1][s], x2 = xc[2][s];
            const float d[3][4] = {{x0.x - x1.x, x0.y - x1.y, x0.z - x1.z, x0.w - x1.w}, {x1.x, x1.y, x1.z, x1.w},
                                   {x2.x - x1.x, x2.y - x1.y, x2.z - x1.z, x2.w - x1.w}};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* Bj = Bs + j * 64 * L4_BLD + bf + 16 * s;
                float4 w[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) w[cb] = *reinterpret_cast<const float4*>(&Bj[cb * 16 * L4_BLD]);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j][0], w[cb].x, acc[j][cb], 0, 0, 0);
                    acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j][1], w[cb].y, acc[j][cb], 0, 0, 0);
                    acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j][2], w[cb].z, acc[j][cb], 0, 0, 0);
                    acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j][3], w[cb].w, acc[j][cb], 0, 0, 0);
                }
            }
        }
        {
            const float dt[3] = {xtc[0] - xtc[1], xtc[1], xtc[2] - xtc[1]};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* Bj = Bs + j * 64 * L4_BLD + bf - 4 * kq + 48 + kq;      // channel 48 + kq of column q
                float wt[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) wt[cb] = Bj[cb * 16 * L4_BLD];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dt[j], wt[cb], acc[j][cb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: out[u] = m1 + m2, out[u + 1] = m2 + m3, + bias[c], sigmoid (the formula of CdaeL4Op), 4-byte buffer stores:
    // accumulator row r of this lane is pair 4 kq + r of the wave, its column n0 + 16 cb + q
    const float* bias = a.pool + t.bias_off;
    const float bias0 = bias[0], bias1 = bias[1];
    const __amdgpu_buffer_rsrc_t rm = buf_rsrc(a.masks + t.out_off, 0x40000000u);
    unsigned ob[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ob[r] = obase[wave * 16 + 4 * kq + r];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int n = n0 + 16 * cb + q;
        const int c = n >= hop ? 1 : 0;
        const float bs = c ? bias1 : bias0;
        const unsigned coff = n < W ? 4u * (unsigned)(c * F * ST + n - c * hop) : BUF_OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m1 = acc[0][cb][r], m2 = acc[1][cb][r], m3 = acc[2][cb][r];
            const float y0 = __builtin_amdgcn_rcpf(1.f + __expf(-((m1 + m2) + bs)));
            const float y1 = __builtin_amdgcn_rcpf(1.f + __expf(-((m2 + m3) + bs)));
            const unsigned vo = (ob[r] != 0xffffffffu && !((XSQ_L4F_ABL & 16) && y0 != 1.2345e-30f)) ? ob[r] + coff : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), rm, (int)vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), rm, (int)vo, 4 * hop, 0);
        }
    }
}

__global__ __launch_bounds__(256, XSQ_L4F_WAVES_PER_EU) void cdae_l4f_kernel(CdaeArgs a, const L4fTileDev* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float Bs[L4_BROWS * L4_BLD];
    __shared__ unsigned obase[L4_PAIRS];                         // byte offset of a pair's first output (channel 0, dt 0) inside the target's masks
    const L4fTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.F), "s"(t.F1), "s"(t.in_off), "s"(t.out_off), "s"(t.bias_off), "s"(t.u_off),
                 "s"(t.hop), "s"(t.n0), "s"(t.P));
    const int rem = l4f_cols(2 * t.hop) - t.n0;                  // (workgroup-uniform)
    if (rem >= 64) cdae_l4f_body<4>(a, t, Bs, obase);
    else if (rem == 48) cdae_l4f_body<3>(a, t, Bs, obase);
    else if (rem == 32) cdae_l4f_body<2>(a, t, Bs, obase);
    else cdae_l4f_body<1>(a, t, Bs, obase);
}

}  // namespace xsq
