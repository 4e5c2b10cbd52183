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
// Row tile = 64 consecutive output pairs of the flattened (b, f, pair) space x one column tile of 16 NCB <= 64 columns;
// 256 threads = 4 waves x 16 pairs; a workgroup runs `run` consecutive row tiles of one column tile.  To = 2 S is even:
// every pair has both outputs.
//  * B operand: the column tile's three weight tiles of ONE frequency tap ([component][column][52 k], 13 KB per 16 columns)
//    go into LDS whole, once per tap -- for the 67 of 70 blocks with one tap they stay there for the workgroup's whole run of row
//    tiles: no barrier, no weight traffic and no exposed load (the next row tile's operands are requested a tile ahead).
//    Row stride 56 words (cdae_wino.h explains the bank pattern of the 16-byte reads).
//  * A operand: not staged (cdae_l1f.h): a lane (pair q, k-quad kq) loads channels 4 kq .. 4 kq + 3 of each chunk and channel
//    48 + kq of its pair's three positions 2 p - 1, 2 p, 2 p + 1 straight into the MFMA layout -- nine 16-byte and three
//    4-byte buffer loads per tap, all in flight together --, two subtractions per value make the three operands.
//  * per tap and wave: 39 NCB v_mfma_f32_16x16x4_f32 (4,992 cycles at NCB = 4) where the GEMM issues 7,168 for the same outputs.
//  * the weights are the MFMA's ROW operand: a lane's four accumulator registers are four consecutive outputs of its own pair's
//    row -- epilogue: output sums, bias, sigmoid, one 16-byte buffer store per quad and output (8-byte halves when hop % 4 == 2).
//   LDS 43.0 KB: three workgroups per CU.
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

#ifndef XSQ_L4F_WAVES_PER_EU
#define XSQ_L4F_WAVES_PER_EU 3
#endif
#ifndef XSQ_L4F_SCHED
#define XSQ_L4F_SCHED 0     // what may cross the scheduling barrier behind a group of MFMAs: nothing (measured: 0.535 ms; everything but LDS
                            // operations, 0x7e: 0.551; no barrier at all, 0x3ff: 0.550 -- profiles/r11_ab_runs.txt r11p)
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
    int Q0, kf, F, run;            // first pair of the run in the flattened (b, f, pair) space; taps; output rows; row tiles of 64 pairs this
                                   // workgroup runs (1 with several taps)
    int64_t in_off, out_off;       // the (block, target)'s act3, in floats / its masks, in floats (= its estimates inside Y, in float2)
    int64_t bias_off, u_off;       // output bias (2) inside the pool / the column tile's weights of tap 0 inside the Winograd pool
    int hop, n0, P, x_off;         // hop, first column of the tile, pairs per (b, f) row = S, the block's mix coefficients inside X in float2
};
static_assert(sizeof(L4fTileDev) == 64, "L4fTileDev is meant to be one 64-byte scalar load");

template <int NCB, bool WITH_Y, bool RES = false>
__device__ __forceinline__ void cdae_l4f_body(const CdaeArgs& a, const L4fTileDev& t, float* const Bs) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const int kf = t.kf, F = t.F, F1 = t.F - t.kf + 1, P = t.P, hop = t.hop, n0 = t.n0, run = t.run;
    const int T1 = a.T1, W = 2 * hop;
    const int ST = a.S * W;                                      // floats of one (b, c, f) output row
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.act3 + t.in_off, 4u * (unsigned)(a.Bn * F1 * T1 * CS));   // (< 2^30 bytes: cdae_launch_layer)
    const int pl = wave * 16 + q;

    // ---- weight tiles of a tap -> LDS.  Global: [component][column < 16 NCB][52 k] = 39 x 16 NCB float4; float4 x -> LDS row
    // component * 64 + column, words 4 (x % 13) ..
    // RES (the multi-tap launch, column tiles of <= 32 columns): the weight tiles of ALL taps stay in LDS, compact -- rows
    // (tap * 3 + component) * 16 NCB + column; otherwise one tap, rows component * 64 + column
    constexpr int JS = RES ? 16 * NCB : 64, TAPW = 3 * JS * L4_BLD;
    constexpr int NB4 = 3 * 16 * NCB * (CS / 4);
    constexpr int NLD = (NB4 + 255) / 256;
    const int tapf = l4f_tap_floats(W);
    const __amdgpu_buffer_rsrc_t ru = buf_rsrc(a.upool + t.u_off, 4u * (unsigned)((kf - 1) * tapf + 3 * CS * 16 * NCB));
    int b_lds[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int x = tid + 256 * r, row = x / (CS / 4), k4 = x - row * (CS / 4);
        const int j = row / (16 * NCB), col = row - j * (16 * NCB);
        b_lds[r] = x < NB4 ? (j * JS + col) * L4_BLD + 4 * k4 : -1;
    }
    float4 gb[NLD];
    auto load_b = [&](int df) {
        if (XSQ_L4F_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < NLD; ++r) gb[r] = buf_ld4(ru, (r < NLD - 1 || tid + 256 * r < NB4) ? 16u * (unsigned)(tid + 256 * r) : BUF_OOB, 4 * df * tapf);
    };
    auto store_b = [&](int tap = 0) {
        if (XSQ_L4F_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < NLD; ++r)
            if (r < NLD - 1 || b_lds[r] >= 0) *reinterpret_cast<float4*>(&Bs[tap * TAPW + b_lds[r]]) = gb[r];
    };

    // ---- a lane's pair of row tile `it` of the run: pair Qg of the flattened (b, f, pair) space
    struct Pair { int b, f, p; bool ok; };
    auto pair_of = [&](int it) {
        Pair pr;
        const int Qg = t.Q0 + it * L4_PAIRS + pl;
        pr.ok = Qg < a.Bn * F * P;
        pr.b = Qg / (F * P);
        const int Q = Qg - pr.b * (F * P);
        pr.f = Q / P; pr.p = Q - pr.f * P;
        return pr;
    };
    // ---- operands of (pair, tap): input row f - df, positions 2 p - 1 (exists for p > 0), 2 p, 2 p + 1 (exists below T1 =
    // 2 S - 1: not for the row's last pair), channels 16 s + 4 kq .. (s < 3) and 48 + kq
    struct Ops { float4 x[3][3]; float xt[3]; };
    struct Addr { unsigned v0, v1, v2; };                        // byte offsets of the three positions' channel quad 4 kq (BUF_OOB: reads as zero)
    auto addr_of = [&](const Pair& pr, int df) {
        const int fi = pr.f - df;
        const bool row_ok = pr.ok && (unsigned)fi < (unsigned)F1;
        const unsigned base = 4u * (unsigned)(((pr.b * F1 + fi) * T1 + 2 * pr.p - 1) * CS + 4 * kq);
        Addr ad;
        ad.v0 = (row_ok && pr.p > 0) ? base : BUF_OOB;
        ad.v1 = row_ok ? base + 4u * CS : BUF_OOB;
        ad.v2 = (row_ok && 2 * pr.p + 1 < T1) ? base + 8u * CS : BUF_OOB;
        return ad;
    };
    // chunk s < 3: channels 16 s + 4 kq ..; s == 3: channel 48 + kq, the lane's own word of the tail quad (the offsets carry 4 kq
    // channels = 16 kq bytes: 192 + 4 kq from the row = offset + 192 - 12 kq)
    auto load_chunk_a = [&](Ops& o, const Addr& ad, int s) {
        if (XSQ_L4F_ABL & 8) {
            if (s < 3) { o.x[0][s] = make_float4(1.f, 2.f, 3.f, (float)ad.v0); o.x[1][s] = make_float4(2.f, 3.f, 1.f, (float)ad.v1); o.x[2][s] = make_float4(3.f, 1.f, 2.f, (float)ad.v2); }
            else { o.xt[0] = 1.f; o.xt[1] = 2.f; o.xt[2] = 3.f; }
            return;
        }
        if (s < 3) {
            o.x[0][s] = buf_ld4(rin, ad.v0, 64 * s);
            o.x[1][s] = buf_ld4(rin, ad.v1, 64 * s);
            o.x[2][s] = buf_ld4(rin, ad.v2, 64 * s);
        } else {
            o.xt[0] = buf_ld1(rin, ad.v0 - 12u * (unsigned)kq, 192);
            o.xt[1] = buf_ld1(rin, ad.v1 - 12u * (unsigned)kq, 192);
            o.xt[2] = buf_ld1(rin, ad.v2 - 12u * (unsigned)kq, 192);
        }
    };
    auto load_a = [&](Ops& o, const Addr& ad) {
#pragma unroll
        for (int s = 0; s < 4; ++s) load_chunk_a(o, ad, s);
    };

    f32x4 acc[3][NCB];
    auto clear_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const int bf = q * L4_BLD + 4 * kq;                          // weight tile: column q of a 16-column block, k-quad kq
    // one tap of one row tile against the weight tiles in LDS: nine groups (chunk s, component j) of 4 NCB MFMAs and the tail
    // group.  The fragments of group g + 1 are read while group g computes, and a scheduling barrier closes every group: left
    // alone the compiler hoists the fragment reads of ALL groups to the top (36 NCB registers: 205 spilled registers in the
    // 64-column body, measured by tools/isa_budget.py's probe kernels).
    // `refill`: chunk s of the NEXT operand set (`nx`: the next row tile of the run, or the next tap) is requested into the
    // registers of chunk s as soon as that chunk has been transformed -- one operand set in registers instead of two (a second
    // set in flight cost 39 registers: 58 spilled in the 64-column body); the request is a whole row tile ahead of its use.
    auto contract = [&](Ops& o, const Addr& nx, bool refill, int tap = 0) {
        const float* const Bt = Bs + tap * TAPW;
        float4 w[2][NCB];
        auto read_w = [&](int g, float4 (&dst)[NCB]) {
            const int s = g / 3, j = g - 3 * s;
            const float* Bj = Bt + j * JS * L4_BLD + bf + 16 * s;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) dst[cb] = *reinterpret_cast<const float4*>(&Bj[cb * 16 * L4_BLD]);
        };
        read_w(0, w[0]);
        float d[3][4];
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int s = g / 3, j = g - 3 * s;
            if (j == 0) {
                const float4 x0 = o.x[0][s], x1 = o.x[1][s], x2 = o.x[2][s];
                d[0][0] = x0.x - x1.x; d[0][1] = x0.y - x1.y; d[0][2] = x0.z - x1.z; d[0][3] = x0.w - x1.w;
                d[1][0] = x1.x; d[1][1] = x1.y; d[1][2] = x1.z; d[1][3] = x1.w;
                d[2][0] = x2.x - x1.x; d[2][1] = x2.y - x1.y; d[2][2] = x2.z - x1.z; d[2][3] = x2.w - x1.w;
            }
            if (j == 0 && refill) load_chunk_a(o, nx, s);
            if (g < 8) read_w(g + 1, w[(g + 1) & 1]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                // (the WEIGHTS are the MFMA's row operand: accumulator register r of a lane is column 4 kq + r of the block for the
                //  lane's OWN pair q -- four consecutive outputs of one row: 16-byte stores, no exchange of output offsets)
                const float4 wv = w[g & 1][cb];
                acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, d[j][0], acc[j][cb], 0, 0, 0);
                acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, d[j][1], acc[j][cb], 0, 0, 0);
                acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, d[j][2], acc[j][cb], 0, 0, 0);
                acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, d[j][3], acc[j][cb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(XSQ_L4F_SCHED);
        }
        const float dt[3] = {o.xt[0] - o.xt[1], o.xt[1], o.xt[2] - o.xt[1]};
        if (refill) load_chunk_a(o, nx, 3);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float* Bj = Bt + j * JS * L4_BLD + bf - 4 * kq + 48 + kq;      // channel 48 + kq of column q
            float wt[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) wt[cb] = Bj[cb * 16 * L4_BLD];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[cb], dt[j], acc[j][cb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(XSQ_L4F_SCHED);
    };
    const float* bias = a.pool + t.bias_off;
    const float bias0 = bias[0], bias1 = bias[1];
    const __amdgpu_buffer_rsrc_t rm = buf_rsrc(a.masks ? a.masks + t.out_off : a.pool, a.masks ? 0x40000000u : 0u);
    // WITH_Y (the module API, Unmix.forward: the estimates are materialised): Y = mask * X, a rounded product per component as
    // CdaeL4Op stores it and as the masked decoder forms it -- the masks-only call and this one stay bitwise interchangeable
    const __amdgpu_buffer_rsrc_t rx = buf_rsrc(WITH_Y ? reinterpret_cast<const float2*>(a.X) + t.x_off : nullptr, WITH_Y ? 0x80000000u : 0u);
    const __amdgpu_buffer_rsrc_t ry = buf_rsrc(WITH_Y ? reinterpret_cast<float2*>(a.Y) + t.out_off : nullptr, WITH_Y ? 0x80000000u : 0u);
    const bool quads = (hop & 3) == 0;                           // (uniform)
    // the values of u (y0) and of u + 1 (y1) at element offset om.  The hop's displacement of u + 1 is ADDED TO THE LANE OFFSET,
    // not passed as the buffer instruction's scalar offset: a 16-byte buffer store whose scalar offset sits in an SGPR lost its
    // FIRST data dword to the vector instruction behind it -- 16 lanes of a wave stored the integer a following v_add had just
    // put into that register (the next row tile's pair index) instead of mask * X, in 1-3 of 30 runs, only in the run-of-tiles
    // form (profiles/r11_ab_runs.txt, r11o).  The ISA manual's store-data hazard (a wait state between a > 64-bit store and a
    // write of its data registers) is documented -- and handled by the compiler -- only for stores WITHOUT an SGPR offset;
    // with the offset in the lane operand the compiler places that wait state.
    auto emit4 = [&](int om, bool ok, float4 y0, float4 y1) {
        if (!WITH_Y || a.masks) {
            const unsigned vo = ok ? 4u * (unsigned)om : BUF_OOB;
            buf_st4(y0, rm, vo, 0);
            buf_st4(y1, rm, vo + 4u * (unsigned)hop, 0);
        }
        if constexpr (WITH_Y) {
            // (the mix and the estimates are 8 bytes per element: ranges of up to 2^31 bytes -- the launch checks it -- and BUF_OOB =
            //  2^31 as the switched-off offset: the sums below must not wrap past 2^32 back into the range)
            const unsigned vx = ok ? 8u * (unsigned)om : BUF_OOB, vx1 = vx + 8u * (unsigned)hop;
            const float4 xa0 = buf_ld4(rx, vx, 0), xa1 = buf_ld4(rx, vx, 16), xb0 = buf_ld4(rx, vx1, 0), xb1 = buf_ld4(rx, vx1, 16);
            buf_st4(make_float4(y0.x * xa0.x, y0.x * xa0.y, y0.y * xa0.z, y0.y * xa0.w), ry, vx, 0);
            buf_st4(make_float4(y0.z * xa1.x, y0.z * xa1.y, y0.w * xa1.z, y0.w * xa1.w), ry, vx, 16);
            buf_st4(make_float4(y1.x * xb0.x, y1.x * xb0.y, y1.y * xb0.z, y1.y * xb0.w), ry, vx1, 0);
            buf_st4(make_float4(y1.z * xb1.x, y1.z * xb1.y, y1.w * xb1.z, y1.w * xb1.w), ry, vx1, 16);
        }
    };
    auto emit2 = [&](int om, bool ok, float2 y0, float2 y1) {
        if (!WITH_Y || a.masks) {
            const unsigned vo = ok ? 4u * (unsigned)om : BUF_OOB;
            buf_st2(y0, rm, vo, 0);
            buf_st2(y1, rm, vo + 4u * (unsigned)hop, 0);
        }
        if constexpr (WITH_Y) {
            const unsigned vx = ok ? 8u * (unsigned)om : BUF_OOB, vx1 = vx + 8u * (unsigned)hop;
            const float4 xa0 = buf_ld4(rx, vx, 0), xb0 = buf_ld4(rx, vx1, 0);
            buf_st4(make_float4(y0.x * xa0.x, y0.x * xa0.y, y0.y * xa0.z, y0.y * xa0.w), ry, vx, 0);
            buf_st4(make_float4(y1.x * xb0.x, y1.x * xb0.y, y1.y * xb0.z, y1.y * xb0.w), ry, vx1, 0);
        }
    };
    auto epilogue = [&](const Pair& pr) {
        const int ob = (pr.b * 2 * F + pr.f) * ST + 2 * pr.p * hop;          // element offset of (channel 0, dt 0) of the pair's first output
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int n = n0 + 16 * cb + 4 * kq;                 // first column of the quad
            float y0[4], y1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bs = n + r >= hop ? bias1 : bias0;
                const float m1 = acc[0][cb][r], m2 = acc[1][cb][r], m3 = acc[2][cb][r];
                y0[r] = __builtin_amdgcn_rcpf(1.f + __expf(-((m1 + m2) + bs)));
                y1[r] = __builtin_amdgcn_rcpf(1.f + __expf(-((m2 + m3) + bs)));
            }
            const bool live = pr.ok && !((XSQ_L4F_ABL & 16) && y0[0] != 1.2345e-30f);
            if (quads) {
                const int c = n >= hop ? 1 : 0;
                emit4(ob + c * F * ST + n - c * hop, live && n < W, make_float4(y0[0], y0[1], y0[2], y0[3]), make_float4(y1[0], y1[1], y1[2], y1[3]));
            } else {
                const int c0 = n >= hop ? 1 : 0, c1 = n + 2 >= hop ? 1 : 0;
                emit2(ob + c0 * F * ST + n - c0 * hop, live && n < W, make_float2(y0[0], y0[1]), make_float2(y1[0], y1[1]));
                emit2(ob + c1 * F * ST + n + 2 - c1 * hop, live && n + 2 < W, make_float2(y0[2], y0[3]), make_float2(y1[2], y1[3]));
            }
        }
    };

    if constexpr (RES) {
        // the multi-tap blocks on a launch of their own (64.5 KB of LDS: two workgroups per CU, which the one-tap tiles do not want:
        // r11z): all kf taps' weight tiles resident for the workgroup's run of row tiles -- no barrier and no weight traffic
        // inside the run; the operand set is refilled in place for the next (row tile, tap)
        Ops o;
        Pair pr = pair_of(0);
        for (int df = 0; df < kf; ++df) { load_b(df); store_b(df); }
        load_a(o, addr_of(pr, 0));
        __syncthreads();
        for (int it = 0; it < run; ++it) {
            const bool more = it + 1 < run;
            const Pair nxp = pair_of(more ? it + 1 : it);
            clear_acc();
            for (int df = 0; df < kf; ++df) {
                const bool last = df + 1 == kf;
                contract(o, last ? addr_of(nxp, 0) : addr_of(pr, df + 1), !last || more, df);
            }
            epilogue(pr);
            pr = nxp;
        }
        return;
    }
    if (kf == 1) {
        // ONE frequency tap (67 of the 70 Bark-262 blocks): the column tile's weights go into LDS once and STAY for the run of
        // `run` consecutive row tiles this workgroup owns; the next row tile's operands are requested chunk by chunk into the
        // registers the current one has just transformed -- no barrier and no weight traffic inside the run.  (One row tile per workgroup, the first form, spent half
        // its time in the prologue: without the weight loads -27 %, without the operand loads -21 %, without both -50 %,
        // profiles/r11_ab_runs.txt r11j.)
        Ops o;
        Pair pr = pair_of(0);
        load_b(0);
        load_a(o, addr_of(pr, 0));
        store_b();
        __syncthreads();
        for (int it = 0; it < run; ++it) {
            const bool more = it + 1 < run;
            const Pair nxp = pair_of(more ? it + 1 : it);
            const Addr nx = addr_of(nxp, 0);
            clear_acc();
            contract(o, nx, more);
            epilogue(pr);
            pr = nxp;
        }
        return;
    }
    // several frequency taps (column tiles of <= 32 columns; every Bark-262 block with more than one tap has W <= 24, other plans
    // go to the implicit GEMM: cdae_launch_layer): one row tile, the weight tiles re-staged per tap, the next tap's weights and
    // operands requested while the tap computes
    if constexpr (NCB <= 2) {
        Ops o;
        const Pair pr = pair_of(0);
        clear_acc();
        load_b(0);
        load_a(o, addr_of(pr, 0));
        for (int df = 0; df < kf; ++df) {
            if (df > 0) __syncthreads();                         // every wave is past the tap before: its weight tiles may go
            store_b();
            __syncthreads();
            const bool more = df + 1 < kf;
            if (more) load_b(df + 1);
            contract(o, addr_of(pr, more ? df + 1 : df), more);
        }
        epilogue(pr);
    }
}

template <bool WITH_Y>
__global__ __launch_bounds__(256, XSQ_L4F_WAVES_PER_EU) void cdae_l4f_kernel(CdaeArgs a, const L4fTileDev* __restrict__ tiles, int ntiles) {
#ifndef XSQ_L4F_LDS_ROWS
#define XSQ_L4F_LDS_ROWS L4_BROWS
#endif
    __shared__ __attribute__((aligned(16))) float Bs[XSQ_L4F_LDS_ROWS * L4_BLD];
    const L4fTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.F), "s"(t.run), "s"(t.in_off), "s"(t.out_off), "s"(t.bias_off), "s"(t.u_off),
                 "s"(t.hop), "s"(t.n0), "s"(t.P), "s"(t.x_off));
    const int rem = l4f_cols(2 * t.hop) - t.n0;                  // (workgroup-uniform)
    if (rem >= 64) cdae_l4f_body<4, WITH_Y>(a, t, Bs);
    else if (rem == 48) cdae_l4f_body<3, WITH_Y>(a, t, Bs);
    else if (rem == 32) cdae_l4f_body<2, WITH_Y>(a, t, Bs);
    else cdae_l4f_body<1, WITH_Y>(a, t, Bs);
}

// the multi-tap blocks' launch (column tiles of <= 32 columns, all taps resident: up to 3 taps x 96 rows or 5 taps x 48 rows of 56 words)
constexpr int L4_RES_ROWS = 288;
template <bool WITH_Y>
__global__ __launch_bounds__(256, 2) void cdae_l4f_taps_kernel(CdaeArgs a, const L4fTileDev* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float Bs[L4_RES_ROWS * L4_BLD];
    const L4fTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.F), "s"(t.run), "s"(t.in_off), "s"(t.out_off), "s"(t.bias_off), "s"(t.u_off),
                 "s"(t.hop), "s"(t.n0), "s"(t.P), "s"(t.x_off));
    const int rem = l4f_cols(2 * t.hop) - t.n0;                  // (workgroup-uniform; <= 32 by construction of the tile table)
    if (rem == 32) cdae_l4f_body<2, WITH_Y, true>(a, t, Bs);
    else cdae_l4f_body<1, WITH_Y, true>(a, t, Bs);
}

}  // namespace xsq
