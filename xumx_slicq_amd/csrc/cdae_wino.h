// Layers 2 and 3 of the CDAE (fp32) as Winograd F(2, 4) along the four TIME taps.
//
// Both layers are (kf x 4)-tap convolutions over 52-channel rows (/root/reference/xumx_slicq_v2/model.py:140-170; layer 3 is
// the transposed convolution written as a gather, cdae.hip): y[t] = sum_dt w[dt] . x[t + dt - PAD].  The slab kernels
// (cdae_slab.h) run that contraction on the fp32 matrix pipe at 0.72 of its peak, and on gfx950 the fp32 MFMAs share the
// SIMD's vector issue port with every other vector instruction (band_dft4.h, "Vector issue"): the only lever left on them
// is fewer MFMA flops.  F(2, 4) computes two neighbouring outputs (t, t + 1) from five inputs with FIVE products per
// (input channel, output channel) instead of eight:
//     V_j = sum_i BT[j][i] d_i        (input transform, five values from the five positions 2q .. 2q + 4)
//     M_j = V_j . U_j                 (U_j = sum_dt G[j][dt] w[dt]: transformed weights, made on the host in fp64)
//     y_0 = M_0 + M_1 + M_2 + M_3,    y_1 = M_1 - M_2 + 2 M_3 + M_4
// with the Cook-Toom matrices of the points {0, 1, -1, 2, inf}: BT has integer entries (exact products, the transform is
// nine fp32 adds / FMAs per channel and pair), the fractions live in G.  Measured on the CPU against fp64 (this layer's
// sizes, ReLU inputs): 1.9e-7 RMS where the direct fp32 sum has 7e-8 -- two orders of magnitude inside the parity bar.
//
// As a GEMM: rows = output PAIRS, five accumulator sets of (pairs x 52 columns), K = 52 channels per set and frequency
// tap: 5 * 52 = 260 products per pair and column where the direct form has 2 * 208 = 416 (0.625).
//
// Tile = 64 consecutive pairs of one batch item in the flattened (f, pair) space (at most two (b, f) rows: P >= 64 pairs per
// row), 256 threads = 4 waves x 16 pairs.  The tile's DISTINCT input positions (2 * 64 + 3 per touched row) sit once in LDS
// as two planes -- even and odd slab positions -- at a row stride of 52 words, so that the five 16-byte reads of a lane
// (its pair's positions 2q .. 2q + 4, four channels) are conflict-free.  The lane transforms them in registers (each
// transformed value feeds exactly one lane's MFMA operand: transforming at staging time would cost the same instructions
// and 2.5x the LDS) and runs, per component and 16-channel chunk, four v_mfma_f32_16x16x4_f32 per 16-column block with the
// k permutation of cdae_slab.h (MFMA i takes channel 4 kq + i of the chunk from k-quad kq).  Columns 48..50 are summed on
// the vector ALU from the same transformed values (as in the slab kernels).  The transformed weights of a (chunk,
// component) -- 52 columns x 16 k, 3.3 KB -- stream through a ring of five LDS tiles (slot = component), loaded four steps
// and written two steps ahead of their use.
//   LDS: 27.9 KB planes + 20.8 KB ring = 48.7 KB -> three 256-thread workgroups per CU.
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

#ifndef XSQ_WINO_WAVES_PER_EU
#define XSQ_WINO_WAVES_PER_EU 3
#endif

namespace xsq {

constexpr int WN_PAIRS = 64;                              // output pairs per tile (4 waves x 16)
constexpr int WN_MAXSEG = 2;                              // (b, f) rows a tile may touch (needs P >= WN_PAIRS)
constexpr int WN_EROWS = WN_PAIRS + 2 * WN_MAXSEG;        // even-plane rows: pairs + 2 per segment
constexpr int WN_OROWS = WN_PAIRS + 1 * WN_MAXSEG;        // odd-plane rows:  pairs + 1 per segment
constexpr int WN_POS = 2 * WN_PAIRS + 3 * WN_MAXSEG;      // slab positions of a tile (134)
constexpr int WN_BLD = 20;                                // ring tile row: 16 k + 4 pad words (conflict-free ds_read_b128)
constexpr int WN_BTILE = CS * WN_BLD;                     // words per ring tile
constexpr int WN_UFULL = 16 * CS;                         // words of a (chunk < 3, component) tile in global memory: [col][16 k]
constexpr int WN_UTAIL = 4 * CS;                          // chunk 3 = channels 48..51: [col][4 k]
constexpr int WN_UDF = 5 * (3 * WN_UFULL + WN_UTAIL);     // words per frequency tap: [chunk][component][col][k]
constexpr int WN_STEPS = 20;                              // (chunk, component) steps per frequency tap

// Cook-Toom F(2, 4), points {0, 1, -1, 2, inf}: G (5 x 4), applied to the folded weights on the host (cdae.hip)
static const double WN_G[5][4] = {{0.5, 0, 0, 0}, {-0.5, -0.5, -0.5, -0.5}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6},
                                  {1.0 / 6, 1.0 / 3, 2.0 / 3, 4.0 / 3}, {0, 0, 0, 1}};

// word offset of tile (chunk s, component j) inside a frequency tap's block of the transformed weights
__host__ __device__ constexpr int wino_u_off(int s, int j) { return s < 3 ? (s * 5 + j) * WN_UFULL : 15 * WN_UFULL + j * WN_UTAIL; }

struct WinoTileDev {               // 64 bytes: one scalar load
    int Q0, kf, Fo, Fi;            // first pair of the tile (f * P + q inside batch item b)
    int64_t in_off, out_off;       // input / output activations of the (block, target), relative to the layer's arenas
    int64_t shift_off, u_off;      // shift vector / transformed weights inside the pool
    int b, f0, q0, P;              // batch item, (f, q) of the first pair, pairs per (b, f) row = (To + 1) / 2
};
static_assert(sizeof(WinoTileDev) == 64, "WinoTileDev is meant to be one 64-byte scalar load");

// input transform of one channel: d[0..4] -> v[0..4] (BT of the header comment; 9 operations, FMAs spelled out so that
// every instantiation rounds alike)
__device__ __forceinline__ void wino_bt(float d0, float d1, float d2, float d3, float d4, float& v0, float& v1, float& v2, float& v3, float& v4) {
    v3 = d3 - d1;
    v1 = fmaf(-2.f, d1, d3 - d2);
    v2 = fmaf(2.f, d1, fmaf(-3.f, d2, d3));
    v0 = fmaf(2.f, d0 - d2, v3);
    v4 = fmaf(-2.f, v3, d4 - d2);
}

template <bool TRANSPOSED>
__global__ __launch_bounds__(256, XSQ_WINO_WAVES_PER_EU) void cdae_wino_kernel(CdaeArgs a, const WinoTileDev* __restrict__ tiles, int ntiles) {
#pragma clang fp contract(off)
    constexpr int PAD = TRANSPOSED ? 3 : 0;
    constexpr int NV = TRANSPOSED ? H1 - 48 : H2 - 48;           // real channels past 47: 2 (layer 3 -> 50) or 3 (layer 2 -> 51)
    constexpr int PLANE_O = WN_EROWS * CS;                       // word offset of the odd plane
    __shared__ __attribute__((aligned(16))) float slab[(WN_EROWS + WN_OROWS) * CS];
    __shared__ __attribute__((aligned(16))) float Bs[5 * WN_BTILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const WinoTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.Fo), "s"(t.Fi), "s"(t.in_off), "s"(t.out_off), "s"(t.shift_off), "s"(t.u_off),
                 "s"(t.b), "s"(t.f0), "s"(t.q0), "s"(t.P));
    const int kf = t.kf, Fo = t.Fo, Fi = t.Fi, P = t.P, b = t.b, f0 = t.f0, q0 = t.q0;
    const int To = TRANSPOSED ? a.T1 : a.T2, Ti = TRANSPOSED ? a.T2 : a.T1;
    const float* in = (TRANSPOSED ? a.act2 : a.act1) + t.in_off;
    const float* U = a.pool + t.u_off;

    const int npairs = min(WN_PAIRS, Fo * P - t.Q0);             // pairs of this tile that exist
    const int n0 = min(npairs, P - q0);                          // ... in segment 0 (row f0); the rest in segment 1 (row f0 + 1)

    // ---- slab staging: lane = (position lane p0 = tid / 13 of 19, channel quad c4 = tid % 13), load r -> slab position p0 + 19 r.
    // Segment i covers slab positions [A_i, A_i + 2 np_i + 3): local position j' is input position 2 qs_i - PAD + j' of input
    // row f0 + i -+ df and lands in plane j' & 1, row (j' >> 1) + (rows of the segments before).
    constexpr int SPL = 256 / (CS / 4);                          // position lanes (19)
    constexpr int NLD = (WN_POS + SPL - 1) / SPL;                // loads per lane and slab (8)
    const int s_p0 = tid / (CS / 4), s_c4 = tid - s_p0 * (CS / 4);
    const bool s_on = tid < SPL * (CS / 4);
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(in, 0x40000000u);    // (a (block, target)'s input is < 2^30 bytes: cdae_launch_layer)
    const int A1 = 2 * n0 + 3;
    auto stage_slab = [&](int df) {
        float4 v[NLD];
        int lds[NLD];
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int j = s_p0 + SPL * r;
            const int seg = j >= A1 ? 1 : 0;
            const int jj = j - (seg ? A1 : 0);
            const int np = seg ? npairs - n0 : n0;
            const int pos = 2 * (seg ? 0 : q0) - PAD + jj;           // input position
            const int fi = TRANSPOSED ? f0 + seg - df : f0 + seg + df;
            const bool exists = s_on && jj < 2 * np + 3 && np > 0;
            const bool inr = exists && (unsigned)pos < (unsigned)Ti && (unsigned)fi < (unsigned)Fi;
            const unsigned vo = 4u * (unsigned)(((b * Fi + fi) * Ti + pos) * CS + 4 * s_c4);
            v[r] = buf_ld4(rin, inr ? vo : BUF_OOB, 0);
            const int row = (jj >> 1) + (seg ? ((jj & 1) ? n0 + 1 : n0 + 2) : 0);
            lds[r] = exists ? ((jj & 1) ? PLANE_O : 0) + row * CS + 4 * s_c4 : -1;
        }
#pragma unroll
        for (int r = 0; r < NLD; ++r)
            if (lds[r] >= 0) *reinterpret_cast<float4*>(&slab[lds[r]]) = v[r];
    };

    // ---- weight stream: tile g = (df, chunk s, component j) -> ring slot j.  208 float4 per full tile, 52 per tail tile.
    float4 gb[5];
    const int b_row = tid >> 2, b_k4 = tid & 3;
    auto load_b = [&](int df, int s, int j) {
        if (s < 3) { if (tid < 208) gb[j] = *reinterpret_cast<const float4*>(U + (int64_t)df * WN_UDF + wino_u_off(s, j) + 4 * tid); }
        else if (tid < CS) gb[j] = *reinterpret_cast<const float4*>(U + (int64_t)df * WN_UDF + wino_u_off(3, j) + 4 * tid);
    };
    auto store_b = [&](int s, int j) {
        if (s < 3) { if (tid < 208) *reinterpret_cast<float4*>(&Bs[j * WN_BTILE + b_row * WN_BLD + 4 * b_k4]) = gb[j]; }
        else if (tid < CS) *reinterpret_cast<float4*>(&Bs[j * WN_BTILE + tid * WN_BLD]) = gb[j];
    };

    // ---- this lane's operands: pair pl of the tile, k-quad kq
    const int pl = wave * 16 + q;
    const int myseg = pl >= n0 ? 1 : 0;
    const int eb = (pl + 2 * myseg) * CS + 4 * kq;               // even plane: rows er, er + 1, er + 2 = positions 0, 2, 4 of the pair
    const int ob = PLANE_O + (pl + myseg) * CS + 4 * kq;         // odd plane: rows or, or + 1 = positions 1, 3
    const int bf = q * WN_BLD + 4 * kq;                          // ring tile: column q of a 16-column block, k-quad kq
    const int bv = 48 * WN_BLD + 4 * kq;                         // the vector columns' rows

    f32x4 acc[5][3];
    float accv[5][NV];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) accv[j][cc] = 0.f;
    }

    // ---- prologue: slab of tap 0, tiles 0 and 1 in LDS, tiles 2 and 3 requested
    load_b(0, 0, 0); load_b(0, 0, 1);
    stage_slab(0);
    store_b(0, 0); store_b(0, 1);
    load_b(0, 0, 2); load_b(0, 0, 3);
    __syncthreads();

    for (int df = 0; df < kf; ++df) {
        const bool more = df + 1 < kf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // raw positions of this lane's pair: four channels (chunk s < 3) or one (the tail chunk: channel 48 + kq)
            float d[5][4];
            if (s < 3) {
                const float4 e0 = *reinterpret_cast<const float4*>(&slab[eb + 16 * s]);
                const float4 o0 = *reinterpret_cast<const float4*>(&slab[ob + 16 * s]);
                const float4 e1 = *reinterpret_cast<const float4*>(&slab[eb + CS + 16 * s]);
                const float4 o1 = *reinterpret_cast<const float4*>(&slab[ob + CS + 16 * s]);
                const float4 e2 = *reinterpret_cast<const float4*>(&slab[eb + 2 * CS + 16 * s]);
                d[0][0] = e0.x; d[0][1] = e0.y; d[0][2] = e0.z; d[0][3] = e0.w;
                d[1][0] = o0.x; d[1][1] = o0.y; d[1][2] = o0.z; d[1][3] = o0.w;
                d[2][0] = e1.x; d[2][1] = e1.y; d[2][2] = e1.z; d[2][3] = e1.w;
                d[3][0] = o1.x; d[3][1] = o1.y; d[3][2] = o1.z; d[3][3] = o1.w;
                d[4][0] = e2.x; d[4][1] = e2.y; d[4][2] = e2.z; d[4][3] = e2.w;
            } else {
                const int te = eb - 4 * kq + 48 + kq, to = ob - 4 * kq + 48 + kq;
                d[0][0] = slab[te]; d[1][0] = slab[to]; d[2][0] = slab[te + CS]; d[3][0] = slab[to + CS]; d[4][0] = slab[te + 2 * CS];
            }
            constexpr int NI = 4;
            float v[5][4];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (s < 3 || i == 0) wino_bt(d[0][i], d[1][i], d[2][i], d[3][i], d[4][i], v[0][i], v[1][i], v[2][i], v[3][i], v[4][i]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                // step n = 5 s + j of this tap: the tile four steps ahead is requested, the tile two steps ahead (requested two
                // steps ago) goes into its ring slot -- last read three steps ago, two barriers back
                {
                    const int n4 = 5 * s + j + 4, n2 = 5 * s + j + 2;
                    if (n4 < WN_STEPS) load_b(df, n4 / 5, n4 % 5);
                    else if (more) load_b(df + 1, (n4 - WN_STEPS) / 5, (n4 - WN_STEPS) % 5);
                    if (n2 < WN_STEPS) store_b(n2 / 5, n2 % 5);
                    else if (more) store_b((n2 - WN_STEPS) / 5, (n2 - WN_STEPS) % 5);
                }
                const float* Bt = &Bs[j * WN_BTILE];
                if (s < 3) {
                    const float4 w0 = *reinterpret_cast<const float4*>(&Bt[bf]);
                    const float4 w1 = *reinterpret_cast<const float4*>(&Bt[bf + 16 * WN_BLD]);
                    const float4 w2 = *reinterpret_cast<const float4*>(&Bt[bf + 32 * WN_BLD]);
                    const float wa[4] = {w0.x, w0.y, w0.z, w0.w}, wb[4] = {w1.x, w1.y, w1.z, w1.w}, wc[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wa[i], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wb[i], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wc[i], acc[j][2], 0, 0, 0);
                    }
#pragma unroll
                    for (int cc = 0; cc < NV; ++cc) {
                        const float4 u = *reinterpret_cast<const float4*>(&Bt[bv + cc * WN_BLD]);
                        asm volatile("v_fmac_f32 %0, %1, %5\n\tv_fmac_f32 %0, %2, %6\n\tv_fmac_f32 %0, %3, %7\n\tv_fmac_f32 %0, %4, %8"
                                     : "+v"(accv[j][cc])
                                     : "v"(v[j][0]), "v"(v[j][1]), "v"(v[j][2]), "v"(v[j][3]), "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
                    }
                } else {
                    const int tb = bf - 4 * kq + kq;
                    acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][0], Bt[tb], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][0], Bt[tb + 16 * WN_BLD], acc[j][1], 0, 0, 0);
                    acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][0], Bt[tb + 32 * WN_BLD], acc[j][2], 0, 0, 0);
#pragma unroll
                    for (int cc = 0; cc < NV; ++cc) {
                        const float u = Bt[48 * WN_BLD + cc * WN_BLD + kq];
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accv[j][cc]) : "v"(v[j][0]), "v"(u));
                    }
                }
                __syncthreads();
            }
        }
        if (more) {                  // every wave is past the barrier behind the slab's last reader
            stage_slab(df + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: output transform, shift + ReLU, through a per-wave LDS image (the planes are free: the loop ended on a
    // barrier), out as 16-byte stores.  Image row 2 p + r = output r of the wave's pair p.
    float* img = slab + wave * 32 * CS;
    const float* shift = a.pool + t.shift_off;
    {
        const int rq = lane >> 4;
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            const int col = 16 * cb + q;
            const float sh = shift[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][cb][r], m1 = acc[1][cb][r], m2 = acc[2][cb][r], m3 = acc[3][cb][r], m4 = acc[4][cb][r];
                const float y0 = ((m0 + m1) + (m2 + m3));
                const float y1 = fmaf(2.f, m3, m1 - m2) + m4;
                img[(2 * (4 * rq + r)) * CS + col] = fmaxf(y0 + sh, 0.f);
                img[(2 * (4 * rq + r) + 1) * CS + col] = fmaxf(y1 + sh, 0.f);
            }
        }
        float y0v[4] = {0.f, 0.f, 0.f, 0.f}, y1v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) {
            float m[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {            // the four k-quads' partial sums meet here (fixed order)
                float x = accv[j][cc];
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                m[j] = x;
            }
            y0v[cc] = ((m[0] + m[1]) + (m[2] + m[3]));
            y1v[cc] = fmaf(2.f, m[3], m[1] - m[2]) + m[4];
        }
        if (kq == 0) {
            const float4 sh = *reinterpret_cast<const float4*>(shift + 48);
            *reinterpret_cast<float4*>(img + (2 * q) * CS + 48) =
                make_float4(fmaxf(y0v[0] + sh.x, 0.f), fmaxf(y0v[1] + sh.y, 0.f), fmaxf(y0v[2] + sh.z, 0.f), fmaxf(y0v[3] + sh.w, 0.f));
            *reinterpret_cast<float4*>(img + (2 * q + 1) * CS + 48) =
                make_float4(fmaxf(y1v[0] + sh.x, 0.f), fmaxf(y1v[1] + sh.y, 0.f), fmaxf(y1v[2] + sh.z, 0.f), fmaxf(y1v[3] + sh.w, 0.f));
        }
    }
    __builtin_amdgcn_wave_barrier();
    // image row -> output row: pair pl = 16 wave + (row >> 1) of the tile, segment by n0, t = 2 q + (row & 1); rows of pairs that
    // do not exist and the phantom second row of an odd To are switched out of the descriptor's range
    float* out = (TRANSPOSED ? a.act3 : a.act2) + t.out_off;
    const __amdgpu_buffer_rsrc_t ro = buf_rsrc(out, 0x40000000u);
#pragma unroll
    for (int it = 0; it < (32 * (CS / 4) + 63) / 64; ++it) {
        const int slot = lane + 64 * it;
        const int row = slot / (CS / 4), c4 = slot - row * (CS / 4);
        const int p = wave * 16 + (row >> 1);
        const int sg = p >= n0 ? 1 : 0;
        const int qq = sg ? p - n0 : q0 + p;
        const int tt = 2 * qq + (row & 1);
        const bool ok = slot < 32 * (CS / 4) && p < npairs && tt < To;
        const unsigned vo = 4u * (unsigned)((((b * Fo + f0 + sg) * To) + tt) * CS + 4 * c4);
        const float4 val = *reinterpret_cast<const float4*>(img + 4 * min(slot, 32 * (CS / 4) - 1));
        buf_st4(val, ro, ok ? vo : BUF_OOB, 0);
    }
}

}  // namespace xsq
