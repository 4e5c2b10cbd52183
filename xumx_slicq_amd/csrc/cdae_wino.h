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
// as two planes -- even and odd slab positions -- at a row stride of 56 words (14 sixteen-byte slots), the weight tiles at
// 24 words (6 slots): a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS), i.e.
// MFMA rows 0-3 and 12-15 of one k-quad together with rows 4-11 of the next -- with a row stride of 2 x odd slots the first
// set lands on the even slots of the 256-byte bank row and the second, one slot further, on the odd ones.  (At 13 and 5
// slots per row, conflict-free for CONTIGUOUS groups of 16 lanes, the counters read 3.3-3.8 conflict cycles per LDS
// instruction: profiles/r08_ab_runs.txt.)  The lane transforms them in registers (each
// transformed value feeds exactly one lane's MFMA operand: transforming at staging time would cost the same instructions
// and 2.5x the LDS) and runs, per component and 16-channel chunk, four v_mfma_f32_16x16x4_f32 per 16-column block with the
// k permutation of cdae_slab.h (MFMA i takes channel 4 kq + i of the chunk from k-quad kq).  Columns 48..50 are summed on
// the vector ALU from the same transformed values (as in the slab kernels).  A frequency tap is three chunks of 16, 16 and
// 16 + 4 channels; the transformed weights of a chunk -- five components x 51 columns x 16 (20) k, 16 (20) KB -- stream
// through two LDS buffers: requested at the start of the chunk before, written at its end, ONE barrier per chunk; the next
// tap's slab is requested during the tap's last chunk and written behind its barrier.  (Several consecutive sub-tiles per
// workgroup, the next one's slab and weights requested during the last chunk of the one before, cost the chunk loop its
// registers and measured SLOWER -- profiles/r08_ab_runs.txt, r08h / r08i -- and so did a loader-wave form, r08n / r08o: removed.)
//   LDS: 30.0 KB planes + 2 x 25.0 KB weight tiles = 79.9 KB -> two 256-thread workgroups per CU, 256 registers each.
#pragma once
#include "cdae_api.h"
#include "gemm_tile.h"

#ifndef XSQ_WINO_WAVES_PER_EU
#define XSQ_WINO_WAVES_PER_EU 2
#endif
#ifndef XSQ_WINO_RAW_AHEAD
#define XSQ_WINO_RAW_AHEAD 0   // 1: the pair's raw positions of chunk s + 1 are read while chunk s computes (20 more registers; round 5's default.
                               // With the scheduling barriers of round 6: 0.855 / 0.872 against 0.836 / 0.861 without it, r11v)
#endif
#ifndef XSQ_WINO_SWAP
#define XSQ_WINO_SWAP 1     // 1: the WEIGHTS are the MFMA's row operand -- a lane's accumulator registers are four consecutive output channels
                            // of its own pair: 16-byte stores straight from registers; 0: pairs as rows, outputs through a per-wave LDS image
#endif
#ifndef XSQ_WINO_SCHED
#define XSQ_WINO_SCHED 0    // >= 0: a scheduling barrier with this mask behind every component's MFMAs; -1: none.  With the weights as the row
                            // operand (XSQ_WINO_SWAP) mask 0 measured 0.841-0.843 / 0.859-0.863 ms against 0.889-0.908 / 0.878-0.891 for the
                            // round-5 form; either change alone: nothing (profiles/r11_ab_runs.txt r11r, r11t)
#endif
#ifndef XSQ_WINO_SCHED2
#define XSQ_WINO_SCHED2 -1  // >= 0: ... and one behind the NEXT component's fragment reads, which pins them in front of this component's MFMAs (the
                            // compiler otherwise sinks them to the last MFMA of the group and waits one MFMA later); A/B r11s2
#endif
#ifndef XSQ_WINO_ABL
#define XSQ_WINO_ABL 0      // diagnostic builds (wrong results, timings only): 2 no vector columns, 4 no weight stream, 8 no slab loads, 16 no epilogue stores, 32 no input transform
#endif

namespace xsq {

#ifndef XSQ_WINO_STAMPS
#define XSQ_WINO_STAMPS 0    // diagnostic build: thread 0 of every workgroup adds its phase times (s_memrealtime ticks of 10 ns) to g_wn_stamps:
                             // 0 prologue, 1 chunk loop, 2 epilogue, 3 tiles, 4 of the loop: thread 0 waiting at its barriers (tools/w4_stamps.py)
#endif
#if XSQ_WINO_STAMPS
__device__ unsigned long long g_wn_stamps[8];
#define WN_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_wn_stamps[i], now_ - wn_t0); wn_t0 = now_; } } while (0)
#define WN_SUB(i, expr) do { const unsigned long long a_ = wall_clock64(); expr; if (tid == 0) atomicAdd(&g_wn_stamps[i], wall_clock64() - a_); } while (0)
#else
#define WN_STAMP(i) do { } while (0)
#define WN_SUB(i, expr) do { expr; } while (0)
#endif

constexpr int WN_PAIRS = 64;                              // output pairs per tile (4 waves x 16)
constexpr int WN_MAXSEG = 2;                              // (b, f) rows a tile may touch (needs P >= WN_PAIRS)
constexpr int WN_EROWS = WN_PAIRS + 2 * WN_MAXSEG;        // even-plane rows: pairs + 2 per segment
constexpr int WN_OROWS = WN_PAIRS + 1 * WN_MAXSEG;        // odd-plane rows:  pairs + 1 per segment
constexpr int WN_POS = 2 * WN_PAIRS + 3 * WN_MAXSEG;      // slab positions of a tile (134)
constexpr int WN_SLD = 56;                                // plane row stride in words: 52 channels + 4 pad (14 slots: header comment)
constexpr int WN_BLD = 24;                                // weight tile row: 16 k | 4 tail k | 4 pad words (6 slots)
constexpr int WN_COLS = 51;                               // columns that are staged: 48 on the matrix pipe + up to 3 on the vector ALU
constexpr int WN_BTILE = WN_COLS * WN_BLD;                // words per component tile in LDS (row = component * 51 + column)
constexpr int WN_U16 = 5 * WN_COLS * 16;                  // words of a chunk's main part in global memory: [component][col][16 k] = 1020 float4
constexpr int WN_UT = 5 * WN_COLS * 4;                    // the tail channels 48..51: [component][col][4 k] = 255 float4 (staged with chunk 2)
constexpr int WN_UDF = 3 * WN_U16 + WN_UT;                // words per frequency tap: three main parts, then the tail
// (float4 x of a main part lands at LDS row x >> 2, words 4 (x & 3) ..; float4 x of the tail at row x, words 16..19: no division
//  anywhere in the staging)

// Cook-Toom F(2, 4), points {0, 1, -1, 2, inf}: G (5 x 4), applied to the folded weights on the host (cdae.hip)
static const double WN_G[5][4] = {{0.5, 0, 0, 0}, {-0.5, -0.5, -0.5, -0.5}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6},
                                  {1.0 / 6, 1.0 / 3, 2.0 / 3, 4.0 / 3}, {0, 0, 0, 1}};

// word offset of (component j, column col, input channel ci) inside a frequency tap's block of the transformed weights
__host__ __device__ constexpr int wino_u_off(int j, int col, int ci) {
    return ci < 48 ? (ci / 16) * WN_U16 + (j * WN_COLS + col) * 16 + ci % 16 : 3 * WN_U16 + (j * WN_COLS + col) * 4 + (ci - 48);
}

struct WinoTileDev {               // 64 bytes: one scalar load
    int Q0, kf, Fo, Fi;            // first pair of the tile (f * P + q inside batch item b)
    int64_t in_off, out_off;       // input / output activations of the (block, target), relative to the layer's arenas
    int64_t shift_off, u_off;      // shift vector inside the pool / transformed weights inside the Winograd pool
    int b, pad0, pad1, P;          // batch item, -, -, pairs per (b, f) row = (To + 1) / 2
};
static_assert(sizeof(WinoTileDev) == 64, "WinoTileDev is meant to be one 64-byte scalar load");

// input transform of one channel: d[0..4] -> v[0..4] (BT of the header comment; 9 operations, FMAs spelled out so that
// every instantiation rounds alike)
__device__ __forceinline__ void wino_bt(float d0, float d1, float d2, float d3, float d4, float& v0, float& v1, float& v2, float& v3, float& v4) {
    if (XSQ_WINO_ABL & 32) { v0 = d0; v1 = d1; v2 = d2; v3 = d3; v4 = d4; return; }
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
    constexpr int PLANE_O = WN_EROWS * WN_SLD;                   // word offset of the odd plane
    __shared__ __attribute__((aligned(16))) float slab[(WN_EROWS + WN_OROWS) * WN_SLD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 5 * WN_BTILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kq = lane >> 4;
    const WinoTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.Fo), "s"(t.Fi), "s"(t.in_off), "s"(t.out_off), "s"(t.shift_off), "s"(t.u_off),
                 "s"(t.b), "s"(t.P));
#if XSQ_WINO_STAMPS
    unsigned long long wn_t0 = wall_clock64();
#endif
    const int kf = t.kf, Fo = t.Fo, Fi = t.Fi, P = t.P, b = t.b;
    const int To = TRANSPOSED ? a.T1 : a.T2, Ti = TRANSPOSED ? a.T2 : a.T1;
    const float* in = (TRANSPOSED ? a.act2 : a.act1) + t.in_off;

    // ---- slab staging: lane = (position lane p0 = tid / 13 of 19, channel quad c4 = tid % 13), load r -> slab position p0 + 19 r.
    // Segment i covers slab positions [A_i, A_i + 2 np_i + 3): local position j' is input position 2 qs_i - PAD + j' of input
    // row f0 + i -+ df and lands in plane j' & 1, row (j' >> 1) + (rows of the segments before).  What does not depend on the
    // frequency tap -- LDS address, byte offset at df = 0 (or BUF_OOB), segment bit -- is formed once per tile.
    constexpr int SPL = 256 / (CS / 4);                          // position lanes (19)
    constexpr int NLD = (WN_POS + SPL - 1) / SPL;                // loads per lane and slab (8)
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(in, 0x40000000u);    // (a (block, target)'s input is < 2^30 bytes: cdae_launch_layer)
    unsigned s_vo[NLD], s_seg = 0;
    unsigned s_lds[NLD / 2];          // two 16-bit float4 indices per word (0xffff: the position does not exist)
    int s_f0 = 0;
    const int s_p0 = tid / (CS / 4), s_c4 = tid - s_p0 * (CS / 4);
    auto setup_staging = [&](int f0, int q0, int npairs, int n0) {
        const bool s_on = tid < SPL * (CS / 4);
        const int A1 = 2 * n0 + 3;
        s_seg = 0; s_f0 = f0;
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int j = s_p0 + SPL * r;
            const int seg = j >= A1 ? 1 : 0;
            const int jj = j - (seg ? A1 : 0);
            const int np = seg ? npairs - n0 : n0;
            const int pos = 2 * (seg ? 0 : q0) - PAD + jj;           // input position
            const bool exists = s_on && jj < 2 * np + 3 && np > 0;
            const bool inr = exists && (unsigned)pos < (unsigned)Ti;
            s_vo[r] = inr ? 4u * (unsigned)(((b * Fi + f0 + seg) * Ti + pos) * CS + 4 * s_c4) : BUF_OOB;
            s_seg |= seg ? 1u << r : 0u;
            const int row = (jj >> 1) + (seg ? ((jj & 1) ? n0 + 1 : n0 + 2) : 0);
            const unsigned l4 = exists ? (unsigned)(((jj & 1) ? PLANE_O : 0) + row * WN_SLD + 4 * s_c4) >> 2 : 0xffffu;
            if (r & 1) s_lds[r >> 1] |= l4 << 16; else s_lds[r >> 1] = l4;
        }
    };
    float4 sv[NLD];
    auto load_slab = [&](int df) {
        const int fa = TRANSPOSED ? s_f0 - df : s_f0 + df;                               // input row of segment 0; segment 1: + 1
        const bool ok0 = (unsigned)fa < (unsigned)Fi, ok1 = (unsigned)(fa + 1) < (unsigned)Fi;       // (uniform)
        const unsigned delta = 4u * (unsigned)((TRANSPOSED ? -df : df) * Ti * CS);      // (a switched-off offset stays past the range)
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const bool on = ((s_seg >> r) & 1u) ? ok1 : ok0;
            sv[r] = (XSQ_WINO_ABL & 8) ? make_float4(1.f, 2.f, 3.f, 4.f) : buf_ld4(rin, on ? s_vo[r] + delta : BUF_OOB, 0);
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const unsigned l4 = (r & 1) ? s_lds[r >> 1] >> 16 : s_lds[r >> 1] & 0xffffu;
            if (l4 != 0xffffu) *reinterpret_cast<float4*>(&slab[4 * l4]) = sv[r];
        }
    };

    // ---- weight stream: chunk (df, s) = five component tiles of 51 columns -> the LDS buffer the chunk counter picks.  Chunks
    // 0 / 1 are 1020 float4 (thread tid takes float4 tid + 256 r, r < 4), chunk 2 is 1275 (r < 5); the spare threads load past
    // the descriptor (zeros) and write into pad words nobody reads: no predicate around any load or store.
    const __amdgpu_buffer_rsrc_t ru = buf_rsrc(a.upool + t.u_off, 4u * (unsigned)(kf * WN_UDF));
    // float4 x = tid + 256 r of a chunk's main part -> LDS row x >> 2, k quad x & 3 (the four spare slots of r = 3 -> pad words);
    // float4 tid of the tail part (chunk 2 only) -> row tid, words 16..19
    const int b_lds0 = (tid >> 2) * WN_BLD + 4 * (tid & 3);
    float4 gb[5];
    auto load_chunk = [&](int df, int s) {
        if (XSQ_WINO_ABL & 4) return;
        const int so = 4 * (df * WN_UDF + s * WN_U16);
#pragma unroll
        for (int r = 0; r < 4; ++r) gb[r] = buf_ld4(ru, (r < 3 || tid < WN_U16 / 4 - 768) ? 16u * (unsigned)(tid + 256 * r) : BUF_OOB, so);
        if (s == 2) gb[4] = buf_ld4(ru, tid < WN_UT / 4 ? 16u * (unsigned)tid : BUF_OOB, 4 * (df * WN_UDF + 3 * WN_U16));
    };
    auto store_chunk = [&](int s, int buf) {
        if (XSQ_WINO_ABL & 4) return;
        float* Bw = Bs + buf * 5 * WN_BTILE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(&Bw[(r < 3 || tid < WN_U16 / 4 - 768) ? b_lds0 + 64 * r * WN_BLD : (tid - (WN_U16 / 4 - 768)) * WN_BLD + 20]) = gb[r];
        if (s == 2) *reinterpret_cast<float4*>(&Bw[tid < WN_UT / 4 ? tid * WN_BLD + 16 : 20]) = gb[4];
    };

    const int bf = q * WN_BLD + 4 * kq;                          // weight tile: column q of a 16-column block, k-quad kq
    // the vector columns' weights: ONE word per lane -- lane c of row kq holds channel 4 kq + (c & 3) of the chunk; the four
    // v_fmac_f32_dpp of a column take it with row_newbcast:i (lane i of every row of 16 to the whole row): a 4-byte LDS read
    // (2 cycles) and one register per column where the 16-byte broadcast read took 4 cycles and four registers
    const int bv = 48 * WN_BLD + 4 * kq + (q & 3);
    const int pl = wave * 16 + q;                                // this lane's pair of the tile

    f32x4 acc[5][3];
    float accv[5][NV];
    auto clear_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) accv[j][cc] = 0.f;
        }
    };

    // A chunk's operands are requested one step early: the pair's raw positions of chunk s + 1 while chunk s computes (the
    // planes only change between frequency taps), a component's weight fragments while the component before it computes.
    struct Frag { float4 w[3]; float u[NV]; float wt[3]; float ut[NV]; };
    auto read_frag = [&](Frag& f, const float* Bt, bool tail) {
        f.w[0] = *reinterpret_cast<const float4*>(&Bt[bf]);
        f.w[1] = *reinterpret_cast<const float4*>(&Bt[bf + 16 * WN_BLD]);
        f.w[2] = *reinterpret_cast<const float4*>(&Bt[bf + 32 * WN_BLD]);
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) f.u[cc] = Bt[bv + cc * WN_BLD];
        if (tail) {                                                // channel 48 + kq: word 16 + kq of the row
            const int tb = q * WN_BLD + 16 + kq;
            f.wt[0] = Bt[tb]; f.wt[1] = Bt[tb + 16 * WN_BLD]; f.wt[2] = Bt[tb + 32 * WN_BLD];
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) f.ut[cc] = Bt[(48 + cc) * WN_BLD + 16 + kq];
        }
    };

    // ---- prologue: the slab of tap 0 and chunk 0 in LDS
    const int f0 = t.Q0 / P, q0 = t.Q0 - f0 * P;
    const int npairs = min(WN_PAIRS, Fo * P - t.Q0);             // pairs of the tile that exist
    const int n0 = min(npairs, P - q0);                          // ... in segment 0 (row f0); the rest in segment 1 (row f0 + 1)
    int cnt = 0;                                                 // chunks issued so far: chunk c lives in buffer c & 1
    load_chunk(0, 0);
    setup_staging(f0, q0, npairs, n0);
    load_slab(0);
    store_slab();
    store_chunk(0, 0);
    __syncthreads();
    WN_STAMP(0);

    {
        const int myseg = pl >= n0 ? 1 : 0;
        const int eb = (pl + 2 * myseg) * WN_SLD + 4 * kq;       // even plane: rows er, er + 1, er + 2 = positions 0, 2, 4 of the pair
        const int ob = PLANE_O + (pl + myseg) * WN_SLD + 4 * kq; // odd plane: rows or, or + 1 = positions 1, 3
        float4 dn[5];
        auto read_raw = [&](int s) {
            dn[0] = *reinterpret_cast<const float4*>(&slab[eb + 16 * s]);
            dn[1] = *reinterpret_cast<const float4*>(&slab[ob + 16 * s]);
            dn[2] = *reinterpret_cast<const float4*>(&slab[eb + WN_SLD + 16 * s]);
            dn[3] = *reinterpret_cast<const float4*>(&slab[ob + WN_SLD + 16 * s]);
            dn[4] = *reinterpret_cast<const float4*>(&slab[eb + 2 * WN_SLD + 16 * s]);
        };
        clear_acc();
        for (int df = 0; df < kf; ++df) {
            const bool more = df + 1 < kf;
            if (XSQ_WINO_RAW_AHEAD) read_raw(0);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                // the next chunk's weights are requested here and written into the other buffer at the end of this chunk; the next
                // tap's slab is requested in the tap's last chunk and written behind its barrier
                const int cur = cnt & 1;
                if (s < 2) load_chunk(df, s + 1);
                else if (more) { load_chunk(df + 1, 0); load_slab(df + 1); }
                const float* Bc = Bs + cur * 5 * WN_BTILE;
                if (!XSQ_WINO_RAW_AHEAD) read_raw(s);
                Frag fr[2];
                read_frag(fr[0], Bc, s == 2);
                // this lane's pair: four channels of the chunk, and in the last chunk also channel 48 + kq
                float v[5][4], vt[5];
                {
                    const float d[5][4] = {{dn[0].x, dn[0].y, dn[0].z, dn[0].w}, {dn[1].x, dn[1].y, dn[1].z, dn[1].w}, {dn[2].x, dn[2].y, dn[2].z, dn[2].w},
                                           {dn[3].x, dn[3].y, dn[3].z, dn[3].w}, {dn[4].x, dn[4].y, dn[4].z, dn[4].w}};
#pragma unroll
                    for (int i = 0; i < 4; ++i) wino_bt(d[0][i], d[1][i], d[2][i], d[3][i], d[4][i], v[0][i], v[1][i], v[2][i], v[3][i], v[4][i]);
                }
                if (s == 2) {
                    const int te = eb - 4 * kq + 48 + kq, to = ob - 4 * kq + 48 + kq;
                    wino_bt(slab[te], slab[to], slab[te + WN_SLD], slab[to + WN_SLD], slab[te + 2 * WN_SLD], vt[0], vt[1], vt[2], vt[3], vt[4]);
                } else if (XSQ_WINO_RAW_AHEAD) {
                    read_raw(s + 1);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const Frag& f = fr[j & 1];
                    if (j < 4) read_frag(fr[(j + 1) & 1], Bc + (j + 1) * WN_BTILE, s == 2);
                    if (XSQ_WINO_SCHED2 >= 0) __builtin_amdgcn_sched_barrier(XSQ_WINO_SCHED2);
                    const float wa[4] = {f.w[0].x, f.w[0].y, f.w[0].z, f.w[0].w}, wb[4] = {f.w[1].x, f.w[1].y, f.w[1].z, f.w[1].w};
                    const float wc[4] = {f.w[2].x, f.w[2].y, f.w[2].z, f.w[2].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (XSQ_WINO_SWAP) {
                            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], v[j][i], acc[j][0], 0, 0, 0);
                            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i], v[j][i], acc[j][1], 0, 0, 0);
                            acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[i], v[j][i], acc[j][2], 0, 0, 0);
                            continue;
                        }
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wa[i], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wb[i], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][i], wc[i], acc[j][2], 0, 0, 0);
                    }
                    if (s == 2) {
                        if (XSQ_WINO_SWAP) {
                            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[0], vt[j], acc[j][0], 0, 0, 0);
                            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[1], vt[j], acc[j][1], 0, 0, 0);
                            acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[2], vt[j], acc[j][2], 0, 0, 0);
                        } else {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[j], f.wt[0], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[j], f.wt[1], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[j], f.wt[2], acc[j][2], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < NV; ++cc) {
                        if (XSQ_WINO_ABL & 2) { accv[j][cc] += v[j][0] + f.u[cc]; continue; }
                        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                            : "+v"(accv[j][cc])
                            : "v"(f.u[cc]), "v"(v[j][0]), "v"(v[j][1]), "v"(v[j][2]), "v"(v[j][3]));
                        if (s == 2) asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[j][cc]) : "v"(vt[j]), "v"(f.ut[cc]));
                    }
                    if (XSQ_WINO_SCHED >= 0) __builtin_amdgcn_sched_barrier(XSQ_WINO_SCHED);
                }
                // other buffer: last read in the chunk before, every wave is past that chunk's barrier
                if (s < 2) store_chunk(s + 1, cur ^ 1);
                else if (more) store_chunk(0, cur ^ 1);
                cnt += 1;
                WN_SUB(4, __syncthreads());
            }
            if (more) {                  // every wave is past the barrier behind the slab's last reader
                store_slab();
                WN_SUB(4, __syncthreads());
            }
        }
        WN_STAMP(1);

        // ---- epilogue: output transform, shift + ReLU, through a per-wave LDS image (the planes are free: the
        // loop ended on a barrier), out as 16-byte stores.  Image row 2 p + r = output r of the wave's pair p.
        const float* shift = a.pool + t.shift_off;
        if (XSQ_WINO_SWAP) {
            // weights as the row operand: accumulator register r of this lane is output channel 16 cb + 4 kq + r of the lane's OWN
            // pair -- output transform, shift + ReLU, one 16-byte store per column block and output, no LDS image, no index exchange
            float* out = (TRANSPOSED ? a.act3 : a.act2) + t.out_off;
            const __amdgpu_buffer_rsrc_t ro = buf_rsrc(out, 0x40000000u);
            const int sg = pl >= n0 ? 1 : 0;
            const int qq = sg ? pl - n0 : q0 + pl;
            const bool ok0 = pl < npairs, ok1 = ok0 && 2 * qq + 1 < To;
            const unsigned vo = 4u * (unsigned)((((b * Fo + f0 + sg) * To) + 2 * qq) * CS + 4 * kq);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const float4 sh = *reinterpret_cast<const float4*>(shift + 16 * cb + 4 * kq);
                const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
                float y0[4], y1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][cb][r], m1 = acc[1][cb][r], m2 = acc[2][cb][r], m3 = acc[3][cb][r], m4 = acc[4][cb][r];
                    y0[r] = fmaxf(((m0 + m1) + (m2 + m3)) + shv[r], 0.f);
                    y1[r] = fmaxf((fmaf(2.f, m3, m1 - m2) + m4) + shv[r], 0.f);
                }
                const bool live = !((XSQ_WINO_ABL & 16) && y0[0] != 1.2345e-30f);
                // (displacements in the LANE offset, scalar offset 0: the store-data hazard of 16-byte stores with an SGPR offset, common.h)
                buf_st4(make_float4(y0[0], y0[1], y0[2], y0[3]), ro, (ok0 && live) ? vo + 64u * cb : BUF_OOB, 0);
                buf_st4(make_float4(y1[0], y1[1], y1[2], y1[3]), ro, (ok1 && live) ? vo + 64u * cb + 4u * CS : BUF_OOB, 0);
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
            const float4 sh = *reinterpret_cast<const float4*>(shift + 48);
            const unsigned vt48 = vo - 16u * (unsigned)kq;                   // (channel 0 of the row)
            buf_st4(make_float4(fmaxf(y0v[0] + sh.x, 0.f), fmaxf(y0v[1] + sh.y, 0.f), fmaxf(y0v[2] + sh.z, 0.f), fmaxf(y0v[3] + sh.w, 0.f)), ro,
                    (ok0 && kq == 0) ? vt48 + 192u : BUF_OOB, 0);
            buf_st4(make_float4(fmaxf(y1v[0] + sh.x, 0.f), fmaxf(y1v[1] + sh.y, 0.f), fmaxf(y1v[2] + sh.z, 0.f), fmaxf(y1v[3] + sh.w, 0.f)), ro,
                    (ok1 && kq == 0) ? vt48 + 192u + 4u * CS : BUF_OOB, 0);
            WN_STAMP(2);
#if XSQ_WINO_STAMPS
            if (tid == 0) atomicAdd(&g_wn_stamps[3], 1ull);
#endif
            return;
        }
        float* img = slab + wave * 32 * CS;
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
        // image row -> output row: pair pl = 16 wave + (row >> 1) of the tile, segment by n0, t = 2 q + (row & 1); rows of pairs
        // that do not exist and the phantom second row of an odd To are switched out of the descriptor's range
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
            buf_st4(val, ro, (ok && !((XSQ_WINO_ABL & 16) && val.x != 1.2345e-30f)) ? vo : BUF_OOB, 0);
        }
    }
}

}  // namespace xsq
