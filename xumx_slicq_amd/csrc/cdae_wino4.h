// Layers 2 and 3 of the CDAE (fp32) as Winograd F(4, 4) along the four TIME taps -- the A/B arm of cdae_wino.h's F(2, 4)
// (bit 8 of xsq_model_set_winograd; /root/reference/xumx_slicq_v2/model.py:140-170 are the layers).
//
// F(4, 4) computes FOUR neighbouring outputs from seven inputs with SEVEN products per (input channel, output channel) where
// F(2, 4) takes ten and the direct form sixteen: Cook-Toom on the points {0, 1, -1, 2, -2, 1/2, inf},
//     V_j = sum_i BT[j][i] d_i,   M_j = V_j . U_j  (U = G w, made on the host in fp64),   y_k = sum_j AT[k][j] M_j
// with integer BT (the fractions live in G; row 5 carries 1/32 of its textbook scale, G's row 32 x).  In an fp32 simulation of
// these layers (tools/probe/wino_f34_error.py: 156 channels x taps, ReLU inputs) the form sits 5.4e-7 RMS from fp64 where
// F(2, 4) has 3.3e-7 and the direct fp32 sum 2.1e-7 -- the accumulation over the channels dominates all three.
//
// What kept F(4, 4) out of round 5 was its footprint next to F(2, 4)'s tile (84 accumulator registers, a 59 KB slab, 2 x 34 KB
// of weight tiles: no two workgroups per CU).  This kernel changes the split instead: ONE 512-thread workgroup per CU (the
// same two waves per SIMD), tile = 64 consecutive output QUADS of one batch item in the flattened (f, quad) space, and the
// eight waves are 4 quad groups x 2 K HALVES: wave (g, h) contracts channels 4 kq + 2 h + {0, 1} of every 16-channel chunk for
// the quads 16 g .. 16 g + 15 -- half the raw reads, half the input transforms, half the MFMAs of the chunk per wave, perfectly
// balanced (the four tail channels 48..51 of a frequency tap go to h = 1).  The two halves of a quad group meet once per tile:
// each applies the (linear) output transform to its partial sums, hands the two outputs it does not own to its partner
// through LDS (the weight buffers and the planes are free by then), adds, shifts, clamps and stores 16 bytes per column block and output.
//
// LDS: the slab as FOUR planes (position mod 4) of 66 rows x 52 words (a lane's seven positions are rows r, r + 1 of planes
// 0..2 and row r of plane 3: consecutive lanes read consecutive rows) = 54.9 KB, two weight buffers of 7 components x 51
// columns x 20 words = 57.1 KB: 112 KB of the CU's 160.  A wave's K half of a 16-channel chunk is channels 8 h + 2 kq + {0, 1}
// (MFMA step i takes channel 8 h + 2 kq + i from k-quad kq, both operands alike): 8-byte reads at row * stride + 8 h + 2 kq
// words, and with row strides of 4 x odd words (20, 52) the 32 lanes of a ds_read_b64 group tile the 64 banks exactly.
#pragma once
#include "cdae_wino.h"
#include <type_traits>

#ifndef XSQ_WINO4_SCHED
#define XSQ_WINO4_SCHED 0    // >= 0: a scheduling barrier with this mask behind every component's MFMAs; -1: none
#endif
#ifndef XSQ_WINO4_SCHED2
#define XSQ_WINO4_SCHED2 0   // >= 0: ... and one behind the NEXT component's fragment reads (issued in front of this component's MFMAs); -1: none
#endif
#ifndef XSQ_WINO4_ABL
#define XSQ_WINO4_ABL 0      // diagnostic builds (wrong results, timings only): 2 no vector columns, 4 no weight stream, 8 no slab loads, 32 no input transform, 64 no barrier per chunk, 128 no raw reads
#endif

namespace xsq {

#ifndef XSQ_WINO4_STAMPS
#define XSQ_WINO4_STAMPS 0   // diagnostic build: thread 0 of every workgroup adds its phase times (s_memrealtime ticks of 10 ns) to g_w4_stamps:
                             // 0 prologue, 1 chunk loop, 2 epilogue, 3 tiles, 4 of the loop: thread 0 waiting at its barriers
#endif
#if XSQ_WINO4_STAMPS
__device__ unsigned long long g_w4_stamps[8];
#define W4_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_w4_stamps[i], now_ - w4_t0); w4_t0 = now_; } } while (0)
#define W4_SUB(i, expr) do { const unsigned long long a_ = wall_clock64(); expr; if (tid == 0) atomicAdd(&g_w4_stamps[i], wall_clock64() - a_); } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#define W4_SUB(i, expr) do { expr; } while (0)
#endif

constexpr int W4_QUADS = 64;                              // output quads per tile (4 quad groups x 16)
constexpr int W4_MAXSEG = 2;                              // (b, f) rows a tile may touch (needs P >= W4_QUADS)
constexpr int W4_PROWS = W4_QUADS + W4_MAXSEG;            // rows per plane: quads + 1 per segment
constexpr int W4_POS = 4 * W4_QUADS + 3 * W4_MAXSEG;      // slab positions of a tile (262)
constexpr int W4_NC = 7;                                  // components
constexpr int W4_SLD = 52;                                // plane row stride in words: the 52 channels, no pad (13 slots; header comment, LDS)
constexpr int W4_BLD = 20;                                // weight tile row: 16 k | 4 tail k (5 slots)
constexpr int W4_BTILE = WN_COLS * W4_BLD;                // words per component tile in LDS (row = component * 51 + column)
constexpr int W4_U16 = W4_NC * WN_COLS * 16;              // words of a chunk's main part in global memory: [component][col][16 k] = 1428 float4
constexpr int W4_UT = W4_NC * WN_COLS * 4;                // the tail channels 48..51: [component][col][4 k] = 357 float4 (staged with chunk 2)
constexpr int W4_UDF = 3 * W4_U16 + W4_UT;                // words per frequency tap

// Cook-Toom F(4, 4), points {0, 1, -1, 2, -2, 1/2, inf}: G (7 x 4), applied to the folded weights on the host (cdae.hip)
static const double W4_G[7][4] = {{1.0 / 4, 0, 0, 0},
                                  {1.0 / 6, 1.0 / 6, 1.0 / 6, 1.0 / 6},
                                  {1.0 / 18, -1.0 / 18, 1.0 / 18, -1.0 / 18},
                                  {1.0 / 72, 1.0 / 36, 1.0 / 18, 1.0 / 9},
                                  {1.0 / 120, -1.0 / 60, 1.0 / 30, -1.0 / 15},
                                  {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45},
                                  {0, 0, 0, 1.0 / 2}};

// word offset of (component j, column col, input channel ci) inside a frequency tap's block of the transformed weights
__host__ __device__ constexpr int wino4_u_off(int j, int col, int ci) {
    return ci < 48 ? (ci / 16) * W4_U16 + (j * WN_COLS + col) * 16 + ci % 16 : 3 * W4_U16 + (j * WN_COLS + col) * 4 + (ci - 48);
}

// input transform of one channel: d[0..6] -> v[0..6] (BT of the header comment; FMAs spelled out so that every instantiation
// rounds alike)
__device__ __forceinline__ void wino4_bt(const float (&d)[7], float (&v)[7]) {
    if (XSQ_WINO4_ABL & 32) {
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = d[j];
        return;
    }
    v[0] = fmaf(4.f, d[0], fmaf(-8.f, d[1], fmaf(-5.f, d[2], fmaf(10.f, d[3], fmaf(-2.f, d[5], d[4])))));
    v[1] = fmaf(-4.f, d[1], fmaf(4.f, d[2], fmaf(9.f, d[3], fmaf(-2.f, d[5], -d[4]))));
    v[2] = fmaf(-4.f, d[1], fmaf(12.f, d[2], fmaf(-7.f, d[3], fmaf(-3.f, d[4], 2.f * d[5]))));
    v[3] = fmaf(2.f, d[1], fmaf(-3.f, d[2], fmaf(-4.f, d[3], fmaf(3.f, d[4], 2.f * d[5]))));
    v[4] = fmaf(2.f, d[1], fmaf(-5.f, d[2], fmaf(5.f, d[4], -2.f * d[5])));
    v[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
    v[6] = fmaf(-4.f, d[1], fmaf(8.f, d[2], fmaf(5.f, d[3], fmaf(-10.f, d[4], fmaf(2.f, d[6], -d[5])))));
}

// output transform: m[0..6] -> y[0..3] (AT: rows (1 1 1 1 1 1 0), (0 1 -1 2 -2 1/2 0), (0 1 1 4 4 1/4 0), (0 1 -1 8 -8 1/8 1))
__device__ __forceinline__ void wino4_at(const float (&m)[7], float (&y)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = (m[0] + s12) + (s34 + m[5]);
    y[1] = fmaf(0.5f, m[5], fmaf(2.f, d34, d12));
    y[2] = fmaf(0.25f, m[5], fmaf(4.f, s34, s12));
    y[3] = fmaf(0.125f, m[5], fmaf(8.f, d34, d12)) + m[6];
}

template <bool TRANSPOSED>
__global__ __launch_bounds__(512, 2) void cdae_wino4_kernel(CdaeArgs a, const WinoTileDev* __restrict__ tiles, int ntiles) {
#pragma clang fp contract(off)
    constexpr int PAD = TRANSPOSED ? 3 : 0;
    constexpr int NV = TRANSPOSED ? H1 - 48 : H2 - 48;           // real channels past 47: 2 (layer 3 -> 50) or 3 (layer 2 -> 51)
    constexpr int PLANE = W4_PROWS * W4_SLD;                     // words per plane
    __shared__ __attribute__((aligned(16))) float slab[4 * PLANE];
    __shared__ __attribute__((aligned(16))) float Bs[2 * W4_NC * W4_BTILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, kq = lane >> 4;
    const int g = wave & 3, h = wave >> 2;                       // quad group, K half
    const int To = TRANSPOSED ? a.T1 : a.T2, Ti = TRANSPOSED ? a.T2 : a.T1;
#if XSQ_WINO4_STAMPS
    unsigned long long w4_t0 = wall_clock64();
#endif

    // (A run of consecutive tiles per workgroup with the next tile's first slab and weight chunk requested in front of the
    //  epilogue -- nothing else on the CU hides a tile's first loads -- was built and measured: 96-144 bytes of scratch in the
    //  chunk loop and 1.13-1.19 / 1.01-1.08 ms against 0.88 / 0.87 for this form, slower with every tile added to the run:
    //  profiles/r11_ab_runs.txt, r11w4.)
    struct Ctx {                     // a tile as the loops see it (wave-uniform)
        int kf, Fo, Fi, P, b, f0, q0, nquads, n0;
        int64_t out_off, shift_off;
        __amdgpu_buffer_rsrc_t rin, ru;
    };
    auto make_ctx = [&](const WinoTileDev& t) {
        asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.Fo), "s"(t.Fi), "s"(t.in_off), "s"(t.out_off), "s"(t.shift_off), "s"(t.u_off),
                     "s"(t.b), "s"(t.P));
        Ctx c;
        c.kf = t.kf; c.Fo = t.Fo; c.Fi = t.Fi; c.P = t.P; c.b = t.b;
        c.f0 = t.Q0 / t.P; c.q0 = t.Q0 - c.f0 * t.P;
        c.nquads = min(W4_QUADS, t.Fo * t.P - t.Q0);             // quads of the tile that exist
        c.n0 = min(c.nquads, t.P - c.q0);                        // ... in segment 0 (row f0); the rest in segment 1 (row f0 + 1)
        c.out_off = t.out_off; c.shift_off = t.shift_off;
        c.rin = buf_rsrc((TRANSPOSED ? a.act2 : a.act1) + t.in_off, 0x40000000u);      // (a (block, target)'s input is < 2^30 bytes: cdae_launch_layer)
        c.ru = buf_rsrc(a.upool + t.u_off, 4u * (unsigned)(t.kf * W4_UDF));
        return c;
    };

    // ---- slab staging: lane = (position lane p0 = tid / 13 of 39, channel quad c4 = tid % 13), load r -> slab position p0 + 39 r.
    // Segment i covers 4 np_i + 3 slab positions: local position j' is input position 4 qs_i - PAD + j' of input row
    // f0 + i -+ df and lands in plane j' & 3, row (j' >> 2) + (0 | n0 + 1).
    constexpr int SPL = 512 / (CS / 4);                          // position lanes (39)
    constexpr int NLD = (W4_POS + SPL - 1) / SPL;                // loads per lane and slab (7)
    unsigned s_vo[NLD], s_seg = 0;
    unsigned s_lds[(NLD + 1) / 2];    // two 16-bit float4 indices per word (0xffff: the position does not exist)
    int s_f0 = 0;
    const int s_p0 = tid / (CS / 4), s_c4 = tid - s_p0 * (CS / 4);
    auto setup_staging = [&](const Ctx& c) {
        const bool s_on = tid < SPL * (CS / 4);
        const int A1 = 4 * c.n0 + 3;
        s_seg = 0; s_f0 = c.f0;
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int j = s_p0 + SPL * r;
            const int seg = j >= A1 ? 1 : 0;
            const int jj = j - (seg ? A1 : 0);
            const int np = seg ? c.nquads - c.n0 : c.n0;
            const int pos = 4 * (seg ? 0 : c.q0) - PAD + jj;         // input position
            const bool exists = s_on && jj < 4 * np + 3 && np > 0;
            const bool inr = exists && (unsigned)pos < (unsigned)Ti;
            s_vo[r] = inr ? 4u * (unsigned)(((c.b * c.Fi + c.f0 + seg) * Ti + pos) * CS + 4 * s_c4) : BUF_OOB;
            s_seg |= seg ? 1u << r : 0u;
            const int row = (jj >> 2) + (seg ? c.n0 + 1 : 0);
            const unsigned l4 = exists ? (unsigned)((jj & 3) * PLANE + row * W4_SLD + 4 * s_c4) >> 2 : 0xffffu;
            if (r & 1) s_lds[r >> 1] |= l4 << 16; else s_lds[r >> 1] = l4;
        }
    };
    float4 sv[NLD];
    auto load_slab = [&](const Ctx& c, int df) {
        const int fa = TRANSPOSED ? s_f0 - df : s_f0 + df;                               // input row of segment 0; segment 1: + 1
        const bool ok0 = (unsigned)fa < (unsigned)c.Fi, ok1 = (unsigned)(fa + 1) < (unsigned)c.Fi;   // (uniform)
        const unsigned delta = 4u * (unsigned)((TRANSPOSED ? -df : df) * Ti * CS);      // (a switched-off offset stays past the range)
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const bool on = ((s_seg >> r) & 1u) ? ok1 : ok0;
            sv[r] = (XSQ_WINO4_ABL & 8) ? make_float4(1.f, 2.f, 3.f, 4.f) : buf_ld4(c.rin, on ? s_vo[r] + delta : BUF_OOB, 0);
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const unsigned l4 = (r & 1) ? s_lds[r >> 1] >> 16 : s_lds[r >> 1] & 0xffffu;
            if (l4 != 0xffffu) *reinterpret_cast<float4*>(&slab[4 * l4]) = sv[r];
        }
    };

    // ---- weight stream: chunk (df, s) = seven component tiles of 51 columns -> the LDS buffer the chunk counter picks.  A main
    // part is 1428 float4 (thread tid takes float4 tid + 512 r, r < 3), chunk 2 adds the 357 float4 of the tail channels.
    const int b_lds0 = (tid >> 2) * W4_BLD + 4 * (tid & 3);
    constexpr int SPARE0 = W4_U16 / 4 - 1024;                    // threads of r = 2 that carry data (404)
    float4 gb[4];
    auto load_chunk = [&](const Ctx& c, int df, int s) {
        if (XSQ_WINO4_ABL & 4) return;
        const int so = 4 * (df * W4_UDF + s * W4_U16);
#pragma unroll
        for (int r = 0; r < 3; ++r) gb[r] = buf_ld4(c.ru, (r < 2 || tid < SPARE0) ? 16u * (unsigned)(tid + 512 * r) : BUF_OOB, so);
        if (s == 2) gb[3] = buf_ld4(c.ru, tid < W4_UT / 4 ? 16u * (unsigned)tid : BUF_OOB, 4 * (df * W4_UDF + 3 * W4_U16));
    };
    auto store_chunk = [&](int s, int buf) {
        if (XSQ_WINO4_ABL & 4) return;
        float* Bw = Bs + buf * W4_NC * W4_BTILE;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < 2 || tid < SPARE0) *reinterpret_cast<float4*>(&Bw[b_lds0 + 128 * r * W4_BLD]) = gb[r];
        if (s == 2 && tid < W4_UT / 4) *reinterpret_cast<float4*>(&Bw[tid * W4_BLD + 16]) = gb[3];
    };

    const int bf = q * W4_BLD + 8 * h + 2 * kq;                  // weight tile: row q of a 16-row block, this wave's two channels of k-quad kq
    // the vector columns' weights: ONE word per lane -- lane c of row kq holds channel 8 h + 2 kq + (c & 1) of the chunk; the two
    // v_fmac_f32_dpp of a column take it with row_newbcast:0 / 1 (cdae_wino.h)
    const int bv = 48 * W4_BLD + 8 * h + 2 * kq + (q & 1);
    const int pl = g * 16 + q;                                   // this lane's quad of the tile

    // (the three 8-byte reads of a component from three OPAQUE bases: from one base the compiler pairs them into ds_read2_b64,
    //  which is banked mod 32 over groups of 16 lanes at half the rate -- MI355X_MICROARCH.md, LDS -- where a plain ds_read_b64 of
    //  these rows, 20 q + 8 h + 2 kq words, is conflict-free)
    int bfc0 = bf, bfc1 = bf + 16 * W4_BLD, bfc2 = bf + 32 * W4_BLD;
    asm volatile("" : "+v"(bfc0), "+v"(bfc1), "+v"(bfc2));
    struct Frag { float2 w[3]; float u[NV]; float wt[3]; float ut[NV]; };
    auto read_frag = [&](Frag& f, const float* Bt, bool tail) {
        f.w[0] = *reinterpret_cast<const float2*>(&Bt[bfc0]);
        f.w[1] = *reinterpret_cast<const float2*>(&Bt[bfc1]);
        f.w[2] = *reinterpret_cast<const float2*>(&Bt[bfc2]);
#pragma unroll
        for (int cc = 0; cc < NV; ++cc) f.u[cc] = Bt[bv + cc * W4_BLD];
        if (tail) {                                                // channel 48 + kq: word 16 + kq of the row
            const int tb = q * W4_BLD + 16 + kq;
            f.wt[0] = Bt[tb]; f.wt[1] = Bt[tb + 16 * W4_BLD]; f.wt[2] = Bt[tb + 32 * W4_BLD];
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) f.ut[cc] = Bt[(48 + cc) * W4_BLD + 16 + kq];
        }
    };

    // ---- prologue: the slab of tap 0 and chunk 0 in LDS
    const Ctx c = make_ctx(tiles[xcd_remap(blockIdx.x, ntiles)]);
    int cnt = 0;                                                 // chunks issued so far: chunk c lives in buffer c & 1
    load_chunk(c, 0, 0);
    setup_staging(c);
    load_slab(c, 0);
    store_slab();
    store_chunk(0, 0);
    __syncthreads();
    W4_STAMP(0);

    {
        const int kf = c.kf;
        f32x4 acc[W4_NC][3];
        float accv[W4_NC][NV];
#pragma unroll
        for (int j = 0; j < W4_NC; ++j) {
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[j][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) accv[j][cc] = 0.f;
        }
        const int myseg = pl >= c.n0 ? 1 : 0;
        const int rb = (pl + myseg) * W4_SLD + 8 * h + 2 * kq;   // row r of every plane: positions 0..3 of the quad; row r + 1 of planes 0..2: 4..6
        for (int df = 0; df < kf; ++df) {
            const bool more = df + 1 < kf;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                // the next chunk's weights are requested here and written into the other buffer at the end of this chunk; the next
                // tap's slab is requested in the tap's last chunk and written behind its barrier
                const int cur = cnt & 1;
                if (s < 2) load_chunk(c, df, s + 1);
                else if (more) { load_chunk(c, df + 1, 0); load_slab(c, df + 1); }
                const float* Bc = Bs + cur * W4_NC * W4_BTILE;
                const bool tail = s == 2 && h;                        // (wave-uniform)
                float2 dn[7];
#pragma unroll
                for (int p = 0; p < 7; ++p) dn[p] = (XSQ_WINO4_ABL & 128) ? make_float2(1.f + p, 2.f) : *reinterpret_cast<const float2*>(&slab[(p & 3) * PLANE + rb + (p >> 2) * W4_SLD + 16 * s]);
                Frag fr[2];
                read_frag(fr[0], Bc, tail);
                float v[2][7], vt[7];
                {
                    const float d0[7] = {dn[0].x, dn[1].x, dn[2].x, dn[3].x, dn[4].x, dn[5].x, dn[6].x};
                    const float d1[7] = {dn[0].y, dn[1].y, dn[2].y, dn[3].y, dn[4].y, dn[5].y, dn[6].y};
                    wino4_bt(d0, v[0]);
                    wino4_bt(d1, v[1]);
                }
                if (tail) {
                    const int tb = rb - 8 * h - 2 * kq + 48 + kq;
                    float dt_[7];
#pragma unroll
                    for (int p = 0; p < 7; ++p) dt_[p] = slab[(p & 3) * PLANE + tb + (p >> 2) * W4_SLD];
                    wino4_bt(dt_, vt);
                }
#pragma unroll
                for (int j = 0; j < W4_NC; ++j) {
                    const Frag& f = fr[j & 1];
                    if (j + 1 < W4_NC) read_frag(fr[(j + 1) & 1], Bc + (j + 1) * W4_BTILE, tail);
                    if (XSQ_WINO4_SCHED2 >= 0) __builtin_amdgcn_sched_barrier(XSQ_WINO4_SCHED2);
                    const float wa[2] = {f.w[0].x, f.w[0].y}, wb[2] = {f.w[1].x, f.w[1].y}, wc[2] = {f.w[2].x, f.w[2].y};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], v[i][j], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i], v[i][j], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[i], v[i][j], acc[j][2], 0, 0, 0);
                    }
                    if (tail) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[0], vt[j], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[1], vt[j], acc[j][1], 0, 0, 0);
                        acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.wt[2], vt[j], acc[j][2], 0, 0, 0);
                    }
#pragma unroll
                    for (int cc = 0; cc < NV; ++cc) {
                        if (XSQ_WINO4_ABL & 2) { accv[j][cc] += v[0][j] + f.u[cc]; continue; }
                        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf"
                            : "+v"(accv[j][cc])
                            : "v"(f.u[cc]), "v"(v[0][j]), "v"(v[1][j]));
                        if (tail) asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[j][cc]) : "v"(vt[j]), "v"(f.ut[cc]));
                    }
                    if (XSQ_WINO4_SCHED >= 0) __builtin_amdgcn_sched_barrier(XSQ_WINO4_SCHED);
                }
                // other buffer: last read in the chunk before, every wave is past that chunk's barrier
                if (s < 2) store_chunk(s + 1, cur ^ 1);
                else if (more) store_chunk(0, cur ^ 1);
                cnt += 1;
                if (!(XSQ_WINO4_ABL & 64)) W4_SUB(4, __syncthreads());
            }
            if (more) {                  // every wave is past the barrier behind the slab's last reader
                store_slab();
                W4_SUB(4, __syncthreads());
            }
        }
        W4_STAMP(1);
        // ---- epilogue.  Accumulator register r of this lane is output channel 16 cb + 4 kq + r of the lane's OWN quad, summed over
        // this wave's half of the channels.  Output transform (linear) on the partial sums; the half h = 0 owns outputs 0, 1 of the
        // quad, h = 1 outputs 2, 3: the two outputs a wave does not own go to its partner through LDS (slot x lane float4 images:
        // the h = 0 waves' in the weight buffers, the h = 1 waves' in the planes -- the loop ended on a barrier), the partner adds
        // them to its own partial sums (a + b = b + a: both halves of a quad round alike), shift + ReLU, one 16-byte store per
        // column block and output.
        float yo[2][3][4];               // owned outputs: [ko][cb][r]
        float yv[2][4];                  // owned outputs of the vector columns
        float* xw = (h ? slab : Bs) + g * (8 * 64 * 4) + 4 * lane;
        const float* xr = (h ? Bs : slab) + g * (8 * 64 * 4) + 4 * lane;
        static_assert(4 * 8 * 64 * 4 <= 2 * W4_NC * W4_BTILE && 4 * 8 * 64 * 4 <= 4 * W4_PROWS * W4_SLD, "exchange images do not fit");
        auto part1 = [&](auto HC) {
            constexpr int H = decltype(HC)::value;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                float y[4][4];               // [r][k]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m[7] = {acc[0][cb][r], acc[1][cb][r], acc[2][cb][r], acc[3][cb][r], acc[4][cb][r], acc[5][cb][r], acc[6][cb][r]};
                    wino4_at(m, y[r]);
                }
#pragma unroll
                for (int ko = 0; ko < 2; ++ko) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) yo[ko][cb][r] = y[r][2 * H + ko];
                    *reinterpret_cast<float4*>(&xw[(ko * 3 + cb) * 256]) =
                        make_float4(y[0][2 - 2 * H + ko], y[1][2 - 2 * H + ko], y[2][2 - 2 * H + ko], y[3][2 - 2 * H + ko]);
                }
            }
            float yk[4][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // [k][cc]
#pragma unroll
            for (int cc = 0; cc < NV; ++cc) {
                float m[7], y[4];
#pragma unroll
                for (int j = 0; j < W4_NC; ++j) {        // the four k-quads' partial sums meet here (fixed order)
                    float x = accv[j][cc];
                    x += __shfl_xor(x, 16);
                    x += __shfl_xor(x, 32);
                    m[j] = x;
                }
                wino4_at(m, y);
#pragma unroll
                for (int k = 0; k < 4; ++k) yk[k][cc] = y[k];
            }
#pragma unroll
            for (int ko = 0; ko < 2; ++ko) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) yv[ko][cc] = yk[2 * H + ko][cc];
                *reinterpret_cast<float4*>(&xw[(6 + ko) * 256]) =
                    make_float4(yk[2 - 2 * H + ko][0], yk[2 - 2 * H + ko][1], yk[2 - 2 * H + ko][2], yk[2 - 2 * H + ko][3]);
            }
        };
        if (h) part1(std::integral_constant<int, 1>{}); else part1(std::integral_constant<int, 0>{});
        __syncthreads();

        {
            const float* shift = a.pool + c.shift_off;
            float* out = (TRANSPOSED ? a.act3 : a.act2) + c.out_off;
            const __amdgpu_buffer_rsrc_t ro = buf_rsrc(out, 0x40000000u);
            const int qq = myseg ? pl - c.n0 : c.q0 + pl;
            const int t0 = 4 * qq + 2 * h;                               // first owned output position
            const bool okq = pl < c.nquads;
            const unsigned vo = 4u * (unsigned)((((c.b * c.Fo + c.f0 + myseg) * To) + t0) * CS + 4 * kq);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const float4 sh = *reinterpret_cast<const float4*>(shift + 16 * cb + 4 * kq);
#pragma unroll
                for (int ko = 0; ko < 2; ++ko) {
                    const float4 o = *reinterpret_cast<const float4*>(&xr[(ko * 3 + cb) * 256]);
                    const float4 y = make_float4(fmaxf((yo[ko][cb][0] + o.x) + sh.x, 0.f), fmaxf((yo[ko][cb][1] + o.y) + sh.y, 0.f),
                                                 fmaxf((yo[ko][cb][2] + o.z) + sh.z, 0.f), fmaxf((yo[ko][cb][3] + o.w) + sh.w, 0.f));
                    // (displacements in the LANE offset, scalar offset 0: the store-data hazard of 16-byte stores with an SGPR offset, common.h)
                    buf_st4(y, ro, (okq && t0 + ko < To) ? vo + 64u * cb + 4u * CS * ko : BUF_OOB, 0);
                }
            }
            const float4 sh = *reinterpret_cast<const float4*>(shift + 48);
            const unsigned vt48 = vo - 16u * (unsigned)kq;                   // (channel 0 of the row)
#pragma unroll
            for (int ko = 0; ko < 2; ++ko) {
                const float4 o = *reinterpret_cast<const float4*>(&xr[(6 + ko) * 256]);
                const float4 y = make_float4(fmaxf((yv[ko][0] + o.x) + sh.x, 0.f), fmaxf((yv[ko][1] + o.y) + sh.y, 0.f),
                                             fmaxf((yv[ko][2] + o.z) + sh.z, 0.f), fmaxf((yv[ko][3] + o.w) + sh.w, 0.f));
                buf_st4(y, ro, (okq && kq == 0 && t0 + ko < To) ? vt48 + 192u + 4u * CS * ko : BUF_OOB, 0);
            }
        }
        W4_STAMP(2);
#if XSQ_WINO4_STAMPS
        if (tid == 0) atomicAdd(&g_w4_stamps[3], 1ull);
#endif
    }
}

}  // namespace xsq
