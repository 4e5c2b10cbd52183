// Layers 2 and 3 of the CDAE as DIRECT 4-tap convolutions, with the input slab of a tile held ONCE in LDS.
//
// Which kernel runs where (cdae.hip: cdae_launch_layer):
//   fp32 (the default precision), rows of >= 127 positions : cdae_wino.h (Winograd F(2, 4) along the time taps) -- NOT this file
//   fp32, rows of 86..126 positions, or xsq_model_set_winograd(0) : cdae_slab_kernel<., 3, true> below (MODE 3, exact fp32)
//   bf16x6 / bf16x3 (opt-in, xsq_model_set_precision 2 / 1), rows >= 86 : MODE 2 / MODE 1 below
//   shorter rows, and the training step (raw epilogues) : the generic tile engine (gemm_tile*.h)
//
// Both layers are (kf x 4)-tap convolutions over channels-last activations (model.py:140-170; layer 3 is the
// transposed convolution written as a gather, cdae.hip).  Seen as an implicit GEMM, row (f, t) of the A
// operand is, for every df, the run of 4 x 52 consecutive words that starts at position t of input row
// f +- df: rows t and t + 1 share three of their four positions.  The generic engine loads every row's run
// separately -- each activation crosses the L2 -> LDS path 4 kf times.  Here a tile is 256 consecutive output rows of
// one batch item; for each df its DISTINCT input positions (256 + 3 per touched (b, f) row) are copied once,
// coalesced, into LDS -- one fp32 plane (MODE 0 / 2 / 3; row stride 52 words) or two bf16 planes hi | lo (MODE 1) --
// and every MFMA A fragment is read straight out of that image: the 4x overlap is served by LDS.  B (the weights of
// this df, 52 x 208 words) streams through a double-buffered LDS tile shared by the 8 waves of the workgroup.
//   L2 -> LDS traffic per 256 rows and df:  A 54 KB + B 43 KB   (generic engine, 2 x 128 rows: 212 + 106 KB)
//   LDS: 55.7 KB of slab + two B tiles (7.3 KB; 10.6 KB for bf16x6) -> two 512-thread workgroups per CU.
#pragma once
#include "cdae_api.h"
#include "gemm_tile_bf3.h"
#include "gemm_tile_bf6.h"

#ifndef XSQ_SLAB_PROLOGUE_PRIO
#define XSQ_SLAB_PROLOGUE_PRIO 0
#endif
#ifndef XSQ_SLAB_STAMP
#define XSQ_SLAB_STAMP 0    // diagnostic build: phase stamps of the fp32 slab kernel (tools/slab_phases.py)
#endif
#ifndef XSQ_SLAB_PRIO
#define XSQ_SLAB_PRIO 0     // diagnostic A/B builds: 1 static priority for waves 4..7, 2 priority around every MFMA cluster
#endif
#ifndef XSQ_SLAB_ABL
#define XSQ_SLAB_ABL 0      // diagnostic builds: 1 no MFMAs, 2 no fragment reads, 4 no B loads, 8 no slab loads, 16 no epilogue
#endif

namespace xsq {

constexpr int SLAB_ROWS = 256;                 // output rows per tile (8 waves x 32)
constexpr int SLAB_MAXSEG = 4;                 // (b, f) rows a tile may touch (needs To >= 86)
constexpr int SLAB_POS = SLAB_ROWS + 3 * SLAB_MAXSEG;
constexpr int SLAB_KRUN = 4 * CS;              // 208 words per df
constexpr int SLAB_BLD = 36;                   // B tile row: 32 k-values as (16 hi | 16 lo) x 2 + 4 pad words
constexpr int SLAB_BLD6 = 52;                  // bf16x6: 2 chunks x 3 pieces x 8 words + 4 pad words (13 x 4: conflict-free)
constexpr int SLAB_BN = CS;                    // B tile rows: the 52 stored channels (columns 52..63 of the MFMA tile are never stored)

// TRANSPOSED = false: layer 2, in = act1 (Fi = F1 rows of Ti = T1), input row f + df, positions t .. t + 3
// TRANSPOSED = true : layer 3, in = act2 (Fi = F2 rows of Ti = T2), input row f - df, positions t - 3 .. t
// MODE 1 (bf16x3): operands in the split format, v_mfma_f32_32x32x16_bf16 x 3 (gemm_tile_bf3.h)
// MODE 0 (fp32)  : plain fp32 operands, v_mfma_f32_32x32x2_f32 with the engine's k permutation (MFMA i of a
//              16-value chunk takes k = i from the lower half-wave and k = 8 + i from the upper one, gemm_tile.h):
//              the slab is then ONE fp32 plane (same 55.7 KB; row stride 52 words, conflict-free ds_read_b128)
// MODE 3 (fp32, exact-width columns): as MODE 0, but the 52 stored channels are not padded to 64 MFMA columns:
//              columns 0..31 on v_mfma_f32_32x32x2_f32, 32..47 on two v_mfma_f32_16x16x4_f32 row blocks, 48..51 on
//              the vector ALU (32 v_fmac per lane and chunk, on the A fragment the lane already holds; the two
//              half-waves' partial sums meet in the epilogue): 768 instead of 1024 matrix-pipe cycles per 32 rows
//              x 16 k.  (v_mfma_f32_4x4x1_16B_f32 would fit the four columns exactly -- lane maps probed with
//              tools/probe/mfma4x4.hip -- but measured ~7x slower than the whole rest of the kernel.)
// MODE 2 (bf16x6): plain fp32 operands, fp32 slab plane; the A fragment is cut into its three bf16 pieces when
//              it is read (gemm_tile_bf6.h), the B tile holds the three pieces of the weights; six bf16 MFMAs
// LATE: the next df's slab is fetched in the slot that writes it (slot 7, no MFMAs: two load -> store halves whose
//       registers are dead fragment registers) instead of being requested a whole df ahead and held in 28 VGPRs
//       across the MFMA slots -- under the 128-VGPR cap of two workgroups per CU those registers were spilled
//       (96-104 B of scratch per lane, 2.4 GB of HBM traffic per launch against 1.0 GB of activations, r01).  The
//       round trip is exposed to this workgroup only; the other three waves of the SIMD keep the matrix pipe busy.
#if XSQ_SLAB_STAMP
// per tile (layer 3 only): s_memrealtime at 0 start, 1 prologue done (first slab + B tile in LDS), 2 slot loop done, 3 epilogue
// stores issued; [4] = kf, [5] = summed duration of the slab-fetch slots (slot 7 of every df but the last), [6] HW_ID, [7] XCC_ID
constexpr int SLAB_STAMP_TILES = 1 << 16;
static __device__ unsigned long long g_slab_stamps[SLAB_STAMP_TILES * 8];
static __device__ unsigned long long g_slab_stamps2[SLAB_STAMP_TILES * 4];    // prologue detail: 0 tile entry read, 1 slab loads issued, 2 slab data arrived, 3 LDS stores done
#define XSQ_SS(i) do { if (TRANSPOSED && tid == 0 && blockIdx.x < SLAB_STAMP_TILES) g_slab_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSQ_SS(i) do { } while (0)
#endif

// One tile WITH everything the prologue needs of its (block, target): the kernel used to read the tile, then the block
// descriptor it points at, then request the slab -- three dependent round trips before the first MFMA, and stamps
// (tools/slab_phases.py) put the prologue at 8-13 us, 19 % of the summed tile time.  All offsets in floats.
struct SlabTileDev {               // 64 bytes: one scalar load
    int m0, kf, Fo, Fi;
    int64_t in_off, out_off;       // input / output activations of the (block, target), relative to the layer's arenas
    int64_t shift_off, w_off;      // shift vector / weight matrix inside the pool
    int b, f0, t0, pad;            // batch item, first output row (f0, t0) of the tile: m0 = b * Fo * To + f0 * To + t0
};
static_assert(sizeof(SlabTileDev) == 64, "SlabTileDev is meant to be one 64-byte scalar load");

template <bool TRANSPOSED, int MODE, bool LATE = false>
__global__ __launch_bounds__(512, 4) void cdae_slab_kernel(CdaeArgs a, const SlabTileDev* __restrict__ tiles, int ntiles) {
    constexpr bool BF3 = MODE == 1, BF6 = MODE == 2, EXW = MODE == 3;
#ifndef XSQ_SLAB_EARLY2
#define XSQ_SLAB_EARLY2 1
#endif
    constexpr bool EARLY2 = XSQ_SLAB_EARLY2 != 0;
    constexpr bool F32T = MODE == 0 || MODE == 3;                // fp32 B tile / fp32 slab plane
    constexpr int BLD = BF6 ? SLAB_BLD6 : SLAB_BLD;               // words per B tile row
    constexpr int PLANE = SLAB_POS * CS + 64;                    // bf16 elements per plane (a multiple of 8)
    __shared__ __attribute__((aligned(16))) unsigned short slab_raw[2 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned Bs[2 * SLAB_BN * BLD];
    unsigned short* const slabH = slab_raw;                      // BF3: hi plane | lo plane
    unsigned short* const slabL = slab_raw + PLANE;
    float* const slabF = reinterpret_cast<float*>(slab_raw);     // fp32: one plane over the same bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    XSQ_SS(0);
#if XSQ_SLAB_PROLOGUE_PRIO
    // A new workgroup's waves are the YOUNGEST on their SIMDs: beside the resident workgroup's MFMA stream they get the
    // leftover vector-issue slots, and the few hundred instructions in front of the first slab request took 5-10 us
    // (stamps).  Raised priority until the prologue's requests are out.
    __builtin_amdgcn_s_setprio(XSQ_SLAB_PROLOGUE_PRIO);
#endif
    const SlabTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    // every field in a scalar register HERE: left alone, the compiler fetched the entry in three pieces at three places
    // of the prologue, each behind its own s_waitcnt -- three dependent round trips in front of the first slab request
    asm volatile("" :: "s"(t.m0), "s"(t.kf), "s"(t.Fo), "s"(t.Fi), "s"(t.in_off), "s"(t.out_off), "s"(t.shift_off), "s"(t.w_off),
                 "s"(t.b), "s"(t.f0), "s"(t.t0));
    const int kf = t.kf;
#if XSQ_SLAB_STAMP
#define XSQ_SS2(i) do { if (TRANSPOSED && tid == 0 && blockIdx.x < SLAB_STAMP_TILES) g_slab_stamps2[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    if (kf == 12345) __builtin_trap();
#else
#define XSQ_SS2(i) do { } while (0)
#endif
    const int Fo = t.Fo, To = TRANSPOSED ? a.T1 : a.T2;
    const int Fi = t.Fi, Ti = TRANSPOSED ? a.T2 : a.T1;
    const float* in = (TRANSPOSED ? a.act2 : a.act1) + t.in_off;
    const float* Bt = (BF3 ? a.poolB : a.pool) + t.w_off;
    const int ldb = kf * SLAB_KRUN;

    // ---- the tile: rows m0 .. m0 + nrows of batch item b, split into segments (one per (b, f) row) --------
    const int perb = Fo * To;
    const int b = t.b;                                          // (host-side: the two divisions cost the prologue ~100 instructions)
    const int mend = min((b + 1) * perb, t.m0 + SLAB_ROWS);     // the epilogue's row bound
    const int nrows = mend - t.m0;
    const int f0 = t.f0, t0 = t.t0;
    // segment i: output rows [seg_r[i], seg_r[i+1]) of the tile, f = f0 + i, first t = (i ? 0 : t0),
    // slab positions start at seg_r[i] + 3 i
    // (position j of a segment is input position  ts - PAD + j,  PAD = 3 for the transposed layer)
    auto seg_start = [&](int i) { return i == 0 ? 0 : min(nrows, i * To - t0); };

    // A-fragment base of this lane's row (bf16 index into the planes)
    const int myrow = wave * 32 + lrow;
    int my_seg = 0;
#pragma unroll
    for (int i = 1; i < SLAB_MAXSEG; ++i) my_seg += (myrow >= seg_start(i)) ? 1 : 0;
    const int a_base = (myrow + 3 * my_seg) * CS + 8 * lk;          // + k of the fragment

    // ---- staging assignments ------------------------------------------------------------------------------
    // Slab staging.  Lane = (position lane p0 = tid / 13 of 39, channel quad c4 = tid % 13) over 507 of the 512 threads;
    // load q of a lane is slab position p0 + 39 q -- no division per load (the first form divided tid + 512 q by 13 seven
    // times per slab and searched the segment with min() chains per element: stamps put the prologue's address
    // arithmetic at 5-10 us per tile, the slab data itself arrived 0.6-0.9 us after the request).  Per segment i the
    // uniform quantities are formed once per slab: first slab position A_i, float offset G_i of that position in the
    // (block, target)'s input, and the range [lo_i, hi_i) of in-segment positions j that exist.
    constexpr int SLAB_PL = 512 / (CS / 4);                         // position lanes (39)
    constexpr int NLD = (SLAB_POS + SLAB_PL - 1) / SLAB_PL;          // float4 loads per thread and slab (7)
    static_assert(NLD == 7, "slab staging: seven loads per lane");
    const int s_p0 = tid / (CS / 4), s_c4 = tid - s_p0 * (CS / 4);
    const bool s_on = tid < SLAB_PL * (CS / 4);
    float4 sa[NLD];
#if XSQ_SLAB_STAMP
    bool stamp_loads = true;
#endif

    // What of a load does not depend on the frequency tap df is formed ONCE per tile: the segment of its slab position
    // (one-hot, 4 bits per load), whether the position exists inside the segment, and its byte offset in the (block,
    // target)'s input at df = 0 -- or BUF_OOB.  A tap then costs a load five vector instructions (segment bit against the
    // tap's mask of existing input rows, select, displacement) instead of ~30 (segment search, four selects per segment
    // bound, range test, zero fill under an exec mask, address): 210 of the ~520 non-MFMA vector instructions of a tap,
    // which on this part are not hidden behind the MFMAs (band_dft4.h, "Vector issue").  Buffer loads return the zeros.
    static_assert(SLAB_MAXSEG <= 4, "one-hot segment code: four bits per load");
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(in, 0x40000000u);      // (a (block, target)'s input is < 2^30 bytes: cdae_launch_layer)
    unsigned s_vo[NLD], s_seg = 0;
#pragma unroll
    for (int q = 0; q < NLD; ++q) s_vo[q] = BUF_OOB;
    // Segment by segment, empty segments skipped by a scalar branch (To >= 256: a tile touches one or two (b, f) rows): per
    // segment and load an unsigned range test, an offset and two selects.  (Load by load with a segment search and a four-way
    // select of the segment's constants this block was 175 of the prologue's 320 vector instructions.)
#pragma unroll
    for (int i = 0; i < SLAB_MAXSEG; ++i) {
        const int st = seg_start(i);
        const int rows_i = (i + 1 < SLAB_MAXSEG ? seg_start(i + 1) : nrows) - st;      // output rows of the segment
        const int f = f0 + i, ts = i ? 0 : t0;
        const int first = ts - (TRANSPOSED ? 3 : 0);                                    // input position of j = 0
        if ((XSQ_SLAB_ABL & 8) || f >= Fo || rows_i <= 0) continue;                      // (uniform)
        const int A = st + 3 * i;                                                       // first slab position of the segment
        const int G = ((b * Fi + f) * Ti + first) * CS;                                 // float offset of that position, input row f (df = 0)
        const int lo = first < 0 ? -first : 0, hi = min(rows_i + 3, Ti - first);        // j with 0 <= first + j < Ti and j < rows_i + 3
        const int j0 = s_p0 - A;
        const unsigned base = 4u * (unsigned)(G + j0 * CS + 4 * s_c4);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int j = j0 + SLAB_PL * q;
            const bool in = s_on && (unsigned)(j - lo) < (unsigned)(hi - lo);
            s_vo[q] = in ? base + 4u * (unsigned)(SLAB_PL * q * CS) : s_vo[q];
            s_seg |= in ? 1u << (4 * q + i) : 0u;
        }
    }
    auto load_slab = [&](int df, int q0 = 0, int q1 = NLD) {
        unsigned okm = 0;                         // segments whose input row f -+ df exists (uniform)
#pragma unroll
        for (int i = 0; i < SLAB_MAXSEG; ++i) {
            const int fi = TRANSPOSED ? f0 + i - df : f0 + i + df;
            okm |= (fi >= 0 && fi < Fi) ? 1u << i : 0u;
        }
        const unsigned delta = 4u * (unsigned)((TRANSPOSED ? -df : df) * Ti * CS);      // (a switched-off offset stays past the range: |delta| << 2^30)
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const bool on = ((s_seg >> (4 * q)) & okm) != 0;
            sa[q] = buf_ld4(rin, on ? s_vo[q] + delta : BUF_OOB, 0);
#if XSQ_SLAB_STAMP
            if (stamp_loads && (q == 0 || q == 3)) { XSQ_SS2(q == 0 ? 0 : 3); }      // (reuses two prologue-detail slots: request 0 / request 3 issued)
#endif
        }
    };
    auto store_slab = [&](int q0 = 0, int q1 = NLD) {
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const int pos = s_p0 + SLAB_PL * q;
            const int e = pos * (CS / 4) + s_c4;
            const bool on = s_on && pos < SLAB_POS;
            if (!BF3) {
                if (on) *reinterpret_cast<float4*>(&slabF[4 * e]) = sa[q];
            } else if (on) {
                const unsigned e0 = __builtin_bit_cast(unsigned, sa[q].x), e1 = __builtin_bit_cast(unsigned, sa[q].y);
                const unsigned e2 = __builtin_bit_cast(unsigned, sa[q].z), e3 = __builtin_bit_cast(unsigned, sa[q].w);
                *reinterpret_cast<uint2*>(&slabH[4 * e]) = make_uint2(__builtin_amdgcn_perm(e1, e0, 0x07060302u), __builtin_amdgcn_perm(e3, e2, 0x07060302u));
                *reinterpret_cast<uint2*>(&slabL[4 * e]) = make_uint2(__builtin_amdgcn_perm(e1, e0, 0x05040100u), __builtin_amdgcn_perm(e3, e2, 0x05040100u));
            }
        }
    };
    // ---- B stream: one K-step = 52 rows x 32 words, loaded by threads 0..415 (one float4 each) two slots
    // before it is written into the LDS tile that the NEXT slot reads (two tiles, two register sets) ------------
    const bool b_ld = tid < SLAB_BN * 8;
    const int b_row = tid >> 3, b_k = (tid & 7) * 4;                 // word offset inside the K-step
    const float* bp = Bt + (int64_t)b_row * ldb + b_k;
    float4 gb[2];
    auto load_b = [&](int set, int df, int ks) {
        if (b_ld && !(XSQ_SLAB_ABL & 4)) gb[set] = (32 * ks + b_k < SLAB_KRUN) ? *reinterpret_cast<const float4*>(bp + df * SLAB_KRUN + 32 * ks)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_b = [&](int set, int buf) {
        if (!b_ld) return;
        unsigned* row = Bs + buf * SLAB_BN * BLD + b_row * BLD;
        if (F32T) { *reinterpret_cast<float4*>(row + b_k) = gb[set]; return; }
        const int c = b_k >> 4, q2 = (b_k & 15) >> 1;
        if (BF6) {
            unsigned a1, a2, a3, b1, b2, b3;
            bf6_cut2(gb[set].x, gb[set].y, a1, a2, a3);
            bf6_cut2(gb[set].z, gb[set].w, b1, b2, b3);
            *reinterpret_cast<uint2*>(row + 24 * c + q2) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(row + 24 * c + 8 + q2) = make_uint2(a2, b2);
            *reinterpret_cast<uint2*>(row + 24 * c + 16 + q2) = make_uint2(a3, b3);
            return;
        }
        const unsigned e0 = __builtin_bit_cast(unsigned, gb[set].x), e1 = __builtin_bit_cast(unsigned, gb[set].y);
        const unsigned e2 = __builtin_bit_cast(unsigned, gb[set].z), e3 = __builtin_bit_cast(unsigned, gb[set].w);
        *reinterpret_cast<uint2*>(row + 16 * c + q2) = make_uint2(__builtin_amdgcn_perm(e1, e0, 0x07060302u), __builtin_amdgcn_perm(e3, e2, 0x07060302u));
        *reinterpret_cast<uint2*>(row + 16 * c + 8 + q2) = make_uint2(__builtin_amdgcn_perm(e1, e0, 0x05040100u), __builtin_amdgcn_perm(e3, e2, 0x05040100u));
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    auto afrag = [&](const unsigned short* plane, int k) {
        const uint2 lo = *reinterpret_cast<const uint2*>(&plane[a_base + k]);
        const uint2 hi = *reinterpret_cast<const uint2*>(&plane[a_base + k + 4]);
        return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
    };
    auto bfrag = [&](const unsigned* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p)); };
    const int b_frag0 = lrow * BLD + (F32T ? 8 : 4) * lk;
    const int b_frag1 = min(lrow + 32, SLAB_BN - 1) * BLD + (F32T ? 8 : 4) * lk;     // columns >= 52 are never stored: any row will do
#define XSQ_MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
    // MODE 3: operands of the two 16-row blocks (lane = row l & 15, k quad q = l >> 4) and of the vector columns
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int q16 = lane >> 4;
    int a16_base[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int r16 = wave * 32 + 16 * rb + (lane & 15);
        int sg = 0;
#pragma unroll
        for (int i = 1; i < SLAB_MAXSEG; ++i) sg += (r16 >= seg_start(i)) ? 1 : 0;
        a16_base[rb] = (r16 + 3 * sg) * CS + 4 * q16;
    }
    const int b16_frag = (32 + (lane & 15)) * BLD + 4 * q16;
    const int bv_frag = 48 * BLD + 8 * lk;
    f32x4_t acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    constexpr int NV = TRANSPOSED ? H1 - 48 : H2 - 48;        // real channels past 47: 2 (layer 3 -> 50) or 3 (layer 2 -> 51); the rest is zero padding
    float accv[4] = {0.f, 0.f, 0.f, 0.f};

    // Slots: every df owns 8 slots -- 7 K-steps of 32 (the last half empty) and one in which the next slab goes
    // into LDS -- so that slot parity = LDS tile = register set and every index below is compile-time.  The
    // K-step of slot s is loaded in slot s - 3, written to tile s & 1 in slot s - 1 (last read in slot s - 2).
    constexpr int NKS = 7;
#if XSQ_SLAB_PRIO == 1       // diagnostic A/B: static priority for the second-dispatched half of the workgroup
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
#if XSQ_SLAB_STAMP
    if (a_base == -12345 || a16_base[0] == -12345 || a16_base[1] == -12345 || s_p0 == -12345) __builtin_trap();    // per-lane setup complete
    if (TRANSPOSED && tid == 0 && blockIdx.x < SLAB_STAMP_TILES) g_slab_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    load_slab(0);
    load_b(0, 0, 0);
    XSQ_SS2(1);
#if XSQ_SLAB_STAMP
    if (sa[0].x == 1.2345e-30f) __builtin_trap();
    XSQ_SS2(2);
#endif
    store_slab();
    store_b(0, 0);
#if XSQ_SLAB_STAMP
    stamp_loads = false;
#endif
    load_b(1, 0, 1);
    load_b(0, 0, 2);
    if (kf > 1 && !LATE) load_slab(1);
#if XSQ_SLAB_PROLOGUE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    XSQ_SS(1);
#if XSQ_SLAB_STAMP
    unsigned long long fetch_t = 0;
#endif
    for (int df = 0; df < kf; ++df) {
        const bool more = df + 1 < kf;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < NKS) {
                // LATE: two of the next slab's seven 16-byte loads per lane leave one slot early (slot 6 holds half a K-step:
                // its fragment registers leave room), so that slot 7 is ONE request -> store round trip instead of two
                if constexpr (LATE && EARLY2) { if (ks == NKS - 1 && more) load_slab(df + 1, 0, XSQ_SLAB_EARLY2 == 1 ? 2 : XSQ_SLAB_EARLY2); }
                const unsigned* Bb = Bs + (ks & 1) * SLAB_BN * BLD;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int k = 32 * ks + 16 * c;
                    if (k < SLAB_KRUN) {       // compile-time: the second half of the last K-step does not exist
                        if constexpr (MODE == 3) {
#if XSQ_SLAB_PRIO == 2       // diagnostic A/B: priority raised around every MFMA cluster
                            __builtin_amdgcn_s_setprio(1);
#endif
                            const float* Bf = reinterpret_cast<const float*>(Bb);
                            const float4 alo = *reinterpret_cast<const float4*>(&slabF[a_base + k]);
                            const float4 ahi = *reinterpret_cast<const float4*>(&slabF[a_base + k + 4]);
                            const float4 p0 = *reinterpret_cast<const float4*>(&Bf[b_frag0 + 16 * c]), p1 = *reinterpret_cast<const float4*>(&Bf[b_frag0 + 16 * c + 4]);
                            const float4 x0 = *reinterpret_cast<const float4*>(&slabF[a16_base[0] + k]);
                            const float4 x1 = *reinterpret_cast<const float4*>(&slabF[a16_base[1] + k]);
                            const float4 y = *reinterpret_cast<const float4*>(&Bf[b16_frag + 16 * c]);
                            const float av[8] = {alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w};
                            const float b0[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                            for (int kk = 0; kk < 8; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                            const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w}, yb[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (XSQ_SLAB_ABL & 64) continue;
                                acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], yb[j], acc16[0], 0, 0, 0);
                                acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], yb[j], acc16[1], 0, 0, 0);
                            }
#pragma unroll
                            for (int cc = 0; cc < NV; ++cc) {  // channels 48.. on the vector ALU: this lane's row, its 8 k-values
                                const float4 v0 = *reinterpret_cast<const float4*>(&Bf[bv_frag + cc * BLD + 16 * c]);
                                const float4 v1 = *reinterpret_cast<const float4*>(&Bf[bv_frag + cc * BLD + 16 * c + 4]);
                                // volatile asm: plain fmaf chains were sunk past the K-step barrier with their operands spilled
                                // (2 KB of scratch per lane, 7x slower); packed v_pk_fma_f32 is not available in this build (Makefile)
                                asm volatile("v_fmac_f32 %0, %1, %9\n\tv_fmac_f32 %0, %2, %10\n\tv_fmac_f32 %0, %3, %11\n\tv_fmac_f32 %0, %4, %12\n\t"
                                             "v_fmac_f32 %0, %5, %13\n\tv_fmac_f32 %0, %6, %14\n\tv_fmac_f32 %0, %7, %15\n\tv_fmac_f32 %0, %8, %16"
                                             : "+v"(accv[cc])
                                             : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]),
                                               "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
                            }
#if XSQ_SLAB_PRIO == 2
                            __builtin_amdgcn_s_setprio(0);
#endif
                        } else if constexpr (MODE == 0) {
                            const float* Bf = reinterpret_cast<const float*>(Bb);
                            const float4 alo = *reinterpret_cast<const float4*>(&slabF[a_base + k]);
                            const float4 ahi = *reinterpret_cast<const float4*>(&slabF[a_base + k + 4]);
                            const float4 p0 = *reinterpret_cast<const float4*>(&Bf[b_frag0 + 16 * c]), p1 = *reinterpret_cast<const float4*>(&Bf[b_frag0 + 16 * c + 4]);
                            const float4 q0 = *reinterpret_cast<const float4*>(&Bf[b_frag1 + 16 * c]), q1 = *reinterpret_cast<const float4*>(&Bf[b_frag1 + 16 * c + 4]);
                            const float av[8] = {alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w};
                            const float b0[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                            const float b1[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                            for (int kk = 0; kk < 8; ++kk) {
                                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b0[kk], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b1[kk], acc1, 0, 0, 0);
                            }
                        } else if constexpr (MODE == 2) {
                            const float4 alo = *reinterpret_cast<const float4*>(&slabF[a_base + k]);
                            const float4 ahi = *reinterpret_cast<const float4*>(&slabF[a_base + k + 4]);
                            unsigned u1[4], u2[4], u3[4];
                            bf6_cut2(alo.x, alo.y, u1[0], u2[0], u3[0]);
                            bf6_cut2(alo.z, alo.w, u1[1], u2[1], u3[1]);
                            bf6_cut2(ahi.x, ahi.y, u1[2], u2[2], u3[2]);
                            bf6_cut2(ahi.z, ahi.w, u1[3], u2[3], u3[3]);
                            const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, make_uint4(u1[0], u1[1], u1[2], u1[3]));
                            const bf16x8_t a2 = __builtin_bit_cast(bf16x8_t, make_uint4(u2[0], u2[1], u2[2], u2[3]));
                            const bf16x8_t a3 = __builtin_bit_cast(bf16x8_t, make_uint4(u3[0], u3[1], u3[2], u3[3]));
                            const unsigned* B0 = Bb + b_frag0 + 24 * c;
                            const unsigned* B1 = Bb + b_frag1 + 24 * c;
                            const bf16x8_t p1 = bfrag(B0), p2 = bfrag(B0 + 8), p3 = bfrag(B0 + 16);
                            const bf16x8_t q1 = bfrag(B1), q2 = bfrag(B1 + 8), q3 = bfrag(B1 + 16);
                            XSQ_MF(a1, p3, acc0); XSQ_MF(a1, q3, acc1);      // smallest terms first
                            XSQ_MF(a3, p1, acc0); XSQ_MF(a3, q1, acc1);
                            XSQ_MF(a2, p2, acc0); XSQ_MF(a2, q2, acc1);
                            XSQ_MF(a1, p2, acc0); XSQ_MF(a1, q2, acc1);
                            XSQ_MF(a2, p1, acc0); XSQ_MF(a2, q1, acc1);
                            XSQ_MF(a1, p1, acc0); XSQ_MF(a1, q1, acc1);
                        } else {
                            const bf16x8_t ah = afrag(slabH, k), al = afrag(slabL, k);
                            const bf16x8_t b0h = bfrag(Bb + b_frag0 + 16 * c), b0l = bfrag(Bb + b_frag0 + 16 * c + 8);
                            const bf16x8_t b1h = bfrag(Bb + b_frag1 + 16 * c), b1l = bfrag(Bb + b_frag1 + 16 * c + 8);
                            XSQ_MF(al, b0h, acc0); XSQ_MF(al, b1h, acc1);
                            XSQ_MF(ah, b0l, acc0); XSQ_MF(ah, b1l, acc1);
                            XSQ_MF(ah, b0h, acc0); XSQ_MF(ah, b1h, acc1);
                        }
                    }
                }
            } else if (more) {                 // slot 7: every wave has passed the barrier behind the slab's last reader
                if constexpr (LATE) {
#if XSQ_SLAB_STAMP
                    const unsigned long long f0 = __builtin_amdgcn_s_memrealtime();
#endif
                    if constexpr (EARLY2) { load_slab(df + 1, XSQ_SLAB_EARLY2 == 1 ? 2 : XSQ_SLAB_EARLY2, NLD); store_slab(0, NLD); }
                    else {
                        load_slab(df + 1, 0, 4); store_slab(0, 4);
                        load_slab(df + 1, 4, NLD); store_slab(4, NLD);
                    }
#if XSQ_SLAB_STAMP
                    fetch_t += __builtin_amdgcn_s_memrealtime() - f0;
#endif
                } else {
                    store_slab();
                    if (df + 2 < kf) load_slab(df + 2);
                }
            }
            // K-step of slot s1 = ks + 1 -> tile s1 & 1 (read last in slot ks - 1); its register set is then refilled
            // with the K-step of slot s3 = ks + 3
            if (ks + 1 < NKS) store_b((ks + 1) & 1, (ks + 1) & 1);
            else if (ks + 1 == 8 && more) store_b(0, 0);                               // (df + 1, 0)
            if (ks + 3 < NKS) load_b((ks + 3) & 1, df, ks + 3);
            else if (ks + 3 >= 8 && more) load_b((ks + 3) & 1, df + 1, ks + 3 - 8);
            __syncthreads();
        }
    }
#undef XSQ_MF
    XSQ_SS(2);
#if XSQ_SLAB_STAMP
    if (TRANSPOSED && tid == 0 && blockIdx.x < SLAB_STAMP_TILES) {
        g_slab_stamps[blockIdx.x * 8 + 4] = kf; g_slab_stamps[blockIdx.x * 8 + 5] = fetch_t;
        g_slab_stamps[blockIdx.x * 8 + 6] = (unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    }
#endif

    // (the epilogue's pointers are formed HERE, from the tile entry: formed in the prologue they were held -- spilled, 12 B
    //  of scratch per lane -- across the whole slot loop)
    CdaeGroup g;
    g.out = (TRANSPOSED ? a.act3 : a.act2) + t.out_off;
    g.shift = a.pool + t.shift_off;
    g.M = mend;
    if (XSQ_SLAB_ABL & 16) {      // every accumulator set of the mode stays live (or its MFMAs would be removed with the epilogue)
        float sacc = acc0[0] + acc0[7];
        if constexpr (EXW) sacc += acc16[0][1] + acc16[1][2] + accv[0] + accv[1] + accv[2] + accv[3];
        else sacc += acc1[1];
        if (sacc == 1.2345e-30f) __builtin_trap();
        return;
    }
    if constexpr (EXW) {
        // (the slot loop ended on a barrier: the slab planes are free; each wave transposes its 32 x 52 block through
        //  6.6 KB of them and stores it as whole 16-byte lanes -- relu_shift_epilogue_xw, cdae.hip)
        static_assert(8 * 32 * CS * 4 <= 2 * PLANE * 2, "epilogue images do not fit the slab planes");
        relu_shift_epilogue_xw(g, t.m0 + wave * 32, lane, acc0, acc16, accv, NV, slabF + wave * 32 * CS);
        XSQ_SS(3);
        return;
    }
    relu_shift_epilogue(g, t.m0 + wave * 32 + 4 * lk, lrow, acc0, acc1, false, BF3);
}

}  // namespace xsq
