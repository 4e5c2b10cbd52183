// fp32 MFMA tile engine for the grouped (ragged) GEMM-shaped kernels of the hot path:
// per-band DFTs of the sliCQT / isliCQT and the four CDAE layers as implicit GEMMs.
//
// One workgroup = 256 threads = 4 wavefronts (64 lanes) stacked along M.  Tile 128 x 64
// (or 128 x 32 for N tails, TileDev.narrow), K-step 16, v_mfma_f32_32x32x2_f32 (exact fp32:
// the parity bar is 1e-4 RMS against a torch-CPU fp32 reference, so operands stay fp32).
// Operand maps (cdna guide section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C: reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
//
// Both operands are K-contiguous: A comes from an operator-specific loader (gathers,
// reflections, zero padding), B is a dense matrix stored TRANSPOSED, Bt[n][k], padded to
// (K%16==0, N%64==0) with zeros, so its loads are unconditional 16-byte loads.  In LDS a
// tile row is 16 k-values with a stride of 20 floats (80 B: 16-byte aligned and conflict-free
// for ds_read_b128 over 16 consecutive rows).  The sum over k is order-free, so MFMA step i
// of a K-step takes k = i from the lower half-wave and k = 8+i from the upper one: every lane
// reads its 8 operand values as two ds_read_b128, all fragment reads of a K-step are issued
// up front and the 16 MFMAs then run back to back.  LDS is double-buffered (one barrier per
// K-step); the next K-step's global loads are in flight during the MFMAs.
#pragma once
#include <type_traits>

#include "common.h"

// Diagnostic builds only (tools/ablate.sh): bit 0 skip MFMAs, bit 1 skip A global loads,
// bit 2 skip B global loads, bit 3 skip LDS fragment reads, bit 4 skip the epilogue.  0 in the product build.
#ifndef XSQ_ABLATE
#define XSQ_ABLATE 0
#endif

#ifndef XSQ_GEMM_STAMP
#define XSQ_GEMM_STAMP 0     // diagnostic build: phase stamps of the tile engine (tools/gemm_phases.py); one copy per translation unit
#endif

namespace xsq {

#if XSQ_GEMM_STAMP
// per tile: s_memrealtime (100 MHz) at 0 start, 1 first K-step staged, 2 K loop done, 3 epilogue stores issued; [4] = tile kind, [5] = K-steps
constexpr int GEMM_STAMP_TILES = 1 << 17;
static __device__ unsigned long long g_gemm_stamps[GEMM_STAMP_TILES * 8];
#define XSQ_GS(i) do { if (stamped_of<Op>::value && tid == 0 && blockIdx.x < GEMM_STAMP_TILES) g_gemm_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XSQ_GS(i) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Optional second operand stream of an operator (e.g. the masks of the masked synthesis): `typedef ... Aux`,
// `Aux load_aux(g, row, k)` is issued next to load_a4 and `float4 finish(v, aux)` is applied when the K-step
// goes into LDS -- never inside the load phase, where consuming a value means waiting for it.
struct NoAux {};
template <class Op, class = void> struct aux_of { typedef NoAux type; static constexpr bool on = false; };
template <class Op> struct aux_of<Op, std::void_t<typename Op::Aux>> { typedef typename Op::Aux type; static constexpr bool on = true; };

// Optional K cursor of an operator whose operand address is an expensive function of k (layer 1: two integer
// divisions by run-time values per load -- 80 of the 180 vector instructions of its K-step, in a kernel whose K-step
// holds 770 cycles of MFMAs).  The engine asks for the operand of k = s_kq, s_kq + 16, s_kq + 32, ... in that order:
// `Cursor cursor(g, k)` positions a cursor, `load_a4(g, row, cursor)` loads through it, `advance(g, cursor)` moves it
// 16 further; the operator keeps the k-dependent part of the address incrementally.
// operators that carry `static constexpr bool STAMPED = true` are the ones a stamped diagnostic build records
template <class Op, class = void> struct stamped_of { static constexpr bool value = false; };
template <class Op> struct stamped_of<Op, std::void_t<decltype(Op::STAMPED)>> { static constexpr bool value = Op::STAMPED; };
struct NoCursor {};
template <class Op, class = void> struct cursor_of { typedef NoCursor type; static constexpr bool on = false; };
template <class Op> struct cursor_of<Op, std::void_t<typename Op::Cursor>> { typedef typename Op::Cursor type; static constexpr bool on = true; };

// Blocks b and b+8 share an XCD (observed round-robin placement; speed only, never
// correctness).  Tiles that are neighbours in the table share the B matrix of their group and
// often their A rows, so they should meet in one XCD's 4 MiB L2 -- but the work per tile varies
// by 20x across groups, so an XCD must not own one contiguous eighth of the table.  Deal the
// table out in chunks of 16 tiles: chunk c goes to XCD c % 8.  Bijective for any grid size
// (the ragged tail of fewer than 128 tiles keeps the identity order).
#ifndef XSQ_XCD_CHUNK
#define XSQ_XCD_CHUNK 16      // diagnostic builds: -DXSQ_XCD_CHUNK=n
#endif
__device__ inline int xcd_remap(int bid, int nblocks) {
    constexpr int G = XSQ_XCD_CHUNK;
    const int full = (nblocks / (8 * G)) * (8 * G);
    if (bid >= full) return bid;
    const int x = bid & 7, l = bid >> 3;
    return ((l / G) * 8 + x) * G + (l % G);
}

constexpr int GEMM_BM = 128, GEMM_BN = 64, GEMM_BK = 16, GEMM_LD = 20;

__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// MT = 32-row MFMA tiles per wave along M: tile height BM = 128 * MT (MT = 2 for the long, regular
// CDAE layers: twice the MFMAs per barrier and per B-tile load).
// XW = 1 (operators whose 64-column tile holds 52 stored channels, Op::NV real ones past 47; MT = 1): the second
// 32-column block is not padded -- channels 32..47 run on two v_mfma_f32_16x16x4_f32 row blocks and 48..47+NV on
// the vector ALU (cdae_slab.h, MODE 3, explains the layout); the operator supplies epilogue_xw.
// XW = 2 (layer 4, whose N = coefficients per slice is any multiple of 4 from 16 to 292): tiles come in four widths,
// TileDev.narrow = 0: 64 columns (two 32x32 blocks), 1: 32, 3: 48 (32x32 block + 16x16x4 blocks), 2: 16 (16x16x4
// blocks only); the operator supplies epilogue16 for the 16-column blocks.  Padding N to 32/64 cost 38 % there.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// One tile.  KIND >= 0 (XW = 2, layer 4): the tile's width class is a COMPILE-TIME fact of this instantiation -- the kernel
// switches once, outside everything.  With the class a run-time value the K loop carried the MFMA sequences of all four
// classes and, where their accumulator assignments met, 70-90 v_accvgpr_write / v_accvgpr_mov / v_mov per K-step: more
// than half of its vector instructions (round 3, read off the ISA), in a kernel whose K-steps are bound by exactly those
// (a wave that is not issuing MFMAs gets few vector issue slots beside the ones that are).
template <class Op, int MT, int XW, int KIND>
__device__ __forceinline__ void gemm_tile_body(const Op& op, const TileDev t, float* const lds) {
    constexpr int BM = GEMM_BM * MT, BN = GEMM_BN, BK = GEMM_BK, LD = GEMM_LD;
    constexpr int RA = BM / 64;     // A rows staged per thread
    float* const As0 = lds;
    float* const Bs0 = lds + 2 * BM * LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // wave-uniform.  XW = 1 operators have N = 52: one tile, always the full width -- a compile-time fact, or the K-step
    // carries both MFMA sequences and 16 accumulator copies where their register assignments meet
    const bool wide = XW == 1 ? true : (KIND >= 0 ? KIND == 0 : t.narrow == 0);
    const int kind = KIND >= 0 ? KIND : (XW == 2 ? t.narrow : (wide ? 0 : 1));
    const typename Op::Group g = op.group(t.group);
    const int K = (XSQ_ABLATE & 32) ? 16 : g.K;      // bit 5: one K-step only (epilogue cost in isolation)

    // ---- staging assignment: one float4 (4 consecutive k) of one row per thread and slot ----
    const int s_row = tid >> 2;          // 0..63
    const int s_kq = (tid & 3) * 4;      // 0,4,8,12
    typename Op::RowA ra[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = op.row_a(g, t.m0 + s_row + 64 * i);
    // Bt[n][k] through a buffer descriptor: per-thread byte offset once, the K-step as the scalar offset (no vector
    // address arithmetic inside the K loop; the matrix is padded to whole tiles, so no range is needed)
    const __amdgpu_buffer_rsrc_t rB = buf_rsrc(g.B, 0x7FFFFFFFu);
    const unsigned bvo = 4u * (unsigned)((t.n0 + s_row) * g.ldb + s_kq);
    const bool b_on = wide || s_row < (XW == 2 ? (kind == 1 ? 32 : kind == 3 ? 48 : 16) : 32);

    // Global loads run TWO K-steps ahead of the MFMAs (two register sets, loop unrolled by two so
    // the set index is static): with 4 waves sharing a SIMD one K-step lasts about as long as an
    // HBM/L2 round trip, so a distance of one left the loads on the critical path.
    float4 ga[2][RA];
    float4 gb[2];
    typename aux_of<Op>::type gx[2][RA];
    // (Measured: issuing the loads unconditionally -- out-of-range addresses redirected to a zero buffer so that
    // the compiler can count the loads in flight instead of waiting with vmcnt(0) -- made the short-K operators
    // SLOWER (layer 1: 0.74 -> 0.85 ms): two extra load sets per tile and a 64-bit select per address.)
    typename cursor_of<Op>::type kc;               // k = s_kq + 16 * (number of load_set calls so far)
    if constexpr (cursor_of<Op>::on) kc = op.cursor(g, s_kq);
    auto load_set = [&](int set, int k) {          // set is a compile-time constant at every call site; k advances by 16 per call
        if (k < K) {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                if (!(XSQ_ABLATE & 2) || k == 0) {
                    if constexpr (cursor_of<Op>::on) ga[set][i] = op.load_a4(g, ra[i], kc);
                    else ga[set][i] = op.load_a4(g, ra[i], k + s_kq);
                    if constexpr (aux_of<Op>::on) gx[set][i] = op.load_aux(g, ra[i], k + s_kq);
                }
            if (b_on && (!(XSQ_ABLATE & 4) || k == 0)) gb[set] = buf_ld4(rB, bvo, 4 * k);
            if constexpr (cursor_of<Op>::on) op.advance(g, kc);
        }
    };
    auto store_set = [&](int set, int buf) {
        float* Aw = As0 + buf * BM * LD;
        float* Bw = Bs0 + buf * BN * LD;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float4 v = ga[set][i];
            if constexpr (aux_of<Op>::on) v = op.finish(v, gx[set][i]);
            *reinterpret_cast<float4*>(&Aw[(s_row + 64 * i) * LD + s_kq]) = v;
        }
        *reinterpret_cast<float4*>(&Bw[s_row * LD + s_kq]) = gb[set];
    };
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[0][i] = ga[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[0] = gb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    load_set(0, 0);
    const float4 fa_lo = ga[0][0], fa_hi = ga[0][0], fb_lo = gb[0], fb_hi = gb[0];   // ablation builds only

    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; acc1[i][r] = 0.f; }

    const int lrow = lane & 31, lk = lane >> 5;
    const int a_frag = (wave * 32 * MT + lrow) * LD + 8 * lk;
    const int b_frag = lrow * LD + 8 * lk;
    // XW: 16-row blocks (lane = row l & 15, k quad l >> 4) and the vector columns
    const int q16 = lane >> 4;
    const int a16_frag = (wave * 32 + (lane & 15)) * LD + 4 * q16;
    const int b16_frag = ((XW == 2 && kind == 2 ? 0 : 32) + (lane & 15)) * LD + 4 * q16;
    const int bv_frag = 48 * LD + 8 * lk;
    f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float accv[4] = {0.f, 0.f, 0.f, 0.f};

    auto mfma_step = [&](int buf) {
        const float* As = As0 + buf * BM * LD;
        const float* Bs = Bs0 + buf * BN * LD;
        float a[MT][8];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float4 lo = (XSQ_ABLATE & 8) ? fa_lo : *reinterpret_cast<const float4*>(&As[a_frag + i * 32 * LD]);
            const float4 hi = (XSQ_ABLATE & 8) ? fa_hi : *reinterpret_cast<const float4*>(&As[a_frag + i * 32 * LD + 4]);
            a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = lo.z; a[i][3] = lo.w;
            a[i][4] = hi.x; a[i][5] = hi.y; a[i][6] = hi.z; a[i][7] = hi.w;
        }
        const float4 b0_lo = (XSQ_ABLATE & 8) ? fb_lo : *reinterpret_cast<const float4*>(&Bs[b_frag]);
        const float4 b0_hi = (XSQ_ABLATE & 8) ? fb_hi : *reinterpret_cast<const float4*>(&Bs[b_frag + 4]);
        const float b0[8] = {b0_lo.x, b0_lo.y, b0_lo.z, b0_lo.w, b0_hi.x, b0_hi.y, b0_hi.z, b0_hi.w};
        if constexpr (XW == 2) {
            if (kind != 2) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b0[kk], acc0[0], 0, 0, 0);
            }
            if (kind == 0) {
                const float4 b1_lo = *reinterpret_cast<const float4*>(&Bs[b_frag + 32 * LD]);
                const float4 b1_hi = *reinterpret_cast<const float4*>(&Bs[b_frag + 32 * LD + 4]);
                const float b1[8] = {b1_lo.x, b1_lo.y, b1_lo.z, b1_lo.w, b1_hi.x, b1_hi.y, b1_hi.z, b1_hi.w};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b1[kk], acc1[0], 0, 0, 0);
            } else if (kind >= 2) {
                const float4 x0 = *reinterpret_cast<const float4*>(&As[a16_frag]);
                const float4 x1 = *reinterpret_cast<const float4*>(&As[a16_frag + 16 * LD]);
                const float4 y = *reinterpret_cast<const float4*>(&Bs[b16_frag]);
                const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w}, yb[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], yb[j], acc16[0], 0, 0, 0);
                    acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], yb[j], acc16[1], 0, 0, 0);
                }
            }
        } else if constexpr (XW == 1) {
            if (wide) {
                const float4 x0 = *reinterpret_cast<const float4*>(&As[a16_frag]);
                const float4 x1 = *reinterpret_cast<const float4*>(&As[a16_frag + 16 * LD]);
                const float4 y = *reinterpret_cast<const float4*>(&Bs[b16_frag]);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b0[kk], acc0[0], 0, 0, 0);
                const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w}, yb[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], yb[j], acc16[0], 0, 0, 0);
                    acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], yb[j], acc16[1], 0, 0, 0);
                }
#pragma unroll
                for (int cc = 0; cc < Op::NV; ++cc) {
                    const float4 v0 = *reinterpret_cast<const float4*>(&Bs[bv_frag + cc * LD]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&Bs[bv_frag + cc * LD + 4]);
                    asm volatile("v_fmac_f32 %0, %1, %9\n\tv_fmac_f32 %0, %2, %10\n\tv_fmac_f32 %0, %3, %11\n\tv_fmac_f32 %0, %4, %12\n\t"
                                 "v_fmac_f32 %0, %5, %13\n\tv_fmac_f32 %0, %6, %14\n\tv_fmac_f32 %0, %7, %15\n\tv_fmac_f32 %0, %8, %16"
                                 : "+v"(accv[cc])
                                 : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[0][4]), "v"(a[0][5]), "v"(a[0][6]), "v"(a[0][7]),
                                   "v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], b0[kk], acc0[0], 0, 0, 0);
            }
        } else if (wide) {
            const float4 b1_lo = (XSQ_ABLATE & 8) ? fb_hi : *reinterpret_cast<const float4*>(&Bs[b_frag + 32 * LD]);
            const float4 b1_hi = (XSQ_ABLATE & 8) ? fb_lo : *reinterpret_cast<const float4*>(&Bs[b_frag + 32 * LD + 4]);
            const float b1[8] = {b1_lo.x, b1_lo.y, b1_lo.z, b1_lo.w, b1_hi.x, b1_hi.y, b1_hi.z, b1_hi.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (XSQ_ABLATE & 1) { acc0[i][kk] += a[i][kk] * b0[kk]; acc1[i][kk] += a[i][kk] * b1[kk]; continue; }
                    acc0[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b0[kk], acc0[i], 0, 0, 0);
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b1[kk], acc1[i], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (XSQ_ABLATE & 1) { acc0[i][kk] += a[i][kk] * b0[kk]; continue; }
                    acc0[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b0[kk], acc0[i], 0, 0, 0);
                }
        }
    };

    // prologue: K-step 0 -> LDS buffer 0; K-steps 1 and 2 in flight in register sets 1 and 0
    store_set(0, 0);
    XSQ_GS(1);
    load_set(1, BK);
    load_set(0, 2 * BK);
    __syncthreads();

    if constexpr (KIND >= 0) {
        // K-steps in pairs (buffer / register-set parity static), an odd last step behind the loop.  (One loop with a
        // break between its two halves gave the accumulators two exit paths: the compiler copied all of them -- 32-64
        // v_accvgpr_write / v_accvgpr_mov per iteration -- where the paths met.)  The other operators keep the loop
        // below: in this form the masked band synthesis operator needs 136 registers (three workgroups per CU).
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
            __syncthreads();        // (the epilogues reuse the staging buffers)
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 2 * BK) {
            // even K-step: compute buffer 0; set 1 (K-step k0+16) -> buffer 1; reload set 1 with k0+48
            mfma_step(0);
            if (k0 + BK < K) store_set(1, 1);
            load_set(1, k0 + 3 * BK);
            __syncthreads();
            if (k0 + BK >= K) break;
            // odd K-step: compute buffer 1; set 0 (K-step k0+32) -> buffer 0; reload set 0 with k0+64
            mfma_step(1);
            if (k0 + 2 * BK < K) store_set(0, 0);
            load_set(0, k0 + 4 * BK);
            __syncthreads();
        }
    }

    XSQ_GS(2);
#if XSQ_GEMM_STAMP
    if (stamped_of<Op>::value && tid == 0 && blockIdx.x < GEMM_STAMP_TILES) {
        g_gemm_stamps[blockIdx.x * 8 + 4] = kind; g_gemm_stamps[blockIdx.x * 8 + 5] = (K + 15) / 16;
        // where the workgroup ran: HW_ID (wave slot, SIMD, CU, shader array, shader engine) and XCC_ID
        g_gemm_stamps[blockIdx.x * 8 + 6] = (unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        g_gemm_stamps[blockIdx.x * 8 + 7] = (unsigned)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
    }
#endif
    // ---- epilogue: accumulator register r of this lane is row row0 + acc_row(r), columns n and n+32 ----
    if (XSQ_ABLATE & 16) {      // diagnostic: no epilogue (keep the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc0[i][r] + acc1[i][r];
        if (sacc == 1.2345e-30f) __builtin_trap();
        return;
    }
    if constexpr (XW == 1) {
        // (the K loop ended on a barrier: the staging buffers are free, each wave takes 32 x 52 floats of them)
        if (wide) { op.epilogue_xw(g, t.m0 + wave * 32, lane, acc0[0], acc16, accv, lds + wave * 32 * 52); XSQ_GS(3); return; }
    }
    if constexpr (XW == 2) {
        if (kind >= 2) op.epilogue16(g, t.m0 + wave * 32, lane, t.n0 + (kind == 3 ? 32 : 0), acc16);
        if (kind == 2) { XSQ_GS(3); return; }
        op.epilogue(g, t.m0 + wave * 32 + 4 * lk, t.n0 + lrow, acc0[0], acc1[0], kind == 0);
        XSQ_GS(3);
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
        op.epilogue(g, t.m0 + (wave * MT + i) * 32 + 4 * lk, t.n0 + lrow, acc0[i], acc1[i], wide);
}

template <class Op, int MT = 1, int XW = 0>
__global__ __launch_bounds__(256) void grouped_gemm_kernel(Op op, const TileDev* __restrict__ tiles,
                                                            int ntiles) {
    static_assert(XW == 0 || MT == 1, "exact-width columns: MT = 1 only");
    static_assert(XW != 1 || 4 * 32 * 52 <= 2 * (GEMM_BM + GEMM_BN) * GEMM_LD, "XW = 1 epilogue image does not fit the staging buffers");
    __shared__ __attribute__((aligned(16))) float lds[2 * (GEMM_BM * MT + GEMM_BN) * GEMM_LD];
#if XSQ_GEMM_STAMP
    const int tid = threadIdx.x;
#endif
    XSQ_GS(0);
    const TileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    if constexpr (XW == 2) {
        switch (t.narrow) {          // workgroup-uniform
            case 0: gemm_tile_body<Op, MT, XW, 0>(op, t, lds); break;
            case 1: gemm_tile_body<Op, MT, XW, 1>(op, t, lds); break;
            case 2: gemm_tile_body<Op, MT, XW, 2>(op, t, lds); break;
            default: gemm_tile_body<Op, MT, XW, 3>(op, t, lds); break;
        }
    } else {
        gemm_tile_body<Op, MT, XW, -1>(op, t, lds);
    }
}

}  // namespace xsq
