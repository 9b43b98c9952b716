// All weight gradients of a PNA layer from ONE launch (+ one fixed-order reduction).
//
// Replaces the backward of nn.Linear w.r.t. its weight (reference models/base_layers.py:101, autograd: dW = dY^T X) for
// every Linear of a PNA layer at once: posttrans h-block, the per-degree posttrans blocks, the later pretrans blocks
// (against a BatchNorm output that was never materialised), the [W_s | W_d] halves of the edge block and the bond-table
// block.  Round 2 issued them as 5 split-K GEMMs + 1 row-segment GEMM + 6 slice reductions + a weight fold-back per layer
// (18 + 4 + 22 launches per step, ~25 % of the kernel time at 22-28 % of the fp32 MFMA peak): 32x32 output tiles
// re-read both operands 7x through L2 and a wave held ONE accumulator.
//
// Shape of the problem: C[M,N] = sum_k A[row(k), m] B[row(k), n] with M, N a few hundred and K = rows of the batch
// (8-17 k).  Both operands are k-major as they lie in memory (a row of dY and a row of X), so a K-tile is staged in LDS
// as it is read - T[k][idx] - and a lane's MFMA fragment is one ds_read_b32.  A workgroup (4 waves, one per SIMD) owns a
// PANEL of up to 208 x 208 outputs = 13 x 13 tiles of v_mfma_f32_16x16x4_f32 for one K-slice: a wave holds 43 independent
// accumulator tiles (3 tile columns x 13 tile rows + its share of the 13th column), so per k-step of 4 rows it issues 43
// MFMAs against 21 fragment reads and every operand row is read from memory ONCE per panel.  F = 200 pads to 208 (8 %
// waste; 32-wide tiles would pad to 224 = 25 %).
//
// Staging is LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  A wave's load of one
// operand row lands as 1 KiB at a wave-uniform LDS address, lane-linear - so the row pitch is 256 floats, and the bank
// steering of the 16-wide fragment reads (rows k and k + 1 of a 32-lane access must hit different bank halves) is done
// on the SOURCE side: LDS float4 slot p of row k holds source float4 p ^ ((k & 1) << 2), a fragment read of column idx in
// row k goes to idx ^ ((k & 1) << 4) (the same involution on both sides).  Rows past the slice, padding rows (-1) and
// columns past the panel are loaded from an out-of-range offset: the buffer unit returns 0 to LDS.  The loads are inline
// asm: with the builtin the compiler, which knows that they write LDS, drains vmcnt right after issuing them (before the
// fragment reads of the tile being computed) - the whole global latency exposed once per K-tile; as asm they are ordered
// by the explicit s_waitcnt vmcnt(0) in front of the tile's barrier, and they are issued in four pieces BETWEEN the MFMA
// groups of the tile (an in-order wave with one MFMA in flight issues a handful of other instructions per MFMA for free).
//
// The K-slices of all problems of a launch are work units of one grid (problem table in the kernel arguments); every unit
// stores its partial panel to a scratch slab in MFMA-tile order (fully coalesced 1 KiB per tile), and ONE reduction
// launch sums the slices of every output in a fixed order (bit-deterministic, no atomics) with the problem's epilogue:
// plain store (two-block outputs), the BatchNorm fix-up of fused_bn.hip, or the fold-back of the per-degree gradients
// into the scaler blocks dW_s = sum_D c_s(D) dW_D (grouped.hip).
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace i3d {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// PREC of the panel kernel: how a product a * b of two fp32 operands is formed
//   0: exact, on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 32 cycles per 16x16x4)
//   1: bf16 matmul mode - both operands rounded to bf16 (RNE) when a lane has read its fragment, ONE v_mfma_f32_16x16x16_bf16
//      per 16 rows of k (the rounding the other GEMMs of that mode apply), fp32 accumulation
//   2: split operands a = a_h + a_l (a_h = bf16(a), a_l = bf16(a - a_h)): a_h b_h + a_h b_l + a_l b_h on the bf16 pipe, fp32
//      accumulation - the a_l b_l term and the rounding of the low parts are dropped: |error| <= ~2^-15 |a b| per product
//      against 2^-24 of the exact form (opt-in: I3D_WGRAD_SPLIT_BF16=1)
constexpr int PREC_F32 = 0, PREC_BF16 = 1, PREC_BF16X3 = 2;

__device__ __forceinline__ void split4(const float v[4], shortx4& hi, shortx4& lo, const bool want_lo) {
    const bf16x4_t h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    hi = __builtin_bit_cast(shortx4, h);
    if (want_lo) {
        const bf16x4_t l = {(__bf16)(v[0] - (float)h[0]), (__bf16)(v[1] - (float)h[1]), (__bf16)(v[2] - (float)h[2]),
                            (__bf16)(v[3] - (float)h[3])};
        lo = __builtin_bit_cast(shortx4, l);
    }
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int WG_T = 13;                        // tiles per panel side
constexpr int WG_LD = 208;                      // widest panel (columns)
constexpr int WG_LDP = 256;                     // LDS pitch of an operand row (floats): one 1 KiB DMA per wave
constexpr int WG_BK = 16;                       // rows per K-tile
// K-tiles of both operands resident in LDS: one computed on, the others in flight.  Two for the exact fp32 form (its 5500 MFMA
// cycles per tile cover one tile's load; a third stage measured 88.8 -> 93.9 us per layer), three for the bf16 forms, whose
// K loop is bound by the loads (bf16 60.6 -> 53.0 us, split 68.9 -> 67.5)
constexpr int wg_nst(int prec) { return prec == 0 ? 2 : 3; }
constexpr int WG_UNIT_FLOATS = WG_T * WG_T * 256;   // slab floats of one unit (tile-major)
constexpr int WG_MAX_PROBLEMS = 40;
constexpr int WG_MAX_SLICE = 1024;              // rows of one K-slice (their row indices are staged in LDS)
constexpr int WG_MAX_OUTPUTS = 8;

struct WgProb {
    const float* A;
    const float* B;
    const int* rows;
    unsigned a_bytes, b_bytes;        // extent of the operand views (buffer descriptor num_records)
    int lda, ldb;
    int M, N;
    int mcw, ncw;                     // panel widths (columns of A / B per chunk), multiples of 4, <= 208
    int m_chunks, n_chunks;
    int k_begin, k_end, k_per_slice;
    int unit_begin;
};

struct WgArgs {
    float* slab;
    unsigned long long* stamps;       // probe builds (-DWG_TIMING): 8 counters per workgroup
    int n_problems, n_units;
    WgProb p[WG_MAX_PROBLEMS];
};

// unit id of this workgroup such that the workgroups of one XCD (observed: block id % 8) hold CONSECUTIVE units: the
// panels of one K-slice read the same operand rows and then share that XCD's L2 (speed only, any mapping is correct)
__device__ __forceinline__ int xcd_unit(int total) {
    const int id = blockIdx.x;
    const int q = total / 8, r = total % 8, xcd = id % 8;
    return xcd * q + min(xcd, r) + id / 8;
}

#ifdef WG_TIMING
#define WG_STAMP(i) do { if (threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif

// MT: tile rows the code is unrolled for (>= the panel's: surplus tiles are computed on whatever the LDS rows hold and
// never stored - no per-MFMA branch); NFULL: the panel has all 13 tile columns (no branch at all in the K loop)
template <int MT, bool NFULL, int PREC>
__device__ __forceinline__ void wgrad_unit(const WgProb& p, const int slice, const int mc, const int nc, float* __restrict__ out,
                                           float* __restrict__ smem, int* __restrict__ kidx, unsigned long long* stamps) {
    WG_STAMP(0);
    constexpr int NQ = MT == WG_T ? 4 : (MT + 3) / 4;       // tiles of the 13th column per wave (MT 13: + tile row 12)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lk = lane >> 4;
    const int k0 = p.k_begin + slice * p.k_per_slice;
    const int klen = min(p.k_end - k0, p.k_per_slice);
    const int mvalid = min(p.mcw, p.M - mc * p.mcw), nvalid = min(p.ncw, p.N - nc * p.ncw);
    const int mtl = (mvalid + 15) / 16, ntl = NFULL ? WG_T : (nvalid + 15) / 16;

    for (int i = threadIdx.x; i < klen; i += 256) kidx[i] = p.rows != nullptr ? p.rows[k0 + i] : k0 + i;

    const unsigned a_rec = p.a_bytes - (unsigned)(mc * p.mcw) * 4u, b_rec = p.b_bytes - (unsigned)(nc * p.ncw) * 4u;
    const unsigned long long pa = (unsigned long long)(p.A + mc * p.mcw), pb = (unsigned long long)(p.B + nc * p.ncw);
    const intx4 da = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)a_rec, 0x00020000};
    const intx4 db = {(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)b_rec, 0x00020000};
    // a wave loads rows wave, wave + 4, ... of a K-tile: their parity is the wave's
    const int pw = wave & 1;
    const int s4 = lane ^ (pw << 2);                 // source float4 of this lane's LDS slot
    const bool a_ok = s4 * 4 < mvalid, b_ok = s4 * 4 < nvalid;
    const unsigned lda4 = (unsigned)p.lda * 4u, ldb4 = (unsigned)p.ldb * 4u, col = (unsigned)s4 * 16u;

    float* const As = smem;                          // [NST][BK][LDP]
    constexpr int NST = wg_nst(PREC);
    float* const Bs = smem + NST * WG_BK * WG_LDP;     // [NST][BK][LDP]
    const unsigned as_base = (unsigned)(unsigned long long)(lds_ptr_t)As, bs_base = (unsigned)(unsigned long long)(lds_ptr_t)Bs;
    __syncthreads();                                 // kidx visible

    int rnext[4];                                    // row indices of the wave's four rows of the tile being loaded
    auto read_rows = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rnext[i] = kidx[min(t * WG_BK + wave + 4 * i, klen - 1)];
    };
    auto issue_piece = [&](int t, int buf, int i) {  // rows wave + 4 i of both operands of K-tile t
        const int kr = wave + 4 * i;
        const int kl = t * WG_BK + kr;
        const bool live = kl < klen && rnext[i] >= 0;
        const unsigned oa = (live && a_ok) ? (unsigned)rnext[i] * lda4 + col : a_rec;
        const unsigned ob = (live && b_ok) ? (unsigned)rnext[i] * ldb4 + col : b_rec;
        const unsigned la = as_base + (unsigned)((buf * WG_BK + kr) * WG_LDP) * 4u;
        const unsigned lb = bs_base + (unsigned)((buf * WG_BK + kr) * WG_LDP) * 4u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(__builtin_amdgcn_readfirstlane(la)), "v"(oa), "s"(da), "s"(__builtin_amdgcn_readfirstlane(lb)), "v"(ob), "s"(db)
                     : "memory");
    };

    floatx4 acc[3][MT], accx[NQ];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) accx[q] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (klen + WG_BK - 1) / WG_BK;
#pragma unroll
    for (int t0 = 0; t0 < NST - 1; ++t0) {
        read_rows(t0);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(t0, t0, i);
    }
    // a tile is 8 loads per wave (4 pieces x 2 operands, issued whether its rows exist or not): tile 0 has landed when at most
    // the later tiles' loads are still in flight
    if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    WG_STAMP(1);
    const bool extra = ntl == WG_T;                  // the 13th tile column is shared by the waves (tile rows wave + 4 q)
    // fragment addresses: row 4 kk + lk (parity lk & 1); tile i sits at (i ^ parity) * 16 = i * 16 +- 16
    const int b = lk & 1;
    const int even_off = lk * WG_LDP + l15 + 16 * b, odd_off = lk * WG_LDP + l15 - 16 * b;
    const int own_off = pw ? odd_off : even_off;     // tiles wave + 4 j have the wave's parity
    for (int t = 0; t < nkt; ++t) {
        const int cur = t % NST, nxt = (t + NST - 1) % NST;      // nxt was last read in tile t - 1, behind that tile's barrier
        read_rows(t + NST - 1);                      // (clamped past the end; those rows load zeros)
        const float* as = As + cur * WG_BK * WG_LDP;
        const float* bs = Bs + cur * WG_BK * WG_LDP;
        if constexpr (PREC != PREC_F32) {
            // the SAME fragment reads as the fp32 form - row 4 kk + lk of the K-tile, kk = 0 .. 3 - but a lane's four values of a
            // fragment become ONE packed bf16x4 operand: v_mfma_f32_16x16x16_bf16 takes 4 consecutive k per lane group, and any
            // assignment of the tile's 16 rows to (lane group, element) is right as long as A and B use the same one
            constexpr bool X3 = PREC == PREC_BF16X3;
            auto frag = [&](const float* base, int off, shortx4& hi, shortx4& lo) {
                float v[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) v[kk] = base[kk * 4 * WG_LDP + off];
                split4(v, hi, lo, X3);
            };
            auto fma3 = [&](floatx4& c, const shortx4& bh, const shortx4& bl, const shortx4& ah, const shortx4& al) {
                if constexpr (X3) {      // small terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bl, ah, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh, al, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh, ah, c, 0, 0, 0);
            };
            shortx4 bh[3], bl[3], bxh, bxl;
#pragma unroll
            for (int j = 0; j < 3; ++j) frag(bs, own_off + (wave + 4 * j) * 16, bh[j], bl[j]);
            frag(bs, even_off + 12 * 16, bxh, bxl);
            // tile rows in two halves (the packed fragments of all 13 next to 172 accumulator registers do not fit 256)
            constexpr int H0 = (MT + 1) / 2;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i0 = half ? H0 : 0, cnt = half ? MT - H0 : H0;
                shortx4 ah[H0], al[H0];
#pragma unroll
                for (int i = 0; i < H0; ++i)
                    if (i < cnt) frag(as, (((i0 + i) & 1) ? odd_off : even_off) + (i0 + i) * 16, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (NFULL || wave + 4 * j < min(ntl, 12)) {
#pragma unroll
                        for (int i = 0; i < H0; ++i)
                            if (i < cnt) fma3(acc[j][i0 + i], bh[j], bl[j], ah[i], al[i]);
                    }
                    if (j < 2) {     // the next tile's loads in four pieces between the MFMA groups
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(t + NST - 1, nxt, half * 2 + j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (NFULL || extra) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    shortx4 xh, xl;
                    if (MT == WG_T && q == 3) frag(as, even_off + 12 * 16, xh, xl);
                    else frag(as, own_off + (wave + 4 * q) * 16, xh, xl);
                    fma3(accx[q], bxh, bxl, xh, xl);
                }
            }
        } else
#pragma unroll
        for (int kk = 0; kk < WG_BK / 4; ++kk) {
            const float* ae = as + kk * 4 * WG_LDP + even_off;
            const float* ao = as + kk * 4 * WG_LDP + odd_off;
            const float* aw = as + kk * 4 * WG_LDP + own_off;
            const float* bw = bs + kk * 4 * WG_LDP + own_off;
            const float* be = bs + kk * 4 * WG_LDP + even_off;
            float af[MT], bf[3], ax[NQ], bx;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = (i & 1) ? ao[i * 16] : ae[i * 16];
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = bw[(wave + 4 * j) * 16];
#pragma unroll
            for (int q = 0; q < NQ; ++q) ax[q] = (MT == WG_T && q == 3) ? ae[12 * 16] : aw[(wave + 4 * q) * 16];
            bx = be[12 * 16];
            // mfma(B fragment, A fragment): the accumulator holds C^T tiles - a lane owns 4 consecutive n of one m
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (NFULL || wave + 4 * j < min(ntl, 12)) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[j][i], 0, 0, 0);
                }
                if (j == 0) {
                    // one piece of the next tile's loads behind the first MFMA group of every k-step: buffer cur ^ 1 was
                    // last read before the barrier that ended the previous trip
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(t + NST - 1, nxt, kk);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (NFULL || extra) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) accx[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(bx, ax[q], accx[q], 0, 0, 0);
            }
        }
        if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile t + 1 has landed (tile t + 2's loads may still fly)
        __syncthreads();
    }
    if constexpr (NST > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (zero) over-fetch past the last tile

    // epilogue: tile (mt, nt) -> out[(mt * 13 + nt) * 256 + lane * 4 ..]: one coalesced 1 KiB store per tile
    WG_STAMP(2);
    float4* const o4 = reinterpret_cast<float4*>(out) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nt = wave + 4 * j;
        if (nt < min(ntl, 12)) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i < mtl) o4[(i * WG_T + nt) * 64] = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
        }
    }
    if (extra) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const bool row12 = MT == WG_T && q == 3;     // every wave computed it, wave 0 stores it
            const int mt = row12 ? 12 : wave + 4 * q;
            if ((!row12 || wave == 0) && mt < min(mtl, row12 ? 13 : 12))
                o4[(mt * WG_T + 12) * 64] = make_float4(accx[q][0], accx[q][1], accx[q][2], accx[q][3]);
        }
    }
}

// two workgroups per CU for the exact form (203 registers, 68 KB of LDS); the bf16 forms stage three K-tiles (100 KB of LDS): one
// workgroup per CU whatever the registers do - asking for two there only made the compiler report a missed target
template <int PREC>
__global__ void __launch_bounds__(256, PREC == 0 ? 2 : 1) wgrad_multi_kernel(const WgArgs a) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * wg_nst(PREC) * WG_BK * WG_LDP];
    __shared__ int kidx[WG_MAX_SLICE];
    const int u = xcd_unit(a.n_units);
    int pi = 0;
    while (pi + 1 < a.n_problems && a.p[pi + 1].unit_begin <= u) ++pi;
    const WgProb& p = a.p[pi];
    const int local = u - p.unit_begin;
    const int nc = local % p.n_chunks, mc = (local / p.n_chunks) % p.m_chunks, slice = local / (p.n_chunks * p.m_chunks);
    float* out = a.slab + (long)u * WG_UNIT_FLOATS;
    const int mvalid = min(p.mcw, p.M - mc * p.mcw), nvalid = min(p.ncw, p.N - nc * p.ncw);
    const int mtl = (mvalid + 15) / 16;
    const bool nfull = nvalid > 192;
#ifdef WG_TIMING
    unsigned long long* stamps = a.stamps + (long)blockIdx.x * 8;
    if (threadIdx.x == 0) { stamps[5] = u; stamps[7] = pi; }
#else
    unsigned long long* stamps = nullptr;
#endif
#define WG_DISPATCH(MTV)                                                          \
    do {                                                                          \
        if (nfull) wgrad_unit<MTV, true, PREC>(p, slice, mc, nc, out, smem, kidx, stamps);  \
        else wgrad_unit<MTV, false, PREC>(p, slice, mc, nc, out, smem, kidx, stamps);       \
    } while (0)
    if (mtl > 7) WG_DISPATCH(13);
    else if (mtl > 4) WG_DISPATCH(7);
    else if (mtl > 2) WG_DISPATCH(4);
    else WG_DISPATCH(2);
#undef WG_DISPATCH
#ifdef WG_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) stamps[3] = __builtin_readcyclecounter();
#endif
}

// ---- the reduction ------------------------------------------------------------------------------------------------
struct WgOut {
    int kind, n_groups, first_problem;
    int M, N, mcw, ncw, m_chunks, n_chunks;
    int ldc, c_split, n_scalers;
    long c_delta, scaler_stride;
    float* C;
    const float* aff;        // [3N] mean | scale | shift (kind BN)
    const float* row;        // [M]
    long item_begin;
};

struct WgReduceArgs {
    const float* slab;
    int n_outputs;
    long total_items;
    WgOut o[WG_MAX_OUTPUTS];
    int unit_begin[WG_MAX_PROBLEMS];
    int n_slices[WG_MAX_PROBLEMS];
    float coef[32][4];
};

template <int S>
__device__ __forceinline__ void wgrad_reduce_item(const WgReduceArgs& a, const WgOut& o, const long item) {
    const int lane = (int)(item & 63);
    long r = item >> 6;
    const int tile = (int)(r % (WG_T * WG_T));
    r /= WG_T * WG_T;
    const int nc = (int)(r % o.n_chunks), mc = (int)(r / o.n_chunks);
    const int mt = tile / WG_T, nt = tile % WG_T;
    const int ml = mt * 16 + (lane & 15), nl = nt * 16 + 4 * (lane >> 4);
    const int mvalid = min(o.mcw, o.M - mc * o.mcw), nvalid = min(o.ncw, o.N - nc * o.ncw);
    if (ml >= mvalid || nl >= nvalid) return;
    const int m = mc * o.mcw + ml, n = nc * o.ncw + nl;
    float4 tot[S];
#pragma unroll
    for (int s = 0; s < S; ++s) tot[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < o.n_groups; ++g) {
        const int pi = o.first_problem + g;
        const int zn = a.n_slices[pi];
        const long stride = (long)o.m_chunks * o.n_chunks * WG_UNIT_FLOATS;
        const float* p = a.slab + ((long)a.unit_begin[pi] + (long)mc * o.n_chunks + nc) * WG_UNIT_FLOATS + tile * 256 + lane * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int FL = 8;      // slices in flight per trip (round 3: 4 - a panel has ~12 slices: three dependent round trips), summed in slice order
        for (int z0 = 0; z0 < zn; z0 += FL) {
            float4 x[FL];
#pragma unroll
            for (int k = 0; k < FL; ++k) x[k] = *reinterpret_cast<const float4*>(p + (long)min(z0 + k, zn - 1) * stride);
#pragma unroll
            for (int k = 0; k < FL; ++k)
                if (z0 + k < zn) { v.x += x[k].x; v.y += x[k].y; v.z += x[k].z; v.w += x[k].w; }
        }
        if (o.kind == I3D_WGRAD_COMBINE) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float c = a.coef[g][s];
                tot[s].x += c * v.x; tot[s].y += c * v.y; tot[s].z += c * v.z; tot[s].w += c * v.w;
            }
        } else {
            tot[0] = v;
        }
    }
    float* c = o.C + (long)m * o.ldc + n + (m >= o.c_split ? o.c_delta : 0);
    if (o.kind == I3D_WGRAD_BN) {
        const float rw = o.row[m];
        const float4 mu = *reinterpret_cast<const float4*>(o.aff + n);
        const float4 sc = *reinterpret_cast<const float4*>(o.aff + o.N + n);
        const float4 sh = *reinterpret_cast<const float4*>(o.aff + 2 * o.N + n);
        tot[0].x = (tot[0].x - rw * mu.x) * sc.x + rw * sh.x;
        tot[0].y = (tot[0].y - rw * mu.y) * sc.y + rw * sh.y;
        tot[0].z = (tot[0].z - rw * mu.z) * sc.z + rw * sh.z;
        tot[0].w = (tot[0].w - rw * mu.w) * sc.w + rw * sh.w;
    }
    if (o.kind == I3D_WGRAD_COMBINE) {
#pragma unroll
        for (int s = 0; s < S; ++s) *reinterpret_cast<float4*>(c + (long)s * o.scaler_stride) = tot[s];
    } else {
        *reinterpret_cast<float4*>(c) = tot[0];
    }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgReduceArgs a) {
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= a.total_items) return;
    int oi = 0;
    while (oi + 1 < a.n_outputs && a.o[oi + 1].item_begin <= item) ++oi;
    const WgOut& o = a.o[oi];
    const long local = item - o.item_begin;
    if (o.kind == I3D_WGRAD_COMBINE) {
        switch (o.n_scalers) {
            case 1: wgrad_reduce_item<1>(a, o, local); break;
            case 2: wgrad_reduce_item<2>(a, o, local); break;
            case 3: wgrad_reduce_item<3>(a, o, local); break;
            default: wgrad_reduce_item<4>(a, o, local); break;
        }
    } else {
        wgrad_reduce_item<1>(a, o, local);
    }
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e != nullptr && e[0] != 0) ? atoi(e) : dflt;
}

int chunk_width(int n, int& chunks) {
    chunks = cdiv(n, WG_LD);
    return cdiv(cdiv(n, chunks), 4) * 4;
}

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" int i3d_wgrad_multi_supported(const I3dWgradProblem* problems, int n_problems, const I3dWgradOutput* outputs,
                                         int n_outputs) {
    if (problems == nullptr || outputs == nullptr || n_problems < 1 || n_problems > WG_MAX_PROBLEMS || n_outputs < 1 ||
        n_outputs > WG_MAX_OUTPUTS)
        return 0;
    int combines = 0;
    for (int i = 0; i < n_outputs; ++i) {
        const I3dWgradOutput& o = outputs[i];
        if (o.kind == I3D_WGRAD_COMBINE) {
            ++combines;
            if (o.n_groups < 1 || o.n_groups > 32 || o.n_scalers < 1 || o.n_scalers > 4 || o.coef == nullptr || o.scaler_stride % 4 != 0)
                return 0;
        } else if (o.n_groups != 1) {
            return 0;
        }
        if (o.C == nullptr || o.ldc % 4 != 0 || o.c_delta % 4 != 0 || (((uintptr_t)o.C) & 15) != 0) return 0;
        if (o.kind == I3D_WGRAD_BN && (o.aff == nullptr || o.row == nullptr || (((uintptr_t)o.aff) & 15) != 0)) return 0;
        if (o.first_problem < 0 || o.first_problem + o.n_groups > n_problems) return 0;
    }
    if (combines > 1) return 0;
    for (int i = 0; i < n_problems; ++i) {
        const I3dWgradProblem& p = problems[i];
        if (p.A == nullptr || p.B == nullptr || p.M < 4 || p.N < 4 || p.M % 4 != 0 || p.N % 4 != 0 || p.lda % 4 != 0 ||
            p.ldb % 4 != 0 || p.lda < p.M || p.ldb < p.N || ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) != 0 || p.k_count < 0 ||
            p.rows_total < 1)
            return 0;
        if (((p.rows_total - 1) * p.lda + p.M) * 4 >= (1L << 32) - 16 || ((p.rows_total - 1) * p.ldb + p.N) * 4 >= (1L << 32) - 16)
            return 0;
    }
    return 1;
}

// rows per K-slice of a problem when a full 13 x 13 panel gets `kps`: a panel with fewer tiles costs less per row (down to
// ~0.45: then its loads bound it - measured on the [64 x 200] bond-table product), so it gets longer slices - every unit of a launch then takes about the same time
static int problem_kps(const I3dWgradProblem& p, int kps) {
    int mch, nch;
    const int mt = std::min(WG_T, cdiv(chunk_width(p.M, mch), 16)), nt = std::min(WG_T, cdiv(chunk_width(p.N, nch), 16));
    const int mt_code = mt > 7 ? 13 : (mt > 4 ? 7 : (mt > 2 ? 4 : 2));            // the unroll class of the kernel
    const int tiles = mt_code * std::min(nt, 12) + (nt == WG_T ? mt_code : 0);
    const long scaled = (long)kps * WG_T * WG_T / std::max(tiles, WG_T * WG_T * 9 / 20);
    return (int)std::min<long>(WG_MAX_SLICE, scaled / WG_BK * WG_BK);
}

// units of a plan with `kps` rows per slice of a full panel
static long plan_units(const I3dWgradProblem* problems, int n, int kps) {
    long u = 0;
    for (int i = 0; i < n; ++i) {
        int mch, nch;
        chunk_width(problems[i].M, mch);
        chunk_width(problems[i].N, nch);
        u += (long)mch * nch * cdiv(problems[i].k_count, problem_kps(problems[i], kps));
    }
    return u;
}

extern "C" long i3d_wgrad_multi_workspace_bytes(int max_units) { return (long)max_units * WG_UNIT_FLOATS * 4; }

extern "C" long i3d_wgrad_multi_min_workspace_bytes(const I3dWgradProblem* problems, int n_problems) {
    if (problems == nullptr || n_problems < 1) return -1;
    return (plan_units(problems, n_problems, WG_MAX_SLICE) + 1) * (long)WG_UNIT_FLOATS * 4 + (1 << 16);
}

extern "C" int i3d_wgrad_multi(const I3dWgradProblem* problems, int n_problems, const I3dWgradOutput* outputs, int n_outputs,
                               void* workspace, long workspace_bytes, void* stream) {
    I3D_CHECK_ARG(i3d_wgrad_multi_supported(problems, n_problems, outputs, n_outputs), "unsupported problem set");
    I3D_CHECK_ARG(workspace != nullptr && (((uintptr_t)workspace) & 15) == 0, "scratch required");
#ifdef WG_TIMING
    const long max_units = (workspace_bytes - (1 << 16)) / ((long)WG_UNIT_FLOATS * 4);
#else
    const long max_units = workspace_bytes / ((long)WG_UNIT_FLOATS * 4);
#endif
    // rows per K-slice of a full panel: the smallest multiple of 16 with at most I3D_WGRAD_UNITS (default 256 = one unit
    // per CU: one round of workgroups) units that fit the scratch - measured at batch 512 (tools/ab.sh, step time): 256 rows
    // (~370 units) 2.39 ms, 384 rows (~240) 2.25-2.29, 512 rows (~180) 2.37, this rule (~360 rows, ~250 units) 2.24.
    // I3D_WGRAD_ROWS fixes the length instead (doubled only while the slabs do not fit).
    static const int fixed_rows = env_int("I3D_WGRAD_ROWS", 0) / WG_BK * WG_BK;
    static const int target_units = env_int("I3D_WGRAD_UNITS", 256);
    int kps;
    if (fixed_rows > 0) {
        kps = std::min(WG_MAX_SLICE, fixed_rows);
        while (kps < WG_MAX_SLICE && plan_units(problems, n_problems, kps) > max_units) kps = std::min(WG_MAX_SLICE, kps * 2);
    } else {
        const long target = std::min<long>(max_units, target_units);
        int lo = 128 / WG_BK, hi = WG_MAX_SLICE / WG_BK;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (plan_units(problems, n_problems, mid * WG_BK) <= target) hi = mid;
            else lo = mid + 1;
        }
        kps = lo * WG_BK;
    }
    I3D_CHECK_ARG(plan_units(problems, n_problems, kps) <= max_units, "scratch too small");
    WgArgs a;
    WgReduceArgs r;
    a.slab = (float*)workspace;
    a.stamps = nullptr;
#ifdef WG_TIMING
    a.stamps = (unsigned long long*)((char*)workspace + workspace_bytes - (1 << 16));     // (probe build: 1024 workgroups)
#endif
    a.n_problems = n_problems;
    int units = 0;
    for (int i = 0; i < n_problems; ++i) {
        const I3dWgradProblem& s = problems[i];
        WgProb& p = a.p[i];
        p.A = s.A; p.B = s.B; p.rows = s.rows;
        p.a_bytes = (unsigned)(((s.rows_total - 1) * s.lda + s.M) * 4);
        p.b_bytes = (unsigned)(((s.rows_total - 1) * s.ldb + s.N) * 4);
        p.lda = s.lda; p.ldb = s.ldb; p.M = s.M; p.N = s.N;
        p.mcw = chunk_width(s.M, p.m_chunks);
        p.ncw = chunk_width(s.N, p.n_chunks);
        p.k_begin = s.k_begin; p.k_end = s.k_begin + s.k_count;
        // equal slices inside a problem (no short tail slice)
        const int n_slices = cdiv(s.k_count, problem_kps(s, kps));
        p.k_per_slice = n_slices > 0 ? cdiv(cdiv(s.k_count, n_slices), WG_BK) * WG_BK : WG_BK;
        p.unit_begin = units;
        r.unit_begin[i] = units;
        r.n_slices[i] = s.k_count > 0 ? cdiv(s.k_count, p.k_per_slice) : 0;
        units += p.m_chunks * p.n_chunks * r.n_slices[i];
    }
    I3D_CHECK_ARG(units <= max_units, "scratch too small");
    a.n_units = units;
    hipStream_t st = (hipStream_t)stream;
    if (units > 0) {
        // how the products are formed: the process-wide matmul precision (bf16 mode: bf16 operands), or - fp32 mode, opt-in -
        // split bf16 operands (see PREC above)
        const char* const sb = getenv("I3D_WGRAD_SPLIT_BF16");      // (read per call: a test switches it inside one process)
        const bool split_bf16 = sb != nullptr && sb[0] == '1';
        const int prec = i3d_get_matmul_precision() != 0 ? PREC_BF16 : (split_bf16 ? PREC_BF16X3 : PREC_F32);
        if (prec == PREC_BF16) hipLaunchKernelGGL(wgrad_multi_kernel<PREC_BF16>, dim3(units), dim3(256), 0, st, a);
        else if (prec == PREC_BF16X3) hipLaunchKernelGGL(wgrad_multi_kernel<PREC_BF16X3>, dim3(units), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(wgrad_multi_kernel<PREC_F32>, dim3(units), dim3(256), 0, st, a);
        I3D_CHECK_LAUNCH();
    }
    r.slab = a.slab;
    r.n_outputs = n_outputs;
    long items = 0;
    std::memset(r.coef, 0, sizeof(r.coef));
    for (int i = 0; i < n_outputs; ++i) {
        const I3dWgradOutput& s = outputs[i];
        const WgProb& p = a.p[s.first_problem];
        WgOut& o = r.o[i];
        o.kind = s.kind; o.n_groups = s.n_groups; o.first_problem = s.first_problem;
        o.M = p.M; o.N = p.N; o.mcw = p.mcw; o.ncw = p.ncw; o.m_chunks = p.m_chunks; o.n_chunks = p.n_chunks;
        for (int g = 1; g < s.n_groups; ++g)
            I3D_CHECK_ARG(a.p[s.first_problem + g].M == p.M && a.p[s.first_problem + g].N == p.N, "groups of one output differ in shape");
        o.ldc = s.ldc; o.c_split = s.c_split > 0 ? s.c_split : 0x7fffffff; o.c_delta = s.c_delta;
        o.n_scalers = s.n_scalers; o.scaler_stride = s.scaler_stride;
        o.C = s.C; o.aff = s.aff; o.row = s.row;
        o.item_begin = items;
        items += (long)p.m_chunks * p.n_chunks * WG_T * WG_T * 64;
        if (s.kind == I3D_WGRAD_COMBINE)
            for (int g = 0; g < s.n_groups; ++g)
                for (int k = 0; k < s.n_scalers; ++k) r.coef[g][k] = s.coef[g * s.n_scalers + k];
    }
    r.total_items = items;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(items, 256)), dim3(256), 0, st, r);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
