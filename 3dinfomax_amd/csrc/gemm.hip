// Dense tower GEMMs on the fp32 matrix cores of gfx950.
//
// Replaces aten::addmm / aten::mm behind nn.Linear in FCLayer (reference models/base_layers.py:101)
// for the forward (Y = X W^T + b), the data gradient (dX = dY W) and the weight gradient
// (dW = dY^T X).  Exact fp32: v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 are bit-for-bit an fp32 fmaf
// chain (MI355X guide 3), so parity with the reference's CPU fp32 matmul is a summation-order question only.
//
// Structure (wave64, 4 waves / workgroup):
//  * both operand tiles are staged in LDS k-major  T[k][idx]  (idx = m or n), so the MFMA fragment of a lane
//    is one conflict-free ds_read_b32: for the 16-wide MFMA the two 16-lane halves of a 32-lane group are
//    steered to different bank halves by XOR-ing bit 4 of idx with ((k ^ (k>>2)) & 1); a 32-wide fragment read is
//    a permutation of 32 consecutive banks either way.
//  * register-staged pipeline with a ring of PF K-tiles in flight: the loads of tile t+PF are issued before the
//    MFMAs of tile t, tile t+1 is written to the other LDS buffer afterwards: one barrier per K-tile.  Every load
//    of the loop is issued unconditionally (tiles past the end read out of bounds and return 0): a load inside a
//    branch makes the compiler's vmcnt bookkeeping fall back to vmcnt(0) at the join, which drains the ring.
//  * loads go through buffer descriptors: lanes outside the matrix pass an offset beyond num_records and the
//    hardware returns 0 - no per-lane branch around a load (MI355X guide 5 trap (c)); the operand layouts are
//    template parameters, per-slot address parts are hoisted out of the K loop.
//  * the MFMA is issued with (W-fragment, X-fragment) so the accumulator holds C^T tiles: a lane owns 4
//    consecutive output columns of one row -> the epilogue (bias, accumulate, split-K atomics) is 16-byte accesses.
//  * k-contiguous operands (X[m][k], W[n][k]) are loaded with 16-byte loads along k and transposed on the
//    LDS write; idx-contiguous operands (dY^T, W for dX) are loaded along idx and written as b128.
//  * split-K (grid.z) with fp32 atomics ONLY for the row-reduction GEMMs of the backward pass.
//  * row indirection (m_rows / k_rows) and per-m-tile weight selection (tile_group) for the degree-grouped
//    posttrans GEMMs of the PNA layer (i3d_gemm_f32_grouped): rows of one in-degree share the combined weight
//    W_D = W_id + amp(D) W_amp + att(D) W_att, which cuts K from 12F to 4F.
#include "common.h"
#include <algorithm>

namespace i3d {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int accumulate;  // C += ...
    int k_per_split; // multiple of BK
    int atomic_out;  // split-K: atomicAdd into C
    int c_vec;       // 16-byte access to C allowed
    unsigned a_bytes, b_bytes;  // extent of the operand views in bytes (buffer descriptor num_records)
    const int* m_rows;      // [M] or null: logical row m lives at row m_rows[m] of A (k-contiguous A only) and of C;
                            //          -1 = padding row (loads return 0, nothing is stored)
    const int* k_rows;      // row-segment kernel: reduction index k lives at row k_rows[k] of both operands
    const int* tile_group;  // [ceil(M/BM)] or null: B of m-tile t is g.B + tile_group[t] * b_group_stride
    long b_group_stride;    // floats
    // two-block operands (the [W_s | W_d] halves of an edge-MLP weight addressed as ONE matrix, P trick of edge.hip):
    // B index (n when B is k-contiguous, k otherwise) >= b_split adds b_delta floats to the address; C rows >= c_split
    // add c_delta.  split = INT_MAX: plain operand.
    int b_split, c_split;
    long b_delta, c_delta;
    float* slab;            // split-K / row segments: blockIdx.z slice z stores its partial tile to slab[z][M][N]
                            // (plain 16-byte stores), slab_reduce_kernel sums the slices in a fixed order: deterministic
                            // and cheaper than fp32 atomics on top of a zero-fill.  null: atomics.
    // fused BatchNorm (FUSE variants of the forward layout only, see gemm_body):
    const float* a_aff;     // [3K] mean | scale | shift: the A operand is read as (A[m][k] - mean[k]) * scale[k] + shift[k] -
                            // the BatchNorm-apply of the block in front, folded into the LDS staging of the consumer
    float* stats;           // [m tiles][3][N]: per 64-row tile and column {sum, M2 about the tile mean, row count} of the
                            // values this launch stores (after bias / accumulate / epi_act): the BatchNorm statistics of
                            // THIS block without a pass over its output (bn_finalize_partials_kernel combines the tiles)
    int epi_act;            // activation applied to the stored value (I3D_ACT_*)
    const float* Cin;       // FUSE & 2 with accumulate: the addend is read from Cin (row pitch ldcin) instead of C - the addend may
    int ldcin;              // be a column block of a wider matrix (the merged [P | lin_h] product of a PNA layer); null: C itself
    int c_bf16;             // FUSE & 2: C is stored as bf16 (row r at (bf16*)C + r * ldc; rounded RNE where it is stored, the
                            // statistics are taken from the fp32 values): the bf16 mode's storage form of the messages
    // batched products (gemm_f32_kernel only; i3d_gemm_f32_batched): blockIdx.z = batch * z_splits + K-slice, batch b works on
    // A + b a_batch, B + b b_batch, C + b c_batch (floats) - the diagonal blocks of a block-diagonal product in ONE launch
    int n_batch, z_splits;  // n_batch <= 1: plain product
    long a_batch, b_batch, c_batch;
};

__device__ __forceinline__ int swz(int k, int idx) { return idx ^ ((((k) ^ (k >> 2)) & 1) << 4); }

// KC: the operand is k-contiguous (T[idx*ld + k]); otherwise idx-contiguous (T[k*ld + idx])
// IM (k-contiguous operands only): the LDS image is idx-major, T[idx][BK + 2]: the 16-byte global load of 4 consecutive
// k is stored with two 8-byte writes (instead of four transposing 4-byte writes) and a lane's MFMA operands for two
// consecutive k-steps come from one 8-byte read; the row pitch BK + 2 keeps both conflict-free.
// IH (k-contiguous operands, bf16 matmul mode): the LDS image is idx-major in BF16, T16[idx][BK + 8] - the operand is rounded
// ONCE, when the tile is staged (the 16-byte global load of 4 consecutive k becomes one 8-byte write), and a lane's MFMA
// fragment (8 consecutive k of one row) is ONE ds_read_b128: half the LDS bytes of the fp32 image, an eighth of its
// fragment reads (round 2 kept the fp32 image and converted at every fragment read).  Row pitch BK + 8 halves = 12 / 20 /
// 36 dwords: the 16 lanes of a b128 access group land on disjoint banks.
// IH = 3 (split products, i3d_set_fp32_products(1)): THREE such images hi | mid | lo with x = hi + mid + lo exactly (each part
// the bf16 rounding of what the parts in front of it leave), one behind the other at IMG_HALVES.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// (a, b) -> packed bf16 pairs hi | mid | lo with a = hi.a + mid.a + lo.a exactly (same for b): every part is the RNE bf16 of what
// the parts in front of it leave (v_cvt_pk_bf16_f32 on the pair), the remainders are exact fp32 differences (v_pk_add_f32)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    auto pack = [](f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); };
    auto widen = [](unsigned u) { f32x2_t r; r.x = __uint_as_float(u << 16); r.y = __uint_as_float(u & 0xffff0000u); return r; };
    f32x2_t v;
    v.x = a; v.y = b;
    hi = pack(v);
    const f32x2_t r1 = v - widen(hi);
    mid = pack(r1);
    const f32x2_t r2 = r1 - widen(mid);
    lo = pack(r2);
}
template <int R, int LD, int BK, bool KC, bool IM = false, int NT = 256, int IH = 0>       // NT: threads of the workgroup
struct TileStage {
    static constexpr int LDK = BK + 2;
    static constexpr int LDKH = BK + 8;                      // bf16 image: row pitch in halves
    static constexpr int IMG_HALVES = R * LDKH;              // one bf16 image
    static constexpr int LDS_FLOATS = IH ? IH * R * LDKH / 2 : (IM ? R * LDK : BK * LD);
    static constexpr int SLOTS = R * BK / 4;                 // float4 slots in a tile
    static constexpr int KQ = BK / 4;                        // float4 slots along k of one row
    static constexpr int PER_THREAD = (SLOTS + NT - 1) / NT;
    struct Regs { float4 v[PER_THREAD]; };                   // one K-tile in flight (the kernel keeps a ring of PF of them)
    unsigned base[PER_THREAD];   // loop-invariant byte offset of the slot (row part or idx part)
    int kloc[PER_THREAD];        // k of the slot inside a K-tile
    int iloc[PER_THREAD];        // idx of the slot (idx-contiguous: first of 4)
    int rowi[PER_THREAD];        // k-contiguous operands: physical row of the slot (through `rows` when given; 0 for a padding slot)
    bool ok[PER_THREAD];

    // KC : slot -> (idx = s / KQ, kq = s % KQ), 4 consecutive k of one row
    // else: slot -> (k = s / (R/4), iq = s % (R/4)), 4 consecutive idx of one k
    __device__ __forceinline__ void prepare(int ld, int idx0, int idx_max, const int* __restrict__ rows, int split = 0x7fffffff,
                                            long delta = 0) {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int s = threadIdx.x + it * NT;
            if (KC) {
                const int idx = idx0 + s / KQ;
                kloc[it] = (s % KQ) * 4;
                int row = -1;
                if (idx < idx_max && s < SLOTS) row = rows ? rows[idx] : idx;
                ok[it] = row >= 0;
                iloc[it] = idx;
                rowi[it] = max(row, 0);
                base[it] = (unsigned)((long)max(row, 0) * ld + kloc[it] + (idx >= split ? delta : 0)) * 4u;
            } else {
                kloc[it] = s / (R / 4);
                iloc[it] = idx0 + (s % (R / 4)) * 4;
                ok[it] = iloc[it] < idx_max && s < SLOTS;
                rowi[it] = 0;
                base[it] = (unsigned)iloc[it] * 4u;
            }
        }
    }

    // ROWS: kidx is an LDS copy of k_rows[k_begin .. k_end) (LDS reads count on lgkmcnt, so they do not drain the
    // in-order vmcnt queue of the tiles already in flight)
    template <bool VEC, bool ROWS>
    __device__ __forceinline__ void load(Regs& r, __amdgpu_buffer_rsrc_t rsrc, unsigned oob, int ld, int idx_max, int k0,
                                         int k_begin, int k_end, const int* kidx, int split = 0x7fffffff,
                                         long delta = 0) const {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int k = k0 + kloc[it];
            unsigned off;
            const bool valid = ok[it] && k < k_end;
            if (KC) {
                off = base[it] + (unsigned)k0 * 4u;
            } else {
                int krow = k;
                if (ROWS) krow = kidx[max(min(k, k_end - 1) - k_begin, 0)];
                off = (unsigned)((long)krow * ld + (k >= split ? delta : 0)) * 4u + base[it];
            }
            if (VEC) {   // contiguous extent is a multiple of 4: a valid first element implies a valid float4
                auto q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, valid ? off : oob, 0, 0);
                static_assert(sizeof(q) == 16, "b128 load");
                r.v[it] = __builtin_bit_cast(float4, q);
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool oku = valid && (KC ? (k + u < k_end) : (iloc[it] + u < idx_max));
                    e[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oku ? off + 4u * u : oob, 0, 0));
                }
                r.v[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }

    // store for the fused BatchNorm-apply prologue (IM image of a k-contiguous operand): aff = LDS copy of [3][KP] mean |
    // scale | shift (zeros beyond K, so out-of-range k stay finite and meet the zeros of the other operand)
    __device__ __forceinline__ static void put_bf16x4(float* __restrict__ T, int idx, int k, float a, float b, float c, float d) {
        __bf16* const p = reinterpret_cast<__bf16*>(T) + idx * LDKH + k;
        if constexpr (IH == 3) {
            uint2 h, m, l;
            split_pair(a, b, h.x, m.x, l.x);
            split_pair(c, d, h.y, m.y, l.y);
            *reinterpret_cast<uint2*>(p) = h;
            *reinterpret_cast<uint2*>(p + IMG_HALVES) = m;
            *reinterpret_cast<uint2*>(p + 2 * IMG_HALVES) = l;
        } else {
            bf16x4 h;
            h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
            *reinterpret_cast<bf16x4*>(p) = h;
        }
    }

    __device__ __forceinline__ void store_aff(const Regs& r, float* __restrict__ T, const float* aff, int KP, int k0) const {
        static_assert(IM || IH, "affine prologue: idx-major image only");
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * NT;
            if (s < SLOTS) {
                const int idx = s / KQ, k = (s % KQ) * 4;
                const int kg = min(k0 + k, KP - 4);
                const float4 mu = *reinterpret_cast<const float4*>(aff + kg);
                const float4 sc = *reinterpret_cast<const float4*>(aff + KP + kg);
                const float4 sh = *reinterpret_cast<const float4*>(aff + 2 * KP + kg);
                const float4 v = r.v[it];
                if constexpr (IH != 0) {
                    put_bf16x4(T, idx, k, (v.x - mu.x) * sc.x + sh.x, (v.y - mu.y) * sc.y + sh.y, (v.z - mu.z) * sc.z + sh.z,
                               (v.w - mu.w) * sc.w + sh.w);
                    continue;
                }
                *reinterpret_cast<float2*>(&T[idx * LDK + k]) = make_float2((v.x - mu.x) * sc.x + sh.x, (v.y - mu.y) * sc.y + sh.y);
                *reinterpret_cast<float2*>(&T[idx * LDK + k + 2]) = make_float2((v.z - mu.z) * sc.z + sh.z, (v.w - mu.w) * sc.w + sh.w);
            }
        }
    }

    __device__ __forceinline__ void store(const Regs& r, float* __restrict__ T) const {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * NT;
            if (s < SLOTS) {
                if (IH) {
                    put_bf16x4(T, s / KQ, (s % KQ) * 4, r.v[it].x, r.v[it].y, r.v[it].z, r.v[it].w);
                } else if (IM) {
                    int idx = s / KQ, k = (s % KQ) * 4;
                    *reinterpret_cast<float2*>(&T[idx * LDK + k]) = make_float2(r.v[it].x, r.v[it].y);
                    *reinterpret_cast<float2*>(&T[idx * LDK + k + 2]) = make_float2(r.v[it].z, r.v[it].w);
                } else if (KC) {
                    int idx = s / KQ, k = (s % KQ) * 4;
                    T[(k + 0) * LD + swz(k + 0, idx)] = r.v[it].x;
                    T[(k + 1) * LD + swz(k + 1, idx)] = r.v[it].y;
                    T[(k + 2) * LD + swz(k + 2, idx)] = r.v[it].z;
                    T[(k + 3) * LD + swz(k + 3, idx)] = r.v[it].w;
                } else {
                    int k = s / (R / 4), idx = (s % (R / 4)) * 4;
                    *reinterpret_cast<float4*>(&T[k * LD + swz(k, idx)]) = r.v[it];
                }
            }
        }
    }
};

template <int MT> struct Acc;
template <> struct Acc<16> { typedef floatx4 type; static constexpr int REGS = 4; };
template <> struct Acc<32> { typedef floatx16 type; static constexpr int REGS = 16; };

// Row-subset / segmented reductions (weight gradients of the degree groups): blockIdx.z walks a table of
// (k range, output offset) segments, every segment accumulates with atomics into its group's output.
constexpr int MAX_SEGS = 96;
constexpr int SEG_MAX_K = 2048;          // k_rows of one segment are staged in LDS
struct Seg { int k_begin, k_end; long c_off; };
struct SegTable { Seg s[MAX_SEGS]; };

// Tile shape: MT = MFMA tile (16: v_mfma_f32_16x16x4_f32, 32: v_mfma_f32_32x32x2_f32), a wave computes WM_T x WN_T
// such tiles; PF = K-tiles in flight per workgroup.
template <int MT_, int WAVES_M_, int WAVES_N_, int WM_T_, int WN_T_, int BK_, int PF_, bool KP_ = false>
struct Shape {
    static constexpr int MT = MT_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, WM_T = WM_T_, WN_T = WN_T_, BK = BK_, PF = PF_;
    static constexpr bool KP = KP_;     // idx-major LDS image for k-contiguous operands (TileStage IM), 32x32x2 MFMA only
    static constexpr int BM = WAVES_M * WM_T * MT, BN = WAVES_N * WN_T * MT;
    static constexpr int NT = 64 * WAVES_M * WAVES_N;      // threads of the workgroup
};

// blockIdx -> (m tile, n tile, z) such that the workgroups that run on one XCD (observed: linear block id % 8; a speed
// assumption only) are CONSECUTIVE work items in (z, m tile, n tile) order: the n-tiles of one m-tile - which read
// the same rows of A - and the tiles of one K split - which read the same rows of both operands - then share that
// XCD's L2 instead of each pulling their own copy through the fabric (the L2s of the 8 XCDs are separate).
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz) {
    const int nx = gridDim.x, ny = gridDim.y;
    const int total = nx * ny * gridDim.z;
    const int id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = total / 8, r = total % 8, xcd = id % 8;
    const int w = xcd * q + min(xcd, r) + id / 8;      // XCD x owns q (+1 if x < r) consecutive items
    by = w % ny;
    bx = (w / ny) % nx;
    bz = w / (ny * nx);
}

// FUSE (forward layout, idx-major image, 64-row tiles only): bit 0 = BatchNorm-apply prologue on A (g.a_aff), bit 1 =
// activation + per-tile column statistics of the stored values (g.epi_act, g.stats).  K <= FUSE_MAX_K for bit 0.
constexpr int FUSE_MAX_K = 1024;

// BF16 (i3d_set_matmul_precision(1)): the operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) when a
// lane reads its MFMA fragments from the fp32 LDS image and multiplied on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16 /
// v_mfma_f32_16x16x32_bf16: one instruction per K-tile instead of 8) - fp32 accumulation, fp32 bias / epilogue /
// statistics, fp32 tensors in HBM.  Shapes whose K-tile is one such instruction (MT 32 + BK 16, MT 16 + BK 32).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <class S> constexpr bool bf16_shape() { return (S::MT == 32 && S::BK % 16 == 0) || (S::MT == 16 && S::BK % 32 == 0); }

__device__ __forceinline__ bf16x8 to_bf16x8(const float* f) {
    bf16x8 r;
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = (__bf16)f[q];
    return r;
}

// f = hi + mid + lo exactly: hi = bf16(f), mid = bf16(f - hi), lo = f - hi - mid (<= 8 significant bits: a bf16); operands
// whose LDS image is fp32 (idx-contiguous ones) are split when a lane has read its fragment
__device__ __forceinline__ void split_bf16x8(const float* f, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    uint4 h, m, l;
    split_pair(f[0], f[1], h.x, m.x, l.x);
    split_pair(f[2], f[3], h.y, m.y, l.y);
    split_pair(f[4], f[5], h.z, m.z, l.z);
    split_pair(f[6], f[7], h.w, m.w, l.w);
    hi = __builtin_bit_cast(bf16x8, h);
    mid = __builtin_bit_cast(bf16x8, m);
    lo = __builtin_bit_cast(bf16x8, l);
}

template <class S, bool VEC, bool A_KC, bool B_KC, bool ROWS, int FUSE = 0, int BF16 = 0>       // BF16: 0 fp32 MFMA, 1 bf16-rounded operands, 2 split products
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int k_begin, const int k_end,
                                          float* __restrict__ Cout, const int* kidx, const bool first_split) {
    constexpr int MT = S::MT, WAVES_N = S::WAVES_N, WM_T = S::WM_T, WN_T = S::WN_T, BK = S::BK, PF = S::PF;
    constexpr int BM = S::BM, BN = S::BN;
    constexpr int LDA = (BM + 31) / 32 * 32, LDB = (BN + 31) / 32 * 32;
    constexpr int KSTEP = (MT == 16) ? 4 : 2;         // k per MFMA
    constexpr bool SPLIT = BF16 == 2;
    constexpr bool A_IH = BF16 != 0 && VEC && S::KP && A_KC && MT == 32, B_IH = BF16 != 0 && VEC && S::KP && B_KC && MT == 32;   // bf16 images
    constexpr bool A_IM = S::KP && A_KC && MT == 32 && !A_IH, B_IM = S::KP && B_KC && MT == 32 && !B_IH;
    typedef TileStage<BM, LDA, BK, A_KC, A_IM, S::NT, A_IH ? (SPLIT ? 3 : 1) : 0> StageA;
    typedef TileStage<BN, LDB, BK, B_KC, B_IM, S::NT, B_IH ? (SPLIT ? 3 : 1) : 0> StageB;
    // one block of LDS: the operand double buffers, reused by the statistics epilogue as the [BM][BN + 1] output tile
    constexpr int OPER_FLOATS = 2 * StageA::LDS_FLOATS + 2 * StageB::LDS_FLOATS;
    constexpr int CT_PITCH = BN + 1;
    constexpr int EPI_FLOATS = (FUSE & 2) ? BM * CT_PITCH + 2 * 4 * BN + BM : 0;
    __shared__ __attribute__((aligned(16))) float smem[OPER_FLOATS > EPI_FLOATS ? OPER_FLOATS : EPI_FLOATS];
    __shared__ __attribute__((aligned(16))) float affL[(FUSE & 1) ? 3 * FUSE_MAX_K : 4];
    float* const As = smem;
    float* const Bs = smem + 2 * StageA::LDS_FLOATS;
    static_assert(FUSE == 0 || ((A_IM || A_IH) && BM == 64), "fused variants: idx-major A image, 64-row tiles");

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = bx * BM, n0 = by * BN;

    typename Acc<MT>::type acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < Acc<MT>::REGS; ++r) acc[i][j][r] = 0.f;

    StageA sa;
    StageB sb;
    typename StageA::Regs ra_[PF];
    typename StageB::Regs rb_[PF];
    // descriptors are built from kernel arguments / blockIdx only (wave-uniform: no waterfall loops, guide T20)
    const float* Bp = g.B;
    if (g.tile_group != nullptr) Bp += (long)g.tile_group[bx] * g.b_group_stride;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, g.b_bytes, 0x00020000);
    sa.prepare(g.lda, m0, g.M, g.m_rows);
    sb.prepare(g.ldb, n0, g.N, nullptr, g.b_split, g.b_delta);
    const int nk = (k_end - k_begin + BK - 1) / BK;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        sa.template load<VEC, ROWS>(ra_[u], ra, g.a_bytes, g.lda, g.M, k_begin + u * BK, k_begin, k_end, kidx);
        sb.template load<VEC, ROWS>(rb_[u], rb, g.b_bytes, g.ldb, g.N, k_begin + u * BK, k_begin, k_end, kidx, g.b_split, g.b_delta);
    }
    const int KP = (g.K + 3) & ~3;
    if constexpr ((FUSE & 1) != 0) {          // per-k mean | scale | shift of the BatchNorm in front, zero-padded
        for (int i = threadIdx.x; i < 3 * KP; i += S::NT) {
            const int part = i / KP, k = i - part * KP;
            affL[i] = k < g.K ? g.a_aff[part * g.K + k] : 0.f;
        }
        __syncthreads();
        sa.store_aff(ra_[0], As, affL, KP, k_begin);
    } else {
        sa.store(ra_[0], As);
    }
    sb.store(rb_[0], Bs);
    __syncthreads();
    const int lt = lane % MT, lk = lane / MT;
    for (int kt = 0; kt < nk; kt += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            // tile t is in LDS buffer t & 1; ring slot u (tile t) was written to LDS one iteration ago and is free
            const int t = kt + u;
            const int cur = (PF % 2 == 0) ? (u & 1) : (t & 1);
            sa.template load<VEC, ROWS>(ra_[u], ra, g.a_bytes, g.lda, g.M, k_begin + (t + PF) * BK, k_begin, k_end, kidx);
            sb.template load<VEC, ROWS>(rb_[u], rb, g.b_bytes, g.ldb, g.N, k_begin + (t + PF) * BK, k_begin, k_end, kidx, g.b_split,
                                        g.b_delta);
            if (t < nk) {
                const float* as = As + cur * StageA::LDS_FLOATS;
                const float* bs = Bs + cur * StageB::LDS_FLOATS;
                if constexpr (BF16 != 0) {
                    static_assert(bf16_shape<S>(), "bf16 MFMA: the K-tile is a whole number of instructions");
                    static_assert(!SPLIT || MT == 32, "split products: 32x32 tiles");
                    constexpr int KI = (MT == 32) ? 16 : 32;       // k per instruction
                    constexpr int NI = SPLIT ? 3 : 1;              // images: the rounded operand, or hi | mid | lo
#pragma unroll
                    for (int ks = 0; ks < BK / KI; ++ks) {
                        // lane (lt, lk) supplies k = ks KI + 8 lk .. + 7 of row lt of its tiles
                        const int k0 = ks * KI + 8 * lk;
                        bf16x8 a8[NI][WM_T], b8[NI][WN_T];
#pragma unroll
                        for (int i = 0; i < WM_T; ++i) {
                            const int idx = (wm * WM_T + i) * MT + lt;
                            if constexpr (A_IH) {
#pragma unroll
                                for (int p = 0; p < NI; ++p)
                                    a8[p][i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(as) + p * StageA::IMG_HALVES +
                                                                                idx * StageA::LDKH + k0);
                            } else {
                                float f[8];
                                if constexpr (A_IM) {
#pragma unroll
                                    for (int q = 0; q < 8; q += 2) {
                                        const float2 v = *reinterpret_cast<const float2*>(&as[idx * StageA::LDK + k0 + q]);
                                        f[q] = v.x; f[q + 1] = v.y;
                                    }
                                } else {
#pragma unroll
                                    for (int q = 0; q < 8; ++q) f[q] = as[(k0 + q) * LDA + swz(k0 + q, idx)];
                                }
                                if constexpr (SPLIT) split_bf16x8(f, a8[0][i], a8[1][i], a8[2][i]);
                                else a8[0][i] = to_bf16x8(f);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < WN_T; ++j) {
                            const int idx = (wn * WN_T + j) * MT + lt;
                            if constexpr (B_IH) {
#pragma unroll
                                for (int p = 0; p < NI; ++p)
                                    b8[p][j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(bs) + p * StageB::IMG_HALVES +
                                                                                idx * StageB::LDKH + k0);
                            } else {
                                float f[8];
                                if constexpr (B_IM) {
#pragma unroll
                                    for (int q = 0; q < 8; q += 2) {
                                        const float2 v = *reinterpret_cast<const float2*>(&bs[idx * StageB::LDK + k0 + q]);
                                        f[q] = v.x; f[q + 1] = v.y;
                                    }
                                } else {
#pragma unroll
                                    for (int q = 0; q < 8; ++q) f[q] = bs[(k0 + q) * LDB + swz(k0 + q, idx)];
                                }
                                if constexpr (SPLIT) split_bf16x8(f, b8[0][j], b8[1][j], b8[2][j]);
                                else b8[0][j] = to_bf16x8(f);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < WM_T; ++i)
#pragma unroll
                            for (int j = 0; j < WN_T; ++j) {
                                if constexpr (SPLIT) {
                                    // a b = sum of the six part products of order <= 2 (each exact in fp32), small ones first; the three
                                    // dropped ones are <= 2^-24 |a b| each: the rounding class of one fp32 product
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[2][j], a8[0][i], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0][j], a8[2][i], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[1][j], a8[1][i], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[1][j], a8[0][i], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0][j], a8[1][i], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0][j], a8[0][i], acc[i][j], 0, 0, 0);
                                } else if constexpr (MT == 16) {
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8[0][j], a8[0][i], acc[i][j], 0, 0, 0);
                                } else {
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0][j], a8[0][i], acc[i][j], 0, 0, 0);
                                }
                            }
                    }
                } else if constexpr (S::KP && MT == 32) {
                    // two k-steps per trip: lane half lk supplies k = 4 j + 2 lk (step 2j) and 4 j + 2 lk + 1 (step 2j+1)
#pragma unroll
                    for (int j4 = 0; j4 < BK / 4; ++j4) {
                        const int k0 = 4 * j4 + 2 * lk;
                        float a0[WM_T], a1[WM_T], b0[WN_T], b1[WN_T];
#pragma unroll
                        for (int i = 0; i < WM_T; ++i) {
                            const int idx = (wm * WM_T + i) * MT + lt;
                            if constexpr (A_IM) {
                                const float2 v = *reinterpret_cast<const float2*>(&as[idx * StageA::LDK + k0]);
                                a0[i] = v.x; a1[i] = v.y;
                            } else {
                                a0[i] = as[k0 * LDA + swz(k0, idx)];
                                a1[i] = as[(k0 + 1) * LDA + swz(k0 + 1, idx)];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < WN_T; ++j) {
                            const int idx = (wn * WN_T + j) * MT + lt;
                            if constexpr (B_IM) {
                                const float2 v = *reinterpret_cast<const float2*>(&bs[idx * StageB::LDK + k0]);
                                b0[j] = v.x; b1[j] = v.y;
                            } else {
                                b0[j] = bs[k0 * LDB + swz(k0, idx)];
                                b1[j] = bs[(k0 + 1) * LDB + swz(k0 + 1, idx)];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < WM_T; ++i)
#pragma unroll
                            for (int j = 0; j < WN_T; ++j) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a0[i], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a1[i], acc[i][j], 0, 0, 0);
                            }
                    }
                } else
#pragma unroll
                for (int kk = 0; kk < BK / KSTEP; ++kk) {
                    const int kr = kk * KSTEP + lk;
                    float af[WM_T], bf[WN_T];
#pragma unroll
                    for (int i = 0; i < WM_T; ++i) af[i] = as[kr * LDA + swz(kr, (wm * WM_T + i) * MT + lt)];
#pragma unroll
                    for (int j = 0; j < WN_T; ++j) bf[j] = bs[kr * LDB + swz(kr, (wn * WN_T + j) * MT + lt)];
#pragma unroll
                    for (int i = 0; i < WM_T; ++i)
#pragma unroll
                        for (int j = 0; j < WN_T; ++j) {
                            if constexpr (MT == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[i][j], 0, 0, 0);
                        }
                }
            }
            if constexpr ((FUSE & 1) != 0)
                sa.store_aff(ra_[(u + 1) % PF], As + (cur ^ 1) * StageA::LDS_FLOATS, affL, KP, k_begin + (t + 1) * BK);
            else
                sa.store(ra_[(u + 1) % PF], As + (cur ^ 1) * StageA::LDS_FLOATS);     // tile t+1 (zeros past the end)
            sb.store(rb_[(u + 1) % PF], Bs + (cur ^ 1) * StageB::LDS_FLOATS);
            __syncthreads();
        }
    }

    // epilogue.  D[row = n][col = m]; a lane owns one m (lane % MT) and groups of 4 consecutive n:
    //   MT = 16: n = 4*(lane>>4) + 0..3 (one group);  MT = 32: n = 8*grp + 4*(lane>>5) + 0..3, grp = 0..3
    const bool add_bias = g.bias != nullptr && first_split;
    constexpr int GROUPS = (MT == 16) ? 1 : 4;
    float* const Ct = smem;                               // statistics epilogue: [BM][CT_PITCH] stored values
    float* const red = smem + BM * CT_PITCH;              // [2][4][BN]
    float* const rvalid = red + 2 * 4 * BN;               // [BM] 1 / 0
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
        const int ml = (wm * WM_T + i) * MT + lt;
        const int m = m0 + ml;
        int row = -1;
        if (m < g.M) row = g.m_rows ? g.m_rows[m] : m;
        if constexpr ((FUSE & 2) != 0) {
            if (wn == 0 && lk == 0) rvalid[ml] = row >= 0 ? 1.f : 0.f;
        }
        if (row < 0) continue;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
#pragma unroll
            for (int grp = 0; grp < GROUPS; ++grp) {
                const int nl = (wn * WN_T + j) * MT + ((MT == 16) ? lk * 4 : 8 * grp + 4 * lk);
                const int n = n0 + nl;
                if (n >= g.N) continue;
                float r[4] = {acc[i][j][4 * grp + 0], acc[i][j][4 * grp + 1], acc[i][j][4 * grp + 2], acc[i][j][4 * grp + 3]};
                float* c = Cout + (long)row * g.ldc + n + (row >= g.c_split ? g.c_delta : 0);
                const bool full = (n + 3 < g.N);
                if (add_bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < g.N) r[q] += g.bias[n + q];
                }
                if constexpr ((FUSE & 2) != 0) {
                    // C (+)= ... then the activation; the stored values also go to the LDS tile for the column statistics
                    if (g.accumulate) {
                        const float* ci = g.Cin != nullptr ? g.Cin + (long)row * g.ldcin + n : c;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (n + q < g.N) r[q] += ci[q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = apply_act_c<false>(r[q], g.epi_act);      // (none / ReLU / LeakyReLU: checked by the host)
                    if (g.c_bf16) {      // (N % 4 == 0 checked by the host: a group of 4 is whole)
                        auto rne = [](float x) { const unsigned u = __float_as_uint(x); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
                        unsigned short* c16 = reinterpret_cast<unsigned short*>(Cout) + (long)row * g.ldc + n;
                        *reinterpret_cast<uint2*>(c16) = make_uint2(rne(r[0]) | (rne(r[1]) << 16), rne(r[2]) | (rne(r[3]) << 16));
                    } else if (full && g.c_vec) {
                        *reinterpret_cast<float4*>(c) = make_float4(r[0], r[1], r[2], r[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (n + q < g.N) c[q] = r[q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) Ct[ml * CT_PITCH + nl + q] = r[q];
                    continue;
                }
                if (g.atomic_out) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < g.N) unsafeAtomicAdd(c + q, r[q]);
                } else if (full && g.c_vec) {
                    float4 o = make_float4(r[0], r[1], r[2], r[3]);
                    if (g.accumulate) {
                        float4 old = *reinterpret_cast<const float4*>(c);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(c) = o;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < g.N) c[q] = g.accumulate ? c[q] + r[q] : r[q];
                }
            }
        }
    }
    if constexpr ((FUSE & 2) != 0) {
        // per-tile column statistics: 4 row quarters x BN columns; sum -> tile mean -> M2 about it (no cancellation),
        // every combination in a fixed order (deterministic)
        static_assert(S::NT == 4 * BN, "statistics epilogue: four row quarters per column");
        __syncthreads();
        const int cl = threadIdx.x % BN, part = threadIdx.x / BN;
        constexpr int RQ = BM / 4;
        float x[RQ], v[RQ];
        float s1 = 0.f, cnt = 0.f;
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
            const int ml = part * RQ + k;
            v[k] = rvalid[ml];
            x[k] = v[k] != 0.f ? Ct[ml * CT_PITCH + cl] : 0.f;
            s1 += x[k];
            cnt += v[k];
        }
        red[part * BN + cl] = s1;
        red[4 * BN + part * BN + cl] = cnt;
        __syncthreads();
        const float tot = ((red[cl] + red[BN + cl]) + red[2 * BN + cl]) + red[3 * BN + cl];
        const float n_rows = ((red[4 * BN + cl] + red[5 * BN + cl]) + red[6 * BN + cl]) + red[7 * BN + cl];
        const float mean_t = n_rows > 0.f ? tot / n_rows : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
            const float d = x[k] - mean_t;
            if (v[k] != 0.f) m2 += d * d;
        }
        __syncthreads();
        red[part * BN + cl] = m2;
        __syncthreads();
        if (part == 0 && n0 + cl < g.N) {
            float* o = g.stats + (long)bx * 3 * g.N + n0 + cl;
            o[0] = tot;
            o[g.N] = ((red[cl] + red[BN + cl]) + red[2 * BN + cl]) + red[3 * BN + cl];
            o[2 * g.N] = n_rows;
        }
    }
}

template <class S, bool VEC, bool A_KC, bool B_KC, int BF16 = 0>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) {
    I3D_CHAIN_PRIO();
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    const int slice = bz;              // (slab index: one per batch and K-slice)
    if (g.n_batch > 1) {               // (uniform) batched product: this workgroup's batch, then its K-slice
        const int bb = bz / g.z_splits;
        bz -= bb * g.z_splits;
        g.A += bb * g.a_batch; g.B += bb * g.b_batch; g.C += bb * g.c_batch;
    }
    const int k_begin = bz * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    if (g.slab != nullptr) {           // every slice writes (zeros for an empty one): the reduction reads them all
        GemmArgs gl = g;
        gl.ldc = g.N; gl.accumulate = 0; gl.atomic_out = 0; gl.c_vec = (g.N % 4 == 0); gl.bias = nullptr;
        gl.c_split = 0x7fffffff; gl.c_delta = 0;
        gemm_body<S, VEC, A_KC, B_KC, false, 0, BF16>(gl, bx, by, min(k_begin, g.K), k_end, g.slab + (long)slice * g.M * g.N, nullptr,
                                                      false);
        return;
    }
    if (k_begin >= k_end && !(bz == 0)) return;
    gemm_body<S, VEC, A_KC, B_KC, false, 0, BF16>(g, bx, by, k_begin, k_end, g.C, nullptr, bz == 0);
}

template <class S, bool VEC, int BF16 = 0>
__global__ void __launch_bounds__(256)
gemm_f32_rowseg_kernel(GemmArgs g, SegTable segs) {
    I3D_CHAIN_PRIO();
    __shared__ int kidx[SEG_MAX_K];
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    const Seg sg = segs.s[bz];
    for (int i = threadIdx.x; i < sg.k_end - sg.k_begin; i += S::NT) kidx[i] = g.k_rows[sg.k_begin + i];
    __syncthreads();
    if (g.slab != nullptr) {
        GemmArgs gl = g;
        gl.ldc = g.N; gl.accumulate = 0; gl.atomic_out = 0; gl.c_vec = (g.N % 4 == 0);
        gl.c_split = 0x7fffffff; gl.c_delta = 0;
        gemm_body<S, VEC, false, false, true, 0, BF16>(gl, bx, by, sg.k_begin, sg.k_end, g.slab + (long)bz * g.M * g.N, kidx, false);
        return;
    }
    gemm_body<S, VEC, false, false, true, 0, BF16>(g, bx, by, sg.k_begin, sg.k_end, g.C + sg.c_off, kidx, false);
}

// C_g[m, n] (+)= sum_{z in [seg_ptr[g], seg_ptr[g+1])} slab[z][m][n]   (fixed order: deterministic)
struct SlabReduce {
    const float* slab;
    float* C;
    const float* bias;
    int M, N, ldc, accumulate, n_groups;
    long c_group_stride;
    int c_split;       // rows >= c_split of C are displaced by c_delta floats (two-block outputs)
    long c_delta;
    int seg_ptr[34];
    // weight gradient against a BatchNorm output that was never materialised (fused_bn.hip): the GEMM ran on the raw
    // activation x, y = (x - mean) * scale + shift per column n, so  dW[m][n] = (sum - row[m] mean[n]) scale[n] + row[m] shift[n]
    // with row[m] = sum over the rows of dY[:, m] (the bias gradient).  null: plain sum.
    const float* post_aff;   // [3N] mean | scale | shift
    const float* post_row;   // [M]
};

template <int V>
__global__ void __launch_bounds__(256) slab_reduce_kernel(SlabReduce a) {
    I3D_CHAIN_PRIO();
    const int NV = a.N / V;
    const long per_group = (long)a.M * NV;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per_group * a.n_groups) return;
    const int gi = (int)(t / per_group);
    const long r = t - gi * per_group;
    const int m = (int)(r / NV), n = (int)(r - (long)m * NV) * V;
    float* c = a.C + gi * a.c_group_stride + (long)m * a.ldc + n + (m >= a.c_split ? a.c_delta : 0);
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = a.accumulate ? c[i] : 0.f;
    const long slice = (long)a.M * a.N;
    const float* p = a.slab + (long)m * a.N + n;
    const int z_end = a.seg_ptr[gi + 1];
    for (int z0 = a.seg_ptr[gi]; z0 < z_end; z0 += 4) {     // four slices in flight per trip, summed in slice order
        float x[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* q = p + (long)min(z0 + k, z_end - 1) * slice;
            if (V == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(q);
                x[k][0] = t4.x; x[k][1 % V] = t4.y; x[k][2 % V] = t4.z; x[k][3 % V] = t4.w;
            } else {
                x[k][0] = q[0];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (z0 + k < z_end) {
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += x[k][i];
            }
        }
    }
    if (a.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] += a.bias[n + i];
    }
    if (a.post_aff != nullptr) {
        const float rw = a.post_row[m];
#pragma unroll
        for (int i = 0; i < V; ++i)
            acc[i] = (acc[i] - rw * a.post_aff[n + i]) * a.post_aff[a.N + n + i] + rw * a.post_aff[2 * a.N + n + i];
    }
    if (V == 4) *reinterpret_cast<float4*>(c) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
    else c[0] = acc[0];
}

// the same fix-up as a kernel of its own (the product did not go through the slab)
__global__ void __launch_bounds__(256)
wgrad_bn_fixup_kernel(float* __restrict__ C, int M, int N, int ldc, const float* __restrict__ aff, const float* __restrict__ row) {
    I3D_CHAIN_PRIO();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)M * N) return;
    const int m = (int)(t / N), n = (int)(t - (long)m * N);
    const float rw = row[m];
    C[(long)m * ldc + n] = (C[(long)m * ldc + n] - rw * aff[n]) * aff[N + n] + rw * aff[2 * N + n];
}

// few outputs, many slices (the 20-wide 3D network: 400 outputs, ~270 slices): 16 lanes share the slices of one output
// item and are combined through LDS in lane order (still a fixed summation order)
constexpr int ZL = 16;
template <int V>
__global__ void __launch_bounds__(256) slab_reduce_small_kernel(SlabReduce a) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[256 / ZL][ZL][V];
    const int NV = a.N / V;
    const long per_group = (long)a.M * NV;
    const int il = threadIdx.x / ZL, zl = threadIdx.x % ZL;
    const long t = (long)blockIdx.x * (256 / ZL) + il;
    const bool live = t < per_group * a.n_groups;
    int gi = 0, m = 0, n = 0;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    if (live) {
        gi = (int)(t / per_group);
        const long r = t - gi * per_group;
        m = (int)(r / NV);
        n = (int)(r - (long)m * NV) * V;
        const long slice = (long)a.M * a.N;
        const float* p = a.slab + (long)m * a.N + n;
        const int z_end = a.seg_ptr[gi + 1];
        for (int z0 = a.seg_ptr[gi] + zl; z0 < z_end; z0 += 4 * ZL) {      // four of this lane's slices in flight per trip
            float x[4][V];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int z = z0 + k * ZL;
                const float* q = p + (long)(z < z_end ? z : z0) * slice;
                if (V == 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(q);
                    x[k][0] = t4.x; x[k][1 % V] = t4.y; x[k][2 % V] = t4.z; x[k][3 % V] = t4.w;
                } else {
                    x[k][0] = q[0];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (z0 + k * ZL < z_end) {
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[i] += x[k][i];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) sm[il][zl][i] = acc[i];
    __syncthreads();
    if (!live || zl != 0) return;
    float* c = a.C + gi * a.c_group_stride + (long)m * a.ldc + n + (m >= a.c_split ? a.c_delta : 0);
    float tot[V];
#pragma unroll
    for (int i = 0; i < V; ++i) tot[i] = a.accumulate ? c[i] : 0.f;
    for (int k = 0; k < ZL; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) tot[i] += sm[il][k][i];
    if (a.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < V; ++i) tot[i] += a.bias[n + i];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) c[i] = tot[i];
}

static void launch_slab_reduce(const SlabReduce& a, hipStream_t s) {
    const bool v4 = a.N % 4 == 0 && a.ldc % 4 == 0 && a.c_group_stride % 4 == 0 && a.c_delta % 4 == 0 &&
                    (((uintptr_t)a.C | (uintptr_t)a.slab) & 15) == 0;
    const long items = (long)a.n_groups * a.M * (v4 ? a.N / 4 : a.N);
    const int slices = a.seg_ptr[a.n_groups] - a.seg_ptr[0];
    if (items <= 8192 && slices >= 4 * ZL && a.post_aff == nullptr) {
        dim3 grid(cdiv(items, 256 / ZL));
        if (v4) hipLaunchKernelGGL(slab_reduce_small_kernel<4>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(slab_reduce_small_kernel<1>, grid, dim3(256), 0, s, a);
        return;
    }
    if (v4) hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3(cdiv(items, 256)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3(cdiv(items, 256)), dim3(256), 0, s, a);
}

// layout: 0 forward (A and B k-contiguous), 1 data gradient (A k-contiguous, B idx-contiguous),
//         2 weight gradient (both idx-contiguous), 3 (A idx-contiguous, B k-contiguous; not on the training path)
// 64x32 tiles of two waves instead of 64x64: when N pads much better in 32-wide column tiles (200 -> 224 vs 256) and the
// launch is small enough for the larger number of smaller workgroups to matter
static bool narrow_pays(int N, long tiles64) {
    return tiles64 < 1100 && (long)cdiv(N, 64) * 64 * 10 > (long)cdiv(N, 32) * 32 * 11;
}
// the fused (BatchNorm prologue / statistics epilogue) forward GEMM at K < 400 keeps 64x64 tiles: measured 2.464 ms per
// step with the narrow tiling against 2.426 ms without (tools/ab.sh, 4 interleaved runs) (left off)
static bool fuse_narrow_short_k() { return false; }

// process-level: 0 = fp32 MFMA (exact fp32 products), 1 = bf16 MFMA on bf16-rounded operands (i3d_set_matmul_precision)
static int g_matmul_bf16 = 0;
// fp32 mode: how the 32x32-tile kernels with an idx-major LDS image (the forward and data-gradient GEMMs of the chain) form a product
// of two fp32 operands (i3d_set_fp32_products; I3D_FP32_PRODUCTS=native|split at load):
//   0 native: v_mfma_f32_32x32x2_f32, 64 cycles per 2 k (exact products, one rounding per accumulation)
//   1 split : a = a_h + a_m + a_l, b likewise (three bf16 parts each, the decomposition is exact); a b ~ the six part products of
//             order <= 2 on the bf16 pipe (6 x v_mfma_f32_32x32x16_bf16 = 192 cycles per 16 k against 512), fp32 accumulation.  Every
//             part product is exact in fp32; the three dropped ones are <= 2^-24 |a b| each - what the rounding of ONE fp32
//             accumulation costs.  Against an fp64 product the error is that of an fp32 GEMM in another summation order
//             (tests/test_gpu_ops.py: test_split_products_are_fp32_products).
static int g_fp32_split = [] { const char* e = getenv("I3D_FP32_PRODUCTS"); return (e != nullptr && e[0] == 'n') ? 0 : 1; }();
template <class S> constexpr bool split_shape() {      // the 64x64 and 64x32 tilings of the chain (Cfg9, Cfg11)
    return S::KP && S::MT == 32 && S::BK == 16 && S::WAVES_M == 2 && S::WM_T == 1 && S::WN_T == 1;
}

template <class S>
static void launch(const GemmArgs& g, int layout, int splits, bool vec, hipStream_t s) {
    dim3 grid(cdiv(g.M, S::BM), cdiv(g.N, S::BN), splits * (g.n_batch > 1 ? g.n_batch : 1)), block(S::NT);
    if constexpr (split_shape<S>()) {
        if (!g_matmul_bf16 && g_fp32_split && vec && layout <= 1) {
            if (layout == 0) hipLaunchKernelGGL((gemm_f32_kernel<S, true, true, true, 2>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<S, true, true, false, 2>), grid, block, 0, s, g);
            return;
        }
    }
    if constexpr (bf16_shape<S>()) {
        if (g_matmul_bf16 && vec) {      // (the unaligned variants are not on the training path: they stay fp32)
            switch (layout) {
                case 0: hipLaunchKernelGGL((gemm_f32_kernel<S, true, true, true, true>), grid, block, 0, s, g); break;
                case 1: hipLaunchKernelGGL((gemm_f32_kernel<S, true, true, false, true>), grid, block, 0, s, g); break;
                default: hipLaunchKernelGGL((gemm_f32_kernel<S, true, false, false, true>), grid, block, 0, s, g); break;
            }
            return;
        }
    }
#define I3D_GEMM_LAUNCH(V, AK, BKC) hipLaunchKernelGGL((gemm_f32_kernel<S, V, AK, BKC>), grid, block, 0, s, g)
    switch (layout * 2 + (vec ? 1 : 0)) {
        case 0: I3D_GEMM_LAUNCH(false, true, true); break;
        case 1: I3D_GEMM_LAUNCH(true, true, true); break;
        case 2: I3D_GEMM_LAUNCH(false, true, false); break;
        case 3: I3D_GEMM_LAUNCH(true, true, false); break;
        case 4: I3D_GEMM_LAUNCH(false, false, false); break;
        default: I3D_GEMM_LAUNCH(true, false, false); break;
    }
#undef I3D_GEMM_LAUNCH
}

template <class S>
static void launch_rowseg(const GemmArgs& g, const SegTable& t, int n_segs, bool vec, hipStream_t s) {
    dim3 grid(cdiv(g.M, S::BM), cdiv(g.N, S::BN), n_segs);
    if constexpr (bf16_shape<S>()) {
        if (g_matmul_bf16 && vec) {
            hipLaunchKernelGGL((gemm_f32_rowseg_kernel<S, true, true>), grid, dim3(S::NT), 0, s, g, t);
            return;
        }
    }
    if (vec) hipLaunchKernelGGL((gemm_f32_rowseg_kernel<S, true>), grid, dim3(S::NT), 0, s, g, t);
    else hipLaunchKernelGGL((gemm_f32_rowseg_kernel<S, false>), grid, dim3(S::NT), 0, s, g, t);
}

// tile configurations; the numbering is part of the tuning entry i3d_gemm_f32_ex
//   0: 128x128x16 (32x32x2)  1: 256x32x16 (16x16x4)  2: 64x64x16 (32x32x2)  3: 32x64x32 (16x16x4)
//   4: 64x64x32 (32x32x2)    5: 128x64x16 (32x32x2)  6 / 7: as 2 with 1 / 2 K-tiles in flight instead of 4
//   8: 32x32x32 (16x16x4): weight gradients of the narrow (hidden_dim 20) 3D network, K = number of edges
//   9 / 10: as 2 / 0 with the idx-major LDS image (TileStage IM) for k-contiguous operands
//   11: 64x32x16, two waves (32x32x2, idx-major image)   12: 32x32x16, one wave
typedef Shape<32, 2, 2, 2, 2, 16, 2> Cfg0;
typedef Shape<16, 4, 1, 4, 2, 16, 2> Cfg1;
typedef Shape<32, 2, 2, 1, 1, 16, 4> Cfg2;
typedef Shape<16, 2, 2, 1, 2, 32, 2> Cfg3;
typedef Shape<32, 2, 2, 1, 1, 32, 2> Cfg4;
typedef Shape<32, 2, 2, 2, 1, 16, 4> Cfg5;
typedef Shape<32, 2, 2, 1, 1, 16, 1> Cfg6;
typedef Shape<32, 2, 2, 1, 1, 16, 2> Cfg7;
typedef Shape<16, 2, 2, 1, 1, 32, 2> Cfg8;
typedef Shape<32, 2, 2, 1, 1, 16, 4, true> Cfg9;       // as 2 with the idx-major LDS image for k-contiguous operands
typedef Shape<32, 2, 2, 2, 2, 16, 2, true> Cfg10;      // as 0 with it
typedef Shape<32, 2, 1, 1, 1, 16, 4, true> Cfg11;      // 64x32 tile, two waves: N = 200 in 7 column tiles instead of 4 x 64
typedef Shape<32, 1, 1, 1, 1, 16, 4, true> Cfg12;      // 32x32 tile, one wave
// bf16 matmul mode, operands with a bf16 LDS image: with one cheap MFMA per 16 k the K-tile's barrier and staging are what
// a tile costs - longer K-tiles (2 / 4 instructions per tile and barrier)
typedef Shape<32, 2, 2, 1, 1, 32, 2, true> Cfg13;      // as 9, K-tile 32
typedef Shape<32, 2, 1, 1, 1, 32, 2, true> Cfg14;      // as 11, K-tile 32
typedef Shape<32, 2, 2, 1, 1, 64, 2, true> Cfg15;      // as 9, K-tile 64
typedef Shape<32, 2, 1, 1, 1, 64, 2, true> Cfg16;      // as 11, K-tile 64
constexpr int N_CFG = 17;
static const int CFG_BM[N_CFG] = {Cfg0::BM, Cfg1::BM, Cfg2::BM, Cfg3::BM, Cfg4::BM, Cfg5::BM, Cfg6::BM, Cfg7::BM, Cfg8::BM, Cfg9::BM, Cfg10::BM, Cfg11::BM, Cfg12::BM, Cfg13::BM, Cfg14::BM, Cfg15::BM, Cfg16::BM};
static const int CFG_BN[N_CFG] = {Cfg0::BN, Cfg1::BN, Cfg2::BN, Cfg3::BN, Cfg4::BN, Cfg5::BN, Cfg6::BN, Cfg7::BN, Cfg8::BN, Cfg9::BN, Cfg10::BN, Cfg11::BN, Cfg12::BN, Cfg13::BN, Cfg14::BN, Cfg15::BN, Cfg16::BN};
static const int CFG_BK[N_CFG] = {Cfg0::BK, Cfg1::BK, Cfg2::BK, Cfg3::BK, Cfg4::BK, Cfg5::BK, Cfg6::BK, Cfg7::BK, Cfg8::BK, Cfg9::BK, Cfg10::BK, Cfg11::BK, Cfg12::BK, Cfg13::BK, Cfg14::BK, Cfg15::BK, Cfg16::BK};

// K-tile of the bf16-image configurations (32: measured against 16 and 64 in round 3)
static int bf16_bk() {
    return 32;
}

// the (A idx-contiguous, B k-contiguous) layout is computed as layout 2 would need B transposed: it only exists for
// API completeness, through one configuration
template <bool VEC>
__global__ void __launch_bounds__(256)
gemm_f32_tt_kernel(GemmArgs g) {
    I3D_CHAIN_PRIO();
    gemm_body<Cfg6, VEC, false, true, false>(g, blockIdx.x, blockIdx.y, 0, g.K, g.C, nullptr, true);
}

// forward layout (A and B k-contiguous, 16-byte loads) with the fused BatchNorm prologue / statistics epilogue
template <class S, int FUSE, int BF16 = 0>
__global__ void __launch_bounds__(256)
gemm_f32_fused_kernel(GemmArgs g) {
    I3D_CHAIN_PRIO();
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    gemm_body<S, true, true, true, false, FUSE, BF16>(g, bx, by, 0, g.K, g.C, nullptr, true);
}

// The same with five waves per SIMD asked of the register allocator, for the four-wave 64x64 shape in fp32 (82 + 16 registers
// -> 88, no spill; the two-wave 64x32 shape would spill).  [E, 200] x [200, 200] is 1040 tiles: at four workgroups per CU the
// chip holds 1024 and the last 16 run as a second round - the statistics variant of that GEMM 35.4 -> 30.7 us, the fused
// (prologue + statistics) one 40.2 -> 34.0 us back to back (tools/fused_gemm_bench.py), step 2.163 -> 2.144 ms.
template <class S, int FUSE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5)))
gemm_f32_fused_o5_kernel(GemmArgs g) {
    I3D_CHAIN_PRIO();
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    gemm_body<S, true, true, true, false, FUSE, false>(g, bx, by, 0, g.K, g.C, nullptr, true);
}

template <class S>
static void launch_fused(const GemmArgs& g, int fuse, hipStream_t s) {
    dim3 grid(cdiv(g.M, S::BM), cdiv(g.N, S::BN), 1), block(S::NT);
    if (g_matmul_bf16) {
        switch (fuse) {
            case 1: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 1, true>), grid, block, 0, s, g); break;
            case 2: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 2, true>), grid, block, 0, s, g); break;
            default: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 3, true>), grid, block, 0, s, g); break;
        }
        return;
    }
    if constexpr (split_shape<S>()) {
        if (g_fp32_split) {
            switch (fuse) {
                case 1: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 1, 2>), grid, block, 0, s, g); break;
                case 2: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 2, 2>), grid, block, 0, s, g); break;
                default: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 3, 2>), grid, block, 0, s, g); break;
            }
            return;
        }
    }
    constexpr bool occ5 = true;
    if constexpr (S::WAVES_N == 2 && S::BK == 16) {
        if (occ5) {
            switch (fuse) {
                case 1: hipLaunchKernelGGL((gemm_f32_fused_o5_kernel<S, 1>), grid, block, 0, s, g); break;
                case 2: hipLaunchKernelGGL((gemm_f32_fused_o5_kernel<S, 2>), grid, block, 0, s, g); break;
                default: hipLaunchKernelGGL((gemm_f32_fused_o5_kernel<S, 3>), grid, block, 0, s, g); break;
            }
            return;
        }
    }
    switch (fuse) {
        case 1: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 1>), grid, block, 0, s, g); break;
        case 2: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 2>), grid, block, 0, s, g); break;
        default: hipLaunchKernelGGL((gemm_f32_fused_kernel<S, 3>), grid, block, 0, s, g); break;
    }
}

struct Extra {
    const int* m_rows = nullptr;
    const int* k_rows = nullptr;
    const int* tile_group = nullptr;
    long b_group_stride = 0;
    long a_rows_total = -1;   // number of physical rows of A / C when m_rows is used (for the descriptor extent)
    long k_rows_total = -1;   // number of physical rows of the operands when k_rows is used
    // row-subset reductions: n_groups disjoint ranges [start, start + count) of k_rows, one output per group
    int n_groups = 0;
    const int* group_start = nullptr;
    const int* group_count = nullptr;
    long c_group_stride = 0;
    // scratch for the two-stage split-K / row-segment reduction (slices x M x N floats); null or too small: atomics
    void* workspace = nullptr;
    long workspace_bytes = 0;
    // two-block operands (GemmArgs::b_split ...); b_view_floats = addressable floats behind B when b_split is used
    int b_split = 0x7fffffff, c_split = 0x7fffffff;
    long b_delta = 0, c_delta = 0, b_view_floats = 0;
    const float* post_aff = nullptr;   // SlabReduce::post_aff / post_row
    const float* post_row = nullptr;
    int n_batch = 1;                   // GemmArgs::n_batch ...
    long a_batch = 0, b_batch = 0, c_batch = 0;
};

static int fill_views(GemmArgs& g, int trans_a, int trans_b, int M, int N, int K, int lda, int ldb, const Extra& ex) {
    long a_rows = trans_a ? K : M, a_cols = trans_a ? M : K, b_rows = trans_b ? N : K, b_cols = trans_b ? K : N;
    if (!trans_a && ex.m_rows) a_rows = ex.a_rows_total;
    if (trans_a && ex.k_rows) a_rows = ex.k_rows_total;
    if (!trans_b && ex.k_rows) b_rows = ex.k_rows_total;
    const long ab = a_rows > 0 ? ((a_rows - 1) * lda + a_cols) * 4 : 0, bb = b_rows > 0 ? ((b_rows - 1) * ldb + b_cols) * 4 : 0;
    I3D_CHECK_ARG(ab < (1L << 32) - 16 && bb < (1L << 32) - 16, "operand view larger than 4 GiB (32-bit buffer offsets)");
    g.a_bytes = (unsigned)ab;
    g.b_bytes = (unsigned)bb;
    return I3D_OK;
}

static int gemm_impl(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int accumulate, int force_cfg, int force_splits,
                     const Extra& ex, void* stream) {
    I3D_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    I3D_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "leading dimension too small");
    if (M == 0 || N == 0) return I3D_OK;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g = {};
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate ? 1 : 0;
    g.m_rows = ex.m_rows; g.k_rows = nullptr; g.tile_group = ex.tile_group; g.b_group_stride = ex.b_group_stride;
    g.slab = nullptr;
    g.a_aff = nullptr; g.stats = nullptr; g.epi_act = I3D_ACT_NONE; g.Cin = nullptr; g.ldcin = 0; g.c_bf16 = 0;
    g.b_split = ex.b_split; g.b_delta = ex.b_delta; g.c_split = ex.c_split; g.c_delta = ex.c_delta;
    int rc = fill_views(g, trans_a, trans_b, M, N, K, lda, ldb, ex);
    if (rc != I3D_OK) return rc;
    if (ex.b_view_floats > 0) {
        I3D_CHECK_ARG(ex.b_view_floats * 4 < (1L << 32) - 16, "operand view larger than 4 GiB (32-bit buffer offsets)");
        g.b_bytes = (unsigned)(ex.b_view_floats * 4);
    }
    const bool a_al = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0), b_al = (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0);
    g.c_vec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);
    // fast path: 16-byte loads need aligned pointers / leading dimensions and contiguous extents that are
    // multiples of 4 (K for k-contiguous operands, M or N for the others)
    const int nb = ex.n_batch > 1 ? ex.n_batch : 1;
    const bool vec = a_al && b_al && ((trans_a ? M : K) % 4 == 0) && ((trans_b ? K : N) % 4 == 0) &&
                     (ex.b_group_stride % 4 == 0) && (ex.b_delta % 4 == 0) && (nb == 1 || (ex.a_batch % 4 == 0 && ex.b_batch % 4 == 0));
    if (ex.c_delta % 4 != 0 || (nb > 1 && ex.c_batch % 4 != 0)) g.c_vec = 0;
    const int layout = trans_a ? (trans_b ? 3 : 2) : (trans_b ? 0 : 1);
    if (nb > 1) {
        I3D_CHECK_ARG(nb <= 32 && layout != 3 && (trans_a == 0 || (ex.m_rows == nullptr && ex.tile_group == nullptr)) &&
                          ex.b_split == 0x7fffffff && ex.c_split == 0x7fffffff && bias == nullptr && ex.post_aff == nullptr,
                      "batched product: plain (or row-grouped forward / data-gradient) operands, at most 32 batches");
        g.n_batch = nb; g.a_batch = ex.a_batch; g.b_batch = ex.b_batch; g.c_batch = ex.c_batch;
    }

    if (layout == 3) {
        I3D_CHECK_ARG(ex.m_rows == nullptr && ex.tile_group == nullptr, "row indirection needs a k-contiguous A");
        g.k_per_split = K; g.atomic_out = 0;
        dim3 grid(cdiv(M, Cfg6::BM), cdiv(N, Cfg6::BN), 1);
        if (vec) hipLaunchKernelGGL((gemm_f32_tt_kernel<true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_f32_tt_kernel<false>), grid, dim3(256), 0, s, g);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }

    // tile configuration, measured on MI355X at the step's shapes (tools/gemm_bench.py, profiles/r01_gemm_bench_*.log):
    // many 64x64 tiles beat fewer big ones - also with thousands of tiles (batch 4096, `gemm_bench.py --nodes 67434 --edges
    // 133440 --all-cfgs --wide-sweep`, profiles/r02_gemm_sweep_B4096.log: the 128x128 tiling the first version switched
    // to at >= 4096 tiles is 20-30 % slower than 64x64 on every shape of the step).
    int cfg;
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64) * nb;
    const bool have_ws = ex.workspace != nullptr && (((uintptr_t)ex.workspace & 15) == 0) && ex.m_rows == nullptr;
    if (trans_a && M <= 32 && N <= 32) cfg = 8;
    else if (trans_a && have_ws && tiles64 <= 16) cfg = 8;   // small weight gradients: 32x32 tiles, slices through the scratch
    else if (N <= 32) cfg = (nb > 1 && !trans_a && trans_b) ? 11 : 1;   // (batched: many 64x32 tiles instead of few 256x32 ones)
    else if (trans_a && tiles64 < 512) cfg = 3;   // weight gradients: few output tiles, long K
    else cfg = 2;
    if (ex.tile_group != nullptr) cfg = 2;        // the group padding of m_rows is 64 rows
    if (cfg == 2 && (!trans_a || trans_b)) cfg = 9;   // a k-contiguous operand: idx-major LDS image (2-4 % faster, r01_gemm_bench_v5)
    // N = 200 fills 6.25 of the 8 32-wide wave tiles of four 64-wide column tiles (a quarter of the waves do nothing
    // useful); in 64x32 tiles of two waves it is 7 column tiles, 12 % padding: -18..-21 % on the long-K forward shapes
    // (post4 [N,4F] x [F,4F]^T: 50 -> 41 us), slower when K is short (r01_gemm_bench_v6)
    // Round 2 sweeps (profiles/r02_gemm_sweep_*.log): it also wins at K = 200 while the launch has few tiles (batch 512:
    // P 17.5 -> 15.2 us, [E,F]x[F,F] 25.9 -> 24.0 us), and loses at every K once there are > ~1000 64x64 tiles (QMugs
    // shape, post4: 96 vs 87 us): the tile count decides, not K.
    if (cfg == 9 && !trans_a && narrow_pays(N, tiles64)) cfg = 11;
    // Launches of at most ~4 32x32 tiles per CU (the projection head and the loss at batch 512: [512,600]x[200,600]^T is 80
    // 64x64 tiles on 256 CUs) are bound by ONE workgroup's serial chain, not by throughput: 32x32 tiles of four 16x16 waves
    // have a quarter of the MFMA cycles per k and wave (tools/small_gemm_sweep.py, profiles/r03_small_gemm_sweep.txt: K = 600 18.9 -> 9.6 us,
    // K = 200 9.0 -> 5.3 us,
    // break-even at ~1000 tiles); weight-gradient layouts only while K is short (long K: slices through the scratch, above)
    const long tiles32 = (long)cdiv(M, 32) * cdiv(N, 32) * nb;
    constexpr bool small_tiles = true;
    if (small_tiles && ex.tile_group == nullptr && N > 32 && tiles32 <= 1024 && (!trans_a || K < 2048)) cfg = 8;
    if (g_matmul_bf16 && vec && (cfg == 9 || cfg == 11) && bf16_bk() != 16)      // bf16 LDS image: longer K-tiles
        cfg = (cfg == 9 ? 13 : 14) + (bf16_bk() == 64 ? 2 : 0);
    if (force_cfg >= 0) {
        I3D_CHECK_ARG(force_cfg < N_CFG, "tile_cfg out of range");
        I3D_CHECK_ARG(ex.tile_group == nullptr || CFG_BM[force_cfg] == 64, "grouped GEMM needs 64-row tiles");
        cfg = force_cfg;
    }
    const int bm = CFG_BM[cfg], bn = CFG_BN[cfg], BK = CFG_BK[cfg];
    int tiles = cdiv(M, bm) * cdiv(N, bn) * nb;
    int splits = 1;
    // split-K ONLY for the row-reduction GEMMs of the backward pass (trans_a: dW = dY^T X, K = number of rows).
    // Forward GEMMs stay single-pass and bit-deterministic, so that the arg-max/arg-min routing of the aggregators and
    // readouts cannot flip from run to run on near-ties.  With scratch the slices are summed in a fixed order by
    // slab_reduce_kernel (>= 1024 of K per slice, up to ~800 workgroups); without it they are fp32 atomics on top of a
    // zero-fill (not free: up to ~512 workgroups, >= 512 of K per slice).
    if (trans_a && have_ws && tiles < 800 && K >= 2048) {
        // ~800 workgroups at batch 512 (K = 8-17 k rows); with K in the 100 k (batch 4096) more, shorter slices win:
        // [200,200] K = 133 k: 16 slices 192 us, 64 slices 166 us
        const int target = K >= 65536 ? 3072 : (K >= 32768 ? 1600 : 800);
        splits = (target + tiles / 2) / tiles;
        int max_splits = K / 1024;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    } else if (trans_a && tiles < 512 && K >= 1024) {
        splits = (512 + tiles - 1) / tiles;
        int max_splits = K / 512;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (force_splits > 0) splits = force_splits;
    int kps = cdiv(cdiv(K, splits), BK) * BK;
    if (kps < BK) kps = BK;
    splits = K > 0 ? cdiv(K, kps) : 1;
    g.k_per_split = kps;
    g.atomic_out = splits > 1;
    const bool use_slab = splits > 1 && ex.workspace != nullptr && (long)splits * nb * M * N * 4 <= ex.workspace_bytes &&
                          (((uintptr_t)ex.workspace & 15) == 0) && ex.m_rows == nullptr;
    g.z_splits = splits;
    I3D_CHECK_ARG(nb == 1 || splits == 1 || use_slab, "batched product with K-slices needs the scratch");
    if (use_slab) {
        g.slab = (float*)ex.workspace;
        g.atomic_out = 0;
    }
    if (splits > 1 && !accumulate && !use_slab) {
        // atomics accumulate on top of zeros (a two-block C: both blocks)
        const int m1 = std::min(M, ex.c_split);
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), m1, s);
        if (e == hipSuccess && M > m1)
            e = hipMemset2DAsync(C + (long)m1 * ldc + ex.c_delta, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M - m1, s);
        if (e != hipSuccess) {
            set_error("i3d_gemm_f32: memset failed");
            return I3D_ERR_LAUNCH;
        }
    }
    switch (cfg) {
        case 0: launch<Cfg0>(g, layout, splits, vec, s); break;
        case 1: launch<Cfg1>(g, layout, splits, vec, s); break;
        case 2: launch<Cfg2>(g, layout, splits, vec, s); break;
        case 3: launch<Cfg3>(g, layout, splits, vec, s); break;
        case 4: launch<Cfg4>(g, layout, splits, vec, s); break;
        case 5: launch<Cfg5>(g, layout, splits, vec, s); break;
        case 6: launch<Cfg6>(g, layout, splits, vec, s); break;
        case 7: launch<Cfg7>(g, layout, splits, vec, s); break;
        case 8: launch<Cfg8>(g, layout, splits, vec, s); break;
        case 9: launch<Cfg9>(g, layout, splits, vec, s); break;
        case 10: launch<Cfg10>(g, layout, splits, vec, s); break;
        case 11: launch<Cfg11>(g, layout, splits, vec, s); break;
        case 12: launch<Cfg12>(g, layout, splits, vec, s); break;
        case 13: launch<Cfg13>(g, layout, splits, vec, s); break;
        case 14: launch<Cfg14>(g, layout, splits, vec, s); break;
        case 15: launch<Cfg15>(g, layout, splits, vec, s); break;
        default: launch<Cfg16>(g, layout, splits, vec, s); break;
    }
    I3D_CHECK_LAUNCH();
    if (use_slab) {
        SlabReduce r;
        r.slab = g.slab; r.C = C; r.bias = bias; r.M = M; r.N = N; r.ldc = ldc; r.accumulate = accumulate ? 1 : 0;
        r.n_groups = nb; r.c_group_stride = ex.c_batch;
        for (int b = 0; b <= nb; ++b) r.seg_ptr[b] = b * splits;
        r.c_split = ex.c_split; r.c_delta = ex.c_delta;
        r.post_aff = ex.post_aff; r.post_row = ex.post_row;
        launch_slab_reduce(r, s);
        I3D_CHECK_LAUNCH();
    } else if (ex.post_aff != nullptr) {
        hipLaunchKernelGGL(wgrad_bn_fixup_kernel, dim3(cdiv((long)M * N, 256)), dim3(256), 0, s, C, M, N, ldc, ex.post_aff,
                           ex.post_row);
        I3D_CHECK_LAUNCH();
    }
    return I3D_OK;
}

// C_g[M,N] (+)= sum_{j in group g} A[k_rows[j], 0:M]^T B[k_rows[j], 0:N] for all groups in ONE launch: the reduction
// ranges are cut into <= MAX_SEGS segments of <= SEG_MAX_K rows, each a blockIdx.z slice that accumulates with atomics.
static int rowseg_impl(int M, int N, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int accumulate,
                       int force_cfg, int force_seg_rows, const Extra& ex, void* stream) {
    I3D_CHECK_ARG(M > 0 && N > 0 && ex.n_groups > 0 && ex.k_rows != nullptr, "bad arguments");
    I3D_CHECK_ARG(lda >= M && ldb >= N && ldc >= N, "leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    long total = 0;
    int k_hi = 0;
    for (int gi = 0; gi < ex.n_groups; ++gi) {
        I3D_CHECK_ARG(ex.group_start[gi] >= 0 && ex.group_count[gi] >= 0, "bad group range");
        total += ex.group_count[gi];
        k_hi = std::max(k_hi, ex.group_start[gi] + ex.group_count[gi]);
    }
    GemmArgs g = {};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = k_hi;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = 1; g.atomic_out = 1; g.k_per_split = 0;
    g.m_rows = nullptr; g.k_rows = ex.k_rows; g.tile_group = nullptr; g.b_group_stride = 0;
    g.slab = nullptr;
    g.a_aff = nullptr; g.stats = nullptr; g.epi_act = I3D_ACT_NONE; g.Cin = nullptr; g.ldcin = 0; g.c_bf16 = 0;
    g.b_split = g.c_split = 0x7fffffff; g.b_delta = g.c_delta = 0;
    int rc = fill_views(g, 1, 0, M, N, k_hi, lda, ldb, ex);
    if (rc != I3D_OK) return rc;
    g.c_vec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);
    // upper bound of the number of segments (>= 512 rows each unless forced): enough scratch -> two-stage reduction
    const bool want_slab = ex.workspace != nullptr && (((uintptr_t)ex.workspace & 15) == 0) && ex.n_groups <= 32;
    if (!accumulate && !want_slab) {
        for (int gi = 0; gi < ex.n_groups; ++gi) {
            float* Cg = C + gi * ex.c_group_stride;
            hipError_t e = (ldc == N) ? hipMemsetAsync(Cg, 0, (size_t)M * N * sizeof(float), s)
                                      : hipMemset2DAsync(Cg, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
            if (e != hipSuccess) {
                set_error("i3d_gemm_f32_rowsubset: memset failed");
                return I3D_ERR_LAUNCH;
            }
        }
    }
    if (total == 0) {
        if (!accumulate && want_slab)
            for (int gi = 0; gi < ex.n_groups; ++gi)
                if (hipMemset2DAsync(C + gi * ex.c_group_stride, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess)
                    return I3D_ERR_LAUNCH;
        return I3D_OK;
    }
    int cfg = want_slab ? 2 : 3;      // measured (tools/gemm_bench.py): 64x64 tiles once the epilogue is a plain store
    if (force_cfg >= 0) {
        I3D_CHECK_ARG(force_cfg >= 2 && force_cfg <= 4, "row-subset GEMM: tile_cfg must be 2, 3 or 4");
        cfg = force_cfg;
    }
    const int BK = CFG_BK[cfg];
    const int tiles = cdiv(M, CFG_BM[cfg]) * cdiv(N, CFG_BN[cfg]);
    // the atomics of the epilogue are the expensive part of a segment: ~1000 workgroups, >= 1024 rows per segment
    long want = std::max<long>(ex.n_groups, (1024 + tiles / 2) / tiles);
    long seg_rows = std::max<long>(want_slab ? 512 : 1024, (long)cdiv((int)cdiv((int)total, (int)want), BK) * BK);
    if (force_seg_rows > 0) seg_rows = (long)cdiv(force_seg_rows, BK) * BK;
    SegTable t;
    int seg_first[33];          // first segment of every group (segments of a group are consecutive table entries)
    int n_segs;
    for (;;) {
        if (seg_rows > SEG_MAX_K) seg_rows = SEG_MAX_K;
        n_segs = 0;
        bool fits = true;
        for (int gi = 0; gi < ex.n_groups && fits; ++gi) {
            const int b = ex.group_start[gi], e = b + ex.group_count[gi];
            if (gi < 33) seg_first[gi] = n_segs;
            for (int k = b; k < e; k += (int)seg_rows) {
                if (n_segs == MAX_SEGS) { fits = false; break; }
                t.s[n_segs++] = Seg{k, std::min<int>(e, k + (int)seg_rows), gi * ex.c_group_stride};
            }
        }
        if (fits) break;
        I3D_CHECK_ARG(seg_rows < SEG_MAX_K, "row-subset GEMM: more than MAX_SEGS * SEG_MAX_K rows");
        seg_rows *= 2;
    }
    const bool vec = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0) && (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0) &&
                     (M % 4 == 0) && (N % 4 == 0);
    const bool use_slab = want_slab && (long)n_segs * M * N * 4 <= ex.workspace_bytes;
    if (use_slab) {
        g.slab = (float*)ex.workspace;
        g.atomic_out = 0;
    } else if (want_slab && !accumulate) {       // scratch too small after all: atomics on top of zeros
        for (int gi = 0; gi < ex.n_groups; ++gi) {
            float* Cg = C + gi * ex.c_group_stride;
            hipError_t e = (ldc == N) ? hipMemsetAsync(Cg, 0, (size_t)M * N * sizeof(float), s)
                                      : hipMemset2DAsync(Cg, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
            if (e != hipSuccess) {
                set_error("i3d_gemm_f32_rowsubset: memset failed");
                return I3D_ERR_LAUNCH;
            }
        }
    }
    switch (cfg) {
        case 2: launch_rowseg<Cfg2>(g, t, n_segs, vec, s); break;
        case 3: launch_rowseg<Cfg3>(g, t, n_segs, vec, s); break;
        default: launch_rowseg<Cfg4>(g, t, n_segs, vec, s); break;
    }
    I3D_CHECK_LAUNCH();
    if (use_slab) {
        SlabReduce r;
        r.slab = g.slab; r.C = C; r.bias = nullptr; r.M = M; r.N = N; r.ldc = ldc; r.accumulate = accumulate ? 1 : 0;
        r.n_groups = ex.n_groups; r.c_group_stride = ex.c_group_stride;
        r.c_split = 0x7fffffff; r.c_delta = 0;
        r.post_aff = nullptr; r.post_row = nullptr;
        for (int gi = 0; gi < ex.n_groups; ++gi) r.seg_ptr[gi] = seg_first[gi];
        r.seg_ptr[ex.n_groups] = n_segs;
        launch_slab_reduce(r, s);
        I3D_CHECK_LAUNCH();
    }
    return I3D_OK;
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_set_matmul_precision(int bf16) {
    I3D_CHECK_ARG(bf16 == 0 || bf16 == 1, "0: fp32 MFMA, 1: bf16 MFMA on bf16-rounded operands");
    g_matmul_bf16 = bf16;
    return I3D_OK;
}

extern "C" int i3d_get_matmul_precision(void) { return g_matmul_bf16; }

extern "C" int i3d_set_fp32_products(int split) {
    I3D_CHECK_ARG(split == 0 || split == 1, "0: fp32 MFMA, 1: three-part bf16 split of both operands, six part products");
    g_fp32_split = split;
    return I3D_OK;
}

extern "C" int i3d_get_fp32_products(void) { return g_fp32_split; }

extern "C" int i3d_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, int accumulate, void* stream) {
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, -1, 0, Extra(), stream);
}

extern "C" int i3d_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, int accumulate, int tile_cfg,
                               int splits, void* workspace, long workspace_bytes, void* stream) {
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, tile_cfg, splits, ex, stream);
}

extern "C" int i3d_gemm_f32_ws(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, int accumulate, void* workspace,
                               long workspace_bytes, void* stream) {
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, -1, 0, ex, stream);
}

// i3d_gemm_f32_ws with a two-block B and/or C (include/infomax3d_hip.h)
extern "C" int i3d_gemm_f32_blocks(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                                   int ldb, int b_split, long b_delta, long b_view_floats, float* C, int ldc, int c_split,
                                   long c_delta, int accumulate, void* workspace, long workspace_bytes, void* stream) {
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    if (b_split > 0) { ex.b_split = b_split; ex.b_delta = b_delta; ex.b_view_floats = b_view_floats; }
    if (c_split > 0) { ex.c_split = c_split; ex.c_delta = c_delta; }
    I3D_CHECK_ARG(!(trans_a && trans_b), "layout not supported");
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}

// n_batch products of one shape in ONE launch (include/infomax3d_hip.h)
extern "C" int i3d_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, long a_batch,
                                    const float* B, int ldb, long b_batch, float* C, int ldc, long c_batch, int n_batch,
                                    int accumulate, void* workspace, long workspace_bytes, void* stream) {
    I3D_CHECK_ARG(n_batch >= 1 && !(trans_a && trans_b), "n_batch >= 1, layout not supported");
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    ex.n_batch = n_batch; ex.a_batch = a_batch; ex.b_batch = b_batch; ex.c_batch = c_batch;
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}

// i3d_gemm_f32_grouped with n_batch diagonal blocks per group (include/infomax3d_hip.h)
extern "C" int i3d_gemm_f32_grouped_batched(int trans_b, int m_padded, int N, int K, const float* A, int lda, long a_batch,
                                            long a_rows_total, const int* m_rows, const int* tile_group, const float* B, int ldb,
                                            long b_group_stride, long b_batch, float* C, int ldc, long c_batch, int n_batch,
                                            int accumulate, void* stream) {
    I3D_CHECK_ARG(m_rows != nullptr && tile_group != nullptr && m_padded % 64 == 0 && n_batch >= 1, "grouped GEMM needs 64-padded m_rows");
    Extra ex;
    ex.m_rows = m_rows; ex.tile_group = tile_group; ex.b_group_stride = b_group_stride; ex.a_rows_total = a_rows_total;
    ex.n_batch = n_batch; ex.a_batch = a_batch; ex.b_batch = b_batch; ex.c_batch = c_batch;
    return gemm_impl(0, trans_b, m_padded, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}

// C[m_rows[m], :] (+)= A[m_rows[m], :] * op(B_g),  g = tile_group[m / 64];  m_rows is padded with -1 to 64 per group
extern "C" int i3d_gemm_f32_grouped(int trans_b, int m_padded, int N, int K, const float* A, int lda, long a_rows_total,
                                    const int* m_rows, const int* tile_group, const float* B, int ldb,
                                    long b_group_stride, float* C, int ldc, int accumulate, void* stream) {
    I3D_CHECK_ARG(m_rows != nullptr && tile_group != nullptr && m_padded % 64 == 0, "grouped GEMM needs 64-padded m_rows");
    Extra ex;
    ex.m_rows = m_rows; ex.tile_group = tile_group; ex.b_group_stride = b_group_stride; ex.a_rows_total = a_rows_total;
    return gemm_impl(0, trans_b, m_padded, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}

// C[M,N] = sum_{j < n_rows} A[k_rows[j], 0:M]^T * B[k_rows[j], 0:N]   (weight gradient over a subset of rows)
extern "C" int i3d_gemm_f32_rowsubset(int M, int N, int n_rows, const float* A, int lda, const float* B, int ldb,
                                      const int* k_rows, long rows_total, float* C, int ldc, int accumulate,
                                      void* stream) {
    I3D_CHECK_ARG(k_rows != nullptr && n_rows >= 0, "k_rows required");
    const int start = 0;
    Extra ex;
    ex.k_rows = k_rows; ex.k_rows_total = rows_total;
    ex.n_groups = 1; ex.group_start = &start; ex.group_count = &n_rows; ex.c_group_stride = 0;
    return rowseg_impl(M, N, A, lda, B, ldb, C, ldc, accumulate, -1, 0, ex, stream);
}

// C_g[M,N] = sum_{group_start[g] <= j < group_start[g] + group_count[g]} A[k_rows[j], 0:M]^T B[k_rows[j], 0:N],
// C_g = C + g * c_group_stride: the weight gradients of all in-degree groups in one launch.
extern "C" int i3d_gemm_f32_rowsubset_multi(int M, int N, int n_groups, const int* group_start, const int* group_count,
                                            const float* A, int lda, const float* B, int ldb, const int* k_rows,
                                            long rows_total, float* C, long c_group_stride, int ldc, int accumulate,
                                            int tile_cfg, int seg_rows, void* workspace, long workspace_bytes,
                                            void* stream) {
    I3D_CHECK_ARG(k_rows != nullptr && group_start != nullptr && group_count != nullptr && n_groups > 0, "bad arguments");
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    ex.k_rows = k_rows; ex.k_rows_total = rows_total;
    ex.n_groups = n_groups; ex.group_start = group_start; ex.group_count = group_count; ex.c_group_stride = c_group_stride;
    return rowseg_impl(M, N, A, lda, B, ldb, C, ldc, accumulate, tile_cfg, seg_rows, ex, stream);
}

// C = act((op(A) W^T) + bias (+ C))  with the BatchNorm of the block in front applied to A on the fly (a_aff) and / or the
// column statistics of the stored values produced per 64-row tile (stats): include/infomax3d_hip.h
extern "C" int i3d_gemm_f32_fused(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb,
                                  float* C, int ldc, const float* bias, int accumulate, const float* a_aff, int epi_act,
                                  float* stats, const int* m_rows, const int* tile_group, long b_group_stride,
                                  void* stream) {
    return i3d_gemm_f32_fused_src(M, N, K, A, lda, a_rows_total, W, ldb, C, ldc, nullptr, 0, bias, accumulate, a_aff, epi_act, stats,
                                  m_rows, tile_group, b_group_stride, stream);
}

// i3d_gemm_f32_fused (statistics variant, no accumulate) with C stored as bf16: row r at (bf16*)C + r * ldc
static thread_local int g_fused_c_bf16 = 0;
extern "C" int i3d_gemm_f32_fused_bf16out(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb,
                                          void* C, int ldc, const float* bias, const float* a_aff, int epi_act, float* stats,
                                          void* stream) {
    I3D_CHECK_ARG(stats != nullptr && N % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)C) & 15) == 0, "bf16 output: statistics variant, N and ldc multiples of 4");
    g_fused_c_bf16 = 1;
    const int rc = i3d_gemm_f32_fused_src(M, N, K, A, lda, a_rows_total, W, ldb, (float*)C, ldc, nullptr, 0, bias, 0, a_aff, epi_act, stats,
                                          nullptr, nullptr, 0, stream);
    g_fused_c_bf16 = 0;
    return rc;
}

// i3d_gemm_f32_fused with the addend of `accumulate` read from c_in (row pitch ldcin) instead of C
extern "C" int i3d_gemm_f32_fused_src(int M, int N, int K, const float* A, int lda, long a_rows_total, const float* W, int ldb,
                                      float* C, int ldc, const float* c_in, int ldcin, const float* bias, int accumulate,
                                      const float* a_aff, int epi_act, float* stats, const int* m_rows, const int* tile_group,
                                      long b_group_stride, void* stream) {
    I3D_CHECK_ARG(c_in == nullptr || (stats != nullptr && accumulate && ldcin >= N && ldcin % 4 == 0 && (((uintptr_t)c_in) & 15) == 0),
                  "c_in needs the statistics variant with accumulate, a pitch >= N that is a multiple of 4, 16-byte alignment");
    I3D_CHECK_ARG(M > 0 && N > 0 && K > 0, "empty GEMM");
    I3D_CHECK_ARG(a_aff != nullptr || stats != nullptr, "nothing to fuse: use i3d_gemm_f32");
    I3D_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "leading dimension too small");
    I3D_CHECK_ARG(a_aff == nullptr || K <= FUSE_MAX_K, "BatchNorm prologue: K <= 1024");
    I3D_CHECK_ARG((m_rows == nullptr) == (tile_group == nullptr), "grouped GEMM needs m_rows and tile_group");
    I3D_CHECK_ARG(m_rows == nullptr || M % 64 == 0, "grouped GEMM needs 64-padded m_rows");
    I3D_CHECK_ARG(relu_class(epi_act), "epilogue activation: none, ReLU or LeakyReLU (the others: a pass of their own)");
    const bool al = ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0) && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 &&
                    K % 4 == 0 && b_group_stride % 4 == 0;
    I3D_CHECK_ARG(al, "fused GEMM needs 16-byte aligned operands and K, leading dimensions multiples of 4");
    GemmArgs g = {};
    g.A = A; g.B = W; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate ? 1 : 0;
    g.k_per_split = K; g.atomic_out = 0; g.c_vec = 1;
    g.m_rows = m_rows; g.k_rows = nullptr; g.tile_group = tile_group; g.b_group_stride = b_group_stride;
    g.b_split = g.c_split = 0x7fffffff; g.b_delta = g.c_delta = 0;
    g.slab = nullptr;
    g.a_aff = a_aff; g.stats = stats; g.epi_act = epi_act; g.Cin = c_in; g.ldcin = ldcin; g.c_bf16 = g_fused_c_bf16;
    Extra ex;
    ex.m_rows = m_rows; ex.a_rows_total = a_rows_total;
    int rc = fill_views(g, 0, 1, M, N, K, lda, ldb, ex);
    if (rc != I3D_OK) return rc;
    const int fuse = (a_aff != nullptr ? 1 : 0) | (stats != nullptr ? 2 : 0);
    hipStream_t s = (hipStream_t)stream;
    // tile choice as in gemm_impl for the forward layout: 64x64 (idx-major image), 64x32 when K is long and N pads badly
    const bool narrow = narrow_pays(N, (long)cdiv(M, 64) * cdiv(N, 64)) && (K >= 400 || fuse_narrow_short_k());
    if (g_matmul_bf16 && bf16_bk() == 32) {
        if (narrow) launch_fused<Cfg14>(g, fuse, s);
        else launch_fused<Cfg13>(g, fuse, s);
    } else if (g_matmul_bf16 && bf16_bk() == 64) {
        if (narrow) launch_fused<Cfg16>(g, fuse, s);
        else launch_fused<Cfg15>(g, fuse, s);
    } else if (narrow) launch_fused<Cfg11>(g, fuse, s);
    else launch_fused<Cfg9>(g, fuse, s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// dW[f_out, f_in] = dY^T y  for  y = (x - mean) * scale + shift  (aff = mean | scale | shift over f_in) computed from the raw
// x: the BatchNorm output y is never materialised (fused_bn.hip).  grad_bias[f_out] = column sums of dY.
extern "C" int i3d_gemm_f32_wgrad_bn(int f_out, int f_in, int rows, const float* dY, int ldy, const float* x, int ldx,
                                     float* dW, int ldw, const float* grad_bias, const float* aff, void* workspace,
                                     long workspace_bytes, void* stream) {
    I3D_CHECK_ARG(grad_bias != nullptr && aff != nullptr, "grad_bias and aff required");
    Extra ex;
    ex.workspace = workspace; ex.workspace_bytes = workspace_bytes;
    ex.post_aff = aff; ex.post_row = grad_bias;
    return gemm_impl(1, 0, f_out, f_in, rows, dY, ldy, x, ldx, dW, ldw, nullptr, 0, -1, 0, ex, stream);
}
