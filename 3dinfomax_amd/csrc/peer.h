// One-shot peer-write exchange for the small vectors of synchronised BatchNorm (provider 3 of comm.hip's collective table).
//
// The reference normalises over the whole batch it is given (models/base_layers.py:87, 100-111); with molecules sharded
// over ranks every BatchNorm needs a [3F]-float or [2F+1]-double vector from every rank, ~50 times per step, on the step's
// dependent chain.  A ring collective of a library is latency-bound there (RCCL at world 1: 11-17 us per call, several
// times that over 8 GPUs).  Here every rank owns a MAILBOX in device memory that every other rank has mapped
// (hipIpcGetMemHandle / hipIpcOpenMemHandle once, at set-up): a kernel writes its vector straight into every peer's
// mailbox over xGMI and reads the `world` vectors that arrive in its own (each word says which collective it belongs to) - no
// communicator, no extra launch when the exchange sits inside the kernel that produces / consumes the vector
// (bn_finalize_partials_kernel<true>), ~2 us next to an empty kernel otherwise (profiles/r04_peer_probe.txt).
//
// Protocol.  Collectives are numbered by a host-side counter `seq` (every rank issues the same sequence of BatchNorm
// calls on ONE stream, as with RCCL).  Collective `seq` uses slot seq % PEER_SLOTS.  Every value travels as ONE 8-byte word
// {low: the value's 32 bits, high: the low 32 bits of seq} written with ONE system-scope store (a double as two such words):
// rank r writes word i of its payload to box[p]->pay[slot][r][i] for every p, the reader spins on box[self]->pay[slot][q][i]
// until the word carries this collective's tag and has the value with it.  No flag, no ordering between stores over the
// fabric is assumed (the first form of this round - payload stores, s_waitcnt, then a flag store per workgroup - relied on
// two stores of one wave to one peer arriving in order; 8 bytes are written atomically whatever the route), and the value is
// there one fabric traversal after it was produced instead of two.
// A slot is reused PEER_SLOTS collectives later: a rank can only be there once it has READ every peer's words of the
// collectives in between, which a peer writes from the kernel of that collective - after its kernel of the earlier one has
// finished (same stream): two slots would do, four are used.  Every wait is bounded (PeerDev::timeout ticks of the 100 MHz wall
// clock): on expiry the thread records `seq` in a host-visible status word and in its own mailbox (later waits return at
// once) and the next enqueue returns an error - a lost rank costs a timeout, not a hung GPU.
// Sums are formed in rank order by every rank: bit-identical results on all ranks, deterministic.
// The tag is the low 32 bits of seq; sequence numbers with a zero tag are skipped (a zero-initialised, never-written word must
// not validate; the status word's 0 means healthy).  A word that still carries the tag of 2^32 collectives ago would need to
// have gone unwritten for that long: every word a collective reads is written by the same collective of the step before.
#pragma once
#include "common.h"

namespace i3d {

constexpr int PEER_MAX_WORLD = 16;
constexpr int PEER_SLOTS = 4;
constexpr int PEER_PAYLOAD_WORDS = 16 << 10;        // per (slot, rank): [3F] floats up to F = 5461, [3F] doubles up to F = 2730

struct PeerBox {
    unsigned long long dead;                        // != 0: a wait of this rank timed out (sequence number)
    unsigned long long pad_[15];
    unsigned long long pay[PEER_SLOTS][PEER_MAX_WORLD][PEER_PAYLOAD_WORDS];      // {value bits, tag} words
};

struct PeerDev {
    int world, rank;
    unsigned long long seq;
    long long timeout;                              // ticks of wall_clock64()
    int* status;                                    // host-visible
    PeerBox* box[PEER_MAX_WORLD];
};

// The peer provider's state for collectives issued on `stream` while it is the provider set in the collective table
// (comm.hip), else null.  A context is one mailbox + one sequence counter + one scratch, and its collectives must all be
// issued on ONE stream (slot reuse relies on stream order): the default context serves every stream that has no context of
// its own, i3d_peer_bind_stream gives a second stream (the 3D network's, which runs beside the 2D network) its own.
struct PeerCtx;
PeerCtx* peer_active(void* stream);
void* peer_scratch(PeerCtx* ctx);
long peer_scratch_bytes(PeerCtx* ctx);
void peer_deactivate();      // comm.hip: another table (or none) was set
// next collective of the sequence: fills `d` (seq advanced) or returns an error when an earlier wait has timed out
int peer_next(PeerCtx* ctx, PeerDev* d);
// fused exchanges used by the BatchNorm entry points (peer.hip); `in` has n values, `rows` is appended as value n
// when append_rows; the sums over the ranks (rank order, fp64) go to out64[n(+1)] / outf[n] (either may be null),
// 1 / (last summed value) to inv_last
int peer_sum_f64(PeerCtx* ctx, const double* in, int n, int append_rows, double rows, double* out64, float* outf, float* inv_last,
                 void* stream);
int peer_sum_f32(PeerCtx* ctx, const float* in, int n, int append_rows, double rows, double* out64, float* outf, float* inv_last,
                 void* stream);

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long* peer_send_slot(const PeerDev& d, int p) {          // where my words go in peer p
    return d.box[p]->pay[(int)(d.seq % PEER_SLOTS)][d.rank];
}
__device__ __forceinline__ const unsigned long long* peer_recv_slot(const PeerDev& d, int q) {    // where rank q's words arrive here
    return d.box[d.rank]->pay[(int)(d.seq % PEER_SLOTS)][q];
}
__device__ __forceinline__ void peer_put_word(const PeerDev& d, int p, int w, unsigned bits) {
    __hip_atomic_store(peer_send_slot(d, p) + w, ((unsigned long long)(unsigned)d.seq << 32) | bits, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// words w[0..K) of rank q's payload of THIS collective: all K loads in flight, repeated (bounded) until every word carries the tag
template <int K>
__device__ __forceinline__ void peer_get_words(const PeerDev& d, int q, const int (&w)[K], unsigned (&out)[K]) {
    const unsigned long long* a = peer_recv_slot(d, q);
    const unsigned tag = (unsigned)d.seq;
    unsigned long long v[K];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = __hip_atomic_load(a + w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int k = 0; k < K; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
    if (!ok && __hip_atomic_load(&d.box[d.rank]->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        const long long t0 = wall_clock64();
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = __hip_atomic_load(a + w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = true;
#pragma unroll
            for (int k = 0; k < K; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
            if (ok) break;
            if (wall_clock64() - t0 > d.timeout) {
                __hip_atomic_store(&d.box[d.rank]->dead, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(d.status, (int)d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = (unsigned)v[k];
}
// word w of the payloads of ranks q0 .. q0 + Q - 1 (clamped to the last rank) of THIS collective: the Q loads in flight together
template <int Q>
__device__ __forceinline__ void peer_get_word_ranks(const PeerDev& d, int q0, int w, unsigned (&out)[Q]) {
    const unsigned tag = (unsigned)d.seq;
    const unsigned long long* a[Q];
    unsigned long long v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) a[k] = peer_recv_slot(d, min(q0 + k, d.world - 1)) + w;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < Q; ++k) v[k] = __hip_atomic_load(a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int k = 0; k < Q; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
    if (!ok && __hip_atomic_load(&d.box[d.rank]->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        const long long t0 = wall_clock64();
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int k = 0; k < Q; ++k) v[k] = __hip_atomic_load(a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = true;
#pragma unroll
            for (int k = 0; k < Q; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
            if (ok) break;
            if (wall_clock64() - t0 > d.timeout) {
                __hip_atomic_store(&d.box[d.rank]->dead, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(d.status, (int)d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) out[k] = (unsigned)v[k];
}
// element i of a float payload is word i, of a double payload words 2 i (low half) and 2 i + 1
__device__ __forceinline__ void peer_put_f32(const PeerDev& d, int p, int i, float v) { peer_put_word(d, p, i, __float_as_uint(v)); }
__device__ __forceinline__ void peer_put_f64(const PeerDev& d, int p, int i, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    peer_put_word(d, p, 2 * i, (unsigned)b);
    peer_put_word(d, p, 2 * i + 1, (unsigned)(b >> 32));
}
__device__ __forceinline__ float peer_get_f32(const PeerDev& d, int q, int i) {
    const int w[1] = {i};
    unsigned o[1];
    peer_get_words<1>(d, q, w, o);
    return __uint_as_float(o[0]);
}
__device__ __forceinline__ double peer_words_f64(unsigned lo, unsigned hi) {
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double peer_get_f64(const PeerDev& d, int q, int i) {
    const int w[2] = {2 * i, 2 * i + 1};
    unsigned o[2];
    peer_get_words<2>(d, q, w, o);
    return peer_words_f64(o[0], o[1]);
}
// the three values i0, i1, i2 of rank q's payload with one wait
__device__ __forceinline__ void peer_get3_f32(const PeerDev& d, int q, int i0, int i1, int i2, float (&out)[3]) {
    const int w[3] = {i0, i1, i2};
    unsigned o[3];
    peer_get_words<3>(d, q, w, o);
    out[0] = __uint_as_float(o[0]); out[1] = __uint_as_float(o[1]); out[2] = __uint_as_float(o[2]);
}
__device__ __forceinline__ void peer_get3_f64(const PeerDev& d, int q, int i0, int i1, int i2, double (&out)[3]) {
    const int w[6] = {2 * i0, 2 * i0 + 1, 2 * i1, 2 * i1 + 1, 2 * i2, 2 * i2 + 1};
    unsigned o[6];
    peer_get_words<6>(d, q, w, o);
    out[0] = peer_words_f64(o[0], o[1]); out[1] = peer_words_f64(o[2], o[3]); out[2] = peer_words_f64(o[4], o[5]);
}
#endif

}  // namespace i3d
