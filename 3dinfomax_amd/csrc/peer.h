// One-shot peer-write exchange for the small vectors of synchronised BatchNorm (provider 3 of comm.hip's collective table).
//
// The reference normalises over the whole batch it is given (models/base_layers.py:87, 100-111); with molecules sharded
// over ranks every BatchNorm needs a [3F]-float or [2F+1]-double vector from every rank, ~50 times per step, on the step's
// dependent chain.  A ring collective of a library is latency-bound there (RCCL at world 1: 11-17 us per call, several
// times that over 8 GPUs).  Here every rank owns a MAILBOX in device memory that every other rank has mapped
// (hipIpcGetMemHandle / hipIpcOpenMemHandle once, at set-up): a kernel writes its vector and then a sequence flag straight
// into every peer's mailbox over xGMI, waits for the `world` flags in its own mailbox and reads the vectors there - no
// communicator, no extra launch when the exchange sits inside the kernel that produces / consumes the vector
// (bn_finalize_partials_kernel<true>), ~2 us next to an empty kernel otherwise (profiles/r04_peer_probe.txt).
//
// Protocol.  Collectives are numbered by a host-side counter `seq` (every rank issues the same sequence of BatchNorm
// calls on ONE stream, as with RCCL).  Collective `seq` uses slot seq % PEER_SLOTS.  Writer (rank r, workgroup w): payload
// to box[p]->pay[slot][r] for every p (system-scope stores), wait until they are performed, barrier, then flag
// box[p]->flag[slot][r][w] = seq.  Reader: spin on box[self]->flag[slot][q][w] == seq for every q, barrier, read
// box[self]->pay[slot][q] with system-scope loads.
// A slot is reused PEER_SLOTS collectives later: a rank can only be there once every peer has raised its flag for the
// collective in between, which a peer does after its kernel of the earlier collective has finished (same stream) -
// two slots would do, four are used.  The wait is bounded (PeerDev::timeout ticks of the 100 MHz wall clock): on expiry
// the kernel records `seq` in a host-visible status word and in its own mailbox (later kernels do not wait again) and the
// next enqueue returns an error - a lost rank costs a timeout, not a hung GPU.
// Sums are formed in rank order by every rank: bit-identical results on all ranks, deterministic.
#pragma once
#include "common.h"

namespace i3d {

constexpr int PEER_MAX_WORLD = 16;
constexpr int PEER_SLOTS = 4;
constexpr int PEER_MAX_WG = 512;                    // workgroups of one exchanging launch
constexpr int PEER_PAYLOAD_BYTES = 64 << 10;        // per (slot, rank): [3F] floats / [2F+1] doubles up to F = 4095

struct PeerBox {
    unsigned long long flag[PEER_SLOTS][PEER_MAX_WORLD][PEER_MAX_WG];
    unsigned long long dead;                        // != 0: a wait of this rank timed out (sequence number)
    unsigned long long pad_[15];
    char pay[PEER_SLOTS][PEER_MAX_WORLD][PEER_PAYLOAD_BYTES];
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
__device__ __forceinline__ char* peer_send_slot(const PeerDev& d, int p) {          // where my payload goes in peer p
    return d.box[p]->pay[(int)(d.seq % PEER_SLOTS)][d.rank];
}
__device__ __forceinline__ const char* peer_recv_slot(const PeerDev& d, int q) {    // where rank q's payload arrives here
    return d.box[d.rank]->pay[(int)(d.seq % PEER_SLOTS)][q];
}
__device__ __forceinline__ void peer_put_f32(const PeerDev& d, int p, int i, float v) {
    __hip_atomic_store((float*)peer_send_slot(d, p) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_put_f64(const PeerDev& d, int p, int i, double v) {
    __hip_atomic_store((double*)peer_send_slot(d, p) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float peer_get_f32(const PeerDev& d, int q, int i) {
    return __hip_atomic_load((const float*)peer_recv_slot(d, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double peer_get_f64(const PeerDev& d, int q, int i) {
    return __hip_atomic_load((const double*)peer_recv_slot(d, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Called by EVERY thread of the workgroup after its peer_put_* calls: publishes them to every rank, then waits until every
// rank's payload of this workgroup index has arrived here.  `wg` < PEER_MAX_WG identifies the workgroup's share of the
// payload (the same partition on every rank).  blockDim.x >= world.
__device__ __forceinline__ void peer_signal_and_wait(const PeerDev& d, int wg) {
    const int slot = (int)(d.seq % PEER_SLOTS);
    // Ordering without cache maintenance: payload and flags are system-scope atomic accesses to UNCACHED memory (write-
    // through stores, loads that miss everywhere), so all that is needed is that this workgroup's payload stores have been
    // PERFORMED before its flag stores are issued - s_waitcnt vmcnt(0) per wave, then the barrier.  (A system-scope release
    // fence here is an L2 write-back of everything the step has dirtied, an acquire fence an L2 invalidate for the kernels
    // that follow: measured 3.7 us per exchange inside the step against 1.8 us in an idle probe.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = threadIdx.x + threadIdx.y * blockDim.x;
    if (t < d.world) {
        __hip_atomic_store(&d.box[t]->flag[slot][d.rank][wg], d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* f = &d.box[d.rank]->flag[slot][t][wg];
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != d.seq) {
            const bool dead = __hip_atomic_load(&d.box[d.rank]->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            const long long t0 = wall_clock64();
            while (!dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != d.seq) {
                if (wall_clock64() - t0 > d.timeout) {
                    __hip_atomic_store(&d.box[d.rank]->dead, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(d.status, (int)d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    // the payload is read with peer_get_* only (system-scope loads: never served from the vector L1 or the L2)
    __syncthreads();
    asm volatile("" ::: "memory");
}
#endif

}  // namespace i3d
