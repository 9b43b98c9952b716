// Peer-write provider of the process-wide collectives (peer.h: protocol; comm.hip: the table the BatchNorm entry points
// consult).  Host side: mailbox allocation + IPC handle, opening the peers' mailboxes, the sequence counter, the generic
// all-gather / all-reduce kernels behind the table's function pointers and the fused "sum over ranks" exchange the
// BatchNorm backward, the activation statistics and the 3D network's edge stage use (one launch instead of
// convert -> collective -> convert).  Replaces nothing of the reference (it has no multi-GPU path); it is what
// makes the reference's whole-batch BatchNorm semantics (models/base_layers.py:87, 100-111) affordable over ranks.
#include "peer.h"

#include <cstdlib>
#include <cstring>

namespace i3d {

struct PeerCtx {
    PeerDev dev;
    PeerBox* own = nullptr;
    void* opened[PEER_MAX_WORLD] = {};
    int* status_host = nullptr;
    void* scratch = nullptr;
    long scratch_bytes = 0;
    void* stream_key = nullptr;      // the stream this context serves (bound contexts)
    bool owns_box = false;
};

namespace {

constexpr int MAX_BOUND = 4;
PeerCtx* g_active = nullptr;            // default context (any stream without one of its own)
PeerCtx* g_bound[MAX_BOUND] = {};      // contexts bound to a stream

// generic all-gather: recv[world][count] <- every rank's send[count]; a thread moves one element to every rank and fetches
// the `world` elements of its index that arrive here
__global__ void __launch_bounds__(256) peer_all_gather_f32_kernel(const PeerDev d, const float* __restrict__ send,
                                                                  float* __restrict__ recv, int count) {
    I3D_CHAIN_PRIO();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float v = send[i];
    for (int p = 0; p < d.world; ++p) peer_put_f32(d, p, i, v);
    // four ranks per round: their words in flight together (a rank after the other is one memory latency per rank)
    for (int q0 = 0; q0 < d.world; q0 += 4) {
        unsigned r[4];
        peer_get_word_ranks<4>(d, q0, i, r);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q0 + k < d.world) recv[(long)(q0 + k) * count + i] = __uint_as_float(r[k]);
    }
}

// sum over the ranks in rank order (fp64): 16 threads per element - thread (e, q) fetches rank q's value of element e, the
// element's first thread adds them in rank order; element n = `rows` when append_rows.  Outputs: out64[n_total] and / or
// outf[n] (the appended element is not written to outf), inv_last = 1 / sum of the last element.
template <typename In>
__global__ void __launch_bounds__(256) peer_sum_kernel(const PeerDev d, const In* in, int n, int append_rows,
                                                       double rows, double* out64, float* __restrict__ outf,
                                                       float* __restrict__ inv_last) {      // (in == out64: in place)
    I3D_CHAIN_PRIO();
    __shared__ double sv[256 / PEER_MAX_WORLD][PEER_MAX_WORLD];
    const int total = n + (append_rows ? 1 : 0);
    const int el = threadIdx.x / PEER_MAX_WORLD, q = threadIdx.x % PEER_MAX_WORLD;
    const int i = blockIdx.x * (256 / PEER_MAX_WORLD) + el;
    if (i < total && q < d.world) {
        const double v = i < n ? (double)in[i] : rows;
        peer_put_f64(d, q, i, v);              // (thread q of the element writes to rank q)
        sv[el][q] = peer_get_f64(d, q, i);
    }
    __syncthreads();
    if (i >= total || q != 0) return;
    double s = 0.0;
    for (int r = 0; r < d.world; ++r) s += sv[el][r];
    if (out64 != nullptr) out64[i] = s;
    if (outf != nullptr && i < n) outf[i] = (float)s;
    if (inv_last != nullptr && i == total - 1) inv_last[0] = (float)(1.0 / s);
}

int table_all_gather_f32(void* user, const float* send, float* recv, long count, void* stream) {
    (void)user;
    PeerCtx* ctx = peer_active(stream);
    I3D_CHECK_ARG(ctx != nullptr, "peer provider not installed");
    I3D_CHECK_ARG(count > 0 && count <= PEER_PAYLOAD_WORDS, "payload too large for the mailbox");
    PeerDev d;
    if (int rc = peer_next(ctx, &d)) return rc;
    hipLaunchKernelGGL(peer_all_gather_f32_kernel, dim3(cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, d, send, recv, (int)count);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

int table_all_reduce_f64(void* user, double* buf, long count, void* stream) {
    (void)user;
    return peer_sum_f64(peer_active(stream), buf, (int)count, 0, 0.0, buf, nullptr, nullptr, stream);
}

template <typename In>
int peer_sum_impl(PeerCtx* ctx, const In* in, int n, int append_rows, double rows, double* out64, float* outf, float* inv_last,
                  void* stream) {
    const int total = n + (append_rows ? 1 : 0);
    I3D_CHECK_ARG(ctx != nullptr && in != nullptr && n > 0 && 2L * total <= PEER_PAYLOAD_WORDS,
                  "bad arguments / payload too large for the mailbox");
    PeerDev d;
    if (int rc = peer_next(ctx, &d)) return rc;
    hipLaunchKernelGGL(peer_sum_kernel<In>, dim3(cdiv(total, 256 / PEER_MAX_WORLD)), dim3(256), 0, (hipStream_t)stream, d, in, n, append_rows, rows,
                       out64, outf, inv_last);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

}  // namespace

PeerCtx* peer_active(void* stream) {
    if (g_active == nullptr) return nullptr;
    for (int i = 0; i < MAX_BOUND; ++i)
        if (g_bound[i] != nullptr && g_bound[i]->stream_key == stream) return g_bound[i];
    return g_active;
}

void* peer_scratch(PeerCtx* ctx) { return ctx->scratch; }
long peer_scratch_bytes(PeerCtx* ctx) { return ctx->scratch_bytes; }

void peer_deactivate() {
    g_active = nullptr;
    for (int i = 0; i < MAX_BOUND; ++i) g_bound[i] = nullptr;
}

int peer_next(PeerCtx* ctx, PeerDev* d) {
    if (ctx->status_host != nullptr && *(volatile int*)ctx->status_host != 0) {
        set_error("peer exchange: a wait for the other ranks timed out at collective %d (a rank is missing or not issuing the "
                  "same sequence of BatchNorm calls)", *(volatile int*)ctx->status_host);
        return I3D_ERR_LAUNCH;
    }
    ctx->dev.seq += 1;
    // tag 0 is what a never-written word of the zero-initialised mailbox carries, and the status words store (int)seq with 0 =
    // healthy: sequence numbers whose low 32 bits are 0 are skipped (every rank skips the same ones; the slot pattern shifts by
    // one, a slot is then reused three collectives later instead of four - two would do, peer.h)
    if ((unsigned)ctx->dev.seq == 0u) ctx->dev.seq += 1;
    *d = ctx->dev;
    return I3D_OK;
}

int peer_sum_f64(PeerCtx* ctx, const double* in, int n, int append_rows, double rows, double* out64, float* outf, float* inv_last,
                 void* stream) {
    return peer_sum_impl<double>(ctx, in, n, append_rows, rows, out64, outf, inv_last, stream);
}

int peer_sum_f32(PeerCtx* ctx, const float* in, int n, int append_rows, double rows, double* out64, float* outf, float* inv_last,
                 void* stream) {
    return peer_sum_impl<float>(ctx, in, n, append_rows, rows, out64, outf, inv_last, stream);
}

}  // namespace i3d

using namespace i3d;

#define HIP_TRY(call, what)                                                        \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            i3d::set_error("%s: %s failed: %s", __func__, what, hipGetErrorString(e_)); \
            return I3D_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

extern "C" long i3d_peer_mailbox_bytes(void) { return (long)sizeof(PeerBox); }

extern "C" int i3d_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// Allocate this rank's mailbox on the current device (uncached device memory: peers write it over xGMI and this rank
// polls it - no L2 line may go stale; fine-grained as the fallback) and export it.
extern "C" int i3d_peer_alloc(void** mailbox, char* handle_out) {
    I3D_CHECK_ARG(mailbox != nullptr && handle_out != nullptr, "null");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, sizeof(PeerBox), hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, sizeof(PeerBox), hipDeviceMallocFinegrained);
    }
    HIP_TRY(e, "hipExtMallocWithFlags (uncached / fine-grained device memory)");
    HIP_TRY(hipMemset(p, 0, sizeof(PeerBox)), "hipMemset");
    HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        HIP_TRY(e, "hipIpcGetMemHandle");
    }
    std::memcpy(handle_out, &h, sizeof(h));
    *mailbox = p;
    return I3D_OK;
}

// a mailbox that never made it into a context (set-up abandoned because another rank failed)
extern "C" int i3d_peer_free(void* mailbox) {
    if (mailbox != nullptr) (void)hipFree(mailbox);
    return I3D_OK;
}

// Open every peer's mailbox (handles[world][i3d_peer_handle_bytes()], rank order; this rank's own entry is not opened).
extern "C" int i3d_peer_open(void* mailbox, const char* handles, int rank, int world, double timeout_s, void** ctx_out) {
    I3D_CHECK_ARG(mailbox != nullptr && ctx_out != nullptr && world >= 1 && world <= PEER_MAX_WORLD && rank >= 0 && rank < world,
                  "1 <= world <= 16, 0 <= rank < world");
    I3D_CHECK_ARG(world == 1 || handles != nullptr, "handles required");
    PeerCtx* ctx = new PeerCtx();
    ctx->own = (PeerBox*)mailbox;
    ctx->owns_box = true;
    ctx->dev.world = world; ctx->dev.rank = rank; ctx->dev.seq = 0;
    // I3D_PEER_SEQ0 (tests: the same value on every rank): start the sequence there - a soak run then crosses the wrap of
    // the 32-bit tag after minutes instead of a day
    if (const char* t = getenv("I3D_TESTING"); t != nullptr && t[0] == '1')
        if (const char* s0 = getenv("I3D_PEER_SEQ0")) ctx->dev.seq = strtoull(s0, nullptr, 0);
    const char* env = getenv("I3D_PEER_TIMEOUT_S");
    if (env != nullptr && atof(env) > 0) timeout_s = atof(env);
    if (timeout_s <= 0) timeout_s = 30.0;
    ctx->dev.timeout = (long long)(timeout_s * 1e8);          // wall_clock64: 100 MHz
    for (int q = 0; q < PEER_MAX_WORLD; ++q) ctx->dev.box[q] = ctx->own;
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (long)q * sizeof(h), sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < q; ++k)
                if (ctx->opened[k] != nullptr) (void)hipIpcCloseMemHandle(ctx->opened[k]);
            delete ctx;
            HIP_TRY(e, "hipIpcOpenMemHandle");
        }
        ctx->opened[q] = p;
        ctx->dev.box[q] = (PeerBox*)p;
    }
    auto undo = [&]() {        // nothing stays mapped or allocated behind a failed open (the mailbox stays the caller's)
        for (int k = 0; k < world; ++k)
            if (ctx->opened[k] != nullptr) (void)hipIpcCloseMemHandle(ctx->opened[k]);
        if (ctx->status_host != nullptr) (void)hipHostFree(ctx->status_host);
        delete ctx;
    };
    void* st = nullptr;
    hipError_t e = hipHostMalloc(&st, 64, hipHostMallocMapped);
    if (e != hipSuccess) {
        undo();
        HIP_TRY(e, "hipHostMalloc");
    }
    std::memset(st, 0, 64);
    ctx->status_host = (int*)st;
    void* st_dev = nullptr;
    e = hipHostGetDevicePointer(&st_dev, st, 0);
    if (e != hipSuccess) {
        undo();
        HIP_TRY(e, "hipHostGetDevicePointer");
    }
    ctx->dev.status = (int*)st_dev;
    *ctx_out = ctx;
    return I3D_OK;
}

// 0: healthy; otherwise the sequence number of the collective whose wait timed out
extern "C" int i3d_peer_status(void* ctx_) {
    PeerCtx* ctx = (PeerCtx*)ctx_;
    return (ctx != nullptr && ctx->status_host != nullptr) ? *(volatile int*)ctx->status_host : 0;
}

extern "C" long long i3d_peer_sequence(void* ctx_) { return ctx_ != nullptr ? (long long)((PeerCtx*)ctx_)->dev.seq : 0; }

extern "C" int i3d_set_collectives_peer(void* ctx_, void* scratch, long scratch_bytes) {
    PeerCtx* ctx = (PeerCtx*)ctx_;
    I3D_CHECK_ARG(ctx != nullptr, "no context");
    I3dCollectives c;
    c.world = ctx->dev.world; c.all_gather_f32 = table_all_gather_f32; c.all_reduce_f64 = table_all_reduce_f64; c.user = ctx;
    c.scratch = scratch; c.scratch_bytes = scratch_bytes;
    I3D_CHECK_ARG(scratch != nullptr && scratch_bytes >= 4096, "device scratch required");
    const int rc = i3d_set_collectives(&c);
    if (rc == I3D_OK) {
        ctx->scratch = scratch; ctx->scratch_bytes = scratch_bytes;
        g_active = ctx;
    }
    return rc;
}

// A context of its own (own mailbox, own sequence, own scratch) for the collectives issued on `stream` - the 3D network's
// side stream: with it the 3D network keeps running beside the 2D network under synchronised BatchNorm.  Every rank binds
// the same streams in the same roles.  After i3d_set_collectives_peer.
extern "C" int i3d_peer_bind_stream(void* ctx_, void* stream, void* scratch, long scratch_bytes) {
    PeerCtx* ctx = (PeerCtx*)ctx_;
    I3D_CHECK_ARG(ctx != nullptr && g_active != nullptr && ctx != g_active, "install the default context first; a context of its own");
    I3D_CHECK_ARG(scratch != nullptr && scratch_bytes >= 4096 && (((uintptr_t)scratch) & 15) == 0, "device scratch required");
    for (int i = 0; i < MAX_BOUND; ++i) {
        if (g_bound[i] == nullptr || g_bound[i] == ctx) {
            ctx->stream_key = stream; ctx->scratch = scratch; ctx->scratch_bytes = scratch_bytes;
            g_bound[i] = ctx;
            return I3D_OK;
        }
    }
    I3D_CHECK_ARG(false, "too many bound streams");
}

// Close the peers' mappings, free the mailbox (every rank must have stopped issuing collectives: the caller synchronises
// its stream and runs a barrier first).
extern "C" int i3d_peer_close(void* ctx_) {
    PeerCtx* ctx = (PeerCtx*)ctx_;
    if (ctx == nullptr) return I3D_OK;
    if (g_active == ctx) i3d_set_collectives(nullptr);      // (also drops the bound contexts: they are closed by their owner)
    for (int i = 0; i < MAX_BOUND; ++i)
        if (g_bound[i] == ctx) g_bound[i] = nullptr;
    for (int q = 0; q < PEER_MAX_WORLD; ++q)
        if (ctx->opened[q] != nullptr) (void)hipIpcCloseMemHandle(ctx->opened[q]);
    if (ctx->status_host != nullptr) (void)hipHostFree(ctx->status_host);
    if (ctx->owns_box && ctx->own != nullptr) (void)hipFree(ctx->own);
    delete ctx;
    return I3D_OK;
}
