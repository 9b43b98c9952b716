// Synchronised BatchNorm from inside the C sequencers: process-wide ("ambient") collectives.
//
// The reference normalises over the whole batch it is given (models/base_layers.py:87, 100-111: nn.BatchNorm1d in
// training mode); with the batch sharded by molecule over ranks only BatchNorm statistics over ALL ranks reproduce its
// loss.  Round 2 could do that only on the per-block Python path (every BatchNorm = statistics kernel ->
// torch.distributed all-reduce -> finalisation, 1.95x the step): the fused-BatchNorm layer and the whole-model
// sequencer produce a block's statistics in one kernel's epilogue and consume them in the next kernel's loads, inside
// ONE C call.  Here the three entry points that finalise BatchNorm statistics (i3d_bn_finalize_partials,
// i3d_act_stats_fwd*, i3d_bn_bwd*) look at a process-wide collective table: when one is set they merge their LOCAL result
// into a small vector, run the collective on the caller's stream (one more enqueue in the sequence - no host round
// trip, no event) and finalise over the ranks.  Every sequencer above them - block composites, PNA layer, whole model,
// the 3D network - becomes synchronised without knowing it.
//
// Three providers (the third, peer.h / peer.hip, is the one meant for production: a one-shot peer-write exchange over IPC-
// mapped mailboxes, inside the BatchNorm kernels themselves where it can be):
//   * RCCL, bound at run time (dlopen: the library has no link-time dependency on it) through a communicator of the
//     library's own, created from an id the caller distributes (i3d_rccl_unique_id on rank 0 -> every rank
//     i3d_rccl_init): ncclAllGather / ncclAllReduce enqueued on the caller's stream;
//   * caller-supplied callbacks (tests: two ranks sharing one GPU over gloo, where RCCL refuses duplicate devices).
// One training per process, one stream issuing collectives (the 3D network's side stream is switched off by dist.setup
// when this is on): RCCL needs every rank to issue the collectives of a communicator in the same order.
#include "common.h"
#include "peer.h"

#include <dlfcn.h>

#include <mutex>

namespace i3d {

namespace {

I3dCollectives g_coll = {0, nullptr, nullptr, nullptr, nullptr, 0};

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------
typedef struct { char internal[128]; } NcclUniqueId;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(void**, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
};

constexpr int NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0;

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: the copy torch has already mapped (its own librccl.so), else the ROCm one
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (r.lib != nullptr) break;
        }
        for (int i = 0; r.lib == nullptr && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (r.lib == nullptr) return;
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
        r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
        r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
        r.all_reduce = (AllReduceFn)dlsym(r.lib, "ncclAllReduce");
        r.error_string = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
    });
    return (r.get_unique_id && r.comm_init_rank && r.all_gather && r.all_reduce) ? &r : nullptr;
}

int rccl_fail(const char* what, int rc) {
    Rccl* r = rccl();
    set_error("%s failed: %s", what, (r && r->error_string) ? r->error_string(rc) : "RCCL error");
    return I3D_ERR_LAUNCH;
}

int rccl_all_gather_f32(void* user, const float* send, float* recv, long count, void* stream) {
    const int rc = rccl()->all_gather(send, recv, (size_t)count, NCCL_FLOAT32, user, (hipStream_t)stream);
    return rc == 0 ? I3D_OK : rccl_fail("ncclAllGather", rc);
}

int rccl_all_reduce_f64(void* user, double* buf, long count, void* stream) {
    const int rc = rccl()->all_reduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, user, (hipStream_t)stream);
    return rc == 0 ? I3D_OK : rccl_fail("ncclAllReduce", rc);
}

}  // namespace

const I3dCollectives* collectives() { return g_coll.world > 0 && g_coll.all_gather_f32 != nullptr ? &g_coll : nullptr; }

}  // namespace i3d

using namespace i3d;

#define I3D_CHECK_HIP(call)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            i3d::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_));     \
            return I3D_ERR_LAUNCH;                                                           \
        }                                                                                    \
    } while (0)

extern "C" int i3d_set_collectives(const I3dCollectives* c) {
    peer_deactivate();      // the fused peer-exchange paths belong to i3d_set_collectives_peer, which re-arms them after this call
    if (c == nullptr) {
        g_coll = I3dCollectives{0, nullptr, nullptr, nullptr, nullptr, 0};
        return I3D_OK;
    }
    I3D_CHECK_ARG(c->world >= 1 && c->all_gather_f32 != nullptr && c->all_reduce_f64 != nullptr, "world >= 1 and both collectives");
    I3D_CHECK_ARG(c->scratch != nullptr && c->scratch_bytes >= 4096 && (((uintptr_t)c->scratch) & 15) == 0, "device scratch required");
    g_coll = *c;
    return I3D_OK;
}

extern "C" int i3d_collectives_world(void) { return collectives() != nullptr ? g_coll.world : 0; }

// The installed provider's two collectives, callable by the host (a binding's self-test; tests/test_gpu_dist.py): exactly what the
// BatchNorm entry points call when no fused exchange applies - the vectors staged in the table's scratch, as they stage theirs (a
// host-staged provider may only know that buffer).  Every rank must make the same calls in the same order.
extern "C" int i3d_collectives_all_gather_f32(const float* send, float* recv, long count, void* stream) {
    const I3dCollectives* c = collectives();
    I3D_CHECK_ARG(c != nullptr, "no collectives installed");
    I3D_CHECK_ARG(send != nullptr && recv != nullptr && count > 0, "bad arguments");
    const long cnt4 = (count + 3) & ~3L;
    I3D_CHECK_ARG((cnt4 + (long)c->world * count) * 4 <= c->scratch_bytes, "vector too long for the provider's scratch");
    float* s_send = (float*)c->scratch;
    float* s_recv = s_send + cnt4;
    hipStream_t st = (hipStream_t)stream;
    I3D_CHECK_HIP(hipMemcpyAsync(s_send, send, (size_t)count * 4, hipMemcpyDeviceToDevice, st));
    const int rc = c->all_gather_f32(c->user, s_send, s_recv, count, stream);
    if (rc != I3D_OK) return rc;
    I3D_CHECK_HIP(hipMemcpyAsync(recv, s_recv, (size_t)c->world * count * 4, hipMemcpyDeviceToDevice, st));
    return I3D_OK;
}

extern "C" int i3d_collectives_all_reduce_f64(double* buf, long count, void* stream) {
    const I3dCollectives* c = collectives();
    I3D_CHECK_ARG(c != nullptr, "no collectives installed");
    I3D_CHECK_ARG(buf != nullptr && count > 0, "bad arguments");
    I3D_CHECK_ARG(count * 8 <= c->scratch_bytes, "vector too long for the provider's scratch");
    double* s_buf = (double*)c->scratch;
    hipStream_t st = (hipStream_t)stream;
    I3D_CHECK_HIP(hipMemcpyAsync(s_buf, buf, (size_t)count * 8, hipMemcpyDeviceToDevice, st));
    const int rc = c->all_reduce_f64(c->user, s_buf, count, stream);
    if (rc != I3D_OK) return rc;
    I3D_CHECK_HIP(hipMemcpyAsync(buf, s_buf, (size_t)count * 8, hipMemcpyDeviceToDevice, st));
    return I3D_OK;
}

extern "C" int i3d_rccl_available(void) { return rccl() != nullptr ? 1 : 0; }

extern "C" int i3d_rccl_unique_id(char* out128) {
    I3D_CHECK_ARG(out128 != nullptr, "null");
    Rccl* r = rccl();
    I3D_CHECK_ARG(r != nullptr, "librccl.so could not be loaded");
    NcclUniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    for (int i = 0; i < 128; ++i) out128[i] = id.internal[i];
    return I3D_OK;
}

extern "C" int i3d_rccl_init(const char* id128, int rank, int world, void** comm) {
    I3D_CHECK_ARG(id128 != nullptr && comm != nullptr && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    Rccl* r = rccl();
    I3D_CHECK_ARG(r != nullptr, "librccl.so could not be loaded");
    NcclUniqueId id;
    for (int i = 0; i < 128; ++i) id.internal[i] = id128[i];
    const int rc = r->comm_init_rank(comm, world, id, rank);
    return rc == 0 ? I3D_OK : rccl_fail("ncclCommInitRank", rc);
}

extern "C" int i3d_rccl_destroy(void* comm) {
    Rccl* r = rccl();
    if (r != nullptr && r->comm_destroy != nullptr && comm != nullptr) r->comm_destroy(comm);
    return I3D_OK;
}

extern "C" int i3d_set_collectives_rccl(void* comm, int world, void* scratch, long scratch_bytes) {
    I3D_CHECK_ARG(comm != nullptr && rccl() != nullptr, "no communicator");
    I3dCollectives c;
    c.world = world; c.all_gather_f32 = rccl_all_gather_f32; c.all_reduce_f64 = rccl_all_reduce_f64; c.user = comm;
    c.scratch = scratch; c.scratch_bytes = scratch_bytes;
    return i3d_set_collectives(&c);
}
