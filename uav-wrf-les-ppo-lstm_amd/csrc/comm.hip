// comm.hip -- the iteration's three exchanges over RCCL (xGMI) behind the C ABI: SURVEY 8(b) `uav_allreduce`, 8(e).
//
// One communicator per handle (= per process = per GPU).  RCCL is bound at RUN time (dlopen of librccl, the soname torch's own
// copy carries, so inside a PyTorch process this is the library torch already loaded): libuavppo.so has no link-time
// dependency on it and loads on a box without RCCL; only uav_comm_* / uav_all* fail there, loudly.  Every exchange is issued on
// the caller's stream, in the caller's program order; nothing here synchronises with the host.
//
// What rides on it (uavppo/dist_utils.py): ONE in-place sum of the flat f32 gradient per optimiser step (the
// loss kernels already scaled by 1 / global sample count, so the sum IS the mean; the clip norm is taken after it, so every
// rank clips identically: the reference's single-process clip_grad_norm_ + Adam of train_ppo2.0.py:85-88), one 3-double sum
// for the whole-buffer advantage statistics (train_ppo2.0.py:35-39), one fixed-size byte all-gather of the packed success
// bits for the replicated curriculum (model.py:131-164).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <type_traits>
#include "common.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

// nullptr + uav_set_error when RCCL cannot be bound
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("UAV_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) {
            snprintf(r.why, sizeof(r.why), "librccl not found (tried $UAV_RCCL_LIB, librccl.so.1, librccl.so, /opt/rocm/lib): %s", dlerror());
            return;
        }
        bool ok = true;
        auto bind = [&](auto& fn, const char* sym) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.so, sym));
            if (!fn) { ok = false; snprintf(r.why, sizeof(r.why), "librccl lacks %s", sym); }
        };
        bind(r.GetVersion, "ncclGetVersion");
        bind(r.GetUniqueId, "ncclGetUniqueId");
        bind(r.CommInitRank, "ncclCommInitRank");
        bind(r.CommDestroy, "ncclCommDestroy");
        bind(r.AllReduce, "ncclAllReduce");
        bind(r.AllGather, "ncclAllGather");
        bind(r.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(r.so); r.so = nullptr; }
    });
    if (!r.so) { uav_set_error("RCCL unavailable: %s", r.why); return nullptr; }
    return &r;
}

}  // namespace

#define UAV_CHECK_RCCL(R, expr)                                                                          \
    do {                                                                                                 \
        ncclResult_t e_ = (expr);                                                                        \
        if (e_ != ncclSuccess) {                                                                         \
            uav_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (R)->GetErrorString(e_));        \
            return 3;                                                                                    \
        }                                                                                                \
    } while (0)

extern "C" {

int uav_rccl_version(int* out /*host*/) {
    UAV_REQUIRE(out, "uav_rccl_version: NULL argument");
    Rccl* r = rccl();
    if (!r) return 3;
    UAV_CHECK_RCCL(r, r->GetVersion(out));
    return 0;
}

int uav_comm_unique_id(void* id_out /*host, UAV_COMM_ID_BYTES*/) {
    static_assert(UAV_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "uavppo.h: UAV_COMM_ID_BYTES");
    UAV_REQUIRE(id_out, "uav_comm_unique_id: NULL argument");
    Rccl* r = rccl();
    if (!r) return 3;
    ncclUniqueId id;
    UAV_CHECK_RCCL(r, r->GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int uav_comm_init(uav_ctx* ctx, const void* id /*host*/, int rank, int world) {
    UAV_REQUIRE(ctx && id, "uav_comm_init: NULL argument");
    UAV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "uav_comm_init: rank %d of %d", rank, world);
    UAV_REQUIRE(ctx->comm == nullptr, "uav_comm_init: this handle already has a communicator (uav_comm_destroy first)");
    Rccl* r = rccl();
    if (!r) return 3;
    UAV_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    UAV_CHECK_RCCL(r, r->CommInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return 0;
}

int uav_comm_world(const uav_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_world : 0; }
int uav_comm_rank(const uav_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_rank : -1; }

int uav_comm_destroy(uav_ctx* ctx) {
    UAV_REQUIRE(ctx, "uav_comm_destroy: NULL argument");
    if (!ctx->comm) return 0;
    Rccl* r = rccl();
    if (!r) return 3;
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    ctx->comm_rank = -1;
    UAV_CHECK_RCCL(r, r->CommDestroy(comm));
    return 0;
}

int uav_allreduce(uav_ctx* ctx, float* flat_grad, int64_t count, uav_stream stream) {
    UAV_REQUIRE(ctx && flat_grad && count > 0, "uav_allreduce: bad argument");
    UAV_REQUIRE(ctx->comm, "uav_allreduce: no communicator on this handle (uav_comm_init)");
    Rccl* r = rccl();
    if (!r) return 3;
    UAV_CHECK_RCCL(r, r->AllReduce(flat_grad, flat_grad, (size_t)count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(ctx->comm),
                                   static_cast<hipStream_t>(stream)));
    return 0;
}

int uav_allreduce_f64(uav_ctx* ctx, double* buf, int64_t count, uav_stream stream) {
    UAV_REQUIRE(ctx && buf && count > 0, "uav_allreduce_f64: bad argument");
    UAV_REQUIRE(ctx->comm, "uav_allreduce_f64: no communicator on this handle (uav_comm_init)");
    Rccl* r = rccl();
    if (!r) return 3;
    UAV_CHECK_RCCL(r, r->AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(ctx->comm),
                                   static_cast<hipStream_t>(stream)));
    return 0;
}

int uav_allgather_bytes(uav_ctx* ctx, const void* send, void* recv, int64_t bytes_per_rank, uav_stream stream) {
    UAV_REQUIRE(ctx && send && recv && bytes_per_rank > 0, "uav_allgather_bytes: bad argument");
    UAV_REQUIRE(ctx->comm, "uav_allgather_bytes: no communicator on this handle (uav_comm_init)");
    Rccl* r = rccl();
    if (!r) return 3;
    UAV_CHECK_RCCL(r, r->AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, static_cast<ncclComm_t>(ctx->comm),
                                   static_cast<hipStream_t>(stream)));
    return 0;
}

}  // extern "C"
