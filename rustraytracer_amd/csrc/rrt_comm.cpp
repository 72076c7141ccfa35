// The frame's one exchange step inside the library (include/rrt.h: rrt_comm_*): a gather of the ranks' compact
// tile buffers over RCCL, so that a host that is not Python can use all GPUs of a node.
// One process per GPU, as everywhere here.  RCCL is resolved at run time (dlopen): a process that already holds an
// RCCL (PyTorch brings its own) shares it, and a single-GPU host never loads it at all.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/rrt.h"
#include "host_io.hpp"   // rrt::set_last_error

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL this process already holds first (same SONAME: PyTorch's own copy), then the system's.
        // RRT_RCCL_LIB names the library instead (a host that ships its own RCCL; the test of the "not found" path).
        const char* override_name = getenv("RRT_RCCL_LIB");
        const char* defaults[] = {"librccl.so.1", "librccl.so"};
        const char* one[] = {override_name};
        const char* const* names = override_name ? one : defaults;
        const int n_names = override_name ? 1 : 2;
        for (int i = 0; i < n_names; ++i)
            if (!r.lib) r.lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
        for (int i = 0; i < n_names; ++i)
            if (!r.lib) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            const char* e = dlerror();      // ONE call: dlerror() clears the message it returns
            r.why = std::string("RCCL not found: ") + (e ? e : "dlopen failed");
            return;
        }
        const auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p && r.why.empty()) r.why = std::string("RCCL lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int cfail(int code, const std::string& m) {
    rrt::set_last_error(m);
    return code;
}

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        const ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return cfail(RRT_E_HIP, std::string(#expr) + ": " + R->GetErrorString(r_));    \
    } while (0)

}  // namespace

struct rrt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int rrt_comm_unique_id(void* id128) {
    if (!id128) return cfail(RRT_E_INVALID, "null argument");
    Rccl* R = rccl();
    if (!R->why.empty()) return cfail(RRT_E_STATE, R->why);
    static_assert(sizeof(ncclUniqueId) == RRT_COMM_ID_BYTES, "rrt.h states the size of the id");
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return RRT_OK;
}

int rrt_comm_init(int rank, int world, const void* id128, rrt_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return cfail(RRT_E_INVALID, "bad rank / world / id");
    Rccl* R = rccl();
    if (!R->why.empty()) return cfail(RRT_E_STATE, R->why);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    rrt_comm* c = new (std::nothrow) rrt_comm();
    if (!c) return cfail(RRT_E_NOMEM, "out of host memory");
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);      // uses the device current on this thread (rrt_set_device)
    if (r != ncclSuccess) {
        delete c;
        return cfail(RRT_E_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
    }
    *out = c;
    return RRT_OK;
}

void rrt_comm_destroy(rrt_comm* c) {
    if (!c) return;
    Rccl* R = rccl();
    if (c->comm && R->CommDestroy) (void)R->CommDestroy(c->comm);
    delete c;
}

static int gather_tiles(rrt_comm* c, const void* d_local, void* d_gathered, uint32_t width, uint32_t height, int root,
                        void* stream_v, uint32_t dealt) {
    if (!c || !d_local || width == 0 || height == 0 || root < -1 || root >= c->world) return cfail(RRT_E_INVALID, "bad argument");
    const bool receives = root < 0 || root == c->rank;
    if (receives && !d_gathered) return cfail(RRT_E_INVALID, "the receiving rank needs d_gathered");
    Rccl* R = rccl();
    hipStream_t stream = (hipStream_t)stream_v;
    const size_t bytes = (size_t)rrt_tile_buffer_bytes(width, height, (uint32_t)c->world, dealt);
    if (root < 0) {
        NCCL_TRY(R->AllGather(d_local, d_gathered, bytes, ncclUint8, c->comm, stream));
        return RRT_OK;
    }
    // gather to one rank: every peer sends straight to the root -- xGMI is point to point, seven links end at the
    // root, so the seven messages (0.78 MB each at 1080p on 8 GPUs) arrive side by side; no ring is needed.
    // The root's own share is a plain copy, issued before the group opens.
    if (c->rank == root) {
        const hipError_t e = hipMemcpyAsync((char*)d_gathered + (size_t)root * bytes, d_local, bytes, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return cfail(RRT_E_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    NCCL_TRY(R->GroupStart());
    // Every exit from here on closes the group: a group left open queues every later collective of this thread and
    // never issues it (a hang).  The first error is kept, the remaining calls are skipped, GroupEnd always runs.
    ncclResult_t first = ncclSuccess;
    const char* where = "";
    if (c->rank == root) {
        for (int r = 0; r < c->world && first == ncclSuccess; ++r) {
            if (r == root) continue;
            first = R->Recv((char*)d_gathered + (size_t)r * bytes, bytes, ncclUint8, r, c->comm, stream);
            where = "ncclRecv";
        }
    } else {
        first = R->Send(d_local, bytes, ncclUint8, root, c->comm, stream);
        where = "ncclSend";
    }
    const ncclResult_t end = R->GroupEnd();
    if (first != ncclSuccess) return cfail(RRT_E_HIP, std::string(where) + ": " + R->GetErrorString(first));
    if (end != ncclSuccess) return cfail(RRT_E_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(end));
    return RRT_OK;
}

int rrt_gather_tiles(rrt_comm* c, const void* d_local, void* d_gathered, uint32_t width, uint32_t height, int root,
                     void* stream) {
    return gather_tiles(c, d_local, d_gathered, width, height, root, stream, 0);
}
int rrt_gather_tiles_dealt(rrt_comm* c, const void* d_local, void* d_gathered, uint32_t width, uint32_t height, int root,
                           void* stream) {
    return gather_tiles(c, d_local, d_gathered, width, height, root, stream, 1);
}

}  // extern "C"
