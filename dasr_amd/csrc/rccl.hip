// RCCL at the C-ABI boundary: the gradient exchange that replaces the reference's single-process nn.DataParallel
// (codes/SRN/models/networks.py:144-146: per-step parameter broadcast + gather on GPU 0) -- one process per GPU, SUM all-reduce of the
// flat fp32 gradient buffer over xGMI.  librccl is resolved lazily (dlopen) so that the compute library has no hard link dependency;
// the communicator is created from a 128-byte unique id that rank 0 hands to the other ranks through any side channel
// (dasr_amd/dist.py uses the torch.distributed store).
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
typedef void* comm_t;
struct uid128 { char b[128]; };   // ncclUniqueId (passed BY VALUE to ncclCommInitRank)
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(comm_t*, int, uid128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
} g_rccl;
typedef decltype(Rccl::CommInitRank) init_fn_t;

int load() {
    if (g_rccl.h) return 0;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return DASR_EINVAL;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (init_fn_t)dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.CommDestroy = (int (*)(comm_t))dlsym(h, "ncclCommDestroy");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.Broadcast || !g_rccl.CommDestroy) return DASR_EINVAL;
    g_rccl.h = h;
    return 0;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h
}  // namespace

extern "C" int dasr_rccl_unique_id(void* id128) {
    if (!id128) return DASR_EINVAL;
    if (int rc = load()) return rc;
    return g_rccl.GetUniqueId(id128);
}

extern "C" int dasr_rccl_init(const void* id128, int32_t rank, int32_t world, void** comm_out) {
    if (!id128 || !comm_out || rank < 0 || rank >= world) return DASR_EINVAL;
    if (int rc = load()) return rc;
    uid128 id;
    memcpy(id.b, id128, 128);
    comm_t c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, world, id, rank);
    *comm_out = c;
    return rc;
}

extern "C" int dasr_allreduce(void* comm, float* buf, int64_t count, void* stream) {
    if (!comm || !buf || count <= 0 || !g_rccl.h) return DASR_EINVAL;
    return g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, comm, as_stream(stream));
}

extern "C" int dasr_broadcast(void* comm, float* buf, int64_t count, int32_t root, void* stream) {
    if (!comm || !buf || count <= 0 || !g_rccl.h) return DASR_EINVAL;
    return g_rccl.Broadcast(buf, buf, (size_t)count, kNcclFloat32, root, comm, as_stream(stream));
}

extern "C" int dasr_rccl_destroy(void* comm) {
    if (!comm || !g_rccl.h) return DASR_EINVAL;
    return g_rccl.CommDestroy(comm);
}
