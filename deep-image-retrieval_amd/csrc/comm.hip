// comm.hip — the one exchange step of the path as a C-ABI call: all-gather of per-GPU descriptor blocks
// over RCCL / xGMI, for a SINGLE process that drives several MI355X (the shape of the reference's
// nn.DataParallel use, dirtorch/utils/common.py:150-175: one Python process, all GPUs).  The
// one-process-per-GPU deployment reaches the same collective through torch.distributed
// (dirtorch_amd/distributed.py, backend "nccl" = RCCL); this file is the boundary for hosts without
// torch.  SURVEY.md §8b: dir_comm_init_all / dir_allgather_desc.
//
// librccl is bound at first use with dlopen (the library itself has no link-time dependency on it, so
// it loads on boxes without RCCL and in the CPU-only build container); a process that already loaded
// torch gets torch's own copy of the same soname.
#include "dir_common.h"

#include <dlfcn.h>
#include <mutex>
#include <vector>

namespace dir {

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;   // ncclSuccess == 0
typedef int ncclDataType_t; // ncclFloat32 == 7 (nccl.h / rccl.h: ncclInt8 0 .. ncclFloat16 6, ncclFloat32 7)
static constexpr ncclDataType_t kNcclFloat32 = 7;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.handle, s);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

static int nccl_fail(const char* what, ncclResult_t rc) {
    Rccl& r = rccl();
    return fail(DIR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

}  // namespace dir

struct dir_comm {
    std::vector<int> devices;
    std::vector<dir::ncclComm_t> comms;
};

using namespace dir;

extern "C" {

int dir_comm_init_all(int ndev, const int* devices, dir_comm** out) {
    if (ndev <= 0 || !out) return fail(DIR_ERR_INVALID, "comm_init_all: bad argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(DIR_ERR_STATE, r.error);
    int have = 0;
    DIR_HIP_CHECK(hipGetDeviceCount(&have));
    dir_comm* c = new dir_comm();
    for (int i = 0; i < ndev; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= have) {
            delete c;
            return fail(DIR_ERR_INVALID, "comm_init_all: no such device " + std::to_string(d));
        }
        c->devices.push_back(d);
    }
    c->comms.resize(ndev);
    const ncclResult_t rc = r.CommInitAll(c->comms.data(), ndev, c->devices.data());
    if (rc != 0) {
        delete c;
        return nccl_fail("ncclCommInitAll", rc);
    }
    *out = c;
    return DIR_OK;
}

int dir_comm_size(const dir_comm* c, int* ndev) {
    if (!c || !ndev) return fail(DIR_ERR_INVALID, "comm_size: null argument");
    *ndev = (int)c->comms.size();
    return DIR_OK;
}

int dir_comm_destroy(dir_comm* c) {
    if (!c) return DIR_OK;
    Rccl& r = rccl();
    for (ncclComm_t k : c->comms)
        if (k && r.CommDestroy) (void)r.CommDestroy(k);
    delete c;
    return DIR_OK;
}

int dir_allgather_desc(dir_comm* c, const float* const* send, float* const* recv, size_t rows, int D,
                       void* const* streams) {
    if (!c || !send || !recv) return fail(DIR_ERR_INVALID, "allgather_desc: null argument");
    if (D <= 0) return fail(DIR_ERR_INVALID, "allgather_desc: D must be positive");
    if (rows == 0) return DIR_OK;
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(DIR_ERR_STATE, r.error);
    int cur = 0;
    DIR_HIP_CHECK(hipGetDevice(&cur));
    const size_t count = rows * (size_t)D;   // equal counts per rank: the caller pads the last shard
    ncclResult_t rc = r.GroupStart();
    if (rc != 0) return nccl_fail("ncclGroupStart", rc);
    for (size_t i = 0; i < c->comms.size() && rc == 0; ++i) {
        if (!send[i] || !recv[i]) {
            (void)r.GroupEnd();
            (void)hipSetDevice(cur);
            return fail(DIR_ERR_INVALID, "allgather_desc: null buffer for rank " + std::to_string(i));
        }
        (void)hipSetDevice(c->devices[i]);
        rc = r.AllGather(send[i], recv[i], count, kNcclFloat32, c->comms[i],
                         (hipStream_t)(streams ? streams[i] : nullptr));
    }
    const ncclResult_t rc2 = r.GroupEnd();
    (void)hipSetDevice(cur);
    if (rc != 0) return nccl_fail("ncclAllGather", rc);
    if (rc2 != 0) return nccl_fail("ncclGroupEnd", rc2);
    return DIR_OK;
}

}  // extern "C"
