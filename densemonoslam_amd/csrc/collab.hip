// The collaborative session's exchange step over RCCL (include/dmslam_collab.h).  librccl is resolved at run time: the
// types come from ROCm's own header, the entry points from dlsym, so a single-camera process neither links nor loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <link.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/dmslam_collab.h"
#include "common.hpp"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

// ONE copy of RCCL per process.  A host that already has one mapped (torch.distributed's "nccl" backend maps torch/lib/librccl.so,
// a C++ front end may link its own) must not get a second instance on the same device beside it: the copy that is loaded is the
// copy that is used (RTLD_NOLOAD on the path dl_iterate_phdr reports; then by soname), and only a process without any loads one -
// DMS_RCCL_PATH if set, else librccl.so.1 / librccl.so from the loader's path, else ROCm's.  dms_collab_library_path() reports
// what the entry points resolved to (dladdr).
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  const char* path = info->dlpi_name;
  if (!path || !*path) return 0;
  const char* base = strrchr(path, '/');
  base = base ? base + 1 : path;
  if (strncmp(base, "librccl.so", 10) != 0) return 0;
  *(std::string*)out = path;
  return 1;
}

std::string g_rccl_path;  // written once under rccl()'s once_flag

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) r.so = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!r.so) r.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (!r.so)
      if (const char* e = getenv("DMS_RCCL_PATH")) r.so = dlopen(e, RTLD_NOW | RTLD_LOCAL);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!r.so) r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!r.so) return;
    auto sym = [&](const char* n) { return dlsym(r.so, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce && r.Send && r.Recv && r.GetErrorString;
    Dl_info di;
    if (r.GetUniqueId && dladdr((void*)r.GetUniqueId, &di) && di.dli_fname) g_rccl_path = di.dli_fname;
  });
  return r;
}

int need_rccl(const char* who) {
  if (rccl().ok) return DMS_OK;
  dms::set_error("%s: librccl.so.1 is not loadable (%s)", who, rccl().so ? "missing entry points" : dlerror());
  return DMS_ERR_UNSUPPORTED;
}

int comm_fail(const char* who, ncclResult_t rc) {
  dms::set_error("%s: RCCL error %d (%s)", who, (int)rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
  return DMS_ERR_COMM;
}

}  // namespace

struct dms_collab {
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
};

static_assert(sizeof(ncclUniqueId) == DMS_COLLAB_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" {

int dms_collab_unique_id(void* id128) {
  DMS_REQUIRE(id128, "null argument");
  int rc = need_rccl("dms_collab_unique_id");
  if (rc) return rc;
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return comm_fail("dms_collab_unique_id", r);
  memcpy(id128, &id, sizeof(id));
  return DMS_OK;
}

int dms_collab_create(dms_collab** out, int rank, int nranks, const void* id128) {
  DMS_REQUIRE(out && id128, "null argument");
  DMS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank / nranks out of range");
  int rc = need_rccl("dms_collab_create");
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  dms_collab* c = new dms_collab();
  c->rank = rank;
  c->size = nranks;
  // ncclCommInitRank blocks until every rank has arrived: a rank that never comes (a crashed peer, a wrong id) must end in an error
  // here, not in a hang.  The call runs on a helper thread bound to the caller's device; the caller waits DMS_RCCL_INIT_TIMEOUT_S
  // seconds (default 120).  After a timeout the helper is left behind (there is no way to cancel the call) and the caller gets
  // DMS_ERR_TIMEOUT - the process is expected to report and exit.
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    delete c;
    dms::set_error("dms_collab_create: no current device");
    return DMS_ERR_HIP;
  }
  struct Init {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t r = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto st = std::make_shared<Init>();
  std::thread([st, dev, nranks, id, rank] {
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclSystemError;
    if (hipSetDevice(dev) == hipSuccess) r = rccl().CommInitRank(&comm, nranks, id, rank);
    std::lock_guard<std::mutex> g(st->m);
    st->r = r;
    st->comm = comm;
    st->done = true;
    st->cv.notify_all();
  }).detach();
  double limit = 120.0;
  if (const char* e = getenv("DMS_RCCL_INIT_TIMEOUT_S")) limit = atof(e) > 0.0 ? atof(e) : limit;
  {
    std::unique_lock<std::mutex> g(st->m);
    if (!st->cv.wait_for(g, std::chrono::duration<double>(limit), [&] { return st->done; })) {
      delete c;
      dms::set_error("dms_collab_create: ncclCommInitRank (rank %d of %d) did not return within %.0f s - a rank is missing", rank, nranks, limit);
      return DMS_ERR_TIMEOUT;
    }
  }
  const ncclResult_t r = st->r;
  c->comm = st->comm;
  if (r != ncclSuccess) {
    delete c;
    return comm_fail("dms_collab_create", r);
  }
  *out = c;
  return DMS_OK;
}

const char* dms_collab_library_path(void) {
  (void)rccl();
  return g_rccl_path.c_str();
}

int dms_collab_rank(const dms_collab* c) { return c ? c->rank : -1; }
int dms_collab_size(const dms_collab* c) { return c ? c->size : 0; }

int dms_collab_allgather(dms_collab* c, const void* send_dev, void* recv_dev, size_t bytes, dms_stream s) {
  DMS_REQUIRE(c && send_dev && recv_dev && bytes > 0, "bad argument");
  const ncclResult_t r = rccl().AllGather(send_dev, recv_dev, bytes, ncclUint8, c->comm, (hipStream_t)s);
  return r == ncclSuccess ? DMS_OK : comm_fail("dms_collab_allgather", r);
}

int dms_collab_send(dms_collab* c, const void* src_dev, size_t bytes, int peer, dms_stream s) {
  DMS_REQUIRE(c && src_dev && bytes > 0 && peer >= 0 && peer < c->size && peer != c->rank, "bad argument");
  const ncclResult_t r = rccl().Send(src_dev, bytes, ncclUint8, peer, c->comm, (hipStream_t)s);
  return r == ncclSuccess ? DMS_OK : comm_fail("dms_collab_send", r);
}

int dms_collab_recv(dms_collab* c, void* dst_dev, size_t bytes, int peer, dms_stream s) {
  DMS_REQUIRE(c && dst_dev && bytes > 0 && peer >= 0 && peer < c->size && peer != c->rank, "bad argument");
  const ncclResult_t r = rccl().Recv(dst_dev, bytes, ncclUint8, peer, c->comm, (hipStream_t)s);
  return r == ncclSuccess ? DMS_OK : comm_fail("dms_collab_recv", r);
}

int dms_collab_allreduce_max_f64(dms_collab* c, double* value_dev, dms_stream s) {
  DMS_REQUIRE(c && value_dev, "bad argument");
  const ncclResult_t r = rccl().AllReduce(value_dev, value_dev, 1, ncclFloat64, ncclMax, c->comm, (hipStream_t)s);
  return r == ncclSuccess ? DMS_OK : comm_fail("dms_collab_allreduce_max_f64", r);
}

int dms_collab_destroy(dms_collab* c) {
  if (!c) return DMS_OK;
  ncclResult_t r = ncclSuccess;
  if (c->comm) r = rccl().CommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? DMS_OK : comm_fail("dms_collab_destroy", r);
}

}  // extern "C"
