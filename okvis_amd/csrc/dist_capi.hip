// The one collective of the multi-GPU design without a Python launcher in the data path: an all-gather
// of the fixed-size per-window records over RCCL.  librccl.so is loaded at run time (dlopen: the library has no link-time
// dependency on it and single-GPU users never touch it); the ncclUniqueId travels from rank 0 to the other ranks through a
// file on a file system all ranks of the node see (one node, one process per GPU: /dev/shm or /tmp).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/okvis_amd_ba.h"

namespace {

// the part of the NCCL / RCCL C API used here (nccl.h: ncclUniqueId is 128 opaque bytes, ncclUint8 = 1, ncclSuccess = 0)
struct UniqueId {
  char internal[128];
};
typedef void* Comm;
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    const char* names[] = {std::getenv("OKVIS_BA_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) return false;
    GetUniqueId = (int (*)(UniqueId*))dlsym(handle, "ncclGetUniqueId");
    CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(handle, "ncclCommInitRank");
    AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))dlsym(handle, "ncclAllGather");
    CommDestroy = (int (*)(Comm))dlsym(handle, "ncclCommDestroy");
    GetErrorString = (const char* (*)(int))dlsym(handle, "ncclGetErrorString");
    return GetUniqueId && CommInitRank && AllGather && CommDestroy;
  }
};

bool read_id(const std::string& file, UniqueId* id) {
  FILE* f = std::fopen(file.c_str(), "rb");
  if (!f) return false;
  const size_t n = std::fread(id->internal, 1, sizeof(id->internal), f);
  std::fclose(f);
  return n == sizeof(id->internal);
}

}  // namespace

extern "C" {

int okvis_ba_gather_records(int32_t rank, int32_t world, int device, const char* id_file, double timeout_s,
                            const okvis_ba_window_record* mine, int32_t n_per_rank, okvis_ba_window_record* all) {
  if (world <= 0 || rank < 0 || rank >= world || n_per_rank < 0 || !all || (n_per_rank > 0 && !mine)) return OKVIS_BA_ERR_ARG;
  if (world > 1 && (!id_file || !*id_file)) return OKVIS_BA_ERR_ARG;
  const size_t bytes = sizeof(okvis_ba_window_record) * (size_t)n_per_rank;
  if (n_per_rank == 0) return OKVIS_BA_OK;
  Rccl R;
  UniqueId id;
  std::memset(&id, 0, sizeof(id));
  // every rank loads RCCL BEFORE an id changes hands: a rank that cannot (OKVIS_BA_ERR_UNSUPPORTED) then fails without having
  // made the others wait inside ncclCommInitRank for it
  if (!R.load()) return OKVIS_BA_ERR_UNSUPPORTED;   // no librccl.so to be found
  // ---- the id: rank 0 makes it and publishes it with an atomic rename, the others wait for the file.  The path belongs to
  //      ONE gather: rank 0 clears whatever an earlier (crashed) job left there before it makes the id, and removes the file
  //      again as soon as the communicator stands (every rank has read it by then) ----
  if (rank == 0) {
    if (world > 1) {
      (void)std::remove(id_file);
      (void)std::remove((std::string(id_file) + ".tmp").c_str());
    }
    if (R.GetUniqueId(&id) != 0) return OKVIS_BA_ERR_STATE;
    if (world > 1) {
      const std::string tmp = std::string(id_file) + ".tmp";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (!f) return OKVIS_BA_ERR_ARG;
      const bool ok = std::fwrite(id.internal, 1, sizeof(id.internal), f) == sizeof(id.internal);
      std::fclose(f);
      if (!ok || std::rename(tmp.c_str(), id_file) != 0) return OKVIS_BA_ERR_ARG;
    }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    while (!read_id(id_file, &id)) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return OKVIS_BA_ERR_STATE;
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  struct RemoveIdFile {   // rank 0, on every way out from here on (a failed rank 0 must not leave an id nobody will honour)
    const char* f;
    ~RemoveIdFile() { if (f) (void)std::remove(f); }
  } remove_id{(rank == 0 && world > 1) ? id_file : nullptr};
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return OKVIS_BA_ERR_NO_DEVICE;
  if (device < 0 || device >= n_dev) return OKVIS_BA_ERR_ARG;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return OKVIS_BA_HIP_ERROR_BASE + (int)e;
  Comm comm = nullptr;
  if (R.CommInitRank(&comm, world, id, rank) != 0) return OKVIS_BA_ERR_STATE;
  if (remove_id.f) {   // the communicator stands: every rank has the id
    (void)std::remove(remove_id.f);
    remove_id.f = nullptr;
  }
  unsigned char *d_send = nullptr, *d_recv = nullptr;
  hipStream_t st = nullptr;
  int rc = OKVIS_BA_OK;
  auto hip = [&](hipError_t err) {
    if (err != hipSuccess && rc == OKVIS_BA_OK) rc = OKVIS_BA_HIP_ERROR_BASE + (int)err;
    return err == hipSuccess;
  };
  if (hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) && hip(hipMalloc((void**)&d_send, bytes)) &&
      hip(hipMalloc((void**)&d_recv, bytes * (size_t)world)) &&
      hip(hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, st))) {
    if (R.AllGather(d_send, d_recv, bytes, /*ncclUint8*/ 1, comm, st) != 0) rc = OKVIS_BA_ERR_STATE;
    if (rc == OKVIS_BA_OK && hip(hipMemcpyAsync(all, d_recv, bytes * (size_t)world, hipMemcpyDeviceToHost, st)))
      hip(hipStreamSynchronize(st));
  }
  if (d_send) (void)hipFree(d_send);
  if (d_recv) (void)hipFree(d_recv);
  if (st) (void)hipStreamDestroy(st);
  (void)R.CommDestroy(comm);
  return rc;
}

int okvis_ba_batch_run_gathered(int device, int32_t rank, int32_t world, int32_t n_total, const okvis_ba_window* all_windows,
                                const okvis_ba_options* opt, int num_iter, const char* id_file, double timeout_s,
                                okvis_ba_window_record* all_records) {
  if (world <= 0 || n_total <= 0 || !all_records) return OKVIS_BA_ERR_ARG;
  const int32_t per = (n_total + world - 1) / world;
  std::vector<okvis_ba_window_record> mine((size_t)per);
  for (auto& r : mine) r.window_id = 0xffffffffu, r.iterations = 0, r.final_cost = 0, r.seconds = 0;   // padding records
  int32_t n_mine = 0;
  int rc = okvis_ba_batch_run(device, rank, world, n_total, all_windows, opt, num_iter, mine.data(), &n_mine);
  if (rc != OKVIS_BA_OK) return rc;
  std::vector<okvis_ba_window_record> gathered((size_t)per * (size_t)world);
  rc = okvis_ba_gather_records(rank, world, device, id_file, timeout_s, mine.data(), per, gathered.data());
  if (rc != OKVIS_BA_OK) return rc;
  // records in window order (window i ran on rank i mod world as its (i / world)-th window)
  for (int32_t i = 0; i < n_total; ++i) all_records[i] = gathered[(size_t)(i % world) * (size_t)per + (size_t)(i / world)];
  return OKVIS_BA_OK;
}

}  // extern "C"
