// fdnn_group.cpp -- one process, N MI355X of one node, behind the same C-ABI / JNI surface.
//
// BASELINE north_star: "frame batches shard across the 8 GPUs of one node with an RCCL broadcast
// of the quantized weights over xGMI at load time only".  fast-dnn_amd/dist.py does that for one
// process per GPU (torch.distributed); this file does it for ONE host process -- the shape a JVM
// has: the unmodified Java class calls Java_suskun_nn_QuantizedDnn_calculate
// (jni_dnn.cc:35-62) on one handle, and the frames of that call are scored on all devices.
//
//   load:       the leader device parses + quantizes once (fdnn_model_load_on); the packed blob
//               (~45 MB for 7x2048 -> 8000) goes device-to-device to every peer -- hipMemcpyPeer
//               (xGMI on one node), or one ncclBroadcast when FDNN_GROUP_BCAST=rccl (librccl is
//               dlopen'ed, not linked: a process that also holds PyTorch must not see two RCCLs)
//               -- and each peer adopts it (fdnn_model_import_blob), so all replicas hold
//               bit-identical weights.
//   calculate:  contiguous frame shards (sizes differ by at most one frame, the same rule as
//               dist.frame_shards), one host thread per device, each shard through the ordinary
//               single-device path into its slice of the caller's output.  No collective, no
//               device-to-device traffic in steady state: frames are independent.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fdnn_internal.hpp"

using fdnn::DeviceGuard;
using fdnn::fail;

struct fdnn_group {
  std::vector<fdnn_model *> models;  // models[0] = leader (quantized the weights)
  std::vector<int> devices;
  std::string bcast;                 // how the weights travelled: "peer-copy" | "rccl" | "none"
};

namespace {

// ---- RCCL through dlopen (single process, one communicator per device)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  bool load() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
  }
};
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)

// One ncclBroadcast of `bytes` from bufs[0] (device devs[0]) into bufs[i] (device devs[i]).
bool rccl_broadcast(const std::vector<int> &devs, const std::vector<void *> &bufs, size_t bytes, std::string *why) {
  for (size_t i = 0; i < devs.size(); ++i)
    for (size_t j = 0; j < i; ++j)
      if (devs[i] == devs[j]) {
        *why = "a device appears twice in the group (RCCL needs distinct devices)";
        return false;
      }
  Rccl r;
  if (!r.load()) {
    *why = "librccl could not be loaded";
    return false;
  }
  const int n = int(devs.size());
  std::vector<void *> comms(size_t(n), nullptr);
  if (r.CommInitAll(comms.data(), n, devs.data()) != 0) {
    *why = "ncclCommInitAll failed";
    return false;
  }
  std::vector<hipStream_t> streams(size_t(n), nullptr);
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    DeviceGuard g(devs[size_t(i)]);
    ok = g.ok && hipStreamCreateWithFlags(&streams[size_t(i)], hipStreamNonBlocking) == hipSuccess;
  }
  if (ok) {
    r.GroupStart();
    for (int i = 0; i < n; ++i) {
      DeviceGuard g(devs[size_t(i)]);
      if (r.Broadcast(bufs[size_t(i)], bufs[size_t(i)], bytes, kNcclUint8, 0, comms[size_t(i)], streams[size_t(i)]) != 0) ok = false;
    }
    if (r.GroupEnd() != 0) ok = false;
  }
  for (int i = 0; i < n; ++i) {
    DeviceGuard g(devs[size_t(i)]);
    if (streams[size_t(i)]) {
      if (hipStreamSynchronize(streams[size_t(i)]) != hipSuccess) ok = false;
      hipStreamDestroy(streams[size_t(i)]);
    }
    if (comms[size_t(i)]) r.CommDestroy(comms[size_t(i)]);
  }
  if (!ok) *why = "ncclBroadcast failed";
  return ok;
}

}  // namespace

namespace fdnn {

// [start, stop) of shard r when n frames are cut into `world` contiguous shards
void frame_shard(int n, int world, int r, int *start, int *stop) {
  const int base = n / world, extra = n % world;
  *start = r * base + (r < extra ? r : extra);
  *stop = *start + base + (r < extra ? 1 : 0);
}

int calculate_on_one_device(fdnn_model *m, const float *x, int n, int dim, int batch_hint, float *out);  // fdnn_runtime.cpp

}  // namespace fdnn

extern "C" {

int fdnn_group_load(const char *path, float cutoff, const int *devices, int n_devices, fdnn_group **out) {
  if (!path || !devices || !out) return fail(FDNN_E_ARG, "null argument");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return fail(FDNN_E_ARG, "a group holds 1..64 devices");
  fdnn_group *g = new fdnn_group();
  g->devices.assign(devices, devices + n_devices);
  fdnn_model *leader = nullptr;
  int rc = fdnn_model_load_on(path, cutoff, devices[0], &leader);
  if (rc) {
    delete g;
    return rc;
  }
  g->models.push_back(leader);
  g->bcast = "none";
  const size_t bytes = leader->hm.blob.size();
  // staging buffers for the received blob on every peer (the leader's own d_blob is the source)
  std::vector<void *> bufs(size_t(n_devices), nullptr);
  bufs[0] = leader->d_blob;
  hipError_t e = hipSuccess;
  for (int i = 1; i < n_devices && e == hipSuccess; ++i) {
    DeviceGuard dg(devices[i]);
    if (!dg.ok) {
      e = hipErrorInvalidDevice;
      break;
    }
    e = hipMalloc(&bufs[size_t(i)], bytes);
  }
  const char *mode = std::getenv("FDNN_GROUP_BCAST");
  const bool want_rccl = mode && std::strcmp(mode, "rccl") == 0;
  if (e == hipSuccess && (n_devices > 1 || want_rccl)) {  // (a one-device group still runs the collective when asked: plumbing test)
    bool done = false;
    if (want_rccl) {
      std::string why;
      done = rccl_broadcast(g->devices, bufs, bytes, &why);
      if (done) g->bcast = "rccl";
    }
    if (!done) {
      for (int i = 1; i < n_devices && e == hipSuccess; ++i) {
        if (devices[i] == devices[0]) {
          DeviceGuard dg(devices[i]);
          e = hipMemcpy(bufs[size_t(i)], leader->d_blob, bytes, hipMemcpyDeviceToDevice);
        } else {
          e = hipMemcpyPeer(bufs[size_t(i)], devices[i], leader->d_blob, devices[0], bytes);
        }
      }
      if (n_devices > 1) g->bcast = "peer-copy";
    }
  }
  if (e != hipSuccess) rc = fail(FDNN_E_DEVICE, std::string("group weight distribution: ") + hipGetErrorString(e));
  for (int i = 1; i < n_devices && !rc; ++i) {
    fdnn_model *peer = nullptr;
    rc = fdnn_model_import_blob(bufs[size_t(i)], bytes, devices[i], &peer);
    if (!rc) g->models.push_back(peer);
  }
  for (int i = 1; i < n_devices; ++i)
    if (bufs[size_t(i)]) {
      DeviceGuard dg(devices[i]);
      hipFree(bufs[size_t(i)]);
    }
  if (rc) {
    fdnn_group_free(g);
    return rc;
  }
  *out = g;
  return FDNN_OK;
}

void fdnn_group_free(fdnn_group *g) {
  if (!g) return;
  for (fdnn_model *m : g->models) {
    if (m->group == g) m->group = nullptr;
    fdnn_model_free(m);
  }
  delete g;
}

int fdnn_group_size(const fdnn_group *g) { return g ? int(g->models.size()) : -1; }

fdnn_model *fdnn_group_model(const fdnn_group *g, int index) {
  if (!g || index < 0 || index >= int(g->models.size())) return nullptr;
  return g->models[size_t(index)];
}

const char *fdnn_group_weight_transport(const fdnn_group *g) { return g ? g->bcast.c_str() : ""; }

int fdnn_group_calculate(fdnn_group *g, const float *x, int n, int dim, int batch_hint, float *out) {
  if (!g || n < 0) return fail(FDNN_E_ARG, "bad argument");
  if (n == 0) return FDNN_OK;
  if (!x || !out) return fail(FDNN_E_ARG, "null buffer");
  fdnn_model *leader = g->models[0];
  const int D = leader->hm.hdr.in_dim, O = leader->hm.hdr.out_dim;
  if (dim != D)
    return fail(FDNN_E_ARG, "input vector size " + std::to_string(dim) + " must be equal with network input size " + std::to_string(D));
  const int world = int(g->models.size());
  std::vector<int> rcs(size_t(world), FDNN_OK);
  std::vector<std::string> msgs(static_cast<size_t>(world));
  auto shard = [&](int r) {
    int a, b;
    fdnn::frame_shard(n, world, r, &a, &b);
    if (b == a) return;
    fdnn_model *m = g->models[size_t(r)];
    m->l0_fma = leader->l0_fma;  // one numeric flavour per group
    rcs[size_t(r)] = fdnn::calculate_on_one_device(m, x + size_t(a) * D, b - a, dim, batch_hint, out + size_t(a) * O);
    if (rcs[size_t(r)]) msgs[size_t(r)] = fdnn_last_error();  // thread-local: carry it to the caller
  };
  std::vector<std::thread> th;
  for (int r = 1; r < world; ++r) th.emplace_back(shard, r);
  shard(0);
  for (auto &t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (rcs[size_t(r)]) return fail(rcs[size_t(r)], "device " + std::to_string(g->devices[size_t(r)]) + ": " + msgs[size_t(r)]);
  return FDNN_OK;
}

void fdnn_group_shard(int n, int world, int rank, int *start, int *stop) {
  int a = 0, b = 0;
  if (world > 0 && rank >= 0 && rank < world && n >= 0) fdnn::frame_shard(n, world, rank, &a, &b);
  if (start) *start = a;
  if (stop) *stop = b;
}

int fdnn_group_attach(fdnn_group *g) {
  if (!g) return fail(FDNN_E_ARG, "null group");
  g->models[0]->group = g;  // fdnn_calculate(leader, ...) now shards over the group; fdnn_model_free(leader) frees it
  return FDNN_OK;
}

}  // extern "C"
