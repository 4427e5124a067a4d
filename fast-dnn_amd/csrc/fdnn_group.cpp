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
//   calculate:  a LARGE call is cut into contiguous frame shards (sizes differ by at most one frame, the same rule
//               as dist.frame_shards), one shard per device, each through the ordinary single-device path into its
//               slice of the caller's output, on that device's own persistent host thread (pinned to the CPUs of
//               the device's NUMA node, so that its staging copies stay local).  A SMALL call -- the JNI serving
//               shape: 100-frame utterances from many Java threads -- stays whole and goes to ONE replica, chosen
//               round robin, on the caller's own thread: eight devices then serve eight callers at a time instead
//               of every call paying eight thread hand-offs, eight H2D copies and 72 launches for 12 frames each.
//               No collective, no device-to-device traffic in steady state: frames are independent.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdnn_internal.hpp"

using fdnn::DeviceGuard;
using fdnn::fail;

// One persistent host thread per replica: runs the shards of large calls on its device.
struct GroupWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  bool stop = false;
  std::string cpus;  // the CPU list it pinned itself to ("" = not pinned)
};

struct fdnn_group {
  std::vector<fdnn_model *> models;  // models[0] = leader (quantized the weights)
  std::vector<int> devices;
  std::string bcast;                 // how the weights travelled: "peer-copy" | "rccl" | "none"
  std::vector<GroupWorker *> workers;  // started on the first large call
  std::mutex workers_mu;
  std::atomic<unsigned> next_replica{0};  // round robin of the small calls
  int split_min = 4096;              // calls of fewer frames stay on one device (FDNN_GROUP_SPLIT_MIN)
  int shard_min = 1024;              // a shard is at least this many frames (fewer devices take part otherwise)
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


// "0-15,128-143" -> the calling thread's affinity mask; false when nothing usable was parsed
bool pin_to_cpulist(const std::string &list) {
  cpu_set_t set;
  CPU_ZERO(&set);
  int found = 0;
  for (const char *q = list.c_str(); *q;) {
    char *end = nullptr;
    const long a = std::strtol(q, &end, 10);
    if (end == q) break;
    long b = a;
    q = end;
    if (*q == '-') {
      b = std::strtol(q + 1, &end, 10);
      q = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
      CPU_SET(static_cast<int>(c), &set);
      ++found;
    }
    if (*q == ',') ++q;
  }
  return found > 0 && pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
}

// CPUs local to the device's PCIe root (its NUMA node): /sys/bus/pci/devices/<bus id>/local_cpulist
std::string device_local_cpus(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return "";
  for (char *c = bus; *c; ++c)
    if (*c >= 'A' && *c <= 'Z') *c = static_cast<char>(*c - 'A' + 'a');
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
  std::string out;
  if (FILE *f = std::fopen(path.c_str(), "r")) {
    char buf[512];
    if (std::fgets(buf, sizeof(buf), f)) out = buf;
    std::fclose(f);
  }
  while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
  return out;
}

void worker_loop(GroupWorker *w, int device) {
  static const bool no_pin = std::getenv("FDNN_GROUP_NO_PIN") != nullptr;
  if (!no_pin) {
    const std::string cpus = device_local_cpus(device);
    if (!cpus.empty() && pin_to_cpulist(cpus)) {
      std::lock_guard<std::mutex> lk(w->mu);
      w->cpus = cpus;
    }
  }
  for (;;) {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->stop || !w->jobs.empty(); });
      if (w->jobs.empty()) return;  // stop, and nothing left to run
      job = std::move(w->jobs.front());
      w->jobs.pop_front();
    }
    job();
  }
}

void start_workers(fdnn_group *g) {
  std::lock_guard<std::mutex> lk(g->workers_mu);
  if (!g->workers.empty()) return;
  for (size_t r = 0; r < g->models.size(); ++r) {
    GroupWorker *w = new GroupWorker();
    w->th = std::thread(worker_loop, w, g->devices[r]);
    g->workers.push_back(w);
  }
}

void stop_workers(fdnn_group *g) {
  std::lock_guard<std::mutex> lk(g->workers_mu);
  for (GroupWorker *w : g->workers) {
    {
      std::lock_guard<std::mutex> l2(w->mu);
      w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  g->workers.clear();
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
  if (const char *e = std::getenv("FDNN_GROUP_SPLIT_MIN")) g->split_min = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("FDNN_GROUP_SHARD_MIN")) g->shard_min = std::max(1, std::atoi(e));
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
    // FDNN_BATCHER on every replica or on none: the leader got its batcher in fdnn_model_load_on
    if (!rc && leader->batcher) {
      int mf = 0, depth = 2, linger = 0;
      const char *env = std::getenv("FDNN_BATCHER");
      if (env && std::sscanf(env, "%d:%d:%d", &mf, &depth, &linger) >= 1 && mf > 0) rc = fdnn_model_enable_batcher(peer, mf, depth, linger);
    }
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
  stop_workers(g);
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
  // small call: whole, on one replica (round robin), on the caller's thread
  if (world == 1 || n < g->split_min) {
    const unsigned r = world == 1 ? 0u : g->next_replica.fetch_add(1, std::memory_order_relaxed) % unsigned(world);
    fdnn_model *m = g->models[r];
    m->l0_fma = leader->l0_fma;  // one numeric flavour per group
    return fdnn::calculate_on_one_device(m, x, n, dim, batch_hint, out);
  }
  // large call: contiguous shards over as many replicas as keep a shard at shard_min frames or more, each on its
  // device's persistent worker; the caller waits
  const int use = std::max(1, std::min(world, n / std::max(1, g->shard_min)));
  start_workers(g);
  const unsigned first = g->next_replica.fetch_add(unsigned(use), std::memory_order_relaxed);
  std::vector<int> rcs(size_t(use), FDNN_OK);
  std::vector<std::string> msgs(static_cast<size_t>(use));
  std::mutex done_mu;
  std::condition_variable done_cv;
  int left = use;
  for (int k = 0; k < use; ++k) {
    const int r = int((first + unsigned(k)) % unsigned(world));
    int a, b;
    fdnn::frame_shard(n, use, k, &a, &b);
    GroupWorker *w = g->workers[size_t(r)];
    auto job = [&, k, r, a, b] {
      if (b > a) {
        fdnn_model *m = g->models[size_t(r)];
        m->l0_fma = leader->l0_fma;
        rcs[size_t(k)] = fdnn::calculate_on_one_device(m, x + size_t(a) * D, b - a, dim, batch_hint, out + size_t(a) * O);
        if (rcs[size_t(k)]) msgs[size_t(k)] = fdnn_last_error();  // thread-local: carry it to the caller
      }
      std::lock_guard<std::mutex> lk(done_mu);
      if (--left == 0) done_cv.notify_all();
    };
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->jobs.emplace_back(job);
    }
    w->cv.notify_one();
  }
  {
    std::unique_lock<std::mutex> lk(done_mu);
    done_cv.wait(lk, [&] { return left == 0; });
  }
  for (int k = 0; k < use; ++k)
    if (rcs[size_t(k)])
      return fail(rcs[size_t(k)], "device " + std::to_string(g->devices[size_t((first + unsigned(k)) % unsigned(world))]) + ": " + msgs[size_t(k)]);
  return FDNN_OK;
}

// Where the worker thread of replica `index` pinned itself ("" = no worker yet / not pinned).  Diagnostics.
const char *fdnn_group_worker_cpus(fdnn_group *g, int index) {
  if (!g || index < 0) return "";
  std::lock_guard<std::mutex> lk(g->workers_mu);
  if (size_t(index) >= g->workers.size()) return "";
  GroupWorker *w = g->workers[size_t(index)];
  std::lock_guard<std::mutex> l2(w->mu);
  return w->cpus.c_str();
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
