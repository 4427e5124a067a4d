// fdnn_server.cpp -- the multi-stream scoring loop: many utterances in flight
// on one MI355X behind one model handle.
//
// The reference's serving shape is caller-side threads over independent utterances, one
// CalculationContext per call (QuantizedDnn.java:72-107, MultiThreadedStressTest.java:48-69), and
// the README blames the per-call round trips for what the lazy path fails to gain (README.md:45).
// On the GPU the same shape wants two things the per-call API cannot give:
//
//   * batches in flight.  A full-size batch fills the chip kernel by kernel, so consecutive
//     batches are serialised on one compute stream; but the soft-max scale of batch i is a pure
//     HBM pass with the matrix and vector pipes idle, and layer 0 of batch i+1 is a pure fp32
//     VALU kernel with HBM idle.  The scale pass therefore goes to a tail stream, behind an event,
//     and runs under the next batch's layer 0.  That needs per-batch scratch (a context per
//     slot) and a completion handle per batch instead of "the stream is the handle": tickets.
//   * coalescing.  A 100-frame utterance is 1/100 of a launch that fills the chip.  Host-pointer
//     submissions from any number of threads are queued, packed into one batch per slot (up to
//     max_frames), scored once, and scattered back to the callers' buffers.  Frames are
//     independent and every kernel is batch-size invariant (tests: any split of a batch gives the
//     same bits), so a coalesced utterance is bit-identical to the same utterance scored alone.
//
// Small batches (too few tiles to fill the chip) run whole on the slot's own stream instead, so
// that several of them overlap.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fdnn_internal.hpp"

using fdnn::DeviceGuard;
using fdnn::fail;

namespace {

// below this many frames a batch leaves most CUs idle: run it on its own stream next to others
constexpr int kSmallBatch = 2560;

struct Piece {  // one caller's rows inside a coalesced batch
  uint64_t ticket;
  float *out;      // caller's destination of these rows
  int row0, rows;  // rows [row0, row0 + rows) of the batch
  bool last;       // the ticket's final piece
  int state = 0;   // 0 batch in flight, 1 rows ready in the slot's pinned buffer, 2 being copied out, 3 done
  const uint64_t *bits = nullptr;  // lazy submissions with bit masks: the caller's words of these rows (compacted return)
};

struct TicketState {
  int status = 0;      // 0 running, != 0 failed with that fdnn_status
  int created = 0;     // pieces packed so far
  int done = 0;        // pieces copied out
  int failed = 0;      // pieces lost with their batch (never copied)
  bool closed = false; // the final piece has been packed, or the rest of the request was dropped after a failure
  // a piece is in flight while created > done + failed: wait() does not hand a failed ticket back to its caller before
  // that is over -- the packer still reads the caller's frames and the copy threads still write the caller's rows
};

struct Request {
  uint64_t ticket;
  const float *x;
  const int8_t *masks;  // may be null
  float *out;
  int n;
  int taken = 0;  // frames already packed into earlier batches
  const uint64_t *bits = nullptr;  // fdnn_server_submit_lazy_bits: [n][ceil(O / 64)] (then masks is null)
  int most = 0;                    // ... and the largest number of active nodes in any of its rows (counted by the submitter)
};

// what a host batch carries: requests of different kinds do not share a batch, except that dense callers may ride in a
// byte-mask batch (all active)
enum BatchKind { kDense = 0, kBytes = 1, kBits = 2 };
inline int kind_of(const Request &r) { return r.bits ? kBits : r.masks ? kBytes : kDense; }

struct Slot {
  fdnn_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;   // small batches: the whole batch; host batches: copies
  hipEvent_t gemm_done = nullptr, tail_done = nullptr, staged = nullptr, done = nullptr;
  uint64_t ticket = 0;            // device submissions: the ticket this slot last carried
  bool used = false;              // `done` has been recorded at least once
  // host submissions (allocated on first use)
  float *h_x = nullptr, *d_out = nullptr;
  float *h_out = nullptr;         // whole-row batches of several callers: the rows' landing area on the host (pinned; ONE transfer
  size_t h_out_floats = 0;        // per batch at the link's rate instead of one pageable copy per caller), allocated on first use
  bool rows_on_host = false;      // current batch: its rows have been brought to h_out
  int8_t *h_mask = nullptr;
  uint64_t *h_bits = nullptr;     // bit-mask batches: the batch's words (pinned; first such batch allocates)
  bool h_bits_pageable = false;   // ... from malloc: pinned memory was refused
  float *d_comp = nullptr;        // ... its compacted result rows [rows][stride]
  float *h_comp = nullptr;        // ... and where they land on the host (pinned): ONE transfer per batch, behind the compaction
  size_t comp_floats = 0;
  int stride = 0;                 // current batch: floats per compacted row (0: the rows leave whole)
  std::vector<Piece> pieces;
  int frames = 0;
  int pieces_left = 0;            // pieces of the current host batch not copied out yet
  bool in_flight = false;         // host batch enqueued, not yet scattered
};

}  // namespace

struct fdnn_server {
  fdnn_model *m = nullptr;
  int max_frames = 0, depth = 0;
  hipStream_t s_main = nullptr, s_tail = nullptr;
  std::vector<Slot> slots;
  std::mutex mu;                  // submission order + slot table
  uint64_t next_ticket = 1;
  uint64_t next_slot = 0;       // device submissions (under mu)
  uint64_t host_next_slot = 0;  // host batches (under qmu)
  // host path
  std::mutex qmu;
  std::condition_variable qcv, done_cv, slot_cv;
  bool packer_done = false;       // set by fdnn_server_free once the packer thread has been joined (under qmu)
  std::deque<Request> queue;
  std::deque<int> flying;         // slots with a host batch enqueued, in launch order
  std::unordered_map<uint64_t, TicketState> pending;  // host tickets not yet complete (a failed one stays until waited for)
  std::thread packer, finisher;
  // Staging of the batch being packed (frames / masks into the slot's pinned buffers) is shared with the callers blocked in
  // fdnn_server_wait: sixteen 100-frame utterances are 4.4 MB of memcpy, half a millisecond on the packer thread alone and
  // the longest stage of the loop; pieces are handed out one by one under qmu (stage_one).
  struct StageJob {
    Slot *sl = nullptr;
    const std::vector<Request> *taken = nullptr;
    std::vector<int> row0;
    int kind = 0;
    bool any_mask = false;
    int next = 0, done = 0, total = 0;
  } stage;
  bool staging = false;
  std::condition_variable stage_cv;
  bool stop = false;
  bool host_ready = false;
  int linger_us = 0;
  // statistics
  std::atomic<uint64_t> n_batches{0}, n_frames{0}, n_requests{0}, n_coalesced{0};
};

namespace {

int alloc_host_side(fdnn_server *s) {
  const fdnn::BlobHeader &h = s->m->hm.hdr;
  for (Slot &sl : s->slots) {
    const size_t n = size_t(s->max_frames);
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&sl.h_x), sizeof(float) * n * h.in_dim, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&sl.h_mask), n * h.out_dim, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&sl.d_out), sizeof(float) * n * h.out_dim);
    // the slot's context was made lean: the device-side landing buffers of host batches live here
    if (e == hipSuccess && !sl.ctx->d_x) e = hipMalloc(reinterpret_cast<void **>(&sl.ctx->d_x), sizeof(float) * n * h.in_dim);
    if (e == hipSuccess && !sl.ctx->d_mask) e = hipMalloc(reinterpret_cast<void **>(&sl.ctx->d_mask), n * h.out_dim);
    if (e != hipSuccess)
      return fail(e == hipErrorOutOfMemory ? FDNN_E_NOMEM : FDNN_E_DEVICE, std::string("server staging: ") + hipGetErrorString(e));
  }
  return FDNN_OK;
}

// Enqueue one batch that is already on the device.  Large batches: compute on the shared main
// stream, soft-max scale on the tail stream; small ones: everything on the slot's stream.
// `after` (may be null) is an event the compute must wait for (the batch's H2D copy);
// returns with `sl.done` recorded behind the last kernel.
int enqueue_batch(fdnn_server *s, Slot &sl, const float *d_x, int n, const int8_t *d_masks, float *d_out, hipEvent_t after,
                  hipStream_t *last_stream, const uint64_t *d_bits = nullptr, float *d_comp = nullptr, int stride = 0) {
  fdnn_ctx *c = sl.ctx;
  c->n = n;
  c->last = -1;
  const bool small = n <= kSmallBatch;
  // Large batches whose output layer cannot scale its own soft-max (odd widths, FDNN_FUSE_NORM=0; dense and batched-lazy
  // batches of the usual shapes can, and have no scale pass at all): the scale pass runs in line behind the output GEMM.
  // FDNN_SERVER_OVERLAP=1 restores round 2's arrangement -- the pass as a background kernel on the tail stream under
  // the next batch's layer 0, which must then be the all-VALU chain kernel (the matrix-pipe kernels fill the register
  // file) -- worth it while layer 0 took 270 us; with the int8 screening (140 us) the chain kernel's 325 us lose.
  static const bool overlap = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_SERVER_OVERLAP");
    return e ? std::atoi(e) != 0 : false;
  }();
  const std::vector<std::pair<int, int>> chunks = small ? std::vector<std::pair<int, int>>{} : fdnn::frame_chunks(n, c->m);
  bool all_fused = !small;
  for (const auto &ch : chunks) all_fused = all_fused && (d_bits || fdnn::output_will_fuse(c, ch.second, d_masks));  // (a short tail chunk may take the unfused kernels)
  c->l0_chain_only = !small && overlap && !all_fused;  // see fdnn_ctx: an overlapped scale pass needs room beside layer 0
  hipStream_t cs = small ? sl.stream : s->s_main;
  if (after) HIP_TRY(hipStreamWaitEvent(cs, after, 0));
  HIP_TRY(fdnn::ctx_enter(c, cs));
  int rc = FDNN_OK;
  hipStream_t end = cs;
  if (small) {
    rc = fdnn::run_hidden(c, d_x, cs, nullptr);
    if (!rc) rc = fdnn::run_output(c, 0, n, d_masks, d_out, cs, nullptr, nullptr, nullptr, nullptr, d_bits);
  } else {
    // very large batches go chunk by chunk (fdnn::frame_chunks), and the chunks overlap like batches do: chunk j's
    // scale pass runs on the tail stream under chunk j+1's layer 0.  One context serves all chunks, so the compute
    // stream may not overwrite the soft-max partial sums (chunk j+1's output GEMM) before chunk j's scale pass has
    // read them: it waits for `tail_done` there.
    const size_t D = size_t(s->m->hm.hdr.in_dim), O = size_t(s->m->hm.hdr.out_dim), wpr = (O + 63) / 64;
    bool first = true;
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      const auto &ch = chunks[ci];
      c->n = ch.second;
      rc = fdnn::run_hidden(c, d_x + size_t(ch.first) * D, cs, nullptr);
      if (rc) break;
      if (!first && overlap) HIP_TRY(hipStreamWaitEvent(cs, sl.tail_done, 0));
      rc = fdnn::run_output(c, 0, ch.second, d_masks ? d_masks + size_t(ch.first) * O : nullptr, d_out + size_t(ch.first) * O, cs,
                            nullptr, nullptr, overlap ? s->s_tail : nullptr, overlap ? sl.gemm_done : nullptr,
                            d_bits ? d_bits + size_t(ch.first) * wpr : nullptr);
      if (rc) break;
      // (without the overlap everything is on the compute stream, in order; a record is 3-4 us of queue time)
      if (overlap && ci + 1 < chunks.size()) HIP_TRY(hipEventRecord(sl.tail_done, s->s_tail));
      first = false;
    }
    c->n = n;
    end = overlap ? s->s_tail : cs;
  }
  // bit-mask host batches: the rows compacted (active probabilities + one value per frame) for the trip over PCIe
  if (!rc && d_bits && d_comp && stride > 0)
    fdnn::launch_lazy_compact(d_out, d_bits, d_comp, n, s->m->hm.hdr.out_dim, stride, end);
  fdnn::ctx_leave(c, end);
  if (rc) return rc;
  *last_stream = end;
  return FDNN_OK;
}

// A batch carrying `ticket` failed: whatever of that request is still queued will never be packed (the packer would read
// caller memory that the caller, told of the failure, is free to release).  Call with qmu held.
void drop_queued_rest(fdnn_server *s, uint64_t ticket) {
  for (auto it = s->queue.begin(); it != s->queue.end();) {
    if (it->ticket == ticket)
      it = s->queue.erase(it);
    else
      ++it;
  }
  auto t = s->pending.find(ticket);
  if (t != s->pending.end()) t->second.closed = true;
}

// Hands rows back to their callers.  The rows of a finished batch sit in the slot's device buffer; whoever gets
// there first copies a piece out -- the caller blocked in fdnn_server_wait (so that many callers copy in
// parallel: 3.2 MB per 100-frame utterance, one thread's copies would be the slowest stage of the loop) or the
// finisher thread (so that a ticket nobody waits for cannot hold a slot).  Call with qmu held; the lock is
// dropped around each memcpy.  `only_ticket` = 0 takes any ready piece of the slot.
void copy_ready_pieces(fdnn_server *s, std::unique_lock<std::mutex> &lk, Slot &sl, uint64_t only_ticket) {
  const size_t O = size_t(s->m->hm.hdr.out_dim);
  for (size_t i = 0; i < sl.pieces.size(); ++i) {
    Piece &p = sl.pieces[i];
    if (p.state != 1 || (only_ticket && p.ticket != only_ticket)) continue;
    p.state = 2;
    const Piece job = p;  // the vector is stable while in_flight, but copy what the unlocked part needs anyway
    lk.unlock();
    // straight from the slot's device buffer into the caller's memory, on the copying thread's own stream: no pinned
    // bounce buffer and no second pass over the 32 KB per frame
    hipError_t ce;
    const size_t stride = size_t(sl.stride);  // (stable while the slot is in flight)
    if (job.bits && stride > 0) {
      // lazy rows, compacted: the batch's rows arrived in the slot's pinned buffer with the batch (one transfer, enqueued
      // behind the compaction); THIS thread -- the caller's own, normally, so that sixteen callers work side by side --
      // expands its rows into the caller's block (fdnn::lazy_expand_rows_from)
      ce = hipSuccess;
      fdnn::lazy_expand_rows_from(job.out, sl.h_comp + size_t(job.row0) * stride, job.rows, O, stride, job.bits);
    } else if (sl.rows_on_host) {
      // whole rows of a coalesced batch: they came to the slot's pinned buffer with the batch; this thread moves its own
      ce = hipSuccess;
      std::memcpy(job.out, sl.h_out + size_t(job.row0) * O, sizeof(float) * size_t(job.rows) * O);
    } else {
      DeviceGuard dg(s->m->device);
      ce = hipMemcpyAsync(job.out, sl.d_out + size_t(job.row0) * O, sizeof(float) * size_t(job.rows) * O, hipMemcpyDeviceToHost,
                          hipStreamPerThread);
      if (ce == hipSuccess) ce = hipStreamSynchronize(hipStreamPerThread);
    }
    lk.lock();
    sl.pieces[i].state = 3;
    auto it = s->pending.find(job.ticket);
    bool finished = false;
    if (it != s->pending.end()) {
      it->second.done++;
      if (ce != hipSuccess) {
        it->second.status = FDNN_E_DEVICE;
        drop_queued_rest(s, job.ticket);
      }
      if (it->second.closed && it->second.done == it->second.created && it->second.status == 0) {
        s->pending.erase(it);
        finished = true;
      }
    }
    if (--sl.pieces_left == 0) {
      sl.in_flight = false;
      s->slot_cv.notify_all();
    }
    if (finished || ce != hipSuccess) s->done_cv.notify_all();
  }
}

// One request's frames (and masks) into the slot's pinned buffers, by whoever asks: the packer, or a caller waiting for its
// ticket.  Call with qmu held (dropped around the copies); false = nothing left to hand out.
bool stage_one(fdnn_server *s, std::unique_lock<std::mutex> &lk) {
  if (!s->staging || s->stage.next >= s->stage.total) return false;
  const int i = s->stage.next++;
  Slot &sl = *s->stage.sl;
  const Request r = (*s->stage.taken)[size_t(i)];
  const int r0 = s->stage.row0[size_t(i)], kind = s->stage.kind;
  const bool any_mask = s->stage.any_mask;
  lk.unlock();
  const fdnn::BlobHeader &h = s->m->hm.hdr;
  const size_t D = size_t(h.in_dim), O = size_t(h.out_dim), wpr = (O + 63) / 64;
  std::memcpy(sl.h_x + size_t(r0) * D, r.x + size_t(r.taken) * D, sizeof(float) * size_t(r.n) * D);
  if (kind == kBits && sl.h_bits) {  // (only bit-mask requests are in such a batch)
    std::memcpy(sl.h_bits + size_t(r0) * wpr, r.bits + size_t(r.taken) * wpr, sizeof(uint64_t) * size_t(r.n) * wpr);
  } else if (any_mask) {
    if (r.masks)
      std::memcpy(sl.h_mask + size_t(r0) * O, r.masks + size_t(r.taken) * O, size_t(r.n) * O);
    else
      std::memset(sl.h_mask + size_t(r0) * O, 1, size_t(r.n) * O);  // dense caller inside a lazy batch: all active
  }
  lk.lock();
  if (++s->stage.done == s->stage.total) s->stage_cv.notify_all();
  return true;
}

// Packs queued requests into batches and enqueues them.
void packer_loop(fdnn_server *s) {
  DeviceGuard g(s->m->device);
  const fdnn::BlobHeader &h = s->m->hm.hdr;
  const size_t D = size_t(h.in_dim), O = size_t(h.out_dim);
  for (;;) {
    std::unique_lock<std::mutex> lk(s->qmu);
    s->qcv.wait(lk, [&] { return s->stop || !s->queue.empty(); });
    if (s->stop && s->queue.empty()) return;
    bool gpu_busy = false;  // lingering only pays while an earlier batch keeps the device busy: an idle server launches at once
    for (const Slot &sl : s->slots) gpu_busy |= sl.in_flight;
    if (s->linger_us > 0 && !s->stop && gpu_busy) {  // give concurrent callers a moment to join the batch
      size_t have = 0;
      for (const Request &r : s->queue) have += size_t(r.n - r.taken);
      if (have < size_t(s->max_frames))
        s->qcv.wait_for(lk, std::chrono::microseconds(s->linger_us), [&] {
          size_t hv = 0;
          for (const Request &r : s->queue) hv += size_t(r.n - r.taken);
          return s->stop || hv >= size_t(s->max_frames);
        });
    }
    // a free slot (its previous host batch scattered)
    int si = -1;
    s->slot_cv.wait(lk, [&] {
      for (int k = 0; k < s->depth; ++k) {
        const int cand = int((s->host_next_slot + uint64_t(k)) % uint64_t(s->depth));
        if (!s->slots[size_t(cand)].in_flight) {
          si = cand;
          return true;
        }
      }
      return s->stop;
    });
    if (si < 0) return;
    Slot &sl = s->slots[size_t(si)];
    // take requests, whole or in part, until the batch is full
    std::vector<Request> taken;
    sl.pieces.clear();
    int rows = 0, most = 0;
    bool any_mask = false;
    const size_t wpr = (O + 63) / 64;
    // a batch carries bit-mask requests only, or none (dense and byte-mask callers share batches as before: the dense rows
    // of such a batch get all-active masks); a request of the other sort waits for the next batch
    int kind = kDense;
    while (!s->queue.empty() && rows < s->max_frames) {
      Request &r = s->queue.front();
      const int rk = kind_of(r);
      if (rows == 0)
        kind = rk;
      else if ((rk == kBits) != (kind == kBits))
        break;
      else if (rk == kBytes)
        kind = kBytes;
      most = std::max(most, r.most);
      const int take = std::min(r.n - r.taken, s->max_frames - rows);
      Request part = r;
      part.taken = r.taken;
      part.n = take;  // rows of this request in THIS batch
      taken.push_back(part);
      any_mask |= r.masks != nullptr;
      r.taken += take;
      const bool last = r.taken == r.n;
      sl.pieces.push_back(Piece{r.ticket, r.out + size_t(part.taken) * O, rows, take, last, 0, r.bits ? r.bits + size_t(part.taken) * wpr : nullptr});
      {
        auto it = s->pending.find(r.ticket);
        if (it != s->pending.end()) {
          it->second.created++;
          if (last) it->second.closed = true;
        }
      }
      rows += take;
      if (last) s->queue.pop_front();
    }
    sl.frames = rows;
    // compacted return (bit-mask batches): worth it while a row is at most 3/4 active nodes
    const size_t stride = size_t(most) + 1;
    sl.stride = kind == kBits && stride * 4 <= O * 3 ? int(stride) : 0;
    sl.pieces_left = int(sl.pieces.size());
    sl.in_flight = true;
    s->host_next_slot = uint64_t(si) + 1;
    lk.unlock();

    // stage (pinned), copy, compute, copy back -- all asynchronous from here
    hipError_t e = hipSuccess;
    // Whole rows of a batch that several callers share (and that is not too large to pin: 1 GB of pinned memory per slot at
    // most) go to the host in one transfer; a single caller's batch is copied by that caller straight into its memory.
    const bool compacted = kind == kBits && sl.stride > 0;
    sl.rows_on_host = false;
    if (!compacted && taken.size() > 1 && size_t(s->max_frames) * O * sizeof(float) <= (size_t(1) << 30)) {
      const size_t need = size_t(s->max_frames) * O;
      if (sl.h_out_floats < need) {
        e = hipHostMalloc(reinterpret_cast<void **>(&sl.h_out), sizeof(float) * need, hipHostMallocDefault);
        if (e == hipSuccess) sl.h_out_floats = need;
      }
      sl.rows_on_host = e == hipSuccess;
      e = hipSuccess;  // (no pinned memory to be had: the per-caller copies as before)
    }
    if (kind == kBits) {  // staging of the first bit-mask batch of this slot
      if (!sl.h_bits) {
        const size_t bytes = sizeof(uint64_t) * size_t(s->max_frames) * wpr;
        if (hipHostMalloc(reinterpret_cast<void **>(&sl.h_bits), bytes, hipHostMallocDefault) != hipSuccess) {
          // (advisor, round 5) no pinned memory to be had: pageable staging -- the copy to the device is then a synchronous
          // one, the batch is not lost
          (void)hipGetLastError();
          sl.h_bits = static_cast<uint64_t *>(std::malloc(bytes));
          sl.h_bits_pageable = sl.h_bits != nullptr;
          if (!sl.h_bits) e = hipErrorOutOfMemory;
        }
      }
      const size_t need = size_t(s->max_frames) * (O * 3 / 4 + 1);
      if (e == hipSuccess && sl.stride > 0 && sl.comp_floats < need) {
        // both halves or neither: a device buffer without its pinned landing place was leaked by the next batch's attempt,
        // and a failure here is not the batch's -- its rows go back uncompacted (per-caller copies)
        float *d_new = nullptr, *h_new = nullptr;
        hipError_t ec = hipMalloc(reinterpret_cast<void **>(&d_new), sizeof(float) * need);
        if (ec == hipSuccess) ec = hipHostMalloc(reinterpret_cast<void **>(&h_new), sizeof(float) * need, hipHostMallocDefault);
        if (ec == hipSuccess) {
          if (sl.d_comp) hipFree(sl.d_comp);
          if (sl.h_comp) hipHostFree(sl.h_comp);
          sl.d_comp = d_new;
          sl.h_comp = h_new;
          sl.comp_floats = need;
        } else {
          (void)hipGetLastError();
          if (d_new) hipFree(d_new);
          sl.stride = 0;
        }
      }
    }
    if (e == hipSuccess) {  // the pieces' copies: this thread and whoever is blocked in fdnn_server_wait, piece by piece
      lk.lock();
      s->stage.sl = &sl;
      s->stage.taken = &taken;
      s->stage.row0.clear();
      int r0 = 0;
      for (const Request &r : taken) {
        s->stage.row0.push_back(r0);
        r0 += r.n;
      }
      s->stage.kind = kind;
      s->stage.any_mask = any_mask;
      s->stage.next = s->stage.done = 0;
      s->stage.total = int(taken.size());
      s->staging = true;
      if (taken.size() > 1) s->done_cv.notify_all();
      while (stage_one(s, lk)) {
      }
      s->stage_cv.wait(lk, [&] { return s->stage.done == s->stage.total; });
      s->staging = false;
      lk.unlock();
    }
    int rc = FDNN_OK;
    if (e == hipSuccess) {
      std::lock_guard<std::mutex> order(s->mu);  // launch order against device submissions
      fdnn_ctx *c = sl.ctx;
      const bool bits = kind == kBits;
      e = hipMemcpyAsync(c->d_x, sl.h_x, sizeof(float) * size_t(rows) * D, hipMemcpyHostToDevice, sl.stream);
      if (e == hipSuccess && bits)
        e = hipMemcpyAsync(c->d_mask_bits, sl.h_bits, sizeof(uint64_t) * size_t(rows) * wpr, hipMemcpyHostToDevice, sl.stream);
      else if (e == hipSuccess && any_mask)
        e = hipMemcpyAsync(c->d_mask, sl.h_mask, size_t(rows) * O, hipMemcpyHostToDevice, sl.stream);
      if (e == hipSuccess) e = hipEventRecord(sl.staged, sl.stream);
      hipStream_t last = sl.stream;
      if (e == hipSuccess)
        rc = enqueue_batch(s, sl, c->d_x, rows, !bits && any_mask ? c->d_mask : nullptr, sl.d_out, sl.staged, &last, bits ? c->d_mask_bits : nullptr,
                           sl.d_comp, sl.stride);
      if (e == hipSuccess && !rc) {
        if (last != sl.stream) {  // results leave on the slot's stream, behind the tail stream's scale pass
          e = hipEventRecord(sl.gemm_done, last);  // (gemm_done is free again: the scale pass already waits on its earlier record)
          if (e == hipSuccess) e = hipStreamWaitEvent(sl.stream, sl.gemm_done, 0);
        }
        if (e == hipSuccess && bits && sl.stride > 0)  // compacted rows: to the host with the batch
          e = hipMemcpyAsync(sl.h_comp, sl.d_comp, sizeof(float) * size_t(rows) * size_t(sl.stride), hipMemcpyDeviceToHost, sl.stream);
        else if (e == hipSuccess && sl.rows_on_host)
          e = hipMemcpyAsync(sl.h_out, sl.d_out, sizeof(float) * size_t(rows) * O, hipMemcpyDeviceToHost, sl.stream);
        if (e == hipSuccess) e = hipEventRecord(sl.done, sl.stream);  // the rows leave later, piece by piece (copy_ready_pieces)
      }
    }
    if (e != hipSuccess) rc = fail(FDNN_E_DEVICE, std::string("server batch: ") + hipGetErrorString(e));
    s->n_batches++;
    s->n_frames += uint64_t(rows);
    if (taken.size() > 1) s->n_coalesced += taken.size();
    {
      std::lock_guard<std::mutex> lk2(s->qmu);
      if (rc) {  // nothing usable was enqueued: fail the tickets, free the slot
        sl.in_flight = false;
        sl.pieces_left = 0;
        for (const Piece &p : sl.pieces) {
          auto it = s->pending.find(p.ticket);
          if (it != s->pending.end()) {
            it->second.status = rc;
            it->second.failed++;
            drop_queued_rest(s, p.ticket);
          }
        }
      } else {
        s->flying.push_back(si);
      }
    }
    if (rc) {
      s->done_cv.notify_all();
      s->slot_cv.notify_all();
    } else {
      s->qcv.notify_all();  // wakes the finisher
    }
  }
}

// Waits for enqueued host batches in launch order, marks their rows ready and helps copying them out.
void finisher_loop(fdnn_server *s) {
  DeviceGuard g(s->m->device);
  for (;;) {
    int si;
    {
      std::unique_lock<std::mutex> lk(s->qmu);
      // (the finisher outlives the packer: at shutdown the packer still drains the queue, and a batch it enqueues after the
      // finisher had gone would never be marked ready or scattered)
      s->qcv.wait(lk, [&] { return (s->stop && s->packer_done) || !s->flying.empty(); });
      if (s->flying.empty()) {
        if (s->stop && s->packer_done) return;
        continue;
      }
      si = s->flying.front();
      s->flying.pop_front();
    }
    Slot &sl = s->slots[size_t(si)];
    const hipError_t e = hipEventSynchronize(sl.done);
    std::unique_lock<std::mutex> lk(s->qmu);
    if (e != hipSuccess) {  // the batch is lost: fail its tickets, free the slot
      for (Piece &p : sl.pieces) {
        p.state = 3;
        auto it = s->pending.find(p.ticket);
        if (it != s->pending.end()) {
          it->second.status = FDNN_E_DEVICE;
          it->second.failed++;
          drop_queued_rest(s, p.ticket);
        }
      }
      sl.pieces_left = 0;
      sl.in_flight = false;
      s->slot_cv.notify_all();
      s->done_cv.notify_all();
      continue;
    }
    for (Piece &p : sl.pieces) p.state = 1;
    s->done_cv.notify_all();             // callers blocked in wait() copy their own rows ...
    copy_ready_pieces(s, lk, sl, 0);     // ... and this thread takes whatever nobody has claimed
  }
}

int start_host_side(fdnn_server *s) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->host_ready) return FDNN_OK;
  int rc = alloc_host_side(s);
  if (rc) return rc;
  s->packer = std::thread(packer_loop, s);
  s->finisher = std::thread(finisher_loop, s);
  s->host_ready = true;
  return FDNN_OK;
}

}  // namespace

extern "C" {

int fdnn_server_create(fdnn_model *m, int max_frames, int depth, fdnn_server **out) {
  if (!m || !out) return fail(FDNN_E_ARG, "null argument");
  *out = nullptr;
  if (max_frames <= 0 || depth < 1 || depth > 16) return fail(FDNN_E_ARG, "server needs max_frames > 0 and 1 <= depth <= 16");
  DeviceGuard g(m->device);
  if (!g.ok) return fail(FDNN_E_DEVICE, "hipSetDevice failed");
  std::unique_ptr<fdnn_server> s(new fdnn_server());
  s->m = m;
  s->max_frames = max_frames;
  s->depth = depth;
  s->slots.resize(size_t(depth));
  // the compute stream outranks the tail stream: when layer 0 of batch i+1 and the scale pass of
  // batch i are both runnable, the dispatcher must place layer 0's big workgroups first
  int prio_least = 0, prio_greatest = 0;
  hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  hipError_t e = hipStreamCreateWithPriority(&s->s_main, hipStreamNonBlocking, prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&s->s_tail, hipStreamNonBlocking, prio_least);
  int rc = FDNN_OK;
  for (Slot &sl : s->slots) {
    if (e != hipSuccess || rc) break;
    rc = fdnn::make_ctx(m, max_frames, &sl.ctx, /*lean=*/true);
    if (rc) break;
    e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
    sl.ctx->durable[0] = s->s_main;  // the loop's streams outlive the slot's context: its ordering records can wait until
    sl.ctx->durable[1] = s->s_tail;  // another stream asks for them (fdnn::ctx_leave)
    sl.ctx->durable[2] = sl.stream;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.gemm_done, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.tail_done, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.staged, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
  }
  if (e != hipSuccess && !rc) rc = fail(FDNN_E_DEVICE, std::string("server: ") + hipGetErrorString(e));
  if (rc) {
    fdnn_server_free(s.release());
    return rc;
  }
  *out = s.release();
  return FDNN_OK;
}

void fdnn_server_free(fdnn_server *s) {
  if (!s) return;
  DeviceGuard g(s->m->device);
  {
    std::lock_guard<std::mutex> lk(s->qmu);
    s->stop = true;
  }
  s->qcv.notify_all();
  s->slot_cv.notify_all();
  if (s->packer.joinable()) s->packer.join();
  {
    std::lock_guard<std::mutex> lk(s->qmu);
    s->packer_done = true;
  }
  s->qcv.notify_all();
  if (s->finisher.joinable()) s->finisher.join();
  for (Slot &sl : s->slots) {
    if (sl.used && sl.done) hipEventSynchronize(sl.done);
    if (sl.stream) hipStreamSynchronize(sl.stream);
  }
  for (Slot &sl : s->slots)
    if (sl.stream) fdnn::fuse_chain_retire_stream(s->m->device, sl.stream);
  if (s->s_main) fdnn::fuse_chain_retire_stream(s->m->device, s->s_main);  // (the device's chain of fused launches may still name them)
  if (s->s_tail) fdnn::fuse_chain_retire_stream(s->m->device, s->s_tail);
  if (s->s_main) hipStreamSynchronize(s->s_main);
  if (s->s_tail) hipStreamSynchronize(s->s_tail);
  for (Slot &sl : s->slots) {
    if (sl.ctx) fdnn::destroy_ctx(sl.ctx);
    if (sl.h_x) hipHostFree(sl.h_x);
    if (sl.h_mask) hipHostFree(sl.h_mask);
    if (sl.h_bits && sl.h_bits_pageable) std::free(sl.h_bits);
    else if (sl.h_bits) hipHostFree(sl.h_bits);
    if (sl.d_comp) hipFree(sl.d_comp);
    if (sl.h_comp) hipHostFree(sl.h_comp);
    if (sl.h_out) hipHostFree(sl.h_out);
    if (sl.d_out) hipFree(sl.d_out);
    if (sl.gemm_done) hipEventDestroy(sl.gemm_done);
    if (sl.tail_done) hipEventDestroy(sl.tail_done);
    if (sl.staged) hipEventDestroy(sl.staged);
    if (sl.done) hipEventDestroy(sl.done);
    if (sl.stream) hipStreamDestroy(sl.stream);
  }
  if (s->s_main) hipStreamDestroy(s->s_main);
  if (s->s_tail) hipStreamDestroy(s->s_tail);
  delete s;
}

int fdnn_server_set_linger_us(fdnn_server *s, int microseconds) {
  if (!s || microseconds < 0) return fail(FDNN_E_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(s->qmu);
  s->linger_us = microseconds;
  return FDNN_OK;
}

int fdnn_server_submit_device(fdnn_server *s, const float *d_x, int n, const int8_t *d_masks, float *d_out, uint64_t *ticket) {
  if (!s || !ticket) return fail(FDNN_E_ARG, "null argument");
  if (n <= 0 || n > s->max_frames) return fail(FDNN_E_ARG, "frame count must be in 1..max_frames of the server");
  if (!d_x || !d_out) return fail(FDNN_E_ARG, "null buffer");
  DeviceGuard g(s->m->device);
  std::unique_lock<std::mutex> lk(s->mu);
  // next slot in submission order; its previous batch must have completed (that bounds the batches in flight)
  int si = -1;
  for (;;) {
    const int cand = int(s->next_slot % uint64_t(s->depth));
    Slot &c = s->slots[size_t(cand)];
    bool host_busy;
    {
      std::lock_guard<std::mutex> q(s->qmu);
      host_busy = c.in_flight;
    }
    if (!host_busy) {
      si = cand;
      break;
    }
    lk.unlock();  // a host batch owns it: wait for the finisher, then look again
    {
      std::unique_lock<std::mutex> q(s->qmu);
      s->slot_cv.wait(q, [&] { return !s->slots[size_t(cand)].in_flight; });
    }
    lk.lock();
  }
  Slot &sl = s->slots[size_t(si)];
  if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));
  hipStream_t last = nullptr;
  int rc = enqueue_batch(s, sl, d_x, n, d_masks, d_out, nullptr, &last);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(sl.done, last));
  sl.used = true;
  sl.ticket = s->next_ticket++;
  s->next_slot = uint64_t(si) + 1;
  *ticket = sl.ticket;
  s->n_batches++;
  s->n_frames += uint64_t(n);
  s->n_requests++;
  return FDNN_OK;
}

int fdnn_server_submit(fdnn_server *s, const float *x, int n, const int8_t *masks, float *out, uint64_t *ticket) {
  if (!s || !ticket) return fail(FDNN_E_ARG, "null argument");
  if (n <= 0) return fail(FDNN_E_ARG, "frame count must be positive");
  if (!x || !out) return fail(FDNN_E_ARG, "null buffer");
  int rc = start_host_side(s);
  if (rc) return rc;
  uint64_t t;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    t = s->next_ticket++;
  }
  {
    std::lock_guard<std::mutex> lk(s->qmu);
    s->pending.emplace(t, TicketState{});
    s->queue.push_back(Request{t, x, masks, out, n, 0});
  }
  s->n_requests++;
  s->qcv.notify_all();
  *ticket = t;
  return FDNN_OK;
}

int fdnn_server_submit_lazy_bits(fdnn_server *s, const float *x, int n, const uint64_t *bits, float *out, uint64_t *ticket) {
  if (!s || !ticket) return fail(FDNN_E_ARG, "null argument");
  if (n <= 0) return fail(FDNN_E_ARG, "frame count must be positive");
  if (!x || !out || !bits) return fail(FDNN_E_ARG, "null buffer");
  int rc = start_host_side(s);
  if (rc) return rc;
  // the widest row of the request, counted here (on the caller's thread) for the batch's compacted row length
  const size_t O = size_t(s->m->hm.hdr.out_dim), wpr = (O + 63) / 64;
  const uint64_t tail_mask = (O & 63) ? ((uint64_t(1) << (O & 63)) - 1) : ~uint64_t(0);
  int most = 0;
  for (int f = 0; f < n; ++f) {
    const uint64_t *row = bits + size_t(f) * wpr;
    int k = 0;
    for (size_t w = 0; w + 1 < wpr; ++w) k += __builtin_popcountll(row[w]);
    k += __builtin_popcountll(row[wpr - 1] & tail_mask);
    most = std::max(most, k);
  }
  uint64_t t;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    t = s->next_ticket++;
  }
  {
    std::lock_guard<std::mutex> lk(s->qmu);
    s->pending.emplace(t, TicketState{});
    Request r{t, x, nullptr, out, n, 0};
    r.bits = bits;
    r.most = most;
    s->queue.push_back(r);
  }
  s->n_requests++;
  s->qcv.notify_all();
  *ticket = t;
  return FDNN_OK;
}

int fdnn_server_wait(fdnn_server *s, uint64_t ticket) {
  if (!s) return fail(FDNN_E_ARG, "null argument");
  {  // a host ticket?
    std::unique_lock<std::mutex> lk(s->qmu);
    if (s->pending.find(ticket) != s->pending.end()) {
      for (;;) {
        auto it = s->pending.find(ticket);
        if (it == s->pending.end()) return FDNN_OK;  // complete (the last piece was copied, here or by the finisher)
        if (it->second.status != 0 && it->second.closed && it->second.created == it->second.done + it->second.failed) {
          const int status = it->second.status;  // no piece of it is queued or in flight any more
          s->pending.erase(it);
          return fail(status, "a batch carrying this ticket failed on the device");
        }
        bool copied = false;
        for (Slot &sl : s->slots) {
          if (!sl.in_flight) continue;
          for (const Piece &p : sl.pieces)
            if (p.ticket == ticket && p.state == 1) {
              copy_ready_pieces(s, lk, sl, ticket);
              copied = true;
              break;
            }
        }
        if (!copied && !stage_one(s, lk)) s->done_cv.wait(lk);  // (idle hands: help the packer stage the next batch)
      }
    }
  }
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    if (ticket == 0 || ticket >= s->next_ticket) return fail(FDNN_E_ARG, "unknown ticket");
    for (Slot &sl : s->slots)
      if (sl.used && sl.ticket == ticket) ev = sl.done;
  }
  if (!ev) return FDNN_OK;  // its slot has been reused since: a slot is only reused after completion
  DeviceGuard g(s->m->device);
  HIP_TRY(hipEventSynchronize(ev));
  return FDNN_OK;
}

int fdnn_server_drain(fdnn_server *s) {
  if (!s) return fail(FDNN_E_ARG, "null argument");
  {
    std::unique_lock<std::mutex> lk(s->qmu);
    s->done_cv.wait(lk, [&] {
      for (auto it = s->pending.begin(); it != s->pending.end();) {
        const TicketState &t = it->second;
        if (t.status == 0 || !t.closed || t.created != t.done + t.failed) return false;  // running, or pieces still in flight
        it = s->pending.erase(it);  // failed and quiescent, nobody waited for it: forget it
      }
      return true;
    });
  }
  DeviceGuard g(s->m->device);
  std::lock_guard<std::mutex> lk(s->mu);
  for (Slot &sl : s->slots)
    if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));
  return FDNN_OK;
}

int fdnn_server_stats(fdnn_server *s, uint64_t *batches, uint64_t *frames, uint64_t *requests, uint64_t *coalesced_requests) {
  if (!s) return fail(FDNN_E_ARG, "null argument");
  if (batches) *batches = s->n_batches.load();
  if (frames) *frames = s->n_frames.load();
  if (requests) *requests = s->n_requests.load();
  if (coalesced_requests) *coalesced_requests = s->n_coalesced.load();
  return FDNN_OK;
}

}  // extern "C"
