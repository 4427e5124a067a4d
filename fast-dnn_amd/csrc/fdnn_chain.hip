// fdnn_chain.hip -- the int8 HIDDEN layers of a pass in one launch (gfx950).
//
//   CalculateUntilLastHiddenLayer's loop over the quantized hidden layers (dnn.cc:413-423): per layer
//   QuantizedLayerActivations / quantizedNodeSum (dnn.cc:289-349) + AddBias (:250-264) + QuantizedSigmoid (:267-286)
//
// Layer l+1 of frame tile t needs layer l of frame tile t and nothing else, so the per-layer kernel boundary -- a wait
// for ALL 256 workgroups, a cold first stage, a prologue, the first-to-last-workgroup spread, six times per pass -- is
// more synchronisation than the data flow asks for.  Here one persistent launch walks a list of TASKS
// (layer, frame tile, node tile); a task's only wait is for the MT node tiles of ITS frame tile in the layer before.
//
// Scheduling -- no co-residency assumption anywhere.  Tasks sit in eight queues (one per XCD: queue q owns frame tiles
// q, q + 8, ...), each in layer-major order, so every task's prerequisites come EARLIER IN ITS OWN QUEUE.  A workgroup
// draws its next task with one atomic add on the queue head of the XCD it runs on (s_getreg XCC_ID; when that queue is
// exhausted, the next one: work stealing), and only then waits for the prerequisites.  Whoever holds a task is running,
// the unfinished task with the smallest index in a queue never waits, so by induction every wait ends -- whatever the
// dispatch order, the residency or the placement (which decide speed only: a frame tile's node tiles drawn on one XCD
// exchange their activation rows through that XCD's L2 fill path).  Batch sizes that are not a whole number of rounds
// cost what their tasks cost: a partial last "round" flows into the next layer instead of idling 200 CUs per layer.
//
// Hand-off (placement independent; MI355X_MICROARCH.md "inter-workgroup visibility"): the producer's activation rows
// leave as write-through stores (sc0 sc1), every wave drains them (s_waitcnt vmcnt(0)), a barrier, then ONE relaxed
// agent-scope add on the frame tile's counter of that layer; the consumer polls that counter with relaxed loads from
// one lane, a barrier, and reads the rows with sc0 sc1 LDS-DMA loads (past its CU's vector L1).  The weight stages of
// the next task do not depend on anybody: they are requested BEFORE the wait.
//
// The tile arithmetic (k-loop, pair-saturation corrections, epilogue) is the per-layer kernel's, operation for
// operation (fdnn_gemm.hip): the bytes are identical by construction and by test.
#include <atomic>
#include <climits>
#include <cstdlib>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"
#include "fdnn_tile.hpp"

#ifndef FDNN_CHAIN_SLEEP
#define FDNN_CHAIN_SLEEP 4  // 64-cycle units between two polls of a frame tile's counter
#endif
#ifndef FDNN_CHAIN_CLK
#define FDNN_CHAIN_CLK 0  // 1 (measurement builds): per-task phase clocks into QChainParams::clk
#endif
#ifndef FDNN_CHAIN_A_AUX
#define FDNN_CHAIN_A_AUX 17  // cache policy of the activation rows' LDS-DMA loads: sc0 | sc1 (they were written by other workgroups of this launch)
#endif
#ifndef FDNN_CHAIN_GROUP
#define FDNN_CHAIN_GROUP 1  // frame tiles of a queue that wait for one another as a group between layers (1: each on its own)
#endif
#ifndef FDNN_CHAIN_EARLY_W
#define FDNN_CHAIN_EARLY_W 1  // the next task's weight stages are requested before the wait for its activation rows
#endif

namespace fdnn {
namespace {

// XCD this wave runs on: s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }

// NOFIX: no layer of the chain has saturating pairs -- the k-loop without the entry walk (fdnn_gemm.hip)
template <int NF, int WN, int BK, int STAGES, int WM, bool NOFIX = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void qchain_kernel(QChainParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = GemmCfg<NF, WN, BK, STAGES, WM>;
  constexpr int FT = Cfg::FT, NW = Cfg::NW, G_BM = Cfg::G_BM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  // Scalar registers: the k-loop of the 320-frame tile leaves room for ~40 SGPRs of state beside its own (descriptors, the
  // saturation-entry walk); a task's decoded fields, the layer's pointers and the launch constants together are more than
  // that, and what the allocator cannot keep it spills into VGPR lanes -- of a kernel that sits at 256 VGPRs.  So nothing
  // but the task id and the kernel-argument pointer lives across the k-loop: every phase re-reads what it needs through
  // that pointer (scalar loads from the kernarg segment), and the pointer and the id are laundered between the phases so
  // that the compiler cannot keep the earlier phase's copies alive.
  typedef const __attribute__((address_space(4))) QChainParams *KP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)p;
  char *aux = smem + Cfg::AUX_OFF;
  int *ctl_s = reinterpret_cast<int *>(smem + Cfg::RING);  // [0] the workgroup's next task; [1] its queue, [2] queues found exhausted (thread 0's)

  // the half-step sigmoid table: once per launch (the ring's epilogue tile never reaches it)
#pragma unroll
  for (int piece = 0; piece < 3; ++piece)
    if (piece % NW == wave) {
      const int bytes = (kLut2Size + 15) & ~15;
      const __amdgpu_buffer_rsrc_t rsrc_lut = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(kp->lut2), 0, bytes, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lut, FDNN_LDS_PTR(aux + piece * 1024), 16, lane * 16, piece * 1024, 0, 0);
    }

  static_assert(NW % 2 == 0 || NW == 1, "slab parity per wave needs an even wave count (or one wave)");
  const int srow = lane / Cfg::LPR;
  const int schunk = ((lane % Cfg::LPR) ^ swz<BK>(wave * Cfg::RPI + srow)) << 4;
  const int voff_w = srow * kp->ldw + schunk;
  const int voff_a = srow * kp->lda + schunk;
  constexpr int NLD_W = Cfg::W_SLABS / NW, NLD_A = (Cfg::A_SLABS + NW - 1) / NW, NLD = NLD_W + NLD_A;
  constexpr bool ROT = STAGES == 2 && BK / 32 >= 4;
  constexpr int SUB = BK / 32;
  constexpr int ROT_D0 = (NLD + 1) / 2;
  constexpr int A_D0 = (NLD_A + 1) / 2;
  constexpr int kTS = G_BM + 16;
  const int frow = lane & 31, fch = lane >> 5, half = lane >> 5;
  const int arow0 = wn * 32 * NF;

  // ---- the queues.  Task id = queue * 2^24 + index; index -> layer-major, frame tile, node tile.
  // (thread 0 keeps its home queue and whether that queue still has tasks in LDS: ctl_s[1], ctl_s[2] (0 = yes, 8 = exhausted))
  struct Dec {
    int l, nt, mt;
    int gnt, gtarget;  // the sync group's leading frame tile (its counters are the group's) and its tasks per layer
  };
  auto decode = [&](KP k, int task) -> Dec {
    const int MT = k->rows_pad / G_BM, NT = k->n_pad / FT;
    const int q = task >> 24, ti = task & 0xffffff;
    const int nft = (NT - q + 7) >> 3;
    const int per_layer = nft * MT;
    const int l = ti / per_layer, r = ti - l * per_layer;
    const int ftl = r / MT;
    const int g0 = ftl / FDNN_CHAIN_GROUP * FDNN_CHAIN_GROUP;
    const int gsz = min(FDNN_CHAIN_GROUP, nft - g0);
    return Dec{l, q + 8 * ftl, r - ftl * MT, q + 8 * g0, gsz * MT};
  };
  auto queue_tasks = [&](KP k, int q) { return ((k->n_pad / FT - q + 7) >> 3) * (k->rows_pad / G_BM) * k->n_layers; };  // frame tiles q, q + 8, ... < NT
  // thread 0: a task from the queue with the most tasks left (its own while that has any), or -1.  All eight heads are read
  // in one go (two 16-byte loads past the L1), then one atomic on the chosen queue: two round trips per draw however many
  // queues are exhausted -- the end of a launch is a sequence of such draws by every workgroup.
  auto draw_slow = [&](KP k) -> int {
    const int own = ctl_s[1];
    uint32_t *heads = k->ctl;
    for (int attempt = 0; attempt < 64; ++attempt) {
      v4i h0, h1;
      asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(h0), "=&v"(h1)
                   : "v"(heads)
                   : "memory");
      const int hd[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      int best = -1, best_left = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int qq = (own + j) & 7;  // ties: the nearest queue after my own
        int left = 0, hv = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u == qq) hv = hd[u];
        left = queue_tasks(k, qq) - hv;
        if (j == 0 && left > 0) {
          best = qq;
          best_left = 1 << 30;
        } else if (left > best_left) {
          best = qq;
          best_left = left;
        }
      }
      if (best < 0) break;
      const uint32_t i = __hip_atomic_fetch_add(heads + best, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (i < static_cast<uint32_t>(queue_tasks(k, best))) {
        if (best != own) ctl_s[2] = 8;  // my own queue is exhausted: every further draw comes here
        return (best << 24) | static_cast<int>(i);
      }
    }
    ctl_s[2] = 8;
    return -1;
  };
  if (tid == 0) {
    ctl_s[1] = xcc_id();
    ctl_s[2] = 0;
    ctl_s[0] = draw_slow(kp);
  }
  __syncthreads();
  int task = __builtin_amdgcn_readfirstlane(ctl_s[0]);

#if FDNN_CHAIN_CLK
  long long tc[8];
#define CH_TS(i) tc[i] = __builtin_readcyclecounter()
#else
#define CH_TS(i)
#endif
#define CH_LAUNDER() asm volatile("" : "+s"(kp), "+s"(task)::"memory")

  while (task >= 0) {
    CH_TS(0);
    // ================================================================ phase A: set-up, early loads, the wait
    v16i acc[2][NF];
    __amdgpu_buffer_rsrc_t rsrc_w, rsrc_a;
    int fix_e = 0, fix_end = 0, fix_k_next = INT_MAX, fix_node0 = 0;
    typedef const __attribute__((address_space(4))) uint64_t *FixPtr;
    FixPtr ent_c = nullptr;
    uint64_t fix_raw = 0, fix_raw_nxt = 0;  // {u16 k, s8 w0, s8 w1, s32 node}
    int ldw_s, lda_s, KT;
    {
      const Dec d = decode(kp, task);
      const int m0 = d.mt * G_BM, f0 = d.nt * FT;
      typedef const __attribute__((address_space(4))) QChainLayer *LP;
      const LP L = reinterpret_cast<LP>(&kp->layer[0]) + d.l;
      ldw_s = kp->ldw;
      lda_s = kp->lda;
      KT = kp->K / BK;
      const int8_t *a_in = kp->act[d.l & 1];
      rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(L->w + static_cast<size_t>(m0) * ldw_s), 0, G_BM * ldw_s, 0x00020000);
      rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a_in + static_cast<size_t>(f0) * lda_s), 0, FT * lda_s, 0x00020000);
      // 128 * sum_k w (the s8 = u8 - 128 offset): the accumulators' start values
      const int32_t *wsum = L->wsum;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int4 ws4 = *reinterpret_cast<const int4 *>(wsum + m0 + 64 * wm + 32 * a + 8 * g + 4 * (lane >> 5));
#pragma unroll
          for (int b = 0; b < NF; ++b) {
            acc[a][b][g * 4 + 0] = ws4.x;
            acc[a][b][g * 4 + 1] = ws4.y;
            acc[a][b][g * 4 + 2] = ws4.z;
            acc[a][b][g * 4 + 3] = ws4.w;
          }
        }
      // this tile's biases: they do not depend on the layer before
      if (wave == 3 % NW) {
        const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L->bias + m0), 0, G_BM * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_bias, FDNN_LDS_PTR(aux + 3072), 16, lane * 16, 0, 0, 0);
      }
      // pmaddubsw pair saturation (dnn.cc:337-340): the entry walk of fdnn_gemm.hip
      ent_c = (FixPtr)(uintptr_t)L->fix_ent;
      fix_node0 = m0 + 64 * wm;
      if (!NOFIX && ent_c) {
        const int32_t *fix_grp = L->fix_grp;
        const int grp = (m0 >> 6) + wm;
        fix_e = __builtin_amdgcn_readfirstlane(fix_grp[grp]);
        fix_end = __builtin_amdgcn_readfirstlane(fix_grp[grp + 1]);
        if (fix_e < fix_end) {
          fix_raw = ent_c[fix_e];
          if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
          fix_k_next = static_cast<int>(fix_raw & 0xffff);
        }
      }
    }
    // activation rows: sc0 sc1 (aux 17) -- they were written by other workgroups of this launch; weights: default policy
    auto stage_load = [&](int kt, int buf, int i) {
      char *base = smem + buf * Cfg::STAGE;
      const int koff = kt * BK;
      const int odd = (NW == 1 && BK == 128) ? 64 : 0;
      if (i < NLD_W) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(base + (i * NW + wave) * 1024), 16, (i & 1) ? voff_w ^ odd : voff_w,
                                                 (i * NW + wave) * Cfg::RPI * ldw_s + koff, 0, 0);
      } else {
        const int s = i - NLD_W;
        if (Cfg::A_SLABS % NW == 0 || s * NW + wave < Cfg::A_SLABS)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, FDNN_LDS_PTR(base + Cfg::W_BYTES + (s * NW + wave) * 1024), 16,
                                                   (s & 1) ? voff_a ^ odd : voff_a, (s * NW + wave) * Cfg::RPI * lda_s + koff, 0, FDNN_CHAIN_A_AUX);
      }
    };
    if (FDNN_CHAIN_EARLY_W) {  // the weight halves of the first stages (ROT: both buffers) depend on nobody: before the wait
#pragma unroll
      for (int s = 0; s < (ROT ? 2 : STAGES - 1); ++s)
        if (s < KT) {
#pragma unroll
          for (int i = 0; i < NLD_W; ++i) stage_load(s, s, i);
        }
    }
    asm volatile("" ::: "memory");
    CH_TS(1);
    {  // ---- the one wait of the task: layer l - 1 of this frame tile, all MT node tiles
      const Dec d = decode(kp, task);
      if (d.l > 0) {
        if (tid == 0) {
          const uint32_t *dn = kp->done + static_cast<size_t>(d.gnt) * kp->n_layers + (d.l - 1);
          const uint32_t want = static_cast<uint32_t>(d.gtarget);
          int spins = 0;
          while (__hip_atomic_load(dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(FDNN_CHAIN_SLEEP);
            if (++spins > (1 << 22)) {  // seconds (a legitimate wait is microseconds): counters left dirty by a killed launch
              if (kp->faults) atomicAdd(kp->faults, 1ull);
              if (kp->fault_flag) __hip_atomic_store(kp->fault_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // host-visible: the pass is wrong and says so
              break;
            }
          }
        }
        __syncthreads();
      }
    }
    CH_LAUNDER();
    CH_TS(2);
    // ================================================================ phase B: the k-loop (fdnn_gemm.hip's, ROT form)
    if (!FDNN_CHAIN_EARLY_W) {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) {
#pragma unroll
          for (int i = 0; i < NLD_W; ++i) stage_load(s, s, i);
        }
    }
    // the activation halves: stage 0 whole; ROT: stage 1's go out as step 0's refill (below)
#pragma unroll
    for (int s = 0; s < (ROT ? 1 : STAGES - 1); ++s)
      if (s < KT) {
#pragma unroll
        for (int i = NLD_W; i < NLD; ++i) stage_load(s, s, i);
      }
    asm volatile("" ::: "memory");

    asm volatile(".p2align 8");
    v4i a[2][2], b[2][NF];
    auto load_frags = [&](const char *wt_, const char *at_, int kk, int set) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[set][mi] = read_frag<BK>(wt_, 64 * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) b[set][ni] = read_frag<BK>(at_, arow0 + 32 * ni + frow, kk * 2 + fch);
    };
    if (ROT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage 0 (and the table, the biases, stage 1's weights) landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CH_TS(7);  // (stage 0 has landed)
      load_frags(smem, smem + Cfg::W_BYTES, 0, 0);
      if (KT > 1) {
#pragma unroll
        for (int i = 0; i < A_D0; ++i) stage_load(1, 1, NLD_W + i);
      }
    }

    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
      if (!ROT) {
        if (STAGES > 2 && kt + STAGES - 2 < KT) {
          static_assert((STAGES - 2) * Cfg::MIN_LOADS <= 63, "vmcnt is a 6-bit counter");
          if (kt == 0)  // issue order here: W0 .. W(S-2), A0 .. A(S-2): stage 0 has landed once only the younger A halves are out
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (Cfg::A_SLABS / NW)) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::MIN_LOADS) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      const bool refill = kt + STAGES - 1 < KT;
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      const char *wt = smem + buf * Cfg::STAGE;
      const char *at = wt + Cfg::W_BYTES;
      while (!NOFIX && fix_k_next < (kt + 1) * BK) {  // rare: a risky pair lives in this k-step (screen, then the exact correction)
        const int node = static_cast<int>(fix_raw >> 32) - fix_node0;         // 0..63
        const int kl = static_cast<int>(fix_raw & 0xffff) - kt * BK;          // even, 0..BK-2
        const int w0 = static_cast<int8_t>(fix_raw >> 16), w1 = static_cast<int8_t>(fix_raw >> 24);
        {
          const int wpk = (w0 & 0xff) | ((w1 & 0xff) << 8);
          const int pbase = 128 * (w0 + w1) + 32768;
          bool fire = false;
#pragma unroll
          for (int j = 0; j < (NF + 1) / 2; ++j) {
            const int row = arow0 + 64 * j + lane;
            const int v = *reinterpret_cast<const uint16_t *>(at + row * BK + (((kl >> 4) ^ swz<BK>(row)) << 4) + (kl & 15));
            const int ps = __builtin_amdgcn_sdot4(v, wpk, pbase, false);  // p + 32768
            const bool live = 64 * j + 64 <= 32 * NF || lane < 32 * NF - 64 * j;
            fire |= live && static_cast<unsigned>(ps) > 65535u;
          }
          if (__ballot(fire) == 0ull) {
            ++fix_e;
            fix_raw = fix_raw_nxt;
            fix_k_next = fix_e < fix_end ? static_cast<int>(fix_raw & 0xffff) : INT_MAX;
            if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
            continue;
          }
        }
        const int rr = node & 31;
        const int idx = (node >> 5) * 16 + (rr & 3) + 4 * (rr >> 3);  // mi*16 + reg
        const bool mine = (lane >> 5) == ((rr >> 2) & 1);
        int c[NF];
        uint32_t pair[NF];
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
          const int row = arow0 + 32 * ni + frow;
          pair[ni] = *reinterpret_cast<const uint16_t *>(at + row * BK + (((kl >> 4) ^ swz<BK>(row)) << 4) + (kl & 15));
        }
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) asm volatile("" : "+v"(pair[ni]));
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) {
          const int a0 = static_cast<int>((pair[ni] & 0xff) ^ 0x80), a1 = static_cast<int>((pair[ni] >> 8) ^ 0x80);  // back to u8
          const int prod = a0 * w0 + a1 * w1;
          c[ni] = mine ? max(-32768, min(32767, prod)) - prod : 0;
        }
        int nz = 0;
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) nz |= c[ni];
        if (__ballot(nz != 0) != 0ull) {
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            if (idx == i) {
#pragma unroll
              for (int ni = 0; ni < NF; ++ni) acc[i >> 4][ni][i & 15] += c[ni];
            }
          }
        }
        ++fix_e;
        fix_raw = fix_raw_nxt;
        fix_k_next = fix_e < fix_end ? static_cast<int>(fix_raw & 0xffff) : INT_MAX;
        if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
      }
      if (!ROT) load_frags(wt, at, 0, 0);
#pragma unroll
      for (int kk = 0; kk < SUB; ++kk) {
        if (kk + 1 < SUB) load_frags(wt, at, kk + 1, (kk + 1) & 1);
        if (ROT) {
          if (kk == 0) {
            if (kt == 0) {  // the rest of stage 1's activation half (its weight half went out before the wait)
              if (KT > 1) {
#pragma unroll
                for (int i = A_D0; i < NLD_A; ++i) stage_load(1, 1, NLD_W + i);
                if (!FDNN_CHAIN_EARLY_W) {
#pragma unroll
                  for (int i = 0; i < NLD_W; ++i) stage_load(1, 1, i);
                }
              }
            } else if (refill) {
#pragma unroll
              for (int i = ROT_D0; i < NLD; ++i) stage_load(kt + 1, nb, i);
            }
          }
          if (kk == SUB - 1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < KT) {
              const char *wn_ = smem + nb * Cfg::STAGE;
              load_frags(wn_, wn_ + Cfg::W_BYTES, 0, 0);
            }
            if (kt + 2 < KT) {
#pragma unroll
              for (int i = 0; i < ROT_D0; ++i) stage_load(kt + 2, buf, i);
            }
          }
        } else if (refill) {
#pragma unroll
          for (int i = 0; i < NLD; ++i)
            if (i * SUB / NLD == kk) stage_load(kt + STAGES - 1, nb, i);
        }
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni], 0, 0, 0);
      }
      if (++buf == STAGES) buf = 0;
    }
    CH_LAUNDER();
    CH_TS(3);
    // ================================================================ phase C: epilogue
    // the next task is drawn now (thread 0): the atomic's round trip runs under the epilogue
    uint32_t drawn = 0;
    if (tid == 0 && ctl_s[2] < 8) drawn = __hip_atomic_fetch_add(kp->ctl + ctl_s[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      // dequantise (validated 3-op division), + bias, half-step table, the s8 tile through LDS, whole rows out
      const Dec d = decode(kp, task);
      const int m0 = d.mt * G_BM, f0 = d.nt * FT;
      typedef const __attribute__((address_space(4))) QChainLayer *LP;
      const LP L = reinterpret_cast<LP>(&kp->layer[0]) + d.l;
      const float coef = L->coef, rcp_coef = L->rcp_coef;
      const uint8_t *lut = reinterpret_cast<const uint8_t *>(aux);
      const float *bias_s = reinterpret_cast<const float *>(aux + 3072);
      char *tile_s = smem + 8192;
      __syncthreads();  // every wave is done with the ring: it becomes the output tile
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nbl = 64 * wm + 32 * mi + 8 * g + 4 * half;  // 4 consecutive nodes of the tile
          const float4 b4 = *reinterpret_cast<const float4 *>(bias_s + nbl);
          const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
          uint8_t act[NF][4];
#pragma unroll
          for (int ni = 0; ni < NF; ++ni) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int av = acc[mi][ni][g * 4 + qq];
              const float lin = dequant<true>(av, coef, rcp_coef) + bj[qq];
              const int u = static_cast<int>(lin * 200.0f);  // RN(lin*200) = 2*RN(lin*100) exactly; trunc -> half-step index
              const int idx = max(-kLut2Half, min(kLut2Half, u)) + kLut2Half;
              act[ni][qq] = lut[idx];
            }
          }
#pragma unroll
          for (int ni = 0; ni < NF; ++ni) {
            const uint32_t packed = static_cast<uint32_t>(act[ni][0]) | static_cast<uint32_t>(act[ni][1]) << 8 |
                                    static_cast<uint32_t>(act[ni][2]) << 16 | static_cast<uint32_t>(act[ni][3]) << 24;
            *reinterpret_cast<uint32_t *>(tile_s + (arow0 + 32 * ni + frow) * kTS + nbl) = packed;
          }
        }
      }
      __syncthreads();
      CH_TS(4);
      int8_t *a_out = kp->act[(d.l & 1) ^ 1];
      const size_t lda = static_cast<size_t>(kp->lda);
      const int rows = kp->rows;
      for (int item = tid; item < FT * (G_BM / 16); item += Cfg::THREADS) {
        const int row = item / (G_BM / 16), ch = item % (G_BM / 16);
        const uint4 v = *reinterpret_cast<const uint4 *>(tile_s + row * kTS + ch * 16);
        if (m0 + ch * 16 < rows)  // rows is a multiple of 16
          store_wt(a_out + static_cast<size_t>(f0 + row) * lda + m0 + ch * 16,
                   v4i{static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)});
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my rows have been acknowledged (write-through: they are in memory)
    CH_TS(5);
    // ================================================================ phase D: next task, arrival
    int done_l = 0, done_nt = 0, done_target = 0;
    {
      const Dec d = decode(kp, task);
      done_l = d.l;
      done_nt = d.gnt;
      done_target = d.gtarget;
    }
    if (tid == 0) {
      int t;
      const int cur_q = ctl_s[1];
      if (ctl_s[2] < 8 && drawn < static_cast<uint32_t>(queue_tasks(kp, cur_q))) {
        t = (cur_q << 24) | static_cast<int>(drawn);
      } else {
        ctl_s[2] = 8;  // my queue is exhausted: from now on, whichever queue has the most left
        t = draw_slow(kp);
      }
      ctl_s[0] = t;
    }
    __syncthreads();  // everybody's rows are out; the ring is free; the next task is known
    if (tid == 0) {
      const int NL = kp->n_layers;
      uint32_t *dn = kp->done + static_cast<size_t>(done_nt) * NL;
      const uint32_t prev = __hip_atomic_fetch_add(dn + done_l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done_l == NL - 1 && prev == static_cast<uint32_t>(done_target) - 1u) {  // the frame tile's last task: its counters are ready for the next launch
        for (int i = 0; i < NL; ++i) __hip_atomic_store(dn + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#if FDNN_CHAIN_CLK
    CH_TS(6);
    if (tid == 0 && kp->clk) {
      long long *clk = kp->clk;
      const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t *>(clk), 1u);
      if (slot < static_cast<uint32_t>(kp->clk_cap)) {
        long long *o = clk + 8 + static_cast<size_t>(slot) * 10;
        o[0] = (static_cast<long long>(blockIdx.x) << 32) | static_cast<uint32_t>(task);
        o[1] = (static_cast<long long>(xcc_id()) << 32) | static_cast<uint32_t>((done_l << 16) | done_nt);
        o[2] = tc[7];  // stage 0 landed
        for (int i = 0; i < 7; ++i) o[3 + i] = tc[i];
      }
    }
#endif
    task = __builtin_amdgcn_readfirstlane(ctl_s[0]);
  }
  // ---- leaving: the last workgroup out rewinds the queue heads for the next launch
  if (tid == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(kp->ctl + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1u) {
      for (int i = 0; i < 9; ++i) __hip_atomic_store(kp->ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int NF, int WN, int BK, int STAGES, int WM = 4>
void launch_chain_cfg(const QChainParams &p, hipStream_t s) {
  using Cfg = GemmCfg<NF, WN, BK, STAGES, WM>;
  constexpr int kLds = Cfg::LDS + 64;
  static_assert(kLds <= 160 * 1024, "LDS");
  auto k = qchain_kernel<NF, WN, BK, STAGES, WM>;
  auto k_nofix = qchain_kernel<NF, WN, BK, STAGES, WM, true>;
  static std::atomic<unsigned long long> attr_set{0};
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_nofix), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    cus[dev & 63].store(n_cu, std::memory_order_relaxed);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  const int per_cu = (160 * 1024) / kLds >= 2 && Cfg::THREADS <= 256 ? 2 : 1;
  const long tiles = static_cast<long>(p.rows_pad / Cfg::G_BM) * (p.n_pad / Cfg::FT);
  const int grid = static_cast<int>(std::min<long>(tiles, static_cast<long>(cus[dev & 63].load(std::memory_order_relaxed)) * per_cu));
  bool any_fix = false;
  for (int i = 0; i < p.n_layers; ++i) any_fix = any_fix || p.layer[i].fix_ent != nullptr;
  hipLaunchKernelGGL(any_fix ? k : k_nofix, dim3(grid), dim3(Cfg::THREADS), kLds, s, p);
}

}  // namespace

int qchain_frame_tile(int rows_pad, int n);
static std::atomic<int> g_chain_mode{-1}, g_chain_min{0};
void qchain_set_mode(int mode, int min_frames) {
  g_chain_mode.store(mode, std::memory_order_relaxed);
  g_chain_min.store(min_frames, std::memory_order_relaxed);
}

bool qchain_ok(int rows_pad, int K, int n, int n_layers) {
  static const int env_mode = [] {
    const char *e = std::getenv("FDNN_CHAIN");
    return e ? std::atoi(e) : -1;
  }();
  static const int env_min = [] {
    const char *e = std::getenv("FDNN_CHAIN_MIN");
    return e ? std::atoi(e) : 9800;
  }();
  const int forced = g_chain_mode.load(std::memory_order_relaxed), forced_min = g_chain_min.load(std::memory_order_relaxed);
  // From ~9 800 frames up -- a round of 256-node x 320-frame tiles and more -- the chain is the faster form at EVERY size
  // (tools/chain_sweep.py, layer 0 + six hidden layers, chained / per-layer: 10 000 frames 0.98, 10 241 0.88, 12 000 0.86,
  // 15 360 0.93, 20 480 0.97): its tasks flow across the layers where a launch per layer idles most of the chip in every
  // partial round.  Below, the per-layer path has better tiles for the size (160- / 128-frame four-wave shapes, two workgroups
  // per CU) and the chain's 320-frame tasks leave CUs without work: 9 000 frames 1.04, 8 000 1.21, 6 000 1.33, 4 097 1.33.
  const int mode = forced >= 0 ? forced : env_mode;  // 0: never; otherwise from min_frames up
  const int min_frames = (forced >= 0 && forced_min > 0) ? forced_min : env_min;
  if (mode == 0 || n_layers < 2 || n_layers > kMaxChainLayers || K % 128 != 0 || n < min_frames) return false;
  if (forced == 1) return true;  // (tests / measurements: wherever the shape allows)
  // ... and only where a launch per layer would run a partly filled round: at whole rounds of 320-frame tiles (10 000 /
  // 10 240 / 20 480 frames on a 2048-wide net) the two forms are within +-2 % of each other, the sign depending on the box.
  static const long cus = [] {  // (advisor, round 5: not a hard-coded 256)
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    return static_cast<long>(n_cu);
  }();
  const long tiles = static_cast<long>(rows_pad / 256) * ((n + 319) / 320);
  const long idle = (tiles + cus - 1) / cus * cus - tiles;
  return idle >= 24;
}

int qchain_frame_tile(int rows_pad, int n) {
  (void)rows_pad;
  static const int forced = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_CHAIN_TILE");
    return e ? std::atoi(e) : 0;
  }();
  if (forced == 256 || forced == 320) return forced;
  // 320-frame tiles unless the padding they add is worth more than their better operand reuse
  const int pad320 = (n + 319) / 320 * 320 - n, pad256 = (n + 255) / 256 * 256 - n;
  return pad256 + 64 < pad320 ? 256 : 320;
}

void launch_qchain(const QChainParams &p, hipStream_t s) {
  if (p.frame_tile == 256)
    launch_chain_cfg<4, 2, 128, 2>(p, s);
  else
    launch_chain_cfg<5, 2, 128, 2>(p, s);
}

}  // namespace fdnn
