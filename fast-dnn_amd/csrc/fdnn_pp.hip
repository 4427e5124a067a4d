// fdnn_pp.hip -- an int8 HIDDEN layer of a large batch with the two waves of every SIMD in DIFFERENT roles (gfx950).
//
//   QuantizedLayerActivations / quantizedNodeSum (dnn.cc:289-349) + AddBias (:250-264) + QuantizedSigmoid (:267-286)
//
// Why another kernel.  In fdnn_gemm.hip / fdnn_chain.hip the two waves of a SIMD are in the same phase at all times: both
// in the k-loop (each paying 9 LDS-DMA issues of 60-185 cycles per k-step inside its own MFMA stream), then both in the
// epilogue (14 k cycles of vector work with the matrix pipe idle).  tools/ubench_role.hip measured what the pipe does when
// ONE wave per SIMD issues nothing but fragment reads and MFMAs (one ds_read_b128 pinned behind each of a sub-step's first
// seven MFMAs): 0.97 of the issue rate alone, 0.92 with its partner issuing every LDS-DMA load of the tile, 0.82 with 104
// vector instructions per k-step on top -- against 0.75 for the in-phase k-loop and 0.48 for the in-phase tile as a whole.
//
// Structure.  One 512-thread workgroup per CU = two GROUPS of four waves (waves w and w + 4 share a SIMD).  A workgroup's
// task is a 256-node x 320-frame tile; group g owns its half g (160 frames).  The groups ALTERNATE: in a PHASE (16
// k-steps = "ticks", one s_barrier each) one group is in the
//   COMPUTE role: the k-loop of its half -- fragment reads, MFMAs, the pair-saturation walk -- plus two ds_read_b128 and
//                 two global stores every third tick (the partner's finished rows, below); and the other in the
//   SUPPORT role: every LDS-DMA load of the compute group's operands (13 one-KiB pieces per wave and tick: the activation
//                 rows one tick ahead into a 2-deep ring, the weights two ticks ahead into a 3-deep ring), the EPILOGUE of
//                 the half it computed in the phase before (dequantise, + bias, half-step table: 40 items of four outputs
//                 per lane, 8 per 3 ticks) and the first stages of the half it computes next.
// Only one group stages operands at a time, so the ring holds one half's operands: 3 x 32 KB + 2 x 20 KB.  The support
// group parks each finished 32-frame block as bytes in LDS; after the tick's barrier the COMPUTE group writes it out as
// whole 256-byte row segments: its vector-memory queue is otherwise empty, whereas a store in a support wave's queue would
// sit in front of the loads that wave's counted s_waitcnt vmcnt waits for (loads and stores retire in order).
// Schedule of a workgroup with tiles T0, T1, ...: [prepare T0.0] | T0.0 | T0.1 + epilogue T0.0 | T1.0 + epilogue T0.1 |
// ... | epilogue of the last half (nothing left to hide it under).  All of it is static: no queue, no poll, no flag.
//
// The tile arithmetic is fdnn_gemm.hip's operation for operation (the epilogue is the validated 3-operation division +
// bias + half-step table; the pair-saturation walk is its walk): identical bytes by construction and by test
// (tests/test_gpu_pp.py).
#include <atomic>
#include <climits>
#include <cstdlib>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"
#include "fdnn_tile.hpp"

#ifndef FDNN_PP_CLK
#define FDNN_PP_CLK 0  // 1 (measurement builds): printf of per-phase clocks from a few workgroups
#endif
#ifndef FDNN_PP_PIPE
#define FDNN_PP_PIPE 0  // 2: the epilogue's three stages alternate with the tick's weight pieces (the pieces' issue time covers the LDS round trips); 1: a tick's table gathers are packed and parked in the NEXT tick (they return under the barrier); 0: in the same tick
#endif
#ifndef FDNN_PP_DEBUG
#define FDNN_PP_DEBUG 0  // timing experiments: 1 no epilogue arithmetic, 2 no MFMAs, 4 no operand loads, 8 no stores
#endif

namespace fdnn {
namespace {

[[maybe_unused]] constexpr int kNF = 5, kBK = 128, kBM = 256, kHT = 32 * kNF, kFT = 2 * kHT, kKT = 16, kTS = kBM + 16;
[[maybe_unused]] constexpr int kWStages = 3;
[[maybe_unused]] constexpr int kLutBytes = (kLut2Size + 15) & ~15;

template <bool NOFIX>
__global__ __launch_bounds__(512, 2) void qpp_kernel(QGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Separate static arrays: the compiler then knows that the epilogue's LDS traffic cannot alias the LDS-DMA destinations;
  // through one extern block every ds_read waits for every load in flight (tools/ubench_role.hip, first run: 52 k cycles a tick).
  __shared__ __attribute__((aligned(16))) uint8_t lut_s[kLutBytes];
  __shared__ __attribute__((aligned(16))) char ringW[kWStages][kBM * kBK];
  __shared__ __attribute__((aligned(16))) char ringA[2][kHT * kBK];
  __shared__ __attribute__((aligned(16))) char tile_s[2][32 * kTS];  // finished 32-frame blocks, bytes, [frame][node]
  __shared__ __attribute__((aligned(16))) char wsum_s[kBM * 4];      // the next half's accumulator start values (LDS-DMA; read with inline asm only)
  __shared__ __attribute__((aligned(16))) float bias_s[2][kBM];      // per group: its tile's biases (plain stores, plain loads: never an LDS-DMA target)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, grp = wave >> 2;
  const int frow = lane & 31, fch = lane >> 5, half = lane >> 5;
  // Scalar registers: the parameters are re-read through the kernel-argument pointer where they are used (scalar loads from
  // the kernarg segment) instead of living in SGPRs across the phases -- by value, 211 of them were spilled into VGPR lanes.
  typedef const __attribute__((address_space(4))) QGemmParams *KP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)p;
#define p (*kp)

  // ---- the workgroup's tiles: t = blockIdx, + gridDim, ...  Workgroup b runs on XCD b % 8 (a locality hint, as in
  // fdnn_gemm.hip): XCD x owns the frame pairs x, x + 8, ... and walks THOSE fastest, so its activation rows stay in its
  // L2 and a weight tile is pulled into an XCD once.  The only holes are at the end of an XCD's list.
  const int MT = p.rows_pad / kBM, NP = p.n_pad / kFT;
  auto tile_of = [&](int i, int &mt, int &pair) -> bool {  // the i-th tile of this workgroup
    const int t = static_cast<int>(blockIdx.x) + i * static_cast<int>(gridDim.x);
    const int x = t & 7, u = t >> 3;
    const int npx = (NP - x + 7) >> 3;  // pairs of XCD x
    if (u >= npx * MT) return false;
    mt = u / npx;
    pair = x + 8 * (u - mt * npx);
    return true;
  };
  int n_tiles = 0;
  {
    int a_, b_;
    while (tile_of(n_tiles, a_, b_)) ++n_tiles;
  }
  if (n_tiles == 0) return;

  // ---- prologue: the half-step table (plain loads: no LDS-DMA ever targets lut_s)
  for (int i = tid; i < kLutBytes / 16; i += 512)
    *reinterpret_cast<uint4 *>(lut_s + 16 * i) = *reinterpret_cast<const uint4 *>(p.lut2 + 16 * i);
  __syncthreads();


  v16i acc[2][kNF];
  float bias_v = 0.0f;  // this lane's bias of the half its group computes next: requested in the last support tick, parked in bias_s at the start of the compute phase
#if FDNN_PP_CLK
  long long pclk = __builtin_readcyclecounter();
  long long pcl[12];
  int npcl = 0;
#endif

  // phase ph (>= 0): group ph & 1 computes half ph & 1 of tile ph >> 1; phase -1: group 0 only prepares tile 0's half 0;
  // after the last phase group 1 runs its last epilogue on its own (no barriers, its own stores)
  const int n_ph = 2 * n_tiles;
  for (int ph = -1; ph < n_ph; ++ph) {
    const int cg = ph & 1, sg = cg ^ 1;
    const bool cvalid = ph >= 0;                      // the compute group has a half to compute
    const bool evalid = ph >= 1;                      // the support group has an epilogue to run (the half of phase ph - 1)
    const bool nvalid = ph + 1 < n_ph;                // the support group computes in phase ph + 1
    const int kt0 = (cvalid || evalid) ? 0 : kKT - 2;  // (phase -1: only the last two ticks do anything)
    const int gt0 = (ph + 1) * kKT;                   // global tick of this phase's tick 0: stage of tick T lives in ringW[T % 3], ringA[T & 1]
    int c_mt = 0, c_pair = 0, e_mt = 0, e_pair = 0, n_mt = 0, n_pair = 0;
    if (cvalid) tile_of(ph >> 1, c_mt, c_pair);
    if (evalid) tile_of((ph - 1) >> 1, e_mt, e_pair);
    if (nvalid) tile_of((ph + 1) >> 1, n_mt, n_pair);
#if FDNN_PP_CLK
    if (tid == 0) {
      const long long now = __builtin_readcyclecounter();
      if (npcl < 12) pcl[npcl++] = now - pclk;
      pclk = now;
    }
#endif

    // (lane-derived values that only one phase needs are formed from a laundered copy: hoisted out of the phase loop they
    // stayed live across every k-loop and two of them were spilled -- with a scratch reload and its vmcnt(0) in a support tick)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (grp == cg) {
      // ============================================================== COMPUTE role
      int fix_e = 0, fix_end = 0, fix_k_next = INT_MAX;
      typedef const __attribute__((address_space(4))) uint64_t *FixPtr;
      const FixPtr ent_c = (FixPtr)(uintptr_t)p.fix_ent;
      uint64_t fix_raw = 0, fix_raw_nxt = 0;
      const int fix_node0 = c_mt * kBM + 64 * wm;
      // my share of the stages I requested in my last support ticks (the second weight stage) has landed.  The BUILTIN, not
      // inline asm: the compiler's wait-count pass must see this wait -- otherwise it protects the first ring read of the
      // k-loop against the LDS-DMA loads of the support path with a vmcnt(0) of its own INSIDE the loop, where it then waits
      // for the store duty's stores tick after tick.
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      asm volatile("" ::: "memory");
      // (... and so has my bias: each wave parks its own 64 -- plain stores; only this wave reads them, in its next support phase)
      if (cvalid) bias_s[grp][64 * wm + ln] = bias_v;
      if (cvalid) {
        // accumulators start at 128 * sum_k w (wsum_s: requested by this group two ticks before its support phase ended)
        const uint32_t wa = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(FDNN_LDS_PTR(wsum_s))) + (64 * wm + 4 * half) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            v4i ws4;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(ws4) : "v"(wa), "n"((32 * a + 8 * g) * 4) : "memory");
#pragma unroll
            for (int b = 0; b < kNF; ++b) {
              acc[a][b][g * 4 + 0] = ws4.x;
              acc[a][b][g * 4 + 1] = ws4.y;
              acc[a][b][g * 4 + 2] = ws4.z;
              acc[a][b][g * 4 + 3] = ws4.w;
            }
          }
        if (!NOFIX && p.fix_ent) {
          const int gi = (c_mt * kBM >> 6) + wm;
          fix_e = __builtin_amdgcn_readfirstlane(p.fix_grp[gi]);
          fix_end = __builtin_amdgcn_readfirstlane(p.fix_grp[gi + 1]);
          if (fix_e < fix_end) {
            fix_raw = ent_c[fix_e];
            if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
            fix_k_next = static_cast<int>(fix_raw & 0xffff);
          }
        }
      }
      // where the partner's finished blocks go (store duty): 16-byte chunk c of a 32 x 256-byte block; second pass c + 64.
      // Buffer stores through the compiler (write-through: sc0 sc1), not inline asm: with the store hidden in an asm block the
      // first dword of every first-pass chunk of the tile's last 64 columns came out wrong whenever the k-loop ran beside it
      // (first GPU run of this kernel) -- a hazard the compiler's recognizer handles for the instructions it can see.
      const int st_c = 128 * wm + ln;
      const int st_m0 = e_mt * kBM;
      const bool st_cols = evalid && st_m0 + 16 * (st_c & 15) < p.rows;  // (rows is a multiple of 16; chunk c + 64 has the same column)
      const int st_voff = (st_c >> 4) * p.act_ld + st_m0 + 16 * (st_c & 15);
      const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          p.act_out + static_cast<size_t>(e_pair * kFT + sg * kHT) * p.act_ld, 0, kHT * p.act_ld, 0x00020000);

      v4i fa[2][2], fb[2][kNF];
      auto load_frags = [&](int T, int kk, int set) {
        const char *wt = ringW[T % kWStages], *at = ringA[T & 1];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[set][mi] = read_frag<kBK>(wt, 64 * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
        for (int ni = 0; ni < kNF; ++ni) fb[set][ni] = read_frag<kBK>(at, 32 * ni + frow, kk * 2 + fch);
      };
      auto mfmas = [&](int set) {
#if !(FDNN_PP_DEBUG & 2)
#pragma unroll
        for (int ni = 0; ni < kNF; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][mi], fb[set][ni], acc[mi][ni], 0, 0, 0);
#else
        (void)set;
#endif
      };
      // one fragment read behind each of the sub-step's first seven MFMAs (tools/ubench_role.hip: 0.97 of the issue rate from
      // one wave; with the reads in front of the MFMAs 0.82, in the compiler's own order 0.77)
      auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
      };

      if (cvalid) load_frags(gt0, 0, 0);
      for (int kt = kt0; kt < kKT; ++kt) {
        const int T = gt0 + kt;
        if (cvalid) {
          const char *at = ringA[T & 1];
          while (!NOFIX && fix_k_next < (kt + 1) * kBK) {  // rare: a risky pair lives in this k-step (fdnn_gemm.hip's walk)
            const int node = static_cast<int>(fix_raw >> 32) - fix_node0;
            const int kl = static_cast<int>(fix_raw & 0xffff) - kt * kBK;
            const int w0 = static_cast<int8_t>(fix_raw >> 16), w1 = static_cast<int8_t>(fix_raw >> 24);
            {
              const int wpk = (w0 & 0xff) | ((w1 & 0xff) << 8);
              const int pbase = 128 * (w0 + w1) + 32768;
              bool fire = false;
#pragma unroll
              for (int j = 0; j < (kNF + 1) / 2; ++j) {
                const int row = 64 * j + lane;
                const bool live = row < kHT;
                const int rr_ = live ? row : 0;
                const int v = *reinterpret_cast<const uint16_t *>(at + rr_ * kBK + (((kl >> 4) ^ swz<kBK>(rr_)) << 4) + (kl & 15));
                const int ps = __builtin_amdgcn_sdot4(v, wpk, pbase, false);  // p + 32768
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
            const int idx = (node >> 5) * 16 + (rr & 3) + 4 * (rr >> 3);  // mi * 16 + reg
            const bool mine = (lane >> 5) == ((rr >> 2) & 1);
            int c[kNF];
            uint32_t pr[kNF];
#pragma unroll
            for (int ni = 0; ni < kNF; ++ni) {
              const int row = 32 * ni + frow;
              pr[ni] = *reinterpret_cast<const uint16_t *>(at + row * kBK + (((kl >> 4) ^ swz<kBK>(row)) << 4) + (kl & 15));
            }
#pragma unroll
            for (int ni = 0; ni < kNF; ++ni) asm volatile("" : "+v"(pr[ni]));
#pragma unroll
            for (int ni = 0; ni < kNF; ++ni) {
              const int a0 = static_cast<int>((pr[ni] & 0xff) ^ 0x80), a1 = static_cast<int>((pr[ni] >> 8) ^ 0x80);  // back to u8
              const int prod = a0 * w0 + a1 * w1;
              c[ni] = mine ? max(-32768, min(32767, prod)) - prod : 0;
            }
            int nz = 0;
#pragma unroll
            for (int ni = 0; ni < kNF; ++ni) nz |= c[ni];
            if (__ballot(nz != 0) != 0ull) {
#pragma unroll
              for (int i = 0; i < 32; ++i) {
                if (idx == i) {
#pragma unroll
                  for (int ni = 0; ni < kNF; ++ni) acc[i >> 4][ni][i & 15] += c[ni];
                }
              }
            }
            ++fix_e;
            fix_raw = fix_raw_nxt;
            fix_k_next = fix_e < fix_end ? static_cast<int>(fix_raw & 0xffff) : INT_MAX;
            if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
          }
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            load_frags(T, kk + 1, (kk + 1) & 1);
            mfmas(kk & 1);
            interleave();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage's fragments are all in registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (cvalid && kt + 1 < kKT) load_frags(T + 1, 0, 0);
        if (evalid && kt % 3 == 0 && kt > 0) {
          // the partner's 32-frame block kt / 3 - 1 is complete in tile_s[(kt / 3 - 1) & 1]: out as 256-byte row segments
          const int ni = kt / 3 - 1;
          const char *src = tile_s[ni & 1];
          typedef unsigned int v4u __attribute__((ext_vector_type(4)));
          const v4u v0 = *reinterpret_cast<const v4u *>(src + (st_c >> 4) * kTS + 16 * (st_c & 15));
          const v4u v1 = *reinterpret_cast<const v4u *>(src + ((st_c >> 4) + 4) * kTS + 16 * (st_c & 15));
#if !(FDNN_PP_DEBUG & 8)
          if (st_cols) {
            __builtin_amdgcn_raw_buffer_store_b128(v0, st_rsrc, st_voff, 32 * ni * p.act_ld, 17);
            __builtin_amdgcn_raw_buffer_store_b128(v1, st_rsrc, st_voff, (32 * ni + 4) * p.act_ld, 17);
          }
#endif
        }
        if (cvalid) {
          mfmas(1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // ============================================================== SUPPORT role
      // (descriptors are built unconditionally -- an invalid phase's indices are 0 and its loads are never issued)
      const __amdgpu_buffer_rsrc_t rw =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.w + static_cast<size_t>(c_mt * kBM) * p.ldw), 0, kBM * p.ldw, 0x00020000);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<int8_t *>(p.a + static_cast<size_t>(c_pair * kFT + cg * kHT) * p.lda), 0, kHT * p.lda, 0x00020000);
      auto rw_next = [&]() {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.w + static_cast<size_t>(n_mt * kBM) * p.ldw), 0, kBM * p.ldw, 0x00020000);
      };
      auto ra_next = [&]() {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.a + static_cast<size_t>(n_pair * kFT + sg * kHT) * p.lda), 0, kHT * p.lda,
                                                 0x00020000);
      };
      // my epilogue reads its biases from bias_s[grp] (each wave parked its own 64 there with plain stores in the last tick of
      // its previous support phase): plain LDS loads that the compiler schedules with the table gathers.  With one inline-asm
      // read + wait per item -- as long as bias_s was an LDS-DMA target, which the compiler must not see read -- the epilogue
      // cost 33 k cycles a phase: an LDS round trip beside the partner's fragment reads and the LDS-DMA writes is ~400 cycles.
      const float *bias_l = bias_s[grp] + 64 * wm + 4 * (ln >> 5);
      const int tile_w = (ln & 31) * kTS + 64 * wm + 4 * (ln >> 5);  // this lane's byte position in a parked block (+ 32 mi + 8 g)
      // (the strides are laundered per phase: left visible, the 16 x 13 scalar offsets of the unrolled ticks are hoisted out of
      // the phase loop and 190 SGPRs spill into VGPR lanes)
      int ldw_s = p.ldw, lda_s = p.lda;
      asm volatile("" : "+s"(ldw_s), "+s"(lda_s));
      // per-lane part of the LDS-DMA addresses (four waves take slabs wm, wm + 4, ...: the slab's parity is wm & 1); formed per
      // phase from the laundered strides -- kept across the compute phases they were the two registers that spilled
      const int srow = ln >> 3;
      const int schunk = ((ln & 7) ^ swz<kBK>(wm * 8 + srow)) << 4;
      const int voff_w = srow * ldw_s + schunk;
      const int voff_a = srow * lda_s + schunk;
      auto stage_w = [&](__amdgpu_buffer_rsrc_t r, int chunk, int buf, int i) {  // piece i (0..7) of a weight stage
        const int slab = i * 4 + wm;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(&ringW[buf][slab * 1024]), 16, voff_w, slab * 8 * ldw_s + chunk * kBK, 0, 0);
      };
      auto stage_a = [&](__amdgpu_buffer_rsrc_t r, int chunk, int buf, int i) {  // piece i (0..4) of an activation stage
        const int slab = i * 4 + wm;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(&ringA[buf][slab * 1024]), 16, voff_a, slab * 8 * lda_s + chunk * kBK, 0, 0);
      };
      // The epilogue, software-pipelined over the ticks (item = four consecutive nodes of one frame; fdnn_chain.hip's
      // arithmetic).  Tick t: the bytes gathered in tick t - 1 are packed and parked; the items of tick t go through the
      // arithmetic and their table gathers are ISSUED -- they return under the tick's weight pieces, its wait and the barrier.
      // (An LDS round trip beside the partner's fragment reads and the LDS-DMA writes is ~400 cycles; unpipelined, two of
      // them per item, the epilogue cost 33 k cycles a phase; one per tick, 10 k.)
      uint32_t gb[3][4];  // gathered table bytes of the previous tick's items
      auto epi_bias = [&](v4f_t (&bq)[3], int it0, int it1) {
#pragma unroll
        for (int it = it0; it < it1; ++it) bq[it - it0] = *reinterpret_cast<const v4f_t *>(bias_l + 32 * ((it >> 2) & 1) + 8 * (it & 3));
      };
      auto epi_math = [&](const v4f_t (&bq)[3], int it0, int it1) {
#if !(FDNN_PP_DEBUG & 1)
        int idx[3][4];
#pragma unroll
        for (int it = it0; it < it1; ++it) {
          const int ni = it >> 3, mi = (it >> 2) & 1, g = it & 3;
          const float bj[4] = {bq[it - it0].x, bq[it - it0].y, bq[it - it0].z, bq[it - it0].w};
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int av = acc[mi][ni][g * 4 + qq];
            const float lin = dequant<true>(av, p.coef, p.rcp_coef) + bj[qq];
            const int u = static_cast<int>(lin * 200.0f);  // RN(lin * 200) = 2 RN(lin * 100) exactly; trunc -> half-step index
            idx[it - it0][qq] = max(-kLut2Half, min(kLut2Half, u)) + kLut2Half;
          }
        }
#pragma unroll
        for (int it = it0; it < it1; ++it)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) gb[it - it0][qq] = lut_s[idx[it - it0][qq]];
#else
        (void)bq;
        (void)it0;
        (void)it1;
#endif
      };
      auto epi_park = [&](int it0, int it1) {
#if !(FDNN_PP_DEBUG & 1)
#pragma unroll
        for (int it = it0; it < it1; ++it) {
          const int ni = it >> 3, mi = (it >> 2) & 1, g = it & 3;
          const uint32_t packed = gb[it - it0][0] | gb[it - it0][1] << 8 | gb[it - it0][2] << 16 | gb[it - it0][3] << 24;
          *reinterpret_cast<uint32_t *>(tile_s[ni & 1] + tile_w + 32 * mi + 8 * g) = packed;
        }
#else
        (void)it0;
        (void)it1;
#endif
      };

#pragma unroll
      for (int kt = 0; kt < kKT; ++kt) {
        if (kt >= kt0) {
          const int T = gt0 + kt;
          bool wq = false;  // this tick ends with eight weight pieces that its wait leaves in flight
          constexpr int kSched[17] = {0, 3, 6, 8, 11, 14, 16, 19, 22, 24, 27, 30, 32, 35, 38, 40, 40};
          if (FDNN_PP_PIPE == 1 && evalid && kt >= 1) epi_park(kSched[kt - 1], kSched[kt]);  // (tick 15 parks the last two items)
          // ---- (a) the activation rows of the next tick (needed at this tick's barrier: first in the queue)
#if !(FDNN_PP_DEBUG & 4)
          if (kt < kKT - 1) {
            if (cvalid) {
#pragma unroll
              for (int i = 0; i < 5; ++i) stage_a(ra, kt + 1, (T + 1) & 1, i);
            }
          } else if (nvalid) {
            const __amdgpu_buffer_rsrc_t ra_n = ra_next();
#pragma unroll
            for (int i = 0; i < 5; ++i) stage_a(ra_n, 0, (T + 1) & 1, i);
          }
          // last tick (my epilogue has read its last biases): my next half's accumulator start values and biases (1 KiB each, one wave each)
          if (kt == kKT - 1 && nvalid) {
            if (wm == 0) {
              const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.wsum + n_mt * kBM), 0, kBM * 4, 0x00020000);
              __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(wsum_s), 16, ln * 16, 0, 0, 0);
            }
            bias_v = p.bias[n_mt * kBM + 64 * wm + ln];  // (parked in bias_s at the start of my compute phase, behind the wait I make there anyway)
          }
#endif
          // ---- (b) my epilogue: 8 items per 3 ticks (3, 3, 2), parked a tick later: block ni is complete before the barrier of tick 3 ni + 3
          // PIPE 2: bias reads | weight pieces 0..3 | arithmetic, gathers | pieces 4..7 | pack, park -- a wave issues in order, and
          // the pieces' issue time (~80 cycles each) is what covers the two LDS round trips
          const bool do_epi = evalid && kt < 15;
          v4f_t bq[3];
          if (do_epi && FDNN_PP_PIPE != 3) epi_bias(bq, kSched[kt], kSched[kt + 1]);
          if (do_epi && FDNN_PP_PIPE < 2) {
            epi_math(bq, kSched[kt], kSched[kt + 1]);
            if (!FDNN_PP_PIPE) epi_park(kSched[kt], kSched[kt + 1]);
          }
          // ---- (c) the weights two ticks ahead (youngest in the queue: the tick's wait leaves them in flight)
#if !(FDNN_PP_DEBUG & 4)
          const bool w_cur = kt < kKT - 2 && cvalid, w_nxt = kt >= kKT - 2 && nvalid;
          wq = w_cur || w_nxt;
          const __amdgpu_buffer_rsrc_t rw_t = w_nxt ? rw_next() : rw;
          const int w_chunk = kt < kKT - 2 ? kt + 2 : kt - (kKT - 2);
          __builtin_amdgcn_sched_barrier(0);
          if (wq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_w(rw_t, w_chunk, (T + 2) % kWStages, i);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (do_epi && FDNN_PP_PIPE >= 2) {
            if (FDNN_PP_PIPE == 3) epi_bias(bq, kSched[kt], kSched[kt + 1]);
            epi_math(bq, kSched[kt], kSched[kt + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (wq) {
#pragma unroll
            for (int i = 4; i < 8; ++i) stage_w(rw_t, w_chunk, (T + 2) % kWStages, i);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (do_epi && FDNN_PP_PIPE >= 2) epi_park(kSched[kt], kSched[kt + 1]);
#endif
          // everything but the youngest weight stage (first tick: also the stores of my compute phase).  The BUILTIN: with the wait
          // hidden in inline asm the compiler's own bookkeeping never sees a load retire, and it protects every reuse of a register
          // that a (scratch) load once wrote with a vmcnt wait of its own -- in front of this tick's pieces, i.e. a full stall.
          if (wq) __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8)
          else __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
          asm volatile("" ::: "memory");
          // my block bytes (parked at the top of the tick) are in tile_s: LDS operations complete in order, so at most the 12
          // gathers issued since may still be out (scalar loads share the counter and return out of order, but every completion
          // that is not one of them is an older LDS operation: the count still guarantees the stores)
          if (FDNN_PP_PIPE == 1) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
    }
  }
  // ---- group 1's last epilogue: nothing left to hide it under, nobody left to synchronise with.  Each wave parks its own
  // 64 columns of a block and writes them out itself (64-byte row segments): no barrier, and group 0's waves are gone.
  if (grp == 1) {
    int e_mt = 0, e_pair = 0;
    tile_of(n_tiles - 1, e_mt, e_pair);
    const int m0 = e_mt * kBM;
    const float *bias_l = bias_s[1] + 64 * wm + 4 * half;
    char *mine = tile_s[0] + 64 * wm;  // my 64 columns of the 32 parked rows
    const int c = lane;                // 16-byte chunk of my 32 x 64-byte slice: row c >> 2 (second pass + 16), chunk c & 3
    const bool cols = m0 + 64 * wm + 16 * (c & 3) < p.rows;
    int8_t *dst = p.act_out + static_cast<size_t>(e_pair * kFT + kHT + (c >> 2)) * p.act_ld + m0 + 64 * wm + 16 * (c & 3);
#pragma unroll
    for (int ni = 0; ni < kNF; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const v4f_t b4 = *reinterpret_cast<const v4f_t *>(bias_l + 32 * mi + 8 * g);
          const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
          uint32_t packed = 0;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float lin = dequant<true>(acc[mi][ni][g * 4 + qq], p.coef, p.rcp_coef) + bj[qq];
            const int u = static_cast<int>(lin * 200.0f);
            packed |= static_cast<uint32_t>(lut_s[max(-kLut2Half, min(kLut2Half, u)) + kLut2Half]) << (8 * qq);
          }
          *reinterpret_cast<uint32_t *>(mine + frow * kTS + 32 * mi + 8 * g + 4 * half) = packed;
        }
      typedef unsigned int v4u __attribute__((ext_vector_type(4)));
      const v4u v0 = *reinterpret_cast<const v4u *>(mine + (c >> 2) * kTS + 16 * (c & 3));
      const v4u v1 = *reinterpret_cast<const v4u *>(mine + ((c >> 2) + 16) * kTS + 16 * (c & 3));
#if !(FDNN_PP_DEBUG & 8)
      if (cols) {
        store_wt(dst + static_cast<size_t>(32 * ni) * p.act_ld, v4i{static_cast<int>(v0.x), static_cast<int>(v0.y), static_cast<int>(v0.z), static_cast<int>(v0.w)});
        store_wt(dst + static_cast<size_t>(32 * ni + 16) * p.act_ld, v4i{static_cast<int>(v1.x), static_cast<int>(v1.y), static_cast<int>(v1.z), static_cast<int>(v1.w)});
      }
#endif
    }
  }
#if FDNN_PP_CLK
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 131)) {
    for (int i = npcl; i < 12; ++i) pcl[i] = 0;
    printf("PP %3d tiles %d: pre %lld | ph-1 %lld | %lld %lld %lld %lld %lld %lld %lld %lld | last %lld\n", blockIdx.x, n_tiles, pcl[0], pcl[1], pcl[2], pcl[3], pcl[4],
           pcl[5], pcl[6], pcl[7], pcl[8], pcl[9], static_cast<long long>(__builtin_readcyclecounter() - pclk));
  }
#endif
#undef p
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

static std::atomic<int> g_pp_mode{-1}, g_pp_min{0};
void qpp_set_mode(int mode, int min_frames) {
  g_pp_mode.store(mode, std::memory_order_relaxed);
  g_pp_min.store(min_frames, std::memory_order_relaxed);
}

// The role-split kernel serves the production shape (K = 2048: 16 ticks per phase, 256-node tiles, validated 3-operation
// division).  Where it pays today (profiles/LABBOOK.md, round 6): layers WITHOUT saturating pairs from two tiles per
// workgroup up (16 384 frames on a 2048-wide net: 463 vs 480 us for six layers at 20 000 frames; 260 vs 258 at 10 000, where a
// workgroup has one tile and nothing hides the second half's epilogue).  With the pair-saturation walk in the compute
// role's instruction stream -- one wave per SIMD, nothing to cover its latency chain -- it loses (328 vs 279 us on the
// Gaussian bench net): such layers keep fdnn_gemm.hip's in-phase tiles unless forced (fdnn_debug_set_pp(1, n), FDNN_PP=1).
bool qpp_ok(int rows_pad, int K, int n, bool fastdiv, bool has_fix) {
  static const int env_mode = [] {
    const char *e = std::getenv("FDNN_PP");
    return e ? std::atoi(e) : -1;
  }();
  static const int env_min = [] {
    const char *e = std::getenv("FDNN_PP_MIN");
    return e ? std::atoi(e) : 16384;
  }();
  const int forced = g_pp_mode.load(std::memory_order_relaxed), forced_min = g_pp_min.load(std::memory_order_relaxed);
  const int mode = forced >= 0 ? forced : env_mode;
  const int min_frames = (forced >= 0 && forced_min > 0) ? forced_min : env_min;
  if (mode == 0 || !fastdiv || K != kKT * kBK || rows_pad % kBM != 0) return false;
  if (mode != 1 && has_fix) return false;
  return n >= min_frames;
}

int qpp_frame_tile() { return kFT; }

void launch_qpp_hidden(const QGemmParams &p, hipStream_t s) {
  auto k = qpp_kernel<false>;
  auto k_nofix = qpp_kernel<true>;
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n_cu = cus[dev & 63].load(std::memory_order_relaxed);
  if (n_cu == 0) {
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    cus[dev & 63].store(n_cu, std::memory_order_relaxed);
  }
  const int MT = p.rows_pad / kBM, NP = p.n_pad / kFT;
  const long t_end = 8L * ((NP + 7) / 8) * MT;  // (a multiple of 8: workgroup b and its later tiles b + grid, ... stay on one XCD's list)
  const int grid = static_cast<int>(std::min<long>(t_end, std::max(8, n_cu / 8 * 8)));
  hipLaunchKernelGGL(p.fix_ent ? k : k_nofix, dim3(grid), dim3(512), 0, s, p);
}

}  // namespace fdnn
