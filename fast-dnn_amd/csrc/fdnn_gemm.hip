// fdnn_gemm.hip -- the int8 layer kernel (MFMA 32x32x32 i8) for gfx950.
//
//   QuantizedLayerActivations/quantizedNodeSum + AddBias + QuantizedSigmoid
//     (dnn.cc:289-349, :250-286)                      -> qgemm_kernel<.., OUTPUT=false>
//   CalculateOutput / LazyOutputActivations + SoftMax first loop
//     (dnn.cc:428-454, :355-392, :534-540)            -> qgemm_kernel<.., OUTPUT=true>
//   pmaddubsw int16 pair saturation (dnn.cc:337-340)  -> sparse exact correction inside the k-loop
//
// u8 x s8 on signed MFMA: activations travel as s8 = u8 - 128 (bit 7 flipped),
// so sum_k u8*w = sum_k s8*w + 128*sum_k w; the second term is a per-node int32
// precomputed at load.  Exact: |sum| <= 2^15 * 255 * 128 < 2^31.
//
// C[node][frame] = sum_k W[node][k] * A[frame][k].  Both operands are K-contiguous
// byte rows, so both MFMA fragments are plain 16-byte row slices.
//
// Workgroup tile: 256 nodes x FT = 32*NF*WN frames, 4*WN waves.  Wave (wm, wn) owns
// nodes [64wm, 64wm+64) x frames [32*NF*wn, +32*NF) = 2 x NF MFMA 32x32 tiles
// (32*NF accumulator registers).
//
// L2 -> LDS: buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave instruction) into a
// ring of STAGES buffers, one raw barrier per k-step.  The 8-wave double-buffered shapes
// put that barrier before the step's LAST 32-deep sub-step ("rotated", see ROT below);
// the 4-wave 3-stage shapes keep it at the top with loads in flight across it under a
// counted s_waitcnt vmcnt.  The k-step is BK bytes of every row; with
// BK = 128 each row segment is one whole 128-byte cache line, which the vector
// L1 serves at twice the rate of 64-byte segments (tools/ubench_glds.hip:
// 29 vs 17 TB/s chip-wide from L2).
//
// LDS image: rows of BK bytes; 16-byte chunk c of row r is stored at chunk
// c ^ swz(r) with swz(r) = (r>>2)&3 for 64-byte rows (4 rows per 256-B bank row)
// and (r>>1)&7 for 128-byte rows (2 rows per bank row).  The XOR is applied to the
// per-lane GLOBAL address while the LDS destination stays lane-linear, and again
// on the ds_read_b128 fragment reads, so every 16-lane read group hits 16
// distinct 16-byte slots (the k-loop's reads are conflict free; what SQ_LDS_BANK_CONFLICT shows for this
// kernel -- 1.2 M cycles per hidden launch -- are the epilogue's table byte gathers and tile transposition).
#include <atomic>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"
#include "fdnn_tile.hpp"

#include <cstdlib>
#include <type_traits>

// Timing experiments only (never set in the shipped build): bit 0 = no staging
// loads after the prologue, bit 1 = no MFMA, bit 2 = no LDS fragment reads.
// 1024 = workgroups return at once (launch cost: 2.4 us per kernel in a stream of launches,
// 6.4 us when every launch is bracketed by HIP events as in bench.py's per-kernel pass).
#ifndef FDNN_GEMM_DEBUG
#define FDNN_GEMM_DEBUG 0
#endif
#ifndef FDNN_GEMM_PIN
#define FDNN_GEMM_PIN 0  // 1: fragment reads pinned one behind each MFMA of the sub-step before (sched_group_barrier)
#endif
#ifndef FDNN_FUSE_PARTS
#define FDNN_FUSE_PARTS 2  // parts of the fused soft-max's row-sum exchange (one per 32-frame block at most)
#endif
#ifndef FDNN_FUSE_SLEEP
#define FDNN_FUSE_SLEEP 16  // 64-cycle units between two polls of a frame tile's arrival counter
#endif
#ifndef FDNN_SMALL_STAGES
#define FDNN_SMALL_STAGES 3  // ring depth of the 256-node x 32-frame tile (4: no change measured)
#endif


namespace fdnn {
namespace {

// FAST: the layer's 3-op division was validated against IEEE division at load
// (every layer of a sane net); !FAST keeps the true divide for the rest.
// PLAIN (output layer only): no mask, no taps, rows of whole cache lines (width % 32 == 0) -- the dense
// production call, without the per-group branches of the general epilogue.
// MASKED (with PLAIN): the same branch-free epilogue for the batched lazy call -- mask present,
// no taps; the mask only selects z = 0 for inactive nodes (with ANYW for widths % 4 != 0).
// ANYW (with PLAIN): the dense call for every other output width (pdf counts are arbitrary):
// per-element range test, 4-byte-aligned dwordx4 stores, scalar stores for a row's last group.
// FUSED (with OUTPUT, PLAIN, FAST; 8-wave shapes): the soft-max scale inside this kernel -- every workgroup keeps its
// exp(z) tile in the accumulator registers, the 256-node tiles of a frame tile exchange their row sums through memory,
// and what leaves is probabilities: no exp(z) round trip through HBM, no normalize pass ("fused soft-max" below).
// NOFIX (the large production shapes): a layer WITHOUT saturating pairs (fix_ent == null: trained, heavy-tailed nets) runs an
// instance whose k-loop does not contain the entry walk at all -- the never-taken walk costs the loop 1.4 % (LABBOOK round 5:
// 596.2 -> 587.8 us per 10 000-frame pass of the pair-free bench net).
template <int NF, int WN, int BK, int STAGES, bool OUTPUT, bool TAP, bool FAST, bool PLAIN = false, bool MASKED = false, bool ANYW = false,
          int WM = 4, bool FUSED = false, bool NOFIX = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void qgemm_kernel(QGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (buffer-resource builtins are device-only)
  using Cfg = GemmCfg<NF, WN, BK, STAGES, WM>;
  constexpr int FT = Cfg::FT, NW = Cfg::NW, G_BM = Cfg::G_BM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#if FDNN_GEMM_DEBUG & 64
  long long ts[6], ts_fix = 0;
  int n_fix_done = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  ts[0] = __builtin_readcyclecounter();
#define FDNN_TS(i) ts[i] = __builtin_readcyclecounter()
#else
#define FDNN_TS(i)
#endif
#if FDNN_GEMM_DEBUG & 1024  // launch-cost experiment: the workgroups do nothing at all
  if (p.n >= 0) return;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;

  // Workgroup b runs on XCD b%8 (strictly so for the first round of a launch; later rounds the
  // dispatcher skips a full XCD now and then -- a locality hint, nothing may depend on it).
  // Give each XCD a contiguous band of frame tiles and
  // walk THOSE fastest: the workgroups resident on an XCD at any time then cover all of
  // its frame tiles (activation rows stay in its 4 MiB L2) and only a few node tiles, so
  // every weight tile is pulled into an XCD once instead of once per frame tile (the
  // 8000-node output layer read 548 MB per launch with node tiles fastest).
  const int MT = p.rows_pad / G_BM;
  const int NT = p.n_pad / FT;
  int mt, nt;
  if (FUSED) {
    // fused soft-max: the MT node tiles of a frame tile are CONSECUTIVE workgroups -- dispatched together, resident
    // together (see the wait below) -- and node tile j of every frame tile lands on XCD j % 8 (for MT % 8 == 0): each
    // XCD keeps a fixed band of MT / 8 weight tiles in its L2 for the whole launch and streams activation rows.
    nt = static_cast<int>(blockIdx.x) / MT;
    mt = static_cast<int>(blockIdx.x) - nt * MT;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_xcd = (NT + 7) / 8;
    mt = j / per_xcd;
    nt = xcd * per_xcd + j % per_xcd;
  }
  if (mt >= MT || nt >= NT) return;
  const int m0 = mt * G_BM, f0 = nt * FT;
  if (FUSED && p.fuse_stagger > 0 && static_cast<int>(blockIdx.x) < 8 * MT) {
    // The first round's workgroups all start at once and stay in step: every CU reaches its store phase (328 KB of
    // probabilities per tile) at the same time, and that burst drains at HBM write speed / 256 per CU while the memory sits
    // idle during the k-loops.  A start offset per frame tile spreads the phases; later rounds inherit it.
    for (int i = (nt & 7) * p.fuse_stagger; i > 0; --i) __builtin_amdgcn_s_sleep(8);
  }

  const size_t ldw = static_cast<size_t>(p.ldw), lda = static_cast<size_t>(p.lda);
  // per-lane source of the staging loads: row = slab*RPI + lane/LPR, chunk swizzled by
  // the row (for 128-byte rows the swizzle depends on the slab's parity = wave&1,
  // because every wave takes slabs wave, wave+NW, ... and NW is even)
  //
  // Addressing: buffer loads (buffer_load_dwordx4 ... offen lds) with one descriptor per
  // operand tile, ONE per-lane VGPR offset (row-in-slab * stride + swizzled chunk) and a
  // scalar offset for slab + k-step.  Flat per-load 64-bit addresses cost two VGPRs per
  // load; at 256 VGPRs the compiler spilled them and each scratch reload drained the
  // whole LDS-DMA queue (s_waitcnt vmcnt(0)) inside the k-loop.
  // (a one-wave workgroup takes EVERY slab: the parity then alternates with the slab index, which is a compile-time
  // constant of the unrolled load -- the odd slabs flip bit 6 of the offset, see stage_load)
  static_assert(NW % 2 == 0 || NW == 1, "slab parity per wave needs an even wave count (or one wave)");
  const int srow = lane / Cfg::LPR;
  const int schunk = ((lane % Cfg::LPR) ^ swz<BK>(wave * Cfg::RPI + srow)) << 4;
  const int voff_w = srow * p.ldw + schunk;
  const int voff_a = srow * p.lda + schunk;
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(p.w + static_cast<size_t>(m0) * ldw), 0, G_BM * p.ldw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(p.a + static_cast<size_t>(f0) * lda), 0, FT * p.lda, 0x00020000);

  const int KT = p.K / BK;
  // one wave issues NLD 1-KB loads per stage: weight slabs wave, wave+NW, ... then activation
  // slabs wave, wave+NW, ...
  constexpr int NLD_W = Cfg::W_SLABS / NW, NLD = NLD_W + (Cfg::A_SLABS + NW - 1) / NW;
  auto stage_load = [&](int kt, int buf, int i) {
    char *base = smem + buf * Cfg::STAGE;
    const int koff = kt * BK;
    // swz<128>(8 * slab + srow) = (4 * slab + (srow >> 1)) & 7: an odd slab flips chunk bit 2 = offset bit 6 (the row
    // strides are multiples of 128 bytes, so the flip stays inside the chunk field)
    const int odd = (NW == 1 && BK == 128) ? 64 : 0;
    if (i < NLD_W) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(base + (i * NW + wave) * 1024), 16, (i & 1) ? voff_w ^ odd : voff_w,
                                               (i * NW + wave) * Cfg::RPI * p.ldw + koff, 0, 0);
    } else {
      const int s = i - NLD_W;
      if (Cfg::A_SLABS % NW == 0 || s * NW + wave < Cfg::A_SLABS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, FDNN_LDS_PTR(base + Cfg::W_BYTES + (s * NW + wave) * 1024), 16,
                                                 (s & 1) ? voff_a ^ odd : voff_a, (s * NW + wave) * Cfg::RPI * p.lda + koff, 0, 0);
    }
  };
  auto stage = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_load(kt, buf, i);
  };

  // The epilogue's sigmoid table and biases ride the same LDS-DMA queue, ahead of the ring
  // (older loads complete first, so every wait that covers stage 0 covers them too).
  char *aux = smem + Cfg::AUX_OFF;
  // four 1-KiB pieces (three of table, one of biases) over however many waves there are
#pragma unroll
  for (int piece = 0; piece < 3; ++piece)
  if (!OUTPUT && piece % NW == wave) {
    // The range of a 16-byte LDS-DMA access is checked as a whole: with num_records = the table
    // size, the lane holding the table's last three (FAST) / last one (exact) entries read zeros --
    // u8 128 instead of 255 for every activation with lin >= 6.395 (found by
    // test_net_with_every_pair_saturating; rare on trained nets, never on the round-1 fixtures).
    // The blob pads both tables to 16 bytes (fdnn_model.cpp: place(size + 15)).
    const int bytes = ((FAST ? kLut2Size : kLutExt) + 15) & ~15;
    const __amdgpu_buffer_rsrc_t rsrc_lut =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(FAST ? p.lut2 : p.lut), 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lut, FDNN_LDS_PTR(aux + piece * 1024), 16, lane * 16, piece * 1024, 0, 0);
  }
  if (wave == 3 % NW) {
    const __amdgpu_buffer_rsrc_t rsrc_bias =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias + m0), 0, G_BM * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_bias, FDNN_LDS_PTR(aux + 3072), 16, lane * 16, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < KT) stage(s, s);
  asm volatile("" ::: "memory");  // the ring's first loads go out before anything else

  // The accumulators start at 128*sum_k(w[node][k]) (the s8 = u8 - 128 activation offset),
  // so that the epilogue has no per-output add left.  D layout (32x32): column (frame) =
  // lane&31, row (node) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  v16i acc[2][NF];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int4 ws4 = *reinterpret_cast<const int4 *>(p.wsum + m0 + 64 * wm + 32 * a + 8 * g + 4 * (lane >> 5));
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        acc[a][b][g * 4 + 0] = ws4.x;
        acc[a][b][g * 4 + 1] = ws4.y;
        acc[a][b][g * 4 + 2] = ws4.z;
        acc[a][b][g * 4 + 3] = ws4.w;
      }
    }

  const int frow = lane & 31, fch = lane >> 5;
  const int arow0 = wn * 32 * NF;  // this wave's first frame row inside the tile

  // pmaddubsw saturation (dnn.cc:337-340).  The MFMA sum is exact; the reference
  // saturates every ADJACENT pair a[2j]*w[2j] + a[2j+1]*w[2j+1] to int16.  Only the
  // few (node, pair) entries listed at load time can saturate at all.  Those of this
  // wave's 64 nodes are sorted by k; when the k-step holding an entry's columns is in
  // LDS, the pair is recomputed from the staged activation bytes and sat16(p) - p is
  // added to the accumulator that holds (node, frame).  The entry walk and the register
  // select are wave-uniform; a layer without risky pairs has fix_k_next = INT_MAX.
  //
  int fix_e = 0, fix_end = 0, fix_k_next = INT_MAX;
  // Entries are walked with SCALAR loads (constant address space: s_load_dwordx2 into SGPRs, two
  // entries ahead): no LDS round trip and no v_readfirstlane per entry, and the scalar cache
  // path does not queue behind the LDS-DMA loads as a vector load issued in the loop would
  // (-4.5 % on a Gaussian-weight layer against an LDS copy of the list).
  typedef const __attribute__((address_space(4))) uint64_t *FixPtr;
  const FixPtr ent_c = (FixPtr)(uintptr_t)p.fix_ent;
  uint64_t fix_raw = 0, fix_raw_nxt = 0;  // {u16 k, s8 w0, s8 w1, s32 node}
  if (!NOFIX && p.fix_ent) {
    const int grp = (m0 >> 6) + wm;
    fix_e = __builtin_amdgcn_readfirstlane(p.fix_grp[grp]);
    fix_end = __builtin_amdgcn_readfirstlane(p.fix_grp[grp + 1]);
    if (fix_e < fix_end) {
      fix_raw = ent_c[fix_e];
      if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
      fix_k_next = static_cast<int>(fix_raw & 0xffff);
    }
  }
  FDNN_TS(1);
  // Code placement: the k-loop is sensitive to where it lands in the instruction
  // stream (measured: the same instructions shifted by one dword ran 0.064 vs 0.049 ms
  // per 2048x2048 layer).  Pin the loop to a 256-byte boundary so that edits elsewhere
  // in the kernel cannot move it; FDNN_GEMM_PAD shifts it for placement experiments.
  asm volatile(".p2align 8");
#ifdef FDNN_GEMM_PAD
#pragma unroll
  for (int i = 0; i < FDNN_GEMM_PAD; ++i) asm volatile("s_nop 0");
#endif

  // ROT (double-buffered 8-wave shapes): the step's one barrier sits before its LAST sub-step
  // instead of before its first.  At that point every wave holds the stage's last fragments in
  // registers, so the stage buffer is free (refill starts at once and has a whole step to
  // land) and the next stage, loaded a step ago, is verified; the next step's first fragments
  // are requested right after the barrier and their LDS latency hides behind the last
  // sub-step's MFMAs instead of idling the matrix pipe after every barrier
  // (tools/ubench_tile.hip, k-loop of one layer: 27.7 us classic, 22.9 us rotated).
  constexpr bool ROT = STAGES == 2 && BK / 32 >= 4;
  constexpr int SUB = BK / 32;
  constexpr int ROT_D0 = (NLD + 1) / 2;  // loads issued right after the barrier; the rest before the next sub-step 0
  v4i a[2][2], b[2][NF];  // MFMA operand fragments, double buffered over the sub-steps
  auto load_frags = [&](const char *wt_, const char *at_, int kt_, int kk, int set) {
#if !(FDNN_GEMM_DEBUG & 4)
    (void)kt_;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[set][mi] = read_frag<BK>(wt_, 64 * wm + 32 * mi + frow, kk * 2 + fch);
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) b[set][ni] = read_frag<BK>(at_, arow0 + 32 * ni + frow, kk * 2 + fch);
#else
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[set][mi] = v4i{kt_, kk, mi, lane};
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) b[set][ni] = v4i{kt_, kk, ni, lane};
#endif
  };
  if (ROT) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once: stage 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(smem, smem + Cfg::W_BYTES, 0, 0, 0);
    if (KT > 1 && !(FDNN_GEMM_DEBUG & 1)) {  // stage 1 is step 0's refill: first half now, the rest before sub-step 0's MFMAs
#pragma unroll
      for (int i = 0; i < ROT_D0; ++i) stage_load(1, 1, i);
    }
  }

  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
#if FDNN_GEMM_DEBUG & 64
    if (kt == 1) FDNN_TS(2);
#endif
    if (!ROT) {
      // stage kt has landed once at most (STAGES-2) younger stages are outstanding
      if (STAGES > 2 && kt + STAGES - 2 < KT) {
        static_assert((STAGES - 2) * Cfg::MIN_LOADS <= 63, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::MIN_LOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // everyone's share of stage kt landed; everyone is done reading stage kt-1
      asm volatile("" ::: "memory");
    }
    // The refill of the buffer freed by the barrier is NOT issued in one burst: a wave
    // issues in order, and NLD back-to-back LDS-DMA loads queue behind the other seven
    // waves' loads in the CU's one address path, which keeps the wave's MFMAs waiting
    // (tools/ubench_pipe.hip: 4550 vs 3400 cycles per 128-k step).  The loads are spread
    // over the k sub-steps instead, each group issued just before a block of MFMAs.
    // (ROT: stage kt+1 goes into the buffer freed at the previous barrier; its first ROT_D0
    // loads went out right after that barrier.)
    const bool refill = kt + STAGES - 1 < KT && !(FDNN_GEMM_DEBUG & 1);
    int nb = buf + STAGES - 1;
    if (nb >= STAGES) nb -= STAGES;
    const char *wt = smem + buf * Cfg::STAGE;
    const char *at = wt + Cfg::W_BYTES;
#if FDNN_GEMM_DEBUG & 64
    const long long tf0 = __builtin_readcyclecounter();
    const int fe0 = fix_e;
#endif
    while (!NOFIX && fix_k_next < (kt + 1) * BK) {  // rare: a risky pair lives in this k-step
      const int node = static_cast<int>(fix_raw >> 32) - (m0 + 64 * wm);  // 0..63
      const int kl = static_cast<int>(fix_raw & 0xffff) - kt * BK;          // even, 0..BK-2
      const int w0 = static_cast<int8_t>(fix_raw >> 16), w1 = static_cast<int8_t>(fix_raw >> 24);
      // Screen first.  A pair that CAN saturate almost never does (both activations must be
      // close to 255: 0.2 % of the (pair, frame) combinations of the Gaussian bench net, under
      // 1 % of the pairs for any of a wave's 160 frames), so the common case only has to prove
      // "no frame of this wave saturates": one frame per lane (not per half-wave as the
      // accumulator layout has it), the pair as one 16-bit LDS read, the pair product as one
      // v_dot4c_i32_i8 on the staged s8 bytes (a = s8 + 128, so p = dot + 128*(w0+w1)), one
      // range test.  Only when some lane fires does the wave run the exact correction below.
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
      for (int ni = 0; ni < NF; ++ni) {  // all lanes read (no divergent branch): the NF gathers go out together
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
      // saturation actually firing is rare (both activations of the pair must be near
      // 255): skip the register select when no lane has a non-zero correction
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
#if FDNN_GEMM_DEBUG & 64
    if (fix_e != fe0) {
      ts_fix += __builtin_readcyclecounter() - tf0;
      n_fix_done += fix_e - fe0;
    }
#endif
    // Fragments are double buffered in registers: the ds_read_b128s of sub-step
    // kk+1 are in flight while the 2*NF MFMAs of sub-step kk issue, so a wave's
    // matrix pipe only waits for LDS once per k-step (the first sub-step; never with ROT).
    if (!ROT) load_frags(wt, at, kt, 0, 0);
#pragma unroll
    for (int kk = 0; kk < SUB; ++kk) {
      if (kk + 1 < SUB) load_frags(wt, at, kt, kk + 1, (kk + 1) & 1);
      if (ROT) {
        if (refill && kk == 0) {  // the second half of stage kt+1's loads (the first went out after the barrier)
#pragma unroll
          for (int i = ROT_D0; i < NLD; ++i) stage_load(kt + 1, nb, i);
        }
        if (kk == SUB - 1) {
          // all of this stage's fragments are in registers or in flight: drain my LDS reads and
          // my share of stage kt+1, meet the other waves, then start the next step's first reads
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (kt + 1 < KT) {
            const char *wn_ = smem + nb * Cfg::STAGE;
            load_frags(wn_, wn_ + Cfg::W_BYTES, kt + 1, 0, 0);
          }
          if (kt + 2 < KT && !(FDNN_GEMM_DEBUG & 1)) {  // stage kt+2 into the buffer this step just released
#pragma unroll
            for (int i = 0; i < ROT_D0; ++i) stage_load(kt + 2, buf, i);
          }
        }
      } else if (refill) {
        // classic 3-stage ring (4-wave shapes): spread over the sub-steps
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          if (i * SUB / NLD == kk) stage_load(kt + STAGES - 1, nb, i);
      }
#if !(FDNN_GEMM_DEBUG & 2)
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni], 0, 0, 0);
#if FDNN_GEMM_PIN
      // Round 6 (tools/ubench_role.hip, tools/ubench_tile.hip): one of the next sub-step's fragment reads pinned behind each of
      // this sub-step's first 2 + NF MFMAs.  Left alone the scheduler issues the reads as a burst in front of the MFMAs, and
      // a wave's matrix pipe waits for its own LDS issue (ubench_tile, rotated 8-wave loop: 3 355 -> 3 146-3 212 cycles a k-step).
      if (ROT && NF >= 2 && kk + 1 < SUB) {
#pragma unroll
        for (int i = 0; i < 2 + NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 2 * NF - 2 - NF, 0);
      }
#endif
#else
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(a[kk & 1][mi]), "v"(b[kk & 1][ni]));
#endif
    }
    if (++buf == STAGES) buf = 0;
  }

  // ------------------------------------------------------------ epilogue
  FDNN_TS(3);
  // The sigmoid table and this tile's 256 biases have been in LDS since the prologue.
  // (They must not be fetched from global memory between the stores below: vmcnt also
  // counts stores, so every such load would wait for all earlier stores of the wave.)
  const uint8_t *lut = reinterpret_cast<const uint8_t *>(aux);
  const float *bias_s = reinterpret_cast<const float *>(aux + 3072);
  char *tile_s = smem + 8192;  // hidden layers: s8 output tile [FT][kTS]
  constexpr int kTS = G_BM + 16;  // row stride: 16-byte aligned, rows 16 apart share a bank (2-way at worst)
  __syncthreads();  // every wave is done with the ring: it becomes the output tile
  FDNN_TS(4);
  // D layout (32x32): column (frame) = lane&31, row (node) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  const int half = lane >> 5;
  const bool vec4 = (p.rows & 3) == 0;
  const int fw0 = f0 + arow0;

  float psum[NF];
#pragma unroll
  for (int ni = 0; ni < NF; ++ni) psum[ni] = 0.0f;
#if FDNN_GEMM_DEBUG & 32
  {  // timing experiment: no epilogue at all (keep the accumulators alive)
    int keep = 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep ^= acc[mi][ni][r];
    if (keep == 0x12345678) p.act_out[0] = 1;
    return;
  }
#endif
  if (OUTPUT && p.acc_probe != nullptr) {  // parity tests: the raw accumulators of every probe_stride-th frame (uniform branch)
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const int f = fw0 + 32 * ni + frow;
      if (f < p.n && f % p.probe_stride == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int node = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (node < p.rows) p.acc_probe[static_cast<size_t>(f / p.probe_stride) * p.rows + node] = acc[mi][ni][r];
          }
      }
    }
  }
  if constexpr (FUSED) {
    static_assert(!FUSED || (OUTPUT && PLAIN && FAST && !TAP && WM == 4), "fused soft-max: the dense / batched-lazy production instances only");
    // (ANYW, round 5: output widths that are not a multiple of 32 -- pdf counts are arbitrary, dnn.cc:428-454 takes any width --
    // per-element range test in phase 1, plain 4-byte-aligned stores and a scalar tail in phase 2)
    // ---------------------------------------------------------------- fused soft-max
    // SoftMax::apply (dnn.cc:534-544) inside the output kernel.  Phase 1: e = exp(z) replaces each accumulator IN
    // PLACE (160 registers per lane stay live), the 64-node partial sums P are formed exactly as in the unfused
    // instance.  The workgroup's four P per frame give S = (P0 + P1) + (P2 + P3) -- the first level of the library's
    // row-total order (normalize_row) -- which it publishes write-through; an arrival counter per frame tile tells when
    // all MT tiles have; every workgroup then reads the MT vectors of S (40 KB), finishes the tree per frame, and
    // phase 2 multiplies its e by RN(1 / total) on the way out.  Placement-independent (sc0 sc1 stores and loads on
    // both sides, one relaxed agent-scope counter); progress: workgroups are dispatched in block order, so the oldest
    // unfinished frame tile always has all its MT workgroups resident.  Every spin is bounded: a workgroup that gives
    // up stores exp(z) unscaled and flags its tile; the frame tile's last workgroup scales it.
    constexpr int kOS = 64 + 4;
    float *wtile = reinterpret_cast<float *>(smem + 8192) + wave * (32 * kOS);
    const int ncol0 = m0 + 64 * wm;
    constexpr int kFuseOff = 8192 + NW * 32 * kOS * 4;
    float *Pw = reinterpret_cast<float *>(smem + kFuseOff);  // [4][FT]
    float *inv_s = Pw + 4 * FT;                               // [FT]
    int *ok_s = reinterpret_cast<int *>(inv_s + FT);          // [4]
    float *Sg = inv_s + FT + 4;                               // [L][FT], L = MT rounded up to a power of two
    int L = 1;
    while (L < MT) L <<= 1;
    // (the host launches this instance only when kFuseOff + (5 FT + 4 + L FT) * 4 <= Cfg::FIX_OFF)
    float ev[2][NF][16];  // exp(z): takes over the accumulators' registers as they die
    // MASKED (batched lazy call): one 64-bit word of mask bits per frame row covers this wave's 64 nodes (launch_mask_pack)
    uint64_t fmw[NF];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const int ff = fw0 + 32 * ni + frow, grp = ncol0 >> 6;
      fmw[ni] = (MASKED && ff < p.n && grp < p.mask_wpr) ? p.mask_bits[static_cast<size_t>(ff) * p.mask_wpr + grp] : 0ull;
    }
    // Round 4: the exchange in TWO PARTS.  A frame's row sum is complete as soon as its own 32-frame block has gone through
    // phase 1, so the sums of the first part's blocks are published (and their arrival counted) before the remaining blocks'
    // exp work; by the time a workgroup has finished phase 1 its siblings' first part has long been published, and while it
    // scales and stores the first part's blocks their second part arrives: of the wait for the slowest of the MT siblings
    // only what exceeds that slack is left.  Same sums, same tree, same bits.  Counters per frame tile: arrivals per part,
    // leavers at [7]; give-up flags per tile: bit = part.  Measured, 10 000 x 8000: one part 229-234 us, two 220-222,
    // three 228, five 248 (every part is two more barriers, an atomic and a gather).
    constexpr int kParts = FDNN_FUSE_PARTS < NF ? FDNN_FUSE_PARTS : NF;  // block ni belongs to part ni * kParts / NF
#ifndef FDNN_FUSE_SPLIT0
#define FDNN_FUSE_SPLIT0 2  // two parts: 32-frame blocks in the first one (0 = half, rounded up); 320-frame tiles, 10 000 x 8000: 2 -> 220.6 us, 3 -> 222.5, 4 -> 228
#endif
    constexpr int kSplit0 = (kParts == 2 && FDNN_FUSE_SPLIT0 > 0) ? (FDNN_FUSE_SPLIT0 < NF ? FDNN_FUSE_SPLIT0 : NF - 1) : 0;
    auto part_lo = [](int part) { return kSplit0 ? (part == 0 ? 0 : part == 1 ? kSplit0 : NF) : (part * NF + kParts - 1) / kParts; };  // first block of a part (part_lo(kParts) = NF)
    auto block_part = [](int ni) { return kSplit0 ? (ni < kSplit0 ? 0 : 1) : ni * kParts / NF; };
    auto exp_block = [&](int ni) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = ncol0 + 32 * mi + 8 * g + 4 * half;
          const v4f_t b4 = *reinterpret_cast<const v4f_t *>(bias_s + (nb - m0));
          const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
          const bool in4 = nb < p.rows;
          typedef float v2f_e __attribute__((ext_vector_type(2)));
          const v2f_e rcp2 = {p.rcp_coef, p.rcp_coef}, coef2 = {p.coef, p.coef};
          const v2f_e log2e2 = {1.44269504088896340736f, 1.44269504088896340736f};
          float e[4];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {  // the operations of the unfused dense instance, one for one
            const v2f_e x = {static_cast<float>(acc[mi][ni][g * 4 + 2 * h2]), static_cast<float>(acc[mi][ni][g * 4 + 2 * h2 + 1])};
            const v2f_e q0 = x * rcp2;
            const v2f_e r = __builtin_elementwise_fma(-q0, coef2, x);
            v2f_e z = __builtin_elementwise_fma(r, rcp2, q0) + v2f_e{bj[2 * h2], bj[2 * h2 + 1]};
            if (MASKED) {  // masked-out nodes keep z = 0 (dnn.cc:366-369)
              const uint32_t nib = static_cast<uint32_t>(fmw[ni] >> (32 * mi + 8 * g + 4 * half)) & 0xfu;
              if (((nib >> (2 * h2)) & 1u) == 0) z.x = 0.0f;
              if (((nib >> (2 * h2 + 1)) & 1u) == 0) z.y = 0.0f;
            }
            const v2f_e y = z * log2e2;
            const v2f_e keep = ANYW ? v2f_e{nb + 2 * h2 < p.rows ? 1.0f : 0.0f, nb + 2 * h2 + 1 < p.rows ? 1.0f : 0.0f} : v2f_e{in4 ? 1.0f : 0.0f, in4 ? 1.0f : 0.0f};
            const v2f_e ex = v2f_e{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)} * keep;
            e[2 * h2] = ex.x;
            e[2 * h2 + 1] = ex.y;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            psum[ni] += e[q];
            ev[mi][ni][g * 4 + q] = e[q];
          }
        }
      }
      const float tot = psum[ni] + __shfl_xor(psum[ni], 32);
      if (half == 0) Pw[wm * FT + arow0 + 32 * ni + frow] = tot;
    };
#if FDNN_GEMM_DEBUG & 64
    long long tf[6], tp[2 * kParts + 1], tw[kParts];
#define FDNN_CLK(x) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
    tf[0] = tf[1] = tf[2] = tf[3] = __builtin_readcyclecounter();
#endif
    float *gS = p.fuse_s + (static_cast<size_t>(nt) * MT + mt) * FT;
    uint32_t *cnt = p.fuse_cnt + 8 * nt;  // {arrived part 0 .. kParts - 1, ..., left at [7]}: all zero between launches
    // frame row f of the tile belongs to block (f % (32 NF)) / 32 of its wave half
    auto part_of_row = [&](int f) { return block_part((f % (32 * NF)) >> 5); };
    auto publish = [&](int part) {
      __syncthreads();  // the part's Pw entries are complete
      if (tid < FT / 4 && part_of_row(4 * tid) == part) {
        v4f_t s4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 4 * tid + q;
          s4[q] = (Pw[f] + Pw[FT + f]) + (Pw[2 * FT + f] + Pw[3 * FT + f]);
        }
        store_wt(gS + 4 * tid, s4);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the S stores have left (inline-asm stores: nobody else waits for them)
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt + part, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t gave_up = 0;  // (meaningful in thread 0 only: ok_s carries the verdict to the others)
    // a part's MT x frames sums, fetched past the (non-coherent) L2s, the tree per frame, inv_s = RN(1 / total)
    auto totals = [&](int part) {
      const int lo = part_lo(part), nb_part = part_lo(part + 1) - lo;  // 32-frame blocks of the part, per wave half
      const int quads = WN * nb_part * 8;                              // 4-frame pieces of the part
      const float *gall = p.fuse_s + static_cast<size_t>(nt) * MT * FT;
      const int items = MT * quads;
      for (int i0 = tid; i0 < items; i0 += 4 * Cfg::THREADS) {  // four 16-byte loads per lane in flight
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + u * Cfg::THREADS, items - 1);  // (clamped duplicates rewrite the same bytes)
          const int row = i / quads, j = i - row * quads, blk = j >> 3, f0q = (blk / nb_part) * (32 * NF) + 32 * (lo + blk % nb_part) + 4 * (j & 7);
          off[u] = row * FT + f0q;
        }
        v4f_t v0, v1, v2, v3;
        asm volatile(
            "global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
            "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
            : "v"(gall + off[0]), "v"(gall + off[1]), "v"(gall + off[2]), "v"(gall + off[3])
            : "memory");
        *reinterpret_cast<v4f_t *>(Sg + off[0]) = v0;
        *reinterpret_cast<v4f_t *>(Sg + off[1]) = v1;
        *reinterpret_cast<v4f_t *>(Sg + off[2]) = v2;
        *reinterpret_cast<v4f_t *>(Sg + off[3]) = v3;
      }
      for (int i = MT * FT + tid; i < L * FT; i += Cfg::THREADS)
        if (part_of_row(i % FT) == part) Sg[i] = 0.0f;  // zero padding: x + 0 = x
      __syncthreads();
      if (tid < FT && part_of_row(tid) == part) {  // adjacent pairs, level by level (normalize_row's tree), one frame per thread, in place
        for (int len = L >> 1; len >= 1; len >>= 1)
          for (int i = 0; i < len; ++i) Sg[i * FT + tid] = Sg[2 * i * FT + tid] + Sg[(2 * i + 1) * FT + tid];
        inv_s[tid] = 1.0f / Sg[tid];  // p_i = e_i * RN(1 / total), as normalize_row
      }
    };
    // wait for the part's MT arrivals, then its totals
    auto collect = [&](int part) {
      if (tid == 0) {
        int ok = (p.debug & 4096) && mt % 3 == part % 3 ? 0 : 1, spins = 0;  // FDNN_GEMM_DEBUG=4096 (tests): some node tiles "give up" on one part
        while (!(FDNN_GEMM_DEBUG & 256) && ok && __hip_atomic_load(cnt + part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < static_cast<uint32_t>(MT)) {  // (256, timing build: no wait)
          __builtin_amdgcn_s_sleep(FDNN_FUSE_SLEEP);
          if (++spins > (1 << 15) * 16 / FDNN_FUSE_SLEEP) {  // tens of milliseconds (a legitimate wait is microseconds): something keeps this frame tile's other workgroups off the chip
            ok = 0;
            break;
          }
        }
        ok_s[0] = ok;
        if (!ok) gave_up |= 1u << part;
      }
      __syncthreads();
#if FDNN_GEMM_DEBUG & 64
      FDNN_CLK(tw[part]);
#endif
      if (ok_s[0] != 0) {
        totals(part);
      } else if (tid < FT && part_of_row(tid) == part) {
        inv_s[tid] = 1.0f;  // this part of the tile leaves unscaled; the frame tile's last workgroup scales it (below)
      }
      __syncthreads();
    };
    // phase 2 for one block: scale, transpose through the wave's LDS tile, 256-byte row segments out
    auto scale_store = [&](int ni) {
      const float iv = inv_s[arow0 + 32 * ni + frow];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = ncol0 + 32 * mi + 8 * g + 4 * half;
          *reinterpret_cast<v4f_t *>(wtile + frow * kOS + (nb - ncol0)) =
              v4f_t{ev[mi][ni][g * 4 + 0] * iv, ev[mi][ni][g * 4 + 1] * iv, ev[mi][ni][g * 4 + 2] * iv, ev[mi][ni][g * 4 + 3] * iv};
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = r * 4 + (lane >> 4), col = (lane & 15) * 4;
        const v4f_t v = *reinterpret_cast<const v4f_t *>(wtile + row * kOS + col);
        const int ff = fw0 + 32 * ni + row;
#ifdef FDNN_FUSE_PLAIN_STORE
        if (ff < p.n && ncol0 + col + 4 <= p.rows) *reinterpret_cast<v4f_t *>(p.final + static_cast<size_t>(ff) * p.rows + ncol0 + col) = v;
#elif defined(FDNN_FUSE_NT_STORE)
        if (ff < p.n && ncol0 + col + 4 <= p.rows) __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(p.final + static_cast<size_t>(ff) * p.rows + ncol0 + col));
#else
#if FDNN_GEMM_DEBUG & 128  // (timing build: the probabilities do not leave)
        if (ff < 0 && ncol0 + col + 4 <= p.rows) store_wt(p.final + static_cast<size_t>(ff) * p.rows + ncol0 + col, v);
#else
        if (!ANYW) {
          if (ff < p.n && ncol0 + col + 4 <= p.rows) store_wt(p.final + static_cast<size_t>(ff) * p.rows + ncol0 + col, v);
        } else if (ff < p.n) {
          // rows of any width: a group of four starts on a 4-byte boundary only, and rows share cache lines -- plain stores
          // (written through, a line shared by two rows becomes partial-line writes to memory), scalar ones for a row's last group
          float *op = p.final + static_cast<size_t>(ff) * p.rows + ncol0 + col;
          if (ncol0 + col + 4 <= p.rows) {
            typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
            *reinterpret_cast<v4f_a4 *>(op) = v4f_a4{v.x, v.y, v.z, v.w};
          } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int q = 0; q < 4; ++q)
              if (ncol0 + col + q < p.rows) op[q] = vv[q];
          }
        }
#endif
#endif
      }
    };
#pragma unroll
    for (int part = 0; part < kParts; ++part) {
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
        if (block_part(ni) == part) exp_block(ni);
      publish(part);
    }
#if FDNN_GEMM_DEBUG & 64
    tf[2] = __builtin_readcyclecounter();
#endif
#if FDNN_GEMM_DEBUG & 64
    FDNN_CLK(tp[0]);
#endif
#pragma unroll
    for (int part = 0; part < kParts; ++part) {
      collect(part);
#if FDNN_GEMM_DEBUG & 64
      FDNN_CLK(tp[2 * part + 1]);
#endif
#pragma unroll
      for (int ni = 0; ni < NF; ++ni)
        if (block_part(ni) == part) scale_store(ni);
#if FDNN_GEMM_DEBUG & 64
      FDNN_CLK(tp[2 * part + 2]);  // (timing build: the part's stores have been acknowledged)
#endif
    }
    // ---- leaving.  A workgroup that gave up on a part has stored that part's block unscaled and raises a flag (bit = part); the
    // LAST workgroup to leave the frame tile -- every sum has been published by then -- scales such blocks and lowers the
    // flags: nothing is left for a second launch (rounds 3's fuse_cleanup_kernel: 4 us per step for, normally, nothing).
    if (tid == 0) ok_s[1] = static_cast<int>(gave_up);
    __syncthreads();
    if (ok_s[1] != 0) {  // (uniform, rare) my unscaled stores must have reached memory before the flag can be seen
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (tid == 0) {
      if (gave_up) {
        // (ANYW stores are plain: written back to memory by the release BEFORE the flag can be seen; the write-through
        // stores of the other instances are there already -- every wave drained them above)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.fuse_flag + static_cast<size_t>(nt) * MT + mt, gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // acq_rel (advisor, round 4): the last workgroup's reads of the flags and of the unscaled blocks below are ordered behind
      // every leaver's add, by the memory model and not only by their cache policy
      const uint32_t prev = __hip_atomic_fetch_add(cnt + 7, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      ok_s[2] = prev == static_cast<uint32_t>(MT) - 1u ? 1 : 0;
    }
    __syncthreads();
    if (ok_s[2] != 0) {  // the last workgroup of the frame tile
      uint32_t *fl_s = reinterpret_cast<uint32_t *>(Pw);  // [MT] (Pw is dead)
      if (tid < MT) fl_s[tid] = __hip_atomic_load(p.fuse_flag + static_cast<size_t>(nt) * MT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      uint32_t any = 0;
      for (int j = 0; j < MT; ++j) any |= fl_s[j];
      if (any) {  // (uniform) the path nobody should ever see
        for (int part = 0; part < kParts; ++part) {
          if (!((any >> part) & 1u)) continue;
          __syncthreads();
          totals(part);  // inv_s of the part's frames
          __syncthreads();
          for (int j = 0; j < MT; ++j) {
            if (!((fl_s[j] >> part) & 1u)) continue;
            // tile j's 256 columns of the part's frames: 64 four-column pieces per frame
            for (int it = tid; it < FT * 64; it += Cfg::THREADS) {
              const int f = it >> 6, c4 = j * 256 + 4 * (it & 63), frame = nt * FT + f;
              if (part_of_row(f) != part || frame >= p.n || c4 >= p.rows) continue;
              float *at = p.final + static_cast<size_t>(frame) * p.rows + c4;
              const float iv = inv_s[f];
              if (c4 + 4 <= p.rows) {
                v4f_t v;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(at) : "memory");
                store_wt(at, v4f_t{v.x * iv, v.y * iv, v.z * iv, v.w * iv});
              } else {  // (ANYW) a row's last, partial group
                for (int q = 0; c4 + q < p.rows; ++q) {
                  const float v = __hip_atomic_load(at + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(at + q, v * iv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
            }
          }
        }
        if (tid < MT && fl_s[tid] != 0u) {
          __hip_atomic_store(p.fuse_flag + static_cast<size_t>(nt) * MT + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p.fuse_giveups) atomicAdd(p.fuse_giveups, 1ull);  // observable: fdnn_model_fuse_giveups
          if (p.fuse_fault) __hip_atomic_fetch_or(p.fuse_fault, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // ... and acted upon: run_output (bit 0: a give-up that was finished after the fact)
        }
      }
      if (tid == 0) {  // the counters are ready for the next launch
#pragma unroll
        for (int i = 0; i < 8; ++i) __hip_atomic_store(cnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#if FDNN_GEMM_DEBUG & 64
    tf[5] = __builtin_readcyclecounter();
    if (tid == 0 && (blockIdx.x % 149) == 0)
      printf("FUSED %4d: prologue %lld first-stage %lld mainloop %lld | exp + publish (all parts) %lld collect + scale + store %lld | total %lld\n", blockIdx.x,
             ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], tf[2] - ts[3], tf[5] - tf[2], tf[5] - ts[0]);
    if (tid == 0 && (blockIdx.x % 149) == 0)
      printf("PARTS %4d: poll0 %d totals0 %d store0 %d poll1 %d totals1 %d store1 %d\n", blockIdx.x, static_cast<int>(tw[0] - tp[0]), static_cast<int>(tp[1] - tw[0]),
             static_cast<int>(tp[2] - tp[1]), static_cast<int>(tw[kParts - 1] - tp[2 * kParts - 2]), static_cast<int>(tp[2 * kParts - 1] - tw[kParts - 1]),
             static_cast<int>(tp[2 * kParts] - tp[2 * kParts - 1]));
#endif
    return;
  }
  if (OUTPUT) {
    // CalculateOutput / LazyOutputActivations: z = sum/coef + bias (masked-out nodes keep
    // z = 0, dnn.cc:366-369), e = exp(z) (SoftMax::apply first loop, dnn.cc:536-540).
    // Each wave parks one 64-node x 32-frame block of e in its own LDS tile and writes
    // it out as 256-byte row segments (a direct float4 store per accumulator group
    // would touch 32 rows x 32 bytes per instruction and is address-processing bound).
    //
    // PLAIN (a separate kernel instance: no mask, no taps, output width a multiple of 4 = the
    // dense production call) carries none of the per-group branches of the general code --
    // those branches, not the 318 MB of stores, were 25-50 k cycles of a 95 k-cycle tile.
    constexpr int kOS = 64 + 4;  // floats per tile row
    float *wtile = reinterpret_cast<float *>(smem + 8192) + wave * (32 * kOS);
    const int ncol0 = m0 + 64 * wm;
    // Lazy contract: this wave's 32 x 64 piece of the mask goes through LDS first (16 dwords per
    // row are one 64-byte segment: 8 wave-loads of 4 segments each) -- reading it straight from
    // global memory where it is needed is 8 loads of 64 different cache lines each per piece,
    // and made the masked call 60 % slower than the dense one.
    constexpr int kMS = 64 + 16;  // bytes per mask tile row
    uint8_t *mtile = reinterpret_cast<uint8_t *>(smem + 8192 + NW * (32 * kOS * 4)) + wave * (32 * kMS);
    static_assert(8192 + NW * (32 * kOS * 4) + NW * (32 * kMS) <= Cfg::FIX_OFF, "mask tiles must not reach the table/biases");
    const bool wt_rows = (p.rows & 31) == 0;
    // The production lazy instances (MASKED) read the mask as BITS: one 64-bit word per frame row covers this wave's 64
    // nodes (launch_mask_pack has turned the caller's 80 MB of bytes into 10 MB of bits at HBM speed).  Eight dwords per
    // lane, an LDS round trip per piece and 30 spilled registers went with the byte version: 0.215 -> 0.18x ms.
    const bool mask_staged = !PLAIN && p.mask != nullptr && vec4;
    auto bits_fetch = [&](int ni) -> uint64_t {
      const int ff = fw0 + 32 * ni + frow, grp = ncol0 >> 6;
      return (ff < p.n && grp < p.mask_wpr) ? p.mask_bits[static_cast<size_t>(ff) * p.mask_wpr + grp] : 0ull;
    };
    // all NF words up front, before the first result store: vmcnt counts stores too, so a load issued between them would
    // wait for every earlier store's acknowledgement (one such wait per frame block: 192 vs 180 us)
    uint64_t mwords[NF];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) mwords[ni] = MASKED ? bits_fetch(ni) : 0ull;
    uint32_t mreg[8];  // the next piece is fetched while the current one is being used
    auto mask_fetch = [&](int ni) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int item = lane + 64 * j, row = item >> 4, c4 = (item & 15) * 4;
        const int ff = fw0 + 32 * ni + row, node = ncol0 + c4;
        mreg[j] = 0;
        if (ff < p.n && node < p.rows) {
          const int8_t *mp = p.mask + static_cast<size_t>(ff) * p.rows + node;
          if (!ANYW) {
            mreg[j] = *reinterpret_cast<const uint32_t *>(mp);
          } else {  // rows % 4 != 0: the piece starts at any byte, and the last row must not be overread
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (node + q < p.rows) mreg[j] |= static_cast<uint32_t>(static_cast<uint8_t>(mp[q])) << (8 * q);
          }
        }
      }
    };
    if ((MASKED || !PLAIN) && mask_staged) mask_fetch(0);
    {
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int f = fw0 + 32 * ni + frow;
        const bool live = f < p.n;
        const uint64_t mword = mwords[ni];
        if ((MASKED || !PLAIN) && mask_staged) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int item = lane + 64 * j;
            *reinterpret_cast<uint32_t *>(mtile + (item >> 4) * kMS + (item & 15) * 4) = mreg[j];
          }
          if (ni + 1 < NF) mask_fetch(ni + 1);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nb = ncol0 + 32 * mi + 8 * g + 4 * half;  // 4 consecutive nodes nb..nb+3
            const float4 b4 = *reinterpret_cast<const float4 *>(bias_s + (nb - m0));
            const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
            uint32_t mbits = 0x01010101u;
            if (MASKED) {  // four bits -> four bytes (bit i to byte i; the products land on distinct bits: no carries)
              const uint32_t nib = static_cast<uint32_t>(mword >> (32 * mi + 8 * g + 4 * half)) & 0xfu;
              mbits = (nib * 0x00204081u) & 0x01010101u;
            } else if ((MASKED || !PLAIN) && mask_staged) {
              mbits = *reinterpret_cast<const uint32_t *>(mtile + frow * kMS + (nb - ncol0));
            } else if (!PLAIN && p.mask && live && nb < p.rows) {
              const int8_t *mp = p.mask + static_cast<size_t>(f) * p.rows + nb;
              if (vec4) {
                mbits = *reinterpret_cast<const uint32_t *>(mp);
              } else {
                mbits = 0;
                for (int q = 0; q < 4; ++q)
                  if (nb + q < p.rows && mp[q]) mbits |= 0xffu << (8 * q);
              }
            }
            // PLAIN: the whole group of 4 nodes is inside or outside the layer (rows % 4 == 0)
            const bool in4 = nb < p.rows;
            float e[4];
            if constexpr (PLAIN && FAST) {
              // The dense / batched-lazy production instances: the same operations per element as the loop below (cvt,
              // x * y, two fma of the exact division, + bias, * log2 e, v_exp_f32), written on pairs so that the five
              // multiplies / fmas / adds go out as packed instructions -- left to itself the compiler packs the hidden
              // layers' epilogue but not this one (965 -> 485 vector instructions per lane and tile).
              typedef float v2f_e __attribute__((ext_vector_type(2)));
              const v2f_e rcp2 = {p.rcp_coef, p.rcp_coef}, coef2 = {p.coef, p.coef};
              const v2f_e log2e2 = {1.44269504088896340736f, 1.44269504088896340736f};
#pragma unroll
              for (int h2 = 0; h2 < 2; ++h2) {
                const v2f_e x = {static_cast<float>(acc[mi][ni][g * 4 + 2 * h2]), static_cast<float>(acc[mi][ni][g * 4 + 2 * h2 + 1])};
                const v2f_e q0 = x * rcp2;
                const v2f_e r = __builtin_elementwise_fma(-q0, coef2, x);
                v2f_e z = __builtin_elementwise_fma(r, rcp2, q0) + v2f_e{bj[2 * h2], bj[2 * h2 + 1]};
                if (MASKED) {
                  if (((mbits >> (16 * h2)) & 0xffu) == 0) z.x = 0.0f;
                  if (((mbits >> (16 * h2 + 8)) & 0xffu) == 0) z.y = 0.0f;
                }
                const v2f_e y = z * log2e2;
                if (ANYW || (FDNN_GEMM_DEBUG & 256)) {
                  e[2 * h2] = nb + 2 * h2 < p.rows ? __builtin_amdgcn_exp2f(y.x) : 0.0f;
                  e[2 * h2 + 1] = nb + 2 * h2 + 1 < p.rows ? __builtin_amdgcn_exp2f(y.y) : 0.0f;
                } else {
                  // nodes past the layer's width (zero weights, zero bias: z = 0, e = 1) must not reach the row sum: one
                  // packed multiply by 0 / 1 per pair instead of a select per element
                  const v2f_e ev = v2f_e{__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)} * v2f_e{in4 ? 1.0f : 0.0f, in4 ? 1.0f : 0.0f};
                  e[2 * h2] = ev.x;
                  e[2 * h2 + 1] = ev.y;
                }
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) psum[ni] += e[q];
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int av = acc[mi][ni][g * 4 + q];
              if (!PLAIN && TAP && live && nb + q < p.rows) p.tap_acc[static_cast<size_t>(f) * p.rows + nb + q] = av;
              float z = dequant<FAST>(av, p.coef, p.rcp_coef) + bj[q];  // sum/coef, then += bias (dnn.cc:311, :446)
              if ((MASKED || !PLAIN) && ((mbits >> (8 * q)) & 0xffu) == 0) z = 0.0f;
              if (!PLAIN && TAP && live && nb + q < p.rows) p.tap_logit[static_cast<size_t>(f) * p.rows + nb + q] = z;
              e[q] = ((PLAIN && !ANYW) ? in4 : (nb + q < p.rows)) ? __expf(z) : 0.0f;
              psum[ni] += e[q];
            }
            *reinterpret_cast<float4 *>(wtile + frow * kOS + (nb - ncol0)) = make_float4(e[0], e[1], e[2], e[3]);
          }
        }
        // wave-private tile: 8 x (4 rows x 256 B) stores
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int row = r * 4 + (lane >> 4), col = (lane & 15) * 4;
          const float4 v = *reinterpret_cast<const float4 *>(wtile + row * kOS + col);
          const int ff = fw0 + 32 * ni + row;
          float *op = p.out + static_cast<size_t>(ff) * p.rows + ncol0 + col;
          // whole groups of four as one store, whatever the output width: a row start off the
          // 16-byte grid (rows % 4 != 0) is still 4-byte aligned, which is all global_store_dwordx4 needs
          if (PLAIN || vec4) {
#if FDNN_GEMM_DEBUG & 16  // ablation: no output stores
            if (v.x == 1234.5f && ff < p.n && ncol0 + col < p.rows) *reinterpret_cast<float4 *>(op) = v;
#else
            if (ff < p.n && ncol0 + col + 4 <= p.rows) {
              typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
              // write-through for rows of whole cache lines only (see normalize_kernel): the PLAIN
              // instance is launched for exactly those, the others test
              if ((FDNN_WT & 2) && !ANYW && ((PLAIN && !MASKED) || wt_rows))
                store_wt(op, v4f_t{v.x, v.y, v.z, v.w});
              else
                *reinterpret_cast<v4f_a4 *>(op) = v4f_a4{v.x, v.y, v.z, v.w};
            } else if (ANYW && ff < p.n) {  // the last, partial group of the row
              const float vv[4] = {v.x, v.y, v.z, v.w};
              for (int q = 0; q < 4; ++q)
                if (ncol0 + col + q < p.rows) op[q] = vv[q];
            }
#endif
          } else if (ff < p.n) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int q = 0; q < 4; ++q)
              if (ncol0 + col + q < p.rows) op[q] = vv[q];
          }
        }
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < (OUTPUT ? 0 : 2); ++mi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = m0 + 64 * wm + 32 * mi + 8 * g + 4 * half;  // 4 consecutive nodes nb..nb+3
      const float4 b4 = *reinterpret_cast<const float4 *>(bias_s + (nb - m0));
      const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
      // AddBias + QuantizedSigmoid for the 4 nodes x NF frames of this accumulator group:
      // all 4*NF table indices first, then the 4*NF LDS byte gathers in one batch (the
      // gathers cannot be hoisted over the tile stores by the compiler -- both live in
      // the same LDS array -- and a wait per 4 gathers was the bulk of the epilogue).
      uint8_t act[NF][4];
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int f = fw0 + 32 * ni + frow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int av = acc[mi][ni][g * 4 + q];
          if (TAP && f < p.n && nb + q < p.rows) p.tap_acc[static_cast<size_t>(f) * p.rows + nb + q] = av;
          const float lin = dequant<FAST>(av, p.coef, p.rcp_coef) + bj[q];
          int idx;
          if (FAST) {
            // RN(lin*200) = 2*RN(lin*100) exactly; trunc -> half-step index (see fdnn_model.cpp)
            const int u = static_cast<int>(lin * 200.0f);
            idx = max(-kLut2Half, min(kLut2Half, u)) + kLut2Half;
          } else {
            idx = lut_index(lin);
          }
#if FDNN_GEMM_DEBUG & 8
          act[ni][q] = static_cast<uint8_t>(idx);
#else
          act[ni][q] = lut[idx];
#endif
        }
      }
      // Park the bytes in the LDS image of the output tile ([frame][256 nodes], row stride
      // kTS); the tile leaves as whole 256-byte rows below.  A direct dword store would
      // touch 32 different rows (8 bytes each) per wave instruction and is
      // address-processing bound.
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const uint32_t packed = static_cast<uint32_t>(act[ni][0]) | static_cast<uint32_t>(act[ni][1]) << 8 |
                                static_cast<uint32_t>(act[ni][2]) << 16 | static_cast<uint32_t>(act[ni][3]) << 24;
        *reinterpret_cast<uint32_t *>(tile_s + (arow0 + 32 * ni + frow) * kTS + (nb - m0)) = packed;
      }
    }
  }
  if (!OUTPUT) {
    // the s8 activation tile: FT rows x 256 bytes, written as 16 bytes per lane
    __syncthreads();
    for (int item = tid; item < FT * (G_BM / 16); item += Cfg::THREADS) {
      const int row = item / (G_BM / 16), ch = item % (G_BM / 16);
      const uint4 v = *reinterpret_cast<const uint4 *>(tile_s + row * kTS + ch * 16);
#if FDNN_GEMM_DEBUG & 16
      if (v.x == 0x12345678u && row == -7)
#else
      if (m0 + ch * 16 < p.rows)  // rows is a multiple of 16
#endif
      {
        int8_t *dst = p.act_out + static_cast<size_t>(f0 + row) * p.act_ld + m0 + ch * 16;
        if (FDNN_WT & 1)
          store_wt(dst, v4i{static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)});
        else
          *reinterpret_cast<uint4 *>(dst) = v;
      }
    }
  }
  if (OUTPUT) {
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const float tot = psum[ni] + __shfl_xor(psum[ni], 32);
      const int f = fw0 + 32 * ni + frow;
      if (half == 0) p.partial[static_cast<size_t>(mt * WM + wm) * p.partial_ld + f] = tot;
    }
  }
#if FDNN_GEMM_DEBUG & 64
  FDNN_TS(5);
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
#if FDNN_GEMM_DEBUG & 128
  if (!OUTPUT && tid == 0) printf("B %d %llu %llu\n", blockIdx.x, rt0, rt1);
#else
  if (tid == 0 && (blockIdx.x % (OUTPUT ? 149 : 37)) == 0)
#endif
#if !(FDNN_GEMM_DEBUG & 128)
    printf("%s %4d  rt %llu (+%llu) | prologue %6lld  first-stage %6lld  mainloop %7lld (fix %d: %lld)  lut %6lld  epilogue %7lld  total %7lld cyc\n",
           OUTPUT ? "OUT" : "blk", blockIdx.x, rt0, rt1 - rt0, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], n_fix_done, ts_fix, ts[4] - ts[3], ts[5] - ts[4],
           ts[5] - ts[0]);
#endif
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int NF, int WN, int BK, int STAGES, bool OUTPUT, bool FAST = true, int WM = 4>
void launch_cfg(const QGemmParams &p, hipStream_t s) {
  using Cfg = GemmCfg<NF, WN, BK, STAGES, WM>;
  const int MT = p.rows_pad / Cfg::G_BM, NT = p.n_pad / Cfg::FT;
  const int blocks = 8 * MT * ((NT + 7) / 8);
  auto k_prod = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, false, false, false, WM>;
  auto k_tap = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, true, FAST, false, false, false, WM>;
  auto k_plain = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, OUTPUT, false, false, WM>;  // hidden layers: same as k_prod
  auto k_masked = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, OUTPUT, OUTPUT, false, WM>;
  auto k_anyw = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, OUTPUT, false, OUTPUT, WM>;
  auto k_masked_anyw = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, OUTPUT, OUTPUT, OUTPUT, WM>;
  constexpr bool kCanFuse = OUTPUT && FAST && WM == 4 && NF >= 4;  // the 8-wave shapes and the 4-wave 128- / 160-frame shapes (two workgroups per CU)
  auto k_fused = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, false, false, WM, kCanFuse>;  // (= k_plain where it cannot)
  auto k_fused_masked = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, kCanFuse, false, WM, kCanFuse>;
  auto k_fused_anyw = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, false, kCanFuse, WM, kCanFuse>;  // widths % 32 != 0
  auto k_fused_masked_anyw = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, kCanFuse, kCanFuse, WM, kCanFuse>;
  // layers without saturating pairs, the 8-wave production shapes: no entry walk in the loop (the other shapes: the same kernels)
#ifdef FDNN_NO_NOFIX  // (measurement builds: the instances with the walk for every layer)
  constexpr bool kNoFix = false;
#else
  constexpr bool kNoFix = FAST && WM == 4 && WN == 2 && NF >= 4;
#endif
  auto k_prod_nofix = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, false, false, false, WM, false, kNoFix && !OUTPUT>;
  auto k_fused_nofix = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, false, false, WM, kCanFuse, kNoFix && kCanFuse>;
  auto k_fused_masked_nofix = qgemm_kernel<NF, WN, BK, STAGES, OUTPUT, false, FAST, kCanFuse, kCanFuse, false, WM, kCanFuse, kNoFix && kCanFuse>;
  // the attribute is per device: a process may hold models on several GPUs
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tap), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_masked), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_anyw), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_masked_anyw), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_masked), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_anyw), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_masked_anyw), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod_nofix), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_nofix), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_masked_nofix), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  if (kCanFuse && p.fuse_s != nullptr) {
    // fused soft-max: the node tiles of a frame tile are consecutive blocks (what a workgroup that gave up waiting left
    // unscaled is scaled by its frame tile's last workgroup: one launch)
    if ((p.rows & 31) != 0)
      hipLaunchKernelGGL(p.mask ? k_fused_masked_anyw : k_fused_anyw, dim3(MT * NT), dim3(Cfg::THREADS), Cfg::LDS, s, p);
    else if (kNoFix && p.fix_ent == nullptr)
      hipLaunchKernelGGL(p.mask ? k_fused_masked_nofix : k_fused_nofix, dim3(MT * NT), dim3(Cfg::THREADS), Cfg::LDS, s, p);
    else
      hipLaunchKernelGGL(p.mask ? k_fused_masked : k_fused, dim3(MT * NT), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  } else if (p.tap_acc)
    hipLaunchKernelGGL(k_tap, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else if (OUTPUT && p.mask == nullptr && (p.rows & 31) == 0)
    hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else if (OUTPUT && p.mask == nullptr)
    hipLaunchKernelGGL(k_anyw, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else if (OUTPUT && p.mask != nullptr && (p.rows & 3) == 0)
    hipLaunchKernelGGL(k_masked, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else if (OUTPUT && p.mask != nullptr)  // 8001 nodes: 0.33 ms against 0.41 through the general epilogue
    hipLaunchKernelGGL(k_masked_anyw, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else if (kNoFix && !OUTPUT && p.fix_ent == nullptr)
    hipLaunchKernelGGL(k_prod_nofix, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
  else
    hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(Cfg::THREADS), Cfg::LDS, s, p);
}

template <bool OUTPUT>
void launch_qgemm(const QGemmParams &p, hipStream_t s) {
  if (!p.fastdiv) {  // layer whose coefficient failed the exact-division check (e.g. 127/0 = inf)
    launch_cfg<4, 1, 64, 3, OUTPUT, false>(p, s);
    return;
  }
  static const int small_bk = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_SMALL_BK");
    return e ? std::atoi(e) : 128;
  }();
  switch (p.frame_tile) {
    // few frames: 32- / 64-frame tiles put four / two times as many workgroups on the chip; 128-byte
    // k-steps (3-stage ring) halve the barriers of the latency-bound loop: 16.6 vs 21 us per
    // 2048 x 2048 layer.  What is left at this size is mostly the launch itself: a 128-node tile
    // (half the operand traffic per workgroup) and whole-step fragment prefetch, both tried, left
    // the 16.6 us untouched.
    // ... and while the 32-frame tiles leave CUs idle or nearly so, a launch waits for ONE workgroup's operand stream: 288
    // rows x 2 KiB at the ~70 GB/s one CU's LDS-DMA path moves = 8 of the 17 us of a 2048 x 2048 layer, whatever the frame
    // count.  64-node tiles (one wave per workgroup) split the same weight rows over four times as many CUs: six hidden
    // layers 101-110 -> 88-98 us from 8 to 700 frames (tools/batch_sweep.py; up to three workgroups per CU, beyond that
    // the extra activation traffic loses).  Hidden layers only: the output layer's exp / transposition epilogue makes its
    // narrow tiles slower (27 vs 21 us).
    case 32: {
      static const int force_wm = [] {
        const char *e = FDNN_TUNE_ENV("FDNN_SMALL_WM");
        return e ? std::atoi(e) : 0;
      }();
      const long wgs256 = static_cast<long>(p.rows_pad / 256) * (p.n_pad / 32);
      const int wm = force_wm ? force_wm : (!OUTPUT && wgs256 * 4 <= 768) ? 1 : 4;
      if (small_bk == 128 && wm == 1)
        launch_cfg<1, 1, 128, 4, OUTPUT, true, 1>(p, s);
      else if (small_bk == 128)
        launch_cfg<1, 1, 128, FDNN_SMALL_STAGES, OUTPUT>(p, s);
      else
        launch_cfg<1, 1, 64, 6, OUTPUT>(p, s);
      break;
    }
    case 64:
      if (small_bk == 128)
        launch_cfg<2, 1, 128, 3, OUTPUT>(p, s);
      else
        launch_cfg<2, 1, 64, 6, OUTPUT>(p, s);
      break;
    // 4 waves, 64-byte k-step, 3-stage ring, two workgroups per CU; when every workgroup has
    // a CU of its own anyway (small batches: one latency-bound k-loop per launch) a 6-stage
    // ring hides twice the load latency per step
    case 128:
      if (p.node_tile == 128) {  // 128 nodes x 128 frames, 2 x 2 waves, double-buffered 128-byte k-steps, two workgroups per CU
        launch_cfg<2, 2, 128, 2, OUTPUT, true, 2>(p, s);
        break;
      }
      if (static_cast<long>(p.rows_pad / 256) * (p.n_pad / 128) <= 256 && !p.tap_acc) {
        if (small_bk == 128)
          launch_cfg<4, 1, 128, 3, OUTPUT>(p, s);
        else
          launch_cfg<4, 1, 64, 6, OUTPUT>(p, s);
      }
      else
        launch_cfg<4, 1, 64, 3, OUTPUT>(p, s);
      break;
    case 160: launch_cfg<5, 1, 64, 3, OUTPUT>(p, s); break;
    // 8 waves, 128-byte k-step (whole cache lines), double buffer, one workgroup per CU
    case 256: launch_cfg<4, 2, 128, 2, OUTPUT>(p, s); break;
    default: launch_cfg<5, 2, 128, 2, OUTPUT>(p, s); break;  // 320
  }
}

}  // namespace

int qgemm_debug_flags() {
  static const int flags = [] {
    const char *e = std::getenv("FDNN_GEMM_DEBUG") /* test hook: bit 4096 makes fused soft-max tiles give up (tests) */;
    return e ? std::atoi(e) : 0;
  }();
  return flags;
}

int qgemm_frame_tile(int rows_pad, int n) {
  static const int forced = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_FRAME_TILE");
    return e ? std::atoi(e) : 0;
  }();
  if (forced == 32 || forced == 64 || forced == 128 || forced == 160 || forced == 256 || forced == 320) return forced;
  const int mt = rows_pad / 256;
  // Few frames: while every workgroup gets a CU of its own the launch is one k-loop deep and
  // latency bound, so the smallest tile that still fits in one round wins -- it has the shortest
  // k-step and puts the most CUs to work (2048 x 2048 layer, 1000 frames: 21 us with 32-frame
  // tiles = 256 workgroups, 24 us at 64, 31 us at 128; 8000-node output layer, 1000 frames:
  // 39 us at 128 = 256 workgroups, 47 us at 64, 65 us at 32 = four rounds).
  for (int ft : {32, 64, 128})
    if (static_cast<long>(mt) * ((n + ft - 1) / ft) <= 256) return ft;
  // Cost model: rounds x frames per tile / relative throughput of the kernel shape.
  // A round fills every CU once (two co-resident workgroups for the 4-wave shapes).
  struct Cand {
    int ft, slots;
    double eff;
  };
  const Cand cands[] = {{128, 512, 0.55}, {160, 512, 0.55}, {256, 256, 1.0}, {320, 256, 1.0}};
  int best = 128;
  double best_cost = -1.0;
  for (const Cand &c : cands) {
    const long blocks = static_cast<long>(mt) * ((n + c.ft - 1) / c.ft);
    const long rounds = (blocks + c.slots - 1) / c.slots;
    // a 4-wave workgroup shares its CU with a second one: a round costs two tiles' time
    const double cost = rounds * c.ft * (c.slots == 512 ? 2.0 : 1.0) / c.eff;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && c.ft > best)) {
      best_cost = cost;
      best = c.ft;
    }
  }
  return best;
}

// Batches up to this many frames take the small-batch kernel (fdnn_small.hip) where the layer allows it.  Measured
// crossovers on the 2048-wide layers (tools/batch_sweep.py, FDNN_SMALL_MAX=0 against the default): six hidden layers
// (64-node tiles from ~160 frames up) 54 vs 88 us at 256 frames, 68 vs 90 at 512, 93 vs 104 at 1000, 104 vs 118 at
// 1200, 117 vs 118 at 1500, 137 vs 121 at 2000; the 8000-node output layer 16.5 vs 24.6 at 256, 25.8 vs 26.3 at 512,
// 44 + 15 (scale pass) vs 40 (fused) at 1000 (a workgroup of the small kernel walks its frame tiles one after the other).
bool qgemm_small_pick(int rows_pad, int K, int n, int fastdiv, bool output) {
  static const int small_max = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_SMALL_MAX");
    return e ? std::atoi(e) : -1;
  }();
  (void)rows_pad;
  const int lim = small_max >= 0 ? small_max : output ? 512 : 1400;
  return n <= lim && qgemm_small_ok(K, fastdiv);
}

// Mid-size batches of the hidden layers: when the layer is between one and two rounds of 128 x 128 tiles (2 049 .. 4 096
// frames on a 2048-node layer), the four-wave 128 x 128 shape -- two workgroups per CU, so one's prologue / epilogue
// hides under the other's k-loop -- beats the 256-node tiles of the same area (tools/batch_sweep.py, six hidden layers:
// 140 vs 158 us at 2 560 frames, 141 vs 158 at 3 000, 151 vs 162 at 4 000; it loses below (123 vs 119 at 2 000: one
// workgroup per CU again) and above (229 vs 202 at 5 000), and on the 8000-node output layer).  Returns 128 or 256;
// with 128 the frame tile is 128 as well.
bool qgemm_fused_ok(const QGemmParams &p) {
  static const bool off = [] {
    const char *e = std::getenv("FDNN_FUSE_NORM");
    return e && std::atoi(e) == 0;
  }();
  if (off || p.small || p.node_tile != 256 || !p.fastdiv || (p.mask && !p.mask_bits) || p.tap_acc || p.tap_logit) return false;
  if (p.frame_tile != 320 && p.frame_tile != 256 && p.frame_tile != 160 && p.frame_tile != 128) return false;
  const int MT = p.rows_pad / 256;
  int L = 1;
  while (L < MT) L <<= 1;
  // the epilogue's LDS: one 32 x 64 float tile per wave, then 4 partial rows + the inverses + L rows of S, below the
  // table / bias area (GemmCfg::FIX_OFF: two 128-byte-step stages for the 8-wave shapes, three 64-byte-step stages
  // for the 4-wave ones)
  const bool eight = p.frame_tile >= 256;
  const long need = 8192 + (eight ? 8 : 4) * 32 * 68 * 4 + (5L * p.frame_tile + 4 + static_cast<long>(L) * p.frame_tile) * 4;
  const long have = static_cast<long>(256 + p.frame_tile) * (eight ? 128 * 2 : 64 * 3);
  return L <= 32 && need <= have;
}

int qgemm_node_tile(int rows_pad, int n, bool output) {
  static const int forced = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_NODE_TILE");
    return e ? std::atoi(e) : 0;
  }();
  if (forced == 128 || forced == 256) return forced;
  if (output) return 256;
  const long tiles = static_cast<long>(rows_pad / 128) * ((n + 127) / 128);
  return (tiles > 256 && tiles <= 512) ? 128 : 256;
}

void launch_qgemm_hidden(const QGemmParams &p, hipStream_t s) {
  if (p.small)
    launch_qgemm_small_hidden(p, s);
  else
    launch_qgemm<false>(p, s);
}
void launch_qgemm_output(const QGemmParams &p, hipStream_t s) {
  if (p.small)
    launch_qgemm_small_output(p, s);
  else
    launch_qgemm<true>(p, s);
}

}  // namespace fdnn
