// fdnn_small.hip -- the int8 layer for SMALL batches (one utterance = 100 frames, the reference's own call
// shape: QuantizedDnn.java:149-167 batch 10, MultiThreadedStressTest.java:48-61, FuncTest.java:40-56).
//
//   QuantizedLayerActivations/quantizedNodeSum + AddBias + QuantizedSigmoid (dnn.cc:289-349, :250-286)
//   CalculateOutput / LazyOutputActivations + SoftMax first loop (dnn.cc:428-454, :355-392, :534-540)
//
// At 100 frames a 2048 x 2048 layer is 0.84 G int8 ops -- 0.4 us of MFMA time -- against 4 MB of weights and
// 200 KB of activations: the launch is bound by how fast ONE workgroup can pull its operands through its CU's
// L2 -> LDS path (~42 B/clk/CU, tools/ubench_dma_waves.hip) and by the latencies in front of and behind that
// stream.  The large-batch kernel (fdnn_gemm.hip) is the wrong shape for this: 256-node tiles put 8..32
// workgroups on the chip, each streaming 512 KB of weights through a k-step ring with a barrier per step
// (14.7 us per layer at 100 frames).  Here instead
//
//   * tiles are 32 nodes x 32 frames (hidden layers: 64 node tiles x 4 frame tiles = 256 workgroups at 100
//     frames, 64 KB + 64 KB of operands each -- the split that minimises the bytes per CU when every CU has one
//     tile) or 64 nodes x 32 frames (output layer: the soft-max partial sums are per 64 nodes and their
//     summation order is part of the batch-size invariance contract, see below);
//   * the whole K extent of the tile is loaded AT ONCE: every one of the 8 waves owns a 256-byte slice of K
//     (in-workgroup split-K), issues the LDS-DMA loads of its own slice of W and A up front and waits only for
//     its own loads -- no ring, no barrier in front of the MFMAs;
//   * W stays in REGISTERS (32 or 64 VGPRs of MFMA fragments per wave): a workgroup that owns several frame
//     tiles (larger batches, the output layer) streams only activation tiles, double buffered, through the LDS;
//   * the eight partial 32 x 32 int32 tiles meet in LDS (integer sums: any order is exact), 4 nodes x 1 frame
//     per thread, and leave through the same dequantise / bias / table (or exp) arithmetic as the large-batch
//     kernel, operation for operation -- so a frame's bits do not depend on the batch it was scored in
//     (tests/test_gpu_server.py compares coalesced batches with per-utterance calls bit for bit).
//
// pmaddubsw saturation (dnn.cc:337-340): the same sparse exact correction as in fdnn_gemm.hip, applied by the
// wave whose K slice holds the pair, from the staged activation bytes.
#include <atomic>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace fdnn {
namespace {

constexpr int kSmWaves = 8;
constexpr int kSmThreads = 64 * kSmWaves;
constexpr int kSmSlice = 256;                      // bytes of K per wave
constexpr int kSmFT = 32;                          // frames per tile
constexpr int kSmABuf = kSmWaves * kSmFT * kSmSlice;  // one activation tile: 64 KiB
constexpr int kSmAuxOff = 2 * kSmABuf;             // table (3 KiB) | biases (256 B) | 128 * sum(w) (256 B) | e scratch (8 KiB)
[[maybe_unused]] constexpr int kSmBiasOff = kSmAuxOff + 3072;  // (used by device code only)
[[maybe_unused]] constexpr int kSmWsumOff = kSmAuxOff + 3072 + 256;
constexpr int kSmEOff = kSmAuxOff + 3584;
constexpr int kSmLds = kSmEOff + 8192 + 64;
static_assert(kSmLds <= 160 * 1024, "LDS");

// chunk c (16 bytes) of row r of a 256-byte-row LDS image lives at chunk position c ^ (r & 15): the 16 lanes of a
// ds_read_b128 group read 16 different rows at the same k and land on 16 different 16-byte slots
[[maybe_unused]] __device__ __forceinline__ int sm_pos(int row, int chunk) { return (row << 8) + (((chunk ^ row) & 15) << 4); }

// NTM: 32-node MFMA tiles per workgroup (1: hidden layers, 2: output layer).
template <int NTM, bool OUTPUT, bool TAP, bool MASKED>
__global__ __launch_bounds__(kSmThreads, 2) void qgemm_small_kernel(QGemmParams p, int groups, int tiles_per_group) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 32 * NTM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FDNN_SM_CLK
  long long tc[8];
  tc[0] = __builtin_readcyclecounter();
#define SM_TS(i) tc[i] = __builtin_readcyclecounter()
#else
#define SM_TS(i)
#endif

  // Workgroup b runs on XCD b % 8 (a locality hint, nothing depends on it): the frame groups of one node tile
  // go to one XCD, so a weight tile is pulled into one L2.
  const int MT = p.rows_pad / NT;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int mt = (j / groups) * 8 + xcd, fg = j % groups;
  if (mt >= MT) return;
  const int m0 = mt * NT;
  const int n_tiles = p.n_pad / kSmFT;
  const int t_begin = fg * tiles_per_group;
  const int t_end = min(n_tiles, t_begin + tiles_per_group);
  if (t_begin >= t_end) return;
  if (OUTPUT && m0 >= p.rows) {  // a node tile of pure padding (rows_pad is a multiple of 256): its partial sums are zero
    for (int i = tid; i < (t_end - t_begin) * kSmFT; i += kSmThreads) p.partial[static_cast<size_t>(m0 >> 6) * p.partial_ld + t_begin * kSmFT + i] = 0.0f;
    return;
  }

  // this wave's K slice; lanes past the layer's K (K is a multiple of 128, the slice 256) fetch nothing
  const int k0 = wave * kSmSlice;
  const bool slice_live = k0 < p.K;
  const int r4 = lane >> 4, c16 = lane & 15;
  char *const abuf0 = smem + wave * (kSmFT * kSmSlice);  // this wave's 8 KiB of activation buffer 0; buffer 1 at + kSmABuf
  // per-lane offsets of the four row phases of a 1-KiB load (rows 4i + r4, i & 3 = ph): SOURCE chunk = lane chunk XOR row;
  // a source chunk past the layer's K is not fetched (an out-of-range offset reads as zeros)
  int voff_w[4], voff_a[4];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    const int ch = ((c16 ^ r4 ^ (4 * ph)) & 15) << 4;
    const bool live = k0 + ch < p.K;
    voff_w[ph] = live ? r4 * p.ldw + ch : 0x7ffffff0;
    voff_a[ph] = live ? r4 * p.lda + ch : 0x7ffffff0;
  }
  // NB the sigmoid table's last 16-byte piece must be inside num_records as a whole (fdnn_gemm.hip)
  char *aux = smem + kSmAuxOff;
  if (!OUTPUT && wave < 3) {
    const int bytes = (kLut2Size + 15) & ~15;
    const __amdgpu_buffer_rsrc_t rsrc_lut = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.lut2), 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lut, FDNN_LDS_PTR(aux + wave * 1024), 16, lane * 16, wave * 1024, 0, 0);
  }
  if (wave == 3 && lane < NT / 4) {
    const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias + m0), 0, NT * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_bias, FDNN_LDS_PTR(smem + kSmBiasOff), 16, lane * 16, 0, 0, 0);
    // 128 * sum_k w[node][k] (the s8 = u8 - 128 activation offset) is added where the partial tiles meet
    const __amdgpu_buffer_rsrc_t rsrc_wsum = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.wsum + m0), 0, NT * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wsum, FDNN_LDS_PTR(smem + kSmWsumOff), 16, lane * 16, 0, 0, 0);
  }
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(p.w + static_cast<size_t>(m0) * p.ldw), 0, NT * p.ldw, 0x00020000);
  // activation rows past the batch read as zeros (no traffic): the descriptor ends at frame n
  auto a_rsrc = [&](int t) {
    const int f0 = t * kSmFT;
    // (readfirstlane: the compiler evaluates the clamp as a vector v_med3, and a descriptor held in VGPRs turns every
    // load into a waterfall loop)
    const int bytes = __builtin_amdgcn_readfirstlane(max(0, min(kSmFT, p.n - f0)) * p.lda);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.a + static_cast<size_t>(f0) * p.lda), 0, bytes, 0x00020000);
  };
  auto load_w = [&](int half, char *dst) {  // 32 weight rows x this wave's slice -> dst (8 KiB, wave private)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, FDNN_LDS_PTR(dst + i * 1024), 16, voff_w[i & 3], (32 * half + 4 * i) * p.ldw + k0, 0, 0);
  };
  auto load_a = [&](int t, char *dst) {
    const __amdgpu_buffer_rsrc_t rsrc_a = a_rsrc(t);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, FDNN_LDS_PTR(dst + i * 1024), 16, voff_a[i & 3], 4 * i * p.lda + k0, 0, 0);
  };
  const int frow = lane & 31, fch = lane >> 5;
  v4i wf[NTM][8];
  auto read_w = [&](int half, const char *src) {
#pragma unroll
    for (int s = 0; s < 8; ++s) wf[half][s] = *reinterpret_cast<const v4i *>(src + sm_pos(frow, 2 * s + fch));
  };

  SM_TS(1);
  // ---- prologue: W through the (not yet needed) second activation buffer into registers, the first activation tile
  // beside it.  64-node shape: the second half of W follows through the same buffer while the first half's MFMAs run
  // (see the first pass of the tile loop) -- both halves at once would need both buffers, and the activation tile
  // could only be requested once they had landed.
  if (slice_live) {
    load_w(0, abuf0 + kSmABuf);
    load_a(t_begin, abuf0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // older loads complete first: table / bias / 128 sum(w) + W half 0 landed
    read_w(0, abuf0 + kSmABuf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (NTM == 1) {
      if (t_begin + 1 < t_end) load_a(t_begin + 1, abuf0 + kSmABuf);
    } else {
      load_w(1, abuf0 + kSmABuf);
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a wave without a slice may still have issued table / bias loads
#pragma unroll
    for (int h = 0; h < NTM; ++h)
#pragma unroll
      for (int s = 0; s < 8; ++s) wf[h][s] = v4i{0, 0, 0, 0};
  }

  // pmaddubsw saturation entries of this tile's 64-node group whose pair lies in this wave's K slice: sorted by k
  // inside the group, so they are one contiguous run.  Its start is found from an interpolated guess (k is spread
  // evenly: a walk from the group's first entry cost the last wave ~60 dependent scalar loads), and the run's first
  // kFixPre entries are fetched into SGPRs here, under the operand loads' flight time.
  typedef const __attribute__((address_space(4))) uint64_t *FixPtr;
  const FixPtr ent_c = (FixPtr)(uintptr_t)p.fix_ent;  // {u16 k, s8 w0, s8 w1, s32 node}
  constexpr int kFixPre = 8;
  int fix_b = 0, fix_end = 0;
  uint64_t fix_pre[kFixPre];
#pragma unroll
  for (int i = 0; i < kFixPre; ++i) fix_pre[i] = 0xffffull;  // k = 65535: past every slice
  if (p.fix_ent && slice_live) {
    const int grp = m0 >> 6;
    typedef const __attribute__((address_space(4))) int *GrpPtr;
    const GrpPtr grp_c = (GrpPtr)(uintptr_t)p.fix_grp;  // scalar loads: no vector-memory wait beside the LDS-DMA queue
    const int g_begin = grp_c[grp];
    fix_end = grp_c[grp + 1];
    int e = g_begin + static_cast<int>(static_cast<long long>(fix_end - g_begin) * k0 / p.K);
    while (e > g_begin && static_cast<int>(ent_c[e - 1] & 0xffff) >= k0) --e;
    while (e < fix_end && static_cast<int>(ent_c[e] & 0xffff) < k0) ++e;
    fix_b = e;
#pragma unroll
    for (int i = 0; i < kFixPre; ++i)
      if (fix_b + i < fix_end) fix_pre[i] = ent_c[fix_b + i];
  }

  const uint8_t *lut = reinterpret_cast<const uint8_t *>(aux);
  const float *bias_s = reinterpret_cast<const float *>(smem + kSmBiasOff);
  float *e_s = reinterpret_cast<float *>(smem + kSmEOff);  // OUTPUT: e values [frame][64 nodes] for the ordered partial sums

  SM_TS(2);
  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    char *at = abuf0 + cur * kSmABuf;
    const int f0 = t * kSmFT;
    // D layout (32x32): column (frame) = lane & 31, row (node) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    v16i acc[NTM];
#pragma unroll
    for (int h = 0; h < NTM; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0;
    if (slice_live) {
      v4i b[8];
      auto read_b = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const v4i *>(at + sm_pos(frow, 2 * s + fch));
      };
      if (NTM == 2 && t == t_begin) {
        // first pass of the 64-node shape: in flight are this tile (8 loads) and W half 1 (8 loads, younger)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (t == t_begin) SM_TS(3);
        read_b();
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][s], b[s], acc[0], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_w(NTM - 1, abuf0 + kSmABuf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < t_end) load_a(t + 1, abuf0 + kSmABuf);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[NTM - 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[NTM - 1][s], b[s], acc[NTM - 1], 0, 0, 0);
      } else {
        if (t + 1 < t_end)
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile t landed, tile t+1 (8 loads) may still fly
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == t_begin) SM_TS(3);
        read_b();
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int h = 0; h < NTM; ++h) acc[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[h][s], b[s], acc[h], 0, 0, 0);
      }
      // saturating pairs (rare): the reference clamps a[2j]*w[2j] + a[2j+1]*w[2j+1] to int16 (dnn.cc:337-340)
      auto fix_one = [&](uint64_t raw) {
        const int node = static_cast<int>(raw >> 32) - m0;
        if (node < 0 || node >= NT) return;  // NTM == 1: the other half of the 64-node group
        const int kl = static_cast<int>(raw & 0xffff) - k0;  // even, 0..254
        const int w0 = static_cast<int8_t>(raw >> 16), w1 = static_cast<int8_t>(raw >> 24);
        const uint32_t pair = *reinterpret_cast<const uint16_t *>(at + sm_pos(frow, kl >> 4) + (kl & 15));
        const int a0 = static_cast<int>((pair & 0xff) ^ 0x80), a1 = static_cast<int>((pair >> 8) ^ 0x80);  // back to u8
        const int prod = a0 * w0 + a1 * w1;
        const int rr = node & 31;
        const bool mine = fch == ((rr >> 2) & 1);
        const int c = mine ? max(-32768, min(32767, prod)) - prod : 0;
        if (__ballot(c != 0) != 0ull) {
          const int idx = (node >> 5) * 16 + (rr & 3) + 4 * (rr >> 3);
#pragma unroll
          for (int i = 0; i < 16 * NTM; ++i)
            if (idx == i) acc[i >> 4][i & 15] += c;
        }
      };
      bool more = true;
#pragma unroll
      for (int i = 0; i < kFixPre; ++i) {
        if (more && static_cast<int>(fix_pre[i] & 0xffff) < k0 + kSmSlice)
          fix_one(fix_pre[i]);
        else
          more = false;
      }
      if (more)  // a run longer than the prefetched entries (nets with weights near +-127 everywhere)
        for (int e = fix_b + kFixPre; e < fix_end; ++e) {
          const uint64_t raw = ent_c[e];
          if (static_cast<int>(raw & 0xffff) >= k0 + kSmSlice) break;
          fix_one(raw);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // partial tile -> this wave's own (just consumed) 8 KiB of the activation buffer: [half][frame][32 nodes] int32,
    // 16-byte chunk q of a frame row at position q ^ (frame & 7)
    int *part = reinterpret_cast<int *>(at);
#pragma unroll
    for (int h = 0; h < NTM; ++h)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<v4i *>(reinterpret_cast<char *>(part) + h * 4096 + frow * 128 + ((((2 * g + fch) ^ frow) & 7) << 4)) =
            v4i{acc[h][g * 4], acc[h][g * 4 + 1], acc[h][g * 4 + 2], acc[h][g * 4 + 3]};
    if (t == t_begin) SM_TS(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t == t_begin) SM_TS(5);

    // ---- reduce + epilogue: thread -> (frame f, node quad Q); NTM == 1: threads 0..255, NTM == 2: all 512
    const int f = (tid >> 3) & 31, q = tid & 7, h = tid >> 8;  // quad Q = 8h + q: nodes m0 + 32h + 4q ..+3
    const bool worker = NTM == 2 || tid < 256;
    const int ff = f0 + f;
    const int nb = m0 + 32 * h + 4 * q;
    float e4[4] = {0.f, 0.f, 0.f, 0.f};
    if (worker) {
      v4i sum = *reinterpret_cast<const v4i *>(smem + kSmWsumOff + (32 * h + 4 * q) * 4);
#pragma unroll
      for (int w = 0; w < kSmWaves; ++w)
        sum += *reinterpret_cast<const v4i *>(smem + cur * kSmABuf + w * (kSmFT * kSmSlice) + h * 4096 + f * 128 + (((q ^ f) & 7) << 4));
      const v4f_t b4 = *reinterpret_cast<const v4f_t *>(bias_s + 32 * h + 4 * q);
      const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
      const int sv[4] = {sum.x, sum.y, sum.z, sum.w};
      if (!OUTPUT) {
        // AddBias + QuantizedSigmoid (half-step table; the layer passed the exact-division and range checks at load)
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (TAP && ff < p.n && nb + i < p.rows) p.tap_acc[static_cast<size_t>(ff) * p.rows + nb + i] = sv[i];
          const float lin = dequant<true>(sv[i], p.coef, p.rcp_coef) + bj[i];
          const int u = static_cast<int>(lin * 200.0f);
          const int idx = max(-kLut2Half, min(kLut2Half, u)) + kLut2Half;
          packed |= static_cast<uint32_t>(lut[idx]) << (8 * i);
        }
        if (nb < p.rows)  // rows is a multiple of 16
          *reinterpret_cast<uint32_t *>(p.act_out + static_cast<size_t>(ff) * p.act_ld + nb) = packed;
      } else {
        // z = sum/coef + bias (masked-out nodes keep z = 0, dnn.cc:366-369), e = exp(z) (dnn.cc:536-540): the
        // operations of the large-batch kernel's dense instance, one for one
        if (p.acc_probe != nullptr && ff < p.n && ff % p.probe_stride == 0)  // parity tests only (see QGemmParams)
          for (int i = 0; i < 4; ++i)
            if (nb + i < p.rows) p.acc_probe[static_cast<size_t>(ff / p.probe_stride) * p.rows + nb + i] = sv[i];
        uint32_t mbits = 0x01010101u;
        if (MASKED && p.mask != nullptr && ff < p.n) {
          mbits = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < p.rows && p.mask[static_cast<size_t>(ff) * p.rows + nb + i]) mbits |= 0xffu << (8 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (TAP && ff < p.n && nb + i < p.rows) p.tap_acc[static_cast<size_t>(ff) * p.rows + nb + i] = sv[i];
          const float x = static_cast<float>(sv[i]);
          const float q0 = x * p.rcp_coef;
          const float r = fmaf(-q0, p.coef, x);
          float z = fmaf(r, p.rcp_coef, q0) + bj[i];
          if (MASKED && ((mbits >> (8 * i)) & 0xffu) == 0) z = 0.0f;
          if (TAP && ff < p.n && nb + i < p.rows) p.tap_logit[static_cast<size_t>(ff) * p.rows + nb + i] = z;
          const float y = z * 1.44269504088896340736f;
          e4[i] = nb + i < p.rows ? __builtin_amdgcn_exp2f(y) : 0.0f;
        }
        if (ff < p.n) {
          float *op = p.out + static_cast<size_t>(ff) * p.rows + nb;
          if (nb + 4 <= p.rows) {
            typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
            if ((p.rows & 31) == 0 && !MASKED)
              store_wt(op, v4f_t{e4[0], e4[1], e4[2], e4[3]});
            else
              *reinterpret_cast<v4f_a4 *>(op) = v4f_a4{e4[0], e4[1], e4[2], e4[3]};
          } else {
            for (int i = 0; i < 4; ++i)
              if (nb + i < p.rows) op[i] = e4[i];
          }
        }
        *reinterpret_cast<v4f_t *>(e_s + f * 64 + 32 * h + 4 * q) = v4f_t{e4[0], e4[1], e4[2], e4[3]};
      }
    }
    if (t == t_begin) SM_TS(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every partial is read: the buffer may be refilled; OUTPUT: e_s is complete
    asm volatile("" ::: "memory");
    if (t + 2 < t_end && slice_live) load_a(t + 2, at);
    if (OUTPUT && wave == 0) {
      // the 64-node partial sum in the large-batch kernel's order: lane half hh takes the node quads
      // 32 mi + 8 g + 4 hh (mi, g ascending), four sequential adds each; total = half 0 + half 1
      float ps = 0.0f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const v4f_t v = *reinterpret_cast<const v4f_t *>(e_s + frow * 64 + 32 * mi + 8 * g + 4 * fch);
          ps += v.x;
          ps += v.y;
          ps += v.z;
          ps += v.w;
        }
      const float tot = ps + __shfl_xor(ps, 32);
      if (fch == 0) p.partial[static_cast<size_t>(m0 >> 6) * p.partial_ld + f0 + frow] = tot;
      // e_s is rewritten only after the next tile's first barrier, which this wave reaches after these reads
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
#ifdef FDNN_SM_CLK
  SM_TS(7);
  if ((tid == 0 || tid == 448) && (blockIdx.x % 97) == 0)
    printf("%s blk %d w%d: setup %lld  prologue %lld  wait-A %lld  mfma+part %lld  barrier %lld  epi %lld  rest %lld  total %lld\n", OUTPUT ? "OUT" : "hid",
           blockIdx.x, wave, tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2], tc[4] - tc[3], tc[5] - tc[4], tc[6] - tc[5], tc[7] - tc[6], tc[7] - tc[0]);
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int NTM, bool OUTPUT>
void launch_small_cfg(const QGemmParams &p, hipStream_t s) {
  auto k_prod = qgemm_small_kernel<NTM, OUTPUT, false, false>;
  auto k_tap = qgemm_small_kernel<NTM, OUTPUT, true, OUTPUT>;
  auto k_masked = qgemm_small_kernel<NTM, OUTPUT, false, OUTPUT>;
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long dev_bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_prod), hipFuncAttributeMaxDynamicSharedMemorySize, kSmLds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_tap), hipFuncAttributeMaxDynamicSharedMemorySize, kSmLds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_masked), hipFuncAttributeMaxDynamicSharedMemorySize, kSmLds);
    attr_set.fetch_or(dev_bit, std::memory_order_release);
  }
  const int NT = 32 * NTM;
  const int MT = p.rows_pad / NT, n_tiles = p.n_pad / kSmFT;
  // frame groups per node tile: as many as keep the launch within one round of workgroups (one per CU), at least one
  const int MT_live = (p.rows + NT - 1) / NT;  // 8000 output nodes: 125 node tiles x 2 frame groups = 250 workgroups
  int groups = std::max(1, std::min(n_tiles, 256 / std::max(1, MT_live)));
  const int tpg = (n_tiles + groups - 1) / groups;
  groups = (n_tiles + tpg - 1) / tpg;
  const int blocks = 8 * ((MT + 7) / 8) * groups;
  if (p.tap_acc)
    hipLaunchKernelGGL(k_tap, dim3(blocks), dim3(kSmThreads), kSmLds, s, p, groups, tpg);
  else if (OUTPUT && p.mask)
    hipLaunchKernelGGL(k_masked, dim3(blocks), dim3(kSmThreads), kSmLds, s, p, groups, tpg);
  else
    hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(kSmThreads), kSmLds, s, p, groups, tpg);
}

}  // namespace

// Small-batch shape available for this layer?  (K up to 8 slices of 256 bytes; the exact-division / bounded-range
// epilogue only; taps of the output layer need a mask-capable instance, which the tap instance is.)
bool qgemm_small_ok(int K, int fastdiv) { return fastdiv && K <= kSmWaves * kSmSlice; }

// Hidden layers: 32-node tiles while a frame tile per workgroup fills the chip (up to ~128 frames on a 2048-node layer);
// beyond, 64-node tiles halve the activation bytes per output (each workgroup then walks fewer frame tiles).
// FDNN_SMALL_NTM=1|2 forces one shape (measurements).
void launch_qgemm_small_hidden(const QGemmParams &p, hipStream_t s) {
  static const int forced = [] {
    const char *e = FDNN_TUNE_ENV("FDNN_SMALL_NTM");
    return e ? std::atoi(e) : 0;
  }();
  const long wg32 = static_cast<long>(p.rows_pad / 32) * (p.n_pad / kSmFT);
  const bool wide = forced ? forced == 2 : wg32 > 320;
  if (wide)
    launch_small_cfg<2, false>(p, s);
  else
    launch_small_cfg<1, false>(p, s);
}
void launch_qgemm_small_output(const QGemmParams &p, hipStream_t s) { launch_small_cfg<2, true>(p, s); }

}  // namespace fdnn
