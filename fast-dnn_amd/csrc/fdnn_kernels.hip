// fdnn_kernels.hip -- soft-max normalisation and small helper kernels for gfx950
// (layer 0 lives in fdnn_l0.hip, the int8 layer kernel in fdnn_gemm.hip).
//
//   SoftMax::apply second loop (dnn.cc:541-543)       -> normalize_kernel
#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace fdnn {
namespace {

// ---------------------------------------------------------------- soft-max normalisation
// SoftMax::apply second loop (dnn.cc:541-543): p_i = e_i / total.  total is the
// sum of the output kernel's per-64-node partials in a fixed order.
// dst == out scales in place; the per-frame lazy call passes a host-mapped dst instead, so the
// probabilities land in the caller's pinned buffer without a copy command.
#ifndef FDNN_NORM_BG_NT
#define FDNN_NORM_BG_NT 1  // non-temporal loads / stores in the background scale kernel: its 640 MB stream through the
                           // L2s that layer 0 (running beside it) keeps its operand images in; +2 % on the overlapped step
#endif
constexpr int kNormMaxTiles = 2048;  // 256-node tiles a row total can span (output layers up to 524 288 nodes)
template <bool NT>
__device__ __forceinline__ void normalize_row(const float *out, float *dst, const float *partial, int f, int partial_ld, int rows,
                                              int n_partial, float *red) {
  const int tid = threadIdx.x;
  // The row total, in THE order every path of this library uses (fdnn_device.hpp: softmax_total_order): the four
  // 64-node partial sums of each 256-node tile first, S_j = (P[4j] + P[4j+1]) + (P[4j+2] + P[4j+3]), then the S_j by
  // adjacent pairs, level by level (zero-padded to a power of two: x + 0 = x).  The large-batch output kernel computes
  // the same tree inside its epilogue (the 256-node tile is its workgroup), which is what lets it write probabilities
  // directly instead of exp(z) for this pass to scale.
  const int MT = n_partial >> 2;  // n_partial = rows_pad / 64, rows_pad a multiple of 256
  int L = 1;
  while (L < MT) L <<= 1;
  for (int t = tid; t < L; t += 256) {
    float sj = 0.0f;
    if (t < MT) {
      const float *pp = partial + static_cast<size_t>(4 * t) * partial_ld + f;
      sj = (pp[0] + pp[partial_ld]) + (pp[2 * static_cast<size_t>(partial_ld)] + pp[3 * static_cast<size_t>(partial_ld)]);
    }
    red[t] = sj;
  }
  __syncthreads();
  for (int len = L >> 1; len >= 1; len >>= 1) {
    float v[kNormMaxTiles / 256];
#pragma unroll
    for (int q = 0; q < kNormMaxTiles / 256; ++q) {
      const int t = tid + 256 * q;
      v[q] = t < len ? red[2 * t] + red[2 * t + 1] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kNormMaxTiles / 256; ++q) {
      const int t = tid + 256 * q;
      if (t < len) red[t] = v[q];
    }
    __syncthreads();
  }
  const float total = red[0];
  // p_i = e_i / total (dnn.cc:541-543) as e_i * RN(1 / total): within one ulp of the quotient
  // (1.2e-7 relative; the hardware exp already differs from glibc's expf by more), and one
  // multiply per element instead of the ~10-instruction IEEE division sequence -- which does not
  // matter to this HBM-bound pass on its own, but is vector-ALU work taken from the next batch's
  // layer 0 when the pass runs underneath it (server loop).
  const float inv = 1.0f / total;
  const float *row = out + static_cast<size_t>(f) * rows;
  float *drow = dst + static_cast<size_t>(f) * rows;
  // Rows of any width (pdf counts are arbitrary): when rows % 4 != 0 a row starts off the 16-byte
  // grid, so up to three head elements are peeled to get there and the body goes as aligned
  // dwordx4 accesses (misaligned dwordx4 accesses work but ran 0.188 ms for an 8001-wide layer,
  // the plain scalar loop 0.153, this 0.108).  Source and destination must share the offset --
  // in place they do; the per-frame lazy call's pinned destination falls back to scalars.
  // write-through stores only when rows are whole cache lines: written through, a line shared by
  // two rows (two workgroups) becomes two partial-line writes to memory (8016-wide layer: 0.187 ms
  // against 0.118 with plain stores; 8000-wide: 0.108 against 0.112)
  const bool wt_rows = (rows & 31) == 0;
  const int head = static_cast<int>((4 - ((reinterpret_cast<uintptr_t>(row) >> 2) & 3)) & 3);
  const bool same = ((reinterpret_cast<uintptr_t>(row) ^ reinterpret_cast<uintptr_t>(drow)) & 15) == 0;
  if (same && rows >= head) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row + head);
    float4 *d4 = reinterpret_cast<float4 *>(drow + head);
    const int groups = (rows - head) >> 2, tail0 = head + 4 * groups;
    for (int i = tid; i < groups; i += 256) {
      float4 v;
      if (NT) {
        const v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(r4 + i));
        v = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v = r4[i];
      }
      v.x = v.x * inv;
      v.y = v.y * inv;
      v.z = v.z * inv;
      v.w = v.w * inv;
      if (NT)
        __builtin_nontemporal_store(v4f_t{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_t *>(d4 + i));
      else if ((FDNN_WT & 32) && wt_rows)
        store_wt(d4 + i, v4f_t{v.x, v.y, v.z, v.w});
      else
        d4[i] = v;
    }
    if (tid < head) drow[tid] = row[tid] * inv;
    if (tid >= 64 && tid - 64 < rows - tail0) drow[tail0 + tid - 64] = row[tail0 + tid - 64] * inv;
  } else {
    for (int i = tid; i < rows; i += 256) drow[i] = row[i] * inv;
  }
}

__global__ __launch_bounds__(256) void normalize_kernel(const float *out, float *dst, const float *partial, int n, int partial_ld,
                                                        int rows, int n_partial) {
  __shared__ float red[kNormMaxTiles];
#ifndef FDNN_NORM_NT
#define FDNN_NORM_NT 0  // non-temporal here too (same box, tools/kstat.py): this pass 105 -> 118 us, the kernels after it
                        // (whose operands it no longer evicts) -12 us together -- nothing in it
#endif
  normalize_row<(FDNN_NORM_NT != 0)>(out, dst, partial, blockIdx.x, partial_ld, rows, n_partial, red);
}

// The pass for SMALL batches (one utterance: 100 rows on 256 CUs, latency bound): the row's e_i are requested first -- eight
// 16-byte loads per thread, nothing they depend on -- and the total is formed meanwhile by ONE wave from registers (the S_j in
// lanes, the tree's levels by lane permutes: the same additions in the same order as normalize_row, no barrier per level);
// one barrier, scale, store.  Three dependent round trips and eleven barriers become two and one: 6.2 -> see LABBOOK (100 rows).
// Rows of whole 16-byte groups, at most 8 192 wide, at most 64 tiles of 256 nodes; anything else takes normalize_kernel.
__global__ __launch_bounds__(256) void normalize_small_kernel(const float *out, float *dst, const float *partial, int n, int partial_ld,
                                                              int rows, int n_partial) {
  __shared__ float tot_s;
  const int tid = threadIdx.x, f = blockIdx.x;
  const float4 *r4 = reinterpret_cast<const float4 *>(out + static_cast<size_t>(f) * rows);
  float4 *d4 = reinterpret_cast<float4 *>(dst + static_cast<size_t>(f) * rows);
  const int groups = rows >> 2;
  float4 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = tid + 256 * q;
    v[q] = i < groups ? r4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (tid < 64) {
    const int MT = n_partial >> 2;
    int L = 1;
    while (L < MT) L <<= 1;
    float sj = 0.0f;
    if (tid < MT) {
      const float *pp = partial + static_cast<size_t>(4 * tid) * partial_ld + f;
      sj = (pp[0] + pp[partial_ld]) + (pp[2 * static_cast<size_t>(partial_ld)] + pp[3 * static_cast<size_t>(partial_ld)]);
    }
    for (int len = L >> 1; len >= 1; len >>= 1) {  // level by level, adjacent pairs: lane t < len takes values 2t, 2t + 1 of the level before
      const float a = __shfl(sj, (2 * tid) & 63), b = __shfl(sj, (2 * tid + 1) & 63);
      sj = a + b;
    }
    if (tid == 0) tot_s = sj;
  }
  __syncthreads();
  const float inv = 1.0f / tot_s;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = tid + 256 * q;
    if (i < groups) d4[i] = make_float4(v[q].x * inv, v[q].y * inv, v[q].z * inv, v[q].w * inv);
  }
}

// The same pass as a BACKGROUND kernel for the server loop: a fixed, small grid of workgroups that
// walk the rows, so that it holds one or two wave slots per SIMD for its whole life instead of
// flooding every CU with thousands of short workgroups.  Launched on the low-priority tail stream
// under the next batch's layer 0, whose 228-register waves need the rest of the register file: with
// the one-workgroup-per-row grid the short workgroups grabbed every slot that came free and layer 0
// starved until the scale pass was done (rocprofv3 timeline: 120 us overlap, layer 0 330 -> 430 us).
__global__ __launch_bounds__(256) void normalize_bg_kernel(const float *out, float *dst, const float *partial, int n,
                                                           int partial_ld, int rows, int n_partial) {
  __shared__ float red[kNormMaxTiles];
  for (int f = blockIdx.x; f < n; f += gridDim.x) {
    normalize_row<(FDNN_NORM_BG_NT != 0)>(out, dst, partial, f, partial_ld, rows, n_partial, red);
    __syncthreads();  // red[] is reused by the next row
  }
}

// ---------------------------------------------------------------- load-time check of the fast division
__global__ __launch_bounds__(256) void fastdiv_check_kernel(float coef, float rcp, unsigned long long *mismatch) {
  const long long lo = -(1ll << 26), hi = (1ll << 26);
  unsigned long long bad = 0;
  for (long long a = lo + blockIdx.x * 256ll + threadIdx.x; a <= hi; a += static_cast<long long>(gridDim.x) * 256ll) {
    const float fast = dequant<true>(static_cast<int>(a), coef, rcp);
    const float ieee = dequant<false>(static_cast<int>(a), coef, rcp);
    if (__float_as_uint(fast) != __float_as_uint(ieee)) ++bad;
  }
  if (bad) atomicAdd(mismatch, bad);
}

__global__ __launch_bounds__(256) void xor80_kernel(const int8_t *in, uint8_t *out, size_t count) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < count) out[i] = static_cast<uint8_t>(in[i]) ^ 0x80;
}

// ---------------------------------------------------------------- lazy masks: bytes -> bits
// One 64-bit word per 64 nodes.  Lane l of a quad takes 16 mask bytes (one dwordx4: a wave reads 1 KiB contiguous),
// turns them into 16 bits, and the quad's first lane collects the four pieces.
__device__ __forceinline__ uint32_t nonzero_bytes_to_bits(uint32_t x) {
  const uint32_t m = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;  // bit 7 of every non-zero byte
  return (((m >> 7) * 0x01020408u) >> 24) & 0xfu;                            // ... gathered into bits 0..3
}
// Rows of whole words (rows % 64 == 0, 16-byte aligned: the 8000-node layer) make the masks ONE contiguous stream with
// piece i <-> bits of word i / 4: no row arithmetic (the general kernel below divides a 64-bit index per piece), eight
// 16-byte loads in flight per thread.  80 MB at 10 000 frames: 19.6 us = 4.1 TB/s general, see DESIGN.md for this one.
__global__ __launch_bounds__(256) void mask_pack_flat_kernel(const int8_t *mask, uint64_t *bits, long long pieces) {
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  constexpr int U = 8;
  for (long long i0 = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i0 < pieces; i0 += U * stride) {
    v4i t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      t[u] = v4i{0, 0, 0, 0};
      if (i < pieces) t[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(mask) + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      const uint32_t b = nonzero_bytes_to_bits(static_cast<uint32_t>(t[u].x)) | nonzero_bytes_to_bits(static_cast<uint32_t>(t[u].y)) << 4 |
                         nonzero_bytes_to_bits(static_cast<uint32_t>(t[u].z)) << 8 | nonzero_bytes_to_bits(static_cast<uint32_t>(t[u].w)) << 12;
      // (pieces and stride are multiples of 4: the four lanes of a quad are in range together)
      const uint32_t b1 = __shfl_down(b, 1), b2 = __shfl_down(b, 2), b3 = __shfl_down(b, 3);
      if (i < pieces && (i & 3) == 0) bits[i >> 2] = static_cast<uint64_t>(b | b1 << 16) | static_cast<uint64_t>(b2 | b3 << 16) << 32;
    }
  }
}

__global__ __launch_bounds__(256) void mask_pack_kernel(const int8_t *mask, uint64_t *bits, int n, int rows, int wpr) {
  const long long total = static_cast<long long>(n) * wpr * 4;  // 16-byte pieces, rows padded to whole words
  const long long stride = static_cast<long long>(gridDim.x) * 256;
  constexpr int U = 4;  // pieces in flight per thread: the pass is one long HBM read
  for (long long i0 = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i0 < total; i0 += U * stride) {
    uint4 v[U];
    int fr[U], pc[U];
    bool fast[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      v[u] = make_uint4(0, 0, 0, 0);
      fr[u] = pc[u] = 0;
      fast[u] = false;
      if (i < total) {
        fr[u] = static_cast<int>(i / (wpr * 4));
        pc[u] = static_cast<int>(i - static_cast<long long>(fr[u]) * (wpr * 4));
        const int8_t *src = mask + static_cast<size_t>(fr[u]) * rows + pc[u] * 16;
        fast[u] = pc[u] * 16 + 16 <= rows && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (fast[u]) {
          const v4i t = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src));
          v[u] = make_uint4(static_cast<uint32_t>(t.x), static_cast<uint32_t>(t.y), static_cast<uint32_t>(t.z), static_cast<uint32_t>(t.w));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      uint32_t b = 0;
      if (fast[u]) {
        b = nonzero_bytes_to_bits(v[u].x) | nonzero_bytes_to_bits(v[u].y) << 4 | nonzero_bytes_to_bits(v[u].z) << 8 |
            nonzero_bytes_to_bits(v[u].w) << 12;
      } else if (i < total) {
        const int col = pc[u] * 16;
        const int8_t *src = mask + static_cast<size_t>(fr[u]) * rows + col;
        for (int q = 0; q < 16; ++q)
          if (col + q < rows && src[q] != 0) b |= 1u << q;
      }
      // the quad's pieces -> one word (the four lanes of a quad are in range together: total and stride are multiples of 4)
      const uint32_t b1 = __shfl_down(b, 1), b2 = __shfl_down(b, 2), b3 = __shfl_down(b, 3);
      if (i < total && (pc[u] & 3) == 0)
        bits[static_cast<size_t>(fr[u]) * wpr + (pc[u] >> 2)] =
            static_cast<uint64_t>(b | b1 << 16) | static_cast<uint64_t>(b2 | b3 << 16) << 32;
    }
  }
}

// the other way (bit-mask entry points, small batches: the small-batch kernels read mask bytes): byte [f][c] = bit c & 63 of word [f][c >> 6]
__global__ __launch_bounds__(256) void mask_unpack_kernel(const uint64_t *bits, int8_t *mask, int n, int rows, int wpr) {
  const long long total = static_cast<long long>(n) * rows;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const int f = static_cast<int>(i / rows), c = static_cast<int>(i - static_cast<long long>(f) * rows);
    mask[i] = static_cast<int8_t>((bits[static_cast<size_t>(f) * wpr + (c >> 6)] >> (c & 63)) & 1ull);
  }
}

// Lazy results for a host caller, compacted: every inactive node of a row reads the same 1 / total (exp(0) terms,
// dnn.cc:366-369, :389), so only the active nodes' probabilities and that one value have to cross PCIe.
//   comp[f][0] = the row's inactive value (0 if the row has no inactive node), comp[f][1 + r] = probability of the row's r-th
//   active node, in node order.  One wave per frame: lane = node within the 64-node word, coalesced reads, dense writes.
__global__ __launch_bounds__(256) void lazy_compact_kernel(const float *out, const uint64_t *bits, float *comp, int n, int rows, int wpr, int stride) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= n) return;
  const float *row = out + static_cast<size_t>(f) * rows;
  float *dst = comp + static_cast<size_t>(f) * stride;
  int at = 0;
  float inactive = 0.0f;
  bool have = false;
  for (int w = 0; w < wpr; ++w) {
    const uint64_t word = bits[static_cast<size_t>(f) * wpr + w];  // (wave-uniform)
    const int node = 64 * w + lane;
    const bool in = node < rows, on = in && ((word >> lane) & 1ull);
    const float v = in ? row[node] : 0.0f;
    if (on) dst[1 + at + __popcll(word & ((1ull << lane) - 1ull))] = v;
    const uint64_t valid = rows - 64 * w >= 64 ? ~0ull : ((1ull << (rows - 64 * w)) - 1ull);
    at += __popcll(word & valid);
    if (!have) {
      const uint64_t off = ~word & valid;
      if (off) {
        inactive = __shfl(v, __ffsll(static_cast<long long>(off)) - 1);
        have = true;
      }
    }
  }
  if (lane == 0) dst[0] = inactive;
}

}  // namespace

void launch_lazy_compact(const float *out, const uint64_t *bits, float *comp, int n, int rows, int stride, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(lazy_compact_kernel, dim3((n + 3) / 4), dim3(256), 0, s, out, bits, comp, n, rows, (rows + 63) / 64, stride);
}

void launch_mask_unpack(const uint64_t *bits, int8_t *mask, int n, int rows, hipStream_t s) {
  const long long total = static_cast<long long>(n) * rows;
  const int blocks = static_cast<int>(std::min<long long>((total + 255) / 256, 256 * 16));
  if (blocks > 0) hipLaunchKernelGGL(mask_unpack_kernel, dim3(blocks), dim3(256), 0, s, bits, mask, n, rows, (rows + 63) / 64);
}

void launch_mask_pack(const int8_t *mask, uint64_t *bits, int n, int rows, hipStream_t s) {
  const int wpr = (rows + 63) / 64;
  const long long total = static_cast<long long>(n) * wpr * 4;
  const int blocks = static_cast<int>(std::min<long long>((total + 255) / 256, 256 * 16));
  if (blocks <= 0) return;
  if (rows % 64 == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0)
    hipLaunchKernelGGL(mask_pack_flat_kernel, dim3(std::min(blocks, 256 * 8)), dim3(256), 0, s, mask, bits, total);
  else
    hipLaunchKernelGGL(mask_pack_kernel, dim3(blocks), dim3(256), 0, s, mask, bits, n, rows, wpr);
}

void launch_normalize(float *out, float *dst, const float *partial, int n, int partial_ld, int rows, int n_partial, hipStream_t s,
                      bool background) {
  if (n <= 0) return;
  if (background) {
    static const int wgs = [] {
      const char *e = FDNN_TUNE_ENV("FDNN_NORM_BG_WGS");
      return e ? std::max(1, std::atoi(e)) : 256;  // sweep, 2 steps in flight: 192-256 best (+7-8 % over one stream), 512 +4 %, 1024 +2 %
    }();
    hipLaunchKernelGGL(normalize_bg_kernel, dim3(std::min(n, wgs)), dim3(256), 0, s, out, dst, partial, n, partial_ld, rows, n_partial);
    return;
  }
  const bool small_ok = n <= 1024 && (rows & 3) == 0 && rows <= 8192 && (n_partial >> 2) <= 64 &&
                        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (small_ok) {
    hipLaunchKernelGGL(normalize_small_kernel, dim3(n), dim3(256), 0, s, out, dst, partial, n, partial_ld, rows, n_partial);
    return;
  }
  hipLaunchKernelGGL(normalize_kernel, dim3(n), dim3(256), 0, s, out, dst, partial, n, partial_ld, rows, n_partial);
}

void launch_fastdiv_check(float coef, float rcp, unsigned long long *d_mismatch, hipStream_t s) {
  hipLaunchKernelGGL(fastdiv_check_kernel, dim3(2048), dim3(256), 0, s, coef, rcp, d_mismatch);
}

void launch_xor80(const int8_t *in, uint8_t *out, size_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(xor80_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s, in, out, count);
}

}  // namespace fdnn
