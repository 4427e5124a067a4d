// fdnn_device.hpp -- device helpers shared by the gfx950 kernels.  Everything here
// is compiled with -ffp-contract=off: each float operation rounds exactly where
// the reference's scalar/SSE code rounds; fused multiply-adds appear only where
// written as fmaf().
#pragma once
#include <hip/hip_runtime.h>

#include <climits>

#include "fdnn_model.hpp"

namespace fdnn {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define FDNN_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define FDNN_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// 16 bytes per lane straight from global memory into LDS at
// wave-uniform base + lane*16 (global_load_lds_dwordx4).
__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds(FDNN_GLOBAL_PTR(g), FDNN_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// Write-through stores (sc0 sc1) for kernel results: the line goes to memory as it is written
// instead of in the write-back every kernel ends with (the eight L2s are not coherent with each
// other, so a kernel boundary flushes them), which shortens the tail of each launch -- measured
// on the hidden layers: 48.5 -> 47.0 us.  FDNN_WT selects which stores use them (experiments).
#ifndef FDNN_WT
#define FDNN_WT 35  // 1 hidden-layer activations, 2 output-layer exp(z), 32 soft-max scale; 4 / 8 / 16 = layer-0 park, activations, image: no effect measured
#endif
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
// The s_nop is part of the store: a vector-memory store of more than 8 bytes reads its data registers a cycle after
// issue, and a vector-ALU write to them in the very next slot corrupts what is stored.  For a store it emits itself the
// compiler's hazard recognizer inserts that wait state; inside inline asm it cannot see the store, and it is free to
// reuse the registers at once -- found when a rescheduled output epilogue wrote a few wrong exp(z) values per launch,
// different ones from run to run (the rows' totals were right, their sums after the scale pass were not).
__device__ __forceinline__ void store_wt(void *p, v4i v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(void *p, v4f_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(void *p, v2f_t v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(void *p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// float(sum) / (multiplier * 255)   -- dnn.cc:298-299, :311.  The fast form is
// the Markstein sequence q = x*y, r = fma(-q, c, x), q' = fma(r, y, q) with
// y = RN(1/c); it is enabled per layer only after launch_fastdiv_check has
// compared it with IEEE division for every possible accumulator.
template <bool FAST>
__device__ __forceinline__ float dequant(int acc, float coef, float rcp) {
  const float x = static_cast<float>(acc);  // v_cvt_f32_i32, RNE like cvtsi2ss
  if (FAST) {
    const float q = x * rcp;
    const float r = fmaf(-q, coef, x);
    return fmaf(r, rcp, q);
  }
  return x / coef;
}

// QuantizedSigmoid::get -- dnn.h:36-43: k = (int)round(x*100), table index
// clamp(k,-640,640)+640 into the extended table.  round() is half away from
// zero; the x86 build turns NaN / |t| >= 2^31 into INT_MIN (-> entry 0).
__device__ __forceinline__ int lut_index(float lin) {
  const float t = lin * 100.0f;
  const float r0 = truncf(t);
  // t - r0 is exact; select, do not branch (the epilogue runs this 32*NF times per lane)
  const float r = r0 + ((fabsf(t - r0) >= 0.5f) ? copysignf(1.0f, t) : 0.0f);
  int k = (fabsf(t) < 2147483648.0f) ? static_cast<int>(r) : INT_MIN;
  k = max(-kLutHalf, min(kLutHalf, k));
  return k + kLutHalf;
}

}  // namespace fdnn
