// fdnn_ppo.hip -- the OUTPUT layer of a large dense batch with the two waves of every SIMD in different roles, soft-max
// scaled inside the kernel (gfx950).
//
//   CalculateOutput (dnn.cc:428-454): QuantizedLayerActivations / quantizedNodeSum (dnn.cc:289-349) + bias + SoftMax::apply
//   (dnn.cc:534-544)
//
// fdnn_pp.hip's structure (one group of four waves in the k-loop of its 256-node x 160-frame half -- fragment reads and
// MFMAs only -- while the other group stages that half's operands and runs the epilogue of the half it computed before),
// with the fused soft-max of fdnn_gemm.hip's FUSED instances as the epilogue: e = exp(z) replaces the accumulators in place,
// the 64-node partial sums P of a frame are formed exactly as there, the workgroup's S = (P0 + P1) + (P2 + P3) is published,
// the 32 node tiles of a half exchange their S through memory, every workgroup finishes the same tree per frame and scales
// its e by RN(1 / total) on the way out: identical bits (tests/test_gpu_ppo.py).
//
// The exchange is run from the COMPUTE role: its vector-memory queue is empty (no loads, no counted waits), so a store, an
// agent-scope add, a poll and the 20 KB gather of the 32 x 160 sums -- LDS-DMA, straight into LDS -- can each be issued after
// one tick's barrier and looked at after the next, under the MFMAs.  The support waves (whose queue the tick's counted wait
// drains in order) only write their partial sums to LDS, and later scale.  Timeline of a phase: exp in ticks 0..9 (four
// items of four outputs per lane and tick), partial sums parked at tick 10, published after barrier 10, arrival counted
// after 11, polled 12 -> 13, gathered 14 -> 15; then the phase is EXTENDED by barrier-only ticks in which the compute group
// forms the totals and the support group scales and stores its 160 outputs per lane (plain 16-byte stores: the rows are read
// by nobody before the kernel ends) -- two such ticks when the siblings are on time, more while they are not (bounded).
//
// Placement: workgroup b owns node tile b % 32 for the whole launch (XCD b % 8 keeps four weight tiles in its L2) and the
// frame pairs b / 32, b / 32 + 8, ...: the 32 workgroups of a pair are consecutive block ids and run the same static
// schedule.  All 256 workgroups must be resident together (one per CU): the launch goes through the device's chain of
// fused launches like fdnn_gemm.hip's (fdnn_runtime.cpp: FuseChain), and every wait is bounded.
#include <atomic>
#include <climits>
#include <cstdlib>

#include "fdnn_device.hpp"
#include "fdnn_kernels.hpp"
#include "fdnn_tile.hpp"

#ifndef FDNN_PPO_CLK
#define FDNN_PPO_CLK 0  // 1 (measurement builds): printf of per-phase clocks from a few workgroups
#endif
#ifndef FDNN_PPO_DEBUG
#define FDNN_PPO_DEBUG 0  // timing experiments: 1 no epilogue arithmetic, 2 no MFMAs, 4 no operand loads, 8 no stores, 16 no exchange and no extension, 32 no scale
#endif

namespace fdnn {
namespace {

[[maybe_unused]] constexpr int kNF = 5, kBK = 128, kBM = 256, kHT = 32 * kNF, kFT = 2 * kHT, kKT = 16, kTS = kBM + 16;
[[maybe_unused]] constexpr int kWStages = 3;
[[maybe_unused]] constexpr int kMT = 32;       // node tiles of the layer (rows_pad = 8192): the 32 x 160 sums of a half are one 20 KB block
#ifndef FDNN_PPO_EXP_PER
#define FDNN_PPO_EXP_PER 6  // epilogue items (of four outputs per lane) per tick (4: 225 us, 5: 255, 6 .. 8: 213 .. 219, 10: 247 -- pair-free net, 10 000 frames)
#endif
[[maybe_unused]] constexpr int kExpPer = FDNN_PPO_EXP_PER;
[[maybe_unused]] constexpr int kPub = (8 * kNF + kExpPer - 1) / kExpPer;  // tick after whose barrier a half's sums are published: the first without exp items
[[maybe_unused]] constexpr int kExtBound = 1 << 16;  // extension ticks (two barriers and a few LDS reads each: ~500 cycles) before a workgroup stops waiting for its siblings: ~15 ms -- a legitimate wait is microseconds

// ---- the accumulators: in the ACCUMULATION registers a0 .. a159, behind the compiler's back.  A wave's ten 32 x 32 tiles
// (tile t = ni * 2 + mi) live in a[16 t : 16 t + 15] for the whole kernel, and every access is inline assembly on literal
// register numbers: the MFMAs, the start values (ds_read_b128 straight into the tiles), the rare pair corrections, exp in
// place, the scale on the way out.  The compiler allocates the architectural registers: amdgpu_num_vgpr(96) confines it
// to v0 .. v95 (a kernel with accumulation registers has its request halved: 192 / 2), the clobber lists make the kernel
// own a0 .. a159, and 96 + 160 is the 256 registers a wave of a 512-thread workgroup can have.
// Why: the epilogue turns int32 sums into float e IN PLACE (there is no room for a second copy) in one role while the other
// role's code wants the same variable as MFMA accumulators.  Left to the compiler -- element writes, whole-tile launders,
// tiles pinned with "+{v[..]}" / "+{a[..]}" operands on every access -- every tile of e was built in a fresh tuple, tiles moved
// between the roles through scratch, 200 .. 1 150 dwords a lane spilled in every variant tried.  Hidden ARCHITECTURAL
// registers (v96 ..) cannot be protected: with 160 KB of LDS the register cap cannot go below 169, and the compiler's
// temporaries landed in them.
// The compiler itself touches accumulation registers for one reason only: to park an architectural register when it runs
// out of those.  So the kernel must compile WITHOUT such spills, and every build proves it: csrc/Makefile runs
// tools/check_hidden_regs.py over the kernel's assembly, which fails the build if an instruction outside the inline-assembly
// blocks names an accumulation register or the split is not 96 + 160.  (profiles/LABBOOK.md, round 6)
// The price: VALU reaches these registers through v_accvgpr_read / _write only (three more instructions per output).
// The hazards the compiler no longer sees are kept apart by construction: an MFMA's result is read by VALU a phase later;
// the pair correction (rare) waits out the XDL write latency itself.
// X(item, byte offset of the item's four start values in the wave's 64, its four registers): item = tile * 4 + g
#define PPO_ITEMS(X) \
  X(0, 0, 0, 1, 2, 3) \
  X(1, 32, 4, 5, 6, 7) \
  X(2, 64, 8, 9, 10, 11) \
  X(3, 96, 12, 13, 14, 15) \
  X(4, 128, 16, 17, 18, 19) \
  X(5, 160, 20, 21, 22, 23) \
  X(6, 192, 24, 25, 26, 27) \
  X(7, 224, 28, 29, 30, 31) \
  X(8, 0, 32, 33, 34, 35) \
  X(9, 32, 36, 37, 38, 39) \
  X(10, 64, 40, 41, 42, 43) \
  X(11, 96, 44, 45, 46, 47) \
  X(12, 128, 48, 49, 50, 51) \
  X(13, 160, 52, 53, 54, 55) \
  X(14, 192, 56, 57, 58, 59) \
  X(15, 224, 60, 61, 62, 63) \
  X(16, 0, 64, 65, 66, 67) \
  X(17, 32, 68, 69, 70, 71) \
  X(18, 64, 72, 73, 74, 75) \
  X(19, 96, 76, 77, 78, 79) \
  X(20, 128, 80, 81, 82, 83) \
  X(21, 160, 84, 85, 86, 87) \
  X(22, 192, 88, 89, 90, 91) \
  X(23, 224, 92, 93, 94, 95) \
  X(24, 0, 96, 97, 98, 99) \
  X(25, 32, 100, 101, 102, 103) \
  X(26, 64, 104, 105, 106, 107) \
  X(27, 96, 108, 109, 110, 111) \
  X(28, 128, 112, 113, 114, 115) \
  X(29, 160, 116, 117, 118, 119) \
  X(30, 192, 120, 121, 122, 123) \
  X(31, 224, 124, 125, 126, 127) \
  X(32, 0, 128, 129, 130, 131) \
  X(33, 32, 132, 133, 134, 135) \
  X(34, 64, 136, 137, 138, 139) \
  X(35, 96, 140, 141, 142, 143) \
  X(36, 128, 144, 145, 146, 147) \
  X(37, 160, 148, 149, 150, 151) \
  X(38, 192, 152, 153, 154, 155) \
  X(39, 224, 156, 157, 158, 159)
// the registers of tile t, as an inline-assembly clobber list
#define PPO_CLOB_0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define PPO_CLOB_1 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define PPO_CLOB_2 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define PPO_CLOB_3 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define PPO_CLOB_4 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79"
#define PPO_CLOB_5 "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define PPO_CLOB_6 "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111"
#define PPO_CLOB_7 "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define PPO_CLOB_8 "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143"
#define PPO_CLOB_9 "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
// X(tile, first, last register)
#define PPO_TILES(X) \
  X(0, 0, 15) \
  X(1, 16, 31) \
  X(2, 32, 47) \
  X(3, 48, 63) \
  X(4, 64, 79) \
  X(5, 80, 95) \
  X(6, 96, 111) \
  X(7, 112, 127) \
  X(8, 128, 143) \
  X(9, 144, 159)
// X(element of the 64-node x 32-frame pair of tiles mi = i >> 4: its register in frame blocks ni = 0 .. 4)
#define PPO_ELEMS(X) \
  X(0, 0, 32, 64, 96, 128) \
  X(1, 1, 33, 65, 97, 129) \
  X(2, 2, 34, 66, 98, 130) \
  X(3, 3, 35, 67, 99, 131) \
  X(4, 4, 36, 68, 100, 132) \
  X(5, 5, 37, 69, 101, 133) \
  X(6, 6, 38, 70, 102, 134) \
  X(7, 7, 39, 71, 103, 135) \
  X(8, 8, 40, 72, 104, 136) \
  X(9, 9, 41, 73, 105, 137) \
  X(10, 10, 42, 74, 106, 138) \
  X(11, 11, 43, 75, 107, 139) \
  X(12, 12, 44, 76, 108, 140) \
  X(13, 13, 45, 77, 109, 141) \
  X(14, 14, 46, 78, 110, 142) \
  X(15, 15, 47, 79, 111, 143) \
  X(16, 16, 48, 80, 112, 144) \
  X(17, 17, 49, 81, 113, 145) \
  X(18, 18, 50, 82, 114, 146) \
  X(19, 19, 51, 83, 115, 147) \
  X(20, 20, 52, 84, 116, 148) \
  X(21, 21, 53, 85, 117, 149) \
  X(22, 22, 54, 86, 118, 150) \
  X(23, 23, 55, 87, 119, 151) \
  X(24, 24, 56, 88, 120, 152) \
  X(25, 25, 57, 89, 121, 153) \
  X(26, 26, 58, 90, 122, 154) \
  X(27, 27, 59, 91, 123, 155) \
  X(28, 28, 60, 92, 124, 156) \
  X(29, 29, 61, 93, 125, 157) \
  X(30, 30, 62, 94, 126, 158) \
  X(31, 31, 63, 95, 127, 159)

[[maybe_unused]] __device__ __forceinline__ void ppo_mfma(int t, v4i a, v4i b) {
  switch (t) {
#define X(T, LO, HI) \
  case T: asm volatile("v_mfma_i32_32x32x32_i8 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(a), "v"(b) : PPO_CLOB_##T); break;
    PPO_TILES(X)
#undef X
    default: break;
  }
}
// item it's four start values: Sum_k 128 w[k] of its four nodes, for every frame alike
[[maybe_unused]] __device__ __forceinline__ void ppo_init4(int it, uint32_t wave_base) {
  switch (it) {
#define X(IT, OFF, R0, R1, R2, R3) \
  case IT: asm volatile("ds_read_b128 a[" #R0 ":" #R3 "], %0 offset:" #OFF ::"v"(wave_base) : "memory", "a" #R0, "a" #R1, "a" #R2, "a" #R3); break;
    PPO_ITEMS(X)
#undef X
    default: break;
  }
}
// element i of the five frame blocks += c[.] (the deferred difference of a saturating pair: fdnn_gemm.hip's walk)
[[maybe_unused]] __device__ __forceinline__ void ppo_add(int i, const int (&c)[5]) {
  int t0, t1, t2, t3, t4;
  switch (i) {
#define X(I, RA, RB, RC, RD, RE)                                                                                                  \
  case I:                                                                                                                          \
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                                                                \
                 "v_accvgpr_read_b32 %0, a" #RA "\n\tv_accvgpr_read_b32 %1, a" #RB "\n\tv_accvgpr_read_b32 %2, a" #RC "\n\t"          \
                 "v_accvgpr_read_b32 %3, a" #RD "\n\tv_accvgpr_read_b32 %4, a" #RE "\n\ts_nop 1\n\t"                                  \
                 "v_add_u32 %0, %0, %5\n\tv_add_u32 %1, %1, %6\n\tv_add_u32 %2, %2, %7\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %9\n\ts_nop 1\n\t" \
                 "v_accvgpr_write_b32 a" #RA ", %0\n\tv_accvgpr_write_b32 a" #RB ", %1\n\tv_accvgpr_write_b32 a" #RC ", %2\n\t"       \
                 "v_accvgpr_write_b32 a" #RD ", %3\n\tv_accvgpr_write_b32 a" #RE ", %4\n\ts_nop 7"                                    \
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)                                                             \
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4])                                                            \
                 : "a" #RA, "a" #RB, "a" #RC, "a" #RD, "a" #RE);                                                                    \
    break;
    PPO_ELEMS(X)
#undef X
    default: break;
  }
}
// Four elements (one lane's four consecutive nodes of one frame): SoftMax::apply's first loop in fdnn_gemm.hip's FUSED
// operations, one for one: x = float(acc); q0 = x rcp; r = fma(-q0, coef, x); z = fma(r, rcp, q0) + bias; e = exp2(z log2e) keep;
// the lane's running sum takes them in order.  (Four chains side by side: a transcendental's result is not read by the next instruction.)
#define PPO_EXP4(R0, R1, R2, R3)                                                                                                     \
  asm volatile("v_accvgpr_read_b32 %[a0], a" #R0 "\n\tv_accvgpr_read_b32 %[a1], a" #R1 "\n\tv_accvgpr_read_b32 %[a2], a" #R2 "\n\tv_accvgpr_read_b32 %[a3], a" #R3 "\n\t" \
               "v_cvt_f32_i32 %[a0], %[a0]\n\tv_cvt_f32_i32 %[a1], %[a1]\n\tv_cvt_f32_i32 %[a2], %[a2]\n\tv_cvt_f32_i32 %[a3], %[a3]\n\t" \
               "v_mul_f32 %[q0], %[rcp], %[a0]\n\tv_mul_f32 %[q1], %[rcp], %[a1]\n\tv_mul_f32 %[q2], %[rcp], %[a2]\n\tv_mul_f32 %[q3], %[rcp], %[a3]\n\t" \
               "v_fma_f32 %[a0], -%[q0], %[coef], %[a0]\n\tv_fma_f32 %[a1], -%[q1], %[coef], %[a1]\n\t"                                \
               "v_fma_f32 %[a2], -%[q2], %[coef], %[a2]\n\tv_fma_f32 %[a3], -%[q3], %[coef], %[a3]\n\t"                                \
               "v_fmac_f32 %[q0], %[rcp], %[a0]\n\tv_fmac_f32 %[q1], %[rcp], %[a1]\n\tv_fmac_f32 %[q2], %[rcp], %[a2]\n\tv_fmac_f32 %[q3], %[rcp], %[a3]\n\t" \
               "v_add_f32 %[q0], %[q0], %[b0]\n\tv_add_f32 %[q1], %[q1], %[b1]\n\tv_add_f32 %[q2], %[q2], %[b2]\n\tv_add_f32 %[q3], %[q3], %[b3]\n\t" \
               "v_mul_f32 %[q0], 0x3fb8aa3b, %[q0]\n\tv_mul_f32 %[q1], 0x3fb8aa3b, %[q1]\n\t"                                          \
               "v_mul_f32 %[q2], 0x3fb8aa3b, %[q2]\n\tv_mul_f32 %[q3], 0x3fb8aa3b, %[q3]\n\t"                                          \
               "v_exp_f32 %[q0], %[q0]\n\tv_exp_f32 %[q1], %[q1]\n\tv_exp_f32 %[q2], %[q2]\n\tv_exp_f32 %[q3], %[q3]\n\t"              \
               "v_mul_f32 %[q0], %[q0], %[keep]\n\tv_mul_f32 %[q1], %[q1], %[keep]\n\t"                                                 \
               "v_mul_f32 %[q2], %[q2], %[keep]\n\tv_mul_f32 %[q3], %[q3], %[keep]\n\t"                                                 \
               "v_accvgpr_write_b32 a" #R0 ", %[q0]\n\tv_accvgpr_write_b32 a" #R1 ", %[q1]\n\t"                                         \
               "v_accvgpr_write_b32 a" #R2 ", %[q2]\n\tv_accvgpr_write_b32 a" #R3 ", %[q3]\n\t"                                         \
               "v_add_f32 %[ps], %[ps], %[q0]\n\tv_add_f32 %[ps], %[ps], %[q1]\n\t"                                                     \
               "v_add_f32 %[ps], %[ps], %[q2]\n\tv_add_f32 %[ps], %[ps], %[q3]"                                                         \
               : [ps] "+v"(psum), [a0] "=&v"(xa0), [a1] "=&v"(xa1), [a2] "=&v"(xa2), [a3] "=&v"(xa3), [q0] "=&v"(xq0), [q1] "=&v"(xq1),       \
                 [q2] "=&v"(xq2), [q3] "=&v"(xq3)                                                                                       \
               : [rcp] "s"(rcp_s), [coef] "s"(coef_s), [b0] "v"(b4.x), [b1] "v"(b4.y), [b2] "v"(b4.z), [b3] "v"(b4.w), [keep] "v"(keep)          \
               : "a" #R0, "a" #R1, "a" #R2, "a" #R3)
// p = e / total as e * RN(1 / total): four elements on their way out
#define PPO_SCL4(R0, R1, R2, R3)                                                                                                     \
  asm volatile("v_accvgpr_read_b32 %0, a" #R0 "\n\tv_accvgpr_read_b32 %1, a" #R1 "\n\tv_accvgpr_read_b32 %2, a" #R2 "\n\tv_accvgpr_read_b32 %3, a" #R3 "\n\t" \
               "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"                            \
               : "=&v"(o4.x), "=&v"(o4.y), "=&v"(o4.z), "=&v"(o4.w)                                                                      \
               : "v"(iv))

template <bool NOFIX>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(96))) void qppo_kernel(QGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Separate static arrays (fdnn_pp.hip): the compiler must know which LDS traffic can alias an LDS-DMA destination.
  __shared__ __attribute__((aligned(16))) char ringW[kWStages][kBM * kBK];
  __shared__ __attribute__((aligned(16))) char ringA[2][kHT * kBK];
  // the 32 x 160 row sums of a half (LDS-DMA target of the gather); its first 4 x 160 floats double as the four waves' partial
  // sums P of the half being exchanged (written with inline asm by the support waves: a plain store to an LDS-DMA target
  // would make the compiler wait for every load in flight)
  __shared__ __attribute__((aligned(16))) float sg_s[kMT * kHT];
  __shared__ __attribute__((aligned(16))) float inv_s[kHT];        // RN(1 / total) per frame of the half
  // the workgroup's node tile never changes: its 256 accumulator start values and biases are fetched once
  __shared__ __attribute__((aligned(16))) int wsum_s[kBM];
  __shared__ __attribute__((aligned(16))) float bias_s[kBM];
  // exchange counters, MONOTONIC over the phases (phase ph's targets are multiples of ph): nothing is ever reset, so a wave that
  // reads late can only see more.  [0] generation whose sums have all arrived, [1] compute waves whose S stores are out,
  // [2] ... whose gather pieces have landed, [3] ... that have written their frames' 1 / total, [4] support waves that have scaled
  __shared__ int xf_s[8];
  __shared__ __attribute__((aligned(16))) uint32_t xpoll_s[4];  // the polled arrival count lands here (LDS-DMA: no register waits for it)

#if FDNN_PPO_CLK
  __shared__ long long clk_s[48];  // per phase: start, end of its 16 ticks, end of its extension (wave 0's clock)
  __shared__ int clk_ext[16];
#define PPO_CLK(i)                                                                      \
  do {                                                                                  \
    if (wave == 0 && ln == 0 && (i) < 48) clk_s[i] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define PPO_CLK(i) \
  do {             \
  } while (0)
#endif
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, grp = wave >> 2;
  typedef const __attribute__((address_space(4))) QGemmParams *KP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  (void)p;
#define p (*kp)

  // ---- the workgroup's tiles: node tile b % 32 for the whole launch, frame pairs b / 32, + grid / 32, ...
  const int NP = p.n_pad / kFT;
  const int my_mt = static_cast<int>(blockIdx.x) % kMT;
  const int pair0 = static_cast<int>(blockIdx.x) / kMT, pair_step = static_cast<int>(gridDim.x) / kMT;
  const int n_tiles = pair0 < NP ? (NP - pair0 + pair_step - 1) / pair_step : 0;
  if (n_tiles == 0) return;
  auto pair_of = [&](int i) { return pair0 + i * pair_step; };
  if (tid < 8) xf_s[tid] = 0;
  if (tid < kBM) {
    wsum_s[tid] = p.wsum[my_mt * kBM + tid];
    bias_s[tid] = p.bias[my_mt * kBM + tid];
  }
  __syncthreads();

  // phase ph (0 <= ph < 2 n_tiles): group ph & 1 computes half ph & 1 of tile ph >> 1, the other group runs the epilogue of the
  // half of phase ph - 1; phase -1: group 0 only prepares tile 0's half 0; phase 2 n_tiles: group 1's last epilogue, group 0 in
  // the compute role for the exchange only.  Every phase with an epilogue ends with its extension ticks.
  const int n_ph = 2 * n_tiles;
  for (int ph = -1; ph <= n_ph; ++ph) {
    const int cg = ph & 1, sg = cg ^ 1;
    const bool cvalid = ph >= 0 && ph < n_ph;
    const bool evalid = ph >= 1;
    const bool nvalid = ph + 1 < n_ph;
    const int kt0 = (cvalid || evalid) ? 0 : kKT - 2;
    const int gt0 = (ph + 1) * kKT;
    const int c_pair = cvalid ? pair_of(ph >> 1) : 0, e_pair = evalid ? pair_of((ph - 1) >> 1) : 0, n_pair = nvalid ? pair_of((ph + 1) >> 1) : 0;
    const int e_half = 2 * e_pair + sg;  // the half whose epilogue runs in this phase (the support group's)
    // the lane id, from the hardware every phase: carried across the phases it would be spilled (the compute role fills the
    // 96 registers) and fetched back from scratch in the support ticks, behind that wave's queue of operand loads
    int zero_s = 0;
    asm volatile("" : "+s"(zero_s));
    int ln = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, static_cast<uint32_t>(zero_s))));
    int ext_iters = 0;
    // FDNN_GEMM_DEBUG=4096 (tests): every third node tile pretends, in its first epilogue, that its wait for the siblings timed out
    const bool sabotage = (p.debug & 4096) && ph == 1 && my_mt % 3 == 1;
    PPO_CLK(3 * (ph + 1));
    if (grp == cg) {
      // ============================================================== COMPUTE role
      const int frow = ln & 31, fch = ln >> 5;
      int fix_e = 0, fix_end = 0, fix_k_next = INT_MAX;
      typedef const __attribute__((address_space(4))) uint64_t *FixPtr;
      const FixPtr ent_c = (FixPtr)(uintptr_t)p.fix_ent;
      uint64_t fix_raw = 0, fix_raw_nxt = 0;
      const int fix_node0 = my_mt * kBM + 64 * wm;
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): my share of the stages I requested in my last support ticks has landed (builtin: fdnn_pp.hip)
      asm volatile("" ::: "memory");
      if (cvalid) {
        // the start values: every frame block's tiles alike (40 16-byte reads into the accumulators)
        const uint32_t wa = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(FDNN_LDS_PTR(wsum_s))) + (64 * wm + 4 * (ln >> 5)) * 4;
#pragma unroll
        for (int it = 0; it < 8 * kNF; ++it) ppo_init4(it, wa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!NOFIX && p.fix_ent) {
          const int gi = (my_mt * kBM >> 6) + wm;
          fix_e = __builtin_amdgcn_readfirstlane(p.fix_grp[gi]);
          fix_end = __builtin_amdgcn_readfirstlane(p.fix_grp[gi + 1]);
          if (fix_e < fix_end) {
            fix_raw = ent_c[fix_e];
            if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
            fix_k_next = static_cast<int>(fix_raw & 0xffff);
          }
        }
      }

      // ---- the exchange of the partner's half (e_half), one step per slot (a slot = the instructions behind a tick's barrier,
      // or an extension tick): 0 publish S | 1 stores out, arrival | 2 poll | 3 all arrived? | 4 gather | 5 landed | 6 totals
      int xs = 0;
      float *gs_half = p.fuse_s + static_cast<size_t>(e_half) * kMT * kHT;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.fuse_cnt + 4 * e_half, 0, 16, 0x00020000);
      uint32_t *cnt = p.fuse_cnt + 4 * e_half;  // [0] node tiles whose sums are out, [1] node tiles that have gathered; zero between launches (four words a half: the context's array is sized for eight per 128 frames)
      auto xslot = [&]() {
#if FDNN_PPO_DEBUG & 16
        return;
#endif
        if (xs == 0) {
          const int t = 64 * wm + ln;
          if (t < kHT) {
            const float s_ = (sg_s[t] + sg_s[kHT + t]) + (sg_s[2 * kHT + t] + sg_s[3 * kHT + t]);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(gs_half + static_cast<size_t>(my_mt) * kHT, 0, kHT * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s_), r, t * 4, 0, 17);  // write-through: read past the other XCDs' L2s
          }
          xs = 1;
        } else if (xs == 1) {
          __builtin_amdgcn_s_waitcnt(0x0f70);  // my S stores have been acknowledged
          asm volatile("" ::: "memory");
          if (ln == 0) {
            const int old = __hip_atomic_fetch_add(&xf_s[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 4 * ph - 1) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (no return value used: no wait)
          }
          xs = 2;
        } else if (xs == 2) {
          // the poll is a load INTO LDS, past the L2s: between this slot and the next nothing but the memory system holds
          // it (an inline-assembly load into a register would be a value the compiler may move before it has arrived)
          if (wm == 0 && ln == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, FDNN_LDS_PTR(xpoll_s), 4, 0, 0, 0, 17);
          xs = 3;
        } else if (xs == 3) {
          if (wm == 0) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            asm volatile("" ::: "memory");
            uint32_t seen;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(FDNN_LDS_PTR(xpoll_s)))) : "memory");
            if (ln == 0) {
              if (seen >= static_cast<uint32_t>(kMT)) xf_s[0] = ph;
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, FDNN_LDS_PTR(xpoll_s), 4, 0, 0, 0, 17);
            }
          }
          if (__builtin_amdgcn_readfirstlane(xf_s[0]) == ph) xs = 4;  // (wave 0 goes on at once; the others see the flag behind the next barrier)
        }
        if (xs == 4) {
          // 20 KB, contiguous in memory and in LDS: five one-KiB pieces per wave, past the L2s (sc0 sc1)
          const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(gs_half, 0, kMT * kHT * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(reinterpret_cast<char *>(sg_s) + (i * 4 + wm) * 1024), 16, ln * 16, (i * 4 + wm) * 1024, 0, 17);
          xs = 5;
        } else if (xs == 5) {
          __builtin_amdgcn_s_waitcnt(0x0f70);
          asm volatile("" ::: "memory");
          if (ln == 0) __hip_atomic_fetch_add(&xf_s[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          xs = 6;
        } else if (xs == 6 && __builtin_amdgcn_readfirstlane(xf_s[2]) >= 4 * ph) {
          const int t = 64 * wm + ln;
          if (t < kHT) {  // adjacent pairs, level by level (normalize_row's tree over the 32 node tiles), depth first
            float l3[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              float l2[2];
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const float *q = sg_s + (16 * a + 8 * b) * kHT + t;
                const float s01 = q[0] + q[kHT], s23 = q[2 * kHT] + q[3 * kHT], s45 = q[4 * kHT] + q[5 * kHT], s67 = q[6 * kHT] + q[7 * kHT];
                float o = (s01 + s23) + (s45 + s67);
                asm volatile("" : "+v"(o));
                l2[b] = o;
              }
              l3[a] = l2[0] + l2[1];
            }
            inv_s[t] = 1.0f / (l3[0] + l3[1]);
          }
          if (wm < 3 && ln == 0) __hip_atomic_fetch_add(&xf_s[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // one more node tile has gathered: the last one re-zeroes the half's counters.  (After the signal: the support waves
          // scale while this round trip is out.)
          if (wm == 0 && ln == 0 && __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<uint32_t>(kMT) - 1u) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          xs = 7;
        }
      };

      // fragment i of a sub-step's seven in the order the MFMAs want them: A0 B0 A1 B1 B2 B3 B4
      v4i fa[2][2], fb[2][kNF];
      auto load_frag = [&](int T, int kk, int set, int i) {
        const char *wt = ringW[T % kWStages], *at = ringA[T & 1];
        if (i == 0) fa[set][0] = read_frag<kBK>(wt, 64 * wm + frow, kk * 2 + fch);
        else if (i == 2) fa[set][1] = read_frag<kBK>(wt, 64 * wm + 32 + frow, kk * 2 + fch);
        else {
          const int ni = i == 1 ? 0 : i - 2;
          fb[set][ni] = read_frag<kBK>(at, 32 * ni + frow, kk * 2 + fch);
        }
      };
      // one sub-step: the ten MFMAs of fragment set `mset`, the seven reads of the next sub-step's set between them
      // (one MFMA issue slot each: measured best in tools/ubench_role.hip); the order is pinned
      auto substep = [&](bool do_mfma, bool do_read, int T, int kk, int rset, int mset) {
#pragma unroll
        for (int i = 0; i < 2 * kNF; ++i) {
#if !(FDNN_PPO_DEBUG & 2)
          if (do_mfma) ppo_mfma(i, fa[mset][i & 1], fb[mset][i >> 1]);
#endif
          __builtin_amdgcn_sched_barrier(0);
          if (do_read && i < 7) {
            load_frag(T, kk, rset, i);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };

      if (cvalid) {
#pragma unroll
        for (int i = 0; i < 7; ++i) load_frag(gt0, 0, 0, i);
      }
      for (int kt = kt0; kt < kKT; ++kt) {
        const int T = gt0 + kt;
        if (cvalid) {
          const char *at = ringA[T & 1];
          // The saturating-pair walk (fdnn_gemm.hip's): for every listed pair of this k-step a screen -- three 16-bit LDS reads,
          // a dot product, a ballot -- and, where a frame fires, the exact correction.  A screen is a latency chain that two
          // waves per SIMD hide from each other and a lone compute wave cannot: the FIRST due entry's reads are issued in
          // front of sub-step 0 and looked at behind its ten MFMAs (an int32 correction may come at any point of the phase).
          auto screen_read = [&](int j) {
            const int kl = static_cast<int>(fix_raw & 0xffff) - kt * kBK;
            const int row = 64 * j + ln;
            const int rr_ = row < kHT ? row : 0;
            return static_cast<int>(*reinterpret_cast<const uint16_t *>(at + rr_ * kBK + (((kl >> 4) ^ swz<kBK>(rr_)) << 4) + (kl & 15)));
          };
          auto walk_one = [&](int v0, int v1, int v2) {
            const int node = static_cast<int>(fix_raw >> 32) - fix_node0;
            const int kl = static_cast<int>(fix_raw & 0xffff) - kt * kBK;
            const int w0 = static_cast<int8_t>(fix_raw >> 16), w1 = static_cast<int8_t>(fix_raw >> 24);
            const int wpk = (w0 & 0xff) | ((w1 & 0xff) << 8);
            const int pbase = 128 * (w0 + w1) + 32768;
            const int vs[3] = {v0, v1, v2};
            bool fire = false;
#pragma unroll
            for (int j = 0; j < (kNF + 1) / 2; ++j) {
              const int ps = __builtin_amdgcn_sdot4(vs[j], wpk, pbase, false);  // p + 32768
              fire |= 64 * j + ln < kHT && static_cast<unsigned>(ps) > 65535u;
            }
            if (__ballot(fire) != 0ull) {
              const int rr = node & 31;
              const int idx = __builtin_amdgcn_readfirstlane((node >> 5) * 16 + (rr & 3) + 4 * (rr >> 3));  // mi * 16 + reg
              const bool mine = (ln >> 5) == ((rr >> 2) & 1);
              int c[kNF];
#pragma unroll
              for (int ni = 0; ni < kNF; ++ni) {
                const int row = 32 * ni + frow;
                const uint32_t pr = *reinterpret_cast<const uint16_t *>(at + row * kBK + (((kl >> 4) ^ swz<kBK>(row)) << 4) + (kl & 15));
                const int a0 = static_cast<int>((pr & 0xff) ^ 0x80), a1 = static_cast<int>((pr >> 8) ^ 0x80);  // back to u8
                const int prod = a0 * w0 + a1 * w1;
                c[ni] = mine ? max(-32768, min(32767, prod)) - prod : 0;
              }
              ppo_add(idx, c);
            }
            ++fix_e;
            fix_raw = fix_raw_nxt;
            fix_k_next = fix_e < fix_end ? static_cast<int>(fix_raw & 0xffff) : INT_MAX;
            if (fix_e + 1 < fix_end) fix_raw_nxt = ent_c[fix_e + 1];
          };
          const bool due = !NOFIX && fix_k_next < (kt + 1) * kBK;
          int sv0 = 0, sv1 = 0, sv2 = 0;
          if (due) {
            sv0 = screen_read(0);
            sv1 = screen_read(1);
            sv2 = screen_read(2);
          }
          __builtin_amdgcn_sched_barrier(0);
          substep(true, true, T, 1, 1, 0);
          if (due) walk_one(sv0, sv1, sv2);
          while (!NOFIX && fix_k_next < (kt + 1) * kBK) walk_one(screen_read(0), screen_read(1), screen_read(2));  // (further entries of the k-step)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kk = 1; kk < 3; ++kk) substep(true, true, T, kk + 1, (kk + 1) & 1, kk & 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage's fragments are all in registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (evalid && kt >= kPub) xslot();
        __builtin_amdgcn_sched_barrier(0);
        if (cvalid) substep(true, kt + 1 < kKT, T + 1, 0, 0, 1);
      }
      PPO_CLK(3 * (ph + 1) + 1);
      // ---- extension ticks: two barriers each (the counters are read between them: nobody adds to one while another wave may
      // still be reading it for the exit decision), my exchange steps behind the second
      if (evalid && !(FDNN_PPO_DEBUG & 16)) {
        for (;;) {
          __builtin_amdgcn_s_barrier();
          const bool all_scaled = __builtin_amdgcn_readfirstlane(xf_s[4]) >= 4 * ph;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (all_scaled || ++ext_iters > kExtBound || sabotage) break;
          xslot();
        }
      }
    } else {
      // ============================================================== SUPPORT role
      const __amdgpu_buffer_rsrc_t rw =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.w + static_cast<size_t>(my_mt * kBM) * p.ldw), 0, kBM * p.ldw, 0x00020000);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<int8_t *>(p.a + static_cast<size_t>(c_pair * kFT + cg * kHT) * p.lda), 0, kHT * p.lda, 0x00020000);
      auto ra_next = [&]() {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.a + static_cast<size_t>(n_pair * kFT + sg * kHT) * p.lda), 0, kHT * p.lda,
                                                 0x00020000);
      };
      const float *bias_l = bias_s + 64 * wm + 4 * (ln >> 5);
      int ldw_s = p.ldw, lda_s = p.lda;
      asm volatile("" : "+s"(ldw_s), "+s"(lda_s));
      const int srow = ln >> 3;
      const int schunk = ((ln & 7) ^ swz<kBK>(wm * 8 + srow)) << 4;
      const int voff_w = srow * ldw_s + schunk;
      const int voff_a = srow * lda_s + schunk;
      auto stage_w = [&](__amdgpu_buffer_rsrc_t r, int chunk, int buf, int i) {
        const int slab = i * 4 + wm;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(&ringW[buf][slab * 1024]), 16, voff_w, slab * 8 * ldw_s + chunk * kBK, 0, 0);
      };
      auto stage_a = [&](__amdgpu_buffer_rsrc_t r, int chunk, int buf, int i) {
        const int slab = i * 4 + wm;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FDNN_LDS_PTR(&ringA[buf][slab * 1024]), 16, voff_a, slab * 8 * lda_s + chunk * kBK, 0, 0);
      };
      float psum = 0.0f;
      const uint32_t pw_a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(FDNN_LDS_PTR(sg_s))) + (wm * kHT + (ln & 31)) * 4;
      float rcp_s = p.rcp_coef, coef_s = p.coef;
      asm volatile("" : "+s"(rcp_s), "+s"(coef_s));
      auto exp_item = [&](int it) {
#if !(FDNN_PPO_DEBUG & 1)
        const int ni = it >> 3, mi = (it >> 2) & 1, g = it & 3;
        const v4f_t b4 = *reinterpret_cast<const v4f_t *>(bias_l + 32 * mi + 8 * g);
        const float keep = my_mt * kBM + 64 * wm + 32 * mi + 8 * g + 4 * (ln >> 5) < p.rows ? 1.0f : 0.0f;
        float xa0, xa1, xa2, xa3, xq0, xq1, xq2, xq3;
        switch (it) {
#define X(IT, OFF, R0, R1, R2, R3) \
  case IT: PPO_EXP4(R0, R1, R2, R3); break;
          PPO_ITEMS(X)
#undef X
          default: break;
        }
        if ((it & 7) == 7) {  // the block's 16 values of this lane are in: + the other half's 16, parked for the publish
          // (the other half's sum by address: __shfl_xor keeps its own copy of the lane id alive across the whole kernel)
          const float tot = psum + __int_as_float(__builtin_amdgcn_ds_bpermute((ln ^ 32) << 2, __float_as_int(psum)));
          if ((ln >> 5) == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pw_a + 32 * ni * 4), "v"(tot) : "memory");
          psum = 0.0f;
        }
#else
        (void)it;
#endif
      };

#pragma unroll
      for (int kt = 0; kt < kKT; ++kt) {
        if (kt >= kt0) {
          const int T = gt0 + kt;
          bool wq = false;
#if !(FDNN_PPO_DEBUG & 4)
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
#endif
          __builtin_amdgcn_sched_barrier(0);
          if (evalid && kt < kPub) {  // my epilogue's arithmetic: kExpPer items a tick (four = one accumulator tile)
#pragma unroll
            for (int it = kExpPer * kt; it < kExpPer * kt + kExpPer && it < 8 * kNF; ++it) {
              exp_item(it);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#if !(FDNN_PPO_DEBUG & 4)
          const bool w_cur = kt < kKT - 2 && cvalid, w_nxt = kt >= kKT - 2 && nvalid;
          wq = w_cur || w_nxt;
          const int w_chunk = kt < kKT - 2 ? kt + 2 : kt - (kKT - 2);
          if (wq) {
#pragma unroll
            for (int i = 0; i < 8; ++i) stage_w(rw, w_chunk, (T + 2) % kWStages, i);  // (the node tile never changes: the next half's weights are this tile's)
          }
#endif
          if (wq) __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8)
          else __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
          asm volatile("" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my partial sums are in LDS
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      PPO_CLK(3 * (ph + 1) + 1);
      // ---- extension ticks: scale and store once the frames' 1 / total are there
      if (evalid && !(FDNN_PPO_DEBUG & 16)) {
        bool scaled = false;
        for (;;) {
          __builtin_amdgcn_s_barrier();
          const bool all_scaled = __builtin_amdgcn_readfirstlane(xf_s[4]) >= 4 * ph;
          const bool inv_ready = __builtin_amdgcn_readfirstlane(xf_s[3]) >= 3 * ph;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (all_scaled || ++ext_iters > kExtBound || sabotage) break;
          if (inv_ready && !scaled) {
#if !(FDNN_PPO_DEBUG & 32)
            // The outputs leave as WHOLE ROW SEGMENTS.  Straight from the accumulators' layout a store instruction is 32 frames
            // x 32 bytes -- 32 different cache lines, a quarter each: 160 such instructions per workgroup and half took 12 000
            // cycles (46 of the kernel's 225 us: measured by leaving them out).  Instead a frame block (32 frames x the wave's 64
            // nodes) is scaled into a park -- row stride 272 bytes -- in the LDS the rings do not need right now (the weight
            // buffer and the row buffer read in tick 15: the next phase's first requests for them come after this tick's last
            // barrier), read back as rows, and every store instruction writes four rows x 256 contiguous bytes.
            // One descriptor for the pair's rows that exist (frames past n fall outside it), one lane offset for all 40 stores,
            // the frames as the scalar offset; nodes past the layer's last (the last node tile only): an offset outside everything.
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            const int f0 = e_pair * kFT;
            const int rows_s = p.rows;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.final + static_cast<size_t>(f0) * rows_s, 0,
                                                                                  max(0, min(kFT, p.n - f0)) * rows_s * 4, 0x00020000);
            const int t_last = gt0 + kKT - 1;  // the tick whose buffers are free now
            char *park = wm < 3 ? &ringW[t_last % kWStages][wm * (32 * kTS)] : &ringA[t_last & 1][0];
            char *park_w = park + (ln & 31) * kTS + (ln >> 5) * 16;            // my frame's row, my half's 16 bytes of every 32
            const char *park_r = park + (ln >> 4) * kTS + (ln & 15) * 16;      // rows 4 j + lane / 16, 16 bytes at (lane % 16) x 16
            const int node_r = my_mt * kBM + 64 * wm + 4 * (ln & 15);
            const int voff = node_r + 4 <= rows_s ? ((sg * kHT + (ln >> 4)) * rows_s + node_r) * 4 : static_cast<int>(0x80000000u);
#pragma clang loop unroll(full)
            for (int ni = 0; ni < kNF; ++ni) {
              const float iv = inv_s[32 * ni + (ln & 31)];
#pragma clang loop unroll(full)
              for (int it = 8 * ni; it < 8 * ni + 8; ++it) {
                const int mi = (it >> 2) & 1, g = it & 3;
                v4f_t o4;
                switch (it) {
#define X(IT, OFF, R0, R1, R2, R3) \
  case IT: PPO_SCL4(R0, R1, R2, R3); break;
                  PPO_ITEMS(X)
#undef X
                  default: break;
                }
                *reinterpret_cast<v4f_t *>(park_w + (32 * mi + 8 * g) * 4) = o4;
              }
#if !(FDNN_PPO_DEBUG & 8)
#pragma clang loop unroll(full)
              for (int j = 0; j < 8; ++j) {
                const v4u v = *reinterpret_cast<const v4u *>(park_r + 4 * j * kTS);
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, voff, (32 * ni + 4 * j) * rows_s * 4, 0);
              }
#endif
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
            scaled = true;
            if (ln == 0) __hip_atomic_fetch_add(&xf_s[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (!scaled && wm == 0 && ln == 0) {  // the siblings never arrived: this half's rows are missing, and it says so
          if (p.fuse_giveups) atomicAdd(p.fuse_giveups, 1ull);
          if (p.fuse_fault) __hip_atomic_fetch_add(p.fuse_fault, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (upper half: halves left UNWRITTEN -- fdnn_calculate compares it around its pass)
        }
      }
    }
    PPO_CLK(3 * (ph + 1) + 2);
#if FDNN_PPO_CLK
    if (wave == 0 && ln == 0 && ph + 1 < 16) clk_ext[ph + 1] = ext_iters;
#endif
  }
#if FDNN_PPO_CLK
  __syncthreads();
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 131 || blockIdx.x == 255)) {
    for (int i = 0; i <= n_ph + 1 && i < 16; ++i)
      printf("PPO wg %3d phase %2d: ticks %6lld ext %6lld (%d extension ticks)%s\n", static_cast<int>(blockIdx.x), i - 1, clk_s[3 * i + 1] - clk_s[3 * i],
             clk_s[3 * i + 2] - clk_s[3 * i + 1], clk_ext[i], i == n_ph + 1 ? "  <- last" : "");
  }
#endif
#undef p
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace

static std::atomic<int> g_ppo_mode{-1};
void qppo_set_mode(int mode) { g_ppo_mode.store(mode, std::memory_order_relaxed); }

// The role-split fused output kernel serves the dense production call of the 8000-node layer (8192 padded rows = 32 node
// tiles: the row sums of a half are one 20 KB block; K = 2048; validated 3-operation division; rows % 4 == 0) on a device
// whose CUs can hold one workgroup per (node tile, frame pair slot): grid = 32 x (CUs / 32).
bool qppo_ok(int rows, int rows_pad, int K, int n, bool fastdiv, bool has_fix) {
  static const int env_mode = [] {
    const char *e = std::getenv("FDNN_PPO");
    return e ? std::atoi(e) : -1;
  }();
  const int forced = g_ppo_mode.load(std::memory_order_relaxed);
  const int mode = forced >= 0 ? forced : env_mode;
  if (mode == 0 || !fastdiv || K != kKT * kBK || rows_pad != kMT * kBM || (rows & 3) != 0) return false;
  if (mode == 1) return true;
  // By default where it was measured ahead of the in-phase fused tiles (tools/ppo_time.py, several boxes; LABBOOK round 6).
  // A launch is ceil(pairs / 8) rounds of frame pairs (8 slots of 32 workgroups on 256 CUs); what matters is how full the
  // last round is and that a workgroup has at least two pairs (a steady state):
  //   a layer without saturating pairs (trained nets): from 14 pairs when the rounds are >= 3/4 full -- 4 480 frames 102 us
  //   against 105, 5 120: 105 / 113, 10 000: 191 .. 212 / 218 .. 226, 20 480: 391 / 444; 3 840 (12 pairs): 100 / 95;
  //   a layer with pairs (the walk runs in a lone compute wave): from 22 pairs when the rounds are >= 4/5 full -- 7 000 frames
  //   170 / 173, 7 680: 171 / 178, 8 320 .. 8 960: 217 / 224 .. 226, 10 000: 223 / 232, 12 000: 272 / 298, 16 000: 382 / 407;
  //   8 000 (25 pairs: four rounds, the last with one pair): 216 / 200; 5 120: 123 / 117.
  const int pairs = (n + kFT - 1) / kFT, rounds = (pairs + 7) / 8;
  return has_fix ? pairs >= 22 && 5 * pairs >= 4 * 8 * rounds : pairs >= 14 && 4 * pairs >= 3 * 8 * rounds;
}

int qppo_frame_tile() { return kFT; }

void launch_qppo_output(const QGemmParams &p, hipStream_t s) {
  auto k = qppo_kernel<false>;
  auto k_nofix = qppo_kernel<true>;
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n_cu = cus[dev & 63].load(std::memory_order_relaxed);
  if (n_cu == 0) {
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    cus[dev & 63].store(n_cu, std::memory_order_relaxed);
  }
  const int NP = p.n_pad / kFT;
  const int slots = std::max(1, std::min(n_cu / kMT, NP));  // frame pairs in flight: every one has all its 32 node tiles resident
  hipLaunchKernelGGL(p.fix_ent ? k : k_nofix, dim3(slots * kMT), dim3(512), 0, s, p);
}

}  // namespace fdnn
